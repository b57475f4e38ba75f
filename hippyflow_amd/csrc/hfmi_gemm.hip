// FP64 tall-skinny contractions on gfx950 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Two shapes carry every operator application, Gram matrix and back-transform on the
// double-pass path (DESIGN.md section "kernels"):
//
//   tsgemm_tn :  C (m x k)  = A^T B     A: N x m, B: N x k   -- reduction over the long axis N
//                (X W in the snapshot-Gram / Jacobian-Gram apply, Q^T B Q, (AQ)^T Q, C W)
//   tsgemm_nn :  Y (N x r)  = A S       A: N x m, S: m x r   -- long axis preserved
//                (X^T G, Q R^{-1}, U = Q V)
//
// All blocks are column-major with every vector contiguous along N, so in tsgemm_tn BOTH
// operands are contiguous along the reduction index: an MFMA k-step may use any 4 reduction
// indices as long as A and B agree, which lets each lane fetch 16 contiguous bytes (2 doubles)
// per operand per two k-steps straight from HBM -- no transposition through LDS.  The operand
// shared by the 4 waves of a workgroup (B in tn, S in nn) is staged through LDS; the streamed
// operand goes HBM -> VGPR directly with a register prefetch ring.  One wave per SIMD holds a
// (16*MT) x (16*NT) fp64 accumulator tile (up to 40 MFMA tiles = 320 VGPRs).
//
// MFMA fragment maps (cdna_hip_programming.md section 3): lane l supplies A[i = l&15][kk = l>>4] and
// B[kk = l>>4][j = l&15]; it receives D[row = (l>>4) + 4*reg][col = l&15], reg = 0..3.
#include "hfmi_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define MFMA_F64_4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// XCD-aware block id remap (8 XCDs, block b runs on XCD b % 8): consecutive logical ids land on the
// same XCD so that workgroups sharing the LDS-staged operand also share an L2.  Bijective for any total.
__device__ __forceinline__ int64_t round_up_dev(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

__device__ __forceinline__ int xcd_remap(int lin, int total) {
  const int xcd = lin & 7, slot = lin >> 3;
  const int fl = total >> 3, rem = total & 7;
  return xcd * fl + (xcd < rem ? xcd : rem) + slot;
}

// =====================================================================================
// tsgemm_tn
// =====================================================================================
constexpr int TN_BK = 32;   // reduction indices per LDS stage
constexpr int TN_LDB = 34;  // LDS leading dimension (doubles): 16-byte aligned rows, spreads banks

// Preconditions (guaranteed by the block layout, hfmi.h): lda, ldb multiples of 32 doubles, rows N..ld-1 of
// every vector are zero, so the reduction runs over whole 32-row stages with NO masks or branches in the loop:
// every load below is unconditional (clamped addresses), which is what lets the prefetches stay in flight.
//
// WAVES = 4: one wave per SIMD with up to 32 accumulator tiles (256 AGPRs).  WAVES = 8: two waves per SIMD with
// up to 16 tiles each -- same workgroup tile, but while one wave of a SIMD sits in a barrier / LDS / HBM wait the
// other keeps the matrix pipe busy.
// R4 > 0: the LAST of the NT column tiles holds at most 4 R4 real columns and is computed with R4
// v_mfma_f64_4x4x4_4b instructions (16 cycles each, four 4x4x4 blocks = the four 4-row groups of a 16-row tile)
// instead of one 16x16x4 (64 cycles): k = 84 costs 5 x 64 + 16 cycles per k-step instead of 6 x 64, k = 74
// 4 x 64 + 48 instead of 5 x 64.  Lane layout of the 4x4x4 instruction (scripts/mfma4x4_probe.hip): A lane
// 16 k + 4 g + i = A_g[i][k], B lane 16 k + 4 g + j = B_g[k][j], D lane 16 i + 4 g + j = D_g[i][j] -- so the 16x16x4
// A fragment (lane (row & 15, k)) already IS the 4x4x4 A operand with block g = rows 4g..4g+3, and B only has to be
// read from LDS with the 4 columns of a group repeated for every block.
template <int MT, int NT, bool TR, int WAVES, int R4>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 8 || MT * NT <= 20) ? 2 : 1) void k_tsgemm_tn(const double* __restrict__ A, int64_t lda, int m,
                                                                    const double* __restrict__ B, int64_t ldb, int k,
                                                                    int64_t Npad, int64_t chunk, int nrb, int nsplit,
                                                                    double* __restrict__ part, int mpad, int kpad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = TN_BK;
  constexpr int COLS = NT * 16;
  constexpr int LDB = BK + 2;
  constexpr int BUFD = COLS * LDB;                // doubles per stage buffer
  constexpr int CSH = 4;                          // log2(16-byte chunks per column per stage)
  constexpr int NTF = R4 > 0 ? NT - 1 : NT;       // full 16-column tiles
  constexpr int NR4 = R4 > 0 ? R4 : 1;
  double* lds = reinterpret_cast<double*>(smem);  // [2][NT*16][LDB]
  constexpr int NTHR = WAVES * 64;
  constexpr int CH = COLS * (BK / 2);             // 16-byte chunks per stage
  constexpr int NQ = (CH + NTHR - 1) / NTHR;      // chunks per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;

  const int logical = xcd_remap(blockIdx.x, nrb * nsplit);
  const int sp = logical / nrb, rb = logical % nrb;
  const int64_t t_begin = (int64_t)sp * chunk;
  int64_t t_end = t_begin + chunk;
  if (t_end > Npad) t_end = Npad;
  const int nstages = (int)((t_end - t_begin) / BK);
  const int64_t t_last = t_end - 8;  // last iteration base that is safe to fetch
  const int rowbase = rb * (16 * MT * WAVES) + wave * (16 * MT);

  const double* a_ptr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = rowbase + mt * 16 + r16;
    if (row > m - 1) row = m - 1;
    a_ptr[mt] = A + (int64_t)row * lda + kk * 2;
  }
  const double* b_ptr[NQ];
#pragma unroll
  for (int qd = 0; qd < NQ; ++qd) {
    int c = tid + NTHR * qd;
    if (c > CH - 1) c = CH - 1;
    int col = c >> CSH;
    if (col > k - 1) col = k - 1;
    b_ptr[qd] = B + (int64_t)col * ldb + (c & ((1 << CSH) - 1)) * 2;
  }

  constexpr int NTA = NTF > 0 ? NTF : 1;
  d4 acc[MT][NTA];
  double acc4[MT][NR4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NTA; ++nt) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NR4; ++q) acc4[mt][q] = 0.0;
  }

  d2 breg[NQ];
  auto stage_load = [&](int64_t ts) {
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) breg[qd] = *reinterpret_cast<const d2*>(b_ptr[qd] + ts);
  };
  auto stage_store = [&](double* L) {
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
      const int c = tid + NTHR * qd;
      const int col = c >> CSH, q = c & ((1 << CSH) - 1);
      const int off = col * LDB + q * 2;
      if (CH % NTHR == 0 || c < CH) *reinterpret_cast<d2*>(L + off) = breg[qd];
    }
  };
  auto load_a = [&](d2(&dst)[MT], int64_t t) {
    if (t > t_last) t = t_last;  // wave-uniform clamp: prefetches past the end re-read valid data
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) dst[mt] = *reinterpret_cast<const d2*>(a_ptr[mt] + t);
  };
  // Full tiles: lane (r16, kk) reads column nt*16 + r16.  The fragments are SINGLE-buffered: the moment the MFMAs
  // of column tile nt have been issued, its registers are refilled with the next iteration's fragment, and the MFMAs of
  // the other column tiles cover the LDS latency (the sched_barriers pin that order; left alone the scheduler hoists
  // all reads to the top, which is the double-buffered register budget again -- 4 NT VGPRs more).  The 4-column groups
  // of the last tile (lane -> column NTF*16 + 4 q + (lane & 3), the same for every row group) are read at the top of
  // their own iteration; the full-tile MFMAs in front of them cover the latency.
  d2 bf[NTA];
  auto ldsb1 = [&](const double* L, int it, int nt) {
    bf[nt] = *reinterpret_cast<const d2*>(L + (nt * 16 + r16) * LDB + it * 8 + kk * 2);
  };
  auto mma = [&](const d2(&a)[MT], const double* L, int it, bool refill) {
    d2 bg[NR4];
    if constexpr (R4 > 0) {
#pragma unroll
      for (int q = 0; q < R4; ++q)
        bg[q] = *reinterpret_cast<const d2*>(L + (NTF * 16 + 4 * q + (lane & 3)) * LDB + it * 8 + kk * 2);
    }
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = TR ? MFMA_F64(bf[nt].x, a[mt].x, acc[mt][nt]) : MFMA_F64(a[mt].x, bf[nt].x, acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = TR ? MFMA_F64(bf[nt].y, a[mt].y, acc[mt][nt]) : MFMA_F64(a[mt].y, bf[nt].y, acc[mt][nt]);
      if (refill) ldsb1(L, it + 1, nt);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (R4 > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < R4; ++q) acc4[mt][q] = MFMA_F64_4(a[mt].x, bg[q].x, acc4[mt][q]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < R4; ++q) acc4[mt][q] = MFMA_F64_4(a[mt].y, bg[q].y, acc4[mt][q]);
    }
    // keep the scheduler from hoisting later iterations' loads above these MFMAs
    // (it would otherwise blow the register budget and spill the accumulators)
    __builtin_amdgcn_sched_barrier(0);
  };

  if (nstages > 0) {
    stage_load(t_begin);
    stage_store(lds);
    __syncthreads();
  }
  // streamed operand: register ring over iterations of 8 reduction indices
  {
    constexpr int NIT = BK / 8;
    d2 a[2][MT];
    load_a(a[0], t_begin);
    for (int s = 0; s < nstages; ++s) {
      const int64_t ts = t_begin + (int64_t)s * BK;
      const bool has_next = s + 1 < nstages;
      if (has_next) stage_load(ts + BK);
      const double* L = lds + (s & 1) * BUFD;
#pragma unroll
      for (int nt = 0; nt < NTF; ++nt) ldsb1(L, 0, nt);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        load_a(a[(it + 1) & 1], ts + 8 * (it + 1));
        mma(a[it & 1], L, it, it + 1 < NIT);
      }
      if (has_next) stage_store(lds + ((s + 1) & 1) * BUFD);
      __syncthreads();
    }
  }

  double* P = part + (int64_t)sp * mpad * kpad;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (TR) {
          const int j = nt * 16 + kk + 4 * r;
          const int i = rowbase + mt * 16 + r16;
          P[(int64_t)j * mpad + i] = acc[mt][nt][r];
        } else {
          const int i = rowbase + mt * 16 + kk + 4 * r;
          const int j = nt * 16 + r16;
          P[(int64_t)i * kpad + j] = acc[mt][nt][r];
        }
      }
  if constexpr (R4 > 0) {
    // 4x4x4 results: lane 16 i + 4 g + j holds (row 4 g + i of the tile, column 4 q + j of the last tile)
    const int i4 = lane >> 4, g4 = (lane >> 2) & 3, j4 = lane & 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < R4; ++q) {
        const int i = rowbase + mt * 16 + 4 * g4 + i4;
        const int j = NTF * 16 + 4 * q + j4;
        if (TR) P[(int64_t)j * mpad + i] = acc4[mt][q];
        else P[(int64_t)i * kpad + j] = acc4[mt][q];
      }
  }
}

// C[i*rs + j*cs] = scale * sum_sp part[sp][...] + beta * C.  A workgroup is 64 outputs x RY split lanes: lane y sums
// the slices y, y+RY, y+2RY, ... (four independent running sums, so four loads are in flight), the RY lane totals
// are combined through LDS in a fixed order -> deterministic, and the sum over hundreds of slices is no longer
// one serial chain per output element (that chain used to cost more than the contraction itself for k <= 138).
// RY = 16 for the many-slice launches of tsgemm_ss, 4 when there are only a few slices (big tn launches: every
// lane then has work and the kernel is a plain HBM stream).
template <int RY>
__global__ __launch_bounds__(64 * RY) void k_reduce_partials(const double* __restrict__ part, int nsplit, int64_t pstride,
                                                             int inner_ld, int tr, int m, int k, double scale, double beta,
                                                             double* __restrict__ C, int64_t rs, int64_t cs) {
  __shared__ double sh[RY][64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int fast = blockIdx.x * 64 + tx;
  const int fastn = tr ? m : k, slown = tr ? k : m;
  for (int slow = blockIdx.y; slow < slown; slow += gridDim.y) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (fast < fastn) {
      const double* p = part + (int64_t)slow * inner_ld + fast;
      int sp = ty;
      for (; sp + 3 * RY < nsplit; sp += 4 * RY) {
        s0 += p[(int64_t)sp * pstride];
        s1 += p[(int64_t)(sp + RY) * pstride];
        s2 += p[(int64_t)(sp + 2 * RY) * pstride];
        s3 += p[(int64_t)(sp + 3 * RY) * pstride];
      }
      for (; sp < nsplit; sp += RY) s0 += p[(int64_t)sp * pstride];
    }
    sh[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && fast < fastn) {
      double s = 0.0;
#pragma unroll
      for (int y = 0; y < RY; ++y) s += sh[y][tx];
      const int i = tr ? fast : slow, j = tr ? slow : fast;
      double* out = C + (int64_t)i * rs + (int64_t)j * cs;
      *out = (beta != 0.0) ? scale * s + beta * (*out) : scale * s;
    }
    __syncthreads();
  }
}

// Flat variant for a row-major result whose row stride equals the partial's (C = G of an operator application,
// m x kpad): the m * kpad entries are one contiguous array per slice, so every lane streams (no 74-of-128 lane waste
// on a narrow fast axis); pad columns j >= k are written as zeros (tsgemm_nn reads G as a zero-padded small matrix).
template <int RY>
__global__ __launch_bounds__(64 * RY) void k_reduce_flat(const double* __restrict__ part, int nsplit, int64_t pstride,
                                                         int64_t total, int kpad, int k, double scale, double beta,
                                                         double* __restrict__ C) {
  __shared__ double sh[RY][64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int64_t idx = (int64_t)blockIdx.x * 64 + tx;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (idx < total) {
    const double* p = part + idx;
    int sp = ty;
    for (; sp + 3 * RY < nsplit; sp += 4 * RY) {
      s0 += p[(int64_t)sp * pstride];
      s1 += p[(int64_t)(sp + RY) * pstride];
      s2 += p[(int64_t)(sp + 2 * RY) * pstride];
      s3 += p[(int64_t)(sp + 3 * RY) * pstride];
    }
    for (; sp < nsplit; sp += RY) s0 += p[(int64_t)sp * pstride];
  }
  sh[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && idx < total) {
    double s = 0.0;
#pragma unroll
    for (int y = 0; y < RY; ++y) s += sh[y][tx];
    const bool real = (int)(idx % kpad) < k;
    C[idx] = real ? ((beta != 0.0) ? scale * s + beta * C[idx] : scale * s) : 0.0;
  }
}

int launch_reduce_partials(hfmi_ctx* ctx, const double* part, int nsplit, int64_t pstride, int inner_ld, bool tr, int m,
                           int k, double scale, double beta, double* C, int64_t rs, int64_t cs) {
  const int fastn = tr ? m : k, slown = tr ? k : m;
  if (!tr && cs == 1 && rs == inner_ld && (int64_t)m * inner_ld >= 65536) {
    const int64_t total = (int64_t)m * inner_ld;
    dim3 fgrid((unsigned)((total + 63) / 64));
    if (nsplit <= 32)
      hipLaunchKernelGGL(k_reduce_flat<4>, fgrid, dim3(64, 4), 0, ctx->stream, part, nsplit, pstride, total, inner_ld, k, scale,
                         beta, C);
    else
      hipLaunchKernelGGL(k_reduce_flat<16>, fgrid, dim3(64, 16), 0, ctx->stream, part, nsplit, pstride, total, inner_ld, k,
                         scale, beta, C);
    HIP_TRY(hipGetLastError());
    return HFMI_OK;
  }
  dim3 grid((fastn + 63) / 64, slown < 32768 ? slown : 32768);
  if (nsplit <= 32)
    hipLaunchKernelGGL(k_reduce_partials<4>, grid, dim3(64, 4), 0, ctx->stream, part, nsplit, pstride, inner_ld, tr ? 1 : 0, m,
                       k, scale, beta, C, rs, cs);
  else
    hipLaunchKernelGGL(k_reduce_partials<16>, grid, dim3(64, 16), 0, ctx->stream, part, nsplit, pstride, inner_ld, tr ? 1 : 0, m,
                       k, scale, beta, C, rs, cs);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

#include <stdlib.h>
#include <string.h>
// Tuning knobs of the MFMA kernels (A/B measurements: environment HFMI_GEMM_WAVES, or
// hfmi_tuning_set at run time).
//   waves: 8 = two waves per SIMD with <= 16 accumulator tiles each; 4 = one wave per SIMD with <= 32 tiles
//   rem4 : compute a last column tile of <= 12 columns with 4x4x4 MFMAs (1, default) or as a full 16-column tile (0)
static int g_waves = 0, g_rem4 = 1, g_nn_waves = 0;  // g_nn_waves: 0 = auto (4 for <= 9 column tiles, else 8)
static int g_nn_hybrid = 1;                          // split only the row tiles beyond the last full round of CUs
static int g_nn_tt = 0;                              // A/B: force the nn wave-tile height (1 = tallest, 2, 3 = next smaller)
static int g_ss = 1;                                 // route skinny x skinny contractions to tsgemm_ss (hfmi_skinny.hip)
static void tuning_init() {
  if (g_waves) return;
  const char* e = getenv("HFMI_GEMM_WAVES");
  g_waves = (e && atoi(e) == 4) ? 4 : (e && atoi(e) == 44) ? 44 : 8;   // measured: 8 waves is best or equal on every shape (scripts/gemm_ab.py)
}
static int gemm_waves() {
  tuning_init();
  return g_waves;
}
extern "C" int hfmi_tuning_set(const char* key, int value) {
  tuning_init();
  if (key && !strcmp(key, "waves") && (value == 4 || value == 8 || value == 44)) g_waves = value;
  else if (key && !strcmp(key, "rem4") && (value == 0 || value == 1)) g_rem4 = value;
  else if (key && !strcmp(key, "nn_waves") && (value == 0 || value == 4 || value == 8)) g_nn_waves = value;
  else if (key && !strcmp(key, "ss") && (value == 0 || value == 1)) g_ss = value;
  else if (key && !strcmp(key, "nn_tt") && value >= 0 && value <= 3) g_nn_tt = value;
  else if (key && !strcmp(key, "nn_hybrid") && (value == 0 || value == 1)) g_nn_hybrid = value;
  else if (key && !strcmp(key, "ss_percu") && value >= 1 && value <= 4) tsgemm_ss_set_percu(value);
  else HFMI_FAIL(HFMI_ERR_INVALID, "tuning_set: unknown key/value");
  return HFMI_OK;
}
// tallest wave tile (in 16-row MFMA tiles) for a panel of nt column tiles
static inline int tn_mt_max(int nt, int waves) {
  static const int t4[17] = {0, 8, 8, 8, 8, 6, 5, 4, 4, 3, 3, 2, 2, 2, 2, 2, 2};
  static const int t8[17] = {0, 5, 5, 5, 4, 3, 3, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1};   // A/B (r01e): <3,6> and <2,9> beat <2,6> / <1,9>; <4,5> does not beat <3,5>
  return waves == 4 ? t4[nt] : t8[nt];
}

template <int MT, int NT, int WAVES, bool TR, int R4>
static int tn_launch_one(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* B, int64_t ldb, int k, int64_t N,
                         int64_t chunk, int nrb, int nsplit, double* part, int mpad, int kpad) {
  const size_t shmem = (size_t)2 * NT * 16 * (TN_BK + 2) * sizeof(double);
  auto kern = k_tsgemm_tn<MT, NT, TR, WAVES, R4>;
  HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kern, dim3(nrb * nsplit), dim3(WAVES * 64), shmem, ctx->stream, A, lda, m, B, ldb, k, N, chunk, nrb,
                     nsplit, part, mpad, kpad);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// r4: number of 4-column groups the last column tile is computed in (0 = as a full 16-column tile)
template <int MT, int NT, int WAVES>
static int tn_launch_inst(hfmi_ctx* ctx, bool tr, int r4, const double* A, int64_t lda, int m, const double* B, int64_t ldb,
                          int k, int64_t N, int64_t chunk, int nrb, int nsplit, double* part, int mpad, int kpad) {
#define TN_R4(R)                                                                                                          \
  case R:                                                                                                                 \
    if (tr) return tn_launch_one<MT, NT, WAVES, true, R>(ctx, A, lda, m, B, ldb, k, N, chunk, nrb, nsplit, part, mpad, kpad); \
    return tn_launch_one<MT, NT, WAVES, false, R>(ctx, A, lda, m, B, ldb, k, N, chunk, nrb, nsplit, part, mpad, kpad);
  if constexpr (WAVES == 8) {
    switch (r4) { TN_R4(1) TN_R4(2) TN_R4(3) }
  }
  switch (0) { TN_R4(0) }
#undef TN_R4
  return HFMI_OK;
}

template <int NT, int WAVES>
static int tn_dispatch_mt(hfmi_ctx* ctx, int mt, bool tr, int r4, const double* A, int64_t lda, int m, const double* B,
                          int64_t ldb, int k, int64_t N, int64_t chunk, int nrb, int nsplit, double* part, int mpad,
                          int kpad) {
  constexpr int LIM = (WAVES == 8) ? 20 : 32;
#define TN_CASE(M)                                                                                                 \
  case M:                                                                                                          \
    if constexpr (M * NT <= LIM)                                                                                   \
      return tn_launch_inst<M, NT, WAVES>(ctx, tr, r4, A, lda, m, B, ldb, k, N, chunk, nrb, nsplit, part, mpad, kpad); \
    break;
  switch (mt) {
    TN_CASE(1) TN_CASE(2) TN_CASE(3) TN_CASE(4) TN_CASE(5) TN_CASE(6) TN_CASE(8)
  }
#undef TN_CASE
  HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_tn: no instance for MT=%d NT=%d WAVES=%d", mt, NT, WAVES);
}

// one panel of at most 256 columns of B
static int tn_panel(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* B, int64_t ldb, int k,
                    int64_t N, double scale, double beta, double* C, int64_t rs, int64_t cs, int nsplit_req) {
  const int nt = (k + 15) / 16;
  const int kpad = nt * 16;
  const int64_t Npad = round_up(N, TN_BK);
  if (lda % 32 != 0 || ldb % 32 != 0 || lda < Npad || ldb < Npad)
    HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_tn: leading dimensions must be multiples of 32 and >= round_up(N,32)");
  const bool tr = (rs == 1 && cs != 1);  // column-major output: coalesce along i
  // wave tile height: as tall as the accumulator budget allows, but no taller than the problem needs
  // waves: 8 = one 8-wave workgroup per CU (two waves per SIMD); 4 = one 4-wave workgroup with big tiles;
  // 44 = 4-wave workgroups with the small (two-per-SIMD) tiles, TWO workgroups per CU: their stage barriers are not
  // synchronised with each other, so one workgroup's MFMAs cover the other's barrier / staging bubbles
  const int wcfg = (nt > 11) ? 4 : gemm_waves();   // very wide panels: the 2-waves/SIMD register budget is too tight
  const bool small4 = wcfg == 44;
  const int waves = small4 ? 4 : wcfg;
  const int row_tiles = (m + 15) / 16;
  int mt = tn_mt_max(nt, small4 ? 8 : waves);
  const int need = (row_tiles + waves - 1) / waves;
  if (need < mt) mt = need;
  if (mt == 7) mt = 6;
  if (mt < 1) mt = 1;
  const int rows_per_block = 16 * waves * mt;
  const int nrb = (m + rows_per_block - 1) / rows_per_block;
  const int mpad = nrb * rows_per_block;
  // split the long axis so that the grid fills the chip in (nearly) whole rounds of CUs
  const int cus = (ctx->num_cus > 0 ? ctx->num_cus : 256) * (small4 ? 2 : 1);   // resident workgroup slots
  int nsplit = nsplit_req;
  if (nsplit <= 0) {
    const int64_t stages = Npad / TN_BK;
    int best = 1;
    double best_cost = 1e300;
    for (int ns = 1; ns <= 128; ++ns) {
      if (ns > 1 && stages / ns < 16) break;
      const int64_t blocks = (int64_t)nrb * ns;
      const int64_t rounds = (blocks + cus - 1) / cus;
      const double eff = (double)blocks / (double)(rounds * cus);
      const double part_ratio = 2.0 * ns * (double)mpad * kpad / ((double)N * (m + k));
      const double cost = 1.0 / eff + part_ratio;
      if (cost < best_cost - 1e-12) {
        best_cost = cost;
        best = ns;
      }
    }
    nsplit = best;
  }
  const int64_t stage_len = TN_BK;
  int64_t chunk = round_up((Npad + nsplit - 1) / nsplit, stage_len);
  if (chunk < stage_len) chunk = stage_len;
  nsplit = (int)((Npad + chunk - 1) / chunk);
  if (nsplit < 1) nsplit = 1;
  void* partv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)nsplit * mpad * kpad * sizeof(double), &partv));
  double* part = (double*)partv;
  // algorithmic work of this launch: flops 2 N m k, bytes 8 (N m + N k + m k)
  // columns of the last tile: up to 12 are done as 1..3 groups of 4 with the 4x4x4 MFMA (16 instead of 64 cycles each)
  const int rem = k - (nt - 1) * 16;
  const int r4 = (g_rem4 && waves == 8 && rem <= 12) ? (rem + 3) / 4 : 0;
  const int pidx = prof_start(ctx, 0, m, k, N);
#define TN_NT(NTV)                                                                                                       \
  case NTV:                                                                                                              \
    if (waves == 8)                                                                                                      \
      HFMI_TRY((tn_dispatch_mt<NTV, 8>(ctx, mt, tr, r4, A, lda, m, B, ldb, k, Npad, chunk, nrb, nsplit, part, mpad, kpad))); \
    else                                                                                                                 \
      HFMI_TRY((tn_dispatch_mt<NTV, 4>(ctx, mt, tr, 0, A, lda, m, B, ldb, k, Npad, chunk, nrb, nsplit, part, mpad, kpad))); \
    break;
  switch (nt) {
    TN_NT(1) TN_NT(2) TN_NT(3) TN_NT(4) TN_NT(5) TN_NT(6) TN_NT(7) TN_NT(8) TN_NT(9) TN_NT(10) TN_NT(11) TN_NT(12)
    TN_NT(13) TN_NT(14) TN_NT(15) TN_NT(16)
    default:
      HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_tn: panel too wide (%d)", k);
  }
#undef TN_NT
  prof_stop(ctx, pidx);
  return launch_reduce_partials(ctx, part, nsplit, (int64_t)mpad * kpad, tr ? mpad : kpad, tr, m, k, scale, beta, C, rs,
                                cs);
}

int launch_tsgemm_tn(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* B, int64_t ldb, int k,
                     int64_t N, double scale, double beta, double* C, int64_t rs, int64_t cs, int nsplit_req) {
  if (m <= 0 || k <= 0 || N <= 0) return HFMI_OK;
  if (g_ss && tsgemm_ss_applicable(m, k, A == B && lda == ldb && m == k))
    return launch_tsgemm_ss(ctx, A, lda, m, B, ldb, k, N, scale, beta, C, rs, cs, nsplit_req);
  for (int k0 = 0; k0 < k; k0 += 256) {
    const int kp = (k - k0 < 256) ? (k - k0) : 256;
    HFMI_TRY(tn_panel(ctx, A, lda, m, B + (int64_t)k0 * ldb, ldb, kp, N, scale, beta, C + (int64_t)k0 * cs, rs, cs,
                      nsplit_req));
  }
  return HFMI_OK;
}

// =====================================================================================
// tsgemm_nn
// =====================================================================================
constexpr int NN_KC = 32;  // reduction indices per LDS stage (8 MFMA k-steps)

// Preconditions: lda multiple of 32 and >= round_up(N,32) (rows beyond N readable); S finite, ld even.
// The reduction axis m may be split over gridDim-many workgroups (msplit > 1): each split writes a raw partial
// block and k_reduce_nn adds them in a fixed order -- this is what balances the grid over the 256 CUs when
// there are only a few row tiles (quantisation), at the price of msplit * N * r * 16 bytes of extra traffic.
// R4 > 0: the last column tile has at most 4 R4 real columns and is computed in R4 groups of 4 columns with the
// 4x4x4 MFMA (see k_tsgemm_tn): the streamed fragment (lane (c16, kk) = long-axis row c16, reduction index kk) is its B
// operand with block g = rows 4g..4g+3, the small-matrix fragment (column 4 q + (lane & 3), the same for every block)
// its A operand, and lane 16 i + 4 g + j receives Y[row 4 g + j = c16][column 4 q + i = 4 q + kk].
template <int TT, int NT, int WAVES, int R4>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_tsgemm_nn(const double* __restrict__ A, int64_t lda, int m,
                                                                    const double* __restrict__ S, int lds_, int r,
                                                                    double* __restrict__ Y, int64_t ldy, int64_t N,
                                                                    int ntiles, int msplit, int mchunk, int64_t pstride,
                                                                    int full_tiles, double* __restrict__ Yfull,
                                                                    int64_t ldfull) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);  // [2][NN_KC][SLD]
  constexpr int COLS = NT * 16;
  constexpr int SLD = COLS + ((NT % 2 == 0) ? 16 : 0);  // SLD % 32 == 16: conflict-free ds_read_b64
  constexpr int TP = TT / 2;                            // tile pairs fed by one 16-byte load per lane
  constexpr bool ODD = (TT & 1) != 0;                   // plus one single tile fed by an 8-byte load
  constexpr int TPA = TP > 0 ? TP : 1;
  constexpr int NTHR = WAVES * 64;
  constexpr int CH = NN_KC * COLS / 2;            // 16-byte pairs per stage
  constexpr int NQ = (CH + NTHR - 1) / NTHR;      // pairs per thread
  constexpr int NTF = R4 > 0 ? NT - 1 : NT;       // full 16-column tiles
  constexpr int NTA = NTF > 0 ? NTF : 1;
  constexpr int NR4 = R4 > 0 ? R4 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, kk = lane >> 4;
  // Row tiles [0, full_tiles) are whole rounds of the CUs: one workgroup each over the full reduction range, written
  // straight to the result.  The remaining (fewer than one round of) tiles are split msplit-ways over the reduction
  // axis so that they, too, occupy every CU for 1/msplit of a tile's time; only those rows go through partials.
  const int tail_tiles = ntiles - full_tiles;
  // whole tiles and tail pieces are remapped over the XCDs separately: one contiguous logical range per XCD would
  // put all the short tail pieces on the last XCDs and leave the whole tiles to the others (full_tiles % 8 == 0)
  const bool whole = (int)blockIdx.x < full_tiles;
  const int logical = whole ? xcd_remap(blockIdx.x, full_tiles)
                            : full_tiles + xcd_remap((int)blockIdx.x - full_tiles, tail_tiles * msplit);
  const int split = whole ? 0 : (logical - full_tiles) / tail_tiles;
  const int tile = whole ? logical : full_tiles + (logical - full_tiles) % tail_tiles;
  const int64_t t0 = (int64_t)tile * (16 * TT * WAVES) + wave * (16 * TT);
  const int i_begin = whole ? 0 : split * mchunk;
  int i_end = whole ? m : i_begin + mchunk;
  if (i_end > m) i_end = m;
  const int nstages = (i_end - i_begin + NN_KC - 1) / NN_KC;
  const int64_t tmax = round_up_dev(N, 32) - 2;
  double* Yo = whole ? Yfull : Y + (int64_t)split * pstride;
  const int64_t ldout = whole ? ldfull : ldy;

  // streamed operand: lane (c16, kk) fetches rows t0 + tp*32 + 2*c16 + {0,1} of vector i0 + kk
  int64_t toff[TPA];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    int64_t t = t0 + tp * 32 + 2 * c16;
    toff[tp] = t > tmax ? tmax : t;  // rows >= N are never stored; keep the address legal
  }
  int64_t toff1 = t0 + TP * 32 + c16;
  if (toff1 > tmax + 1) toff1 = tmax + 1;
  // S stage: NN_KC rows x COLS cols as 16-byte pairs, NQ per thread
  int s_row[NQ], s_cp[NQ];
#pragma unroll
  for (int qd = 0; qd < NQ; ++qd) {
    int c = tid + NTHR * qd;
    if (c > CH - 1) c = CH - 1;
    s_row[qd] = c / (COLS / 2);
    s_cp[qd] = c % (COLS / 2);
  }

  d4 acc[TT][NTA];
  double acc4[TT][NR4];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
    for (int nt = 0; nt < NTA; ++nt) acc[tt][nt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NR4; ++q) acc4[tt][q] = 0.0;
  }

  struct AFrag {
    d2 p[TPA];
    double s;
  };
  d2 sreg[NQ];
  auto stage_load = [&](int is) {
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
      int row = is + s_row[qd];
      if (row > m - 1) row = m - 1;
      sreg[qd] = *reinterpret_cast<const d2*>(S + (int64_t)row * lds_ + s_cp[qd] * 2);
    }
  };
  auto stage_store = [&](double* L, int is) {
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
      d2 v = sreg[qd];
      if (is + s_row[qd] >= i_end) v = d2{0.0, 0.0};  // rows past this split's range contribute nothing
      if (CH % NTHR == 0 || tid + NTHR * qd < CH) *reinterpret_cast<d2*>(L + s_row[qd] * SLD + s_cp[qd] * 2) = v;
    }
  };
  auto load_a = [&](AFrag& dst, int i0) {
    int col = i0 + kk;
    if (col > m - 1) col = m - 1;
    const double* p = A + (int64_t)col * lda;
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) dst.p[tp] = *reinterpret_cast<const d2*>(p + toff[tp]);
    if (ODD) dst.s = p[toff1];
  };
  struct SFrag {
    double f[NTA];   // full tiles: column nt*16 + c16
    double g[NR4];   // 4-column groups of the last tile: column NTF*16 + 4 q + (lane & 3)
  };
  auto ldss = [&](SFrag& sf, const double* L, int ks) {
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt) sf.f[nt] = L[(ks * 4 + kk) * SLD + nt * 16 + c16];
    if constexpr (R4 > 0) {
#pragma unroll
      for (int q = 0; q < R4; ++q) sf.g[q] = L[(ks * 4 + kk) * SLD + NTF * 16 + 4 * q + (lane & 3)];
    }
  };
  auto mma = [&](const AFrag& a, const SFrag& sf) {
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
#pragma unroll
      for (int nt = 0; nt < NTF; ++nt) {
        acc[2 * tp][nt] = MFMA_F64(sf.f[nt], a.p[tp].x, acc[2 * tp][nt]);
        acc[2 * tp + 1][nt] = MFMA_F64(sf.f[nt], a.p[tp].y, acc[2 * tp + 1][nt]);
      }
      if constexpr (R4 > 0) {
#pragma unroll
        for (int q = 0; q < R4; ++q) {
          acc4[2 * tp][q] = MFMA_F64_4(sf.g[q], a.p[tp].x, acc4[2 * tp][q]);
          acc4[2 * tp + 1][q] = MFMA_F64_4(sf.g[q], a.p[tp].y, acc4[2 * tp + 1][q]);
        }
      }
    }
    if (ODD) {
#pragma unroll
      for (int nt = 0; nt < NTF; ++nt) acc[TT - 1][nt] = MFMA_F64(sf.f[nt], a.s, acc[TT - 1][nt]);
      if constexpr (R4 > 0) {
#pragma unroll
        for (int q = 0; q < R4; ++q) acc4[TT - 1][q] = MFMA_F64_4(sf.g[q], a.s, acc4[TT - 1][q]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  stage_load(i_begin);
  stage_store(lds, i_begin);
  __syncthreads();
  // register ring of 4 k-steps for the streamed operand (prefetch distance 3), ping-pong LDS fragments
  AFrag a0, a1, a2, a3;
  SFrag sf0, sf1;
  load_a(a0, i_begin);
  load_a(a1, i_begin + 4);
  load_a(a2, i_begin + 8);
  for (int s = 0; s < nstages; ++s) {
    const int is = i_begin + s * NN_KC;
    const bool has_next = s + 1 < nstages;
    if (has_next) stage_load(is + NN_KC);
    const double* L = lds + (s & 1) * NN_KC * SLD;
    ldss(sf0, L, 0);
    load_a(a3, is + 12);
    ldss(sf1, L, 1);
    mma(a0, sf0);
    load_a(a0, is + 16);
    ldss(sf0, L, 2);
    mma(a1, sf1);
    load_a(a1, is + 20);
    ldss(sf1, L, 3);
    mma(a2, sf0);
    load_a(a2, is + 24);
    ldss(sf0, L, 4);
    mma(a3, sf1);
    load_a(a3, is + 28);
    ldss(sf1, L, 5);
    mma(a0, sf0);
    load_a(a0, is + 32);
    ldss(sf0, L, 6);
    mma(a1, sf1);
    load_a(a1, is + 36);
    ldss(sf1, L, 7);
    mma(a2, sf0);
    load_a(a2, is + 40);
    mma(a3, sf1);
    if (has_next) stage_store(lds + ((s + 1) & 1) * NN_KC * SLD, is + NN_KC);
    __syncthreads();
  }

  // Raw accumulator stores only: any VALU arithmetic on the accumulators here makes hipcc keep them in
  // VGPRs across the loop back-edge (256 v_accvgpr copies per stage); scaling/accumulation is done by the caller.
#pragma unroll
  for (int nt = 0; nt < NTF; ++nt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int j = nt * 16 + kk + 4 * rg;
      if (j < r) {
        double* yc = Yo + (int64_t)j * ldout;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          const int64_t t = t0 + tp * 32 + 2 * c16;
          if (t + 1 < N) {
            *reinterpret_cast<d2*>(yc + t) = d2{acc[2 * tp][nt][rg], acc[2 * tp + 1][nt][rg]};
          } else if (t < N) {
            yc[t] = acc[2 * tp][nt][rg];
          }
        }
        if (ODD) {
          const int64_t t = t0 + TP * 32 + c16;
          if (t < N) yc[t] = acc[TT - 1][nt][rg];
        }
      }
    }
  if constexpr (R4 > 0) {
#pragma unroll
    for (int q = 0; q < R4; ++q) {
      const int j = NTF * 16 + 4 * q + kk;
      if (j < r) {
        double* yc = Yo + (int64_t)j * ldout;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          const int64_t t = t0 + tp * 32 + 2 * c16;
          if (t + 1 < N) {
            *reinterpret_cast<d2*>(yc + t) = d2{acc4[2 * tp][q], acc4[2 * tp + 1][q]};
          } else if (t < N) {
            yc[t] = acc4[2 * tp][q];
          }
        }
        if (ODD) {
          const int64_t t = t0 + TP * 32 + c16;
          if (t < N) yc[t] = acc4[TT - 1][q];
        }
      }
    }
  }
}

// Y[j][t] = sum_s part[s][j][t]  (fixed order), rows row0 <= t < N (row0 even)
__global__ void k_reduce_nn(const double* __restrict__ part, int msplit, int64_t pstride, int64_t ldp, double* __restrict__ Y,
                            int64_t ldy, int64_t row0, int64_t N, int r) {
  for (int j = blockIdx.y; j < r; j += gridDim.y) {
    const double* p = part + (int64_t)j * ldp;
    double* y = Y + (int64_t)j * ldy;
    for (int64_t t = row0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        d2 acc = *reinterpret_cast<const d2*>(p + t);
        for (int sp = 1; sp < msplit; ++sp) {
          const d2 v = *reinterpret_cast<const d2*>(p + (int64_t)sp * pstride + t);
          acc.x += v.x;
          acc.y += v.y;
        }
        *reinterpret_cast<d2*>(y + t) = acc;
      } else {
        double acc = p[t];
        for (int sp = 1; sp < msplit; ++sp) acc += p[(int64_t)sp * pstride + t];
        y[t] = acc;
      }
    }
  }
}

// tallest wave tile (in 16-row tiles along the long axis) for nt column tiles
static inline int nn_tt(int nt, int waves) {
  static const int t4[17] = {0, 8, 8, 8, 8, 6, 5, 4, 4, 3, 3, 2, 2, 2, 2, 2, 2};
  static const int t8[17] = {0, 8, 8, 5, 4, 3, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1};
  return waves == 4 ? t4[nt] : t8[nt];
}

// Launch plan of tsgemm_nn for a tile of `tile_rows` rows: how many ways to split the reduction axis m so that the
// grid fills the 256 CUs in (nearly) whole rounds.  Time model (the kernel is MFMA bound, the split partials only
// cost their own HBM round trip in k_reduce_nn): t = flops / (eff * rate) + (msplit + 1) * N * r * 8 / hbm.
static double nn_plan(hfmi_ctx* ctx, int tile_rows, int m, int r, int64_t N, double rate_factor, int* msplit_out) {
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  const int64_t ntiles = (N + tile_rows - 1) / tile_rows;
  const int stages = (m + NN_KC - 1) / NN_KC;
  const double flops = 2.0 * (double)ntiles * tile_rows * (double)m * (double)(((r + 15) / 16) * 16);
  const double rate = 60e12 * rate_factor, hbm = 4.0e12;
  int best = 1;
  double best_t = 1e300;
  for (int ns = 1; ns <= 64; ++ns) {
    if (ns > 1 && stages / ns < 8) break;
    const int64_t blocks = ntiles * ns;
    const int64_t rounds = (blocks + cus - 1) / cus;
    const double eff = (double)blocks / (double)(rounds * cus);
    const double t = flops / (eff * rate) + (ns > 1 ? (ns + 1.0) * (double)N * r * 8.0 / hbm + 3e-6 : 0.0);
    if (t < best_t - 1e-12) {
      best_t = t;
      best = ns;
    }
  }
  *msplit_out = best;
  return best_t;
}

template <int TT, int NT, int WAVES>
static int nn_launch_inst(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r,
                          double* Y, int64_t ldy, int64_t N, int msplit, bool tail_split = false) {
  constexpr int SLD = NT * 16 + ((NT % 2 == 0) ? 16 : 0);
  const size_t shmem = (size_t)2 * NN_KC * SLD * sizeof(double);
  // columns of the last tile: up to 12 are done as 1..3 groups of 4 with the 4x4x4 MFMA
  const int rem = r - (NT - 1) * 16;
  const int r4 = (g_rem4 && rem <= 12) ? (rem + 3) / 4 : 0;
  auto kern = r4 == 1 ? k_tsgemm_nn<TT, NT, WAVES, 1> : r4 == 2 ? k_tsgemm_nn<TT, NT, WAVES, 2>
            : r4 == 3 ? k_tsgemm_nn<TT, NT, WAVES, 3> : k_tsgemm_nn<TT, NT, WAVES, 0>;
  HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const int tile_rows = 16 * TT * WAVES;
  const int ntiles = (int)((N + tile_rows - 1) / tile_rows);
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  // msplit > 1 means the plan found the row tiles badly quantised over the CUs.  With at least one full round of
  // tiles, only the tiles beyond the last full round are split (see the kernel); otherwise every tile is.
  int full_tiles = 0;
  if ((msplit > 1 || tail_split) && ntiles >= cus && g_nn_hybrid) {
    full_tiles = ntiles / cus * cus;
    const int tail = ntiles - full_tiles;
    if (tail == 0) {
      msplit = 1;
    } else {
      const int stages = (m + NN_KC - 1) / NN_KC;
      int ms = cus / tail;                                  // one round of CUs for the tail
      if (ms > stages / 4) ms = stages / 4;                 // at least four LDS stages per workgroup
      if (ms < 1) ms = 1;
      msplit = ms;
      if (msplit == 1) full_tiles = 0;                      // nothing to split: plain launch
    }
  }
  int mchunk = (int)round_up((m + msplit - 1) / msplit, NN_KC);
  msplit = (m + mchunk - 1) / mchunk;
  double* out = Y;
  int64_t ldo = ldy, pstride = 0;
  if (msplit > 1) {
    ldo = round_up(N, 32);
    pstride = ldo * r;
    void* pv = nullptr;
    HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)msplit * pstride * sizeof(double), &pv));
    out = (double*)pv;
  } else {
    full_tiles = 0;
  }
  const int tail_tiles = ntiles - full_tiles;
  dim3 grid((unsigned)(full_tiles + tail_tiles * msplit)), block(WAVES * 64);
  hipLaunchKernelGGL(kern, grid, block, shmem, ctx->stream, A, lda, m, S, lds_, r, out, ldo, N, ntiles, msplit, mchunk, pstride,
                     full_tiles, Y, ldy);
  HIP_TRY(hipGetLastError());
  if (msplit > 1) {
    const int64_t row0 = (int64_t)full_tiles * tile_rows;   // multiple of 64
    int64_t gx = ((N - row0 + 1) / 2 + 255) / 256;
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_reduce_nn, dim3((unsigned)gx, (unsigned)r), dim3(256), 0, ctx->stream, (const double*)out, msplit,
                       pstride, ldo, Y, ldy, row0, N, r);
    HIP_TRY(hipGetLastError());
  }
  return HFMI_OK;
}

// one-wave-per-SIMD variants: the tile height is chosen among TMAX, TMAX-1, TMAX-2 (16-row tiles per wave) together
// with the reduction split, by the time model above -- a slightly shorter tile often fills the last round of CUs
template <int NT, int TMAX>
static int nn_launch_w4(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double* Y,
                        int64_t ldy, int64_t N) {
  constexpr int T1 = TMAX > 1 ? TMAX - 1 : 1, T2 = TMAX > 2 ? TMAX - 2 : 1;
  int ms0 = 1, ms1 = 1, ms2 = 1;
  const double c0 = nn_plan(ctx, 64 * TMAX, m, r, N, 1.0, &ms0);
  const double c1 = (T1 != TMAX) ? nn_plan(ctx, 64 * T1, m, r, N, 0.98, &ms1) : 1e300;
  const double c2 = (T2 != T1) ? nn_plan(ctx, 64 * T2, m, r, N, 0.96, &ms2) : 1e300;
  // With at least one full round of the tallest tiles the quantisation is handled by splitting only the tail tiles
  // (nn_launch_inst), so the tallest tile -- the best MFMA-to-LDS ratio -- is taken (A/B r01e: config 4 nn 56.5 -> 59.7 TF)
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  if (g_nn_hybrid && g_nn_tt == 0 && (N + 64 * TMAX - 1) / (64 * TMAX) >= cus && m >= 16 * NN_KC)
    return nn_launch_inst<TMAX, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, 1, true);
  if (g_nn_tt == 1) return nn_launch_inst<TMAX, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms0);
  if (g_nn_tt == 2) return nn_launch_inst<T1, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms1);
  if (g_nn_tt == 3) return nn_launch_inst<T2, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms2);
  if (c0 <= c1 && c0 <= c2) return nn_launch_inst<TMAX, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms0);
  if (c1 <= c2) return nn_launch_inst<T1, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms1);
  return nn_launch_inst<T2, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms2);
}

template <int NT, int TT>
static int nn_launch_w8(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double* Y,
                        int64_t ldy, int64_t N) {
  int ms = 1;
  nn_plan(ctx, 128 * TT, m, r, N, 1.0, &ms);
  return nn_launch_inst<TT, NT, 8>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms);
}

static int nn_panel(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double* Y,
                    int64_t ldy, int64_t N) {
  const int nt = (r + 15) / 16;
  tuning_init();
  const int waves = g_nn_waves ? g_nn_waves : (nt >= 10 ? 8 : 4);   // A/B (scripts/gemm_ab.py, r01e): 4 waves win up to 9 column tiles
#define NN_CASE(NTV, TT4, TT8)                                                             \
  case NTV:                                                                                \
    if (waves == 8) return nn_launch_w8<NTV, TT8>(ctx, A, lda, m, S, lds_, r, Y, ldy, N);  \
    return nn_launch_w4<NTV, TT4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N);
  switch (nt) {
    NN_CASE(1, 8, 8) NN_CASE(2, 8, 8) NN_CASE(3, 8, 5) NN_CASE(4, 8, 4) NN_CASE(5, 6, 3) NN_CASE(6, 5, 2)
    NN_CASE(7, 4, 2) NN_CASE(8, 4, 2) NN_CASE(9, 3, 1) NN_CASE(10, 3, 1) NN_CASE(11, 2, 1) NN_CASE(12, 2, 1)
    NN_CASE(13, 2, 1) NN_CASE(14, 2, 1) NN_CASE(15, 2, 1) NN_CASE(16, 2, 1)
  }
#undef NN_CASE
  HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: panel too wide (%d)", r);
}

__global__ void k_small_scale_copy(double* __restrict__ dst, const double* __restrict__ src, int rows, int cols, int ld,
                                   double alpha) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ld) return;
  for (int i = blockIdx.y; i < rows; i += gridDim.y) dst[(int64_t)i * ld + j] = (j < cols) ? alpha * src[(int64_t)i * ld + j] : 0.0;
}

// Y = alpha * A * S + beta * Y.  alpha is folded into a scaled copy of the small matrix; beta goes through a
// scratch block + axpby, so the MFMA kernel itself only ever stores raw accumulators.
int launch_tsgemm_nn(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double alpha,
                     double beta, double* Y, int64_t ldy, int64_t N) {
  if (r <= 0 || N <= 0) return HFMI_OK;
  if (m <= 0) HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: empty reduction");
  if (lds_ % 2 != 0) HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: small-matrix leading dimension must be even");
  if (lda % 32 != 0 || lda < round_up(N, 32))
    HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: leading dimension must be a multiple of 32 and >= round_up(N,32)");
  if (r > 256 && A == Y) HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: in-place update needs r <= 256");
  if (alpha != 1.0) {
    void* sc = nullptr;
    HFMI_TRY(ctx_ws(ctx, WS_MISC, (size_t)m * lds_ * sizeof(double), &sc));
    dim3 block(128), grid((lds_ + 127) / 128, m < 4096 ? m : 4096);
    hipLaunchKernelGGL(k_small_scale_copy, grid, block, 0, ctx->stream, (double*)sc, S, m, r, lds_, alpha);
    HIP_TRY(hipGetLastError());
    S = (const double*)sc;
  }
  double* out = Y;
  int64_t ldo = ldy;
  if (beta != 0.0) {
    void* tmp = nullptr;
    ldo = round_up(N, 32);
    HFMI_TRY(ctx_ws(ctx, WS_STAGE, (size_t)ldo * r * sizeof(double), &tmp));
    out = (double*)tmp;
  }
  for (int r0 = 0; r0 < r; r0 += 256) {
    const int rp = (r - r0 < 256) ? (r - r0) : 256;
    const int pidx = prof_start(ctx, 1, m, rp, N);
    HFMI_TRY(nn_panel(ctx, A, lda, m, S + r0, lds_, rp, out + (int64_t)r0 * ldo, ldo, N));
    prof_stop(ctx, pidx);
  }
  if (beta != 0.0) {
    if (beta != 1.0) HFMI_TRY(launch_scale(ctx, Y, ldy, N, r, beta));
    HFMI_TRY(launch_axpy(ctx, Y, ldy, 1.0, out, ldo, N, r));
  }
  return HFMI_OK;
}
