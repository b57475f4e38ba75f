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
// indices as long as A and B agree, which lets each lane fetch 16 or 32 contiguous bytes of its own
// vector straight from HBM into the fragment layout -- no transposition through LDS.  The operand
// shared by the waves of a workgroup (B in tn, S in nn) is staged through LDS; the streamed
// operand goes HBM -> VGPR directly, prefetched one iteration ahead.  A wave holds a
// (16*MT) x (16*NT) fp64 accumulator tile: up to 16 MFMA tiles with two waves per SIMD (the default),
// up to 32 with one.  A last column tile of at most 12 columns runs on the 4x4x4 MFMA (R4).
// tsgemm_nn lives in hfmi_gemm_nn.hip, the skinny x skinny case of tn in hfmi_skinny.hip.
//
// MFMA fragment maps (cdna_hip_programming.md section 3): lane l supplies A[i = l&15][kk = l>>4] and
// B[kk = l>>4][j = l&15]; it receives D[row = (l>>4) + 4*reg][col = l&15], reg = 0..3.
#include "hfmi_gemm_common.h"

// =====================================================================================
// tsgemm_tn
// =====================================================================================
constexpr int TN_BK = 32;   // reduction indices per LDS stage
// the streamed operand is read once by one workgroup: -DHFMI_TN_NT marks its loads non-temporal (A/B build, scripts/build_variant.sh)
#ifdef HFMI_TN_NT
#define TN_LOAD_A(p) __builtin_nontemporal_load(p)
#else
#define TN_LOAD_A(p) (*(p))
#endif
// the finely split tail of a launch (see the block mapping in k_tsgemm_tn); nrb = 0: none
struct TnTail {
  int nrb, nsplit;
  int64_t chunk;
  double* out;
  int64_t si, sj, sps;
};

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
                                                                    const double* B /* not restrict: see the stage loop */, int64_t ldb, int k,
                                                                    int64_t Npad, int64_t chunk, int nrb, int nsplit,
                                                                    double* __restrict__ out, int64_t si, int64_t sj,
                                                                    int64_t sps, int direct, int probe_arg, TnTail tail) {
  // timing-only diagnostic (scripts/tn_probe.py; results are garbage): compiled in only with -DHFMI_TN_PROBE
  // (HFMI_EXTRA_HIPCC_FLAGS), the production instance carries no run-time branch for it
#ifdef HFMI_TN_PROBE
  const int probe = probe_arg;
#else
  constexpr int probe = 0;
  (void)probe_arg;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = TN_BK;
  constexpr int COLS = NT * 16;
  constexpr int LDB = BK + 2;
  constexpr int BUFD = COLS * LDB;                // doubles per stage buffer
  constexpr int CSH = 4;                          // log2(16-byte chunks per column per stage)
  constexpr int NTF = R4 > 0 ? NT - 1 : NT;       // full 16-column tiles
  constexpr int NR4 = R4 > 0 ? R4 : 1;
  constexpr int NTHR = WAVES * 64;
  constexpr int CH = COLS * (BK / 2);             // 16-byte chunks per stage
  constexpr int NQ = (CH + NTHR - 1) / NTHR;      // chunks per thread
  // AH = 2: one iteration is 16 reduction indices and every lane fetches 32 contiguous bytes of its vector (see
  // load_a); taken where the VGPR estimate leaves room for the 8 MT extra registers, else 8 indices / 16 bytes
  constexpr int EST = MT * NTF * 8 + MT * R4 * 2 + NTF * 4 + R4 * 4 + 16 * MT + NQ * 6;
  constexpr int AH = (EST <= ((WAVES == 8 || MT * NT <= 20) ? 218 : 470)) ? 2 : 1;
  double* lds = reinterpret_cast<double*>(smem);  // [2][NT*16][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;

  // Row blocks [0, nrb) are whole rounds of the CUs, each split nsplit ways over the reduction axis; the tail.nrb row blocks
  // behind them (fewer than one round: m = 1e5 is 261 blocks of 384 rows on 256 CUs) are split tail.nsplit ways -- as finely as
  // it takes to fill one more round of workgroups for 1 / tail.nsplit of a block's time -- and write to their own, compact
  // partial buffer.  The two groups are remapped over the XCDs separately (see k_tsgemm_nn).
  const int nfull = nrb * nsplit;
  const bool in_tail = (int)blockIdx.x >= nfull;
  const int logical = in_tail ? xcd_remap((int)blockIdx.x - nfull, tail.nrb * tail.nsplit) : xcd_remap(blockIdx.x, nfull);
  const int sp = in_tail ? logical / tail.nrb : logical / nrb;
  const int rb = in_tail ? nrb + logical % tail.nrb : logical % nrb;
  if (in_tail) {
    chunk = tail.chunk;
    out = tail.out;
    si = tail.si;
    sj = tail.sj;
    sps = tail.sps;
    direct = 0;
  }
  const int row_off = in_tail ? nrb * (16 * MT * WAVES) : 0;   // the tail's partial buffer starts at its first row
  const int64_t t_begin = (int64_t)sp * chunk;
  int64_t t_end = t_begin + chunk;
  if (t_end > Npad) t_end = Npad;
  const int nstages = (int)((t_end - t_begin) / BK);
  const int64_t t_last = t_end - 8 * AH;  // last iteration base that is safe to fetch
  const int rowbase = rb * (16 * MT * WAVES) + wave * (16 * MT);

  const double* a_ptr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = rowbase + mt * 16 + r16;
    if (row > m - 1) row = m - 1;
    a_ptr[mt] = A + (int64_t)row * lda + kk * 2 * AH;
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
  // AH = 2: one iteration = 16 reduction indices; lane (r16, kk) fetches the 32 contiguous bytes t + 4 kk .. t + 4 kk + 3
  // of vector r16 as two 16-byte loads issued back to back, so the four kk lanes cover a whole 128-byte line of every
  // vector at once (HBM delivers 128-byte runs at 6.3-6.7 TB/s, 64-byte runs at 3.5-3.9 TB/s: scripts/stream_probe2.hip).
  // Any assignment of reduction indices to (lane, k-step) is legal as long as the LDS fragments of B use the same one:
  // half h of the iteration covers indices t + 2 AH kk + 2 h + {0, 1}.
  auto load_a = [&](d2(&dst)[AH][MT], int64_t t) {
    if (t > t_last) t = t_last;  // wave-uniform clamp: prefetches past the end re-read valid data
    if (probe & 1) t = t_begin;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < AH; ++h) dst[h][mt] = TN_LOAD_A(reinterpret_cast<const d2*>(a_ptr[mt] + t + 2 * h));
  };
  // Full tiles: lane (r16, kk) reads column nt*16 + r16.  The fragments are SINGLE-buffered: the moment the MFMAs
  // of column tile nt have been issued, its registers are refilled with the next iteration's fragment, and the MFMAs of
  // the other column tiles cover the LDS latency (the sched_barriers pin that order; left alone the scheduler hoists
  // all reads to the top, which is the double-buffered register budget again -- 4 NT VGPRs more).  The 4-column groups
  // of the last tile (lane -> column NTF*16 + 4 q + (lane & 3), the same for every row group) are read at the top of
  // their own iteration; the full-tile MFMAs in front of them cover the latency.
  d2 bf[NTA];
  // `it` counts half iterations (8 reduction indices): offset of this lane's two indices inside the stage
  auto koff = [&](int it) { return (it / AH) * (8 * AH) + kk * (2 * AH) + (it % AH) * 2; };
  auto ldsb1 = [&](const double* L, int it, int nt) {
    bf[nt] = *reinterpret_cast<const d2*>(L + (nt * 16 + r16) * LDB + koff(it));
  };
  auto mma = [&](const d2(&a)[MT], const double* L, int it, bool refill) {
    d2 bg[NR4];
    if constexpr (R4 > 0) {
#pragma unroll
      for (int q = 0; q < R4; ++q)
        bg[q] = *reinterpret_cast<const d2*>(L + (NTF * 16 + 4 * q + (lane & 3)) * LDB + koff(it));
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
  // streamed operand: two register buffers over iterations of 8 AH reduction indices (prefetch distance one iteration;
  // a ring of 4 x 8 indices at distance 3 measured no faster on any shape -- latency is not what limits the kernel)
  {
    constexpr int NIT = BK / (8 * AH), NH = BK / 8;
    d2 a[2][AH][MT];
    load_a(a[0], t_begin);
    for (int s = 0; s < nstages; ++s) {
      const int64_t ts = t_begin + (int64_t)s * BK;
      const bool has_next = s + 1 < nstages;
      // UNCONDITIONAL (the last stage re-reads its own slab): vmcnt retires loads in issue order, and these loads sit
      // between the streamed operand's fragment fetched one iteration ago and the wait in front of its first MFMA.  Behind
      // a branch (as it was until round 2) the compiler has to assume the path WITHOUT them at the join and emits
      // vmcnt(6) instead of vmcnt(6 + NQ): every stage then began by waiting for these loads' L2 / HBM round trip --
      // 10-12 % of the kernel (round-2 experiment builds, profiles/r02o_tn_variants.txt).
      stage_load(has_next ? ts + BK : ts);
      asm volatile("" ::: "memory");   // keeps the loads HERE: left alone they are sunk to their ds_write and waited for with vmcnt(0)
      const double* L = lds + (s & 1) * BUFD;
#pragma unroll
      for (int nt = 0; nt < NTF; ++nt) ldsb1(L, 0, nt);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        load_a(a[(it + 1) & 1], ts + 8 * AH * (it + 1));
#pragma unroll
        for (int h = 0; h < AH; ++h) mma(a[it & 1][h], L, it * AH + h, it * AH + h + 1 < NH);
      }
      if (!(probe & 2)) {
      if (has_next) stage_store(lds + ((s + 1) & 1) * BUFD);
      __syncthreads();
      }
    }
  }

  // Results: element (i, j) goes to P[i si + j sj].  Split launches write raw partial tiles (padded image of split
  // sp, summed by k_reduce_partials); a launch with ONE split and nothing to scale or accumulate writes the result
  // block itself (direct: bounds-checked) -- config 2's m = 1e5 launches used to spend 0.28 ms copying one partial.
  double* P = out + (int64_t)sp * sps;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = TR ? rowbase + mt * 16 + r16 : rowbase + mt * 16 + kk + 4 * r;
        const int j = TR ? nt * 16 + kk + 4 * r : nt * 16 + r16;
        if (!direct || (i < m && j < k)) P[(int64_t)(i - row_off) * si + (int64_t)j * sj] = acc[mt][nt][r];
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
        if (!direct || (i < m && j < k)) P[(int64_t)(i - row_off) * si + (int64_t)j * sj] = acc4[mt][q];
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

// Vector form of the two kernels above for the layouts the big launches use -- the fast axis is contiguous in the partials AND
// in C, rows 16-byte aligned: every thread owns TWO consecutive outputs and walks the slices with four independent running
// sums of 16-byte loads (the 8-byte, one-load-per-thread forms above moved 0.5 TB/s: 0.12 ms for the 65 MB of a config-4
// launch).  Same fixed summation order on every run.  kreal < rowlen: columns >= kreal of a row are written as zeros.
__global__ __launch_bounds__(256) void k_reduce_vec(const double* __restrict__ part, int nsplit, int64_t pstride, int64_t prow,
                                                    int64_t rowlen, int nrows, int64_t kpad, int kreal, double scale, double beta,
                                                    double* __restrict__ C, int64_t crow) {
  const int64_t f = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (f >= rowlen) return;
  for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
    const double* p = part + (int64_t)row * prow + f;
    d2 s0 = d2{0.0, 0.0}, s1 = s0, s2 = s0, s3 = s0;
    int sp = 0;
    for (; sp + 3 < nsplit; sp += 4) {
      s0 += *reinterpret_cast<const d2*>(p + (int64_t)sp * pstride);
      s1 += *reinterpret_cast<const d2*>(p + (int64_t)(sp + 1) * pstride);
      s2 += *reinterpret_cast<const d2*>(p + (int64_t)(sp + 2) * pstride);
      s3 += *reinterpret_cast<const d2*>(p + (int64_t)(sp + 3) * pstride);
    }
    for (; sp < nsplit; ++sp) s0 += *reinterpret_cast<const d2*>(p + (int64_t)sp * pstride);
    d2 v = ((s0 + s1) + (s2 + s3)) * scale;
    double* out = C + (int64_t)row * crow + f;
    if (beta != 0.0) v += beta * *reinterpret_cast<const d2*>(out);
    if (kreal < kpad) {
      if ((int)(f % kpad) >= kreal) v.x = 0.0;
      if ((int)((f + 1) % kpad) >= kreal) v.y = 0.0;
    }
    *reinterpret_cast<d2*>(out) = v;
  }
}

int launch_reduce_partials(hfmi_ctx* ctx, const double* part, int nsplit, int64_t pstride, int inner_ld, bool tr, int m,
                           int k, double scale, double beta, double* C, int64_t rs, int64_t cs) {
  const int fastn = tr ? m : k, slown = tr ? k : m;
  {
    // vector path: fast axis contiguous and even-strided on both sides, pointers 16-byte aligned, enough work to matter
    const int64_t cfast = tr ? rs : cs, crow = tr ? cs : rs;
    const bool aligned = (((uintptr_t)part | (uintptr_t)C) & 15) == 0 && pstride % 2 == 0 && inner_ld % 2 == 0 && crow % 2 == 0;
    if (cfast == 1 && aligned && (int64_t)fastn * slown >= 65536) {
      int64_t rowlen, prow = inner_ld, kpad = 0;
      int nrows, kreal = 0;
      if (!tr && rs == inner_ld) {          // rows are back to back on both sides: one long row, pad columns zeroed
        rowlen = (int64_t)m * inner_ld;
        nrows = 1;
        kpad = inner_ld;
        kreal = k;
      } else {
        rowlen = fastn & ~1;                // an odd tail element goes through the scalar kernel below
        nrows = slown;
        kpad = kreal = 0;
      }
      if (rowlen == fastn || nrows == 1) {
        dim3 vgrid((unsigned)((rowlen / 2 + 255) / 256), (unsigned)(nrows < 32768 ? nrows : 32768));
        hipLaunchKernelGGL(k_reduce_vec, vgrid, dim3(256), 0, ctx->stream, part, nsplit, pstride, prow, rowlen, nrows, kpad, kreal, scale,
                           beta, C, crow);
        HIP_TRY(hipGetLastError());
        return HFMI_OK;
      }
    }
  }
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
static int g_waves = 0, g_rem4 = 1, g_probe = 0, g_tn_mt = 0;
// where a launch writes: split partials (the reduce kernel sums them) or, for one split with nothing to scale, C itself
struct TnOut {
  bool direct;
  double* C;
  int64_t rs, cs;
  // tail plan (tn_panel): row blocks beyond the whole rounds, their split and their compact partial buffer
  int tail_nrb, tail_nsplit;
  int64_t tail_chunk;
  double* tail_part;
  int tail_mpad;
};
static int g_ss = 1;                                 // route skinny x skinny contractions to tsgemm_ss (hfmi_skinny.hip)
static int g_tn_hybrid = 1;                          // tsgemm_tn: whole rounds of row blocks coarsely split + a finely split tail (A/B: "tn_hybrid")
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
  else if (key && !strcmp(key, "rem4") && (value == 0 || value == 1)) { g_rem4 = value; nn_tuning_set(key, value); }
  else if (key && nn_tuning_set(key, value)) {}
  else if (key && eig_tuning_set(key, value)) {}
  else if (key && chol_tuning_set(key, value)) {}
  else if (key && api_tuning_set(key, value)) {}
  else if (key && !strcmp(key, "ss") && (value == 0 || value == 1)) g_ss = value;
  else if (key && !strcmp(key, "probe")) g_probe = value;
  else if (key && !strcmp(key, "tn_hybrid") && (value == 0 || value == 1)) g_tn_hybrid = value;
  else if (key && !strcmp(key, "tn_mt") && value >= 0 && value <= 8) g_tn_mt = value;   // A/B: wave tile height of tsgemm_tn (0 = automatic)
  else if (key && !strcmp(key, "ss_percu") && value >= 1 && value <= 4) tsgemm_ss_set_percu(value);
  else if (key && !strcmp(key, "ss_blocked") && value >= 0 && value <= 2) tsgemm_ss_set_blocked(value);
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
                         int64_t chunk, int nrb, int nsplit, double* part, int mpad, int kpad, const TnOut& o) {
  const size_t shmem = (size_t)2 * NT * 16 * (TN_BK + 2) * sizeof(double);
  auto kern = k_tsgemm_tn<MT, NT, TR, WAVES, R4>;
  HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  double* out = o.direct ? o.C : part;
  const int64_t si = o.direct ? o.rs : (TR ? 1 : kpad), sj = o.direct ? o.cs : (TR ? mpad : 1);
  TnTail tail = {o.tail_nrb, o.tail_nsplit, o.tail_chunk, o.tail_part, TR ? 1 : kpad, TR ? o.tail_mpad : 1, (int64_t)o.tail_mpad * kpad};
  hipLaunchKernelGGL(kern, dim3(nrb * nsplit + o.tail_nrb * o.tail_nsplit), dim3(WAVES * 64), shmem, ctx->stream, A, lda, m, B, ldb, k,
                     N, chunk, nrb, nsplit, out, si, sj, (int64_t)mpad * kpad, o.direct ? 1 : 0, g_probe, tail);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// r4: number of 4-column groups the last column tile is computed in (0 = as a full 16-column tile)
template <int MT, int NT, int WAVES>
static int tn_launch_inst(hfmi_ctx* ctx, bool tr, int r4, const double* A, int64_t lda, int m, const double* B, int64_t ldb,
                          int k, int64_t N, int64_t chunk, int nrb, int nsplit, double* part, int mpad, int kpad,
                          const TnOut& o) {
#define TN_R4(R)                                                                                                          \
  case R:                                                                                                                 \
    if (tr) return tn_launch_one<MT, NT, WAVES, true, R>(ctx, A, lda, m, B, ldb, k, N, chunk, nrb, nsplit, part, mpad, kpad, o); \
    return tn_launch_one<MT, NT, WAVES, false, R>(ctx, A, lda, m, B, ldb, k, N, chunk, nrb, nsplit, part, mpad, kpad, o);
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
                          int kpad, const TnOut& o) {
  constexpr int LIM = (WAVES == 8) ? 20 : 32;
#define TN_CASE(M)                                                                                                 \
  case M:                                                                                                          \
    if constexpr (M * NT <= LIM)                                                                                   \
      return tn_launch_inst<M, NT, WAVES>(ctx, tr, r4, A, lda, m, B, ldb, k, N, chunk, nrb, nsplit, part, mpad, kpad, o); \
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
  if (g_tn_mt > 0 && g_tn_mt < mt) mt = g_tn_mt;
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
  // Hybrid plan: when the row blocks make at least one whole round of the CUs, the rounds that ARE whole need no fine split at
  // all (ns_full = 1 or 2: their partial traffic is a single slice or none) and only the blocks beyond them are split finely
  // enough to fill one more round for 1 / ns_tail of a block's time.  m = 1e5 (config 2): 261 blocks = 256 whole + 5 x 51 instead
  // of 261 x 19 (the uniform plan's best: 3 % quantisation loss, 1.5 GB of partials written and read back, a 0.3 ms reduction);
  // m = 51200 (config 4): 128 x 2 + 6 x 42 instead of 134 x 21 (0.7 GB of partials, 0.12 ms).
  const bool can_direct = (scale == 1.0 && beta == 0.0 && C != A && C != B);
  int ns_full = 0, nrb_full = 0, ns_tail = 0;
  if (nsplit_req <= 0 && g_tn_hybrid) {
    const int64_t stages = Npad / TN_BK;
    const double ideal = (double)nrb / cus;
    const double part_unit = 2.0 * (double)rows_per_block * kpad / ((double)N * (m + k));   // one slice of one row block
    double uniform_cost;
    {
      const int64_t blocks = (int64_t)nrb * nsplit;
      const int64_t rounds = (blocks + cus - 1) / cus;
      uniform_cost = (double)(rounds * cus) / (double)blocks + part_unit * nsplit * nrb;
    }
    double best_cost = uniform_cost;
    static const int cand[] = {1, 2, 3, 4, 6, 8};
    for (int ci = 0; ci < 6; ++ci) {
      const int nsf = cand[ci];
      if (nsf > 1 && stages / nsf < 16) break;
      const int64_t R = ((int64_t)nrb * nsf) / cus;            // whole rounds of full-region workgroups
      if (R < 1 || (R * cus) % nsf != 0) continue;
      const int nf = (int)(R * cus / nsf);
      const int nt_blocks = nrb - nf;
      if (nt_blocks <= 0) continue;                             // the uniform plan already is this one
      int nst = cus / nt_blocks;
      if (nst > stages / 16) nst = (int)(stages / 16);
      if (nst > 128) nst = 128;
      if (nst < 1) nst = 1;
      const int64_t tail_rounds = ((int64_t)nt_blocks * nst + cus - 1) / cus;
      const double time = (double)R / nsf + (double)tail_rounds / nst;
      const double parts = part_unit * ((nsf == 1 && can_direct ? 0.0 : (double)nsf * nf) + (double)nst * nt_blocks);
      const double cost = time / ideal + parts;
      if (cost < best_cost - 1e-9) {
        best_cost = cost;
        ns_full = nsf;
        nrb_full = nf;
        ns_tail = nst;
      }
    }
  }
  const int rem = k - (nt - 1) * 16;
  const int r4 = (g_rem4 && waves == 8 && rem <= 12) ? (rem + 3) / 4 : 0;
  if (ns_full > 0) {
    const int nrb_t = nrb - nrb_full;
    const int mpad_f = nrb_full * rows_per_block, mpad_t = nrb_t * rows_per_block;
    int64_t chunk_f = round_up((Npad + ns_full - 1) / ns_full, stage_len);
    ns_full = (int)((Npad + chunk_f - 1) / chunk_f);
    int64_t chunk_t = round_up((Npad + ns_tail - 1) / ns_tail, stage_len);
    ns_tail = (int)((Npad + chunk_t - 1) / chunk_t);
    const bool direct_f = ns_full == 1 && can_direct;
    const size_t full_doubles = direct_f ? 0 : (size_t)ns_full * mpad_f * kpad;
    void* partv = nullptr;
    HFMI_TRY(ctx_ws(ctx, WS_PART, (full_doubles + (size_t)ns_tail * mpad_t * kpad) * sizeof(double), &partv));
    double* part_f = (double*)partv;
    double* part_t = part_f + full_doubles;
    const TnOut o = {direct_f, C, rs, cs, nrb_t, ns_tail, chunk_t, part_t, mpad_t};
    const int pidx = prof_start(ctx, 0, m, k, N);
#define TN_NT(NTV)                                                                                                       \
  case NTV:                                                                                                              \
    if (waves == 8)                                                                                                      \
      HFMI_TRY((tn_dispatch_mt<NTV, 8>(ctx, mt, tr, r4, A, lda, m, B, ldb, k, Npad, chunk_f, nrb_full, ns_full, part_f, mpad_f, kpad, o))); \
    else                                                                                                                 \
      HFMI_TRY((tn_dispatch_mt<NTV, 4>(ctx, mt, tr, 0, A, lda, m, B, ldb, k, Npad, chunk_f, nrb_full, ns_full, part_f, mpad_f, kpad, o))); \
    break;
    switch (nt) {
      TN_NT(1) TN_NT(2) TN_NT(3) TN_NT(4) TN_NT(5) TN_NT(6) TN_NT(7) TN_NT(8) TN_NT(9) TN_NT(10) TN_NT(11) TN_NT(12)
      TN_NT(13) TN_NT(14) TN_NT(15) TN_NT(16)
      default:
        HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_tn: panel too wide (%d)", k);
    }
#undef TN_NT
    prof_stop(ctx, pidx);
    const int m_full = mpad_f;                        // every row of the whole rounds is a real row (only the last block is ragged)
    if (!direct_f)
      HFMI_TRY(launch_reduce_partials(ctx, part_f, ns_full, (int64_t)mpad_f * kpad, tr ? mpad_f : kpad, tr, m_full, k, scale, beta, C,
                                      rs, cs));
    return launch_reduce_partials(ctx, part_t, ns_tail, (int64_t)mpad_t * kpad, tr ? mpad_t : kpad, tr, m - m_full, k, scale, beta,
                                  C + (int64_t)m_full * rs, rs, cs);
  }
  int64_t chunk = round_up((Npad + nsplit - 1) / nsplit, stage_len);
  if (chunk < stage_len) chunk = stage_len;
  nsplit = (int)((Npad + chunk - 1) / chunk);
  if (nsplit < 1) nsplit = 1;
  void* partv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)nsplit * mpad * kpad * sizeof(double), &partv));
  double* part = (double*)partv;
  // algorithmic work of this launch: flops 2 N m k, bytes 8 (N m + N k + m k)
  // columns of the last tile: up to 12 are done as 1..3 groups of 4 with the 4x4x4 MFMA (16 instead of 64 cycles each)
  const bool direct = (nsplit == 1 && can_direct);   // blocks never overlap partially
  const TnOut o = {direct, C, rs, cs, 0, 0, 0, nullptr, 0};
  const int pidx = prof_start(ctx, 0, m, k, N);
#define TN_NT(NTV)                                                                                                       \
  case NTV:                                                                                                              \
    if (waves == 8)                                                                                                      \
      HFMI_TRY((tn_dispatch_mt<NTV, 8>(ctx, mt, tr, r4, A, lda, m, B, ldb, k, Npad, chunk, nrb, nsplit, part, mpad, kpad, o))); \
    else                                                                                                                 \
      HFMI_TRY((tn_dispatch_mt<NTV, 4>(ctx, mt, tr, 0, A, lda, m, B, ldb, k, Npad, chunk, nrb, nsplit, part, mpad, kpad, o))); \
    break;
  switch (nt) {
    TN_NT(1) TN_NT(2) TN_NT(3) TN_NT(4) TN_NT(5) TN_NT(6) TN_NT(7) TN_NT(8) TN_NT(9) TN_NT(10) TN_NT(11) TN_NT(12)
    TN_NT(13) TN_NT(14) TN_NT(15) TN_NT(16)
    default:
      HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_tn: panel too wide (%d)", k);
  }
#undef TN_NT
  prof_stop(ctx, pidx);
  if (direct) return HFMI_OK;
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

