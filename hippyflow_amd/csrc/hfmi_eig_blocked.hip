// Whole-GPU symmetric eigensolver for 256 < n <= 16384: the n x n Gram problem of the deterministic POD (la.eigh(G),
// PODProjector.py:812-833 -- any number of snapshots; dataGenerator.py:278-279 hands it the whole training set), i.e. LAPACK
// dsyevd's algorithm family spread over the 256 compute units instead of the one-workgroup kernels of hfmi_eig_dc.hip:
//
//   1. panel Householder tridiagonalisation (dsytrd / dlatrd's recurrences, panels of EB_NB columns).  A column costs TWO
//      launches, both over many workgroups and without a host round trip:
//        k_tri_a  finalises the previous column of W (w = tau (A v - V W^T v - W V^T v) + alpha v; alpha needs v . w', which is
//                 taken from the identity v^T w' = tau (v^T A v - 2 (V^T v).(W^T v)), so no second device-wide reduction), then
//                 forms column j of the reduced matrix, a_j - V W_j^T - W V_j^T, and the partial sums of its norm;
//        k_tri_b  every workgroup derives the reflector scalars from those partial sums (same arithmetic, same bits), keeps v in
//                 LDS, and its waves take whole columns of the (un-updated) trailing block and of the panel's V and W:
//                 y = A v, V^T v, W^T v as plain dot products along contiguous columns (A is symmetric: A v = A^T v), each
//                 summed by ONE wave in a fixed order -- no partial sums cross workgroups except the scalar v . y.
//      The trailing rank-2 EB_NB update A <- A - V W^T - W V^T runs on the fp64 MFMA (k_dgemm, both products in one pass).
//   2. divide and conquer on the tridiagonal matrix: the couplings of the upper tree levels are torn first, the 2^Lf leaves of at
//      most 256 rows are solved by the one-workgroup kernel of hfmi_eig_dc.hip, one leaf per compute unit (k_dc_batch); the upper
//      merges run level by level over the whole GPU: rank sort + deflation (dlaed2's rules; the rotation scan is the only
//      sequential piece, in LDS), one WAVE per root of the secular equation, Gu-Eisenstat vector, eigenvectors of the rank-one
//      problem, and Q <- Q S on the MFMA.
//   3. back-transformation with compact-WY block reflectors: Gram matrices of the panels (one batched MFMA product), the
//      triangular factors, Y = V T, then per panel Z <- Z - Y (V^T Z): two MFMA products.
//   4. eigenvalues sorted on the host (n numbers), eigenvectors permuted + transposed to the row-major output on the device.
//
// Every reduction has a fixed order: results are bit-reproducible from run to run.  tests/helpers/eig_blocked_twin.py is the numpy
// twin (same recurrences, same tearing); tests/test_eig_blocked_twin.py pins it against numpy.linalg.eigh on the CPU.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <numeric>
#include <vector>

#include "hfmi_dc_common.h"

int launch_dc_leaves(hfmi_ctx* ctx, int n, int Lf, const double* dvec, const double* evec, double* Dout, double* Qbig, int64_t ldq,
                     int* fail);
int sym_eig_large_jacobi(hfmi_ctx* ctx, const double* host_T, int n, int sort_by_abs, double* host_d, double* host_V);

namespace {
constexpr int EB_NB = 64;          // panel width of the tridiagonalisation and of the block reflectors
constexpr int EB_MAXN = HFMI_EIG_MAXN;      // 16384: v of k_tri_b lives in LDS (128 KB of 160)
constexpr int GT = 64;             // k_dgemm: C tile
constexpr int GK = 16;             // reduction depth of an LDS stage
constexpr int GLD = 80;            // LDS row stride: = 16 mod 32 doubles, so the four k-rows of an MFMA operand fall into two bank halves

// ------------------------------------------------------------------------------------------------ fp64 MFMA GEMM
// element (t, k) of an operand tile, t = the operand's OWN index (row of op(A) / column of op(B)), for thread tid, slot u of 4
template <bool KC>
__device__ __forceinline__ void tile_idx(int tid, int u, int& t, int& k) {
  if (KC) {           // the reduction index is the contiguous one in memory
    k = tid & 15;
    t = (tid >> 4) + 16 * u;
  } else {            // the operand's own index is contiguous
    t = tid & 63;
    k = (tid >> 6) + 4 * u;
  }
}
template <bool KC>
__device__ __forceinline__ void tile_load(const double* __restrict__ X, int64_t ld, int t0, int k0, int Tdim, int Kdim, int tid,
                                          double (&r)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int t, k;
    tile_idx<KC>(tid, u, t, k);
    const bool ok = t0 + t < Tdim && k0 + k < Kdim;
    const int64_t off = KC ? (int64_t)(k0 + k) + (int64_t)(t0 + t) * ld : (int64_t)(t0 + t) + (int64_t)(k0 + k) * ld;
    r[u] = ok ? X[off] : 0.0;
  }
}
template <bool KC>
__device__ __forceinline__ void tile_store(double (*s)[GLD], int tid, const double (&r)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int t, k;
    tile_idx<KC>(tid, u, t, k);
    s[k][t] = r[u];
  }
}
// C (M x N) = alpha (op(A) op(B) + op(A2) op(B2)) + beta C, everything column-major.  op(A) = A (M x K, lda) or A^T (A is K x M);
// op(B) = B (K x N) or B^T (B is N x K); the second product (K2 columns, same shapes and leading dimensions) is optional
// (K2 = 0) -- it makes the symmetric rank-2k update one pass over C.  blockIdx.z = batch index (element strides sA, sB, sC).
// 64 x 64 tile per workgroup, 4 waves of 32 x 32 (2 x 2 MFMA 16x16x4 tiles), operands staged k-major through two LDS buffers.
// The MFMA's first operand carries the N index, the second the M index: the accumulator then holds 16 consecutive rows of a
// column per register, and the stores of C are 128-byte runs.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_dgemm(int M, int N, int K, double alpha, const double* __restrict__ A, int64_t lda,
                                               const double* __restrict__ B, int64_t ldb, int K2, const double* __restrict__ A2,
                                               const double* __restrict__ B2, double beta, double* __restrict__ C, int64_t ldc,
                                               int64_t sA, int64_t sB, int64_t sC) {
  __shared__ double s_a[2][GK][GLD], s_b[2][GK][GLD];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, li = l & 15, lk = l >> 4;
  const int wm = w & 1, wn = w >> 1;
  const int i0 = blockIdx.x * GT, j0 = blockIdx.y * GT;
  A += (int64_t)blockIdx.z * sA;
  B += (int64_t)blockIdx.z * sB;
  C += (int64_t)blockIdx.z * sC;
  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};
  const int nk1 = (K + GK - 1) / GK, nk2 = (K2 + GK - 1) / GK, nk = nk1 + nk2;
  double ra[4], rb[4];
  auto gload = [&](int kt) {
    const bool second = kt >= nk1;
    const double* Ap = second ? A2 : A;
    const double* Bp = second ? B2 : B;
    const int Kc = second ? K2 : K, k0 = (second ? kt - nk1 : kt) * GK;
    tile_load<TA>(Ap, lda, i0, k0, M, Kc, tid, ra);
    tile_load<!TB>(Bp, ldb, j0, k0, N, Kc, tid, rb);
  };
  if (nk > 0) {
    gload(0);
    tile_store<TA>(s_a[0], tid, ra);
    tile_store<!TB>(s_b[0], tid, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int k4 = 0; k4 < GK / 4; ++k4) {
      double af[2], bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[mi] = s_a[buf][k4 * 4 + lk][wm * 32 + mi * 16 + li];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bf[ni] = s_b[buf][k4 * 4 + lk][wn * 32 + ni * 16 + li];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = MFMA_F64(bf[ni], af[mi], acc[mi][ni]);
    }
    if (kt + 1 < nk) {
      tile_store<TA>(s_a[buf ^ 1], tid, ra);
      tile_store<!TB>(s_b[buf ^ 1], tid, rb);
    }
    __syncthreads();
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = i0 + wm * 32 + mi * 16 + li, j = j0 + wn * 32 + ni * 16 + lk + 4 * reg;
        if (i < M && j < N) {
          double* cp = C + (int64_t)i + (int64_t)j * ldc;
          double v = alpha * acc[mi][ni][reg];
          if (beta != 0.0) v = fma(beta, *cp, v);
          *cp = v;
        }
      }
}

// ------------------------------------------------------------------------------------------------ fp64 MFMA GEMM, software-pipelined
// The same contract as k_dgemm for the large products (trailing rank-2k updates, Q S of the upper merges, the block reflectors of
// the back-transformation), built for the MFMA pipe to stay busy:
//   * workgroup tile BM x 128 (BM = 128 | 64), 4 waves of (BM / 2) x 64: 4 x 4 (2 x 4) accumulator tiles per wave, so a reduction
//     step of 4 costs 4 + 4 (2 + 4) eight-byte LDS reads for 16 (8) MFMAs of 64 cycles each;
//   * the operand fragments of step k + 1 are read into a second register set while the MFMAs of step k issue (register double
//     buffering: the 128 x 128 variant of round 5 read its fragments right before their use and lost to the 64 x 64 one);
//   * the next 16-deep slab of both operands travels global -> registers (16-byte loads) during the whole slab, and is written to
//     the other LDS buffer after the last MFMA: one barrier per 64 (32) MFMAs of a wave;
//   * LDS layout by the operand's memory layout, both conflict-free for the 16-byte stores and the 8-byte fragment reads:
//       own index contiguous in memory -> k-major rows of BT + 16 doubles (= 16 mod 32: the four k-rows of a fragment fall into
//       two bank halves), reduction index contiguous -> own-index-major rows of 18 doubles (36 li mod 64 are 16 distinct multiples
//       of 4: a fragment's 16 rows x 2 columns cover the 64 banks exactly once).
// 16-byte loads need 16-byte aligned operands (even offsets and leading dimensions): `vec` says so, else 8-byte loads.
constexpr int PK = 16;              // reduction depth of an LDS stage
constexpr int PS_KC = 18;           // row stride (doubles) of a stage whose rows are the operand's own index
template <bool KC, int BT>
struct pstage {
  static constexpr int ND2 = BT * PK / 2 / 256;            // 16-byte pieces per thread
  static constexpr int LDK = BT + 16;                      // k-major row stride
  static constexpr int DOUBLES = KC ? BT * PS_KC : PK * LDK;
  // piece u of thread tid: own index t (pair t, t + 1 if !KC), reduction index k (pair k, k + 1 if KC)
  static __device__ __forceinline__ void idx(int tid, int u, int& t, int& k) {
    if (KC) {
      k = 2 * (tid & 7);
      t = (tid >> 3) + 32 * u;
    } else {
      t = 2 * (tid & (BT / 2 - 1));
      k = tid / (BT / 2) + (512 / BT) * u;
    }
  }
  // FAST: the whole slab lies inside the operand and is 16-byte aligned -- plain 16-byte loads, no predicate.  Otherwise 8-byte loads
  // from clamped (always valid) addresses and a select: no lane-divergent branch in either form (a predicated load that shares
  // its destination with another load makes the compiler wait for the first one: eight serialised round trips per slab).
  template <bool FAST>
  static __device__ __forceinline__ void load(const double* __restrict__ X, int64_t ld, int t0, int k0, int Tdim, int Kdim, int tid,
                                              d2 (&r)[ND2], unsigned& valid) {
    valid = ~0u;
#pragma unroll
    for (int u = 0; u < ND2; ++u) {
      int t, k;
      idx(tid, u, t, k);
      const int tt = t0 + t, kk = k0 + k;
      if (FAST) {
        const int64_t off = KC ? (int64_t)kk + (int64_t)tt * ld : (int64_t)tt + (int64_t)kk * ld;
        r[u] = *(const d2*)(X + off);
      } else {
        const int tt1 = KC ? tt : tt + 1, kk1 = KC ? kk + 1 : kk;
        const bool ok0 = tt < Tdim && kk < Kdim, ok1 = tt1 < Tdim && kk1 < Kdim;
        const int tc0 = min(tt, Tdim - 1), kc0 = min(kk, Kdim - 1), tc1 = min(tt1, Tdim - 1), kc1 = min(kk1, Kdim - 1);
        r[u].x = X[KC ? (int64_t)kc0 + (int64_t)tc0 * ld : (int64_t)tc0 + (int64_t)kc0 * ld];
        r[u].y = X[KC ? (int64_t)kc1 + (int64_t)tc1 * ld : (int64_t)tc1 + (int64_t)kc1 * ld];
        if (!ok0) valid &= ~(1u << (2 * u));
        if (!ok1) valid &= ~(2u << (2 * u));
      }
    }
  }
  // (the zeroing of out-of-range entries happens here, after the MFMAs of the slab that hid the loads)
  static __device__ __forceinline__ void store(double* __restrict__ s, int tid, const d2 (&r)[ND2], unsigned valid) {
#pragma unroll
    for (int u = 0; u < ND2; ++u) {
      int t, k;
      idx(tid, u, t, k);
      d2 v = r[u];
      if (!((valid >> (2 * u)) & 1u)) v.x = 0.0;
      if (!((valid >> (2 * u)) & 2u)) v.y = 0.0;
      if (KC) *(d2*)(s + t * PS_KC + k) = v;
      else *(d2*)(s + k * LDK + t) = v;
    }
  }
  // fragment element (own index t, reduction index k)
  static __device__ __forceinline__ double frag(const double* __restrict__ s, int t, int k) { return KC ? s[t * PS_KC + k] : s[k * LDK + t]; }
};
template <bool TA, bool TB, int BM>
__global__ __launch_bounds__(256, 2) void k_dgemm_p(int M, int N, int K, double alpha, const double* __restrict__ A, int64_t lda,
                                                    const double* __restrict__ B, int64_t ldb, int K2, const double* __restrict__ A2,
                                                    const double* __restrict__ B2, double beta, double* __restrict__ C, int64_t ldc,
                                                    int64_t sA, int64_t sB, int64_t sC, int vec, int lower) {
  constexpr int BN = 128, MI = BM / 32, NI = 4;
  typedef pstage<TA, BM> SA;          // op(A) = A^T: the reduction index is the contiguous one
  typedef pstage<!TB, BN> SB;
  extern __shared__ __attribute__((aligned(16))) double s_p[];
  auto s_a = [&](int b) { return s_p + b * SA::DOUBLES; };
  auto s_b = [&](int b) { return s_p + 2 * SA::DOUBLES + b * SB::DOUBLES; };
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, li = l & 15, lk = l >> 4;
  const int wm = w & 1, wn = w >> 1;
  const int i0 = blockIdx.x * BM, j0 = blockIdx.y * BN;
  // symmetric update, lower part only: `lower` - 1 is the position of C's first row / column inside its 128 x 128 block of the matrix;
  // a tile is skipped when it lies entirely above the diagonal 128-blocks (the lower-triangle products read those blocks whole)
  if (lower && ((i0 + lower - 1 + BM - 1) >> 7) < ((j0 + lower - 1) >> 7)) return;
  A += (int64_t)blockIdx.z * sA;
  B += (int64_t)blockIdx.z * sB;
  C += (int64_t)blockIdx.z * sC;
  if (K2 > 0) {
    A2 += (int64_t)blockIdx.z * sA;
    B2 += (int64_t)blockIdx.z * sB;
  }
  d4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};
  const int nk1 = (K + PK - 1) / PK, nk2 = (K2 + PK - 1) / PK, nk = nk1 + nk2;
  d2 ra[SA::ND2], rb[SB::ND2];
  unsigned va = ~0u, vb = ~0u;
  const bool inside = vec && i0 + BM <= M && j0 + BN <= N;      // workgroup-uniform
  auto gload = [&](int kt) {
    const bool second = kt >= nk1;
    const double* Ap = second ? A2 : A;
    const double* Bp = second ? B2 : B;
    const int Kc = second ? K2 : K, k0 = (second ? kt - nk1 : kt) * PK;
    if (inside && k0 + PK <= Kc) {
      SA::template load<true>(Ap, lda, i0, k0, M, Kc, tid, ra, va);
      SB::template load<true>(Bp, ldb, j0, k0, N, Kc, tid, rb, vb);
    } else {
      SA::template load<false>(Ap, lda, i0, k0, M, Kc, tid, ra, va);
      SB::template load<false>(Bp, ldb, j0, k0, N, Kc, tid, rb, vb);
    }
  };
  if (nk > 0) {
    gload(0);
    SA::store(s_a(0), tid, ra, va);
    SB::store(s_b(0), tid, rb, vb);
  }
  __syncthreads();
  const int ta0 = wm * (BM / 2) + li, tb0 = wn * 64 + li;
  for (int kt = 0; kt < nk; ++kt) {
    const double* __restrict__ pa = s_a(kt & 1);
    const double* __restrict__ pb = s_b(kt & 1);
    if (kt + 1 < nk) gload(kt + 1);
    double fa[2][MI], fb[2][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fa[0][mi] = SA::frag(pa, ta0 + 16 * mi, lk);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fb[0][ni] = SB::frag(pb, tb0 + 16 * ni, lk);
#pragma unroll
    for (int k4 = 0; k4 < PK / 4; ++k4) {
      const int cur = k4 & 1, nxt = cur ^ 1;
      if (k4 + 1 < PK / 4) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[nxt][mi] = SA::frag(pa, ta0 + 16 * mi, 4 * (k4 + 1) + lk);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[nxt][ni] = SB::frag(pb, tb0 + 16 * ni, 4 * (k4 + 1) + lk);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = MFMA_F64(fb[cur][ni], fa[cur][mi], acc[mi][ni]);
    }
    if (kt + 1 < nk) {
      SA::store(s_a((kt + 1) & 1), tid, ra, va);
      SB::store(s_b((kt + 1) & 1), tid, rb, vb);
    }
    __syncthreads();
  }
  // epilogue: lane (li, lk), register reg of tile (mi, ni) holds C[i0 + wm BM/2 + 16 mi + li, j0 + wn 64 + 16 ni + lk + 4 reg].
  // beta != 0: the old values are fetched two column tiles at a time -- 8 MI loads per lane in flight (from clamped, always valid
  // addresses: no predicate on a load), then the same number of stores -- two round trips per tile instead of one per column
  const int ib = i0 + wm * (BM / 2) + li, jb = j0 + wn * 64 + lk;
#pragma unroll
  for (int nh = 0; nh < NI; nh += 2) {
    double old[2][4][MI];
    if (beta != 0.0) {
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int i = min(ib + 16 * mi, M - 1), j = min(jb + 16 * (nh + n2) + 4 * reg, N - 1);
            old[n2][reg][mi] = C[(int64_t)i + (int64_t)j * ldc];
          }
    }
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int i = ib + 16 * mi, j = jb + 16 * (nh + n2) + 4 * reg;
          double v = alpha * acc[mi][nh + n2][reg];
          if (beta != 0.0) v = fma(beta, old[n2][reg][mi], v);
          if (i < M && j < N) C[(int64_t)i + (int64_t)j * ldc] = v;
        }
  }
}

struct gemm_desc {
  bool ta, tb;
  int M, N, K;
  double alpha, beta;
  const double *A, *B;
  int64_t lda, ldb;
  double* C;
  int64_t ldc;
  int K2 = 0;
  const double *A2 = nullptr, *B2 = nullptr;
  int batch = 1;
  int64_t sA = 0, sB = 0, sC = 0;
  int lower = 0;            // symmetric update: > 0 = 1 + (first row of C mod 128): tiles above the diagonal 128-blocks are skipped (pipelined kernel only)
};
template <bool TA, bool TB, int BM>
int launch_dgemm_p(hfmi_ctx* ctx, const gemm_desc& g, int vec) {
  typedef pstage<TA, BM> SA;
  typedef pstage<!TB, 128> SB;
  const size_t lds = (size_t)2 * (SA::DOUBLES + SB::DOUBLES) * sizeof(double);
  static bool attr_set = false;       // (one flag per instantiation)
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_dgemm_p<TA, TB, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const dim3 grid((g.M + BM - 1) / BM, (g.N + 127) / 128, g.batch), block(256);
  hipLaunchKernelGGL((k_dgemm_p<TA, TB, BM>), grid, block, lds, ctx->stream, g.M, g.N, g.K, g.alpha, g.A, g.lda, g.B, g.ldb, g.K2, g.A2, g.B2,
                     g.beta, g.C, g.ldc, g.sA, g.sB, g.sC, vec, g.lower);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
int launch_dgemm(hfmi_ctx* ctx, const gemm_desc& g) {
  if (g.M <= 0 || g.N <= 0) return HFMI_OK;
  static const int pipelined = [] {      // HFMI_EIG_GEMM = 0: the 64 x 64 kernel of round 5 everywhere (A/B)
    const char* e = getenv("HFMI_EIG_GEMM");
    return e ? atoi(e) : 1;
  }();
  // the pipelined kernel where its tiles fill the chip: 128 x 128 from 192 tiles on, 64 x 128 from 192 of those; the 64 x 64 kernel below
  // for the small products (panel factors, merges of small nodes)
  const int64_t t128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch;
  const int64_t t64 = (int64_t)((g.M + 63) / 64) * ((g.N + 127) / 128) * g.batch;
  if (pipelined && (t128 >= 192 || t64 >= 192) && g.K + g.K2 >= 32) {
    auto even = [](int64_t v) { return (v & 1) == 0; };
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const int vec = even(g.lda) && even(g.ldb) && even(g.sA) && even(g.sB) && al16(g.A) && al16(g.B) && (g.K2 == 0 || (al16(g.A2) && al16(g.B2)));
    const bool big = t128 >= 192;
#define EB_GEMM_P(TAV, TBV) (big ? launch_dgemm_p<TAV, TBV, 128>(ctx, g, vec) : launch_dgemm_p<TAV, TBV, 64>(ctx, g, vec))
    if (!g.ta && !g.tb) return EB_GEMM_P(false, false);
    if (g.ta && !g.tb) return EB_GEMM_P(true, false);
    if (!g.ta && g.tb) return EB_GEMM_P(false, true);
    return EB_GEMM_P(true, true);
#undef EB_GEMM_P
  }
  const dim3 grid((g.M + GT - 1) / GT, (g.N + GT - 1) / GT, g.batch), block(256);
#define EB_GEMM(TAV, TBV)                                                                                                              \
  hipLaunchKernelGGL((k_dgemm<TAV, TBV>), grid, block, 0, ctx->stream, g.M, g.N, g.K, g.alpha, g.A, g.lda, g.B, g.ldb, g.K2, g.A2, g.B2, \
                     g.beta, g.C, g.ldc, g.sA, g.sB, g.sC)
  if (!g.ta && !g.tb) EB_GEMM(false, false);
  else if (g.ta && !g.tb) EB_GEMM(true, false);
  else if (!g.ta && g.tb) EB_GEMM(false, true);
  else EB_GEMM(true, true);
#undef EB_GEMM
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// ------------------------------------------------------------------------------------------------ load: symmetrise, scale
// raw: the caller's n x n row-major matrix.  A (column-major, ld) = (raw + raw^T) / 2; per-tile max |entry| -> pmax
__global__ __launch_bounds__(256) void k_sym_load(const double* __restrict__ raw, int n, double* __restrict__ A, int64_t ld,
                                                  double* __restrict__ pmax) {
  __shared__ double t1[32][33], t2[32][33];
  __shared__ double s_red[4];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = ty + 8 * q;
    t1[a][tx] = (i0 + a < n && j0 + tx < n) ? raw[(size_t)(i0 + a) * n + (j0 + tx)] : 0.0;     // raw[i][j]
    t2[a][tx] = (j0 + a < n && i0 + tx < n) ? raw[(size_t)(j0 + a) * n + (i0 + tx)] : 0.0;     // raw[j][i]
  }
  __syncthreads();
  double mx = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int b = ty + 8 * q;            // j = j0 + b, i = i0 + tx
    const double v = 0.5 * (t1[tx][b] + t2[b][tx]);
    if (i0 + tx < n && j0 + b < n) {
      A[(size_t)(i0 + tx) + (size_t)(j0 + b) * ld] = v;
      mx = fmax(mx, v == v ? fabs(v) : INFINITY);
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) pmax[blockIdx.y * gridDim.x + blockIdx.x] = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
}
// power-of-two scaling to max |entry| in [1, 2): norms cannot overflow / underflow, the eigenvalues scale back exactly
__global__ __launch_bounds__(1024) void k_scale_exp(const double* __restrict__ pmax, int np, int* __restrict__ sexp_out) {
  __shared__ double s_red[16];
  double mx = 0.0;
  for (int i = threadIdx.x; i < np; i += 1024) mx = fmax(mx, pmax[i]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    double amax = 0.0;
    for (int q = 0; q < 16; ++q) amax = fmax(amax, s_red[q]);
    int sexp = 0;
    if (amax > 0.0 && isfinite(amax)) sexp = ilogb(amax);
    sexp_out[0] = sexp;
    sexp_out[1] = isfinite(amax) ? 0 : 1;        // (fmax drops NaNs: the per-tile pass counts them into the maximum as +inf)
  }
}
__global__ __launch_bounds__(256) void k_scale_apply(double* __restrict__ A, int64_t ld, int n, const int* __restrict__ sexp) {
  const double sc = ldexp(1.0, -sexp[0]);
  const bool bad = sexp[1] != 0;               // non-finite input: the solve runs on zeros (bounded work), the host reports the error
  double* col = A + (size_t)blockIdx.x * ld;
  for (int r = threadIdx.x; r < n; r += 256) col[r] = bad ? 0.0 : col[r] * sc;
}

// ------------------------------------------------------------------------------------------------ tridiagonalisation
struct tri_args {
  int n, j, jj, p0, mode;       // mode bit 0: finalise W[:, jj - 1] (products of step j - 1); bit 1: form column j
  int nr;                       // rows of a column that exist and are zero beyond n: n rounded up to 128
  int64_t ld;                   // leading dimension: nr, or nr + 144 where nr is a power of two (see sym_eig_large)
  double* A;
  double* Vh;                   // reflectors, n x n (column j = v_j with explicit zeros and the unit entry)
  double* W;                    // panel W, n x EB_NB
  double* colbuf;               // column j of the reduced matrix (rows j ..)
  double* ybuf;                 // A v of the last step
  double *x1, *x2;              // V^T v, W^T v of the last step
  double* pvy;                  // partial sums of v . y, one per workgroup of the last k_tri_b
  int npvy;
  double* pn;                   // partial sums of |col[j + 2 ..]|^2, one per workgroup of k_tri_a
  int npn;
  double *dvec, *evec, *tauv;
  double* part;                 // k_tri_bs: partial products, part[k * ld + row], k < nb
  int nb;                       // 128-row blocks of the trailing matrix seen from rs2 = (j + 1) rounded down to 128
};

// 64 rows per workgroup (row r = j + 64 blockIdx.x + lane), the panel columns dealt to the 8 waves (wave w takes k = w, w + 8, ...):
// every thread has at most 8 (V, W) pairs to fetch and issues all of them before anything waits, so a column costs a couple of
// memory latencies instead of a 63-step dependent loop; the 8 partial sums of a row meet in LDS and wave 0 finishes the row.
constexpr int TA_WAVES = 8;
constexpr int TA_KPT = EB_NB / TA_WAVES;      // panel columns per thread
// SLOTS: the last step's products were made by k_tri_bs -- y[r] is the sum of the nb <= 64 partial values in part[k * ld + r] (the slots
// dealt to the 8 waves, at most 8 loads per lane, the 8 sums of a row added in wave order; row j, which every workgroup needs, by the
// same tree: same bits everywhere), and v . y the sum of its per-tile terms v_I . (A_IJ v_J) (pvy, up to 2080 of them).
template <bool SLOTS>
__global__ __launch_bounds__(64 * TA_WAVES) void k_tri_a(tri_args p) {
  __shared__ double s_x1[EB_NB], s_x2[EB_NB], s_vj[EB_NB], s_wj[EB_NB];
  __shared__ double s_py[TA_WAVES][64], s_pc[TA_WAVES][64];
  __shared__ double s_yq[SLOTS ? TA_WAVES : 1][64], s_yj[TA_WAVES];
  __shared__ double s_alpha;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, n = p.n, j = p.j, jj = p.jj;
  const int64_t ld = p.ld;
  const double* __restrict__ Vp = p.Vh + (size_t)p.p0 * ld;
  double* __restrict__ W = p.W;
  const bool fin = (p.mode & 1) && jj > 0, col = (p.mode & 2) != 0;
  const int kf = jj - 1;                       // the column being finalised
  const int kmax = fin ? kf : jj;              // panel columns that are complete
  const int r = j + blockIdx.x * 64 + l;
  const bool live = r < n;
  // this thread's (V, W) pairs first: nothing below waits for them until the barrier
  double vv[TA_KPT], ww[TA_KPT];
#pragma unroll
  for (int u = 0; u < TA_KPT; ++u) {
    const int k = w + TA_WAVES * u;
    const bool ok = live && k < kmax;
    vv[u] = ok ? Vp[(size_t)r + (size_t)k * ld] : 0.0;
    ww[u] = ok ? W[(size_t)r + (size_t)k * ld] : 0.0;
  }
  const double tprev = fin ? p.tauv[j - 1] : 0.0;
  // ... and what wave 0 needs to finish the row: requested now, used after the second barrier
  double vr_pre = 0.0, y_pre = 0.0, a_pre = 0.0;
  if (w == 0 && live) {
    if (fin) {
      vr_pre = Vp[(size_t)r + (size_t)kf * ld];
      if (!SLOTS) y_pre = p.ybuf[r];
    }
    if (col) a_pre = p.A[(size_t)r + (size_t)j * ld];
  }
  if (SLOTS) {
    if (fin) {
      double yq[8], jq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = w + TA_WAVES * u;
        yq[u] = (live && k < p.nb) ? p.part[(size_t)k * ld + r] : 0.0;
        jq[u] = (col && k < p.nb) ? p.part[(size_t)k * ld + j] : 0.0;
      }
      s_yq[w][l] = ((yq[0] + yq[1]) + (yq[2] + yq[3])) + ((yq[4] + yq[5]) + (yq[6] + yq[7]));
      if (l == 0) s_yj[w] = ((jq[0] + jq[1]) + (jq[2] + jq[3])) + ((jq[4] + jq[5]) + (jq[6] + jq[7]));
    }
    __syncthreads();
    if (fin && w == 0)
      y_pre = ((s_yq[0][l] + s_yq[1][l]) + (s_yq[2][l] + s_yq[3][l])) + ((s_yq[4][l] + s_yq[5][l]) + (s_yq[6][l] + s_yq[7][l]));
  }
  if (w == 0) {
    if (fin) {
      // the partial sums of v . y (up to 512 workgroups of k_tri_b / 2080 tiles of k_tri_bs): all loads of a lane in flight at once
      // (a loop over a run-time count waits for every load before it issues the next: 8 x an L2 round trip per column at n = 4096)
      constexpr int NS = SLOTS ? 33 : 8;
      double s8[NS];
#pragma unroll
      for (int u = 0; u < NS; ++u) s8[u] = (l + 64 * u < p.npvy) ? p.pvy[l + 64 * u] : 0.0;
      double s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
      if (SLOTS) {
#pragma unroll
        for (int u = 8; u < NS; ++u) s += s8[u];
      }
      const double x1v = l < kf ? p.x1[l] : 0.0, x2v = l < kf ? p.x2[l] : 0.0;
      const double vjk = (col && l < kf) ? Vp[(size_t)j + (size_t)l * ld] : 0.0, wjk = (col && l < kf) ? W[(size_t)j + (size_t)l * ld] : 0.0;
      const double vjl = col ? Vp[(size_t)j + (size_t)kf * ld] : 0.0;
      const double yj = !col ? 0.0 : SLOTS ? ((s_yj[0] + s_yj[1]) + (s_yj[2] + s_yj[3])) + ((s_yj[4] + s_yj[5]) + (s_yj[6] + s_yj[7])) : p.ybuf[j];
      const double vy = wave_sum(s);
      s_x1[l] = x1v;
      s_x2[l] = x2v;
      const double s12 = wave_sum(x1v * x2v);
      const double alpha = -0.5 * tprev * tprev * (vy - 2.0 * s12);      // -tau/2 (v . w'),  v . w' = tau (v.y - 2 x1.x2)
      if (l == 0) s_alpha = alpha;
      if (col) {      // row j of V and W, the finalised entry W[j, kf] included: every workgroup computes it for itself
        const double yc = wave_sum(fma(vjk, x2v, wjk * x1v));
        const double wjl = fma(tprev, yj - yc, alpha * vjl);
        s_vj[l] = l < kf ? vjk : (l == kf ? vjl : 0.0);
        s_wj[l] = l < kf ? wjk : (l == kf ? wjl : 0.0);
      }
    } else if (col) {
      s_vj[l] = l < jj ? Vp[(size_t)j + (size_t)l * ld] : 0.0;
      s_wj[l] = l < jj ? W[(size_t)j + (size_t)l * ld] : 0.0;
    }
  }
  __syncthreads();
  double accy = 0.0, accc = 0.0;
#pragma unroll
  for (int u = 0; u < TA_KPT; ++u) {
    const int k = w + TA_WAVES * u;
    if (k < kmax) {        // wave-uniform
      if (fin) accy = fma(vv[u], s_x2[k], fma(ww[u], s_x1[k], accy));
      if (col) accc = fma(vv[u], s_wj[k], fma(ww[u], s_vj[k], accc));
    }
  }
  s_py[w][l] = accy;
  s_pc[w][l] = accc;
  __syncthreads();
  if (w != 0) return;
  accy = ((s_py[0][l] + s_py[1][l]) + (s_py[2][l] + s_py[3][l])) + ((s_py[4][l] + s_py[5][l]) + (s_py[6][l] + s_py[7][l]));
  accc = ((s_pc[0][l] + s_pc[1][l]) + (s_pc[2][l] + s_pc[3][l])) + ((s_pc[4][l] + s_pc[5][l]) + (s_pc[6][l] + s_pc[7][l]));
  double colv = 0.0;
  if (live) {
    if (fin) {
      const double vr = vr_pre;
      const double wr = fma(tprev, y_pre - accy, s_alpha * vr);
      W[(size_t)r + (size_t)kf * ld] = wr;
      if (col) accc = fma(vr, s_wj[kf], fma(wr, s_vj[kf], accc));
    }
    if (col) {
      colv = a_pre - accc;
      p.colbuf[r] = colv;
      if (r == j) p.dvec[j] = colv;
    }
  }
  if (col) {
    const double part = wave_sum((live && r >= j + 2) ? colv * colv : 0.0);
    if (l == 0) p.pn[blockIdx.x] = part;
  }
}

// 512 threads; dynamic LDS: v on rows [rs, ld), rs = (j + 1) rounded down to 64
template <int UNR, int CB>
__global__ __launch_bounds__(512) void k_tri_b(tri_args p) {
  extern __shared__ __attribute__((aligned(16))) double s_v[];
  __shared__ double s_part[8];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, n = p.n, j = p.j, jj = p.jj;
  const int64_t ld = p.ld;
  const int rs = (j + 1) & ~63;
  const int nA = n - j - 1, nc = nA + 2 * jj;
  const int npair = (p.nr - rs) >> 1;
  auto column = [&](int q) -> const d2* {
    const double* colp;
    if (q < nA) colp = p.A + (size_t)(j + 1 + q) * ld;
    else if (q < nA + jj) colp = p.Vh + (size_t)(p.p0 + q - nA) * ld;
    else colp = p.W + (size_t)(q - nA - jj) * ld;
    return (const d2*)(colp + rs);
  };
  // the first 2 KB of this wave's first column are requested before anything else: they do not depend on the reflector, and
  // at n <= 1024 (a column is one or two such requests) the kernel is a chain of memory round trips -- this one now runs
  // beside those of the reflector scalars and of v
  const int qfirst = blockIdx.x * 8 + w;
  d2 pre[4];
  {
    const d2* __restrict__ c2 = column(qfirst < nc ? qfirst : 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) pre[u] = (qfirst < nc && l + 64 * u < npair) ? c2[l + 64 * u] : d2{0.0, 0.0};
  }
  // column j of the reduced matrix: every load of it is requested before the first use (a loop over a run-time count would wait
  // for each load before issuing the next: 8 x an L2 round trip per column at n = 4096)
  double cb[CB];           // CB = 8: n <= 4096, 16: n <= 8192, 32: n <= 16384
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    const int r = rs + tid + 512 * u;
    cb[u] = (r > j + 1 && r < n) ? p.colbuf[r] : 0.0;
  }
  const double alpha0 = p.colbuf[j + 1];
  // one partial sum per 64 rows: at most 64 (one per lane) up to n = 4096, two per lane up to 8192, four up to 16384
  double pnl = l < p.npn ? p.pn[l] : 0.0;
#pragma unroll
  for (int u = 1; u < CB / 8; ++u) pnl += l + 64 * u < p.npn ? p.pn[l + 64 * u] : 0.0;
  const double xn2 = wave_sum(pnl);
  double tau = 0.0, beta = alpha0, scl = 0.0;
  if (xn2 > 1e-280) {      // entries are scaled to O(1): below this the column is zero to any precision that matters
    const double nrm = sqrt(fma(alpha0, alpha0, xn2));
    beta = -copysign(nrm, alpha0);
    tau = (beta - alpha0) / beta;
    scl = 1.0 / (alpha0 - beta);
  }
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    const int r = rs + tid + 512 * u;
    if (r < p.nr) s_v[r - rs] = (r <= j || r >= n) ? 0.0 : (r == j + 1 ? 1.0 : cb[u] * scl);
  }
  if (blockIdx.x == 0 && tid == 0) {
    p.evec[j] = beta;
    p.tauv[j] = tau;
  }
  __syncthreads();
  {      // column j of the reflector matrix (explicit zeros above the unit entry), this workgroup's share of the rows, from LDS
    double* __restrict__ vcol = p.Vh + (size_t)j * ld;
    for (int r = blockIdx.x * 512 + tid; r < n; r += gridDim.x * 512) vcol[r] = r < rs ? 0.0 : s_v[r - rs];
  }
  const d2* __restrict__ v2 = (const d2*)s_v;
  double vyp = 0.0;
  for (int q = qfirst; q < nc; q += gridDim.x * 8) {
    const d2* __restrict__ c2 = column(q);
    // UNR 16-byte loads per lane in flight (UNR KB per wave): with one column per wave at n = 4096 the product is bound by the
    // bytes in flight, not by the cache that holds the matrix (profiles/r05b: two loads in flight gave 3.7 TB/s)
    double acc[2 * UNR];
#pragma unroll
    for (int u = 0; u < 2 * UNR; ++u) acc[u] = 0.0;
    int i = l;
    if (q == qfirst) {       // wave-uniform: the prefetched head of the first column
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i + 64 * u < npair) {
          const d2 vv = v2[i + 64 * u];
          acc[2 * u] = fma(pre[u].x, vv.x, acc[2 * u]);
          acc[2 * u + 1] = fma(pre[u].y, vv.y, acc[2 * u + 1]);
        }
      }
      i += 256;
    }
    for (; i + 64 * (UNR - 1) < npair; i += 64 * UNR) {
      d2 x[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) x[u] = c2[i + 64 * u];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const d2 vv = v2[i + 64 * u];
        acc[2 * u] = fma(x[u].x, vv.x, acc[2 * u]);
        acc[2 * u + 1] = fma(x[u].y, vv.y, acc[2 * u + 1]);
      }
    }
    for (; i < npair; i += 64) {
      const d2 x = c2[i];
      const d2 vv = v2[i];
      acc[0] = fma(x.x, vv.x, acc[0]);
      acc[1] = fma(x.y, vv.y, acc[1]);
    }
#pragma unroll
    for (int st = 1; st < 2 * UNR; st *= 2)
#pragma unroll
      for (int u = 0; u + st < 2 * UNR; u += 2 * st) acc[u] += acc[u + st];
    const double y = wave_sum(acc[0]);
    if (q < nA) {
      if (l == 0) p.ybuf[j + 1 + q] = y;
      vyp = fma(y, s_v[j + 1 + q - rs], vyp);
    } else if (q < nA + jj) {
      if (l == 0) p.x1[q - nA] = y;
    } else {
      if (l == 0) p.x2[q - nA - jj] = y;
    }
  }
  if (l == 0) s_part[w] = vyp;
  __syncthreads();
  if (tid == 0) p.pvy[blockIdx.x] = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) + ((s_part[4] + s_part[5]) + (s_part[6] + s_part[7]));
}
// the last 2 x 2 block of the reduced matrix
// The same products from the LOWER TRIANGLE only (large trailing blocks: the stream of the matrix is what a column costs there, so
// half the bytes is nearly half the time).  128 x 128 tiles (I, J), I >= J, rows / columns counted from rs2 = (j + 1) rounded down
// to 128; one workgroup per tile: wave w takes columns w, w + 8, ... of the tile (one 16-byte load per lane and column), and a tile
// below the diagonal serves both  y_I += A_IJ v_J  (accumulated per lane over the wave's columns, the 8 waves summed through LDS in
// a fixed order) and  y_J += A_IJ^T v_I  (one wave sum per column).  Nothing is accumulated across workgroups: tile (I, J) leaves its
// two partial vectors in slots k = J (rows of I) and k = I (rows of J) of `part`, so that every row finds exactly nb partial values,
// one per slot, which the next k_tri_a<true> adds in a fixed order -- bit-reproducible.  The last 2 jj workgroups are the panel's columns of V and
// W (V^T v, W^T v): one column per workgroup, an eighth of the rows per wave.  v is not staged: tiles need 2 x 128 entries, taken
// from column j of the reduced matrix with the reflector's scaling (same arithmetic in every workgroup).
constexpr int TS = 128;
template <int CB>
__global__ __launch_bounds__(512) void k_tri_bs(tri_args p, int ntiles) {
  __shared__ double s_vi[TS], s_vj[TS];
  __shared__ double s_pr[8][TS];
  __shared__ double s_dot[8];
  // (the wave index as a SCALAR: the 16 column addresses of a lane are then one lane offset + scalar strides instead of 16 address pairs
  // in registers -- 108 -> fewer VGPRs, i.e. three resident workgroups per CU instead of two)
  const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = p.n, j = p.j, jj = p.jj;
  const int64_t ld = p.ld;
  const int rs2 = (j + 1) & ~(TS - 1);
  const bool is_tile = (int)blockIdx.x < ntiles;
  // ---- every load of this workgroup is requested before anything waits (round 6: the tile loads used to sit behind the wait
  // for the partial norms -- two dependent memory round trips per workgroup, 4.85 TB/s where the bare tile pattern reads 6.25).
  // The small loads go first: vmcnt retires in order, so waiting for them leaves the tile in flight.
  const double alpha0 = p.colbuf[j + 1];
  double pnv[CB / 8];
#pragma unroll
  for (int u = 0; u < CB / 8; ++u) pnv[u] = l + 64 * u < p.npn ? p.pn[l + 64 * u] : 0.0;
  // tile (I, J), I >= J: blockIdx = I (I + 1) / 2 + J
  int I = 0, J = 0;
  if (is_tile) {
    I = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= (int)blockIdx.x) ++I;
    while (I * (I + 1) / 2 > (int)blockIdx.x) --I;
    J = blockIdx.x - I * (I + 1) / 2;
  }
  const int r0 = rs2 + TS * I, c0 = rs2 + TS * J;
  // a panel column: x1[q] = V[:, q] . v  or  x2[q] = W[:, q] . v
  const int q = is_tile ? 0 : (int)blockIdx.x - ntiles;
  const int npair = (p.nr - rs2) >> 1;               // pairs of rows from rs2
  const int per = (npair + 7) >> 3;                  // per wave (<= 512: at most 8 per lane)
  // ONE straight-line load sequence for both kinds of workgroup (addresses selected, never a branch around a load: the waitcnt
  // pass treats a join behind divergent loads as "everything may be in flight" and waited for the small loads before the big ones)
  const double* colp = q < jj ? p.Vh + (size_t)(p.p0 + q) * ld : p.W + (size_t)(q - jj) * ld;
  const d2* __restrict__ c2 = (const d2*)(colp + rs2);
  const d2* __restrict__ b2 = (const d2*)(p.colbuf + rs2);
  const double* __restrict__ Ab = p.A + (size_t)r0 + (size_t)c0 * ld;
  // this thread's entry of column j for the tile's two row ranges (any valid address for the threads / workgroups without one)
  const double cvec = p.colbuf[is_tile ? (tid < TS ? r0 + tid : c0 + (tid & (TS - 1))) : rs2];
  d2 x[16];
  unsigned okmask = 0xffffu;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int u8 = u & 7;
    const int i = w * per + l + 64 * u8;
    const bool ok = l + 64 * u8 < per && i < npair;
    const int ic = min(i, npair - 1);
    const d2* src = is_tile ? (const d2*)(Ab + (size_t)(w + 8 * u) * ld) + l : (u < 8 ? c2 + ic : b2 + ic);
    x[u] = *src;
    if (!is_tile && !ok) okmask &= ~(1u << u);
  }
  __builtin_amdgcn_sched_barrier(0);
  double pnl = pnv[0];
#pragma unroll
  for (int u = 1; u < CB / 8; ++u) pnl += pnv[u];
  const double xn2 = wave_sum(pnl);
  double tau = 0.0, beta = alpha0, scl = 0.0;
  if (xn2 > 1e-280) {
    const double nrm = sqrt(fma(alpha0, alpha0, xn2));
    beta = -copysign(nrm, alpha0);
    tau = (beta - alpha0) / beta;
    scl = 1.0 / (alpha0 - beta);
  }
  auto vrow = [&](int r, double c) { return (r <= j || r >= n) ? 0.0 : (r == j + 1 ? 1.0 : c * scl); };
  if (blockIdx.x == 0 && tid == 0) {
    p.evec[j] = beta;
    p.tauv[j] = tau;
  }
  if (!is_tile) {
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = rs2 + 2 * (w * per + l + 64 * u);
      if ((okmask >> u) & 1u) {
        acc = fma(x[u].x, vrow(r, x[8 + u].x), acc);
        acc = fma(x[u].y, vrow(r + 1, x[8 + u].y), acc);
      }
    }
    acc = wave_sum(acc);
    if (l == 0) s_dot[w] = acc;
    __syncthreads();
    if (tid == 0) {
      const double s = ((s_dot[0] + s_dot[1]) + (s_dot[2] + s_dot[3])) + ((s_dot[4] + s_dot[5]) + (s_dot[6] + s_dot[7]));
      if (q < jj) p.x1[q] = s;
      else p.x2[q - jj] = s;
    }
    return;
  }
  if (tid < TS) s_vi[tid] = vrow(r0 + tid, cvec);
  else if (tid < 2 * TS) s_vj[tid - TS] = vrow(c0 + tid - TS, cvec);
  __syncthreads();
  if (I == J && tid < TS) p.Vh[(size_t)(r0 + tid) + (size_t)j * ld] = s_vi[tid];      // column j of the reflector matrix (rows above rs2 are zero already)
  const double vi0 = s_vi[2 * l], vi1 = s_vi[2 * l + 1];
  double a0 = 0.0, a1 = 0.0;
  double* __restrict__ pcol = p.part + (size_t)I * ld + c0;       // slot I, rows of block J
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const double vc = s_vj[w + 8 * u];
    a0 = fma(x[u].x, vc, a0);
    a1 = fma(x[u].y, vc, a1);
    if (I != J) {        // workgroup-uniform
      const double dsum = wave_sum(fma(x[u].x, vi0, x[u].y * vi1));
      if (l == 0) pcol[w + 8 * u] = dsum;
    }
  }
  s_pr[w][2 * l] = a0;
  s_pr[w][2 * l + 1] = a1;
  __syncthreads();
  if (tid < TS) {      // waves 0 and 1
    const double s = ((s_pr[0][tid] + s_pr[1][tid]) + (s_pr[2][tid] + s_pr[3][tid])) + ((s_pr[4][tid] + s_pr[5][tid]) + (s_pr[6][tid] + s_pr[7][tid]));
    p.part[(size_t)J * ld + r0 + tid] = s;                         // slot J, rows of block I
    // this tile's term of v . y = v^T A v: v_I . (A_IJ v_J), twice below the diagonal (the mirrored tile is never visited)
    const double tv = wave_sum(s * s_vi[tid]);
    if (l == 0) s_dot[w] = tv;
  }
  __syncthreads();
  if (tid == 0) p.pvy[blockIdx.x] = (I == J ? 1.0 : 2.0) * (s_dot[0] + s_dot[1]);
}
// ------------------------------------------------------------------------------------------------ unblocked tail: ONE launch per column
// For trailing blocks of at most EB_UNB_CAP rows the tridiagonalisation is bound by the two kernel boundaries + two chains of
// dependent memory round trips that a column of the panel algorithm costs (8.7-10 us per column at n <= 2048 whatever the size:
// profiles/r05_eig_large_n2048_kernel_stats.csv), not by bytes.  LAPACK's unblocked recurrence (dsytd2) needs ONE device-wide
// exchange per column when the rank-2 update of reflector j - 1 is delayed into the pass that forms A v_j:
//   every workgroup (redundantly, same arithmetic, same bits) takes y = A v of the last step and v itself, whole:
//     alpha = -tau^2/2 (v . y),  w = tau y + alpha v                                  (dsytd2's symmetric correction)
//     c = A[:, j] - v w_j - w v_j   (column j of the reduced matrix: the delayed update applied to this one column),
//     d_j = c_j, the reflector of c[j + 1:] -> e_j, tau_j, v_new                      (its norm from the whole column: no partial sums)
//   then every WAVE owns columns q > j of the (full, symmetric) trailing block:
//     A[:, q] -= v w_q + w v_q  (written back),  y_new[q] = A[:, q] . v_new           (a column is contiguous: 16-byte loads)
// -- v, w and v_new live in LDS (24 bytes per row), the first column of a wave is requested before anything else.  The matrix is
// read AND written once per column (the panel algorithm reads half of it and writes it once per 64 columns), which is why this
// path stops at EB_UNB_CAP rows: below it the trailing block sits in the caches and the column time is latency, not bytes.
constexpr int EB_UNB_CAP = 2304;      // rows of LDS vectors: (cap + 63 + 127 rounded to 128) * 24 bytes <= 64 KB
template <int UNR>       // 16-byte loads per lane that cover one column: (ld - rs) / 2 <= 64 UNR pairs
__global__ __launch_bounds__(512) void k_tri_u(tri_args p, const double* __restrict__ yprev, double* __restrict__ ynew, int has_prev) {
  constexpr int CB = UNR / 4;         // rows per thread of the whole-vector part: (ld - rs) <= 512 CB
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  __shared__ double s_red[8], s_bc[4];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, n = p.n, j = p.j;
  const int64_t ld = p.ld;
  const int rs = j & ~63;
  const int L = p.nr - rs, npair = L >> 1;
  double* s_vp = s_dyn;
  double* s_w = s_dyn + L;
  double* s_vn = s_dyn + 2 * L;
  const int nA = n - j - 1;           // columns of the trailing block
  // this wave's first column: its loads depend on nothing
  const int qstride = gridDim.x * 8;
  const int q0 = j + 1 + blockIdx.x * 8 + w;
  d2 x[UNR];
  {
    const d2* __restrict__ c2 = (const d2*)(p.A + (size_t)(q0 < n ? q0 : j) * ld + rs);
#pragma unroll
    for (int u = 0; u < UNR; ++u) x[u] = (q0 < n && l + 64 * u < npair) ? c2[l + 64 * u] : d2{0.0, 0.0};
  }
  // ---- the whole-vector part
  double yv[CB], vp[CB], cj[CB];
  const double* __restrict__ vprev = p.Vh + (size_t)(j > 0 ? j - 1 : 0) * ld;
  const double* __restrict__ acol = p.A + (size_t)j * ld;
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    const int r = rs + tid + 512 * u;
    const bool in = r >= j && r < n;
    yv[u] = (has_prev && in) ? yprev[r] : 0.0;
    vp[u] = (has_prev && in) ? vprev[r] : 0.0;
    cj[u] = in ? acol[r] : 0.0;
  }
  const double tp = has_prev ? p.tauv[j - 1] : 0.0;
  auto block_sum = [&](double v) {
    v = wave_sum(v);
    if (l == 0) s_red[w] = v;
    __syncthreads();
    const double t = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
    __syncthreads();
    return t;
  };
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < CB; ++u) acc = fma(vp[u], yv[u], acc);
  const double vy = block_sum(acc);
  const double alpha = -0.5 * tp * tp * vy;
  double wv[CB];
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    const int r = rs + tid + 512 * u;
    wv[u] = fma(tp, yv[u], alpha * vp[u]);
    if (r < p.nr) {
      s_vp[r - rs] = vp[u];
      s_w[r - rs] = wv[u];
    }
  }
  __syncthreads();
  const double wj = s_w[j - rs];      // v_{j-1}[j] = 1
  double c[CB];
  acc = 0.0;
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    const int r = rs + tid + 512 * u;
    c[u] = has_prev ? (cj[u] - fma(vp[u], wj, wv[u])) : cj[u];
    if (r == j) s_bc[0] = c[u];
    if (r == j + 1) s_bc[1] = c[u];
    if (r >= j + 2 && r < n) acc = fma(c[u], c[u], acc);
  }
  const double xn2 = block_sum(acc);  // (its barriers publish s_bc as well)
  const double dj = s_bc[0], alpha0 = j + 1 < n ? s_bc[1] : 0.0;
  double tau = 0.0, beta = alpha0, scl = 0.0;
  if (xn2 > 1e-280) {
    const double nrm = sqrt(fma(alpha0, alpha0, xn2));
    beta = -copysign(nrm, alpha0);
    tau = (beta - alpha0) / beta;
    scl = 1.0 / (alpha0 - beta);
  }
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    const int r = rs + tid + 512 * u;
    if (r < p.nr) s_vn[r - rs] = (r <= j || r >= n) ? 0.0 : (r == j + 1 ? 1.0 : c[u] * scl);
  }
  if (blockIdx.x == 0 && tid == 0) {
    p.dvec[j] = dj;
    if (j + 1 < n) {
      p.evec[j] = beta;
      p.tauv[j] = tau;
    }
  }
  __syncthreads();
  if (nA <= 0) return;
  {      // column j of the reflector matrix (rows above rs are zero already), this workgroup's share of the rows
    double* __restrict__ vcol = p.Vh + (size_t)j * ld;
    for (int r = rs + blockIdx.x * 512 + tid; r < n; r += gridDim.x * 512) vcol[r] = s_vn[r - rs];
  }
  const d2* __restrict__ vp2 = (const d2*)s_vp;
  const d2* __restrict__ w2 = (const d2*)s_w;
  const d2* __restrict__ vn2 = (const d2*)s_vn;
  for (int q = q0; q < n; q += qstride) {
    d2* __restrict__ c2 = (d2*)(p.A + (size_t)q * ld + rs);
    if (q != q0) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) x[u] = l + 64 * u < npair ? c2[l + 64 * u] : d2{0.0, 0.0};
    }
    const double wq = s_w[q - rs], vq = s_vp[q - rs];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = l + 64 * u;
      if (i < npair) {
        const d2 vv = vp2[i], ww = w2[i], nn = vn2[i];
        d2 a = x[u];
        if (has_prev) {
          a.x -= fma(vv.x, wq, ww.x * vq);
          a.y -= fma(vv.y, wq, ww.y * vq);
          c2[i] = a;
        }
        a0 = fma(a.x, nn.x, a0);
        a1 = fma(a.y, nn.y, a1);
      }
    }
    const double y = wave_sum(a0 + a1);
    if (l == 0) ynew[q] = y;
  }
}
__global__ void k_tri_tail(const double* __restrict__ A, int64_t ld, int n, double* __restrict__ dvec, double* __restrict__ evec) {
  if (threadIdx.x == 0) {
    dvec[n - 2] = A[(size_t)(n - 2) + (size_t)(n - 2) * ld];
    evec[n - 2] = A[(size_t)(n - 1) + (size_t)(n - 2) * ld];
    dvec[n - 1] = A[(size_t)(n - 1) + (size_t)(n - 1) * ld];
  }
}

// ------------------------------------------------------------------------------------------------ divide and conquer, upper levels
struct dcl_node {
  int lo, mid, hi, K, nrot, skip;
  double rho, tol;
};
struct dcl_args {
  int n, L;
  int64_t ld;
  const double* evec;
  double *D, *Dnew;             // eigenvalue of every column before / after the level's merges
  double *Q, *Qout, *Qg, *S;
  double *Zv, *Ds, *Zs, *dl, *wv, *tauS, *zhat, *rc, *rs;
  int *Col, *Live, *Ks, *Src, *orgv, *ra, *rb;
  int *xkp, *xli;               // k_dcl_deflate<1 | 2>: its index lists in global memory
  double *xd, *xz;              // k_dcl_deflate<2>: poles and rank-one vector in global memory
  unsigned char *xkept, *xlv;
  dcl_node* nodes;
  int* fail;
};
__device__ __forceinline__ void dcl_range(int n, int L, int t, int& node, int& lo, int& mid, int& hi) {
  node = (((t + 1) << L) - 1) / n;
  lo = (int)(((int64_t)node * n) >> L);
  hi = (int)(((int64_t)(node + 1) * n) >> L);
  mid = (int)(((int64_t)(2 * node + 1) * n) >> (L + 1));
}
// the rank-one vector of every node: last row of the left child's Q, first row of the right child's; tolerance of the node
__global__ __launch_bounds__(1024) void k_dcl_z(dcl_args p) {
  __shared__ double s_d[16], s_z[16];
  const int node = blockIdx.x, n = p.n, L = p.L, tid = threadIdx.x;
  const int lo = (int)(((int64_t)node * n) >> L), hi = (int)(((int64_t)(node + 1) * n) >> L);
  const int mid = (int)(((int64_t)(2 * node + 1) * n) >> (L + 1));
  const double beta = p.evec[mid - 1], sgn = beta >= 0.0 ? 1.0 : -1.0, rho = 2.0 * fabs(beta);
  double dmax = 0.0, zmax = 0.0;
  for (int t = lo + tid; t < hi; t += 1024) {
    const double q = t < mid ? p.Q[(size_t)(mid - 1) + (size_t)t * p.ld] : sgn * p.Q[(size_t)mid + (size_t)t * p.ld];
    const double z = q * 0.70710678118654752440;
    p.Zv[t] = z;
    dmax = fmax(dmax, fabs(p.D[t]));
    zmax = fmax(zmax, fabs(z));
  }
  dmax = wave_max(dmax);
  zmax = wave_max(zmax);
  if ((tid & 63) == 0) {
    s_d[tid >> 6] = dmax;
    s_z[tid >> 6] = zmax;
  }
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < 16; ++q) {
      dmax = fmax(dmax, s_d[q]);
      zmax = fmax(zmax, s_z[q]);
    }
    const double tol = 8.0 * DC_EPS * fmax(dmax, zmax);
    dcl_node nd;
    nd.lo = lo;
    nd.mid = mid;
    nd.hi = hi;
    nd.K = 0;
    nd.nrot = 0;
    nd.skip = rho * zmax <= tol ? 1 : 0;            // nothing couples: every component counts as small
    nd.rho = rho;
    nd.tol = tol;
    p.nodes[node] = nd;
  }
}
// ascending rank of every pole inside its node (ties by column), components below the tolerance drop out (dlaed2's first test).
// 64 poles per workgroup, the comparison partners dealt to its 4 waves (n / 64 workgroups: a level keeps many CUs busy).
__global__ __launch_bounds__(256) void k_dcl_rank(dcl_args p) {
  __shared__ double s_d[256];
  __shared__ unsigned char s_live[256];
  __shared__ int s_cnt[3][4][64];
  const int n = p.n, L = p.L, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int t = blockIdx.x * 64 + l;
  const bool valid = t < n;
  int node = 0, lo = 0, mid = 0, hi = 0;
  if (valid) dcl_range(n, L, t, node, lo, mid, hi);
  int ulo, uhi;
  {
    const int t_first = blockIdx.x * 64, t_last = min(n, t_first + 64) - 1;
    int nd, a, m, b;
    dcl_range(n, L, t_first, nd, a, m, b);
    ulo = a;
    dcl_range(n, L, t_last, nd, a, m, b);
    uhi = b;
  }
  const double dt = valid ? p.D[t] : 0.0;
  int rank = 0, pre = 0, cnt = 0;
  for (int base = ulo; base < uhi; base += 256) {
    const int u = base + tid;
    if (u < uhi) {
      int nu, a, m, b;
      dcl_range(n, L, u, nu, a, m, b);
      const dcl_node& nd = p.nodes[nu];
      s_d[tid] = p.D[u];
      s_live[tid] = (nd.skip || nd.rho * fabs(p.Zv[u]) <= nd.tol) ? 0 : 1;
    }
    __syncthreads();
    if (valid) {
      const int q0 = max(0, lo - base), q1 = min(min(256, uhi - base), hi - base);
      for (int q = q0 + w; q < q1; q += 4) {
        const double du = s_d[q];
        const int live = s_live[q];
        const bool before = du < dt || (du == dt && base + q < t);
        rank += before ? 1 : 0;
        pre += before ? live : 0;
        cnt += live;
      }
    }
    __syncthreads();
  }
  s_cnt[0][w][l] = rank;
  s_cnt[1][w][l] = pre;
  s_cnt[2][w][l] = cnt;
  __syncthreads();
  if (w == 0 && valid) {
    rank = (s_cnt[0][0][l] + s_cnt[0][1][l]) + (s_cnt[0][2][l] + s_cnt[0][3][l]);
    pre = (s_cnt[1][0][l] + s_cnt[1][1][l]) + (s_cnt[1][2][l] + s_cnt[1][3][l]);
    cnt = (s_cnt[2][0][l] + s_cnt[2][1][l]) + (s_cnt[2][2][l] + s_cnt[2][3][l]);
    const dcl_node& nd = p.nodes[node];
    const double zt = p.Zv[t];
    const bool live_t = !(nd.skip || nd.rho * fabs(zt) <= nd.tol);
    p.Ds[lo + rank] = dt;
    p.Zs[lo + rank] = zt;
    p.Col[lo + rank] = t;
    p.Live[lo + rank] = live_t ? 1 : 0;
    if (live_t) p.Ks[lo + pre] = lo + rank;
    if (t == lo) p.nodes[node].K = cnt;
  }
}
// dlaed2's second test (two neighbouring surviving poles so close that a plane rotation decouples one of them): checked on all
// pairs in parallel; only if some pair is close, one thread redoes the scan sequentially with the rotations (everything in LDS).
// Then the kept poles are gathered and the new column order of the node is fixed: kept poles first (ascending), the deflated
// columns behind them (ascending).
// BIG (a node of more than 4160 poles, i.e. the top merge beyond n = 4096): only the poles and the rank-one vector stay in LDS (16
// bytes per pole, 132 KB at 8192), the index lists live in global memory at the node's offset -- every access pattern below is
// "written, workgroup barrier, read by other threads of the SAME workgroup", which global memory serves like LDS.
// MODE 2 (a node of more than 9984 poles: the top merge beyond n = 9984): the poles and the rank-one vector live in global memory
// too (xd, xz): no dynamic LDS at all; the sequential rotation scan, when it is needed at all, then walks global memory.
template <int MODE>
__global__ __launch_bounds__(1024) void k_dcl_deflate(dcl_args p, int cap) {
  constexpr bool BIG = MODE >= 1;
  // dynamic LDS, cap = the largest node of the level rounded up to 64: 30 bytes per pole (125 KB at 4096)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ int s_scan[1024];
  __shared__ int s_any, s_K, s_nrot;
  const int node = blockIdx.x, tid = threadIdx.x;
  const dcl_node nd = p.nodes[node];
  const int lo = nd.lo, hi = nd.hi, nn = hi - lo, K0 = nd.K;
  const double tol = nd.tol;
  double* s_d = MODE == 2 ? p.xd + lo : (double*)s_raw;
  double* s_z = MODE == 2 ? p.xz + lo : s_d + cap;
  int* s_ks = BIG ? p.Ks + lo : (int*)(s_z + cap);
  int* s_kp = BIG ? p.xkp + lo : s_ks + cap;
  int* s_li = BIG ? p.xli + lo : s_kp + cap;   // sorted position -> index in the live list (-1: deflated by the first test)
  unsigned char* s_kept = BIG ? p.xkept + lo : (unsigned char*)(s_li + cap);
  unsigned char* s_lv = BIG ? p.xlv + lo : s_kept + cap;
  if (tid == 0) {
    s_any = 0;
    s_K = K0;
    s_nrot = 0;
  }
  for (int i = tid; i < K0; i += 1024) {
    const int s = p.Ks[lo + i];
    if (!BIG) s_ks[i] = s;
    s_d[i] = p.Ds[s];
    s_z[i] = p.Zs[s];
    s_kp[i] = i;
    s_kept[i] = 1;
  }
  for (int q = tid; q < nn; q += 1024) {
    s_lv[q] = (unsigned char)p.Live[lo + q];
    s_li[q] = -1;
  }
  __syncthreads();
  for (int i = tid; i < K0; i += 1024) s_li[s_ks[i] - lo] = i;
  __syncthreads();
  for (int i = 1 + tid; i < K0; i += 1024) {
    const double zs = s_z[i], zp = s_z[i - 1], t = s_d[i] - s_d[i - 1];
    // |t c s| <= tol with c = zs / tau, s = -zp / tau, tau^2 = zs^2 + zp^2
    if (fabs(t * zs * zp) <= tol * fma(zs, zs, zp * zp)) s_any = 1;
  }
  __syncthreads();
  if (s_any) {
    if (tid == 0) {
      int K = 0, nrot = 0, pj = -1;
      double zp = 0.0, dp = 0.0;
      for (int i = 0; i < K0; ++i) {
        const double zs = s_z[i], ds = s_d[i];
        if (pj < 0) {
          pj = i;
          zp = zs;
          dp = ds;
          continue;
        }
        const double tau = sqrt(fma(zs, zs, zp * zp));     // |z| <= 1: no overflow to guard
        const double c = zs / tau, sn = -zp / tau;
        const double t = ds - dp;
        if (fabs(t * c * sn) <= tol) {
          s_z[pj] = 0.0;
          s_kept[pj] = 0;
          p.ra[lo + nrot] = s_ks[pj];
          p.rb[lo + nrot] = s_ks[i];
          p.rc[lo + nrot] = c;
          p.rs[lo + nrot] = sn;
          ++nrot;
          s_d[pj] = dp * c * c + ds * sn * sn;
          const double dnew = dp * sn * sn + ds * c * c;
          s_d[i] = dnew;
          s_z[i] = tau;
          pj = i;
          zp = tau;
          dp = dnew;
        } else {
          s_kp[K++] = pj;
          pj = i;
          zp = zs;
          dp = ds;
        }
      }
      if (pj >= 0) s_kp[K++] = pj;
      s_K = K;
      s_nrot = nrot;
    }
    __syncthreads();
    for (int i = tid; i < K0; i += 1024)
      if (!s_kept[i]) s_lv[s_ks[i] - lo] = 0;
    __syncthreads();
  }
  const int K = s_K;
  // kept poles -> the secular problem
  for (int i = tid; i < K; i += 1024) {
    const int li = s_kp[i], s = s_ks[li];
    p.dl[lo + i] = s_d[li];
    p.wv[lo + i] = s_z[li];
    p.Src[lo + i] = p.Col[s];
  }
  // deflated columns: position among the deflated, in sorted order (exclusive scan of the dead flags, 4 positions per thread)
  const int per = (nn + 1023) / 1024;
  int dead = 0;
  for (int q = tid * per; q < min(nn, (tid + 1) * per); ++q) dead += s_lv[q] ? 0 : 1;
  s_scan[tid] = dead;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  int pos = s_scan[tid] - dead;
  // the eigenvalue of a deflated column: its pole -- the rotated one (in LDS under its live index) if a rotation deflated it
  for (int q = tid * per; q < min(nn, (tid + 1) * per); ++q) {
    if (!s_lv[q]) {
      const int li = s_li[q];
      p.Src[lo + K + pos] = p.Col[lo + q];
      p.Dnew[lo + K + pos] = li >= 0 ? s_d[li] : p.Ds[lo + q];
      ++pos;
    }
  }
  if (tid == 0) {
    p.nodes[node].K = K;
    p.nodes[node].nrot = s_nrot;
  }
}
// the deflating rotations on the columns of Q, one thread per row, in scan order (a chain of rotations hands its second column on
// as the first column of the next one: that value stays in a register)
__global__ __launch_bounds__(256) void k_dcl_rot(dcl_args p) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= p.n) return;
  int node, lo, mid, hi;
  dcl_range(p.n, p.L, r, node, lo, mid, hi);
  const int nrot = p.nodes[node].nrot;
  int prev_cb = -1;
  double prev_val = 0.0;
  for (int q = 0; q < nrot; ++q) {
    const int ca = p.Col[p.ra[lo + q]], cb = p.Col[p.rb[lo + q]];
    const double c = p.rc[lo + q], sn = p.rs[lo + q];
    const double qa = ca == prev_cb ? prev_val : p.Q[(size_t)r + (size_t)ca * p.ld];
    const double qb = p.Q[(size_t)r + (size_t)cb * p.ld];
    const double na = c * qa + sn * qb, nb = c * qb - sn * qa;
    p.Q[(size_t)r + (size_t)ca * p.ld] = na;
    p.Q[(size_t)r + (size_t)cb * p.ld] = nb;
    prev_cb = cb;
    prev_val = nb;
  }
}
// one wave per root of the secular equation
__global__ __launch_bounds__(256) void k_dcl_secular(dcl_args p) {
  const int l = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= p.n) return;
  int node, lo, mid, hi;
  dcl_range(p.n, p.L, t, node, lo, mid, hi);
  const int K = p.nodes[node].K, i = t - lo;
  if (i >= K) return;
  int org, evals;
  double tau;
  const bool ok = secular_root<64>(p.dl + lo, p.wv + lo, K, i, l, p.nodes[node].rho, org, tau, evals);
  if (l == 0) {
    p.tauS[t] = tau;
    p.orgv[t] = org;
    p.Dnew[t] = p.dl[lo + org] + tau;
    if (!ok) atomicOr(p.fail, 1);
  }
}
// Gu-Eisenstat: the rank-one vector for which the COMPUTED roots are exact
__global__ __launch_bounds__(256) void k_dcl_zhat(dcl_args p) {
  const int l = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= p.n) return;
  int node, lo, mid, hi;
  dcl_range(p.n, p.L, t, node, lo, mid, hi);
  const int K = p.nodes[node].K, i = t - lo;
  if (i >= K) return;
  const double* __restrict__ dl = p.dl + lo;
  const double di = dl[i];
  double prod = 1.0;
  for (int j = l; j < K; j += 64) {
    const double num = (dl[p.orgv[lo + j]] - di) + p.tauS[lo + j];     // lam_j - dl_i
    prod *= (j == i) ? num : num * dc_rcp(dl[j] - di);
  }
  prod = group_prod<64>(prod);
  if (l == 0) p.zhat[t] = copysign(sqrt(fabs(prod)), p.wv[t]);
}
// eigenvectors of the rank-one problem, normalised: column i of the node's S (K x K at S[lo.., lo..], column-major)
__global__ __launch_bounds__(256) void k_dcl_svec(dcl_args p) {
  const int l = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= p.n) return;
  int node, lo, mid, hi;
  dcl_range(p.n, p.L, t, node, lo, mid, hi);
  const int K = p.nodes[node].K, i = t - lo;
  if (i >= K) return;
  const double* __restrict__ dl = p.dl + lo;
  const double* __restrict__ zh = p.zhat + lo;
  const double dorg = dl[p.orgv[t]], tau = p.tauS[t];
  double nrm = 0.0;
  for (int q = l; q < K; q += 64) {
    const double sv = zh[q] * dc_rcp((dl[q] - dorg) - tau);
    nrm = fma(sv, sv, nrm);
  }
  nrm = wave_sum(nrm);
  const double inv = dc_rsqrt(nrm);
  double* __restrict__ scol = p.S + (size_t)lo + (size_t)t * p.ld;
  for (int q = l; q < K; q += 64) scol[q] = zh[q] * dc_rcp((dl[q] - dorg) - tau) * inv;
}
// new column order: kept columns gathered for the product (Qg), deflated columns moved over unchanged (Qout); rows of the node
__global__ __launch_bounds__(256) void k_dcl_gather(dcl_args p) {
  const int t = blockIdx.x;
  int node, lo, mid, hi;
  dcl_range(p.n, p.L, t, node, lo, mid, hi);
  const int K = p.nodes[node].K;
  const double* __restrict__ src = p.Q + (size_t)p.Src[t] * p.ld;
  double* __restrict__ dst = (t - lo < K ? p.Qg : p.Qout) + (size_t)t * p.ld;
  for (int r = lo + threadIdx.x; r < hi; r += 256) dst[r] = src[r];
}

// ------------------------------------------------------------------------------------------------ block reflectors
// triangular factor of a 64-column panel from its Gram matrix (LAPACK dlarft, forward, columnwise): T[i][i] = tau_i,
// T[0:i, i] = -tau_i T[0:i, 0:i] (V^T v_i).  One wave per panel.  The panels are the diagonal 64 x 64 blocks of the EB_WY-wide
// block reflectors (EB_WY = 256 | 512): panel p lives at offset (p % (EB_WY / 64)) * 64 on the diagonal of block p / (EB_WY / 64) of G and T
// (leading dimension EB_WY).
__global__ __launch_bounds__(64) void k_larft(const double* __restrict__ G, const double* __restrict__ tauv, double* __restrict__ Tf, int EB_WY) {
  __shared__ double s_t[EB_NB][EB_NB + 1], s_g[EB_NB];
  const int r = threadIdx.x;
  const size_t off = (size_t)(blockIdx.x / (EB_WY / EB_NB)) * EB_WY * EB_WY + (size_t)(blockIdx.x % (EB_WY / EB_NB)) * EB_NB * (EB_WY + 1);
  const double* g = G + off;
  for (int c = 0; c < EB_NB; ++c) s_t[r][c] = 0.0;
  for (int i = 0; i < EB_NB; ++i) {
    s_g[r] = g[r + (size_t)i * EB_WY];     // column i of the Gram matrix: v_c . v_i
    __syncthreads();
    const double ti = tauv[blockIdx.x * EB_NB + i];
    double v = 0.0;
    if (r < i) {
      double s = 0.0;
      for (int c = r; c < i; ++c) s = fma(s_t[r][c], s_g[c], s);
      v = -ti * s;
    } else if (r == i) {
      v = ti;
    }
    s_t[r][i] = v;
    __syncthreads();
  }
  double* tf = Tf + off;
  for (int c = 0; c < EB_NB; ++c) tf[r + (size_t)c * EB_WY] = s_t[r][c];
}
// out (row-major n x nv) [i][pos] = Z[i, pos]: transposition through LDS
__global__ __launch_bounds__(256) void k_out(const double* __restrict__ Z, int64_t ld, int n, int nv, double* __restrict__ out) {
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = blockIdx.x * 32, p0 = blockIdx.y * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = ty + 8 * q;            // position p0 + a, row i0 + tx
    t[a][tx] = (p0 + a < nv && i0 + tx < n) ? Z[(size_t)(i0 + tx) + (size_t)(p0 + a) * ld] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = ty + 8 * q;            // row i0 + a, position p0 + tx
    if (i0 + a < n && p0 + tx < nv) out[(size_t)(i0 + a) * nv + (p0 + tx)] = t[tx][a];
  }
}
// Zs[:, pos] = Z[:, order[pos]] for pos < nv: the wanted eigenvectors of the tridiagonal matrix in output order, BEFORE the
// back-transformation -- its products then run over nv columns only
__global__ __launch_bounds__(256) void k_pick_columns(const double* __restrict__ Z, double* __restrict__ Zs, int64_t ld, int n,
                                                      const int* __restrict__ order) {
  const double* __restrict__ src = Z + (size_t)order[blockIdx.x] * ld;
  double* __restrict__ dst = Zs + (size_t)blockIdx.x * ld;
  for (int r = threadIdx.x; r < n; r += 256) dst[r] = src[r];
}

// upper triangle of the trailing block <- its lower triangle (A[c, r] = A[r, c], r > c >= t0): the rank-2k updates of the panels
// whose columns take the lower-triangle products only touch the tiles on and below the diagonal; the full-column products
// and the unblocked tail read whole columns
__global__ __launch_bounds__(256) void k_mirror_lower(double* __restrict__ A, int64_t ld, int n, int t0) {
  __shared__ double t[32][33];
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = t0 + bi * 32, c0 = t0 + bj * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = ty + 8 * q;            // element (r0 + tx, c0 + c) of the lower triangle
    t[c][tx] = (r0 + tx < n && c0 + c < n) ? A[(size_t)(r0 + tx) + (size_t)(c0 + c) * ld] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = ty + 8 * q;           // destination (c0 + tx, r0 + rr): row index from the source's column range
    const int dr = c0 + tx, dc = r0 + rr;
    if (dr < n && dc < n && dc > dr) A[(size_t)dr + (size_t)dc * ld] = t[tx][rr];
  }
}

struct phase_clock {
  hfmi_ctx* ctx;
  bool on;
  std::chrono::steady_clock::time_point t0;
  double ms[8];
  int cur;
  explicit phase_clock(hfmi_ctx* c) : ctx(c), on(env_flag("HFMI_EIG_LARGE_TIMING")), cur(0) {
    memset(ms, 0, sizeof(ms));
    if (on) {
      (void)hipStreamSynchronize(ctx->stream);
      t0 = std::chrono::steady_clock::now();
    }
  }
  void mark(int slot) {
    if (!on) return;
    (void)hipStreamSynchronize(ctx->stream);
    const auto t1 = std::chrono::steady_clock::now();
    ms[slot] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    t0 = t1;
  }
};
}  // namespace

// instrumentation: C (M x N) = op(A) op(B) through launch_dgemm (the products of the eigensolver), host operands column-major with
// their natural leading dimensions (A: ta ? K x M : M x K; B: tb ? N x K : K x N); average time of `reps` launches after one warm-up
int eig_dgemm_bench(hfmi_ctx* ctx, int M, int N, int K, int ta, int tb, int reps, const double* host_A, const double* host_B,
                    double* host_C, double* avg_ms) {
  const int64_t ra = ta ? K : M, ca = ta ? M : K, rb = tb ? N : K, cb = tb ? K : N;
  const int64_t lda = round_up(ra, 2), ldb = round_up(rb, 2), ldc = round_up(M, 2);
  void* wv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_STAGE, (size_t)(lda * ca + ldb * cb + ldc * N) * sizeof(double), &wv));
  double *A = (double*)wv, *B = A + lda * ca, *C = B + ldb * cb;
  hipStream_t st = ctx->stream;
  HIP_TRY(hipMemcpy2DAsync(A, lda * sizeof(double), host_A, ra * sizeof(double), ra * sizeof(double), ca, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpy2DAsync(B, ldb * sizeof(double), host_B, rb * sizeof(double), rb * sizeof(double), cb, hipMemcpyHostToDevice, st));
  gemm_desc g;
  g.ta = ta != 0;
  g.tb = tb != 0;
  g.M = M;
  g.N = N;
  g.K = K;
  g.alpha = 1.0;
  g.beta = 0.0;
  g.A = A;
  g.B = B;
  g.lda = lda;
  g.ldb = ldb;
  g.C = C;
  g.ldc = ldc;
  HFMI_TRY(launch_dgemm(ctx, g));
  HIP_TRY(hipEventRecord(ctx->ev0, st));
  for (int r = 0; r < reps; ++r) HFMI_TRY(launch_dgemm(ctx, g));
  HIP_TRY(hipEventRecord(ctx->ev1, st));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  if (avg_ms) *avg_ms = reps > 0 ? ms / reps : 0.0;
  if (host_C) {
    HIP_TRY(hipMemcpy2DAsync(host_C, (size_t)M * sizeof(double), C, ldc * sizeof(double), (size_t)M * sizeof(double), N, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return HFMI_OK;
}

// host_T: n x n row-major symmetric (the symmetric part is used); host_d: n eigenvalues descending (by |d| if
// sort_by_abs); host_V: n x n row-major, eigenvectors in the columns (may be null).
// nvec: how many eigenvectors (the leading ones in output order) are wanted; host_V is n x nvec row-major.
// dev_T != null: the matrix is already in device memory (n x n row-major, what hfmi_block_dot leaves in its workspace) and host_T is
// ignored -- the Gram matrix of the deterministic POD then never crosses PCIe (hfmi_block_gram_eig).
int sym_eig_large(hfmi_ctx* ctx, const double* host_T, int n, int sort_by_abs, double* host_d, double* host_V, int nvec,
                  const double* dev_T) {
  if (n > EB_MAXN) HFMI_FAIL(HFMI_ERR_INVALID, "sym_eig: n=%d exceeds %d", n, EB_MAXN);
  sort_by_abs &= 1;      // the callers hand over their whole flags word (bit 1 = HFMI_EIG_JACOBI selects a method, not an order)
  if (nvec < 0 || nvec > n) nvec = n;
  if (!host_V) nvec = 0;
  static const bool jacobi_env = [] {
    const char* e = getenv("HFMI_EIG_LARGE");
    return e && !strcmp(e, "jacobi");
  }();
  const bool jacobi = jacobi_env && n <= 4096;      // (the A/B route of round 4 stops there)
  std::vector<double> staged;
  if ((jacobi || n < 3) && dev_T) {      // the Jacobi route takes a host matrix
    staged.resize((size_t)n * n);
    HIP_TRY(hipMemcpyAsync(staged.data(), dev_T, staged.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    host_T = staged.data();
  }
  if (jacobi || n < 3) {
    if (nvec == n || !host_V) return sym_eig_large_jacobi(ctx, host_T, n, sort_by_abs, host_d, host_V);
    std::vector<double> full((size_t)n * n);
    HFMI_TRY(sym_eig_large_jacobi(ctx, host_T, n, sort_by_abs, host_d, full.data()));
    for (int i = 0; i < n; ++i) memcpy(host_V + (size_t)i * nvec, full.data() + (size_t)i * n, (size_t)nvec * sizeof(double));
    return HFMI_OK;
  }
  const int NB = EB_NB;
  // width of a block reflector of the back-transformation: V^T Z is a WY x nv product over WY / 64 row tiles -- 512 columns from
  // n = 2048 on keep 8 row tiles x nv / 128 column tiles of the pipelined kernel on the chip where 256 would leave half of it idle
  // (back-transformation at n = 2048 / 4096: 1.9 -> 1.6 / 6.5 -> 4.7 ms; profiles/r06_eig_large.txt)
  static const int wy_env = [] {
    const char* e = getenv("HFMI_EIG_WY");
    const int v = e ? atoi(e) : 0;
    return (v == 256 || v == 512) ? v : 0;
  }();
  const int WY = wy_env ? wy_env : (n >= 2048 ? 512 : 256);
  // rows of a column: n rounded up to 128.  Leading dimension: the same, except where that is a power of two (n = 4096, 8192,
  // 16384): consecutive columns of a 128 x 128 tile or of a wave's column set then sit a power of two apart and crowd the same
  // HBM channels -- the tile pattern of the lower-triangle products read 6.25 TB/s at ld = 8192 and 6.9-7.0 at 8336 / 8720, 5.5
  // against 6.0-6.2 at n = 4096 (scripts/tile_stride_probe.hip, profiles/r06_tile_stride_probe.txt); 144 = 9 cache lines
  static const bool ld_pad = !env_flag("HFMI_EIG_NO_LD_PAD");
  const int nr = (int)round_up(n, 128);
  const int64_t ld = nr + ((ld_pad && nr >= 4096 && (nr & (nr - 1)) == 0) ? 144 : 0);
  const int npad = (int)round_up(n, WY), npanels = npad / NB, nblk = npad / WY;
  const size_t mat = (size_t)ld * npad;
  const size_t vlen = (size_t)npad + 128;
  // ---- workspace
  const size_t n_mats = 5;
  const size_t d_count = n_mats * mat + (size_t)ld * NB + (size_t)WY * npad + 3 * (size_t)nblk * WY * WY + 18 * vlen + 2 * 64 + EB_MAXN / 64 + 2112 +
                         (size_t)(npad / 32 + 1) * (npad / 32 + 1);
  const size_t i_count = 11 * vlen + 64;
  const size_t bytes = d_count * sizeof(double) + i_count * sizeof(int) + 64 * sizeof(dcl_node) + 256;
  void* wv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_STAGE, bytes, &wv));
  double* dp = (double*)wv;
  auto take = [&](size_t count) {
    double* r = dp;
    dp += count;
    return r;
  };
  double* A = take(mat);        // the matrix; later S of the merges, then Y = V T of the back-transformation
  double* Vh = take(mat);
  double* Q1 = take(mat);
  double* Q2 = take(mat);
  double* Qg = take(mat);       // raw upload first, gathered columns of the merges, nothing afterwards
  double* Wp = take((size_t)ld * NB);
  double* W1 = take((size_t)WY * npad);
  double* Gm = take((size_t)nblk * WY * WY);
  double* Tf = take((size_t)nblk * WY * WY);
  double* Tt = take((size_t)nblk * WY * WY);      // products in flight while the triangular factors are merged
  double* colbuf = take(vlen);
  double* ybuf = take(vlen);
  double* dvec = take(vlen);
  double* evec = take(vlen);
  double* tauv = take(vlen);
  double* D0 = take(vlen);
  double* D1 = take(vlen);
  double* Zv = take(vlen);
  double* Ds = take(vlen);
  double* Zs = take(vlen);
  double* dl = take(vlen);
  double* wvv = take(vlen);
  double* tauS = take(vlen);
  double* zhat = take(vlen);
  double* rc = take(vlen);
  double* rs = take(vlen);
  double* xd = take(vlen);
  double* xz = take(vlen);
  double* x1 = take(64);
  double* x2 = take(64);
  double* pn = take(EB_MAXN / 64);  // one partial norm per 64 rows
  double* pvy = take(2112);         // partial sums of v . y: <= 512 workgroups of k_tri_b, <= 2080 tiles of k_tri_bs
  double* pmax = take((size_t)(npad / 32 + 1) * (npad / 32 + 1));
  int* ip = (int*)dp;
  auto take_i = [&](size_t count) {
    int* r = ip;
    ip += count;
    return r;
  };
  int* Col = take_i(vlen);
  int* Live = take_i(vlen);
  int* Ks = take_i(vlen);
  int* Src = take_i(vlen);
  int* orgv = take_i(vlen);
  int* ra = take_i(vlen);
  int* rb = take_i(vlen);
  int* order = take_i(vlen);
  int* xkp = take_i(vlen);
  int* xli = take_i(vlen);
  unsigned char* xkept = (unsigned char*)take_i(vlen / 2);
  unsigned char* xlv = (unsigned char*)take_i(vlen / 2);
  int* fail = take_i(16);
  int* sexp_dev = fail + 1;
  dcl_node* nodes = (dcl_node*)round_up((int64_t)(uintptr_t)(ip + 48), 16);
  hipStream_t st = ctx->stream;
  phase_clock clk(ctx);

  // ---- load
  const double* raw = dev_T ? dev_T : Qg;
  HIP_TRY(hipMemsetAsync(A, 0, mat * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(Vh, 0, mat * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(Q1, 0, 2 * mat * sizeof(double), st));          // Q1 and Q2 are adjacent
  HIP_TRY(hipMemsetAsync(Wp, 0, (size_t)ld * NB * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(colbuf, 0, 16 * vlen * sizeof(double), st));    // the vectors (tau beyond n - 3 must read 0)
  HIP_TRY(hipMemsetAsync(fail, 0, 16 * sizeof(int), st));
  clk.mark(6);
  if (!dev_T) HFMI_TRY(xfer_h2d(ctx, Qg, host_T, (size_t)n * n * sizeof(double)));      // (the fills above run while the host side is read)
  clk.mark(7);
  {
    const int nt = (n + 31) / 32;
    hipLaunchKernelGGL(k_sym_load, dim3(nt, nt), dim3(256), 0, st, raw, n, A, ld, pmax);
    hipLaunchKernelGGL(k_scale_exp, dim3(1), dim3(1024), 0, st, pmax, nt * nt, sexp_dev);
    hipLaunchKernelGGL(k_scale_apply, dim3(n), dim3(256), 0, st, A, ld, n, sexp_dev);
    HIP_TRY(hipGetLastError());
  }
  clk.mark(0);

  // ---- tridiagonalisation
  {
    static const int tri_unr = [] {      // HFMI_EIG_TRI_UNR = 4 | 8: 16-byte loads in flight per lane of k_tri_b (A/B)
      const char* e = getenv("HFMI_EIG_TRI_UNR");
      return (e && atoi(e) == 4) ? 4 : 8;
    }();
    static const int sym_min = [] {      // HFMI_EIG_SYM_MIN: trailing blocks from this size on take the lower-triangle products (0: never)
      const char* e = getenv("HFMI_EIG_SYM_MIN");
      const int v = e ? atoi(e) : 3072;
      return v <= 0 ? (1 << 30) : std::max(v, 256);
    }();
    tri_args ta;
    ta.part = Qg;                        // free between the load and the merges: nb <= 64 partial vectors of ld doubles
    ta.nb = 0;
    ta.n = n;
    ta.nr = nr;
    ta.ld = ld;
    ta.A = A;
    ta.Vh = Vh;
    ta.W = Wp;
    ta.colbuf = colbuf;
    ta.ybuf = ybuf;
    ta.x1 = x1;
    ta.x2 = x2;
    ta.pvy = pvy;
    ta.pn = pn;
    ta.dvec = dvec;
    ta.evec = evec;
    ta.tauv = tauv;
    ta.npvy = 0;
    ta.npn = 0;
    bool prev_slots = false;             // the last column's products were left in slots by k_tri_bs
    if (n > 8192)      // v of the first columns is 128 KB (of 160)
      HIP_TRY(hipFuncSetAttribute((const void*)k_tri_b<8, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(nr * sizeof(double))));
    else if (n > 4096)      // v of the first columns is 64 KB: beyond what a kernel gets without asking
      HIP_TRY(hipFuncSetAttribute((const void*)k_tri_b<8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(nr * sizeof(double))));
    // the lower-triangle products serve trailing blocks of sym_min ... 8192 rows (64 slots of 128 rows, 2080 tiles); the first columns of a
    // larger matrix take the full-column products
    auto uses_bs = [&](int j) { return n - j - 1 >= sym_min && (nr - ((j + 1) & ~(TS - 1))) / TS <= 64; };
    static const int unb_max = [] {      // HFMI_EIG_UNB_MAX: trailing blocks of at most this many rows take one launch per column (0: never)
      const char* e = getenv("HFMI_EIG_UNB_MAX");
      const int v = e ? atoi(e) : 2048;
      return std::max(0, std::min(v, EB_UNB_CAP));
    }();
    int j_unb = -1;                      // first column of the unblocked tail
    bool upper_valid = true;             // the upper triangle of the trailing block is up to date
    static const bool lower_updates = !env_flag("HFMI_EIG_FULL_UPDATE");     // A/B: every rank-2k update over the full block
    auto make_upper_valid = [&](int t0) {
      if (upper_valid) return;
      const int nt = (n - t0 + 31) / 32;
      hipLaunchKernelGGL(k_mirror_lower, dim3(nt, nt), dim3(256), 0, st, A, ld, n, t0);
      upper_valid = true;
    };
    for (int p0 = 0; p0 < n - 2; p0 += NB) {
      if (n - p0 <= unb_max) {
        j_unb = p0;
        make_upper_valid(p0);
        break;
      }
      const int ncols = std::min(NB, n - 2 - p0);
      ta.p0 = p0;
      for (int jj = 0; jj < ncols; ++jj) {
        const int j = p0 + jj;
        ta.j = j;
        ta.jj = jj;
        ta.mode = 3;
        const int ga = (n - j + 63) / 64;
        ta.npn = ga;
        if (prev_slots) hipLaunchKernelGGL(k_tri_a<true>, dim3(ga), dim3(64 * TA_WAVES), 0, st, ta);
        else hipLaunchKernelGGL(k_tri_a<false>, dim3(ga), dim3(64 * TA_WAVES), 0, st, ta);
        const int nc = (n - j - 1) + 2 * jj;
        const int gb = std::max(1, std::min(512, (nc + 7) / 8));
        const int rs0 = (j + 1) & ~63;
        const size_t v_lds = (size_t)(nr - rs0) * sizeof(double);
        if (uses_bs(j)) {
          // large trailing block: the lower triangle only (k_tri_bs); the next k_tri_a adds the partial vectors
          const int rs2 = (j + 1) & ~(TS - 1);
          const int nb = (nr - rs2) / TS, ntiles = nb * (nb + 1) / 2;
          ta.nb = nb;
          if (n > 4096) hipLaunchKernelGGL(k_tri_bs<16>, dim3(ntiles + 2 * jj), dim3(512), 0, st, ta, ntiles);
          else hipLaunchKernelGGL(k_tri_bs<8>, dim3(ntiles + 2 * jj), dim3(512), 0, st, ta, ntiles);
          ta.npvy = ntiles;
          prev_slots = true;
        } else {
          if (n > 8192) hipLaunchKernelGGL((k_tri_b<8, 32>), dim3(gb), dim3(512), v_lds, st, ta);
          else if (n > 4096) hipLaunchKernelGGL((k_tri_b<8, 16>), dim3(gb), dim3(512), v_lds, st, ta);
          else if (tri_unr == 8) hipLaunchKernelGGL((k_tri_b<8, 8>), dim3(gb), dim3(512), v_lds, st, ta);
          else hipLaunchKernelGGL((k_tri_b<4, 8>), dim3(gb), dim3(512), v_lds, st, ta);
          ta.npvy = gb;
          prev_slots = false;
        }
      }
      const int t0 = p0 + ncols;
      ta.j = t0;
      ta.jj = ncols;
      ta.mode = 1;
      if (prev_slots) hipLaunchKernelGGL(k_tri_a<true>, dim3((n - t0 + 63) / 64), dim3(64 * TA_WAVES), 0, st, ta);
      else hipLaunchKernelGGL(k_tri_a<false>, dim3((n - t0 + 63) / 64), dim3(64 * TA_WAVES), 0, st, ta);
      HIP_TRY(hipGetLastError());
      // A[t0:, t0:] -= V W^T + W V^T
      gemm_desc g;
      g.ta = false;
      g.tb = true;
      g.M = g.N = n - t0;
      g.K = g.K2 = ncols;
      g.alpha = -1.0;
      g.beta = 1.0;
      g.A = Vh + (size_t)p0 * ld + t0;
      g.B = Wp + t0;
      g.A2 = Wp + t0;
      g.B2 = Vh + (size_t)p0 * ld + t0;
      g.lda = g.ldb = ld;
      g.C = A + (size_t)t0 * ld + t0;
      g.ldc = ld;
      // every column of the NEXT panel takes the lower-triangle products (and there is a next panel): the tiles above the diagonal are
      // not read again until the full-column / unblocked columns begin -- they are skipped and mirrored back once, there
      const bool next_all_lower = uses_bs(t0) && uses_bs(t0 + NB - 1) && n - t0 > unb_max;
      g.lower = (lower_updates && (next_all_lower || !upper_valid)) ? 1 + (t0 & 127) : 0;
      HFMI_TRY(launch_dgemm(ctx, g));
      if (g.lower) upper_valid = false;
      if (!next_all_lower) make_upper_valid(t0);
    }
    make_upper_valid(0);
    if (j_unb < 0) {
      hipLaunchKernelGGL(k_tri_tail, dim3(1), dim3(64), 0, st, A, ld, n, dvec, evec);
    } else {
      // the matrix is fully updated at a panel boundary: nothing is pending at column j_unb.  Steps j_unb .. n - 1: step j applies
      // reflector j - 1 and forms reflector j; the last two steps only collect d and e of the final 2 x 2 block.
      double* yb[2] = {ybuf, colbuf};      // y of the last step / of this step (the panel kernels' colbuf is free here)
      for (int j = j_unb; j < n; ++j) {
        ta.j = j;
        const int rs0 = j & ~63, L = nr - rs0, nA = n - j - 1;
        const int g = std::max(1, std::min(512, (nA + 7) / 8));
        const size_t lds = (size_t)3 * L * sizeof(double);
        const int has_prev = j > j_unb ? 1 : 0;
        const double* yp = yb[(j - j_unb) & 1];
        double* yn = yb[(j - j_unb + 1) & 1];
        if (L <= 512) hipLaunchKernelGGL(k_tri_u<4>, dim3(g), dim3(512), lds, st, ta, yp, yn, has_prev);
        else if (L <= 1024) hipLaunchKernelGGL(k_tri_u<8>, dim3(g), dim3(512), lds, st, ta, yp, yn, has_prev);
        else if (L <= 2048) hipLaunchKernelGGL(k_tri_u<16>, dim3(g), dim3(512), lds, st, ta, yp, yn, has_prev);
        else hipLaunchKernelGGL(k_tri_u<20>, dim3(g), dim3(512), lds, st, ta, yp, yn, has_prev);
      }
    }
    HIP_TRY(hipGetLastError());
  }
  clk.mark(1);

  // ---- divide and conquer
  static const int leaf_max = [] {      // HFMI_EIG_LEAF = 64 ... 256: largest leaf handed to the one-workgroup solver (A/B)
    const char* e = getenv("HFMI_EIG_LEAF");
    const int v = e ? atoi(e) : 0;
    return (v >= 64 && v <= 256) ? v : 128;      // 128: n = 512 / 1024 5.65 / 11.5 ms against 6.06 / 11.9 with 256-row leaves, equal beyond
  }();
  int Lf = 0;
  while (((n + (1 << Lf) - 1) >> Lf) > leaf_max) ++Lf;
  HFMI_TRY(launch_dc_leaves(ctx, n, Lf, dvec, evec, D0, Q1, ld, fail));
  clk.mark(2);
  double *Dcur = D0, *Dnext = D1, *Qcur = Q1, *Qnext = Q2;
  {
    void* pin = nullptr;
    HFMI_TRY(ctx_pinned(ctx, 64 * sizeof(dcl_node), &pin));
    dcl_node* hnodes = (dcl_node*)pin;
    dcl_args da;
    da.n = n;
    da.ld = ld;
    da.evec = evec;
    da.Qg = Qg;
    da.S = A;
    da.Zv = Zv;
    da.Ds = Ds;
    da.Zs = Zs;
    da.dl = dl;
    da.wv = wvv;
    da.tauS = tauS;
    da.zhat = zhat;
    da.rc = rc;
    da.rs = rs;
    da.Col = Col;
    da.Live = Live;
    da.Ks = Ks;
    da.Src = Src;
    da.orgv = orgv;
    da.ra = ra;
    da.rb = rb;
    da.xkp = xkp;
    da.xli = xli;
    da.xkept = xkept;
    da.xlv = xlv;
    da.xd = xd;
    da.xz = xz;
    da.nodes = nodes;
    da.fail = fail;
    for (int L = Lf - 1; L >= 0; --L) {
      const int nn = 1 << L;
      da.L = L;
      da.D = Dcur;
      da.Dnew = Dnext;
      da.Q = Qcur;
      da.Qout = Qnext;
      hipLaunchKernelGGL(k_dcl_z, dim3(nn), dim3(1024), 0, st, da);
      hipLaunchKernelGGL(k_dcl_rank, dim3((n + 63) / 64), dim3(256), 0, st, da);
      const int cap = (int)round_up((n + nn - 1) / nn + 1, 64);
      if (cap > 9984) {      // the top merge beyond n = 9984: nothing of it in LDS
        hipLaunchKernelGGL(k_dcl_deflate<2>, dim3(nn), dim3(1024), 0, st, da, cap);
      } else if (cap > 4160) {      // the top merge beyond n = 4096
        const size_t defl_lds = (size_t)cap * 16;
        HIP_TRY(hipFuncSetAttribute((const void*)k_dcl_deflate<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)defl_lds));
        hipLaunchKernelGGL(k_dcl_deflate<1>, dim3(nn), dim3(1024), defl_lds, st, da, cap);
      } else {
        const size_t defl_lds = (size_t)cap * 30;
        HIP_TRY(hipFuncSetAttribute((const void*)k_dcl_deflate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)defl_lds));
        hipLaunchKernelGGL(k_dcl_deflate<0>, dim3(nn), dim3(1024), defl_lds, st, da, cap);
      }
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(hnodes, nodes, (size_t)nn * sizeof(dcl_node), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(ctx->ev_side, st));
      hipLaunchKernelGGL(k_dcl_rot, dim3((n + 255) / 256), dim3(256), 0, st, da);
      hipLaunchKernelGGL(k_dcl_secular, dim3((n + 3) / 4), dim3(256), 0, st, da);
      hipLaunchKernelGGL(k_dcl_zhat, dim3((n + 3) / 4), dim3(256), 0, st, da);
      hipLaunchKernelGGL(k_dcl_svec, dim3((n + 3) / 4), dim3(256), 0, st, da);
      hipLaunchKernelGGL(k_dcl_gather, dim3(n), dim3(256), 0, st, da);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventSynchronize(ctx->ev_side));
      for (int i = 0; i < nn; ++i) {
        const dcl_node& nd = hnodes[i];
        if (nd.K <= 0) continue;
        gemm_desc g;
        g.ta = g.tb = false;
        g.M = nd.hi - nd.lo;
        g.N = g.K = nd.K;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.A = Qg + (size_t)nd.lo * ld + nd.lo;
        g.B = A + (size_t)nd.lo * ld + nd.lo;
        g.lda = g.ldb = ld;
        g.C = Qnext + (size_t)nd.lo * ld + nd.lo;
        g.ldc = ld;
        HFMI_TRY(launch_dgemm(ctx, g));
      }
      std::swap(Dcur, Dnext);
      std::swap(Qcur, Qnext);
    }
  }
  clk.mark(3);

  // ---- eigenvalues to the host, output order
  std::vector<double> lam(n);
  int hfail[3] = {0, 0, 0};
  HIP_TRY(hipMemcpyAsync(lam.data(), Dcur, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hfail, fail, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (hfail[2]) HFMI_FAIL(HFMI_ERR_NUMERIC, "sym_eig (n=%d): the matrix has non-finite entries", n);
  if (hfail[0]) HFMI_FAIL(HFMI_ERR_NOT_CONVERGED, "sym_eig (n=%d): a secular equation did not converge", n);
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return sort_by_abs ? fabs(lam[x]) > fabs(lam[y]) : lam[x] > lam[y]; });
  for (int jx = 0; jx < n; ++jx) host_d[jx] = ldexp(lam[perm[jx]], hfail[1]);
  if (!host_V || nvec == 0) return HFMI_OK;
  const int nv = nvec;
  // the wanted eigenvectors of the tridiagonal matrix, in output order, into the other Q buffer: everything below works on nv columns
  HIP_TRY(hipMemcpyAsync(order, perm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_pick_columns, dim3(nv), dim3(256), 0, st, Qcur, Qnext, ld, n, order);
  HIP_TRY(hipGetLastError());
  std::swap(Qcur, Qnext);

  // ---- back-transformation: Z <- (I - V_0 T_0 V_0^T) ... (I - V_last T_last V_last^T) Z with block reflectors of WY = 256 | 512
  // columns (four panels): 64-column products V^T Z would be 64 tiles on 256 compute units.  The triangular factor of a block is
  // assembled from those of its panels: [V_a V_b] has T = [[T_a, -T_a (V_a^T V_b) T_b], [0, T_b]] (twice: 64 -> 128 -> 256).
  {
    double* Z = Qcur;
    double* Y = A;
    const int64_t sWY = (int64_t)WY * WY;
    gemm_desc g;
    g.ta = true;                 // Gram matrices of all blocks
    g.tb = false;
    g.M = g.N = WY;
    g.K = n;
    g.alpha = 1.0;
    g.beta = 0.0;
    g.A = g.B = Vh;
    g.lda = g.ldb = ld;
    g.C = Gm;
    g.ldc = WY;
    g.batch = nblk;
    g.sA = g.sB = (int64_t)WY * ld;
    g.sC = sWY;
    HFMI_TRY(launch_dgemm(ctx, g));
    HIP_TRY(hipMemsetAsync(Tf, 0, (size_t)nblk * sWY * sizeof(double), st));
    hipLaunchKernelGGL(k_larft, dim3(npanels), dim3(64), 0, st, Gm, tauv, Tf, WY);
    HIP_TRY(hipGetLastError());
    for (int w = NB; w < WY; w *= 2) {              // merge neighbours of width w into width 2 w
      for (int a0 = 0; a0 + 2 * w <= WY; a0 += 2 * w) {
        const size_t offTa = (size_t)a0 * (WY + 1), offTb = (size_t)(a0 + w) * (WY + 1);
        const size_t offX = (size_t)a0 + (size_t)(a0 + w) * WY;          // rows of a, columns of b
        gemm_desc m1;            // tmp = (V_a^T V_b) T_b
        m1.ta = m1.tb = false;
        m1.M = m1.N = m1.K = w;
        m1.alpha = 1.0;
        m1.beta = 0.0;
        m1.A = Gm + offX;
        m1.lda = WY;
        m1.B = Tf + offTb;
        m1.ldb = WY;
        m1.C = Tt + offX;
        m1.ldc = WY;
        m1.batch = nblk;
        m1.sA = m1.sB = m1.sC = sWY;
        HFMI_TRY(launch_dgemm(ctx, m1));
        gemm_desc m2 = m1;       // T_ab = -T_a tmp
        m2.alpha = -1.0;
        m2.A = Tf + offTa;
        m2.B = Tt + offX;
        m2.C = Tf + offX;
        HFMI_TRY(launch_dgemm(ctx, m2));
      }
    }
    gemm_desc gy;                // Y = V T, every block
    gy.ta = gy.tb = false;
    gy.M = n;
    gy.N = gy.K = WY;
    gy.alpha = 1.0;
    gy.beta = 0.0;
    gy.A = Vh;
    gy.lda = ld;
    gy.B = Tf;
    gy.ldb = WY;
    gy.C = Y;
    gy.ldc = ld;
    gy.batch = nblk;
    gy.sA = gy.sC = (int64_t)WY * ld;
    gy.sB = sWY;
    HFMI_TRY(launch_dgemm(ctx, gy));
    for (int bi = nblk - 1; bi >= 0; --bi) {
      // rows from p0 on: row p0 of the block's reflectors (and of Y = V T) is zero -- column p0 + c starts at row p0 + c + 1 --
      // so the products are the same as from p0 + 1, and every operand keeps its 16-byte alignment
      const int p0 = bi * WY, r0 = p0;
      if (p0 >= n - 2) continue;
      gemm_desc g1;              // W1 = V^T Z   (rows r0 ..)
      g1.ta = true;
      g1.tb = false;
      g1.M = WY;
      g1.N = nv;
      g1.K = n - r0;
      g1.alpha = 1.0;
      g1.beta = 0.0;
      g1.A = Vh + (size_t)p0 * ld + r0;
      g1.lda = ld;
      g1.B = Z + r0;
      g1.ldb = ld;
      g1.C = W1;
      g1.ldc = WY;
      HFMI_TRY(launch_dgemm(ctx, g1));
      gemm_desc g2;              // Z -= Y W1
      g2.ta = g2.tb = false;
      g2.M = n - r0;
      g2.N = nv;
      g2.K = WY;
      g2.alpha = -1.0;
      g2.beta = 1.0;
      g2.A = Y + (size_t)p0 * ld + r0;
      g2.lda = ld;
      g2.B = W1;
      g2.ldb = WY;
      g2.C = Z + r0;
      g2.ldc = ld;
      HFMI_TRY(launch_dgemm(ctx, g2));
    }
    clk.mark(4);
    double* out = Qnext;
    hipLaunchKernelGGL(k_out, dim3((n + 31) / 32, (nv + 31) / 32), dim3(256), 0, st, Z, ld, n, nv, out);
    HIP_TRY(hipGetLastError());
    HFMI_TRY(xfer_d2h(ctx, host_V, out, (size_t)n * nv * sizeof(double)));
    clk.mark(5);
  }
  if (clk.on)
    fprintf(stderr, "[hfmi eig n=%d] ms: workspace + fills %.3f | upload %.3f | load %.3f | tridiagonalisation %.3f | leaves %.3f | merges %.3f | back-transformation %.3f | output %.3f\n",
            n, clk.ms[6], clk.ms[7], clk.ms[0], clk.ms[1], clk.ms[2], clk.ms[3], clk.ms[4], clk.ms[5]);
  return HFMI_OK;
}
