// Small dense kernels of the Rayleigh-Ritz / Cholesky-QR steps: everything k x k (k <= 256) is done by
// ONE workgroup with the matrix resident in LDS (k <= 140: k*(k+1)*8 bytes <= 160 KB), fp64 throughout.
//
//   k_chol_inv   : G = R^T R (upper), optional diagonal shift on breakdown, R^{-1}, running R product
//   k_jacobi_eig : parallel cyclic Jacobi (round-robin ordering with the matrix kept in seat order, one 2x2-block
//                  rotation phase per round), rotations are logged and replayed on the identity by
//                  k_jacobi_vectors (one workgroup per eigenvector-matrix row) so the eigenvector update never
//                  sits on the critical path.
#include "hfmi_internal.h"

#include <stdlib.h>
#include <string.h>
#define SMALL_THREADS 1024
#define EPS_D 2.220446049250313e-16
// workgroup size of the one-workgroup kernels (tuning knob for experiments: HFMI_SMALL_THREADS)
static int small_threads() {
  static int v = 0;
  if (!v) {
    const char* e = getenv("HFMI_SMALL_THREADS");
    v = e ? atoi(e) : SMALL_THREADS;
    if (v < 64 || v > 1024 || (v & 63)) v = SMALL_THREADS;
  }
  return v;
}

__device__ __forceinline__ double block_sum(double v, double* scratch /* >= 16 doubles, LDS */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) s += scratch[w];
  return s;
}
__device__ __forceinline__ double block_min(double v, double* scratch) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = scratch[0];
  const int nw = blockDim.x >> 6;
  for (int w = 1; w < nw; ++w) s = fmin(s, scratch[w]);
  return s;
}

// ------------------------------------------------------------------------------------------------
// Cholesky + triangular inverse.  M points at the working k x k matrix (LDS when it fits, else a global
// scratch slot), row-major with leading dimension ldm.
//
// One workgroup; everything here is issue/latency bound on a single CU (fp64 VALU is quarter rate: a dependent
// v_fma_f64 costs 32 cycles with 4 waves per SIMD; a software fp64 divide ~300), so the structure avoids
// redundant scalar fp64 math (v_rsq_f64 + Newton instead of sqrt and divide on every thread), maps triangular
// index sets onto ALL lanes (folded triangle, reciprocal-multiply index split) and spends ONE barrier per column
// of the factorisation and one per row of the inverse.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rsqrt(double x) {
  // v_rsq_f64 is good to ~2^-26; ONE third-order step y0 (1 + e/2 + 3 e^2/8), e = 1 - x y0^2, leaves ~e^3 = 2^-78:
  // five dependent fp64 operations (32 cycles each on one CU) instead of the eight of two Newton steps -- this sits
  // on the serial chain of every Cholesky column and every Jacobi round.
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y0), y0, 1.0);
  const double q = e * fma(0.375, e, 0.5);
  return fma(y0, q, y0);
}
// cell e of the folded triangle {(row, col): 0 <= row <= col < n}: ceil(n/2) strips of n+1 cells, strip r holds
// row r (n-r cells) followed by row n-1-r (r+1 cells).  Returns false for the duplicate half of the middle strip.
__device__ __forceinline__ bool tri_cell(int e, int n, float inv_np1, int& row, int& col) {
  int r = (int)((float)e * inv_np1);
  int cc = e - r * (n + 1);
  if (cc < 0) {
    --r;
    cc += n + 1;
  } else if (cc > n) {
    ++r;
    cc -= n + 1;
  }
  if (cc < n - r) {
    row = r;
    col = r + cc;
    return true;
  }
  const int r2 = n - 1 - r;
  row = r2;
  col = r2 + (cc - (n - r));
  return r2 != r;
}

// Generic path (any k <= 256; the matrix in LDS when it fits, else in a global scratch slot -- the pointer is then
// generic and the accesses are FLAT: this kernel is the fallback, k_chol_reg below is the fast path for k <= 139).
__global__ __launch_bounds__(SMALL_THREADS) void k_chol_inv(const double* __restrict__ G, int ldg, int k,
                                                            double* __restrict__ Rout, double* __restrict__ Rinv,
                                                            double* __restrict__ Rtot, double* __restrict__ Rtmp,
                                                            int ldo, int rtot_mode, int full_r, double shift_rel,
                                                            double pivot_tol, double* __restrict__ gscratch,
                                                            int use_lds, double* __restrict__ colnorm0,
                                                            double* __restrict__ rdiag,
                                                            hfmi_status_words* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);       // 32 doubles of reduction scratch
  double* diag0 = red + 32;                            // k original diagonal entries
  double* invd = diag0 + 256;                          // 1 / R_jj
  double* ird0 = invd + 256;                           // 1 / (original diagonal + shift): pivot-ratio bookkeeping
  double* lds_m = ird0 + 256;
  const int ldm = use_lds ? (k | 1) : ldo;
  double* M = use_lds ? lds_m : gscratch;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  __shared__ int s_break;

  long long tk0 = clock64(), tk1, tk2, tk3, tk4;
  // diag and orthonormality defect || D^-1/2 G D^-1/2 - I ||_F of the input
  for (int i = tid; i < k; i += nthr) {
    diag0[i] = G[i * ldg + i];
    if (rtot_mode == 1) colnorm0[i] = sqrt(fmax(diag0[i], 0.0));  // norms of the ORIGINAL columns (first pass)
  }
  __syncthreads();
  for (int i = tid; i < k; i += nthr) invd[i] = diag0[i] > 0.0 ? fast_rsqrt(diag0[i]) : 0.0;
  __syncthreads();
  double dev = 0.0, tr = 0.0;
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) {
      const double g = 0.5 * (G[i * ldg + j] + G[j * ldg + i]);
      const double x = g * invd[i] * invd[j] - (i == j ? 1.0 : 0.0);
      dev += x * x;
      if (i == j) tr += g;
    }
  dev = block_sum(dev, red);
  tr = block_sum(tr, red);
  tk1 = clock64();

  int shifted = 0, failed = 0;
  double min_ratio = 1e300;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const double shift = attempt ? shift_rel * tr : 0.0;
    for (int i = wave; i < k; i += nw)
      for (int j = lane; j < k; j += 64) {
        double g = 0.5 * (G[i * ldg + j] + G[j * ldg + i]);
        if (i == j) g += shift;
        M[i * ldm + j] = (j >= i) ? g : 0.0;
      }
    if (tid == 0) s_break = 0;
    for (int i = tid; i < k; i += nthr) ird0[i] = (diag0[i] + shift > 0.0) ? 1.0 / (diag0[i] + shift) : 0.0;
    __syncthreads();
    double ratio_local = 1e300;
    // Unscaled (LDL^T-style) right-looking factorisation: U[i][c] = G[i][c] - sum_{j<i} U[j][i] U[j][c] / U[j][j].
    // Row j is never rewritten once it is the pivot row, so a column step is ONE barrier phase: every thread reads
    // the pivot, forms 1 / U[j][j] itself (v_rsq_f64 + Newton, squared) and updates its share of the trailing
    // triangle.  R = diag(U)^{-1/2} U is formed by one scaling pass at the end.
    for (int j = 0; j < k; ++j) {
      const double piv = M[j * ldm + j];
      const double ref = diag0[j] + shift;
      if (!(piv > pivot_tol * ref) || !(ref > 0.0)) {  // uniform decision: every thread reads the same words
        if (tid == 0) s_break = 1;
        break;
      }
      const double inv = fast_rsqrt(piv);
      const double rp = inv * inv;
      if (tid == 0) {
        ratio_local = fmin(ratio_local, piv * ird0[j]);
        invd[j] = inv;
      }
      // trailing update of the upper triangle: M[i][c] -= U[j][i] U[j][c] / U[j][j], j < i <= c (folded onto all lanes)
      const int n = k - j - 1;
      if (n > 0) {
        const float inv_np1 = __frcp_rn((float)(n + 1));
        const int cells = ((n + 1) >> 1) * (n + 1);
        const double* rj = M + j * ldm + j + 1;
        for (int e = tid; e < cells; e += nthr) {
          int a, b;
          if (tri_cell(e, n, inv_np1, a, b)) M[(j + 1 + a) * ldm + j + 1 + b] -= rj[a] * rp * rj[b];
        }
      }
      __syncthreads();
    }
    __syncthreads();
    if (!s_break) {
      min_ratio = ratio_local;
      break;
    }
    if (attempt == 0) shifted = 1;
    else failed = 1;
    __syncthreads();
  }
  if (failed) {
    if (tid == 0) {
      status->min_pivot_ratio = 0.0;
      status->gram_dev = sqrt(dev);
      status->shifted = shifted;
      status->failed = 1;
    }
    return;
  }
  // R = diag(U)^{-1/2} U
  for (int i = wave; i < k; i += nw) {
    const double sc = invd[i];
    for (int j = i + lane; j < k; j += 64) M[i * ldm + j] *= sc;
  }
  __syncthreads();
  tk2 = clock64();
  // R out (upper triangle, zeros below)
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) Rout[i * ldo + j] = (j >= i) ? M[i * ldm + j] : 0.0;
  // Inverse X = R^-1 by back substitution, one ROW per barrier phase (bottom up): x_cc = 1/R_cc and, for i < c,
  // x_ic = -(sum_{l=i+1..c} R[i][l] x_lc) / R_ii.  All columns c > i of row i are independent: one 8-lane group per
  // column, both factors of the dot product contiguous along l (x is kept in the unused strictly lower triangle,
  // X[i][c] -> M[c][i]).
  {
    const int g8 = tid & 7, grp8 = tid >> 3, ngrp8 = nthr >> 3;
    for (int i = k - 2; i >= 0; --i) {
      const double* ri = M + i * ldm;
      const double mi = -invd[i];
      for (int c = i + 1 + grp8; c < k; c += ngrp8) {
        const double* xr = M + c * ldm;
        const double xc = invd[c];
        double sacc = 0.0;
        for (int l = i + 1 + g8; l <= c; l += 8) sacc += ri[l] * (l == c ? xc : xr[l]);
        sacc += __shfl_xor(sacc, 4, 64);
        sacc += __shfl_xor(sacc, 2, 64);
        sacc += __shfl_xor(sacc, 1, 64);
        if (g8 == 0) M[c * ldm + i] = sacc * mi;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  tk3 = clock64();
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) Rinv[i * ldo + j] = (j > i) ? M[j * ldm + i] : (j == i ? invd[i] : 0.0);
  // diagonal of the running product R = R_p ... R_1 (its ratio to the original column norms exposes
  // numerically dependent columns); the full product only when the caller wants R
  for (int i = tid; i < k; i += nthr) rdiag[i] = (rtot_mode == 1 ? 1.0 : rdiag[i]) * M[i * ldm + i];
  if (full_r) {
    if (rtot_mode == 1) {
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) Rtot[i * ldo + j] = (j >= i) ? M[i * ldm + j] : 0.0;
    } else {
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) {
          double acc = 0.0;
          if (j >= i)
            for (int l = i; l <= j; ++l) acc += M[i * ldm + l] * Rtot[l * ldo + j];
          Rtmp[i * ldo + j] = acc;
        }
      __syncthreads();
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) Rtot[i * ldo + j] = Rtmp[i * ldo + j];
    }
  }
  tk4 = clock64();
  if (tid == 0) {
    status->min_pivot_ratio = min_ratio;
    status->gram_dev = sqrt(dev);
    status->shifted = shifted;
    status->failed = 0;
    status->tick[0] = tk1 - tk0;  // load + defect
    status->tick[1] = tk2 - tk1;  // Cholesky
    status->tick[2] = tk3 - tk2;  // inverse
    status->tick[3] = tk4 - tk3;  // outputs + R product
    status->tick[4] = 0;

  }
}

// ------------------------------------------------------------------------------------------------
// k_chol_reg: the same contract as k_chol_inv for an LDS-resident matrix (k <= 139), built around what one CU
// actually charges for: INSTRUCTION ISSUE (one VALU and one scalar instruction per SIMD every 4 cycles, shared by
// the waves of the SIMD) and the 32-cycle dependent fp64 latency -- not flops.  512 threads; the upper triangle is
// enumerated bottom row first and dealt cyclically, entry e -> thread e % 512, register slot e / 512, so slot q of
// all threads is one band of adjacent rows.  A factorisation step publishes the pivot row into a fixed LDS row
// buffer (static read addresses), takes one barrier, and updates whole slot groups: per entry two ds_reads, one
// mul, one fma, no mask (an entry whose row is finished is dead: it has been published and may be overwritten
// with garbage), one wave-uniform test per GROUP of four slots, all loads of a group issued before their first
// use.  The inverse runs the same way bottom-up (see below).
// ------------------------------------------------------------------------------------------------
#define CHOL_REG_THREADS 512
template <int EPT>
__global__ __launch_bounds__(CHOL_REG_THREADS, 2) void k_chol_reg(const double* __restrict__ G, int ldg, int k,
                                                               double* __restrict__ Rout, double* __restrict__ Rinv,
                                                               double* __restrict__ Rtot, double* __restrict__ Rtmp,
                                                               int ldo, int rtot_mode, int full_r, double shift_rel,
                                                               double pivot_tol, double* __restrict__ colnorm0,
                                                               double* __restrict__ rdiag,
                                                               hfmi_status_words* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);       // 32 doubles of reduction scratch
  double* diag0 = red + 32;                            // k original diagonal entries
  double* invd = diag0 + 256;                          // 1 / R_jj
  double* rb0 = invd + 256;                            // published row, two alternating buffers
  double* rb1 = rb0 + 256;
  double* M = rb1 + 256;                               // U, then R (upper triangle), row-major
  const int ldm = k | 1;
  constexpr int NT = CHOL_REG_THREADS;
  constexpr int GQ = 4;                                // slots per group
  constexpr int NG = (EPT + GQ - 1) / GQ;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, nw = NT >> 6;
  __shared__ int s_break;

  long long tk0 = clock64(), tk1, tk2, tk3, tk4;
  for (int i = tid; i < k; i += NT) {
    diag0[i] = G[i * ldg + i];
    if (rtot_mode == 1) colnorm0[i] = sqrt(fmax(diag0[i], 0.0));  // norms of the ORIGINAL columns (first pass)
  }
  for (int i = tid; i < 512; i += NT) rb0[i] = 0.0;               // both row buffers
  __syncthreads();
  for (int i = tid; i < k; i += NT) invd[i] = diag0[i] > 0.0 ? fast_rsqrt(diag0[i]) : 0.0;
  __syncthreads();
  double dev = 0.0, tr = 0.0;
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) {
      const double g = 0.5 * (G[i * ldg + j] + G[j * ldg + i]);
      const double x = g * invd[i] * invd[j] - (i == j ? 1.0 : 0.0);
      dev += x * x;
      if (i == j) tr += g;
    }
  dev = block_sum(dev, red);
  tr = block_sum(tr, red);
  tk1 = clock64();

  // Near-orthonormal input (the polishing pass of Cholesky-QR2: G = D^1/2 (I + E) D^1/2 with |E| ~ eps cond^2): the
  // sequential factorisation is replaced by the first-order inverse square root X = D^-1/2 (I - E/2), for which
  // X^T G X = I - 3/4 E^2 + O(E^3) -- below 1e-14 for |E|_F < 1e-7.  X is not triangular, so this is only taken when
  // the caller does not ask for the triangular factor (the fused solves: only the span and the B-orthonormality of Q
  // matter there).  k = 138: 0.27 ms -> 0.02 ms for the second pass of every solve.
  if (!full_r && dev < 1e-14) {
    bool pos = true;
    for (int i = tid; i < k; i += NT) pos = pos && diag0[i] > 0.0;
    if (__syncthreads_and(pos ? 1 : 0)) {
      for (int i = wave; i < k; i += nw) {
        const double di = invd[i];
        for (int j = lane; j < k; j += 64) {
          const double g = 0.5 * (G[i * ldg + j] + G[j * ldg + i]);
          const double dj = invd[j];
          const double e = g * di * dj - (i == j ? 1.0 : 0.0);
          const double id = (i == j) ? 1.0 : 0.0;
          Rinv[i * ldo + j] = di * (id - 0.5 * e);
          Rout[i * ldo + j] = (id + 0.5 * e) * diag0[j] * dj;          // (I + E/2) D^1/2, d^1/2 = d * d^-1/2
          if (i == j) rdiag[i] = (rtot_mode == 1 ? 1.0 : rdiag[i]) * (1.0 + 0.5 * e) * diag0[i] * di;
        }
      }
      if (tid == 0) {
        status->min_pivot_ratio = 1.0;
        status->gram_dev = sqrt(dev);
        status->shifted = 0;
        status->failed = 0;
        status->tick[0] = tk1 - tk0;
        status->tick[1] = 0;
        status->tick[2] = 0;
        status->tick[3] = clock64() - tk1;
        status->tick[4] = 3;
      }
      return;
    }
  }

  // owned entries: slot q holds cell e = tid + q * NT; rows counted from the bottom, m (m+1)/2 <= e < (m+1)(m+2)/2
  const int total = k * (k + 1) / 2;
  int oi[EPT], oc[EPT];   // row (clamped into the matrix for idle slots: they compute garbage nobody reads), column
  bool own[EPT];
  double v[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = tid + q * NT;
    own[q] = e < total;
    const int ee = own[q] ? e : 0;
    int m = (int)((sqrtf(8.0f * (float)ee + 1.0f) - 1.0f) * 0.5f);
    while (m * (m + 1) / 2 > ee) --m;
    while ((m + 1) * (m + 2) / 2 <= ee) ++m;
    oi[q] = k - 1 - m;
    oc[q] = (k - 1 - m) + (ee - m * (m + 1) / 2);
    v[q] = 0.0;
  }

  int shifted = 0, failed = 0;
  double shift = 0.0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    shift = attempt ? shift_rel * tr : 0.0;
    if (tid == 0) s_break = 0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      v[q] = 0.5 * (G[oi[q] * ldg + oc[q]] + G[oc[q] * ldg + oi[q]]);
      if (oi[q] == oc[q]) v[q] += shift;
    }
    // Unscaled (LDL^T-style) right-looking factorisation: U[i][c] = G[i][c] - sum_{j<i} U[j][i] U[j][c] / U[j][j];
    // R = diag(U)^{-1/2} U is formed by one scaling pass at the end.
    for (int j = 0; j < k; ++j) {
      double* rb = (j & 1) ? rb1 : rb0;
      const int m = k - 1 - j;                            // row j counted from the bottom
      const int e0 = m * (m + 1) / 2;                     // its cells are e0 .. e0 + m (k - j of them)
      const int qp0 = e0 / NT, qp1 = (e0 + m) / NT;       // the (at most two) slots that hold them
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (g * GQ <= qp1 && qp0 < g * GQ + GQ) {
#pragma unroll
          for (int q = g * GQ; q < g * GQ + GQ && q < EPT; ++q)
            if (own[q] && oi[q] == j) {                   // row j is final: publish it
              rb[oc[q]] = v[q];
              M[j * ldm + oc[q]] = v[q];
            }
        }
      __syncthreads();
      const double piv = rb[j];
      const double ref = diag0[j] + shift;
      if (!(piv > pivot_tol * ref) || !(ref > 0.0)) {     // uniform decision: every thread reads the same words
        if (tid == 0) s_break = 1;
        break;
      }
      const double inv = fast_rsqrt(piv);
      const double nrp = -(inv * inv);
      if (tid == 0) invd[j] = inv;
      // slots 0 .. qa-1 still hold rows > j (cells below row j are exactly e < e0)
      const int qa = (e0 + NT - 1) / NT;
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (g * GQ < qa) {
          double xa[GQ], xb[GQ];
#pragma unroll
          for (int u = 0; u < GQ; ++u)
            if (g * GQ + u < EPT) {
              xa[u] = rb[oi[g * GQ + u]];
              xb[u] = rb[oc[g * GQ + u]];
            }
#pragma unroll
          for (int u = 0; u < GQ; ++u)
            if (g * GQ + u < EPT) v[g * GQ + u] = fma(xa[u] * nrp, xb[u], v[g * GQ + u]);
        }
    }
    __syncthreads();
    if (!s_break) break;
    if (attempt == 0) shifted = 1;
    else failed = 1;
    __syncthreads();
  }
  if (failed) {
    if (tid == 0) {
      status->min_pivot_ratio = 0.0;
      status->gram_dev = sqrt(dev);
      status->shifted = shifted;
      status->failed = 1;
    }
    return;
  }
  // smallest pivot relative to the (shifted) original diagonal: piv_j = 1 / invd_j^2
  double ratio = 1e300;
  for (int j = tid; j < k; j += NT) ratio = fmin(ratio, 1.0 / (invd[j] * invd[j] * (diag0[j] + shift)));
  const double min_ratio = block_min(ratio, red);
  // R = diag(U)^{-1/2} U
  for (int i = wave; i < k; i += nw) {
    const double sc = invd[i];
    for (int j = i + lane; j < k; j += 64) M[i * ldm + j] *= sc;
  }
  for (int i = tid; i < 512; i += NT) rb0[i] = 0.0;  // the inverse relies on zeros left of the published row
  __syncthreads();
  tk2 = clock64();
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) Rout[i * ldo + j] = (j >= i) ? M[i * ldm + j] : 0.0;
  // Inverse X = R^-1, right-looking and bottom-up: once row l of X is final (x_ll = 1/R_ll, x_lc = -acc / R_ll) it
  // goes to the output and into the row buffer, and every entry (i, c) with i < l accumulates R[i][l] x_lc -- x_lc
  // reads as 0 for c < l (the buffers are zeroed and row l only writes c >= l), finished entries (i >= l) are
  // dead, so again no masks.
#pragma unroll
  for (int q = 0; q < EPT; ++q) v[q] = 0.0;
  for (int l = k - 1; l >= 0; --l) {
    double* xr = (l & 1) ? rb1 : rb0;
    const int m = k - 1 - l;
    const int e0 = m * (m + 1) / 2;
    const int qp0 = e0 / NT, qp1 = (e0 + m) / NT;
    const double il = invd[l];
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (g * GQ <= qp1 && qp0 < g * GQ + GQ) {
#pragma unroll
        for (int q = g * GQ; q < g * GQ + GQ && q < EPT; ++q)
          if (own[q] && oi[q] == l) {
            const double x = (oc[q] == l) ? il : -il * v[q];
            xr[oc[q]] = x;
            Rinv[l * ldo + oc[q]] = x;
          }
      }
    __syncthreads();
    // slots qs .. EPT-1 hold rows < l (cells of rows above row l are exactly e >= e0 + m + 1)
    const int qs = (e0 + m + 1) / NT;
    const double* ml = M + l;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (g * GQ + GQ > qs) {
        double xa[GQ], xb[GQ];
#pragma unroll
        for (int u = 0; u < GQ; ++u)
          if (g * GQ + u < EPT) {
            xa[u] = ml[oi[g * GQ + u] * ldm];
            xb[u] = xr[oc[g * GQ + u]];
          }
#pragma unroll
        for (int u = 0; u < GQ; ++u)
          if (g * GQ + u < EPT) v[g * GQ + u] = fma(xa[u], xb[u], v[g * GQ + u]);
      }
  }
  __syncthreads();
  tk3 = clock64();
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < i; j += 64) Rinv[i * ldo + j] = 0.0;
  // diagonal of the running product R = R_p ... R_1 (its ratio to the original column norms exposes
  // numerically dependent columns); the full product only when the caller wants R
  for (int i = tid; i < k; i += NT) rdiag[i] = (rtot_mode == 1 ? 1.0 : rdiag[i]) * M[i * ldm + i];
  if (full_r) {
    if (rtot_mode == 1) {
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) Rtot[i * ldo + j] = (j >= i) ? M[i * ldm + j] : 0.0;
    } else {
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) {
          double acc = 0.0;
          if (j >= i)
            for (int l = i; l <= j; ++l) acc += M[i * ldm + l] * Rtot[l * ldo + j];
          Rtmp[i * ldo + j] = acc;
        }
      __syncthreads();
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) Rtot[i * ldo + j] = Rtmp[i * ldo + j];
    }
  }
  tk4 = clock64();
  if (tid == 0) {
    status->min_pivot_ratio = min_ratio;
    status->gram_dev = sqrt(dev);
    status->shifted = shifted;
    status->failed = 0;
    status->tick[0] = tk1 - tk0;  // load + defect
    status->tick[1] = tk2 - tk1;  // Cholesky
    status->tick[2] = tk3 - tk2;  // inverse
    status->tick[3] = tk4 - tk3;  // outputs + R product
    status->tick[4] = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// k_chol_tile: k_chol_reg with a two-dimensional ownership map.  With the entries dealt cyclically every entry of a
// thread needs its own two LDS words per step (the pivot row at its row and at its column): 2 EPT reads per step,
// 164 KB per step at k = 138 -- the factorisation was LDS-bandwidth bound (1.3 k of its 2 k cycles per step).  Here a
// thread owns a TR x TC rectangle of the upper triangle and reads TR + TC words for TR TC entries (10 instead of 50
// at k = 138); everything else is the same algorithm: unscaled right-looking factorisation with the pivot row
// published through alternating LDS row buffers, one barrier per column, dead entries instead of masks, and the
// inverse accumulated the same way bottom-up.
// ------------------------------------------------------------------------------------------------
// NTHR = 512, MG = false: k <= 139, the factor in LDS.  NTHR = 1024, MG = true (round 3): 140 <= k <= 256, where k x k doubles
// no longer fit the 160 KB of LDS: the tiles still live in registers (6 x 6 per thread at k = 256), only the copy of the
// factor that the inverse reads column by column goes to an L2-resident global slot, and those reads do not depend on the
// step's barrier, so they are issued one step ahead.  (The generic k_chol_inv with its FLAT accesses took 3.0 ms at k = 256.)
template <int TR, int TC, int NTHR = CHOL_REG_THREADS, bool MG = false>
__global__ __launch_bounds__(NTHR, NTHR == 1024 ? 1 : 2) void k_chol_tile(const double* __restrict__ G, int ldg, int k,
                                                                double* __restrict__ Rout, double* __restrict__ Rinv,
                                                                double* __restrict__ Rtot, double* __restrict__ Rtmp,
                                                                int ldo, int rtot_mode, int full_r, double shift_rel,
                                                                double pivot_tol, double* __restrict__ colnorm0,
                                                                double* __restrict__ rdiag,
                                                                hfmi_status_words* __restrict__ status,
                                                                double* __restrict__ Mglob) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);       // 32 doubles of reduction scratch
  double* diag0 = red + 32;                            // k original diagonal entries
  double* invd = diag0 + 256;                          // 1 / R_jj
  double* rb0 = invd + 256;                            // published row, two alternating buffers
  double* rb1 = rb0 + 256;
  // U, then R (upper triangle), row-major: in LDS, or (MG) in the global slot -- two differently typed names so that every
  // access compiles to ds_* or global_* and never to FLAT
  double* Ml = rb1 + 256;
  const int ldm = MG ? ldo : (k | 1);
  constexpr int NT = NTHR;
#define CHOL_M(idx) (*(MG ? (Mglob + (idx)) : (Ml + (idx))))
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, nw = NT >> 6;
  __shared__ int s_break;

  long long tk0 = clock64(), tk1, tk2, tk3, tk4;
  for (int i = tid; i < k; i += NT) {
    diag0[i] = G[i * ldg + i];
    if (rtot_mode == 1) colnorm0[i] = sqrt(fmax(diag0[i], 0.0));  // norms of the ORIGINAL columns (first pass)
  }
  for (int i = tid; i < 512; i += NT) rb0[i] = 0.0;               // both row buffers
  __syncthreads();
  for (int i = tid; i < k; i += NT) invd[i] = diag0[i] > 0.0 ? fast_rsqrt(diag0[i]) : 0.0;
  __syncthreads();
  double dev = 0.0, tr = 0.0;
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) {
      const double g = 0.5 * (G[i * ldg + j] + G[j * ldg + i]);
      const double x = g * invd[i] * invd[j] - (i == j ? 1.0 : 0.0);
      dev += x * x;
      if (i == j) tr += g;
    }
  dev = block_sum(dev, red);
  tr = block_sum(tr, red);
  tk1 = clock64();

  // near-orthonormal input and no triangular factor wanted: first-order inverse square root (see k_chol_reg)
  if (!full_r && dev < 1e-14) {
    bool pos = true;
    for (int i = tid; i < k; i += NT) pos = pos && diag0[i] > 0.0;
    if (__syncthreads_and(pos ? 1 : 0)) {
      for (int i = wave; i < k; i += nw) {
        const double di = invd[i];
        for (int j = lane; j < k; j += 64) {
          const double g = 0.5 * (G[i * ldg + j] + G[j * ldg + i]);
          const double dj = invd[j];
          const double e = g * di * dj - (i == j ? 1.0 : 0.0);
          const double id = (i == j) ? 1.0 : 0.0;
          Rinv[i * ldo + j] = di * (id - 0.5 * e);
          Rout[i * ldo + j] = (id + 0.5 * e) * diag0[j] * dj;
          if (i == j) rdiag[i] = (rtot_mode == 1 ? 1.0 : rdiag[i]) * (1.0 + 0.5 * e) * diag0[i] * di;
        }
      }
      if (tid == 0) {
        status->min_pivot_ratio = 1.0;
        status->gram_dev = sqrt(dev);
        status->shifted = 0;
        status->failed = 0;
        status->tick[0] = tk1 - tk0;
        status->tick[1] = 0;
        status->tick[2] = 0;
        status->tick[3] = clock64() - tk1;
        status->tick[4] = 3;
      }
      return;
    }
  }

  // this thread's tile: the tid-th rectangle (tile row I ascending, then tile column J) that touches the upper triangle
  const int ntr = (k + TR - 1) / TR, ntc = (k + TC - 1) / TC;
  bool own = false;
  int r0 = 0, c0 = 0;
  {
    int left = tid;
    for (int I = 0; I < ntr; ++I) {
      int jmin = (I * TR - TC + 1 + TC - 1) / TC;          // smallest J with J TC + TC - 1 >= I TR
      if (I * TR - TC + 1 <= 0) jmin = 0;
      const int cnt = ntc - jmin;
      if (cnt <= 0) continue;
      if (left < cnt) {
        own = true;
        r0 = I * TR;
        c0 = (jmin + left) * TC;
        break;
      }
      left -= cnt;
    }
  }
  double v[TR][TC];

  int shifted = 0, failed = 0;
  double shift = 0.0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    shift = attempt ? shift_rel * tr : 0.0;
    if (tid == 0) s_break = 0;
#pragma unroll
    for (int a = 0; a < TR; ++a)
#pragma unroll
      for (int b = 0; b < TC; ++b) {
        const int r = r0 + a, c = c0 + b;
        double g = 0.0;
        if (own && r < k && c < k && c >= r) g = 0.5 * (G[r * ldg + c] + G[c * ldg + r]) + (r == c ? shift : 0.0);
        v[a][b] = g;
      }
    for (int j = 0; j < k; ++j) {
      double* rb = (j & 1) ? rb1 : rb0;
      if (own && j >= r0 && j < r0 + TR) {                 // row j is final: publish it
#pragma unroll
        for (int a = 0; a < TR; ++a)
          if (r0 + a == j) {
#pragma unroll
            for (int b = 0; b < TC; ++b) {
              const int c = c0 + b;
              if (c >= j && c < k) {
                rb[c] = v[a][b];
                CHOL_M(j * ldm + c) = v[a][b];
              }
            }
          }
      }
      __syncthreads();
      const double piv = rb[j];
      const double ref = diag0[j] + shift;
      if (!(piv > pivot_tol * ref) || !(ref > 0.0)) {     // uniform decision: every thread reads the same words
        if (tid == 0) s_break = 1;
        break;
      }
      const double inv = fast_rsqrt(piv);
      const double nrp = -(inv * inv);
      if (tid == 0) invd[j] = inv;
      if (own && r0 + TR - 1 > j) {                        // the tile still holds rows below row j
        double xa[TR], xb[TC];
#pragma unroll
        for (int a = 0; a < TR; ++a) xa[a] = rb[r0 + a];
#pragma unroll
        for (int b = 0; b < TC; ++b) xb[b] = rb[c0 + b];
#pragma unroll
        for (int a = 0; a < TR; ++a) {
          const double f = xa[a] * nrp;
#pragma unroll
          for (int b = 0; b < TC; ++b) v[a][b] = fma(f, xb[b], v[a][b]);
        }
      }
    }
    __syncthreads();
    if (!s_break) break;
    if (attempt == 0) shifted = 1;
    else failed = 1;
    __syncthreads();
  }
  if (failed) {
    if (tid == 0) {
      status->min_pivot_ratio = 0.0;
      status->gram_dev = sqrt(dev);
      status->shifted = shifted;
      status->failed = 1;
    }
    return;
  }
  double ratio = 1e300;
  for (int j = tid; j < k; j += NT) ratio = fmin(ratio, 1.0 / (invd[j] * invd[j] * (diag0[j] + shift)));
  const double min_ratio = block_min(ratio, red);
  // R = diag(U)^{-1/2} U
  for (int i = wave; i < k; i += nw) {
    const double sc = invd[i];
    for (int j = i + lane; j < k; j += 64) CHOL_M(i * ldm + j) *= sc;
  }
  for (int i = tid; i < 512; i += NT) rb0[i] = 0.0;  // the inverse relies on zeros left of the published row
  __syncthreads();
  tk2 = clock64();
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) Rout[i * ldo + j] = (j >= i) ? CHOL_M(i * ldm + j) : 0.0;
  // Inverse X = R^-1, right-looking and bottom-up (see k_chol_reg)
#pragma unroll
  for (int a = 0; a < TR; ++a)
#pragma unroll
    for (int b = 0; b < TC; ++b) v[a][b] = 0.0;
  double xan[TR];                                        // column l of R at this tile's rows, fetched one step ahead
#pragma unroll
  for (int a = 0; a < TR; ++a) {
    const int r = r0 + a < k ? r0 + a : k - 1;
    xan[a] = CHOL_M(r * ldm + (k - 1));
  }
  for (int l = k - 1; l >= 0; --l) {
    double* xr = (l & 1) ? rb1 : rb0;
    const double il = invd[l];
    if (own && l >= r0 && l < r0 + TR) {
#pragma unroll
      for (int a = 0; a < TR; ++a)
        if (r0 + a == l) {
#pragma unroll
          for (int b = 0; b < TC; ++b) {
            const int c = c0 + b;
            if (c >= l && c < k) {
              const double x = (c == l) ? il : -il * v[a][b];
              xr[c] = x;
              Rinv[l * ldo + c] = x;
            }
          }
        }
    }
    __syncthreads();
    double xa[TR];
#pragma unroll
    for (int a = 0; a < TR; ++a) {
      xa[a] = xan[a];
      const int r = r0 + a < k ? r0 + a : k - 1;
      xan[a] = CHOL_M(r * ldm + (l > 0 ? l - 1 : 0));     // R is final: this read does not wait for the next barrier
    }
    if (own && r0 < l) {                                   // the tile still holds rows above row l
      double xb[TC];
#pragma unroll
      for (int b = 0; b < TC; ++b) xb[b] = xr[c0 + b];
#pragma unroll
      for (int a = 0; a < TR; ++a)
#pragma unroll
        for (int b = 0; b < TC; ++b) v[a][b] = fma(xa[a], xb[b], v[a][b]);
    }
  }
  __syncthreads();
  tk3 = clock64();
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < i; j += 64) Rinv[i * ldo + j] = 0.0;
  for (int i = tid; i < k; i += NT) rdiag[i] = (rtot_mode == 1 ? 1.0 : rdiag[i]) * CHOL_M(i * ldm + i);
  if (full_r) {
    if (rtot_mode == 1) {
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) Rtot[i * ldo + j] = (j >= i) ? CHOL_M(i * ldm + j) : 0.0;
    } else {
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) {
          double acc = 0.0;
          if (j >= i)
            for (int l = i; l <= j; ++l) acc += CHOL_M(i * ldm + l) * Rtot[l * ldo + j];
          Rtmp[i * ldo + j] = acc;
        }
      __syncthreads();
      for (int i = wave; i < k; i += nw)
        for (int j = lane; j < k; j += 64) Rtot[i * ldo + j] = Rtmp[i * ldo + j];
    }
  }
  tk4 = clock64();
  if (tid == 0) {
    status->min_pivot_ratio = min_ratio;
    status->gram_dev = sqrt(dev);
    status->shifted = shifted;
    status->failed = 0;
    status->tick[0] = tk1 - tk0;
    status->tick[1] = tk2 - tk1;
    status->tick[2] = tk3 - tk2;
    status->tick[3] = tk4 - tk3;
    status->tick[4] = 0;
  }
}
#undef CHOL_M

static int chol_tiles(int k, int trr, int tcc) {
  const int ntr = (k + trr - 1) / trr, ntc = (k + tcc - 1) / tcc;
  int n = 0;
  for (int I = 0; I < ntr; ++I)
    for (int J = 0; J < ntc; ++J)
      if (J * tcc + tcc - 1 >= I * trr) ++n;
  return n;
}

static int g_chol_default = -1;
int chol_tuning_set(const char* key, int value) {
  if (key && !strcmp(key, "chol") && (value == 0 || value == 1)) {
    g_chol_default = value;
    return 1;
  }
  return 0;
}
int launch_chol_inv(hfmi_ctx* ctx, int k, int slot_gram, int slot_r, int slot_rinv, int slot_rtot, int rtot_mode,
                    int full_r, double shift_rel, double pivot_tol) {
  if (k < 1 || k > SM_MAXK) HFMI_FAIL(HFMI_ERR_INVALID, "chol_inv: k=%d out of range", k);
  // default: the blocked MFMA kernel (hfmi_chol.hip); the column-at-a-time kernels below stay for A/B runs
  // (HFMI_CHOL=tile, tuning "chol" = 1) and as the reference the tests compare the blocked kernel with
  if (g_chol_default < 0) {
    const char* e = getenv("HFMI_CHOL");
    g_chol_default = (e && !strcmp(e, "tile")) ? 1 : 0;
  }
  if (g_chol_default == 0) return launch_chol_mfma(ctx, k, slot_gram, slot_r, slot_rinv, slot_rtot, rtot_mode, full_r, shift_rel, pivot_tol);
  const int use_lds = (k <= 139) ? 1 : 0;   // (32 + 3 * 256) * 8 + 139 * 139 * 8 = 160,968 bytes <= 160 KB (163,840)
  const size_t shmem = (32 + 3 * 256) * sizeof(double) + (use_lds ? (size_t)k * (k | 1) * sizeof(double) : 0);
  if (pivot_tol <= 0.0) pivot_tol = 64.0 * k * EPS_D;
  static const bool chol_generic = env_flag("HFMI_CHOL_GENERIC");   // A/B: the generic one-kernel path
  if (use_lds && !chol_generic) {
    const int ept = (k * (k + 1) / 2 + CHOL_REG_THREADS - 1) / CHOL_REG_THREADS;   // <= 19 for k <= 139
    const size_t shm = (32 + 4 * 256) * sizeof(double) + (size_t)k * (k | 1) * sizeof(double);
#define CHOL_REG(E)                                                                                                   \
  do {                                                                                                                \
    HIP_TRY(hipFuncSetAttribute((const void*)k_chol_reg<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));   \
    hipLaunchKernelGGL(k_chol_reg<E>, dim3(1), dim3(CHOL_REG_THREADS), shm, ctx->stream, sm_ptr(ctx, slot_gram),      \
                       SM_LD, k, sm_ptr(ctx, slot_r), sm_ptr(ctx, slot_rinv), sm_ptr(ctx, slot_rtot),                 \
                       sm_ptr(ctx, SM_TMP2), SM_LD, rtot_mode, full_r, shift_rel, pivot_tol, sm_ptr(ctx, SM_AUX),     \
                       sm_ptr(ctx, SM_AUX) + SM_LD, ctx->status_dev);                                                 \
  } while (0)
    static int use_tile = -1;   // HFMI_CHOL_TILE=0: the cyclic ownership map (A/B measurements)
    if (use_tile < 0) {
      const char* e = getenv("HFMI_CHOL_TILE");
      use_tile = (e && atoi(e) == 0) ? 0 : 1;
    }
#define CHOL_TILE(A, B)                                                                                                 \
  do {                                                                                                                  \
    HIP_TRY(hipFuncSetAttribute((const void*)k_chol_tile<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
    hipLaunchKernelGGL((k_chol_tile<A, B>), dim3(1), dim3(CHOL_REG_THREADS), shm, ctx->stream, sm_ptr(ctx, slot_gram),  \
                       SM_LD, k, sm_ptr(ctx, slot_r), sm_ptr(ctx, slot_rinv), sm_ptr(ctx, slot_rtot),                   \
                       sm_ptr(ctx, SM_TMP2), SM_LD, rtot_mode, full_r, shift_rel, pivot_tol, sm_ptr(ctx, SM_AUX),       \
                       sm_ptr(ctx, SM_AUX) + SM_LD, ctx->status_dev, (double*)nullptr);                                 \
  } while (0)
    // smallest rectangle whose tiles fit the 512 threads
    if (use_tile && k >= 8) {
      if (chol_tiles(k, 1, 2) <= CHOL_REG_THREADS) CHOL_TILE(1, 2);
      else if (chol_tiles(k, 2, 2) <= CHOL_REG_THREADS) CHOL_TILE(2, 2);
      else if (chol_tiles(k, 2, 3) <= CHOL_REG_THREADS) CHOL_TILE(2, 3);
      else if (chol_tiles(k, 3, 3) <= CHOL_REG_THREADS) CHOL_TILE(3, 3);
      else if (chol_tiles(k, 3, 4) <= CHOL_REG_THREADS) CHOL_TILE(3, 4);
      else if (chol_tiles(k, 4, 4) <= CHOL_REG_THREADS) CHOL_TILE(4, 4);
      else if (chol_tiles(k, 4, 5) <= CHOL_REG_THREADS) CHOL_TILE(4, 5);
      else CHOL_TILE(5, 5);
    } else if (ept <= 1) CHOL_REG(1);
    else if (ept <= 2) CHOL_REG(2);
    else if (ept <= 4) CHOL_REG(4);
    else if (ept <= 6) CHOL_REG(6);
    else if (ept <= 8) CHOL_REG(8);
    else if (ept <= 12) CHOL_REG(12);
    else if (ept <= 16) CHOL_REG(16);
    else CHOL_REG(20);
#undef CHOL_REG
#undef CHOL_TILE
  } else if (!use_lds && !chol_generic) {
    // 140 <= k <= 256: tiles in the registers of 1024 threads, the factor's copy in the global slot SM_TMP
    const size_t shmb = (32 + 4 * 256) * sizeof(double);
#define CHOL_BIG(A, B)                                                                                                    \
  hipLaunchKernelGGL((k_chol_tile<A, B, 1024, true>), dim3(1), dim3(1024), shmb, ctx->stream, sm_ptr(ctx, slot_gram), SM_LD, k, \
                     sm_ptr(ctx, slot_r), sm_ptr(ctx, slot_rinv), sm_ptr(ctx, slot_rtot), sm_ptr(ctx, SM_TMP2), SM_LD,   \
                     rtot_mode, full_r, shift_rel, pivot_tol, sm_ptr(ctx, SM_AUX), sm_ptr(ctx, SM_AUX) + SM_LD,          \
                     ctx->status_dev, sm_ptr(ctx, SM_TMP))
    auto tiles1024 = [&](int a, int b) { return chol_tiles(k, a, b) <= 1024; };
    if (tiles1024(4, 4)) CHOL_BIG(4, 4);
    else if (tiles1024(4, 5)) CHOL_BIG(4, 5);
    else if (tiles1024(5, 5)) CHOL_BIG(5, 5);
    else if (tiles1024(5, 6)) CHOL_BIG(5, 6);
    else CHOL_BIG(6, 6);
#undef CHOL_BIG
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)k_chol_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(k_chol_inv, dim3(1), dim3(small_threads()), shmem, ctx->stream, sm_ptr(ctx, slot_gram), SM_LD, k,
                       sm_ptr(ctx, slot_r), sm_ptr(ctx, slot_rinv), sm_ptr(ctx, slot_rtot), sm_ptr(ctx, SM_TMP2), SM_LD,
                       rtot_mode, full_r, shift_rel, pivot_tol, sm_ptr(ctx, SM_TMP), use_lds, sm_ptr(ctx, SM_AUX),
                       sm_ptr(ctx, SM_AUX) + SM_LD, ctx->status_dev);
  }
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

__global__ void k_small_identity(double* M, int ld, int k) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < k * ld) M[e] = ((e / ld) == (e % ld)) ? 1.0 : 0.0;
}
// C (k x r, zero padded to a multiple of 16 columns) = A (k x k) * B[:, :r], all row-major small-arena matrices.  One wave per
// 16 x 16 tile of C on the fp64 MFMA, sixteen k-steps of loads in flight (the scalar version -- one thread per entry, k
// dependent multiply-adds behind two loads each -- took 37 us at k = 138 for 5 Mflop).
typedef double sm_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_small_matmul(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ Cm, int ld,
                                                     int k, int r, int rpad) {
  const int l = threadIdx.x, li = l & 15, lk = l >> 4;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const bool rowok = i0 + li < k, colok = j0 + li < r;
  sm_d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < k; k0 += 64) {
    double a[16], b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int kk = k0 + 4 * u + lk;
      a[u] = (rowok && kk < k) ? A[(i0 + li) * ld + kk] : 0.0;      // A operand: lane (li, lk) = A[row li][k-index lk]
      b[u] = (colok && kk < k) ? B[kk * ld + j0 + li] : 0.0;        // B operand: lane (li, lk) = B[k-index lk][column li]
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int i = i0 + 4 * s + lk, j = j0 + li;                     // accumulator: register s = row 4 s + lk, column li
    if (i < k && j < rpad) Cm[i * ld + j] = acc[s];
  }
}
int launch_small_matmul(hfmi_ctx* ctx, int k, int r, int slot_a, int slot_b, int slot_c) {
  const int rpad = (r + 15) & ~15;
  hipLaunchKernelGGL(k_small_matmul, dim3(rpad / 16, (k + 15) / 16), dim3(64), 0, ctx->stream, sm_ptr(ctx, slot_a), sm_ptr(ctx, slot_b),
                     sm_ptr(ctx, slot_c), SM_LD, k, r, rpad);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
int launch_small_set_identity(hfmi_ctx* ctx, int k, int slot) {
  hipLaunchKernelGGL(k_small_identity, dim3((k * SM_LD + 255) / 256), dim3(256), 0, ctx->stream, sm_ptr(ctx, slot), SM_LD, k);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// ------------------------------------------------------------------------------------------------
// Jacobi eigensolver
// ------------------------------------------------------------------------------------------------
// Round-robin tournament on n = 2 np players kept in SLOT order: the players of pair P always sit in the adjacent
// slots (2P, 2P+1) and the matrix is stored in slot order, S[s][t] = A[player(s)][player(t)].  After a round every
// player but the one in slot 0 moves one seat along the ring  1 -> 2 -> 4 -> ... -> n-2 -> n-1 -> n-3 -> ... -> 3 -> 1
// (Brent-Luk): n - 1 rounds meet every pair once and bring every player back to its own slot, so after whole
// sweeps slot order == player order.  The rotated 2x2 blocks are WRITTEN to their next-round seats, which makes
// every read of the update phase a fixed, aligned 16-byte pair (no index tables, no bank-conflict lottery).
__device__ __forceinline__ int jac_next_slot(int s, int n) {
  if (s == 0 || n == 2) return s;
  if (s & 1) return s == 1 ? 2 : s - 2;  // bottom row walks left; seat 1 climbs to the top row
  return s == n - 2 ? n - 1 : s + 2;     // top row walks right; the last seat drops to the bottom row
}

#define JAC_MAX_SWEEPS 40

// Rotation (c, s) that annihilates the pivot of [[app, apq], [apq, aqq]]:  t = tan(phi) = sign(theta) /
// (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), written with two reciprocal square roots instead of
// two divisions and two square roots (the dependent chain of this phase is pure latency on one CU):
// h = hypot(d, 2 apq), cos(2 phi) = |d| / h, c^2 = (1 + |d|/h) / 2, s = sign(d) apq / (h c).
__device__ __forceinline__ void jac_rotation(double app, double apq, double aqq, double& c, double& s) {
  c = 1.0;
  s = 0.0;
  const double g = 100.0 * fabs(apq);
  if (apq == 0.0 || (fabs(app) + g == fabs(app) && fabs(aqq) + g == fabs(aqq))) return;
  const double d = aqq - app;
  const double h2 = d * d + 4.0 * apq * apq;
  if (!(h2 > 1e-300) || !(h2 < 1e300)) {  // out of the safe range of the squared form: the classical formula
    const double theta = 0.5 * d / apq;
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    c = rsqrt(t * t + 1.0);
    s = t * c;
    return;
  }
  const double rh = fast_rsqrt(h2);
  const double c2 = 0.5 + 0.5 * fabs(d) * rh;
  const double rc = fast_rsqrt(c2);
  c = c2 * rc;
  s = (d >= 0.0 ? apq : -apq) * rh * rc;
}

// NB = 2x2 blocks per thread in the update phase (ceil(cells / threads)); the block coordinates of a thread never
// change, so all its addresses are computed once.
template <int NB, bool LDS>
__global__ __launch_bounds__(SMALL_THREADS) void k_jacobi_eig(const double* __restrict__ T, int ldt, int k,
                                                              double* __restrict__ gwork, int use_lds,
                                                              double2* __restrict__ rotlog, double* __restrict__ dvals,
                                                              int* __restrict__ perm, int sort_by_abs,
                                                              hfmi_status_words* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);                // 32
  double2* rot = reinterpret_cast<double2*>(red + 32);          // 128 (c, s) per pair
  double* Ad = reinterpret_cast<double*>(rot + 128);            // 256 diagonal copy / keys
  double* lds_a = Ad + 256;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int n = (k + 1) & ~1;  // players (one dummy if k is odd: a zero row/column that is never rotated)
  const int np = n / 2;
  // even leading dimension: the (2P, 2P+1) column pairs are 16-byte aligned.  LDS is a template parameter so that
  // the matrix pointer has ONE address space (a runtime LDS/global select would compile to FLAT accesses).
  int lda;
  double* A;
  if constexpr (LDS) {
    lda = n;
    A = lds_a;
  } else {
    lda = SM_LD;
    A = gwork;
  }
  const float inv_np1 = __frcp_rn((float)(np + 1));
  const int cells = ((np + 1) >> 1) * (np + 1);

  double fro = 0.0;
  for (int i = wave; i < n; i += nw)
    for (int j = lane; j < n; j += 64) {
      const double v = (i < k && j < k) ? 0.5 * (T[i * ldt + j] + T[j * ldt + i]) : 0.0;
      A[i * lda + j] = v;
      fro += v * v;
    }
  fro = block_sum(fro, red);
  __syncthreads();
  const double tol2 = EPS_D * EPS_D * fro;

  // this thread's blocks {P <= Q}: read offset of the block's first row and the next-round seat of each of its four
  // entries.  Only the block-upper triangle (pair(row) <= pair(col)) is stored and read: an entry whose new seat
  // falls below it is written transposed, so every entry costs ONE LDS store.
  int rd[NB], P_[NB], Q_[NB], d11[NB], d12[NB], d21[NB], d22[NB];
  bool live[NB];
  auto seat = [&](int r, int c) {
    const bool upper = (r >> 1) != (c >> 1) ? (r >> 1) < (c >> 1) : r <= c;
    return upper ? r * lda + c : c * lda + r;
  };
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) {
    const int e = tid + bi * nthr;
    int P = 0, Q = 0;
    live[bi] = e < cells && tri_cell(e, np, inv_np1, P, Q);
    P_[bi] = P;
    Q_[bi] = Q;
    rd[bi] = 2 * P * lda + 2 * Q;
    const int r1 = jac_next_slot(2 * P, n), r2 = jac_next_slot(2 * P + 1, n);
    const int c1 = jac_next_slot(2 * Q, n), c2 = jac_next_slot(2 * Q + 1, n);
    d11[bi] = seat(r1, c1);
    d12[bi] = seat(r1, c2);
    d21[bi] = seat(r2, c1);
    d22[bi] = seat(r2, c2);
  }

  const long long tj0 = clock64();
  int sweeps = 0;
  double off2 = 0.0;
  for (; sweeps < JAC_MAX_SWEEPS; ++sweeps) {
    off2 = 0.0;
    for (int i = wave; i < n; i += nw)
      for (int j = i + 1 + lane; j < n; j += 64) off2 += 2.0 * A[i * lda + j] * A[i * lda + j];
    off2 = block_sum(off2, red);
    __syncthreads();
    if (off2 <= tol2) break;
    for (int r = 0; r < n - 1; ++r) {
      // phase 1: one rotation per pair from the diagonal block in seats (2P, 2P+1)
      if (tid < np) {
        const double2 top = *reinterpret_cast<const double2*>(A + 2 * tid * lda + 2 * tid);
        const double aqq = A[(2 * tid + 1) * lda + 2 * tid + 1];
        double c, sn;
        jac_rotation(top.x, top.y, aqq, c, sn);
        rot[tid] = make_double2(c, sn);
        rotlog[((size_t)sweeps * (n - 1) + r) * np + tid] = make_double2(c, sn);
      }
      __syncthreads();
      // phase 2: every 2x2 block {P, Q}, P <= Q, <- J_P^T * block * J_Q, written to the seats its players take in
      // the next round (block-upper triangle only).  All reads precede all writes.
      double z11[NB], z12[NB], z21[NB], z22[NB];
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        if (!live[bi]) continue;
        const double2 x1 = *reinterpret_cast<const double2*>(A + rd[bi]);
        double2 x2 = *reinterpret_cast<const double2*>(A + rd[bi] + lda);
        if (P_[bi] == Q_[bi]) x2.x = x1.y;  // below the diagonal of a diagonal block: not stored
        {
          const double2 rp = rot[P_[bi]], rq = rot[Q_[bi]];
          // columns: (x_ip, x_iq) <- (c x_ip - s x_iq, s x_ip + c x_iq) with J_Q
          const double y11 = rq.x * x1.x - rq.y * x1.y, y12 = rq.y * x1.x + rq.x * x1.y;
          const double y21 = rq.x * x2.x - rq.y * x2.y, y22 = rq.y * x2.x + rq.x * x2.y;
          // rows with J_P
          z11[bi] = rp.x * y11 - rp.y * y21;
          z21[bi] = rp.y * y11 + rp.x * y21;
          z12[bi] = rp.x * y12 - rp.y * y22;
          z22[bi] = rp.y * y12 + rp.x * y22;
          if (P_[bi] == Q_[bi] && rp.y != 0.0) z12[bi] = 0.0;   // the annihilated pivot (an identity rotation keeps it)
        }
      }
      __syncthreads();
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        if (!live[bi]) continue;
        A[d11[bi]] = z11[bi];
        A[d22[bi]] = z22[bi];
        if (P_[bi] == Q_[bi]) {
          A[d12[bi]] = z12[bi];  // d12 == d21 here
        } else {
          A[d12[bi]] = z12[bi];
          A[d21[bi]] = z21[bi];
        }
      }
      __syncthreads();
    }
  }
  // whole sweeps bring every player back to its own slot: slot order == index order here.
  // eigenvalues, sort descending (rank by counting; ties broken by index -> a permutation)
  for (int i = tid; i < k; i += nthr) Ad[i] = A[i * lda + i];
  __syncthreads();
  for (int i = tid; i < k; i += nthr) {
    const double ki = sort_by_abs ? fabs(Ad[i]) : Ad[i];
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const double kj = sort_by_abs ? fabs(Ad[j]) : Ad[j];
      if (kj > ki || (kj == ki && j < i)) ++rank;
    }
    perm[rank] = i;
    dvals[rank] = Ad[i];
  }
  if (tid == 0) {
    status->offdiag = fro > 0.0 ? sqrt(off2 / fro) : 0.0;
    status->sweeps = sweeps;
    status->failed = (sweeps >= JAC_MAX_SWEEPS && off2 > tol2) ? 1 : 0;
    status->tick[0] = clock64() - tj0;
    status->tick[1] = 0;
    status->tick[2] = 0;
    status->tick[3] = sweeps;
    status->tick[4] = 1;
  }
}

// The 2x2 block update shared by the update phase and the look-ahead pivots of k_jacobi_eig_db (which must
// reproduce the stored values bit for bit): z = J_P^T [x1; x2] J_Q, rp / rq = (c, s) of pairs P / Q.
__device__ __forceinline__ void jac_block(double2 x1, double2 x2, const double2 rp, const double2 rq, const bool diag,
                                          double& z11, double& z12, double& z21, double& z22) {
  if (diag) x2.x = x1.y;  // below the diagonal of a diagonal block: not stored
  const double y11 = rq.x * x1.x - rq.y * x1.y, y12 = rq.y * x1.x + rq.x * x1.y;
  const double y21 = rq.x * x2.x - rq.y * x2.y, y22 = rq.y * x2.x + rq.x * x2.y;
  z11 = rp.x * y11 - rp.y * y21;
  z21 = rp.y * y11 + rp.x * y21;
  z12 = rp.x * y12 - rp.y * y22;
  z22 = rp.y * y12 + rp.x * y22;
  if (diag && rp.y != 0.0) z12 = 0.0;  // the annihilated pivot (an identity rotation keeps it)
}

// k_jacobi_eig_db: the same tournament with ONE barrier per round instead of three (n <= 138).
//  * The block-upper triangle is PACKED, block (P, Q), P <= Q, at index P np - P (P - 1) / 2 + Q - P, in two planes
//    of 16-byte elements (top row / bottom row of the 2x2 block), and there are TWO such images: a round reads one and
//    writes the players' next seats into the other, so no barrier separates its reads from its writes
//    (2 x 2 x 2415 x 16 B = 154.6 KB at n = 138 -- the full square would not fit twice).
//  * The rotations of round g + 1 are computed DURING round g by the last two waves, which own no blocks: pair P' of the
//    next round is made of two players that sit in different pairs now, its three entries are three values the update
//    phase is about to store, and the look-ahead threads recompute exactly those (jac_block on the three source blocks,
//    same operation order) from the old image and the current rotations.  The dependent chain LDS read -> rotation ->
//    barrier -> LDS read -> update -> LDS write -> barrier of the three-barrier kernel becomes max(update, look-ahead).
template <int NB>
__global__ __launch_bounds__(SMALL_THREADS) void k_jacobi_eig_db(const double* __restrict__ T, int ldt, int k,
                                                                 double2* __restrict__ rotlog, double* __restrict__ dvals,
                                                                 int* __restrict__ perm, int sort_by_abs,
                                                                 hfmi_status_words* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);                // 32
  double2* rot = reinterpret_cast<double2*>(red + 32);          // 2 x 128 (c, s) per pair: this round / next round
  double* Ad = reinterpret_cast<double*>(rot + 256);            // 256 diagonal copy / keys
  double* img = Ad + 256;                                       // [2 images][4 planes][nblk]
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int n = (k + 1) & ~1;  // players (one dummy if k is odd: a zero row/column that is never rotated)
  const int np = n / 2;
  const int nblk = np * (np + 1) / 2;
  const int nupd = nthr - 128;                                   // update threads; the last two waves look ahead
  const float inv_np1 = __frcp_rn((float)(np + 1));
  const int cells = ((np + 1) >> 1) * (np + 1);
  auto bidx = [&](int P, int Q) { return P * np - P * (P - 1) / 2 + (Q - P); };
  // double index (within one image) of entry (r, c) of the seat-ordered matrix; lower entries map to their mirror
  auto seat = [&](int r, int c) {
    const bool upper = (r >> 1) != (c >> 1) ? (r >> 1) < (c >> 1) : r <= c;
    const int rr = upper ? r : c, cc = upper ? c : r;
    return ((rr & 1) * 2 + (cc & 1)) * nblk + bidx(rr >> 1, cc >> 1);
  };

  // this thread's blocks: element index in a plane, and the next-round seats of its four entries
  int bx[NB], d11[NB], d12[NB], d21[NB], d22[NB], P_[NB], Q_[NB];
  bool live[NB];
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) {
    const int e = tid + bi * nupd;
    int P = 0, Q = 0;
    live[bi] = tid < nupd && e < cells && tri_cell(e, np, inv_np1, P, Q);
    P_[bi] = P;
    Q_[bi] = Q;
    bx[bi] = bidx(P, Q);
    const int r1 = jac_next_slot(2 * P, n), r2 = jac_next_slot(2 * P + 1, n);
    const int c1 = jac_next_slot(2 * Q, n), c2 = jac_next_slot(2 * Q + 1, n);
    d11[bi] = seat(r1, c1);
    d12[bi] = seat(r1, c2);
    d21[bi] = seat(r2, c1);
    d22[bi] = seat(r2, c2);
  }
  // look-ahead thread pi: the pair that will sit in seats (2 pi, 2 pi + 1) comes from the seats a, b of this round
  const int pi = tid - nupd;
  const bool ahead = pi >= 0 && pi < np;
  int pa = 0, pb = 0, ia = 0, ib = 0, bab = 0, baa = 0, bbb = 0;
  bool ab_swapped = false;
  if (ahead) {
    int a = 0, b = 0;
    for (int sl = 0; sl < n; ++sl) {
      const int nx = jac_next_slot(sl, n);
      if (nx == 2 * pi) a = sl;
      if (nx == 2 * pi + 1) b = sl;
    }
    pa = a >> 1; ia = a & 1; pb = b >> 1; ib = b & 1;
    ab_swapped = pb < pa;
    bab = ab_swapped ? bidx(pb, pa) : bidx(pa, pb);
    baa = bidx(pa, pa);
    bbb = bidx(pb, pb);
  }

  // load (symmetrised) into image 0, seat order == index order
  double fro = 0.0;
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) {
    if (!live[bi]) continue;
    double v[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = 2 * P_[bi] + i, c = 2 * Q_[bi] + j;
        v[i][j] = (r < k && c < k) ? 0.5 * (T[r * ldt + c] + T[c * ldt + r]) : 0.0;
      }
    img[bx[bi]] = v[0][0];
    img[nblk + bx[bi]] = v[0][1];
    img[2 * nblk + bx[bi]] = v[1][0];
    img[3 * nblk + bx[bi]] = v[1][1];
    if (P_[bi] == Q_[bi]) fro += v[0][0] * v[0][0] + v[1][1] * v[1][1] + 2.0 * v[0][1] * v[0][1];
    else fro += 2.0 * (v[0][0] * v[0][0] + v[0][1] * v[0][1] + v[1][0] * v[1][0] + v[1][1] * v[1][1]);
  }
  fro = block_sum(fro, red);
  __syncthreads();
  const double tol2 = EPS_D * EPS_D * fro;
  const long long max_rounds = (long long)JAC_MAX_SWEEPS * (n - 1);

  // rotations of the very first round, straight from the diagonal blocks
  if (tid < np) {
    const int bd = bidx(tid, tid);
    double c, sn;
    jac_rotation(img[bd], img[nblk + bd], img[3 * nblk + bd], c, sn);
    rot[tid] = make_double2(c, sn);
    rotlog[tid] = make_double2(c, sn);
  }
  __syncthreads();

  // the look-ahead chain is serial and decides the length of a round: its waves get issue priority over the update
  // waves they share a SIMD with (without it k = 74 ran 16 % slower than the three-barrier kernel)
  if (tid >= nupd) __builtin_amdgcn_s_setprio(3);
  const long long tj0 = clock64();
  int sweeps = 0, cur = 0;
  long long g = 0;  // global round counter
  double off2 = 0.0;
  for (; sweeps < JAC_MAX_SWEEPS; ++sweeps) {
    const double* A = img + (size_t)cur * 4 * nblk;
    off2 = 0.0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (!live[bi]) continue;
      const double2 x1 = make_double2(A[bx[bi]], A[nblk + bx[bi]]), x2 = make_double2(A[2 * nblk + bx[bi]], A[3 * nblk + bx[bi]]);
      if (P_[bi] == Q_[bi]) off2 += 2.0 * x1.y * x1.y;
      else off2 += 2.0 * (x1.x * x1.x + x1.y * x1.y + x2.x * x2.x + x2.y * x2.y);
    }
    off2 = block_sum(off2, red);
    __syncthreads();
    if (off2 <= tol2) break;
    for (int r = 0; r < n - 1; ++r, ++g) {
      const double* Ar = img + (size_t)cur * 4 * nblk;
      double* Aw = img + (size_t)(cur ^ 1) * 4 * nblk;
      auto blk = [&](int b, double2& x1, double2& x2) {
        x1 = make_double2(Ar[b], Ar[nblk + b]);
        x2 = make_double2(Ar[2 * nblk + b], Ar[3 * nblk + b]);
      };
      const double2* rt = rot + (g & 1) * 128;
      if (tid < nupd) {
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
          if (!live[bi]) continue;
          double2 x1, x2;
          blk(bx[bi], x1, x2);
          double z11, z12, z21, z22;
          jac_block(x1, x2, rt[P_[bi]], rt[Q_[bi]], P_[bi] == Q_[bi], z11, z12, z21, z22);
          Aw[d11[bi]] = z11;
          Aw[d22[bi]] = z22;
          Aw[d12[bi]] = z12;                          // d12 == d21 for a diagonal block
          if (P_[bi] != Q_[bi]) Aw[d21[bi]] = z21;
        }
      } else if (ahead) {
        double z11, z12, z21, z22;
        double2 x1, x2;
        const double2 ra = rt[pa], rb = rt[pb];
        blk(baa, x1, x2);
        jac_block(x1, x2, ra, ra, true, z11, z12, z21, z22);
        const double naa = ia ? z22 : z11;
        blk(bbb, x1, x2);
        jac_block(x1, x2, rb, rb, true, z11, z12, z21, z22);
        const double nbb = ib ? z22 : z11;
        double nab;
        blk(bab, x1, x2);
        if (pa == pb) {
          jac_block(x1, x2, ra, ra, true, z11, z12, z21, z22);
          nab = z12;
        } else if (!ab_swapped) {
          jac_block(x1, x2, ra, rb, false, z11, z12, z21, z22);
          nab = ia ? (ib ? z22 : z21) : (ib ? z12 : z11);
        } else {  // stored block is (pb, pa): rows belong to b's pair
          jac_block(x1, x2, rb, ra, false, z11, z12, z21, z22);
          nab = ib ? (ia ? z22 : z21) : (ia ? z12 : z11);
        }
        double c, sn;
        jac_rotation(naa, nab, nbb, c, sn);
        rot[((g + 1) & 1) * 128 + pi] = make_double2(c, sn);
        if (g + 1 < max_rounds) rotlog[(size_t)(g + 1) * np + pi] = make_double2(c, sn);
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  // whole sweeps bring every player back to its own slot: slot order == index order here.
  {
    const double* A = img + (size_t)cur * 4 * nblk;
    for (int i = tid; i < k; i += nthr) {
      const int P = i >> 1;
      Ad[i] = (i & 1) ? A[3 * nblk + bidx(P, P)] : A[bidx(P, P)];
    }
  }
  __syncthreads();
  for (int i = tid; i < k; i += nthr) {
    const double ki = sort_by_abs ? fabs(Ad[i]) : Ad[i];
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const double kj = sort_by_abs ? fabs(Ad[j]) : Ad[j];
      if (kj > ki || (kj == ki && j < i)) ++rank;
    }
    perm[rank] = i;
    dvals[rank] = Ad[i];
  }
  if (tid == 0) {
    status->offdiag = fro > 0.0 ? sqrt(off2 / fro) : 0.0;
    status->sweeps = sweeps;
    status->failed = (sweeps >= JAC_MAX_SWEEPS && off2 > tol2) ? 1 : 0;
    status->tick[0] = clock64() - tj0;
    status->tick[1] = 0;
    status->tick[2] = 0;
    status->tick[3] = sweeps;
    status->tick[4] = 2;
  }
}

// Row i of V = e_i^T * (product of all logged rotations); one workgroup per row.  The row is kept in slot order
// in two LDS buffers: lane P rotates the adjacent pair (2P, 2P+1) of the current buffer and writes it to the
// players' next seats in the other one -- one barrier per round, no index arithmetic.
__global__ __launch_bounds__(128) void k_jacobi_vectors(const double2* __restrict__ rotlog, int k,
                                                        const hfmi_status_words* __restrict__ status,
                                                        const int* __restrict__ perm, double* __restrict__ V, int ldv) {
  __shared__ __attribute__((aligned(16))) double row[2][SM_MAXK + 2];
  const int n = (k + 1) & ~1, np = n / 2;
  const int i = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < n; j += blockDim.x) row[0][j] = (j == i) ? 1.0 : 0.0;
  __syncthreads();
  const int rounds = status->sweeps * (n - 1);
  const int s1 = jac_next_slot(2 * tid, n), s2 = jac_next_slot(2 * tid + 1, n);
  int cur = 0;
  // the (c, s) pairs come from global memory and depend on nothing: keep PF rounds of them in flight in registers,
  // otherwise every round pays a full memory round trip between two barriers
  constexpr int PF = 8;
  const bool act = tid < np;
  const double2* lg = rotlog + (act ? tid : 0);
  double2 ring[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) ring[u] = (u < rounds) ? lg[(size_t)u * np] : make_double2(1.0, 0.0);
  for (int r0 = 0; r0 < rounds; r0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int rr = r0 + u;
      if (rr < rounds) {   // uniform
        const double2 cs = ring[u];
        const int nx = rr + PF;
        ring[u] = (nx < rounds) ? lg[(size_t)nx * np] : make_double2(1.0, 0.0);
        if (act) {
          const double2 v = *reinterpret_cast<const double2*>(&row[cur][2 * tid]);
          row[cur ^ 1][s1] = cs.x * v.x - cs.y * v.y;
          row[cur ^ 1][s2] = cs.y * v.x + cs.x * v.y;
        }
        __syncthreads();
        cur ^= 1;
      }
    }
  }
  for (int c = tid; c < k; c += blockDim.x) V[i * ldv + c] = row[cur][perm[c]];
  for (int c = k + tid; c < ((k + 15) & ~15); c += blockDim.x) V[i * ldv + c] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// k_jacobi_vectors_wave: the same replay with one WAVE per row (n <= 128 seats: lane P keeps the pair seated in
// (2P, 2P+1) in registers).  A round is a rotation in registers and three lane shifts (the top row of the tournament
// walks right, the bottom row left, the two end seats swap rows) -- no LDS, no barrier.
// whole-wave shifts by one lane as DPP moves (wave_shr:1 = 0x138, wave_shl:1 = 0x130 on GFX9): two v_mov_b32_dpp per
// double instead of two ds_bpermute round trips (with __shfl the replay was slower than the LDS version)
__device__ __forceinline__ double lane_shr1(double v) {   // lane P <- lane P - 1 (lane 0 keeps its own)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_shl1(double v) {   // lane P <- lane P + 1 (lane 63 keeps its own)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void k_jacobi_vectors_wave(const double2* __restrict__ rotlog, int k,
                                                             const hfmi_status_words* __restrict__ status,
                                                             const int* __restrict__ perm, double* __restrict__ V, int ldv) {
  __shared__ double rowbuf[4][130];
  const int n = (k + 1) & ~1, np = n / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;                     // row of V handled by this wave (wave-uniform)
  const int rounds = status->sweeps * (n - 1);
  const bool act = lane < np;
  double x = (2 * lane == i) ? 1.0 : 0.0, y = (2 * lane + 1 == i) ? 1.0 : 0.0;
  constexpr int PF = 32;   // a round is ~100 cycles here: 8 rounds ahead no longer cover the memory latency
  const double2* lg = rotlog + (act ? lane : 0);
  double2 ring[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) ring[u] = (u < rounds) ? lg[(size_t)u * np] : make_double2(1.0, 0.0);
  if (i < k) {
    for (int r0 = 0; r0 < rounds; r0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int rr = r0 + u;
        if (rr < rounds) {   // uniform
          const double2 cs = act ? ring[u] : make_double2(1.0, 0.0);
          const int nxt = rr + PF;
          ring[u] = (nxt < rounds) ? lg[(size_t)nxt * np] : make_double2(1.0, 0.0);
          const double nx = cs.x * x - cs.y * y, ny = cs.y * x + cs.x * y;
          if (n > 2) {
            const double xr = lane_shr1(nx), yr = lane_shr1(ny), yl = lane_shl1(ny);
            x = (lane == 0) ? nx : (lane == 1) ? yr : xr;    // seat 0 stays, seat 1 climbs to seat 2, 2P-2 -> 2P
            y = (lane == np - 1) ? nx : yl;                  // seat n-2 drops to n-1, 2P+3 -> 2P+1
          } else {
            x = nx;
            y = ny;
          }
        }
      }
    }
    if (act) {
      rowbuf[wave][2 * lane] = x;
      rowbuf[wave][2 * lane + 1] = y;
    }
  }
  __syncthreads();
  if (i < k) {
    for (int c = lane; c < k; c += 64) V[i * ldv + c] = rowbuf[wave][perm[c]];
    for (int c = k + lane; c < ((k + 15) & ~15); c += 64) V[i * ldv + c] = 0.0;
  }
}

// One-sided (Hestenes) Jacobi SVD of a small square matrix R = U diag(sigma) V^T: column pairs of W (= R, then
// R V) are rotated until mutually orthogonal; sigma_j = ||w_j||, U = W diag(1/sigma), V = product of the
// rotations (logged, replayed by k_jacobi_vectors).  Full relative accuracy for small singular values (unlike
// eig(R^T R)).  Used by accuracyEnhancedSVD (hippylib randomizedSVD; activeSubspaceProjector.py:813-834,1026).
// One 16-lane group per column pair, one workgroup barrier per round.
// ------------------------------------------------------------------------------------------------
template <bool LDS>
__global__ __launch_bounds__(SMALL_THREADS) void k_jacobi_svd(const double* __restrict__ R, int ldr, int k,
                                                              double* __restrict__ gwork, int use_lds,
                                                              double2* __restrict__ rotlog, double* __restrict__ svals,
                                                              int* __restrict__ perm, double* __restrict__ Uleft, int ldu,
                                                              hfmi_status_words* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);   // 32
  double* sig = red + 32;                          // 256
  short* who = reinterpret_cast<short*>(sig + 256);  // [2][264]: player seated in each slot (jac_next_slot schedule)
  double* lds_w = sig + 256 + 132;
  int ldw;
  double* W;                                       // column-major: W[col * ldw + row]; one address space (see k_chol_inv)
  if constexpr (LDS) {
    ldw = k | 1;
    W = lds_w;
  } else {
    ldw = SM_LD;
    W = gwork;
  }
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int grp = tid >> 4, gl = tid & 15, ngrp = nthr >> 4;
  const int n = (k + 1) & ~1, np = n / 2;
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) W[j * ldw + i] = R[i * ldr + j];
  for (int j = tid; j < n; j += nthr) who[j] = (short)j;
  __syncthreads();
  int sweeps = 0, cur = 0;
  double worst = 0.0;
  for (; sweeps < JAC_MAX_SWEEPS; ++sweeps) {
    double mx = 0.0;
    for (int r = 0; r < n - 1; ++r) {
      const short* wc = who + cur * 264;
      for (int P = grp; P < np; P += ngrp) {
        // the pair seated in slots (2P, 2P+1); same schedule and orientation as k_jacobi_vectors replays
        const int a = wc[2 * P], b = wc[2 * P + 1];
        double c = 1.0, sn = 0.0;
        if (a < k && b < k) {
          double* wa = W + a * ldw;
          double* wb = W + b * ldw;
          double al = 0.0, be = 0.0, ga = 0.0;
          for (int i = gl; i < k; i += 16) {
            const double x = wa[i], y = wb[i];
            al += x * x;
            be += y * y;
            ga += x * y;
          }
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) {
            al += __shfl_xor(al, off, 64);
            be += __shfl_xor(be, off, 64);
            ga += __shfl_xor(ga, off, 64);
          }
          const double lim = sqrt(al * be);
          const double ratio = lim > 0.0 ? fabs(ga) / lim : 0.0;
          mx = fmax(mx, ratio);
          if (ratio > EPS_D) {
            const double zeta = 0.5 * (be - al) / ga;
            const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            c = rsqrt(1.0 + t * t);
            sn = c * t;
            for (int i = gl; i < k; i += 16) {
              const double x = wa[i], y = wb[i];
              wa[i] = c * x - sn * y;
              wb[i] = sn * x + c * y;
            }
          }
        }
        if (gl == 0) rotlog[((size_t)sweeps * (n - 1) + r) * np + P] = make_double2(c, sn);
      }
      for (int j = tid; j < n; j += nthr) who[(cur ^ 1) * 264 + jac_next_slot(j, n)] = wc[j];
      cur ^= 1;
      __syncthreads();
    }
    // convergence: every pair of the sweep was already orthogonal to round-off
    mx = -block_min(-mx, red);
    worst = mx;
    __syncthreads();
    if (mx <= 4.0 * EPS_D) {
      ++sweeps;
      break;
    }
  }
  // singular values, sort descending
  for (int j = wave; j < k; j += nw) {
    double a2 = 0.0;
    for (int i = lane; i < k; i += 64) a2 += W[j * ldw + i] * W[j * ldw + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a2 += __shfl_down(a2, off, 64);
    if (lane == 0) sig[j] = sqrt(a2);
  }
  __syncthreads();
  for (int i = tid; i < k; i += nthr) {
    const double ki = sig[i];
    int rank = 0;
    for (int j = 0; j < k; ++j)
      if (sig[j] > ki || (sig[j] == ki && j < i)) ++rank;
    perm[rank] = i;
    svals[rank] = ki;
  }
  __syncthreads();
  // left singular vectors U[:, c] = w_perm[c] / sigma  (row-major k x k output)
  for (int cidx = wave; cidx < k; cidx += nw) {
    const int j = perm[cidx];
    const double inv = sig[j] > 0.0 ? 1.0 / sig[j] : 0.0;
    for (int i = lane; i < k; i += 64) Uleft[i * ldu + cidx] = W[j * ldw + i] * inv;
  }
  if (tid == 0) {
    status->offdiag = worst;
    status->sweeps = sweeps > JAC_MAX_SWEEPS ? JAC_MAX_SWEEPS : sweeps;
    status->failed = (worst > 1e-12) ? 1 : 0;
    status->tick[4] = 2;
  }
}

int launch_jacobi_svd(hfmi_ctx* ctx, int k, int slot_r, int slot_u, int slot_v, double* svals) {
  if (k < 1 || k > SM_MAXK) HFMI_FAIL(HFMI_ERR_INVALID, "jacobi_svd: k=%d out of range", k);
  const int use_lds = (k <= 139) ? 1 : 0;   // 139*139*8 + 3.4 KB of tables <= 160 KB
  const int n = (k + 1) & ~1;
  const size_t log_bytes = (size_t)(JAC_MAX_SWEEPS + 1) * (n - 1) * (n / 2) * sizeof(double2) + 1024 * sizeof(int);
  void* logv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_MISC, log_bytes, &logv));
  double2* rotlog = (double2*)logv;
  int* perm = (int*)((char*)logv + (size_t)(JAC_MAX_SWEEPS + 1) * (n - 1) * (n / 2) * sizeof(double2));
  const size_t shmem = (32 + 256 + 132) * sizeof(double) + (use_lds ? (size_t)k * (k | 1) * sizeof(double) : 0);
  if (use_lds) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_svd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(k_jacobi_svd<true>, dim3(1), dim3(small_threads()), shmem, ctx->stream, sm_ptr(ctx, slot_r), SM_LD,
                       k, sm_ptr(ctx, SM_TMP), use_lds, rotlog, svals, perm, sm_ptr(ctx, slot_u), SM_LD, ctx->status_dev);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_svd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(k_jacobi_svd<false>, dim3(1), dim3(small_threads()), shmem, ctx->stream, sm_ptr(ctx, slot_r), SM_LD,
                       k, sm_ptr(ctx, SM_TMP), use_lds, rotlog, svals, perm, sm_ptr(ctx, slot_u), SM_LD, ctx->status_dev);
  }
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_jacobi_vectors, dim3(k), dim3(128), 0, ctx->stream, rotlog, k, ctx->status_dev, perm,
                     sm_ptr(ctx, slot_v), SM_LD);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

int launch_jacobi_eig(hfmi_ctx* ctx, int k, int slot_t, int slot_v, double* dvals, int sort_by_abs) {
  if (k < 1 || k > SM_MAXK) HFMI_FAIL(HFMI_ERR_INVALID, "jacobi_eig: k=%d out of range", k);
  const int n = (k + 1) & ~1, np = n / 2;
  const int use_lds = (n <= 138) ? 1 : 0;   // 138*138*8 + 6.4 KB of tables = 158.8 KB <= 160 KB
  const size_t log_bytes = (size_t)JAC_MAX_SWEEPS * (n - 1) * (n / 2) * sizeof(double2) + 1024 * sizeof(int);
  void* logv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_MISC, log_bytes, &logv));
  double2* rotlog = (double2*)logv;
  int* perm = (int*)((char*)logv + (size_t)JAC_MAX_SWEEPS * (n - 1) * (n / 2) * sizeof(double2));
  const size_t shmem = (32 + 256 + 256) * sizeof(double) + (use_lds ? (size_t)n * n * sizeof(double) : 0);
  const int threads = small_threads();
  const int cells = ((np + 1) >> 1) * (np + 1);
  const int nb = (cells + threads - 1) / threads;
#define JAC_LAUNCH2(NBV, LDSV)                                                                                       \
  do {                                                                                                               \
    HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_eig<NBV, LDSV>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                (int)shmem));                                                                        \
    hipLaunchKernelGGL((k_jacobi_eig<NBV, LDSV>), dim3(1), dim3(threads), shmem, ctx->stream, sm_ptr(ctx, slot_t),   \
                       SM_LD, k, sm_ptr(ctx, SM_TMP), use_lds, rotlog, dvals, perm, sort_by_abs, ctx->status_dev);   \
  } while (0)
#define JAC_LAUNCH(NBV)  \
  do {                   \
    if (use_lds) JAC_LAUNCH2(NBV, true); \
    else JAC_LAUNCH2(NBV, false);        \
  } while (0)
  static int use_db = -1;   // HFMI_JACOBI_DB: A/B measurements
  if (use_db < 0) {
    const char* e = getenv("HFMI_JACOBI_DB");
    use_db = e ? atoi(e) : 1;   // 0: never, 1: where it wins (>= 3 blocks per thread), 2: whenever it fits
  }
  const int nblk = np * (np + 1) / 2;
  const size_t shmem_db = (32 + 512 + 256) * sizeof(double) + (size_t)8 * nblk * sizeof(double);
  const int nb_db = threads > 128 ? (cells + (threads - 128) - 1) / (threads - 128) : 99;
  // measured (scripts/jacobi_ab.py, kernel traces): k = 138 1.21 vs 1.38 ms, but k = 74 0.41 vs 0.35 and k = 84 0.46 vs
  // 0.44 ms -- with one block per thread the look-ahead chain is longer than the three short phases it replaces
  const int db_min_blocks = (use_db == 2) ? 1 : 3;
  if (use_db && nb >= db_min_blocks && n <= 138 && np <= 128 && threads >= 256 && nb_db <= 3 && shmem_db <= 160 * 1024) {
#define JAC_DB(NBV)                                                                                                  \
  do {                                                                                                               \
    HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_eig_db<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                (int)shmem_db));                                                                     \
    hipLaunchKernelGGL((k_jacobi_eig_db<NBV>), dim3(1), dim3(threads), shmem_db, ctx->stream, sm_ptr(ctx, slot_t),   \
                       SM_LD, k, rotlog, dvals, perm, sort_by_abs, ctx->status_dev);                                 \
  } while (0)
    if (nb_db <= 1) JAC_DB(1);
    else if (nb_db <= 2) JAC_DB(2);
    else JAC_DB(3);
#undef JAC_DB
  } else if (nb <= 1) JAC_LAUNCH(1);
  else if (nb <= 2) JAC_LAUNCH(2);
  else if (nb <= 3) JAC_LAUNCH(3);
  else if (nb <= 9) JAC_LAUNCH(9);
  else HFMI_FAIL(HFMI_ERR_INVALID, "jacobi_eig: %d blocks per thread (k=%d, %d threads) not instantiated", nb, k, threads);
#undef JAC_LAUNCH
#undef JAC_LAUNCH2
  HIP_TRY(hipGetLastError());
  static const bool replay_lds = env_flag("HFMI_JACOBI_REPLAY_LDS");   // A/B: the LDS + barrier replay
  if (n <= 128 && !replay_lds)
    hipLaunchKernelGGL(k_jacobi_vectors_wave, dim3((k + 3) / 4), dim3(256), 0, ctx->stream, rotlog, k, ctx->status_dev, perm,
                       sm_ptr(ctx, slot_v), SM_LD);
  else
    hipLaunchKernelGGL(k_jacobi_vectors, dim3(k), dim3(128), 0, ctx->stream, rotlog, k, ctx->status_dev, perm,
                       sm_ptr(ctx, slot_v), SM_LD);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// Rayleigh-Ritz eigensolver selection.  Default: divide and conquer (hfmi_eig_dc.hip; k = 74: 0.68 -> about 0.15 ms).  The
// Jacobi kernels above stay for callers that want the high RELATIVE accuracy of small eigenvalues of graded positive
// definite matrices (hfmi_sym_eig_small flag bit 1, hfmi_double_pass flag bit 3) and for A/B runs (tuning "eig" = 1).
static int g_eig_default = -1;
int eig_tuning_set(const char* key, int value) {
  if (key && !strcmp(key, "eig") && (value == 0 || value == 1)) {
    g_eig_default = value;
    return 1;
  }
  return 0;
}
int launch_sym_eig(hfmi_ctx* ctx, int k, int slot_t, int slot_v, double* dvals, int sort_by_abs, int method) {
  if (g_eig_default < 0) {
    const char* e = getenv("HFMI_EIG");
    g_eig_default = (e && !strcmp(e, "jacobi")) ? 1 : 0;
  }
  if (method == 1 || g_eig_default == 1) return launch_jacobi_eig(ctx, k, slot_t, slot_v, dvals, sort_by_abs);
  return launch_dc_eig(ctx, k, slot_t, slot_v, dvals, sort_by_abs);
}
