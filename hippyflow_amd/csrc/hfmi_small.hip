// Small dense kernels of the Rayleigh-Ritz / Cholesky-QR steps: everything k x k (k <= 256) is done by
// ONE workgroup with the matrix resident in LDS (k <= 140: k*(k+1)*8 bytes <= 160 KB), fp64 throughout.
//
//   k_chol_inv   : G = R^T R (upper), optional diagonal shift on breakdown, R^{-1}, running R product
//   k_jacobi_eig : parallel cyclic Jacobi (round-robin ordering, one 2x2-block rotation phase per round),
//                  rotations are logged and replayed on the identity by k_jacobi_vectors (one workgroup
//                  per eigenvector-matrix row) so the eigenvector update never sits on the critical path.
#include "hfmi_internal.h"

#include <stdlib.h>
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
// index sets onto ALL lanes (folded triangle, reciprocal-multiply index split) and computes the inverse with
// no workgroup barriers at all.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);      // ~2^-26 relative
  y = y * (1.5 - 0.5 * x * y * y);         // two Newton steps -> ~1 ulp
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
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
  double* lds_m = invd + 256;
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
    __syncthreads();
    double ratio_local = 1e300;
    for (int j = 0; j < k; ++j) {
      const double piv = M[j * ldm + j];
      const double ref = diag0[j] + shift;
      if (!(piv > pivot_tol * ref) || !(ref > 0.0)) {  // uniform decision: every thread reads the same words
        if (tid == 0) s_break = 1;
        break;
      }
      if (tid == 0) ratio_local = fmin(ratio_local, piv / ref);
      const double inv = fast_rsqrt(piv);
      const double rjj = piv * inv;
      __syncthreads();
      for (int c = j + tid; c < k; c += nthr) M[j * ldm + c] = (c == j) ? rjj : M[j * ldm + c] * inv;
      if (tid == 0) invd[j] = inv;
      __syncthreads();
      // trailing update of the upper triangle: M[i][c] -= R[j][i] R[j][c], j < i <= c  (folded onto all lanes)
      const int n = k - j - 1;
      if (n > 0) {
        const float inv_np1 = __frcp_rn((float)(n + 1));
        const int cells = ((n + 1) >> 1) * (n + 1);
        const double* rj = M + j * ldm + j + 1;
        for (int e = tid; e < cells; e += nthr) {
          int a, b;
          if (tri_cell(e, n, inv_np1, a, b)) M[(j + 1 + a) * ldm + j + 1 + b] -= rj[a] * rj[b];
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
  tk2 = clock64();
  // R out (upper triangle, zeros below)
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) Rout[i * ldo + j] = (j >= i) ? M[i * ldm + j] : 0.0;
  // Inverse X = R^-1 by back substitution, one column per 16-lane group, NO workgroup barriers: column c is
  // x_c = 1/R_cc, x_i = -(sum_{l=i+1..c} R[i][l] x_l) / R_ii for i < c.  x is written into the (unused)
  // strictly lower triangle, X[i][c] -> M[c][i], so columns never collide with R or with each other.
  {
    const int grp = lane >> 4, gl = lane & 15;
    for (int c0 = 4 * wave; c0 < k; c0 += 4 * nw) {
      const int c = c0 + grp;
      const bool live = c < k;
      const double xc = live ? invd[c] : 0.0;
      const int cmax = (c0 + 3 < k) ? c0 + 3 : k - 1;
      for (int i = cmax - 1; i >= 0; --i) {
        double sacc = 0.0;
        if (live && i < c) {
          const double* ri = M + i * ldm;
          const double* xr = M + c * ldm;
          for (int l = i + 1 + gl; l <= c; l += 16) sacc += ri[l] * (l == c ? xc : xr[l]);
        }
        sacc += __shfl_xor(sacc, 8, 64);
        sacc += __shfl_xor(sacc, 4, 64);
        sacc += __shfl_xor(sacc, 2, 64);
        sacc += __shfl_xor(sacc, 1, 64);
        if (live && i < c && gl == 0) M[c * ldm + i] = -sacc * invd[i];
      }
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

int launch_chol_inv(hfmi_ctx* ctx, int k, int slot_gram, int slot_r, int slot_rinv, int slot_rtot, int rtot_mode,
                    int full_r, double shift_rel, double pivot_tol) {
  if (k < 1 || k > SM_MAXK) HFMI_FAIL(HFMI_ERR_INVALID, "chol_inv: k=%d out of range", k);
  const int use_lds = (k <= 139) ? 1 : 0;   // (32 + 256 + 256) * 8 + 139 * 139 * 8 = 158,920 bytes <= 160 KB
  const size_t shmem = (32 + 256 + 256) * sizeof(double) + (use_lds ? (size_t)k * (k | 1) * sizeof(double) : 0);
  HIP_TRY(hipFuncSetAttribute((const void*)k_chol_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  if (pivot_tol <= 0.0) pivot_tol = 64.0 * k * EPS_D;
  hipLaunchKernelGGL(k_chol_inv, dim3(1), dim3(small_threads()), shmem, ctx->stream, sm_ptr(ctx, slot_gram), SM_LD, k,
                     sm_ptr(ctx, slot_r), sm_ptr(ctx, slot_rinv), sm_ptr(ctx, slot_rtot), sm_ptr(ctx, SM_TMP2), SM_LD,
                     rtot_mode, full_r, shift_rel, pivot_tol, sm_ptr(ctx, SM_TMP), use_lds, sm_ptr(ctx, SM_AUX),
                     sm_ptr(ctx, SM_AUX) + SM_LD, ctx->status_dev);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

__global__ void k_small_identity(double* M, int ld, int k) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < k * ld) M[e] = ((e / ld) == (e % ld)) ? 1.0 : 0.0;
}
int launch_small_set_identity(hfmi_ctx* ctx, int k, int slot) {
  hipLaunchKernelGGL(k_small_identity, dim3((k * SM_LD + 255) / 256), dim3(256), 0, ctx->stream, sm_ptr(ctx, slot), SM_LD, k);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// ------------------------------------------------------------------------------------------------
// Jacobi eigensolver
// ------------------------------------------------------------------------------------------------
// round-robin tournament on n (even) players: round r in [0, n-1), pair p in [0, n/2)
__device__ __forceinline__ void rr_pair(int n, int r, int p, int& a, int& b) {
  if (p == 0) {
    a = n - 1;
    b = r;
  } else {
    a = (r + p) % (n - 1);
    b = (r - p + (n - 1)) % (n - 1);
  }
  if (a > b) {
    const int t = a;
    a = b;
    b = t;
  }
}

#define JAC_MAX_SWEEPS 40

__global__ __launch_bounds__(SMALL_THREADS) void k_jacobi_eig(const double* __restrict__ T, int ldt, int k,
                                                              double* __restrict__ gwork, int use_lds,
                                                              double2* __restrict__ rotlog, double* __restrict__ dvals,
                                                              int* __restrict__ perm, int sort_by_abs,
                                                              hfmi_status_words* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);                // 32
  double2* rot = reinterpret_cast<double2*>(red + 32);          // 128 (c, s) per pair
  double* Ad = reinterpret_cast<double*>(rot + 128);            // 256 diagonal copy / keys
  short* pa = reinterpret_cast<short*>(Ad + 256);               // 128 + 128 pair members of the current round
  short* pb = pa + 128;
  double* lds_a = reinterpret_cast<double*>(pb + 128);
  const int lda = use_lds ? (k | 1) : SM_LD;
  double* A = use_lds ? lds_a : gwork;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int n = (k + 1) & ~1;  // players (one dummy if k is odd)
  const int np = n / 2;
  const float inv_np1 = __frcp_rn((float)(np + 1));
  const int cells = ((np + 1) >> 1) * (np + 1);

  double fro = 0.0;
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) {
      const double v = 0.5 * (T[i * ldt + j] + T[j * ldt + i]);
      A[i * lda + j] = v;
      fro += v * v;
    }
  fro = block_sum(fro, red);
  __syncthreads();
  const double tol2 = EPS_D * EPS_D * fro;

  long long tj0 = clock64(), tp1 = 0, tp2 = 0;
  int sweeps = 0;
  double off2 = 0.0;
  for (; sweeps < JAC_MAX_SWEEPS; ++sweeps) {
    off2 = 0.0;
    for (int i = wave; i < k; i += nw)
      for (int j = i + 1 + lane; j < k; j += 64) off2 += 2.0 * A[i * lda + j] * A[i * lda + j];
    off2 = block_sum(off2, red);
    __syncthreads();
    if (off2 <= tol2) break;
    for (int r = 0; r < n - 1; ++r) {
      // phase 1: the round's pairs (kept in LDS: no modulo arithmetic in the update phase) and one
      // rotation per pair
      const long long ta = clock64();
      if (tid < np) {
        int a, b;
        rr_pair(n, r, tid, a, b);
        pa[tid] = (short)a;
        pb[tid] = (short)b;
        double c = 1.0, s = 0.0;
        if (b < k) {
          const double apq = A[a * lda + b];
          const double app = A[a * lda + a], aqq = A[b * lda + b];
          const double g = 100.0 * fabs(apq);
          if (apq != 0.0 && !(fabs(app) + g == fabs(app) && fabs(aqq) + g == fabs(aqq))) {
            const double theta = 0.5 * (aqq - app) / apq;
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = rsqrt(t * t + 1.0);
            s = t * c;
          }
        }
        rot[tid] = make_double2(c, s);
        rotlog[((size_t)sweeps * (n - 1) + r) * np + tid] = make_double2(c, s);
      }
      __syncthreads();
      const long long tb = clock64();
      tp1 += tb - ta;
      // phase 2: every 2x2 block {P, Q}, P <= Q, <- J_P^T * block * J_Q, mirrored into the lower triangle (full
      // symmetric storage: no min/max index logic).  The np (np + 1) / 2 blocks are laid onto ALL lanes through the
      // folded-triangle map -- this phase is fp64-VALU-issue bound, so idle lanes are the main cost.
      for (int e = tid; e < cells; e += nthr) {
        int P, Q;
        if (!tri_cell(e, np, inv_np1, P, Q)) continue;
        const int p1 = pa[P], q1 = pb[P], p2 = pa[Q], q2 = pb[Q];
        const int p1r = __mul24(p1, lda), q1r = __mul24(q1, lda), p2r = __mul24(p2, lda), q2r = __mul24(q2, lda);
        // a dummy player (k odd) has index >= k: its entries read as 0 and are never written; the rotation of
        // its pair is the identity, so the real member still receives the other pairs' rotations
        const bool r2 = q1 < k, c2 = q2 < k;
        const double2 rp = rot[P], rq = rot[Q];
        const double x11 = A[p1r + p2];
        const double x12 = c2 ? A[p1r + q2] : 0.0;
        const double x21 = r2 ? A[q1r + p2] : 0.0;
        const double x22 = (r2 && c2) ? A[q1r + q2] : 0.0;
        // columns: (x_ip, x_iq) <- (c x_ip - s x_iq, s x_ip + c x_iq) with J_Q
        const double y11 = rq.x * x11 - rq.y * x12, y12 = rq.y * x11 + rq.x * x12;
        const double y21 = rq.x * x21 - rq.y * x22, y22 = rq.y * x21 + rq.x * x22;
        // rows with J_P
        const double z11 = rp.x * y11 - rp.y * y21, z21 = rp.y * y11 + rp.x * y21;
        const double z12 = rp.x * y12 - rp.y * y22, z22 = rp.y * y12 + rp.x * y22;
        if (P == Q) {
          A[p1r + p1] = z11;
          if (r2) {
            A[q1r + q1] = z22;
            A[p1r + q1] = 0.0;        // the annihilated pivot
            A[q1r + p1] = 0.0;
          }
        } else {
          A[p1r + p2] = z11;
          A[p2r + p1] = z11;
          if (c2) {
            A[p1r + q2] = z12;
            A[q2r + p1] = z12;
          }
          if (r2) {
            A[q1r + p2] = z21;
            A[p2r + q1] = z21;
          }
          if (r2 && c2) {
            A[q1r + q2] = z22;
            A[q2r + q1] = z22;
          }
        }
      }
      __syncthreads();
      tp2 += clock64() - tb;
    }
  }
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
    status->tick[1] = tp1;
    status->tick[2] = tp2;
    status->tick[3] = sweeps;
    status->tick[4] = 1;
  }
}

// Row i of V = e_i^T * (product of all logged rotations); one workgroup per row, the row lives in LDS.
// The pair (a, b) of lane p in round r is ((r + p) mod (n-1), (r - p) mod (n-1)) -- advanced by +1 per round
// with a wrap test instead of a modulo; lane 0 pairs the fixed player n-1 with r.
__global__ __launch_bounds__(128) void k_jacobi_vectors(const double2* __restrict__ rotlog, int k,
                                                        const hfmi_status_words* __restrict__ status,
                                                        const int* __restrict__ perm, double* __restrict__ V, int ldv) {
  __shared__ double row[SM_MAXK + 2];
  const int n = (k + 1) & ~1, np = n / 2;
  const int i = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < n; j += blockDim.x) row[j] = (j == i) ? 1.0 : 0.0;
  __syncthreads();
  const int rounds = status->sweeps * (n - 1);
  int a = 0, b = 0;
  if (tid < np) rr_pair(n, 0, tid, a, b);
  int ua = (tid == 0) ? 0 : tid % (n - 1);                      // un-ordered members for the increment
  int ub = (tid == 0) ? 0 : (n - 1 - tid) % (n - 1);
  int r = 0;
  for (int rr = 0; rr < rounds; ++rr) {
    if (tid < np) {
      const double2 cs = rotlog[(size_t)rr * np + tid];
      const double va = row[a], vb = row[b];
      row[a] = cs.x * va - cs.y * vb;
      row[b] = cs.y * va + cs.x * vb;
      // next round's pair
      r = (r + 1 == n - 1) ? 0 : r + 1;
      if (tid == 0) {
        a = r;
        b = n - 1;
      } else {
        ua = (ua + 1 == n - 1) ? 0 : ua + 1;
        ub = (ub + 1 == n - 1) ? 0 : ub + 1;
        a = ua < ub ? ua : ub;
        b = ua < ub ? ub : ua;
      }
    }
    __syncthreads();
  }
  for (int c = tid; c < k; c += blockDim.x) V[i * ldv + c] = row[perm[c]];
  for (int c = k + tid; c < ((k + 15) & ~15); c += blockDim.x) V[i * ldv + c] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// One-sided (Hestenes) Jacobi SVD of a small square matrix R = U diag(sigma) V^T: column pairs of W (= R, then
// R V) are rotated until mutually orthogonal; sigma_j = ||w_j||, U = W diag(1/sigma), V = product of the
// rotations (logged, replayed by k_jacobi_vectors).  Full relative accuracy for small singular values (unlike
// eig(R^T R)).  Used by accuracyEnhancedSVD (hippylib randomizedSVD; activeSubspaceProjector.py:813-834,1026).
// One 16-lane group per column pair, one workgroup barrier per round.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SMALL_THREADS) void k_jacobi_svd(const double* __restrict__ R, int ldr, int k,
                                                              double* __restrict__ gwork, int use_lds,
                                                              double2* __restrict__ rotlog, double* __restrict__ svals,
                                                              int* __restrict__ perm, double* __restrict__ Uleft, int ldu,
                                                              hfmi_status_words* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* red = reinterpret_cast<double*>(smem);   // 32
  double* sig = red + 32;                          // 256
  double* lds_w = sig + 256;
  const int ldw = use_lds ? (k | 1) : SM_LD;
  double* W = use_lds ? lds_w : gwork;             // column-major: W[col * ldw + row]
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int grp = tid >> 4, gl = tid & 15, ngrp = nthr >> 4;
  const int n = (k + 1) & ~1, np = n / 2;
  for (int i = wave; i < k; i += nw)
    for (int j = lane; j < k; j += 64) W[j * ldw + i] = R[i * ldr + j];
  __syncthreads();
  int sweeps = 0;
  double worst = 0.0;
  for (; sweeps < JAC_MAX_SWEEPS; ++sweeps) {
    double mx = 0.0;
    for (int r = 0; r < n - 1; ++r) {
      for (int P = grp; P < np; P += ngrp) {
        int a, b;
        rr_pair(n, r, P, a, b);
        double c = 1.0, sn = 0.0;
        if (b < k) {
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
  const int use_lds = (k <= 140) ? 1 : 0;
  const int n = (k + 1) & ~1;
  const size_t log_bytes = (size_t)(JAC_MAX_SWEEPS + 1) * (n - 1) * (n / 2) * sizeof(double2) + 1024 * sizeof(int);
  void* logv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_MISC, log_bytes, &logv));
  double2* rotlog = (double2*)logv;
  int* perm = (int*)((char*)logv + (size_t)(JAC_MAX_SWEEPS + 1) * (n - 1) * (n / 2) * sizeof(double2));
  const size_t shmem = (32 + 256) * sizeof(double) + (use_lds ? (size_t)k * (k | 1) * sizeof(double) : 0);
  HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_svd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(k_jacobi_svd, dim3(1), dim3(small_threads()), shmem, ctx->stream, sm_ptr(ctx, slot_r), SM_LD, k,
                     sm_ptr(ctx, SM_TMP), use_lds, rotlog, svals, perm, sm_ptr(ctx, slot_u), SM_LD, ctx->status_dev);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_jacobi_vectors, dim3(k), dim3(128), 0, ctx->stream, rotlog, k, ctx->status_dev, perm,
                     sm_ptr(ctx, slot_v), SM_LD);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

int launch_jacobi_eig(hfmi_ctx* ctx, int k, int slot_t, int slot_v, double* dvals, int sort_by_abs) {
  if (k < 1 || k > SM_MAXK) HFMI_FAIL(HFMI_ERR_INVALID, "jacobi_eig: k=%d out of range", k);
  const int use_lds = (k <= 138) ? 1 : 0;   // 138*139*8 + 4.9 KB of tables = 158.3 KB <= 160 KB
  const int n = (k + 1) & ~1;
  const size_t log_bytes = (size_t)JAC_MAX_SWEEPS * (n - 1) * (n / 2) * sizeof(double2) + 1024 * sizeof(int);
  void* logv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_MISC, log_bytes, &logv));
  double2* rotlog = (double2*)logv;
  int* perm = (int*)((char*)logv + (size_t)JAC_MAX_SWEEPS * (n - 1) * (n / 2) * sizeof(double2));
  const size_t shmem = (32 + 256 + 256 + 64) * sizeof(double) + (use_lds ? (size_t)k * (k | 1) * sizeof(double) : 0);
  HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_eig, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(k_jacobi_eig, dim3(1), dim3(small_threads()), shmem, ctx->stream, sm_ptr(ctx, slot_t), SM_LD, k,
                     sm_ptr(ctx, SM_TMP), use_lds, rotlog, dvals, perm, sort_by_abs, ctx->status_dev);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_jacobi_vectors, dim3(k), dim3(128), 0, ctx->stream, rotlog, k, ctx->status_dev, perm,
                     sm_ptr(ctx, slot_v), SM_LD);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
