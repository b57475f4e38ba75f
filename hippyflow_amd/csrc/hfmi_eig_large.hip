// Two-sided Jacobi with a parallel (tournament) ordering, spread over the whole GPU: the eigensolver of rounds 2-4 for
// 256 < n <= 4096 (the n x n Gram problem of PODProjector.py:812-833, la.eigh there).  Launch-bound; since round 5 the default
// is the blocked tridiagonalisation + divide and conquer of hfmi_eig_blocked.hip, and this one stays behind
// HFMI_EIG_LARGE=jacobi as the A/B partner and as an independent check of it on the device.
//
// A round of the tournament holds n/2 disjoint index pairs (p, q).  T <- J^T T J with J the product of the round's
// plane rotations is applied in two launches: (1) one workgroup per pair computes its rotation from T_pp, T_qq, T_pq
// and rotates COLUMNS p, q of T and of the accumulated eigenvector matrix V (contiguous streams); (2) one workgroup
// per column applies all the round's rotations to the ROW pairs inside its column.  A pair whose off-diagonal entry
// is already negligible relative to its diagonal entries (|T_pq| <= eps sqrt(|T_pp T_qq|): high relative accuracy for
// positive definite Gram matrices) is skipped; a sweep without a single rotation ends the iteration -- the host looks
// at one counter per sweep.  (A one-sided Hestenes iteration on the columns of T was tried first: it works on T^2
// implicitly and needed > 40 sweeps at cond(T) = 4e7; the two-sided form converges in 10-20.)
#include <math.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "hfmi_internal.h"

namespace {
constexpr int EL_THREADS = 256;

// launch 1 of a round: rotation of pair b from the current T, columns of T and V rotated, (c, s) kept for launch 2
__global__ void __launch_bounds__(EL_THREADS) k_jacobi_large_cols(double* __restrict__ T, double* __restrict__ V, int64_t ld, int n,
                                                                  const int* __restrict__ pairs, double2* __restrict__ cs,
                                                                  unsigned int* __restrict__ nrot, double abs_floor) {
  const int p = pairs[2 * blockIdx.x], q = pairs[2 * blockIdx.x + 1];
  if (p >= n || q >= n) {             // the bye of an odd tournament
    if (threadIdx.x == 0) cs[blockIdx.x] = make_double2(1.0, 0.0);
    return;
  }
  double* tp = T + (int64_t)p * ld;
  double* tq = T + (int64_t)q * ld;
  // ONE thread reads the pivot entries and decides: the loop below overwrites exactly those entries (rows p and q of the
  // two columns), so a wavefront that started late must not derive its own (c, s) from half-rotated values
  __shared__ double sh_c, sh_s;
  __shared__ int sh_rotate;
  if (threadIdx.x == 0) {
    const double app = tp[p], aqq = tq[q], apq = tq[p];
    const double eps = 2.220446049250313e-16;
    const bool rot = fabs(apq) > eps * sqrt(fabs(app * aqq)) && fabs(apq) > abs_floor;
    double c = 1.0, s = 0.0;
    if (rot) {
      const double tau = (aqq - app) / (2.0 * apq);
      const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      c = 1.0 / sqrt(1.0 + t * t);
      s = t * c;
      atomicAdd(nrot, 1u);
    }
    cs[blockIdx.x] = make_double2(c, s);
    sh_c = c;
    sh_s = s;
    sh_rotate = rot ? 1 : 0;
  }
  __syncthreads();
  if (!sh_rotate) return;            // uniform: every thread reads the same flag after the barrier
  const double c = sh_c, s = sh_s;
  double* vp = V + (int64_t)p * ld;
  double* vq = V + (int64_t)q * ld;
  for (int r = threadIdx.x; r < n; r += EL_THREADS) {
    const double x = tp[r], y = tq[r];
    tp[r] = c * x - s * y;
    tq[r] = s * x + c * y;
    const double u = vp[r], w = vq[r];
    vp[r] = c * u - s * w;
    vq[r] = s * u + c * w;
  }
}
// launch 2: column blockIdx.x of T gets every rotation of the round applied to its row pairs
__global__ void __launch_bounds__(EL_THREADS) k_jacobi_large_rows(double* __restrict__ T, int64_t ld, int n, int npairs,
                                                                  const int* __restrict__ pairs, const double2* __restrict__ cs) {
  double* col = T + (int64_t)blockIdx.x * ld;
  for (int b = threadIdx.x; b < npairs; b += EL_THREADS) {
    const double2 r = cs[b];
    if (r.y == 0.0) continue;
    const int p = pairs[2 * b], q = pairs[2 * b + 1];
    const double x = col[p], y = col[q];
    col[p] = r.x * x - r.y * y;
    col[q] = r.y * x + r.x * y;
  }
}
__global__ void k_diag_large(const double* __restrict__ T, int64_t ld, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = T[(int64_t)i * ld + i];
}
}  // namespace

// host_T: n x n row-major symmetric (the symmetric part is used); host_d: n eigenvalues descending (by |d| if
// sort_by_abs); host_V: n x n row-major, eigenvectors in the columns (may be null).
int sym_eig_large_jacobi(hfmi_ctx* ctx, const double* host_T, int n, int sort_by_abs, double* host_d, double* host_V) {
  if (n > 4096) HFMI_FAIL(HFMI_ERR_INVALID, "sym_eig: n=%d exceeds 4096", n);
  const int64_t ld = round_up(n, 32);
  std::vector<double> A((size_t)ld * n, 0.0);      // column-major, symmetrised
  double fro2 = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      const double v = 0.5 * (host_T[(size_t)i * n + j] + host_T[(size_t)j * n + i]);
      A[(size_t)j * ld + i] = v;
      fro2 += v * v;
    }
  const double fro = sqrt(fro2);
  // tournament schedule: m = n rounded up to even players, m - 1 rounds of m / 2 pairs
  const int m = (n + 1) & ~1, rounds = m - 1, np = m / 2;
  std::vector<int> sched((size_t)rounds * np * 2);
  {
    std::vector<int> pos(m);
    std::iota(pos.begin(), pos.end(), 0);
    for (int r = 0; r < rounds; ++r) {
      for (int p = 0; p < np; ++p) {
        int a = pos[p], b = pos[m - 1 - p];
        if (a > b) std::swap(a, b);
        sched[((size_t)r * np + p) * 2] = a;
        sched[((size_t)r * np + p) * 2 + 1] = b;
      }
      const int last = pos[m - 1];                 // player 0 stays, the others rotate
      for (int i = m - 1; i > 1; --i) pos[i] = pos[i - 1];
      pos[1] = last;
    }
  }
  void *wv = nullptr, *pv = nullptr;
  const size_t mat_bytes = (size_t)ld * n * sizeof(double);
  HFMI_TRY(ctx_ws(ctx, WS_STAGE, 2 * mat_bytes + (size_t)n * sizeof(double) + (size_t)np * sizeof(double2) + 64, &wv));
  double* T = (double*)wv;
  double* V = T + (size_t)ld * n;
  double* diag = V + (size_t)ld * n;
  double2* cs = (double2*)(diag + n + (n & 1));    // 16-byte aligned
  unsigned int* nrot = (unsigned int*)(cs + np);
  HFMI_TRY(ctx_ws(ctx, WS_MISC, sched.size() * sizeof(int), &pv));
  int* dpairs = (int*)pv;
  HIP_TRY(hipMemcpyAsync(dpairs, sched.data(), sched.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(T, A.data(), mat_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));      // A and sched are pageable host memory
  std::fill(A.begin(), A.end(), 0.0);
  for (int j = 0; j < n; ++j) A[(size_t)j * ld + j] = 1.0;
  HIP_TRY(hipMemcpyAsync(V, A.data(), mat_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // entries below eps ||T||_F / 10 are round-off of the matrix itself (a rank-deficient Gram matrix -- mean-shifted or
  // low-rank snapshot sets -- has a whole block of them, which the relative test alone would rotate for ever); what is
  // left unrotated perturbs an eigenvalue by at most n eps ||T||_F / 10, the absolute accuracy of LAPACK's eigh
  const double abs_floor = 0.1 * 2.220446049250313e-16 * fro;
  const int max_sweeps = 60;
  int sweeps = 0;
  unsigned int rotated = 1;
  for (; sweeps < max_sweeps && rotated; ++sweeps) {
    HIP_TRY(hipMemsetAsync(nrot, 0, sizeof(unsigned int), ctx->stream));
    for (int r = 0; r < rounds; ++r) {
      const int* rp = dpairs + (size_t)r * np * 2;
      hipLaunchKernelGGL(k_jacobi_large_cols, dim3(np), dim3(EL_THREADS), 0, ctx->stream, T, V, ld, n, rp, cs, nrot, abs_floor);
      hipLaunchKernelGGL(k_jacobi_large_rows, dim3(n), dim3(EL_THREADS), 0, ctx->stream, T, ld, n, np, rp, cs);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&rotated, nrot, sizeof(rotated), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (rotated) HFMI_FAIL(HFMI_ERR_NOT_CONVERGED, "sym_eig (n=%d): Jacobi still rotating %u pairs after %d sweeps", n, rotated, max_sweeps);
  hipLaunchKernelGGL(k_diag_large, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, T, ld, n, diag);
  HIP_TRY(hipGetLastError());
  std::vector<double> lam(n);
  HIP_TRY(hipMemcpyAsync(lam.data(), diag, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(A.data(), V, mat_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) {
    return sort_by_abs ? fabs(lam[x]) > fabs(lam[y]) : lam[x] > lam[y];
  });
  for (int j = 0; j < n; ++j) host_d[j] = lam[perm[j]];
  if (host_V)
    for (int j = 0; j < n; ++j) {
      const double* col = A.data() + (size_t)perm[j] * ld;
      for (int i = 0; i < n; ++i) host_V[(size_t)i * n + j] = col[i];
    }
  return HFMI_OK;
}
