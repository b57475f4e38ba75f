// Symmetric eigensolve for matrices too large for the one-workgroup Jacobi kernel (n > 256): one-sided (Hestenes)
// Jacobi spread over the whole GPU.  Used by the deterministic POD of PODProjectorFromData when there are more than
// 256 snapshots (the n x n Gram problem of PODProjector.py:812-833, la.eigh there).
//
// Method: with s >= 0 chosen so that A + s I is positive definite, W = A + s I and V = I are rotated together from the
// right, W <- W J, V <- V J, one plane rotation per column pair, until the columns of W are mutually orthogonal.  Then
// W = (A + s I) V has orthogonal columns and V is orthogonal, so the columns of V are eigenvectors and
// lambda_i = ||w_i|| - s.  The n/2 pairs of a round touch disjoint columns: one workgroup per pair, all pairs of a
// round in one launch, rounds in a round-robin tournament; the host looks at the largest |cos| of a sweep once per
// sweep.  Columns are contiguous, so every access is a coalesced stream; a column pair stays in registers between its
// inner products and its rotation.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "hfmi_internal.h"

namespace {
constexpr int EL_THREADS = 256;
constexpr int EL_MAXROWS = 16;        // rows per thread: n <= 4096

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One round: workgroup b rotates columns (pairs[2b], pairs[2b+1]) of W and V.  offmax (device, double bits compared
// as unsigned: all values are non-negative) collects max |w_i . w_j| / (||w_i|| ||w_j||) over the sweep.
__global__ void __launch_bounds__(EL_THREADS) k_jacobi_large_round(double* __restrict__ W, double* __restrict__ V, int64_t ld, int n,
                                                                   const int* __restrict__ pairs, unsigned long long* offmax,
                                                                   double skip_tol) {
  __shared__ double red[3][EL_THREADS / 64];
  int ci = pairs[2 * blockIdx.x], cj = pairs[2 * blockIdx.x + 1];
  if (ci >= n || cj >= n) return;     // the bye of an odd tournament
  double* wi = W + (int64_t)ci * ld;
  double* wj = W + (int64_t)cj * ld;
  double a[EL_MAXROWS], b[EL_MAXROWS];
  double aa = 0.0, bb = 0.0, ab = 0.0;
#pragma unroll
  for (int t = 0; t < EL_MAXROWS; ++t) {
    const int r = threadIdx.x + t * EL_THREADS;
    a[t] = r < n ? wi[r] : 0.0;
    b[t] = r < n ? wj[r] : 0.0;
    aa += a[t] * a[t];
    bb += b[t] * b[t];
    ab += a[t] * b[t];
  }
  aa = wave_sum(aa);
  bb = wave_sum(bb);
  ab = wave_sum(ab);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = aa;
    red[1][wave] = bb;
    red[2][wave] = ab;
  }
  __syncthreads();
  aa = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  bb = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  ab = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  const double denom = sqrt(aa) * sqrt(bb);
  const double cosv = denom > 0.0 ? fabs(ab) / denom : 0.0;
  if (threadIdx.x == 0) atomicMax(offmax, (unsigned long long)__double_as_longlong(cosv));
  if (!(cosv > skip_tol)) return;
  // rotation that makes the two columns orthogonal (Rutishauser's formulas); the larger column ends up first
  const double zeta = (bb - aa) / (2.0 * ab);
  const double tn = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  double c = 1.0 / sqrt(1.0 + tn * tn), s = c * tn;
  // new norms: aa' = aa - tn ab, bb' = bb + tn ab; swap the roles if the second would be larger (de Rijk ordering)
  const bool swap = (bb + tn * ab) > (aa - tn * ab);
  double* vi = V + (int64_t)ci * ld;
  double* vj = V + (int64_t)cj * ld;
#pragma unroll
  for (int t = 0; t < EL_MAXROWS; ++t) {
    const int r = threadIdx.x + t * EL_THREADS;
    if (r < n) {
      const double x = c * a[t] - s * b[t], y = s * a[t] + c * b[t];
      wi[r] = swap ? y : x;
      wj[r] = swap ? x : y;
      const double p = vi[r], q = vj[r];
      const double xv = c * p - s * q, yv = s * p + c * q;
      vi[r] = swap ? yv : xv;
      vj[r] = swap ? xv : yv;
    }
  }
}

// out[j] = ||w_j||
__global__ void __launch_bounds__(EL_THREADS) k_col_norms_large(const double* __restrict__ W, int64_t ld, int n, double* __restrict__ out) {
  __shared__ double red[EL_THREADS / 64];
  const double* w = W + (int64_t)blockIdx.x * ld;
  double acc = 0.0;
  for (int r = threadIdx.x; r < n; r += EL_THREADS) acc += w[r] * w[r];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sqrt(red[0] + red[1] + red[2] + red[3]);
}
}  // namespace

// host_T: n x n row-major symmetric (the symmetric part is used); host_d: n eigenvalues descending (by |d| if
// sort_by_abs); host_V: n x n row-major, eigenvectors in the columns (may be null).
int sym_eig_large(hfmi_ctx* ctx, const double* host_T, int n, int sort_by_abs, double* host_d, double* host_V) {
  if (n > EL_MAXROWS * EL_THREADS) HFMI_FAIL(HFMI_ERR_INVALID, "sym_eig: n=%d exceeds %d", n, EL_MAXROWS * EL_THREADS);
  const int64_t ld = round_up(n, 32);
  std::vector<double> A((size_t)ld * n, 0.0);      // column-major, symmetrised
  double shift = 0.0;
  {
    double fro2 = 0.0, gersh = 0.0;
    for (int j = 0; j < n; ++j) {
      double off = 0.0;
      for (int i = 0; i < n; ++i) {
        const double v = 0.5 * (host_T[(size_t)i * n + j] + host_T[(size_t)j * n + i]);
        A[(size_t)j * ld + i] = v;
        fro2 += v * v;
        if (i != j) off += fabs(v);
      }
      gersh = std::min(gersh, A[(size_t)j * ld + j] - off);
    }
    // positive definite after the shift: every Gershgorin disc to the right of fro * 1e-3 (well away from zero so that
    // no column of A + s I is numerically null); |lambda| <= fro bounds the loss of absolute accuracy to ~2 eps fro
    const double fro = sqrt(fro2);
    if (gersh < 1e-3 * fro) shift = std::min(fro, -gersh) + 1e-3 * fro;
    if (!(fro > 0.0)) shift = 1.0;
    for (int j = 0; j < n; ++j) A[(size_t)j * ld + j] += shift;
  }
  void *wv = nullptr, *pv = nullptr;
  const size_t mat_bytes = (size_t)ld * n * sizeof(double);
  HFMI_TRY(ctx_ws(ctx, WS_STAGE, 2 * mat_bytes + (size_t)n * sizeof(double) + 64, &wv));
  double* W = (double*)wv;
  double* V = W + (size_t)ld * n;
  double* norms = V + (size_t)ld * n;
  unsigned long long* offmax = (unsigned long long*)(norms + n);
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
  HFMI_TRY(ctx_ws(ctx, WS_MISC, sched.size() * sizeof(int), &pv));
  int* dpairs = (int*)pv;
  HIP_TRY(hipMemcpyAsync(dpairs, sched.data(), sched.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(W, A.data(), mat_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));      // A and sched are pageable host memory
  HFMI_TRY(launch_fill(ctx, V, ld, n, ld, 0.0, true));
  {
    std::vector<double> eye((size_t)ld * n, 0.0);
    for (int j = 0; j < n; ++j) eye[(size_t)j * ld + j] = 1.0;
    HIP_TRY(hipMemcpyAsync(V, eye.data(), mat_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  const double eps = 2.220446049250313e-16;
  const double tol = 8.0 * eps;                    // columns orthogonal to working precision
  const int max_sweeps = 40;
  int sweeps = 0;
  double off = 1.0;
  for (; sweeps < max_sweeps; ++sweeps) {
    HIP_TRY(hipMemsetAsync(offmax, 0, sizeof(unsigned long long), ctx->stream));
    for (int r = 0; r < rounds; ++r)
      hipLaunchKernelGGL(k_jacobi_large_round, dim3(np), dim3(EL_THREADS), 0, ctx->stream, W, V, ld, n, dpairs + (size_t)r * np * 2, offmax,
                         0.5 * tol);
    HIP_TRY(hipGetLastError());
    unsigned long long bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, offmax, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(&off, &bits, sizeof(off));
    if (off <= tol) {
      ++sweeps;
      break;
    }
  }
  if (!(off <= tol)) HFMI_FAIL(HFMI_ERR_NOT_CONVERGED, "sym_eig (n=%d): one-sided Jacobi did not converge in %d sweeps (max cosine %.2e)", n, max_sweeps, off);
  hipLaunchKernelGGL(k_col_norms_large, dim3(n), dim3(EL_THREADS), 0, ctx->stream, W, ld, n, norms);
  HIP_TRY(hipGetLastError());
  std::vector<double> lam(n);
  HIP_TRY(hipMemcpyAsync(lam.data(), norms, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(A.data(), V, mat_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int j = 0; j < n; ++j) lam[j] -= shift;
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
