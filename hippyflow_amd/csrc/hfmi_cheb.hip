// M^-1 on a block by Jacobi-preconditioned CHEBYSHEV iteration (prior.Msolver behind hp.Solver2Operator in
// hp.doublePassG(KLE_Operator, prior.M, prior.Msolver, ...), KLEProjector.py:163-164; the mass-orthogonal QR applies it
// once per solve to an N x k block).
//
// Why not the block CG of hfmi_api.hip (kept: it estimates the spectrum once per matrix and is the fallback): per
// iteration CG needs two global reductions -- five launches, eleven passes over N x k blocks, a host look at the
// residual every fourth iteration -- and its SpMM with one thread per ROW gathers 8 bytes per lane from k different
// vectors (1.5 TB/s on config 2 while the elementwise kernels beside it run at 6 TB/s out of the Infinity Cache).
// Chebyshev needs no inner products at all: ONE kernel per iteration, six passes, nothing for the host to wait for.
//
// Layout: the iteration works on a ROW-MAJOR copy of the block (entry (row, j) at [row * k + j]; two transposes per
// solve): with lanes mapped to the k columns of one row every access is a contiguous run of the row -- the gathers of
// the neighbour rows of a sparse-matrix row included -- and the matrix entries of that row are the same for all its
// lanes (broadcast loads).  Any CSR matrix, no ELL image needed.
//
//   theta = (lmax + lmin) / 2, delta = (lmax - lmin) / 2, sigma = theta / delta, rho_0 = 1 / sigma
//   x_0 = 0, x_1 = D^-1 b / theta
//   rho' = 1 / (2 sigma - rho);  x_{n+1} = x_n + rho' rho (x_n - x_{n-1}) + (2 rho' / delta) D^-1 (b - A x_n);  rho = rho'
// -- the three-term form of the iteration: the residual is recomputed from b in every step (no residual or direction
// arrays, no drift), x_{n+1} overwrites x_{n-1} in place, and a step reads b, x_n (with its neighbour rows), x_{n-1} and
// writes x_{n+1}: FOUR passes over three N x k arrays -- 200 MB on config 2, resident in the 256 MB Infinity Cache.
// [lmin, lmax] bracket the spectrum of D^-1 A (hfmi_api.hip: Gershgorin from above, Lanczos of a scalar CG run from
// below); the error after n steps is at most 2 c^n / (1 + c^2n), c = (sqrt(kappa) - 1) / (sqrt(kappa) + 1).
#include <algorithm>

#include "hfmi_internal.h"

namespace {
// thread t of a 256-thread workgroup: row t / k of the workgroup's current group of rows, column t % k (k <= 256: 256 / k
// rows at a time); k > 256: one row at a time, the threads stride over its columns
struct rm_map {
  int rows_per_pass, rl, j0, jstep;
  bool live;
};
__device__ __forceinline__ rm_map rm_thread(int k) {
  rm_map m;
  if (k <= 256) {
    m.rows_per_pass = 256 / k;
    m.rl = threadIdx.x / k;
    m.j0 = threadIdx.x - m.rl * k;
    m.jstep = k;                       // one column per thread
    m.live = m.rl < m.rows_per_pass;
  } else {
    m.rows_per_pass = 1;
    m.rl = 0;
    m.j0 = threadIdx.x;
    m.jstep = 256;
    m.live = true;
  }
  return m;
}

// x_1 = D^-1 B / theta  (x_0 = 0 is never stored: the first step runs with c1 = 0 ... see launch_cheb_step)
__global__ __launch_bounds__(256) void k_cheb_first(const double* __restrict__ B, double* __restrict__ X1, double* __restrict__ X0,
                                                   const double* __restrict__ inv_diag, int64_t nrows, int k, double inv_theta) {
  const rm_map m = rm_thread(k);
  for (int64_t r0 = (int64_t)blockIdx.x * m.rows_per_pass; r0 < nrows; r0 += (int64_t)gridDim.x * m.rows_per_pass) {
    const int64_t row = r0 + m.rl;
    if (!m.live || row >= nrows) continue;
    const double s = inv_diag[row] * inv_theta;
    for (int j = m.j0; j < k; j += m.jstep) {
      const int64_t e = row * k + j;
      X1[e] = s * B[e];
      X0[e] = 0.0;
    }
  }
}

// One step of the three-term iteration, ONE WAVE PER ROW (k even): lane l holds columns 2l, 2l+1 as one 16-byte access, so the
// gather of a neighbour row is one instruction for up to 128 columns, and the row's (column, value) pairs are wave-uniform:
// scalar loads, no vector-memory traffic for the matrix.  UNR rows per wave for memory-level parallelism.
//   RESID = false:  Xp <- Xc + c1 (Xc - Xp) + c2 D^-1 (B - A Xc)      (Xp holds x_{n-1} on entry, x_{n+1} on exit)
//   RESID = true:   Xp <- B - A Xc                                     (the true residual, for the convergence check)
typedef double d2v __attribute__((ext_vector_type(2)));
template <bool RESID, int UNR>
__global__ __launch_bounds__(256) void k_cheb_step_wave(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                       const double* __restrict__ data, const double* __restrict__ inv_diag,
                                                       int64_t nrows, int k, const double* __restrict__ B, const double* __restrict__ Xc,
                                                       double* __restrict__ Xp, double c1, double c2) {
  // the matrix through the constant address space: with wave-uniform addresses the compiler then issues scalar loads (it will
  // not for plain global pointers once the kernel also stores, restrict or not)
  typedef const int64_t __attribute__((address_space(4)))* cptr64;
  typedef const int32_t __attribute__((address_space(4)))* cptr32;
  typedef const double __attribute__((address_space(4)))* cptrd;
  const cptr64 c_indptr = (cptr64)indptr;
  const cptr32 c_indices = (cptr32)indices;
  const cptrd c_data = (cptrd)data, c_inv_diag = (cptrd)inv_diag;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Workgroup b works on the band of
  // rows that belongs to ITS XCD (b % 8), so that the neighbour rows a sparse row gathers were fetched by the same L2.
  const int64_t per_xcd = gridDim.x / 8;                       // the grid is a multiple of 8
  const int64_t logical = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const int64_t row0 = (logical * 4 + wave) * UNR;             // this wave's UNR consecutive rows
  if (row0 >= nrows) return;
  const int half = k >> 1;
  for (int l0 = 0; l0 < half; l0 += 64) {                      // k <= 128: one trip
    const int l = l0 + lane;
    const bool on = l < half;
    const int64_t col2 = on ? 2 * l : 0;
    d2v ax[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) ax[u] = d2v{0.0, 0.0};
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t row = row0 + u < nrows ? row0 + u : nrows - 1;
      const int64_t zb = c_indptr[row], ze = c_indptr[row + 1];
      for (int64_t z = zb; z < ze; ++z) {
        const double v = c_data[z];
        const d2v g = *reinterpret_cast<const d2v*>(Xc + (int64_t)c_indices[z] * k + col2);
        ax[u] += v * g;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t row = row0 + u;
      if (row >= nrows || !on) continue;
      const int64_t e = row * k + col2;
      const d2v res = *reinterpret_cast<const d2v*>(B + e) - ax[u];
      if (RESID) {
        *reinterpret_cast<d2v*>(Xp + e) = res;
      } else {
        const d2v xc = *reinterpret_cast<const d2v*>(Xc + e);
        const d2v xp = *reinterpret_cast<const d2v*>(Xp + e);
        *reinterpret_cast<d2v*>(Xp + e) = xc + c1 * (xc - xp) + (c2 * c_inv_diag[row]) * res;
      }
    }
  }
}

// the same step with one thread per (row, column) -- odd k, or k > 128 columns
template <bool RESID>
__global__ __launch_bounds__(256) void k_cheb_step(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                  const double* __restrict__ data, const double* __restrict__ inv_diag,
                                                  int64_t nrows, int k, const double* __restrict__ B, const double* __restrict__ Xc,
                                                  double* __restrict__ Xp, double c1, double c2) {
  const rm_map m = rm_thread(k);
  if (!m.live) return;
  const int64_t per_xcd = gridDim.x / 8;
  const int64_t logical = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const int64_t row = logical * m.rows_per_pass + m.rl;
  if (row >= nrows) return;
  const int64_t zb = indptr[row], ze = indptr[row + 1];
  const double s = c2 * inv_diag[row];
  for (int j = m.j0; j < k; j += m.jstep) {
    double ax = 0.0;
    for (int64_t z = zb; z < ze; ++z) ax += data[z] * Xc[(int64_t)indices[z] * k + j];
    const int64_t e = row * k + j;
    const double res = B[e] - ax;
    if (RESID) {
      Xp[e] = res;
    } else {
      const double xc = Xc[e];
      Xp[e] = xc + c1 * (xc - Xp[e]) + s * res;
    }
  }
}

// per-column sums of squares of a row-major (nrows x k) array: one partial per (column, workgroup)
__global__ __launch_bounds__(256) void k_rm_colsq(const double* __restrict__ A, int64_t nrows, int k, double* __restrict__ part) {
  extern __shared__ double sh[];       // 256 doubles
  const rm_map m = rm_thread(k);
  for (int jb = 0; jb < k; jb += 256) {                // k <= 256: one trip
    double acc = 0.0;
    const int j = (k <= 256) ? m.j0 : jb + (int)threadIdx.x;
    if (m.live && j < k)
      for (int64_t row = (int64_t)blockIdx.x * m.rows_per_pass + m.rl; row < nrows; row += (int64_t)gridDim.x * m.rows_per_pass) {
        const double v = A[row * k + j];
        acc += v * v;
      }
    sh[threadIdx.x] = (m.live && j < k) ? acc : 0.0;
    __syncthreads();
    if (k <= 256) {
      if ((int)threadIdx.x < k) {                      // fixed order over the row groups of this workgroup
        double s = 0.0;
        for (int g = 0; g < m.rows_per_pass; ++g) s += sh[g * k + threadIdx.x];
        part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
      }
    } else if (j < k) {
      part[(int64_t)j * gridDim.x + blockIdx.x] = sh[threadIdx.x];
    }
    __syncthreads();
  }
}
}  // namespace

static inline unsigned cheb_grid(hfmi_ctx* ctx, int64_t nrows, int k) {
  const int rows_per_pass = k <= 256 ? 256 / k : 1;
  int64_t g = (nrows + rows_per_pass - 1) / rows_per_pass;
  const int64_t cap = (int64_t)(ctx->num_cus > 0 ? ctx->num_cus : 256) * 64;
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}
int launch_cheb_first(hfmi_ctx* ctx, const double* B, double* X1, double* X0, const double* inv_diag, int64_t nrows, int k, double inv_theta) {
  hipLaunchKernelGGL(k_cheb_first, dim3(cheb_grid(ctx, nrows, k)), dim3(256), 0, ctx->stream, B, X1, X0, inv_diag, nrows, k, inv_theta);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
// one step (resid = false) or the residual B - A Xc into Xp (resid = true)
int launch_cheb_step(hfmi_ctx* ctx, const hfmi_csr* M, const double* B, const double* Xc, double* Xp, int k, double c1, double c2, bool resid) {
  static const bool thread_map = env_flag("HFMI_CHEB_THREADMAP");      // A/B switch
  if ((k & 1) == 0 && k <= 128 && !thread_map) {
    // rows per wave: 1 measured best on config 2 (53.5 us a step; 2: 54.4, 4: 63.0, 8: 104 -- profiles/r04c_cheb_unr.txt)
    static const int unr = getenv("HFMI_CHEB_UNR") ? atoi(getenv("HFMI_CHEB_UNR")) : 1;
    int64_t gw = (M->nrows + 4 * unr - 1) / (4 * unr);          // 4 waves x UNR rows per workgroup
    gw = (gw + 7) / 8 * 8;                                      // whole bands for the 8 XCDs
#define HFMI_CHEB_LAUNCH(RES, U) hipLaunchKernelGGL((k_cheb_step_wave<RES, U>), dim3((unsigned)gw), dim3(256), 0, ctx->stream, M->indptr, M->indices, M->data, M->inv_diag, M->nrows, k, B, Xc, Xp, c1, c2)
    if (resid) { gw = ((M->nrows + 3) / 4 + 7) / 8 * 8; HFMI_CHEB_LAUNCH(true, 1); }
    else if (unr == 2) HFMI_CHEB_LAUNCH(false, 2);
    else if (unr == 4) HFMI_CHEB_LAUNCH(false, 4);
    else { gw = ((M->nrows + 3) / 4 + 7) / 8 * 8; HFMI_CHEB_LAUNCH(false, 1); }
#undef HFMI_CHEB_LAUNCH
    HIP_TRY(hipGetLastError());
    return HFMI_OK;
  }
  const int rows_per_pass = k <= 256 ? 256 / k : 1;
  int64_t g = (M->nrows + rows_per_pass - 1) / rows_per_pass;
  g = (g + 7) / 8 * 8;
  if (resid)
    hipLaunchKernelGGL((k_cheb_step<true>), dim3((unsigned)g), dim3(256), 0, ctx->stream, M->indptr, M->indices, M->data, M->inv_diag, M->nrows, k, B, Xc, Xp, c1, c2);
  else
    hipLaunchKernelGGL((k_cheb_step<false>), dim3((unsigned)g), dim3(256), 0, ctx->stream, M->indptr, M->indices, M->data, M->inv_diag, M->nrows, k, B, Xc, Xp, c1, c2);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
// out[j] = sum over rows of A[row][j]^2 (deterministic: fixed partition, fixed order)
int launch_rm_colsq(hfmi_ctx* ctx, const double* A, int64_t nrows, int k, double* out) {
  const int chunks = 1024;
  void* part = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)k * chunks * sizeof(double), &part));
  hipLaunchKernelGGL(k_rm_colsq, dim3(chunks), dim3(256), 256 * sizeof(double), ctx->stream, A, nrows, k, (double*)part);
  HIP_TRY(hipGetLastError());
  return launch_dots_final(ctx, (const double*)part, chunks, k, out);
}
