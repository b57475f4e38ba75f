// HBM-bound helper kernels: block fill/copy/axpy, the Philox probe draw, layout conversion,
// CSR SpMM, per-vector dot products and the fused vector updates of the block PCG, plus the
// micro-benchmarks that measure the roofline denominators in the same job.
#include "hfmi_internal.h"
#include "hfmi_randn_math.h"

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// 2-D elementwise launch geometry: x over row pairs (16 B per lane), y over vectors (grid-stride)
static inline dim3 ew_grid(int64_t N, int nvec) {
  const int64_t pairs = (N + 1) / 2;
  int64_t gx = (pairs + 255) / 256;
  if (gx > 4096) gx = 4096;
  int gy = nvec < 1024 ? nvec : 1024;
  return dim3((unsigned)gx, (unsigned)gy);
}

__global__ void k_fill(double* __restrict__ p, int64_t rows, int nvec, int64_t ld, double value) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < rows; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < rows) *reinterpret_cast<d2*>(c + t) = d2{value, value};
      else c[t] = value;
    }
  }
}
__global__ void k_zero_pad(double* __restrict__ p, int64_t N, int nvec, int64_t ld) {
  const int64_t padn = ld - N;
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld + N;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < padn; t += (int64_t)gridDim.x * blockDim.x) c[t] = 0.0;
  }
}
// Synthetic config-2 covariance: vector j of the block is column j of the Matern-3/2 kernel matrix
// sigma^2 (1 + a) exp(-a), a = sqrt(3) |x_i - x_j| / ell, over the first N nodes (row-major numbering) of an nx x ny grid
// on the unit square.  Written once per run; 16-byte stores.
__global__ void k_matern32(double* __restrict__ p, int64_t N, int nvec, int64_t ld, int nx, double hx, double hy, double sigma2,
                           double inv_ell) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld;
    const double xj = (j % nx) * hx, yj = (j / nx) * hy;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      double v[2];
      for (int e = 0; e < 2; ++e) {
        const int64_t i = t + e;
        const double dx = (i % nx) * hx - xj, dy = (i / nx) * hy - yj;
        const double a = 1.7320508075688772 * sqrt(dx * dx + dy * dy) * inv_ell;
        v[e] = sigma2 * (1.0 + a) * exp(-a);
      }
      if (t + 1 < N) *reinterpret_cast<d2*>(c + t) = d2{v[0], v[1]};
      else c[t] = v[0];
    }
  }
}
int launch_matern32(hfmi_ctx* ctx, double* p, int64_t N, int nvec, int64_t ld, int nx, int ny, double sigma, double ell) {
  hipLaunchKernelGGL(k_matern32, ew_grid(N, nvec), dim3(256), 0, ctx->stream, p, N, nvec, ld, nx, 1.0 / (nx - 1), 1.0 / (ny - 1),
                     sigma * sigma, 1.0 / ell);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
// G (m x k, row-major, ld) <- diag(w) G: the diagonal of hp.LowRankOperator(d, U) between its two contractions
__global__ void k_row_scale(double* __restrict__ G, int ld, int m, int k, const double* __restrict__ w) {
  const int i = blockIdx.x;
  const double wi = w[i];
  for (int j = threadIdx.x; j < k; j += blockDim.x) G[(int64_t)i * ld + j] *= wi;
}
int launch_row_scale(hfmi_ctx* ctx, double* G, int ld, int m, int k, const double* w) {
  if (m <= 0 || k <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_row_scale, dim3(m), dim3(64), 0, ctx->stream, G, ld, m, k, w);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
int launch_fill(hfmi_ctx* ctx, double* p, int64_t N, int nvec, int64_t ld, double value, bool include_pad) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  const int64_t rows = include_pad ? ld : N;
  hipLaunchKernelGGL(k_fill, ew_grid(rows, nvec), dim3(256), 0, ctx->stream, p, rows, nvec, ld, value);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
int launch_zero_pad(hfmi_ctx* ctx, double* p, int64_t N, int nvec, int64_t ld) {
  if (ld <= N || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_zero_pad, dim3(1, nvec < 1024 ? nvec : 1024), dim3(64), 0, ctx->stream, p, N, nvec, ld);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

__global__ void k_copy(double* __restrict__ dst, int64_t ldd, const double* __restrict__ src, int64_t lds, int64_t N, int nvec) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* d = dst + (int64_t)j * ldd;
    const double* s = src + (int64_t)j * lds;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) *reinterpret_cast<d2*>(d + t) = *reinterpret_cast<const d2*>(s + t);
      else d[t] = s[t];
    }
  }
}
int launch_copy(hfmi_ctx* ctx, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t N, int nvec) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_copy, ew_grid(N, nvec), dim3(256), 0, ctx->stream, dst, ldd, src, lds, N, nvec);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

__global__ void k_scale(double* __restrict__ p, int64_t ld, int64_t N, int nvec, double alpha) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        d2 v = *reinterpret_cast<d2*>(c + t);
        v.x *= alpha; v.y *= alpha;
        *reinterpret_cast<d2*>(c + t) = v;
      } else c[t] *= alpha;
    }
  }
}
int launch_scale(hfmi_ctx* ctx, double* p, int64_t ld, int64_t N, int nvec, double alpha) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_scale, ew_grid(N, nvec), dim3(256), 0, ctx->stream, p, ld, N, nvec, alpha);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

__global__ void k_axpy(double* __restrict__ y, int64_t ldy, double alpha, const double* __restrict__ x, int64_t ldx, int64_t N, int nvec) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* yc = y + (int64_t)j * ldy;
    const double* xc = x + (int64_t)j * ldx;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        d2 v = *reinterpret_cast<d2*>(yc + t);
        const d2 u = *reinterpret_cast<const d2*>(xc + t);
        v.x += alpha * u.x; v.y += alpha * u.y;
        *reinterpret_cast<d2*>(yc + t) = v;
      } else yc[t] += alpha * xc[t];
    }
  }
}
int launch_axpy(hfmi_ctx* ctx, double* y, int64_t ldy, double alpha, const double* x, int64_t ldx, int64_t N, int nvec) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_axpy, ew_grid(N, nvec), dim3(256), 0, ctx->stream, y, ldy, alpha, x, ldx, N, nvec);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// ------------------------------------------------------------------ Philox4x32-10 + Box-Muller
// hfmi_randn_math.h: the integer generator, the range-specific fp64 functions and the element map (four normals per
// Philox output, rows 4g .. 4g+3 of column j from counter (g, j, stream)).  Thread = one row group per step, one 32-byte
// store; a wave writes 2 KiB contiguous.  VEC = false: blocks wrapped around foreign memory whose columns are not
// 32-byte aligned.
using hfmi_rng::philox4x32_10;
template <bool VEC>
__global__ __launch_bounds__(256) void k_randn(double* __restrict__ p, int64_t N, int nvec, int64_t ld, uint32_t k0, uint32_t k1,
                                               uint32_t stream, double sigma) {
  const int64_t nfull = N >> 2, ngroups = (N + 3) >> 2;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  const hfmi_rng::normal_consts kc = hfmi_rng::make_consts(sigma);
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld;
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; g < nfull; g += step) {
      uint32_t x[4];
      double z[4];
      philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)j, stream, k0, k1, x);
      hfmi_rng::box_muller4(x, kc, z);
      if (VEC) *reinterpret_cast<d4*>(c + 4 * g) = d4{z[0], z[1], z[2], z[3]};
      else { c[4 * g] = z[0]; c[4 * g + 1] = z[1]; c[4 * g + 2] = z[2]; c[4 * g + 3] = z[3]; }
    }
    if (g < ngroups) {   // the ragged last group: one thread of the grid
      uint32_t x[4];
      double z[4];
      philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)j, stream, k0, k1, x);
      hfmi_rng::box_muller4(x, kc, z);
      for (int i = 0; i < 3; ++i) if (4 * g + i < N) c[4 * g + i] = z[i];
    }
  }
}
static inline dim3 randn_grid(int64_t N, int nvec) {
  const int64_t ngroups = (N + 3) / 4;
  int64_t gx = (ngroups + 256 * 4 - 1) / (256 * 4);   // ~4 row groups per thread
  if (gx < 1) gx = 1;
  if (gx > 4096) gx = 4096;
  return dim3((unsigned)gx, (unsigned)(nvec < 1024 ? nvec : 1024));
}
int launch_randn(hfmi_ctx* ctx, double* p, int64_t N, int nvec, int64_t ld, uint64_t seed, uint32_t stream, double sigma) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  if (sigma == 0.0) return launch_fill(ctx, p, N, nvec, ld, 0.0, false);
  const bool vec = ((uintptr_t)p % 32 == 0) && (ld % 4 == 0 || nvec == 1);
  if (vec) hipLaunchKernelGGL(k_randn<true>, randn_grid(N, nvec), dim3(256), 0, ctx->stream, p, N, nvec, ld, (uint32_t)seed, (uint32_t)(seed >> 32), stream, sigma);
  else hipLaunchKernelGGL(k_randn<false>, randn_grid(N, nvec), dim3(256), 0, ctx->stream, p, N, nvec, ld, (uint32_t)seed, (uint32_t)(seed >> 32), stream, sigma);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
// the raw integer stream behind the draw: out[(j * ngroups + g) * 4 + 0..3], ngroups = ceil(N / 4)
__global__ void k_philox_raw(uint32_t* __restrict__ out, int64_t ngroups, int nvec, uint32_t k0, uint32_t k1, uint32_t stream) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y)
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
      uint32_t x[4];
      philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)j, stream, k0, k1, x);
      uint32_t* o = out + ((int64_t)j * ngroups + g) * 4;
      o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; o[3] = x[3];
    }
}
int launch_philox_raw(hfmi_ctx* ctx, uint32_t* out, int64_t N, int nvec, uint64_t seed, uint32_t stream) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_philox_raw, randn_grid(N, nvec), dim3(256), 0, ctx->stream, out, (N + 3) / 4, nvec, (uint32_t)seed, (uint32_t)(seed >> 32), stream);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// ------------------------------------------------------------------ dense (N x nvec, row-major) <-> block
// 32 x 32 tiles through LDS so both sides are read/written in 256-byte runs.
template <bool TO_BLOCK>
__global__ void k_transpose(const double* __restrict__ src, double* __restrict__ dst, int64_t ld, int64_t ldd, int64_t N, int nvec) {
  __shared__ double tile[32][33];
  const int64_t t0 = (int64_t)blockIdx.x * 32;
  const int j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  if (TO_BLOCK) {  // src dense[t*nvec + j], dst block[j*ld + t]
    for (int r = ty; r < 32; r += 8) {
      const int64_t t = t0 + r; const int j = j0 + tx;
      if (t < N && j < nvec) tile[r][tx] = src[t * ldd + j];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int j = j0 + r; const int64_t t = t0 + tx;
      if (t < N && j < nvec) dst[(int64_t)j * ld + t] = tile[tx][r];
    }
  } else {  // src block[j*ld + t], dst dense[t*nvec + j]
    for (int r = ty; r < 32; r += 8) {
      const int j = j0 + r; const int64_t t = t0 + tx;
      if (t < N && j < nvec) tile[r][tx] = src[(int64_t)j * ld + t];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int64_t t = t0 + r; const int j = j0 + tx;
      if (t < N && j < nvec) dst[t * ldd + j] = tile[tx][r];
    }
  }
}
int launch_dense_to_block(hfmi_ctx* ctx, const double* dense, double* p, int64_t ld, int64_t N, int nvec) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  dim3 grid((unsigned)((N + 31) / 32), (unsigned)((nvec + 31) / 32));
  hipLaunchKernelGGL(k_transpose<true>, grid, dim3(256), 0, ctx->stream, dense, p, ld, (int64_t)nvec, N, nvec);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
int launch_block_to_dense_ld(hfmi_ctx* ctx, const double* p, int64_t ld, double* dense, int64_t ldd, int64_t N, int nvec) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  dim3 grid((unsigned)((N + 31) / 32), (unsigned)((nvec + 31) / 32));
  hipLaunchKernelGGL(k_transpose<false>, grid, dim3(256), 0, ctx->stream, p, dense, ld, ldd, N, nvec);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
int launch_block_to_dense(hfmi_ctx* ctx, const double* p, int64_t ld, double* dense, int64_t N, int nvec) {
  return launch_block_to_dense_ld(ctx, p, ld, dense, nvec, N, nvec);
}

// ELL variant: thread = row, the (index, value) pairs of a slot are contiguous across the rows of a wave, SPMM_EB
// vectors per pass.  DOT: also emit this workgroup's part of <X_j, Y_j> (the p . A p of PCG) -- one partial per
// (vector, workgroup), summed in a fixed order by k_col_dots_final.
constexpr int SPMM_EB = 12;
template <bool DOT>
__global__ __launch_bounds__(256) void k_ell_spmm(const int32_t* __restrict__ eidx, const double* __restrict__ eval,
                                                  int w, int64_t nrows, const double* __restrict__ X, int64_t ldx,
                                                  double* __restrict__ Y, int64_t ldy, int nvec, int accumulate,
                                                  double* __restrict__ part, int gx8, int gy) {
  __shared__ double wsum[4][SPMM_EB];
  // XCD-aware order (1-D grid of gx8 * gy workgroups, gx8 a multiple of 8): workgroups are dealt round-robin to the 8 XCDs;
  // XCD x takes the x-th band of row blocks, for every group of vectors, so that the rows a band gathers from (its own and
  // the neighbouring ones) are fetched into ONE L2.  With the natural order every L2 pulled most of X through the fabric:
  // 453 MB of HBM reads per launch for 143 MB of operands on config 2 (profiles/r04c PMC pass).
  const int per = gx8 >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = xcd * per + slot % per, yg = slot / per;
  const int64_t row = (int64_t)rb * blockDim.x + threadIdx.x;
  const bool live = row < nrows;
  const int64_t r = live ? row : nrows - 1;
  for (int j0 = yg * SPMM_EB; j0 < nvec; j0 += gy * SPMM_EB) {
    double acc[SPMM_EB];
#pragma unroll
    for (int jj = 0; jj < SPMM_EB; ++jj) acc[jj] = 0.0;
    for (int s = 0; s < w; ++s) {
      const double v = eval[(int64_t)s * nrows + r];
      const double* xr = X + eidx[(int64_t)s * nrows + r];
#pragma unroll
      for (int jj = 0; jj < SPMM_EB; ++jj) {
        const int j = j0 + jj < nvec ? j0 + jj : nvec - 1;
        acc[jj] += v * xr[(int64_t)j * ldx];
      }
    }
#pragma unroll
    for (int jj = 0; jj < SPMM_EB; ++jj)
      if (live && j0 + jj < nvec) {
        double* y = Y + (int64_t)(j0 + jj) * ldy + row;
        *y = accumulate ? *y + acc[jj] : acc[jj];
      }
    if (DOT) {
#pragma unroll
      for (int jj = 0; jj < SPMM_EB; ++jj) {
        const int j = j0 + jj < nvec ? j0 + jj : nvec - 1;
        double d = live ? acc[jj] * X[(int64_t)j * ldx + row] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][jj] = d;
      }
      __syncthreads();
      if ((int)threadIdx.x < SPMM_EB && j0 + (int)threadIdx.x < nvec)
        part[(int64_t)(j0 + threadIdx.x) * gx8 + rb] =
            (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------ CSR SpMM: Y[:, j] (+)= M X[:, j]
// One lane per matrix row (consecutive lanes -> consecutive rows: coalesced Y stores; the gathers of X hit
// neighbouring rows for FEM matrices), 8 vectors per pass so the row's (index, value) pairs are read once per 8.
constexpr int SPMM_JB = 8;
__global__ void k_csr_spmm(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                           const double* __restrict__ data, int64_t nrows, const double* __restrict__ X, int64_t ldx,
                           double* __restrict__ Y, int64_t ldy, int nvec, int accumulate, int gx8, int gy) {
  // XCD-aware row bands, as in k_ell_spmm: 1-D grid of gx8 * gy workgroups, XCD x (= blockIdx.x % 8) owns the x-th band of rows
  const int per = gx8 >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = xcd * per + slot % per, yg = slot / per;
  const int64_t row = (int64_t)rb * blockDim.x + threadIdx.x;
  if (row >= nrows) return;
  const int64_t b = indptr[row], e = indptr[row + 1];
  for (int j0 = yg * SPMM_JB; j0 < nvec; j0 += gy * SPMM_JB) {
    double acc[SPMM_JB];
#pragma unroll
    for (int jj = 0; jj < SPMM_JB; ++jj) acc[jj] = 0.0;
    for (int64_t z = b; z < e; ++z) {
      const double v = data[z];
      const double* xr = X + indices[z];
#pragma unroll
      for (int jj = 0; jj < SPMM_JB; ++jj)
        if (j0 + jj < nvec) acc[jj] += v * xr[(int64_t)(j0 + jj) * ldx];
    }
#pragma unroll
    for (int jj = 0; jj < SPMM_JB; ++jj)
      if (j0 + jj < nvec) {
        double* y = Y + (int64_t)(j0 + jj) * ldy + row;
        *y = accumulate ? *y + acc[jj] : acc[jj];
      }
  }
}
int launch_csr_spmm(hfmi_ctx* ctx, const hfmi_csr* M, const double* X, int64_t ldx, double* Y, int64_t ldy, int nvec, bool accumulate) {
  if (M->nrows <= 0 || nvec <= 0) return HFMI_OK;
  if (M->ell_w > 0) {
    const int gx8 = (int)(((M->nrows + 255) / 256 + 7) / 8 * 8), gy = (nvec + SPMM_EB - 1) / SPMM_EB;
    hipLaunchKernelGGL((k_ell_spmm<false>), dim3((unsigned)(gx8 * gy)), dim3(256), 0, ctx->stream, M->ell_idx, M->ell_val, M->ell_w, M->nrows, X,
                       ldx, Y, ldy, nvec, accumulate ? 1 : 0, (double*)nullptr, gx8, gy);
    HIP_TRY(hipGetLastError());
    return HFMI_OK;
  }
  const int gy = (nvec + SPMM_JB - 1) / SPMM_JB;
  const int gx8 = (int)(((M->nrows + 255) / 256 + 7) / 8 * 8);
  hipLaunchKernelGGL(k_csr_spmm, dim3((unsigned)(gx8 * gy)), dim3(256), 0, ctx->stream, M->indptr, M->indices, M->data, M->nrows, X, ldx, Y, ldy, nvec, accumulate ? 1 : 0, gx8, gy);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
__global__ void k_csr_diag_inv(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                               const double* __restrict__ data, int64_t nrows, double* __restrict__ inv_diag) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrows) return;
  double d = 0.0;
  for (int64_t z = indptr[row]; z < indptr[row + 1]; ++z)
    if (indices[z] == row) d += data[z];
  inv_diag[row] = d != 0.0 ? 1.0 / d : 1.0;
}
int launch_csr_diag_inv(hfmi_ctx* ctx, hfmi_csr* M) {
  hipLaunchKernelGGL(k_csr_diag_inv, dim3((unsigned)((M->nrows + 255) / 256)), dim3(256), 0, ctx->stream, M->indptr, M->indices, M->data, M->nrows, M->inv_diag);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// ------------------------------------------------------------------ per-vector dot products (deterministic)
// grid (chunks, vectors): wave-64 shuffle reduction -> LDS across the 4 waves -> one partial per block;
// a second kernel adds the partials in a fixed order.
constexpr int DOT_CHUNKS = 64;
__global__ void k_col_dots_partial(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
                                   int64_t N, int nvec, double* __restrict__ part) {
  __shared__ double wsum[4];
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    const double* a = A + (int64_t)j * lda;
    const double* b = B + (int64_t)j * ldb;
    double s = 0.0;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        const d2 u = *reinterpret_cast<const d2*>(a + t), v = *reinterpret_cast<const d2*>(b + t);
        s += u.x * v.x + u.y * v.y;
      } else s += a[t] * b[t];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(int64_t)j * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}
// one wave per output: lanes stride over the partials, then a fixed-order shuffle tree (deterministic); the serial
// version of this loop cost 50-100 us per call once a fused kernel produced a few hundred partials per vector
__global__ __launch_bounds__(64) void k_col_dots_final(const double* __restrict__ part, int nchunks, int nvec,
                                                       double* __restrict__ out) {
  const int j = blockIdx.x;
  if (j >= nvec) return;
  const double* p = part + (int64_t)j * nchunks;
  double s = 0.0;
  for (int c = threadIdx.x; c < nchunks; c += 64) s += p[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) out[j] = s;
}
int launch_dots_final(hfmi_ctx* ctx, const double* part, int nchunks, int nvec, double* out) {
  hipLaunchKernelGGL(k_col_dots_final, dim3(nvec), dim3(64), 0, ctx->stream, part, nchunks, nvec, out);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
int launch_col_dots(hfmi_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, int64_t N, int nvec, double* out) {
  if (nvec <= 0) return HFMI_OK;
  void* part = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)nvec * DOT_CHUNKS * sizeof(double), &part));
  dim3 grid(DOT_CHUNKS, nvec < 1024 ? nvec : 1024);
  hipLaunchKernelGGL(k_col_dots_partial, grid, dim3(256), 0, ctx->stream, A, lda, B, ldb, N, nvec, (double*)part);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_col_dots_final, dim3(nvec), dim3(64), 0, ctx->stream, (const double*)part, DOT_CHUNKS, nvec, out);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// y_j += sign * coef_j * x_j,  coef_j = num_j / den_j (den == null: num_j); a zero denominator gives 0 (converged vector)
__global__ void k_col_axpy_dev(double* __restrict__ y, int64_t ldy, const double* __restrict__ x, int64_t ldx, int64_t N, int nvec,
                               const double* __restrict__ num, const double* __restrict__ den, double sign) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double coef = num[j];
    if (den) coef = den[j] != 0.0 ? coef / den[j] : 0.0;
    coef *= sign;
    double* yc = y + (int64_t)j * ldy;
    const double* xc = x + (int64_t)j * ldx;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        d2 v = *reinterpret_cast<d2*>(yc + t);
        const d2 u = *reinterpret_cast<const d2*>(xc + t);
        v.x += coef * u.x; v.y += coef * u.y;
        *reinterpret_cast<d2*>(yc + t) = v;
      } else yc[t] += coef * xc[t];
    }
  }
}
int launch_col_axpy_dev(hfmi_ctx* ctx, double* y, int64_t ldy, const double* x, int64_t ldx, int64_t N, int nvec, const double* num, const double* den, double sign) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_col_axpy_dev, ew_grid(N, nvec), dim3(256), 0, ctx->stream, y, ldy, x, ldx, N, nvec, num, den, sign);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
__global__ void k_col_xpby_dev(double* __restrict__ p, int64_t ldp, const double* __restrict__ z, int64_t ldz, int64_t N, int nvec,
                               const double* __restrict__ num, const double* __restrict__ den) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    const double coef = den[j] != 0.0 ? num[j] / den[j] : 0.0;
    double* pc = p + (int64_t)j * ldp;
    const double* zc = z + (int64_t)j * ldz;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        d2 v = *reinterpret_cast<d2*>(pc + t);
        const d2 u = *reinterpret_cast<const d2*>(zc + t);
        v.x = u.x + coef * v.x; v.y = u.y + coef * v.y;
        *reinterpret_cast<d2*>(pc + t) = v;
      } else pc[t] = zc[t] + coef * pc[t];
    }
  }
}
int launch_col_xpby_dev(hfmi_ctx* ctx, double* p, int64_t ldp, const double* z, int64_t ldz, int64_t N, int nvec, const double* num, const double* den) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_col_xpby_dev, ew_grid(N, nvec), dim3(256), 0, ctx->stream, p, ldp, z, ldz, N, nvec, num, den);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
__global__ void k_diag_scale(double* __restrict__ z, int64_t ldz, const double* __restrict__ r, int64_t ldr, const double* __restrict__ inv_diag, int64_t N, int nvec) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* zc = z + (int64_t)j * ldz;
    const double* rc = r + (int64_t)j * ldr;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (int64_t)gridDim.x * blockDim.x) zc[t] = inv_diag[t] * rc[t];
  }
}
int launch_diag_scale(hfmi_ctx* ctx, double* z, int64_t ldz, const double* r, int64_t ldr, const double* inv_diag, int64_t N, int nvec) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_diag_scale, ew_grid(2 * N, nvec), dim3(256), 0, ctx->stream, z, ldz, r, ldr, inv_diag, N, nvec);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

int launch_ell_spmm_dot(hfmi_ctx* ctx, const hfmi_csr* M, const double* X, int64_t ldx, double* Y, int64_t ldy, int nvec,
                        double* dots) {
  if (M->ell_w <= 0) HFMI_FAIL(HFMI_ERR_INVALID, "ell_spmm_dot: matrix has no ELL image");
  const int gx = (int)(((M->nrows + 255) / 256 + 7) / 8 * 8), gy = (nvec + SPMM_EB - 1) / SPMM_EB;
  void* part = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)nvec * gx * sizeof(double), &part));
  hipLaunchKernelGGL((k_ell_spmm<true>), dim3((unsigned)(gx * gy)), dim3(256), 0, ctx->stream, M->ell_idx, M->ell_val, M->ell_w, M->nrows, X, ldx,
                     Y, ldy, nvec, 0, (double*)part, gx, gy);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_col_dots_final, dim3(nvec), dim3(64), 0, ctx->stream, (const double*)part, (int)gx, nvec, dots);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// Fused PCG update (one pass over y, r, p, ap instead of four kernels): per column j, alpha = rz_j / pap_j,
// y += alpha p, r -= alpha ap, partial sums of <r, D^-1 r> and <r, r>.  z = D^-1 r is never stored.
constexpr int PCG_CHUNKS = 128;
__global__ __launch_bounds__(256) void k_pcg_update(double* __restrict__ y, int64_t ldy, double* __restrict__ r, int64_t ldr,
                                                    const double* __restrict__ p, int64_t ldp,
                                                    const double* __restrict__ ap, int64_t ldap,
                                                    const double* __restrict__ dinv, int64_t N, int nvec,
                                                    const double* __restrict__ rz, const double* __restrict__ pap,
                                                    double* __restrict__ part) {
  __shared__ double wsum[4][2];
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    const double a = pap[j] != 0.0 ? rz[j] / pap[j] : 0.0;
    double* yc = y + (int64_t)j * ldy;
    double* rc = r + (int64_t)j * ldr;
    const double* pc = p + (int64_t)j * ldp;
    const double* ac = ap + (int64_t)j * ldap;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        d2 yv = *reinterpret_cast<d2*>(yc + t), rv = *reinterpret_cast<d2*>(rc + t);
        const d2 pv = *reinterpret_cast<const d2*>(pc + t), av = *reinterpret_cast<const d2*>(ac + t);
        const d2 dv = *reinterpret_cast<const d2*>(dinv + t);
        yv.x += a * pv.x; yv.y += a * pv.y;
        rv.x -= a * av.x; rv.y -= a * av.y;
        *reinterpret_cast<d2*>(yc + t) = yv;
        *reinterpret_cast<d2*>(rc + t) = rv;
        s1 += rv.x * (dv.x * rv.x) + rv.y * (dv.y * rv.y);
        s2 += rv.x * rv.x + rv.y * rv.y;
      } else {
        yc[t] += a * pc[t];
        const double rv = rc[t] - a * ac[t];
        rc[t] = rv;
        s1 += rv * (dinv[t] * rv);
        s2 += rv * rv;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s1 += __shfl_down(s1, off, 64);
      s2 += __shfl_down(s2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      wsum[threadIdx.x >> 6][0] = s1;
      wsum[threadIdx.x >> 6][1] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 2)
      part[((int64_t)threadIdx.x * nvec + j) * gridDim.x + blockIdx.x] =
          (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
    __syncthreads();
  }
}
int launch_pcg_update(hfmi_ctx* ctx, double* y, int64_t ldy, double* r, int64_t ldr, const double* p, int64_t ldp,
                      const double* ap, int64_t ldap, const double* inv_diag, int64_t N, int nvec, const double* rz,
                      const double* pap, double* rz_new, double* rr) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  if (rr != rz_new + nvec) HFMI_FAIL(HFMI_ERR_INVALID, "pcg_update: rr must follow rz_new in memory");
  void* part = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)2 * nvec * PCG_CHUNKS * sizeof(double), &part));
  dim3 grid(PCG_CHUNKS, nvec < 1024 ? nvec : 1024);
  hipLaunchKernelGGL(k_pcg_update, grid, dim3(256), 0, ctx->stream, y, ldy, r, ldr, p, ldp, ap, ldap, inv_diag, N, nvec, rz,
                     pap, (double*)part);
  HIP_TRY(hipGetLastError());
  // the two partial arrays are laid out back to back as 2 * nvec "vectors": one final pass fills rz_new | rr
  hipLaunchKernelGGL(k_col_dots_final, dim3(2 * nvec), dim3(64), 0, ctx->stream, (const double*)part, PCG_CHUNKS, 2 * nvec,
                     rz_new);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
__global__ void k_pcg_direction(double* __restrict__ p, int64_t ldp, const double* __restrict__ r, int64_t ldr,
                                const double* __restrict__ dinv, int64_t N, int nvec, const double* __restrict__ num,
                                const double* __restrict__ den) {
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    const double b = den[j] != 0.0 ? num[j] / den[j] : 0.0;
    double* pc = p + (int64_t)j * ldp;
    const double* rc = r + (int64_t)j * ldr;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        d2 pv = *reinterpret_cast<d2*>(pc + t);
        const d2 rv = *reinterpret_cast<const d2*>(rc + t), dv = *reinterpret_cast<const d2*>(dinv + t);
        pv.x = dv.x * rv.x + b * pv.x;
        pv.y = dv.y * rv.y + b * pv.y;
        *reinterpret_cast<d2*>(pc + t) = pv;
      } else pc[t] = dinv[t] * rc[t] + b * pc[t];
    }
  }
}
int launch_pcg_direction(hfmi_ctx* ctx, double* p, int64_t ldp, const double* r, int64_t ldr, const double* inv_diag,
                         int64_t N, int nvec, const double* rz_new, const double* rz) {
  if (N <= 0 || nvec <= 0) return HFMI_OK;
  hipLaunchKernelGGL(k_pcg_direction, ew_grid(N, nvec), dim3(256), 0, ctx->stream, p, ldp, r, ldr, inv_diag, N, nvec, rz_new, rz);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// G_i <- Gamma G_i for every sample's q x k slab (noise precision, operatorWrappers.py:107-109).  One
// workgroup per sample, slab staged in LDS (q*k*8 bytes: 100 x 74 -> 59 KB).
__global__ void k_gamma_apply(double* __restrict__ G, int ldg, int q, int k, const double* __restrict__ gamma, int ldgam) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* slab = reinterpret_cast<double*>(smem);
  double* Gi = G + (int64_t)blockIdx.x * q * ldg;
  for (int e = threadIdx.x; e < q * k; e += blockDim.x) slab[e] = Gi[(int64_t)(e / k) * ldg + (e % k)];
  __syncthreads();
  for (int e = threadIdx.x; e < q * k; e += blockDim.x) {
    const int o = e / k, j = e % k;
    double s = 0.0;
    for (int p = 0; p < q; ++p) s += gamma[o * ldgam + p] * slab[p * k + j];
    Gi[(int64_t)o * ldg + j] = s;
  }
}
int launch_gamma_apply(hfmi_ctx* ctx, double* G, int ldg, int ndata, int q, int k, const double* gamma, int ldgam) {
  const size_t shmem = (size_t)q * k * sizeof(double);
  if (shmem > 150 * 1024) HFMI_FAIL(HFMI_ERR_INVALID, "gamma_apply: q*k slab (%zu bytes) exceeds LDS", shmem);
  HIP_TRY(hipFuncSetAttribute((const void*)k_gamma_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(k_gamma_apply, dim3(ndata), dim3(256), shmem, ctx->stream, G, ldg, q, k, gamma, ldgam);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// column-major variant for the Rayleigh-quotient shortcut: Gc is (ndata*q) x k with leading dimension ld (one
// column per probe vector); out[:, j] slab i = Gamma * Gc[:, j] slab i
__global__ void k_gamma_apply_cm(const double* __restrict__ Gc, double* __restrict__ out, int64_t ld, int q, int k,
                                 const double* __restrict__ gamma, int ldgam) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* slab = reinterpret_cast<double*>(smem);     // [k][q]
  const int64_t base = (int64_t)blockIdx.x * q;
  for (int e = threadIdx.x; e < q * k; e += blockDim.x) {
    const int j = e / q, o = e % q;
    slab[e] = Gc[(int64_t)j * ld + base + o];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < q * k; e += blockDim.x) {
    const int j = e / q, o = e % q;
    double s = 0.0;
    for (int p = 0; p < q; ++p) s += gamma[o * ldgam + p] * slab[j * q + p];
    out[(int64_t)j * ld + base + o] = s;
  }
}
int launch_gamma_apply_cm(hfmi_ctx* ctx, const double* Gc, double* out, int64_t ld, int ndata, int q, int k, const double* gamma,
                          int ldgam) {
  const size_t shmem = (size_t)q * k * sizeof(double);
  if (shmem > 150 * 1024) HFMI_FAIL(HFMI_ERR_INVALID, "gamma_apply: q*k slab (%zu bytes) exceeds LDS", shmem);
  HIP_TRY(hipFuncSetAttribute((const void*)k_gamma_apply_cm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(k_gamma_apply_cm, dim3(ndata), dim3(256), shmem, ctx->stream, Gc, out, ld, q, k, gamma, ldgam);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

// ------------------------------------------------------------------ micro-benchmarks (roofline denominators)
__global__ __launch_bounds__(256, 2) void k_bench_mfma(double* out, int iters) {
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256, 2) void k_bench_fma(double* out, int iters) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-9 + i;
  const double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 123.456) out[0] = s;
}
__global__ void k_bench_copy(const d2* __restrict__ src, d2* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int launch_bench_peaks(hfmi_ctx* ctx, double* mfma_tflops, double* fma_tflops, double* copy_gbs) {
  void* buf = nullptr;
  const size_t bytes = (size_t)2 << 30;  // 1 GiB read + 1 GiB write per copy
  HFMI_TRY(ctx_ws(ctx, WS_STAGE, bytes, &buf));
  double* out = (double*)buf;
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  float ms = 0.f;
  // fp64 MFMA: 4 waves/CU (one per SIMD) x 8 independent accumulators
  const int it1 = 2000, g1 = cus * 16;
  hipLaunchKernelGGL(k_bench_mfma, dim3(g1), dim3(256), 0, ctx->stream, out, 100);
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(k_bench_mfma, dim3(g1), dim3(256), 0, ctx->stream, out, it1);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *mfma_tflops = (double)g1 * 4 * it1 * 8 * (2.0 * 16 * 16 * 4) / (ms * 1e-3) / 1e12;
  // fp64 vector FMA: 8 waves/CU x 16 independent chains
  const int it2 = 2000, g2 = cus * 32;
  hipLaunchKernelGGL(k_bench_fma, dim3(g2), dim3(256), 0, ctx->stream, out, 100);
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(k_bench_fma, dim3(g2), dim3(256), 0, ctx->stream, out, it2);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *fma_tflops = (double)g2 * 256 * it2 * 16 * 2.0 / (ms * 1e-3) / 1e12;
  // HBM copy
  const int64_t n = (int64_t)(bytes / 2 / sizeof(d2));
  d2* src = (d2*)buf;
  d2* dst = src + n;
  hipLaunchKernelGGL(k_bench_copy, dim3(cus * 8), dim3(256), 0, ctx->stream, src, dst, n);
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k_bench_copy, dim3(cus * 8), dim3(256), 0, ctx->stream, src, dst, n);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *copy_gbs = 5.0 * (double)bytes / (ms * 1e-3) / 1e9;
  return HFMI_OK;
}

// The same loop on FULL-MANTISSA operands: every lane holds eight Gaussian values per operand, and each MFMA of an iteration takes
// another pair of them (the register pairs rotate through the eight accumulators), so operand and accumulator bits toggle like
// those of the solve's contractions on Gaussian data.  The constant-operand loop above multiplies 1 +- tid 1e-9: almost no
// toggling, a clock (2.35-2.40 GHz) that random data does not see under the power limit (MI355X_MICROARCH.md, "DVFS give-back").
__global__ __launch_bounds__(256, 2) void k_bench_mfma_rand(const double* __restrict__ vals, double* out, int iters) {
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a[8], b[8];
  const double* src = vals + ((size_t)blockIdx.x % 64) * 4096 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[256 * i];
    b[i] = src[256 * (8 + i)];
  }
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(i + r) & 7], b[(i + 3 * r) & 7], acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;
}
__global__ void k_bench_copy_loop(const d2* __restrict__ src, d2* __restrict__ dst, int64_t n, int reps) {
  for (int rep = 0; rep < reps; ++rep)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
      __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
// The fp64 MFMA rate this box sustains WHILE HBM is being streamed: the constant-operand MFMA loop on the context's stream with
// the copy kernel running beside it on the auxiliary stream (two workgroups per CU, ~3 TB/s).  The big contractions of the
// solve run in exactly that regime (their clock is set by the power limit), so this, not the cold
// micro-benchmark, is the in-job ceiling their fraction should be read against.
int launch_bench_loaded_peak(hfmi_ctx* ctx, double* mfma_tflops, double* copy_gbs) {
  void* buf = nullptr;
  const size_t bytes = (size_t)2 << 30;
  HFMI_TRY(ctx_ws(ctx, WS_STAGE, bytes + 4096, &buf));
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  const int64_t n = (int64_t)(bytes / 2 / sizeof(d2));
  d2* src = (d2*)buf;
  d2* dst = src + n;
  double* out = (double*)(dst + n);
  hipEvent_t c0, c1;
  HIP_TRY(hipEventCreate(&c0));
  HIP_TRY(hipEventCreate(&c1));
  const int it1 = 2000, g1 = cus * 16, copies = 24, reps = 3;
  hipLaunchKernelGGL(k_bench_mfma, dim3(g1), dim3(256), 0, ctx->stream, out, 100);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipEventRecord(c0, ctx->aux_stream));
  // ONE resident launch (its workgroups keep their slots beside the MFMA waves), four workgroups per CU
  hipLaunchKernelGGL(k_bench_copy_loop, dim3(cus * 4), dim3(256), 0, ctx->aux_stream, src, dst, n, copies);
  HIP_TRY(hipEventRecord(c1, ctx->aux_stream));
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k_bench_mfma, dim3(g1), dim3(256), 0, ctx->stream, out, it1);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventSynchronize(c1));
  float ms = 0.f, msc = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  HIP_TRY(hipEventElapsedTime(&msc, c0, c1));
  (void)hipEventDestroy(c0);
  (void)hipEventDestroy(c1);
  *mfma_tflops = (double)reps * g1 * 4 * it1 * 8 * (2.0 * 16 * 16 * 4) / (ms * 1e-3) / 1e12;
  *copy_gbs = (double)copies * (double)bytes / (msc * 1e-3) / 1e9;   // over the whole copy window (part of it runs alone)
  return HFMI_OK;
}

// read-only stream: what a contraction that only READS its big operand can get from HBM (the copy above also writes)
__global__ __launch_bounds__(256) void k_bench_read(const d2* __restrict__ src, double* out, int64_t n) {
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 * 8 + threadIdx.x; i + 7 * 256 < n; i += stride) {
    d2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
  }
  if (s == 123.456) out[0] = s;
}
int launch_bench_read(hfmi_ctx* ctx, double* read_gbs) {
  void* buf = nullptr;
  const size_t bytes = (size_t)2 << 30;
  HFMI_TRY(ctx_ws(ctx, WS_STAGE, bytes + 4096, &buf));
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  const int64_t n = (int64_t)(bytes / sizeof(d2));
  double* out = (double*)((char*)buf + bytes);
  float ms = 0.f;
  hipLaunchKernelGGL(k_bench_read, dim3(cus * 8), dim3(256), 0, ctx->stream, (const d2*)buf, out, n);
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k_bench_read, dim3(cus * 8), dim3(256), 0, ctx->stream, (const d2*)buf, out, n);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *read_gbs = 5.0 * (double)bytes / (ms * 1e-3) / 1e9;
  return HFMI_OK;
}

// fp64 MFMA rate on Gaussian operands: alone, and with the copy kernel streaming HBM beside it (the regime of the solve's big
// contractions) -- the denominators bench.py reports as roofline.frac_of_in_job_random_operand_peak[_while_streaming]
int launch_bench_random_peaks(hfmi_ctx* ctx, double* mfma_tflops, double* mfma_tflops_streaming, double* copy_gbs) {
  void* buf = nullptr;
  const size_t bytes = (size_t)2 << 30;
  const size_t nvals = (size_t)64 * 4096;
  HFMI_TRY(ctx_ws(ctx, WS_STAGE, bytes + 4096 + nvals * sizeof(double), &buf));
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  const int64_t n = (int64_t)(bytes / 2 / sizeof(d2));
  d2* src = (d2*)buf;
  d2* dst = src + n;
  double* out = (double*)(dst + n);
  double* vals = out + 512;
  HFMI_TRY(launch_randn(ctx, vals, (int64_t)nvals, 1, (int64_t)nvals, 0x5eedULL, 7u, 1.0));
  const int it1 = 2000, g1 = cus * 16, copies = 24, reps = 3;
  float ms = 0.f, msc = 0.f;
  hipLaunchKernelGGL(k_bench_mfma_rand, dim3(g1), dim3(256), 0, ctx->stream, vals, out, 104);
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k_bench_mfma_rand, dim3(g1), dim3(256), 0, ctx->stream, vals, out, it1);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *mfma_tflops = (double)reps * g1 * 4 * it1 * 8 * (2.0 * 16 * 16 * 4) / (ms * 1e-3) / 1e12;
  hipEvent_t c0, c1;
  HIP_TRY(hipEventCreate(&c0));
  HIP_TRY(hipEventCreate(&c1));
  HIP_TRY(hipEventRecord(c0, ctx->aux_stream));
  hipLaunchKernelGGL(k_bench_copy_loop, dim3(cus * 4), dim3(256), 0, ctx->aux_stream, src, dst, n, copies);
  HIP_TRY(hipEventRecord(c1, ctx->aux_stream));
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k_bench_mfma_rand, dim3(g1), dim3(256), 0, ctx->stream, vals, out, it1);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventSynchronize(c1));
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  HIP_TRY(hipEventElapsedTime(&msc, c0, c1));
  (void)hipEventDestroy(c0);
  (void)hipEventDestroy(c1);
  *mfma_tflops_streaming = (double)reps * g1 * 4 * it1 * 8 * (2.0 * 16 * 16 * 4) / (ms * 1e-3) / 1e12;
  *copy_gbs = (double)copies * (double)bytes / (msc * 1e-3) / 1e9;
  return HFMI_OK;
}
