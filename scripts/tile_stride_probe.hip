// Does the leading dimension of the n x n matrix of the whole-GPU eigensolver (ld = n rounded up to 128: a power of two at n = 4096 / 8192)
// cost HBM bandwidth in the lower-triangle products of the tridiagonalisation (k_tri_bs: one 128 x 128 tile per workgroup, a column segment
// of a tile = 1 KB contiguous, consecutive columns ld * 8 bytes apart)?  The same access pattern (512 threads, 16 x 16-byte loads per lane,
// tiles I >= J of the trailing block) at several leading dimensions; GB/s over 20 launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hippyflow_amd/build/tile_stride_probe scripts/tile_stride_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k_tiles(const double* __restrict__ A, long ld, double* out) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  int I = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= (int)blockIdx.x) ++I;
  while (I * (I + 1) / 2 > (int)blockIdx.x) --I;
  const int J = blockIdx.x - I * (I + 1) / 2;
  const double* Ab = A + (size_t)(128 * I) + (size_t)(128 * J) * ld;
  d2 x[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) x[u] = ((const d2*)(Ab + (size_t)(w + 8 * u) * ld))[l];
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) s += x[u].x + x[u].y;
  if (s == 123.456) out[blockIdx.x] = s;
}
// the full-column pattern of k_tri_b: a wave per column, 8 x 16-byte loads per lane in flight
__global__ __launch_bounds__(512) void k_cols(const double* __restrict__ A, long ld, int n, double* out) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  double s = 0.0;
  for (int q = blockIdx.x * 8 + w; q < n; q += gridDim.x * 8) {
    const d2* c2 = (const d2*)(A + (size_t)q * ld);
    for (int i = l; i < n / 2; i += 512) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const d2 v = c2[i + 64 * u]; s += v.x + v.y; }
    }
  }
  if (s == 123.456) out[blockIdx.x] = s;
}

int main() {
  const int ns[] = {8192, 6144, 4096};
  const int pads[] = {0, 16, 32, 144, 528};
  double* A = nullptr; double* out = nullptr;
  CK(hipMalloc(&A, (size_t)(8192 + 528) * 8192 * 8 + 4096));
  CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(A, 0, (size_t)(8192 + 528) * 8192 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int n : ns) for (int pad : pads) {
    const long ld = n + pad;
    const int nb = n / 128, ntiles = nb * (nb + 1) / 2;
    for (int kind = 0; kind < 2; ++kind) {
      auto run = [&] { if (kind == 0) hipLaunchKernelGGL(k_tiles, dim3(ntiles), dim3(512), 0, 0, A, ld, out);
                       else hipLaunchKernelGGL(k_cols, dim3(512), dim3(512), 0, 0, A, ld, n, out); };
      run();
      CK(hipEventRecord(e0));
      for (int r = 0; r < 20; ++r) run();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = kind == 0 ? (double)ntiles * 128 * 128 * 8 : (double)n * n * 8;
      printf("n=%d ld=%ld %-28s %8.1f us/launch  %7.1f GB/s\n", n, ld, kind == 0 ? "lower-triangle tiles (bs)" : "full columns (b)", ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e9);
    }
  }
  return 0;
}
