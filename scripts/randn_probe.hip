// Probe for the Box-Muller kernel (a1): times the candidate bodies on one 1e6 x 384 block (3.07 GB written) and
// measures the accuracy of the hardware seeds and of the range-specific functions against libm on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ihippyflow_amd/csrc -o gpurun_out/randn_probe scripts/randn_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hfmi_randn_math.h"
using namespace hfmi_rng;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// V0: the round-3 kernel (OCML log / sqrt / sincospi), one pair per thread and grid step
__global__ void k_v0(double* __restrict__ p, int64_t N, int nvec, int64_t ld, uint32_t k0, uint32_t k1, uint32_t stream, double sigma) {
  const int64_t npairs = (N + 1) / 2;
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * blockDim.x) {
      uint32_t x[4];
      philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)j, stream, k0, k1, x);
      const uint64_t a = (((uint64_t)x[1] << 32) | x[0]) >> 11;
      const uint64_t b = (((uint64_t)x[3] << 32) | x[2]) >> 11;
      const double u1 = ((double)a + 0.5) * 0x1.0p-53;
      const double u2 = ((double)b + 0.5) * 0x1.0p-53;
      const double rad = sigma * sqrt(-2.0 * log(u1));
      double sn, cs;
      sincospi(2.0 * u2, &sn, &cs);
      const int64_t t = 2 * q;
      if (t + 1 < N) *reinterpret_cast<d2*>(c + t) = d2{rad * cs, rad * sn};
      else c[t] = rad * cs;
    }
  }
}

// the product kernel's body (hippyflow_amd/csrc/hfmi_misc.hip k_randn<true>): four normals per Philox output, one 32-byte store
__global__ __launch_bounds__(256) void k_final(double* __restrict__ p, int64_t N, int nvec, int64_t ld, uint32_t k0, uint32_t k1, uint32_t stream, double sigma) {
  const int64_t nfull = N >> 2;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  const normal_consts kc = make_consts(sigma);
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nfull; g += step) {
      uint32_t x[4];
      double z[4];
      philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)j, stream, k0, k1, x);
      box_muller4(x, kc, z);
      *reinterpret_cast<d4*>(c + 4 * g) = d4{z[0], z[1], z[2], z[3]};
    }
  }
}
// the same map through OCML's log / sqrt / sincospi: the accuracy reference on the device
__global__ __launch_bounds__(256) void k_final_ocml(double* __restrict__ p, int64_t N, int nvec, int64_t ld, uint32_t k0, uint32_t k1, uint32_t stream, double sigma) {
  const int64_t nfull = N >> 2;
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nfull; g += (int64_t)gridDim.x * blockDim.x) {
      uint32_t x[4];
      philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)j, stream, k0, k1, x);
      for (int h = 0; h < 2; ++h) {
        const double u1 = ((double)x[2 * h] + 0.5) * 0x1.0p-32, u2 = ((double)x[2 * h + 1] + 0.5) * 0x1.0p-32;
        const double rad = sigma * sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        c[4 * g + 2 * h] = rad * cs;
        c[4 * g + 2 * h + 1] = rad * sn;
      }
    }
  }
}

// Philox only: the cost of the integer stream + the store
__global__ __launch_bounds__(256) void k_philox_only(double* __restrict__ p, int64_t N, int nvec, int64_t ld, uint32_t k0, uint32_t k1, uint32_t stream) {
  const int64_t npairs = (N + 1) / 2;
  for (int j = blockIdx.y; j < nvec; j += gridDim.y) {
    double* c = p + (int64_t)j * ld;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * blockDim.x) {
      uint32_t x[4];
      philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)j, stream, k0, k1, x);
      const int64_t t = 2 * q;
      if (t + 1 < N) *reinterpret_cast<uint4*>(c + t) = uint4{x[0], x[1], x[2], x[3]};
    }
  }
}
__global__ void k_fill(double* __restrict__ p, int64_t n) {
  for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t + 1 < n; t += (int64_t)gridDim.x * blockDim.x * 2)
    *reinterpret_cast<d2*>(p + t) = d2{1.0, 2.0};
}
template <class F>
static double time_ms(F&& launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int64_t N = 1000000; const int nvec = 384; const int64_t ld = N;
  double* p; CK(hipMalloc(&p, sizeof(double) * ld * nvec));
  const double gb = 8.0 * N * nvec / 1e9;
  const int64_t npairs = N / 2;
  auto grid1 = dim3((unsigned)((npairs + 255) / 256 > 4096 ? 4096 : (npairs + 255) / 256), nvec);
  printf("block %lld x %d = %.3f GB\n", (long long)N, nvec, gb);
  double t;
  t = time_ms([&] { hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, p, ld * nvec); }, 20);
  printf("%-40s %.4f ms  %.2f TB/s\n", "plain fill", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(k_v0, grid1, dim3(256), 0, 0, p, N, nvec, ld, 7u, 9u, 3u, 1.0); }, 20);
  printf("%-40s %.4f ms  %.2f TB/s\n", "round-3 kernel (OCML, 53-bit map)", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(k_philox_only, grid1, dim3(256), 0, 0, p, N, nvec, ld, 7u, 9u, 3u); }, 20);
  printf("%-40s %.4f ms  %.2f TB/s\n", "philox only, one output per 16 bytes", t, gb / t);
  for (int gx : {977, 489, 245, 123, 62, 16}) {
    t = time_ms([&] { hipLaunchKernelGGL(k_final, dim3(gx, nvec), dim3(256), 0, 0, p, N, nvec, ld, 7u, 9u, 3u, 1.0); }, 20);
    printf("final kernel, grid.x %-19d %.4f ms  %.2f TB/s  = %.3f of 8 TB/s\n", gx, t, gb / t, gb / t / 8.0);
  }
  {
    const int kv = 8;
    double* p0; CK(hipMalloc(&p0, sizeof(double) * N * kv));
    hipLaunchKernelGGL(k_final_ocml, dim3(245, kv), dim3(256), 0, 0, p0, N, kv, N, 7u, 9u, 3u, 1.5);
    hipLaunchKernelGGL(k_final, dim3(245, kv), dim3(256), 0, 0, p, N, kv, N, 7u, 9u, 3u, 1.5);
    std::vector<double> a(N * kv), b(N * kv);
    CK(hipMemcpy(a.data(), p0, sizeof(double) * N * kv, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), p, sizeof(double) * N * kv, hipMemcpyDeviceToHost));
    double md = 0, mx = 0, m1 = 0, m2 = 0, m4 = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      md = fmax(md, fabs(a[i] - b[i]));
      const double v = b[i] / 1.5;
      mx = fmax(mx, fabs(v)); m1 += v; m2 += v * v; m4 += v * v * v * v;
    }
    const double n = (double)a.size();
    printf("final vs the same map through OCML over %.0e normals: max |diff| %.3e; mean %.2e var %.5f kurtosis %.4f max|z| %.3f\n", n, md, m1 / n,
           m2 / n, m4 / n / (m2 / n) / (m2 / n), mx);
  }
  return 0;
}
