// Can the fp64 MFMA pipe and the fp64 VALU pipe run concurrently on one SIMD?  Half of the waves of each
// workgroup issue v_mfma_f64_16x16x4, the other half v_fma_f64; report both rates together and alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512, 2) void k_mixed(double* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || (mode == 2 && (wave & 1) == 0);   // mode 0: all MFMA, 1: all FMA, 2: half/half (wave w and w+1 share... )
  double s = 0.0;
  if (do_mfma) {
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters * 8; ++it) {   // 8x more iterations: an FMA instruction is 16x less work than an MFMA
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
  }
  if (s == 123.456) out[0] = s;
}

int main() {
  double* out; CK(hipMalloc(&out, 64));
  const int cus = 256, iters = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k_mixed, dim3(cus * 8), dim3(512), 0, 0, out, 50, mode);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_mixed, dim3(cus * 8), dim3(512), 0, 0, out, iters, mode);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)cus * 8 * 8;
    const double mfma_waves = mode == 0 ? waves : (mode == 2 ? waves / 2 : 0);
    const double fma_waves = waves - mfma_waves;
    const double mfma_fl = mfma_waves * iters * 8 * 2048.0;
    const double fma_fl = fma_waves * (double)iters * 8 * 16 * 64 * 2.0;
    printf("mode %d: %.3f ms  mfma %.1f TF + valu-fma %.1f TF = %.1f TF\n", mode, ms, mfma_fl / ms / 1e9, fma_fl / ms / 1e9, (mfma_fl + fma_fl) / ms / 1e9);
  }
  return 0;
}
