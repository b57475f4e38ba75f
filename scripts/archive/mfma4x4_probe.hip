// Lane-layout and throughput probe of v_mfma_f64_4x4x4_4b_f64 (__builtin_amdgcn_mfma_f64_4x4x4f64):
// for every (source lane of A, source lane of B) pair with unit entries, which D lane receives the product?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p scripts/mfma4x4_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_map(int* out) {   // out[la * 64 + lb] = D lane that is nonzero (or -1), value check
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (l == 0) out[la * 64 + lb] = m ? (int)__ffsll((long long)m) - 1 + 1000 * __popcll(m) : -1;
    }
}
__global__ void k_rate(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
int main() {
  int* dm; CK(hipMalloc(&dm, 4096 * 4));
  hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, dm);
  static int h[4096]; CK(hipMemcpy(h, dm, sizeof(h), hipMemcpyDeviceToHost));
  // for each A lane list the B lanes it pairs with and the D lane
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d pairs:", la);
    for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" (B%d->D%d,n%d)", lb, h[la * 64 + lb] % 1000, h[la * 64 + lb] / 1000);
    printf("\n");
  }
  double* out; CK(hipMalloc(&out, 1024 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  hipLaunchKernelGGL(k_rate, dim3(1024), dim3(256), 0, 0, out, 100);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_rate, dim3(1024), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double insts = 1024.0 * 4 * iters * 8;   // wave instructions
  printf("4x4x4_4b: %.3f ms, %.2f TFLOP/s (512 flop per instruction), %.1f ns per instruction per SIMD-slot\n", ms,
         insts * 512 / (ms * 1e-3) / 1e12, ms * 1e6 / (insts / 1024.0));
  return 0;
}
