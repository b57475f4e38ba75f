// Cost of the primitives the one-workgroup eigensolver kernels are built from (shader cycles per iteration, one workgroup):
// wave sums on DPP moves vs ds_bpermute, lane reads, the reciprocal / rsqrt chains, LDS hand-offs through a barrier.
//   hipcc --offload-arch=gfx950 -O3 scripts/prim_probe.hip -o /tmp/prim_probe && /tmp/prim_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_get<0xB1>(v);
  v += dpp_get<0x4E>(v);
  v += dpp_get<0x141>(v);
  v += dpp_get<0x140>(v);
  return v;
}
__device__ __forceinline__ double lane_get(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v = row_sum(v);
  return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
__device__ __forceinline__ double wave_sum_shfl(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double rsqrt_fast(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y0), y0, 1.0);
  const double q = e * fma(0.375, e, 0.5);
  return fma(y0, q, y0);
}
__device__ __forceinline__ double rcp_fast(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

__global__ void k_probe(double* out, long long* cyc, int iters, int mode) {
  __shared__ double sh[4096];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  sh[tid] = tid * 1e-3 + 1.0;
  __syncthreads();
  long long c0 = clock64();
  double x = 1.0 + tid * 1e-9, y = 0.5;
  switch (mode) {
    case 0: for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9); break;                     // dependent fma
    case 1: for (int i = 0; i < iters; ++i) { x = fma(x, 1.0000001, 1e-9); y = fma(y, 0.9999999, 1e-9); } x += y; break;   // 2 chains
    case 2: for (int i = 0; i < iters; ++i) x = wave_sum_dpp(x) * 0.015; break;
    case 3: for (int i = 0; i < iters; ++i) x = wave_sum_shfl(x) * 0.015; break;
    case 4: for (int i = 0; i < iters; ++i) x = row_sum(x) * 0.06; break;
    case 5: for (int i = 0; i < iters; ++i) x = lane_get(x, i & 63) + 1e-9; break;
    case 6: for (int i = 0; i < iters; ++i) x = rsqrt_fast(x + 1.0); break;
    case 7: for (int i = 0; i < iters; ++i) x = rcp_fast(x + 1.0); break;
    case 8: for (int i = 0; i < iters; ++i) x = 1.0 / (x + 1.0); break;
    case 9: for (int i = 0; i < iters; ++i) x = sqrt(x + 1.0); break;
    case 10: for (int i = 0; i < iters; ++i) __syncthreads(); break;
    case 11:   // one wave writes, barrier, everybody reads (a hand-off)
      for (int i = 0; i < iters; ++i) {
        if (w == (i & 3)) sh[l] = x;
        __syncthreads();
        x += sh[(l + 1) & 63];
      }
      break;
    case 12:   // dependent LDS read
      { int idx = tid; for (int i = 0; i < iters; ++i) { x += sh[idx & 1023]; idx = (int)x + i; } }
      break;
    case 13:   // taken scalar branches
      for (int i = 0; i < iters; ++i) {
        if ((i & 1) == w) { asm volatile(""); x += 1.0; } else { asm volatile(""); x -= 1.0; }
      }
      break;
    case 14:   // global store, then a hand-off through LDS + barrier
      for (int i = 0; i < iters; ++i) { out[4096 + ((i * 64 + l) & 65535)] = x; sh[tid] = x; __syncthreads(); x += sh[(tid + 1) & (blockDim.x - 1)]; }
      break;
  }
  long long c1 = clock64();
  if (tid == 0) cyc[0] = c1 - c0;
  out[tid] = x;
}

int main() {
  double* out; long long* cyc;
  CK(hipMalloc(&out, (4096 + 65536 + 1024) * 8)); CK(hipMalloc(&cyc, 16));
  const char* names[] = {"dependent fma", "2 fma chains (per pair)", "wave sum, DPP + readlane", "wave sum, ds_bpermute", "16-lane sum, DPP",
                         "lane_get (2 readlane) + add", "rsqrt estimate + 1 step", "rcp estimate + 2 steps", "IEEE divide", "IEEE sqrt",
                         "barrier", "LDS hand-off + barrier", "dependent LDS read", "taken scalar branch", "global store + LDS hand-off"};
  for (int threads : {64, 256, 512, 1024})
    for (int mode = 0; mode < 15; ++mode) {
      const int iters = 4000;
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(threads), 0, 0, out, cyc, 100, mode);
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(threads), 0, 0, out, cyc, iters, mode);
      CK(hipDeviceSynchronize());
      long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
      printf("threads %4d  %-30s %8.1f cycles/iter\n", threads, names[mode], (double)h[0] / iters);
    }
  return 0;
}
