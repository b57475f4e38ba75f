// Single-workgroup latency probe: effective clock, barrier cost, LDS round trip, fp64 div/sqrt chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_probe(double* out, long long* cyc, int iters, int mode) {
  __shared__ double sh[2048];
  const int tid = threadIdx.x;
  sh[tid] = tid * 1e-3 + 1.0;
  sh[tid + 1024] = 2.0;
  __syncthreads();
  long long c0 = clock64();
  long long w0 = wall_clock64();
  double x = 1.0 + tid * 1e-9;
  if (mode == 0) {  // dependent fp64 FMA chain
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);
  } else if (mode == 1) {  // barriers only
    for (int i = 0; i < iters; ++i) __syncthreads();
  } else if (mode == 2) {  // LDS write -> barrier -> read neighbour (one "phase")
    for (int i = 0; i < iters; ++i) {
      sh[tid] = x;
      __syncthreads();
      x += sh[(tid + 1) & (blockDim.x - 1)];
      __syncthreads();
    }
  } else if (mode == 3) {  // fp64 sqrt + div chain
    for (int i = 0; i < iters; ++i) x = sqrt(x + 1.0) / (x + 0.5);
  } else if (mode == 4) {  // dependent LDS loads (pointer chase-ish)
    int idx = tid;
    for (int i = 0; i < iters; ++i) { x += sh[idx & 1023]; idx = (int)x + i; }
  } else if (mode == 5) {  // global store then barrier
    for (int i = 0; i < iters; ++i) { out[1024 + ((i * 1024 + tid) & 65535)] = x; __syncthreads(); x += 1.0; }
  }
  long long c1 = clock64();
  long long w1 = wall_clock64();
  if (tid == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
  out[tid] = x;
}

int main() {
  double* out; long long* cyc;
  CK(hipMalloc(&out, (1024 + 65536) * 8)); CK(hipMalloc(&cyc, 16));
  const char* names[] = {"fma chain", "barrier", "lds phase (2 barriers)", "sqrt+div chain", "dependent lds load", "global store + barrier"};
  int wall_rate = 0; CK(hipDeviceGetAttribute(&wall_rate, hipDeviceAttributeWallClockRate, 0));
  printf("wall clock rate %d kHz\n", wall_rate);
  for (int threads : {1024, 256, 64})
    for (int mode = 0; mode < 6; ++mode) {
      const int iters = 20000;
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(threads), 0, 0, out, cyc, 1000, mode);
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(threads), 0, 0, out, cyc, iters, mode);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
      printf("threads %4d  %-24s  %8.1f ns/iter  %8.1f shader-cycles/iter  eff clock %.0f MHz (wall ticks %lld)\n", threads, names[mode],
             ms * 1e6 / iters, (double)h[0] / iters, (double)h[0] / (ms * 1e3), h[1]);
    }
  return 0;
}
