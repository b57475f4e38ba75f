// What does ONE device-wide dependency cost on this GPU?  The numbers behind DESIGN.md section 8 item 1 (why the whole-GPU eigensolver stays
// one-stage), measured here rather than quoted: (1) a dependent kernel boundary between trivial kernels; (2) the same with the successor
// reading what the predecessor wrote from other compute units (one fresh-data round trip on the chain); (3) a flag hand-off between two
// workgroups inside one launch (what a pipelined bulge chase pays per task); (4) a grid-wide barrier inside one launch (what a persistent
// panel kernel pays per column).  Every spin is bounded; a time-out is reported, never waited out.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hippyflow_amd/build/sync_price_probe scripts/sync_price_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_trivial(double* buf, int it) {
  if (threadIdx.x == 0) buf[blockIdx.x] = (double)it;
}
// every workgroup sums what ALL workgroups of the previous launch wrote (256 values: fresh data from every XCD), then writes its own
__global__ __launch_bounds__(256) void k_chain(const double* __restrict__ in, double* __restrict__ out, int nb) {
  __shared__ double s[4];
  double v = threadIdx.x < nb ? in[threadIdx.x] : 0.0;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]) + 1.0;
}
// two workgroups, a flag each way: hops round trips of {8-byte payload + flag} through agent-scope relaxed atomics
__global__ __launch_bounds__(64) void k_pingpong(unsigned long long* flags, int hops, int* fail, int partner_stride) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_stride ? 1 : -1);
  if (me < 0 || threadIdx.x != 0) return;
  unsigned long long* mine = flags + 32 * me;
  unsigned long long* other = flags + 32 * (1 - me);
  for (int h = 1; h <= hops; ++h) {
    if (me == 0) __hip_atomic_store(other, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)h) {
      if (++spins > 20000000) { *fail = 1; return; }
    }
    if (me == 1) __hip_atomic_store(other, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// grid barrier: one monotonic counter, lane 0 of every workgroup arrives (release) and polls (relaxed) until all have, then acquires
__global__ __launch_bounds__(256) void k_gridbar(unsigned int* counter, int iters, int* fail, double* data) {
  for (int it = 1; it <= iters; ++it) {
    if (threadIdx.x == 0) data[blockIdx.x] = (double)it;             // something to publish
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int target = (unsigned int)it * gridDim.x;
      long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000) { *fail = 1; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (*fail) return;
  }
}

int main() {
  double *a, *b; unsigned long long* flags; unsigned int* counter; int* fail;
  CK(hipMalloc(&a, 1 << 16)); CK(hipMalloc(&b, 1 << 16)); CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&counter, 256)); CK(hipMalloc(&fail, 64));
  CK(hipMemset(a, 0, 1 << 16)); CK(hipMemset(b, 0, 1 << 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  const int N = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_trivial, dim3(256), dim3(256), 0, 0, a, i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("(1) dependent kernel boundary, trivial 256-workgroup kernels:                 %.2f us per launch\n", ms * 1e3 / N);
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, 256);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("(2) boundary + every workgroup reads the 256 values its predecessor wrote:      %.2f us per launch\n", ms * 1e3 / N);
  int hfail = 0;
  for (int stride : {1, 8, 9}) {                 // partner on another CU of the same XCD (block 8) / on another XCD (blocks 1, 9)
    const int hops = 20000;
    CK(hipMemset(flags, 0, 4096)); CK(hipMemset(fail, 0, 64));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_pingpong, dim3(stride + 1), dim3(64), 0, 0, flags, hops, fail, stride);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost));
    printf("(3) flag hand-off between workgroups 0 and %d (idle chip), one way:              %.2f us%s\n", stride, ms * 1e3 / (2.0 * hops), hfail ? "  [TIMED OUT]" : "");
  }
  for (int wgs : {256, 512}) {
    const int iters = 2000;
    CK(hipMemset(counter, 0, 256)); CK(hipMemset(fail, 0, 64));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gridbar, dim3(wgs), dim3(256), 0, 0, counter, iters, fail, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost));
    printf("(4) grid barrier inside one launch, %d workgroups (counter + release / acquire): %.2f us per barrier%s\n", wgs, ms * 1e3 / iters, hfail ? "  [TIMED OUT]" : "");
  }
  return 0;
}
