// Does it matter for HBM throughput WHICH 256-byte pieces of a column the concurrently running workgroups read?
// blocked: workgroup b owns a contiguous slice of the long axis (tsgemm_ss today); cyclic: workgroup b reads pieces
// b, b + G, b + 2G, ... so that the G workgroups in flight sweep every column front to back together.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe3 scripts/stream_probe3.hip && /tmp/stream_probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <int UNR>
__global__ __launch_bounds__(512) void k_stream(const double* __restrict__ p, int64_t ld, int ncols, int run16, int visits,
                                                int cyclic, u64* __restrict__ out) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int per_visit = ncols * run16;
  u64x2 acc = {0, 0};
  for (int v = 0; v < visits; ++v) {
    const int64_t piece = cyclic ? (int64_t)v * gridDim.x + blockIdx.x : (int64_t)blockIdx.x * visits + v;
    const double* base = p + piece * run16 * 2;
    for (int c0 = tid; c0 < per_visit; c0 += nthr * UNR) {
      u64x2 t[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int c = c0 + u * nthr;
        if (c > per_visit - 1) c = per_visit - 1;
        const int col = c / run16, off = c - col * run16;
        t[u] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(base + (int64_t)col * ld + off * 2));
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc ^= t[u];
    }
  }
  if ((acc.x ^ acc.y) == 0x1234567ull) out[blockIdx.x * nthr + tid] = acc.x;
}

int main() {
  const int64_t N = 1000000, ld = 1000000;
  const int maxcols = 176;
  double* p; u64* out;
  CK(hipMalloc(&p, (size_t)ld * maxcols * 8)); CK(hipMemset(p, 1, (size_t)ld * maxcols * 8));
  CK(hipMalloc(&out, 1 << 24));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%6s %6s %7s %7s | %8s %8s\n", "ncols", "run B", "blocks", "cyclic", "ms", "TB/s");
  for (int ncols : {146, 154, 170})
    for (int run : {256, 512})
      for (int blocks : {256, 512, 768})
        for (int cyclic = 0; cyclic < 2; ++cyclic) {
          const int run16 = run / 16;
          const int visits = (int)(N * 8 / run / blocks);
          auto launch = [&]() { hipLaunchKernelGGL((k_stream<4>), dim3(blocks), dim3(512), 0, 0, p, ld, ncols, run16, visits, cyclic, out); };
          launch();
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < 5; ++i) launch();
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
          const double bytes = (double)visits * blocks * run * ncols;
          printf("%6d %6d %7d %7d | %8.4f %8.3f\n", ncols, run, blocks, cyclic, ms, bytes / (ms * 1e-3) / 1e12);
        }
  return 0;
}
