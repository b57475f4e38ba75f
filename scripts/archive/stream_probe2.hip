// DRAM access-pattern probe for tsgemm_tn's streamed operand: `groups` sets of `ncols` column vectors (leading
// dimension ld), block b reads column set b % groups over the row slice b / groups, visiting every column `run`
// bytes at a time (tn today: run = 64).  No LDS, no MFMA.  Answers: what does HBM deliver for this pattern?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe2 scripts/stream_probe2.hip && /tmp/stream_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <int UNR>
__global__ __launch_bounds__(512) void k_stream(const double* __restrict__ p, int64_t ld, int ncols, int groups, int run16,
                                                int64_t chunk_rows, u64* __restrict__ out) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int g = blockIdx.x % groups, sl = blockIdx.x / groups;
  const int64_t r0 = (int64_t)sl * chunk_rows;
  const int visits = (int)(chunk_rows * 8 / (run16 * 16));
  const int per_visit = ncols * run16;
  const double* pg = p + (int64_t)g * ncols * ld;
  u64x2 acc = {0, 0};
  for (int v = 0; v < visits; ++v) {
    const double* base = pg + r0 + (int64_t)v * run16 * 2;
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
  const int64_t N = 200000, ld = 200000;
  const int ncols = 384, groups = 32;          // 12288 vectors of 1.6 MB = 19.7 GB
  double* p; u64* out;
  CK(hipMalloc(&p, (size_t)ld * ncols * groups * 8)); CK(hipMemset(p, 1, (size_t)ld * ncols * groups * 8));
  CK(hipMalloc(&out, 1 << 24));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%6s %7s | %8s %8s\n", "run B", "slices", "ms", "TB/s");
  for (int run : {64, 128, 256, 512, 1024, 4096})
    for (int slices : {8, 16, 32}) {
      const int run16 = run / 16;
      const int64_t rows_per_visit = run / 8;
      const int64_t chunk = (N / slices) / rows_per_visit * rows_per_visit;
      const int blocks = groups * slices;
      hipLaunchKernelGGL((k_stream<4>), dim3(blocks), dim3(512), 0, 0, p, ld, ncols, groups, run16, chunk, out);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<4>), dim3(blocks), dim3(512), 0, 0, p, ld, ncols, groups, run16, chunk, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      const double bytes = (double)chunk * slices * ncols * groups * 8;
      printf("%6d %7d | %8.4f %8.3f\n", run, slices, ms, bytes / (ms * 1e-3) / 1e12);
    }
  return 0;
}
