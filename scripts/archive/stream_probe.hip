// Access-pattern probe for the column-block layout: how fast can HBM deliver `ncols` column streams when every
// workgroup owns a slice of the long axis and visits each column `run` bytes at a time?  No LDS, no MFMA: each thread
// xors its 16-byte loads into a register.  Answers whether tsgemm_ss's ~4.5 TB/s is the kernel or the pattern.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe scripts/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// each block: rows [b*chunk, (b+1)*chunk); per visit: all ncols columns x `run` bytes.  UNR loads in flight per thread.
template <int UNR, bool NT>
__global__ __launch_bounds__(512) void k_stream(const double* __restrict__ p, int64_t ld, int ncols, int run16,
                                                int64_t chunk_rows, u64* __restrict__ out) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
  const int visits = (int)(chunk_rows * 8 / (run16 * 16));
  const int per_visit = ncols * run16;  // 16-byte chunks per visit
  u64x2 acc = {0, 0};
  for (int v = 0; v < visits; ++v) {
    const double* base = p + r0 + (int64_t)v * run16 * 2;
    for (int c0 = tid; c0 < per_visit; c0 += nthr * UNR) {
      u64x2 t[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int c = c0 + u * nthr;
        if (c > per_visit - 1) c = per_visit - 1;
        const int col = c / run16, off = c - col * run16;
        const u64x2* src = reinterpret_cast<const u64x2*>(base + (int64_t)col * ld + off * 2);
        t[u] = NT ? __builtin_nontemporal_load(src) : *src;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc ^= t[u];
    }
  }
  if ((acc.x ^ acc.y) == 0x1234567ull) out[blockIdx.x * nthr + tid] = acc.x;
}

int main() {
  const int64_t N = 1000000, ld = 1000000 + 0;  // ld multiple of 32
  const int maxcols = 160;
  double* p; u64* out;
  CK(hipMalloc(&p, (size_t)ld * maxcols * 8)); CK(hipMemset(p, 1, (size_t)ld * maxcols * 8));
  CK(hipMalloc(&out, 1 << 24));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%6s %6s %7s %7s %5s %3s | %8s %8s\n", "ncols", "run B", "blocks", "threads", "unr", "nt", "ms", "TB/s");
  for (int ncols : {146, 64})
    for (int run : {128, 256, 512, 1024, 4096})
      for (int blocks : {256, 512, 1024, 2048})
        for (int threads : {256, 512})
          for (int nt = 0; nt < 2; ++nt) {
            const int run16 = run / 16;
            // rows per block: multiple of the run length
            int64_t rows_per_visit = run / 8;
            int64_t chunk = (N / blocks) / rows_per_visit * rows_per_visit;
            if (chunk < rows_per_visit) continue;
            auto launch = [&]() {
              if (nt) hipLaunchKernelGGL((k_stream<4, true>), dim3(blocks), dim3(threads), 0, 0, p, ld, ncols, run16, chunk, out);
              else hipLaunchKernelGGL((k_stream<4, false>), dim3(blocks), dim3(threads), 0, 0, p, ld, ncols, run16, chunk, out);
            };
            launch();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double bytes = (double)chunk * blocks * ncols * 8;
            printf("%6d %6d %7d %7d %5d %3d | %8.4f %8.3f\n", ncols, run, blocks, threads, 4, nt, ms, bytes / (ms * 1e-3) / 1e12);
          }
  // reference: one contiguous stream of the same total size, grid-stride
  return 0;
}
