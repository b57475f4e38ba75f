// Cost of phase A of the blocked Cholesky kernel (the 16 x 16 diagonal block on one wave), in shader cycles per call, with the
// rest of the compute unit idle as it is in the kernel:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ihippyflow_amd/csrc scripts/chol_diag_probe.hip -o /tmp/chol_diag_probe && /tmp/chol_diag_probe
#include "../hippyflow_amd/csrc/hfmi_chol.hip"
#include <stdio.h>
#include <vector>
void hfmi_set_error(const char*, ...) {}   // the launcher of the included file is not used here

__global__ __launch_bounds__(64) void k_probe(const double* A, double* out, long long* cyc, int reps, int part) {
  __shared__ double scr[4 * 272], zpan[256], sref[256], spiv[256], srd[256];
  __shared__ int s_fail;
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) {
    sref[i] = 1.0;
    scr[(i >> 4) * 17 + (i & 15)] = A[i];
  }
  s_fail = 0;
  __syncthreads();
  d4 acc;
  for (int q = 0; q < 4; ++q) acc[q] = A[64 * q + lane];       // accumulator layout: register q = rows 4 q + lane / 16
  const int li = lane & 15, lk = lane >> 4;
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    asm volatile("" : "+v"(acc));
    for (int q = 0; q < 4; ++q) scr[(4 * q + lk) * 17 + li] = acc[q];
    if (part == 0) {
      chol_diag16(scr, scr + 272, zpan, sref, spiv, srd, 0, 1e-12, &s_fail, lane);
    } else if (part == 1) {
      // the input copy and the output writes without the elimination: what the rest costs
      double v[16];
      const int c = lane & 15;
      for (int q = 0; q < 16; ++q) v[q] = scr[q * 17 + c];
      for (int q = 0; q < 16; ++q) {
        if (lane < 32) scr[(lane < 16 ? 0 : 272) + q * 17 + c] = v[q];
        if (lane >= 16 && lane < 32) zpan[16 * q + c] = v[q];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[0] = (t1 - t0) / reps;
  for (int i = lane; i < 256; i += 64) out[i] = scr[(i >> 4) * 17 + (i & 15)], out[256 + i] = zpan[i];
}

int main() {
  std::vector<double> B(40 * 16), A(256);
  for (size_t i = 0; i < B.size(); ++i) B[i] = ((i * 2654435761u) % 1000) / 500.0 - 1.0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int l = 0; l < 40; ++l) s += B[l * 16 + i] * B[l * 16 + j];
      A[i * 16 + j] = s;
    }
  double *dA, *dout;
  long long* dc;
  hipMalloc(&dA, 256 * 8); hipMalloc(&dout, 512 * 8); hipMalloc(&dc, 8);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int part = 0; part < 3; ++part) {
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dout, dc, 200, part);
    long long c = 0;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%s: %lld cycles per call\n", part == 0 ? "chol_diag16 (copy in, eliminate, copy out)" : part == 1 ? "copies only" : "block to scratch only", c);
  }
  std::vector<double> out(512);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dout, dc, 1, 0);
  hipMemcpy(out.data(), dout, 512 * 8, hipMemcpyDeviceToHost);
  double err = 0;   // R^T R - A
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int l = 0; l < 16; ++l) s += out[l * 16 + i] * out[l * 16 + j];
      err = fmax(err, fabs(s - A[i * 16 + j]));
    }
  printf("max |R^T R - A| = %.2e\n", err);
  return 0;
}
