// Issue cost of the VALU instructions the fp64 elementwise kernels are made of (gfx950): each kernel runs 8 independent
// chains of one instruction, 8 waves per SIMD, and reports SIMD cycles per wave-instruction at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hippyflow_amd/build/valu_rate_probe scripts/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 512, CHAINS = 8;

#define KERNEL64(name, ASM)                                                                         \
  __global__ __launch_bounds__(256) void name(double* out, double a, double b) {                    \
    double v[CHAINS];                                                                               \
    for (int i = 0; i < CHAINS; ++i) v[i] = a + i + threadIdx.x;                                    \
    for (int it = 0; it < ITER; ++it) {                                                             \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(a), "v"(b)); \
    }                                                                                               \
    double s = 0; for (int i = 0; i < CHAINS; ++i) s += v[i];                                       \
    if (s == 12345.678) out[threadIdx.x] = s;                                                       \
  }
#define KERNEL32(name, ASM)                                                                         \
  __global__ __launch_bounds__(256) void name(double* out, double a, double b) {                    \
    unsigned v[CHAINS]; unsigned ua = (unsigned)a + 3, ub = (unsigned)b + 5;                        \
    for (int i = 0; i < CHAINS; ++i) v[i] = ua + i + threadIdx.x;                                   \
    for (int it = 0; it < ITER; ++it) {                                                             \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(ua), "v"(ub)); \
    }                                                                                               \
    unsigned s = 0; for (int i = 0; i < CHAINS; ++i) s += v[i];                                     \
    if (s == 12345678u) out[threadIdx.x] = s;                                                       \
  }
// 64-bit result from 32-bit sources
#define KERNELMAD(name, ASM)                                                                        \
  __global__ __launch_bounds__(256) void name(double* out, double a, double b) {                    \
    unsigned long long v[CHAINS]; unsigned ua = (unsigned)a + 3, ub = (unsigned)b + 5;              \
    for (int i = 0; i < CHAINS; ++i) v[i] = ua + i + threadIdx.x;                                   \
    for (int it = 0; it < ITER; ++it) {                                                             \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(ua), "v"(ub) : "vcc"); \
    }                                                                                               \
    unsigned long long s = 0; for (int i = 0; i < CHAINS; ++i) s += v[i];                           \
    if (s == 12345678ull) out[threadIdx.x] = (double)s;                                             \
  }

KERNEL64(k_fma64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_mul64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_add64, "v_add_f64 %0, %0, %1")
KERNEL64(k_max64, "v_max_f64 %0, %0, %1")
KERNEL64(k_rcp64, "v_rcp_f64 %0, %0")
KERNEL64(k_rsq64, "v_rsq_f64 %0, %0")
KERNEL64(k_sqrt64, "v_sqrt_f64 %0, %0")
KERNEL64(k_rndne64, "v_rndne_f64 %0, %0")
KERNEL64(k_frexpm64, "v_frexp_mant_f64 %0, %0")
KERNEL64(k_ldexp64, "v_ldexp_f64 %0, %0, 1")
KERNEL64(k_mov64, "v_mov_b64 %0, %1")
KERNEL32(k_xor32, "v_xor_b32 %0, %0, %1")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL32(k_cnd32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_fma32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_log32, "v_log_f32 %0, %0")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 11")
__global__ __launch_bounds__(256) void k_mad64(double* out, double a, double b) {
  unsigned long long v[CHAINS]; unsigned w[CHAINS]; unsigned ua = (unsigned)a + 3;
  for (int i = 0; i < CHAINS; ++i) w[i] = ua + i + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(v[i]) : "v"(w[i]), "v"(ua) : "vcc");
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) w[i] = (unsigned)(v[i] >> 32);
  }
  unsigned long long s = 0; for (int i = 0; i < CHAINS; ++i) s += v[i];
  if (s == 12345678ull) out[threadIdx.x] = (double)s;
}
__global__ __launch_bounds__(256) void k_cvtu(double* out, double a, double b) {
  double v[CHAINS]; unsigned w[CHAINS]; unsigned ua = (unsigned)a + 3;
  for (int i = 0; i < CHAINS; ++i) w[i] = ua + i + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(v[i]) : "v"(w[i]));
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) w[i] = (unsigned)(__builtin_bit_cast(unsigned long long, v[i]) >> 32);
  }
  double s = 0; for (int i = 0; i < CHAINS; ++i) s += v[i];
  if (s == 12345.678) out[threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kern, double* out, double clock_ghz) {
  const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.25, 0.75);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.25, 0.75);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double per_simd = 5.0 * blocks * 4 / 1024.0 * ITER * CHAINS;   // wave-instructions per SIMD
  printf("%-18s %8.3f ms   %6.2f ns/1000 instr/SIMD   %5.2f cycles per wave-instruction at %.2f GHz\n", name, ms / 5, ms * 1e6 / per_simd * 1e3 / 1e3,
         ms * 1e-3 * clock_ghz * 1e9 / per_simd, clock_ghz);
}

int main() {
  double* out; CK(hipMalloc(&out, 4096));
  int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double ghz = khz / 1e6;
  printf("device clock attribute %.3f GHz (cycle counts assume it is sustained)\n", ghz);
#define RUN(k) run(#k, k, out, ghz)
  RUN(k_fma32); RUN(k_fma64); RUN(k_mul64); RUN(k_add64); RUN(k_max64); RUN(k_mov64); RUN(k_cvtu); RUN(k_rndne64); RUN(k_frexpm64); RUN(k_ldexp64);
  RUN(k_rcp64); RUN(k_rsq64); RUN(k_sqrt64); RUN(k_log32);
  RUN(k_xor32); RUN(k_bitop3); RUN(k_cnd32); RUN(k_alignbit); RUN(k_mullo); RUN(k_mulhi); RUN(k_mad64);
  return 0;
}
