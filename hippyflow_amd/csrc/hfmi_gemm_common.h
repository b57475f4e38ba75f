// Shared by the two MFMA translation units (hfmi_gemm.hip: tsgemm_tn, hfmi_gemm_nn.hip: tsgemm_nn).
#pragma once
#include "hfmi_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define MFMA_F64_4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int64_t round_up_dev(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// XCD-aware block id remap (8 XCDs, block b runs on XCD b % 8): consecutive logical ids land on the
// same XCD so that workgroups sharing the LDS-staged operand also share an L2.  Bijective for any total.
__device__ __forceinline__ int xcd_remap(int lin, int total) {
  const int xcd = lin & 7, slot = lin >> 3;
  const int fl = total >> 3, rem = total & 7;
  return xcd * fl + (xcd < rem ? xcd : rem) + slot;
}

// tuning knobs of tsgemm_nn (set through hfmi_tuning_set, hfmi_gemm.hip)
int nn_tuning_set(const char* key, int value);   // 1 = key handled
