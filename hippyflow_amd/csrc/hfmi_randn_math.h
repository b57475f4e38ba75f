// Philox4x32-10 and the range-specific fp64 elementary functions of the probe draw (SURVEY 8a row a1:
// hp.parRandom.normal(1., Omega), activeSubspaceProjector.py:433-443, PODProjector.py:365-374, KLEProjector.py:151-160).
//
// The draw is VALU-issue bound, not HBM bound (scripts/valu_rate_probe.hip: every fp64 VALU instruction of a wave
// occupies its SIMD for ~4.5 cycles, v_mad_u64_u32 the same, v_rcp/v_rsq_f64 ~17): what counts is the NUMBER of
// instructions per normal.  Box-Muller needs -2 ln(u1) on (0, 1), one square root and (cos, sin)(2 pi u2); OCML's
// general-purpose log / sqrt / sincospi spend ~180 instructions per pair on range checks, double-double arithmetic and
// a Payne-Hanek path these arguments never reach.  Here: one range reduction each, near-minimax polynomials
// (least squares at 200 Chebyshev nodes in 60-digit arithmetic; tests/helpers/randn_math_twin.py re-derives the
// bounds on the CPU) whose every Horner step is ONE v_fma_f64 reading its coefficient from an SGPR pair (hipcc turns
// p = fma(p, z, c) into v_mov_b64 + v_fmac_f64, two issue slots), hardware 24-bit seeds + one residual step, the
// quadrant logic replaced by the double-angle identities, and FOUR normals per Philox output:
//   neg2log_scaled  rel err <= 4e-15, the seed error squared (s = (f-1)/(f+1) from v_rcp_f64 + one residual step, atanh series in s^2, 7 terms)
//   sqrt_pos        rel err <= 3e-16 (v_rsq_f64 + two residual steps)
//   rotate_turn     abs err <= 4e-16 in cos / sin (sin, cos((pi/2) f) on |f| <= 1/2, 7 terms each, then the double angle)
// The integer stream is bit-exact Philox4x32-10 (Random123 known-answer vectors, tests/test_oracle_philox.py).
//
// Element map (oracle/philox.py is the checker): column j, row group g (rows 4g .. 4g+3):
//   x = philox4x32_10(ctr = (g lo, g hi, j, stream), key = (seed lo, seed hi))
//   rows 4g, 4g+1 = sigma sqrt(-2 ln u(x0)) (cos, sin)(2 pi u(x1)),  rows 4g+2, 4g+3 likewise from (x2, x3),
//   u(x) = (x + 1/2) 2^-32  -- the 32-bit uniform lattice hipRAND / cuRAND use for their Philox normal draws
//   (|z| <= 6.76 sigma; the mass beyond that is 1.4e-11).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hfmi_rng {

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// one v_mad_u64_u32 per 32 x 32 -> 64 product, one v_bitop3_b32 per three-way xor: 4 VALU instructions a round
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int rnd = 0; rnd < 10; ++rnd) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double fma_k(double a, double b, double k) {      // a * b + k, k wave-uniform (SGPR pair)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
}

// per-kernel constants: -4 sigma^2, -2 ln 2 sigma^2 and the leading coefficient of each polynomial parked in a VGPR pair
// (an opaque move, so that the loop body does not re-materialise it before every Horner chain)
struct normal_consts { double m4s, ln2s, lead_log, lead_sin, lead_cos; };
__device__ __forceinline__ double park(double k) {
  double v;
  asm volatile("v_mov_b64 %0, %1" : "=v"(v) : "s"(k));
  return v;
}
__device__ __forceinline__ normal_consts make_consts(double sigma) {
  return normal_consts{-4.0 * sigma * sigma, -0x1.62e42fefa39efp+0 * sigma * sigma, park(0x1.2b62ca3da2e33p-4),
                       park(0x1.e3f30e54e0ae1p-25), park(0x1.f3db44f6585e0p-22)};
}

// sigma^2 * -2 ln(u), u in (0, 1)
__device__ __forceinline__ double neg2log_scaled(double u, const normal_consts& k) {
  const double m = __builtin_amdgcn_frexp_mant(u);                 // [1/2, 1)
  int e = __builtin_amdgcn_frexp_exp(u);
  const bool low = m < 0x1.6a09e667f3bcdp-1;                       // sqrt(1/2): f = m 2^low in [sqrt(1/2), sqrt(2))
  const double sc = low ? 2.0 : 1.0;
  e -= low ? 1 : 0;
  const double num = __builtin_fma(m, sc, -1.0), den = __builtin_fma(m, sc, 1.0);
  const double x = __builtin_amdgcn_rcp(den);                      // 2^-24
  double s = num * x;
  s = __builtin_fma(__builtin_fma(-s, den, num), x, s);            // (f - 1) / (f + 1) to 2^-48, |s| <= 0.1716
  const double z = s * s;
  double p = fma_k(k.lead_log, z, 0x1.39fd807189bb6p-4);
  p = fma_k(p, z, 0x1.7462bba41c635p-4);
  p = fma_k(p, z, 0x1.c71c62cbfea3fp-4);
  p = fma_k(p, z, 0x1.2492492e0644ep-3);
  p = fma_k(p, z, 0x1.9999999995219p-3);
  p = fma_k(p, z, 0x1.5555555555558p-2);
  const double at = __builtin_fma(s * z, p, s);                    // atanh(s); ln f = 2 atanh(s)
  return __builtin_fma(at, k.m4s, (double)e * k.ln2s);             // -2 sigma^2 (e ln 2 + 2 atanh s)
}

// sqrt(t), t > 0 (on the 32-bit lattice u <= 1 - 2^-33, so t >= sigma^2 2^-32; sigma = 0 never reaches the kernel)
__device__ __forceinline__ double sqrt_pos(double t) {
  const double y = __builtin_amdgcn_rsq(t);                        // 2^-24
  const double hy = 0.5 * y;
  double g = t * y;
  g = __builtin_fma(__builtin_fma(-g, g, t), hy, g);               // 2^-47
  return __builtin_fma(__builtin_fma(-g, g, t), hy, g);
}

// z0 = rad cos(2 pi u), z1 = rad sin(2 pi u) for u = (x + 1/2) 2^-32: with w = 2u, n = rint(w) in {0, 1, 2} and f = w - n,
// phi = (pi/2) f:  cos(pi w) = (-1)^n (C^2 - S^2), sin(pi w) = (-1)^n 2 S C,  C = cos phi, S = sin phi.
// n and f come straight from the bits of x: n is odd iff bit 31 and bit 30 of x differ, and 2^31 f = (int32)(x ^ odd << 31) + 1/2
// (exactly the f = w - rint(w) of the oracle).  The sign goes onto rad's sign bit, the factor 2 into its exponent field.
__device__ __forceinline__ void rotate_turn(uint32_t x, double rad, const normal_consts& k, double& z0, double& z1) {
  const uint32_t flip = __builtin_amdgcn_bitop3_b32(x, x << 1, 0x80000000u, 0x28);          // (a ^ b) & c
  const double f = __builtin_fma((double)(int32_t)(x ^ flip), 0x1.0p-31, 0x1.0p-32);
  const double y = f * f;
  double ps = fma_k(k.lead_sin, y, -0x1.e300707084987p-19);
  ps = fma_k(ps, y, 0x1.50782fd8d055dp-13);
  ps = fma_k(ps, y, -0x1.32d2cce2e4892p-8);
  ps = fma_k(ps, y, 0x1.466bc677587e9p-4);
  ps = fma_k(ps, y, -0x1.4abbce625be41p-1);
  ps = fma_k(ps, y, 0x1.921fb54442d18p+0);
  double pc = fma_k(k.lead_cos, y, -0x1.a6c9c0485eb30p-16);
  pc = fma_k(pc, y, 0x1.e1f4fb5d356afp-11);
  pc = fma_k(pc, y, -0x1.55d3c7dbfa5f9p-6);
  pc = fma_k(pc, y, 0x1.03c1f081b075cp-2);
  pc = fma_k(pc, y, -0x1.3bd3cc9be458bp+0);
  const double Cc = __builtin_fma(pc, y, 1.0);
  const double S = f * ps;
  const double c2 = __builtin_fma(Cc, Cc, -(S * S));
  const double sc = S * Cc;
  const uint32_t rhi = (uint32_t)__double2hiint(rad) ^ flip, rlo = (uint32_t)__double2loint(rad);
  z0 = __hiloint2double((int)rhi, (int)rlo) * c2;
  z1 = __hiloint2double((int)(rhi + 0x00100000u), (int)rlo) * sc;
}

// (x + 1/2) 2^-32 for a 32-bit x: 2^52 + x is the double with x in its low word, the offset takes the 2^52 out again
__device__ __forceinline__ double u32_to_unit(uint32_t x) {
  const double d = __hiloint2double(0x43300000, (int)x);
  return __builtin_fma(d, 0x1.0p-32, 0x1.0p-32 * (0.5 - 0x1.0p52));
}

// four normals from one Philox output
__device__ __forceinline__ void box_muller4(const uint32_t (&x)[4], const normal_consts& k, double (&z)[4]) {
  const double ra = sqrt_pos(neg2log_scaled(u32_to_unit(x[0]), k));
  const double rb = sqrt_pos(neg2log_scaled(u32_to_unit(x[2]), k));
  rotate_turn(x[1], ra, k, z[0], z[1]);
  rotate_turn(x[3], rb, k, z[2], z[3]);
}

}  // namespace hfmi_rng
