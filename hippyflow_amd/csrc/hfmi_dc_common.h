// Device helpers shared by the two eigensolver translation units (hfmi_eig_dc.hip: one workgroup, k <= 256;
// hfmi_eig_blocked.hip: whole GPU, 256 < n <= 4096): DPP cross-lane reductions, reciprocal / rsqrt from the hardware
// estimates, and the secular-equation solver of the divide-and-conquer merges (LAPACK dlaed4's role).
#pragma once
#include <math.h>

#include "hfmi_gemm_common.h"

namespace {
constexpr double DC_EPS = 2.220446049250313e-16;
// Cross-lane sums on DPP moves (two v_mov_b32_dpp per double and stage) instead of ds_bpermute round trips through the LDS
// crossbar: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.  After a
// stage both partners hold a + b and b + a, i.e. identical bits, so every lane of the group ends with the same value.
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_get(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int G>   // G = 1, 2, 4, 8, 16 consecutive lanes, or the whole wave (64)
__device__ __forceinline__ double group_sum(double v) {
  if (G >= 2) v += dpp_get<0xB1>(v);
  if (G >= 4) v += dpp_get<0x4E>(v);
  if (G >= 8) v += dpp_get<0x141>(v);
  if (G >= 16) v += dpp_get<0x140>(v);
  if (G == 64) v = (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
  return v;
}
template <int G>
__device__ __forceinline__ double group_prod(double v) {
  if (G >= 2) v *= dpp_get<0xB1>(v);
  if (G >= 4) v *= dpp_get<0x4E>(v);
  if (G >= 8) v *= dpp_get<0x141>(v);
  if (G >= 16) v *= dpp_get<0x140>(v);
  if (G == 64) v = (lane_get(v, 0) * lane_get(v, 16)) * (lane_get(v, 32) * lane_get(v, 48));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {   // all 64 lanes
  v = group_sum<16>(v);
  return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
// 1 / sqrt(x) and 1 / x from the hardware estimates plus Newton steps: short dependent chains instead of the IEEE sequences
// (the reflector needs tau and the scale to a few ulp, not correctly rounded)
__device__ __forceinline__ double dc_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y0), y0, 1.0);
  const double q = e * fma(0.375, e, 0.5);
  return fma(y0, q, y0);
}
__device__ __forceinline__ double dc_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

// ------------------------------------------------------------------------------------------------ secular equation
template <int G>
__device__ __forceinline__ void secular_eval(const double* __restrict__ dl, const double* __restrict__ wv, int K, int sub,
                                             double dorg, double tau, double rho, int i0, double& f, double& dpsi, double& dphi,
                                             double& err) {
  double psi = 0.0, phi = 0.0, dps = 0.0, dph = 0.0;
  int j = sub;
  for (; j <= i0; j += G) {                  // poles to the left of the root, then those to the right: no selects
    const double dj = (dl[j] - dorg) - tau;
    const double inv = dc_rcp(dj);           // a couple of ulp: inside the 8 eps (|psi| + |phi|) the stopping test allows
    const double z = wv[j];
    const double t = z * z * inv;
    psi += t;
    dps = fma(t, inv, dps);
  }
  for (; j < K; j += G) {
    const double dj = (dl[j] - dorg) - tau;
    const double inv = dc_rcp(dj);
    const double z = wv[j];
    const double t = z * z * inv;
    phi += t;
    dph = fma(t, inv, dph);
  }
  psi = rho * group_sum<G>(psi);
  phi = rho * group_sum<G>(phi);
  dpsi = rho * group_sum<G>(dps);
  dphi = rho * group_sum<G>(dph);
  f = 1.0 + psi + phi;
  err = 8.0 * (fabs(psi) + fabs(phi)) + 1.0 + fabs(tau) * (dpsi + dphi);
}

// root i of 1 + rho sum_j w_j^2 / (dl_j - lam) = 0: returns the origin pole and tau = lam - dl[origin].
// One round = one evaluation + the "middle way" step; the step is written without data-dependent branches (the lanes of a wave
// work on different roots: a branch taken by one root is paid by all), with the hardware reciprocal / rsqrt estimates plus
// Newton steps instead of the IEEE sequences.  A round is a dependent chain of ~100 fp64 operations either way: that chain
// times the rounds a root needs (4 on average, 8 at most) times the levels is the floor of this kernel.
template <int G>
__device__ __forceinline__ bool secular_root(const double* __restrict__ dl, const double* __restrict__ wv, int K, int i, int sub,
                                             double rho, int& org_out, double& tau_out, int& evals) {
  evals = 0;
  if (K == 1) {
    org_out = 0;
    tau_out = rho * wv[0] * wv[0];
    return true;
  }
  const bool last = (i == K - 1);
  const int i0 = last ? K - 2 : i, i1 = i0 + 1;
  int org;
  double lo, hi, f, dpsi, dphi, err;
  double tau;
  bool have_eval = false;
  if (last) {
    org = K - 1;
    double s2 = 0.0;
    for (int j = sub; j < K; j += G) s2 = fma(wv[j], wv[j], s2);
    lo = 0.0;
    hi = rho * group_sum<G>(s2);
    tau = 0.5 * hi;
  } else {
    const double gap = dl[i + 1] - dl[i];
    tau = 0.5 * gap;
    secular_eval<G>(dl, wv, K, sub, dl[i], tau, rho, i0, f, dpsi, dphi, err);
    have_eval = true;        // the same point in either shifted variable: value and slopes carry over
    if (f >= 0.0) {          // root in the lower half: origin = left pole
      org = i;
      lo = 0.0;
      hi = tau;
    } else {                 // origin = right pole, tau = -gap / 2 there
      org = i + 1;
      lo = -tau;
      hi = 0.0;
      tau = -tau;
    }
  }
  const double dorg = dl[org];
  const double d0 = dl[i0] - dorg, d1 = dl[i1] - dorg;
  bool converged = false;
  for (int it = 0; it < 100; ++it) {
    if (!have_eval) secular_eval<G>(dl, wv, K, sub, dorg, tau, rho, i0, f, dpsi, dphi, err);
    have_eval = false;
    ++evals;
    if (fabs(f) <= DC_EPS * err) {
      converged = true;
      break;
    }
    lo = (f < 0.0) ? tau : lo;
    hi = (f < 0.0) ? hi : tau;
    if (hi - lo <= 2.0 * DC_EPS * fmax(fabs(lo), fabs(hi))) {
      tau = 0.5 * (lo + hi);
      converged = true;
      break;
    }
    // "middle way": the two neighbouring poles kept exact, the rest matched in value and slope; c eta^2 - a eta + b = 0
    const double D0 = d0 - tau, D1 = d1 - tau;
    const double dw = dpsi + dphi;
    const double dd = D0 * D1;
    const double a = fma(D0 + D1, f, -dd * dw);
    const double b = dd * f;
    const double c = f - fma(D0, dpsi, D1 * dphi);
    const double disc = fma(a, a, -4.0 * b * c);
    const bool dok = disc > 0.0 && disc < 1e300;
    const double dsafe = dok ? disc : 1.0;
    const double sq = dsafe * dc_rsqrt(dsafe);
    const double q = 0.5 * (a + copysign(sq, a));
    const double x1 = fma(q, dc_rcp(c), tau);             // q / c
    const double x2 = fma(b, dc_rcp(q), tau);             // b / q
    const double xn = fma(-f, dc_rcp(dw), tau);           // Newton: f is increasing, the step always points at the root
    const bool ok1 = dok && x1 > lo && x1 < hi;           // (comparisons are false for NaN / inf)
    const bool ok2 = dok && x2 > lo && x2 < hi;
    const bool okn = xn > lo && xn < hi;
    const bool take2 = ok2 && (!ok1 || fabs(x2 - tau) < fabs(x1 - tau));
    double next = take2 ? x2 : x1;
    const bool found = ok1 || ok2;
    next = found ? next : xn;
    if (!(found || okn)) {                     // bisection, geometric where the bracket spans decades (rare)
      if (lo > 0.0 && hi > 4.0 * lo) next = sqrt(lo * hi);
      else if (hi < 0.0 && lo < 4.0 * hi) next = -sqrt(lo * hi);
      else if (lo == 0.0) next = hi * 0.0625;
      else if (hi == 0.0) next = lo * 0.0625;
      else next = 0.5 * (lo + hi);
    }
    tau = next;
  }
  org_out = org;
  tau_out = tau;
  return converged;
}
}  // namespace
