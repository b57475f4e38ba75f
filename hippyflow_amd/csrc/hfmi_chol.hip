// Blocked Cholesky factorisation with the triangular inverse, k <= 256, on the fp64 MFMA of ONE compute unit (round 3).
//
// The column-at-a-time kernels of hfmi_small.hip (k_chol_tile) spend a barrier, an LDS hand-off and a reciprocal square
// root on each of the k columns of the factorisation and again on each of the k rows of the inverse: 1.6 k cycles per
// column, 0.2 ms at k = 138 -- twice per solve.  Here the matrix is cut into 16 x 16 blocks that live in the MFMA's
// accumulator layout in registers, and a step handles sixteen columns:
//
//   for K = 0 .. nb-1:   (nb = ceil(k / 16))
//     A  one wave factors the diagonal block A_KK = R_KK^T R_KK by Gaussian elimination on [A_KK | I], lane l holding column
//        l mod 16 of both halves: sixteen steps whose broadcasts are 64-bit DPP operands of the accumulate itself
//        (v_fmac_f64_dpp row_newbcast), no barrier; the augmented half ends as R_KK^-T.
//     B  block row K:   R_KJ = R_KK^-T A_KJ (J > K)   and   Z_KJ = R_KK^-T W_KJ (J < K),  Z_KK = R_KK^-T
//     C  everything below:   A_IJ -= R_KI^T R_KJ (K < I <= J)   and   W_IJ -= R_KI^T Z_KJ (J <= K < I)
//
//   W is the identity carried through the same elimination: Z = R^-T comes out row by row, and X = R^-1 = Z^T is written
//   transposed.  Block (J, I) of A (J < I) dies at step J, exactly when W_IJ is born: they share one register slot, so the
//   nb (nb + 1) / 2 slots of the upper triangle hold everything (17 slots of four doubles per wave at k = 256).
//
// Every product has the form X^T Y with X and Y in the accumulator layout -- lane (li, lk) holds rows 4 s + lk, column li
// in register s -- and that is exactly what v_mfma_f64_16x16x4 wants as its A operand (X^T: A[li][4 s + lk]) and B operand
// (Y: B[4 s + lk][li]) for k-step s: blocks go from accumulator registers (or a row-major 16 x 16 copy in LDS) straight
// into the next MFMA, no shuffles.  Two barriers per block step; the published block row is double-buffered.
//
// Semantics (status words, shift-and-retry on breakdown, first-order polish for an orthonormal input, running product of the
// factors) are those of k_chol_tile.  Kernel time 80 -> 25 us at k = 74, 205 -> 55 us at k = 138, 730 -> 170 us at k = 256; per
// block step about 4.8 k cycles for the diagonal block (scripts/chol_diag_probe.hip), 2.7 k for the block row, 2.3 k for the
// trailing update at k = 138 (HFMI_DEBUG_TIMING=1 prints the three sums).
#include "hfmi_internal.h"

#include <stdlib.h>
#include <string.h>

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define EPS_D 2.220446049250313e-16

namespace {

__device__ __forceinline__ double cm_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y0), y0, 1.0);
  const double q = e * fma(0.375, e, 0.5);
  return fma(y0, q, y0);
}
__device__ __forceinline__ double cm_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double cm_readlane(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cm_block_sum(double v, double* scratch) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) s += scratch[w];
  return s;
}
__device__ __forceinline__ double cm_block_min(double v, double* scratch) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = scratch[0];
  const int nw = blockDim.x >> 6;
  for (int w = 1; w < nw; ++w) s = fmin(s, scratch[w]);
  return s;
}

// a wave-uniform value the optimiser may not see through: keeps per-slot index arithmetic inside the loop it belongs to instead
// of hoisted (and live in registers) across the whole factorisation
__device__ __forceinline__ int cm_opaque(int x) {
  asm volatile("" : "+s"(x));
  return x;
}

// Barrier of the block loop: waits for this wave's LDS traffic only.  __syncthreads() is also a workgroup-scope fence for GLOBAL
// memory (s_waitcnt vmcnt(0)): with it every step would wait for the acknowledgement of the rows of R and R^-1 it has just
// stored, which nobody reads before the kernel ends (or before the __syncthreads() in front of the running product).
__device__ __forceinline__ void cm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int CM_SCR = 4 * 16 * 17;   // padded 16 x 16 blocks of scratch per wave: input / R_KK, Z_KK, two dump blocks

// 64-bit DPP: lane l gets x of lane I of its own row of sixteen (row_newbcast, the one DPP mode the double-precision ALU has) --
// as a plain copy and fused into the accumulate acc += bcast_I(src) * mul.  The compiler neither knows these instructions
// (v_readlane into SGPRs is what it offers: two per double and ~20 cycles each, 60 % of the elimination's time) nor their
// hazard (two wait states between a VALU write of src and its DPP read): the copy, which is always the first DPP reader of a
// freshly written row, carries the s_nop.
template <int I>
__device__ __forceinline__ double cm_bcast(double x) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(I));
  return r;
}
template <int I>
__device__ __forceinline__ void cm_fmac_bcast(double& acc, double src, double mul) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(I));
}
template <int J, int I>
struct cm_elim_rows {   // rows I .. 15 of step J: v[i] -= U[J][i] (v[J] / pivot), the same for the augmented half
  static __device__ __forceinline__ void run(double (&v)[16], double (&w)[16], double tv, double tw) {
    if constexpr (I < 16) {
      cm_fmac_bcast<I>(v[I], v[J], tv);
      cm_fmac_bcast<I>(w[I], v[J], tw);
      cm_elim_rows<J, I + 1>::run(v, w, tv, tw);
    }
  }
};
template <int J>
struct cm_elim_steps {
  static __device__ __forceinline__ void run(double (&v)[16], double (&w)[16], double piv, double inv, double& mypiv, int c) {
    if constexpr (J < 16) {
      if (c == J) mypiv = piv;                       // lane c keeps pivot c: checks, square roots and scaling are done once, after
      const double tv = -(v[J] * inv), tw = -(w[J] * inv);
      double piv_n = 1.0, inv_n = 1.0;
      if constexpr (J + 1 < 16) {
        // row J + 1 first, and the reciprocal of ITS pivot started right away: the dependent chain (broadcast, v_rcp_f64, two
        // Newton steps) runs under the other updates
        cm_fmac_bcast<J + 1>(v[J + 1], v[J], tv);
        cm_fmac_bcast<J + 1>(w[J + 1], v[J], tw);
        piv_n = cm_bcast<J + 1>(v[J + 1]);
        inv_n = cm_rcp(piv_n);
      }
      cm_elim_rows<J, J + 2>::run(v, w, tv, tw);
      cm_elim_steps<J + 1>::run(v, w, piv_n, inv_n, mypiv, c);
    }
  }
};
template <int J>
struct cm_scale_out {   // row J of R = diag(U)^-1/2 U and of Z = diag(U)^-1/2 L^-1 to their blocks
  static __device__ __forceinline__ void run(const double (&v)[16], const double (&w)[16], double rsv, bool mat, int c, double* dst,
                                             double* dsz) {
    if constexpr (J < 16) {
      const double rs = cm_bcast<J>(rsv);
      const double x = (mat ? ((J <= c) ? v[J] : 0.0) : ((J >= c) ? w[J] : 0.0)) * rs;
      dst[J * 17] = x;
      dsz[16 * J] = x;
      cm_scale_out<J + 1>::run(v, w, rsv, mat, c, dst, dsz);
    }
  }
};

// Phase A.  scr: the diagonal block, row-major with stride 17 (written by this wave).  Gaussian elimination on [A_KK | I], lane
// l holds column l mod 16 of both halves (the four rows of sixteen lanes do the same work: the broadcasts stay inside a row).
// Outputs, all in LDS: R_KK over the input (scr, stride 17), Z = R_KK^-T in scz (stride 17) and in zpan (row-major 16 x 16, the
// published block Z_KK; scz + 272 .. scz + 3 * 272 is a dump area), the pivots in spiv, diag(R) in srd; *s_fail on breakdown.  The global copies are written by the
// caller after the barrier, off the critical path.
__device__ __forceinline__ void chol_diag16(double* scr, double* scz, double* zpan, const double* sref, double* spiv, double* srd,
                                            int K, double ptol, int* s_fail, int lane) {
  asm volatile("" : "+v"(lane));                 // lane masks and addresses are recomputed per call, not kept live across calls
  const int c = lane & 15;
  double v[16], w[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    v[r] = scr[r * 17 + c];
    w[r] = (r == c) ? 1.0 : 0.0;
  }
  const int g0 = 16 * K;
  double mypiv = 1.0;
  const double piv0 = cm_bcast<0>(v[0]);
  cm_elim_steps<0>::run(v, w, piv0, cm_rcp(piv0), mypiv, c);
  const double ref = sref[g0 + c];
  const bool bad = !(mypiv > ptol * ref) || !(ref > 0.0);
  const double rsv = cm_rsqrt(mypiv);
  // lanes 0..15: column c of R_KK (zero below the diagonal) over the input; lanes 16..31: column c of Z (zero above it) to scz and
  // to the published block; the other lanes write the same values to a dump block, so that no store sits behind a branch
  // (predicated, the 48 stores cost more than the elimination)
  const int row = lane >> 4;
  const bool mat = row == 0, aug = row == 1;
  double* dst = (mat ? scr : aug ? scz : scz + 272) + c;
  double* dsz = (aug ? zpan : scz + 2 * 272) + c;
  cm_scale_out<0>::run(v, w, rsv, mat, c, dst, dsz);
  spiv[g0 + c] = mypiv;                             // (the four rows of lanes write the same words)
  srd[g0 + c] = mypiv * rsv;                        // R_cc, folded into the running diagonal after a clean run
  if (bad) *s_fail = 1;
}

template <int NW, int SLOTS>
__global__ __launch_bounds__(64 * NW) void k_chol_mfma(const double* __restrict__ G, int ldg, int k, double* __restrict__ Rout,
                                                       double* __restrict__ Rinv, double* __restrict__ Rtot,
                                                       double* __restrict__ Rtmp, int ldo, int rtot_mode, int full_r,
                                                       double shift_rel, double pivot_tol, double* __restrict__ colnorm0,
                                                       double* __restrict__ rdiag, hfmi_status_words* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 64 * NW;
  double* red = reinterpret_cast<double*>(smem);   // 32 doubles of reduction scratch
  double* diag0 = red + 32;                        // original diagonal (1 on the padding)
  double* invd = diag0 + 256;                      // diag0^-1/2
  double* sref = invd + 256;                       // diag0 + shift
  double* spiv = sref + 256;                       // pivots
  double* srd = spiv + 256;                        // diag(R)
  double* scratch = srd + 256;                     // four padded 16 x 16 blocks per wave
  double* panel = scratch + NW * CM_SCR;           // 2 x nb published blocks
  __shared__ int s_fail;

  const int tid = threadIdx.x, lane0 = tid & 63, lane = lane0, li = lane & 15, lk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = (k + 15) >> 4, nslots = nb * (nb + 1) / 2;
  double* myscr = scratch + wave * CM_SCR;

  long long tk0 = clock64(), tk1, tk2, tk3, tk4;
  // slot t of this wave: block pair {a <= b}, dealt cyclically in the order b-major so that every wave's share dies evenly
  int sa[SLOTS], sb[SLOTS];
  bool sv[SLOTS];
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const int q = wave + NW * t;
    sv[t] = q < nslots;
    const int qq = sv[t] ? q : 0;
    int b = 0;
    while ((b + 1) * (b + 2) / 2 <= qq) ++b;               // scalar loop on a wave-uniform value: a and b stay in SGPRs
    sb[t] = __builtin_amdgcn_readfirstlane(b);
    sa[t] = __builtin_amdgcn_readfirstlane(qq - b * (b + 1) / 2);
  }

  for (int i = tid; i < 256; i += NT) {
    const double d = i < k ? G[(size_t)i * ldg + i] : 1.0;
    diag0[i] = d;
    invd[i] = i < k ? (d > 0.0 ? cm_rsqrt(d) : 0.0) : 1.0;
    if (i < k && rtot_mode == 1) colnorm0[i] = sqrt(fmax(d, 0.0));   // norms of the ORIGINAL columns (first pass)
  }
  __syncthreads();

  d4 acc[SLOTS];
  // block (a, b) in the accumulator layout: register s of lane (li, lk) is the element (16 a + 4 s + lk, 16 b + li) of the
  // symmetrised matrix; identity on the padding
  auto load_blocks = [&](double shift, bool measure, double& dev, double& tr) {
    int ln = lane0;
    asm volatile("" : "+v"(ln));                    // (the reload of a shifted retry must not park its addresses in registers)
    const int li = ln & 15, lk = ln >> 4;
    // All loads of a group of slots are issued before the first one is used, unconditionally (clamped indices; a slot beyond
    // the last block pair loads block (0, 0) and is never looked at): the Gram matrix was written by another kernel and is a
    // 3 k-cycle round trip away, which was paid once per ELEMENT behind per-element branches and once per slot without groups.
    constexpr int GS = SLOTS < 6 ? SLOTS : 6;
#pragma unroll
    for (int t0 = 0; t0 < SLOTS; t0 += GS) {
      double g1[GS][4], g2[GS][4];
      int ga[GS], gb[GS];
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        if (t0 + u >= SLOTS) continue;
        ga[u] = cm_opaque(sa[t0 + u]);
        gb[u] = cm_opaque(sb[t0 + u]);
      }
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        if (t0 + u >= SLOTS) continue;
        const int c = 16 * gb[u] + li;
        const int cc = c < k ? c : k - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = 16 * ga[u] + 4 * s + lk;
          const int rc = r < k ? r : k - 1;
          g1[u][s] = G[(size_t)rc * ldg + cc];
          g2[u][s] = G[(size_t)cc * ldg + rc];
        }
      }
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        if (t0 + u >= SLOTS) continue;
        const int t = t0 + u;
        const int c = 16 * gb[u] + li;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = 16 * ga[u] + 4 * s + lk;
          const bool in = r < k && c < k;
          double g = in ? 0.5 * (g1[u][s] + g2[u][s]) : ((r == c) ? 1.0 : 0.0);
          if (measure && sv[t]) {
            const double x = in ? g * invd[r] * invd[c] - (r == c ? 1.0 : 0.0) : 0.0;
            dev += (ga[u] == gb[u] ? 1.0 : 2.0) * x * x;
            if (in && r == c) tr += g;
          }
          if (in && r == c) g += shift;
          acc[t][s] = g;
        }
      }
    }
  };
  double dev = 0.0, tr = 0.0;
  load_blocks(0.0, true, dev, tr);
  dev = cm_block_sum(dev, red);
  tr = cm_block_sum(tr, red);
  tk1 = clock64();

  // near-orthonormal input and no triangular factor wanted: first-order inverse square root (see k_chol_reg)
  if (!full_r && dev < 1e-14) {
    bool pos = true;
    for (int i = tid; i < k; i += NT) pos = pos && diag0[i] > 0.0;
    if (__syncthreads_and(pos ? 1 : 0)) {
#pragma unroll
      for (int t = 0; t < SLOTS; ++t) {
        if (!sv[t]) continue;
        const int c = 16 * sb[t] + li;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = 16 * sa[t] + 4 * s + lk;
          if (r < k && c < k) {
            const double di = invd[r], dj = invd[c];
            const double id = (r == c) ? 1.0 : 0.0;
            const double e = acc[t][s] * di * dj - id;
            Rinv[(size_t)r * ldo + c] = di * (id - 0.5 * e);
            Rout[(size_t)r * ldo + c] = (id + 0.5 * e) * diag0[c] * dj;
            if (sa[t] != sb[t]) {
              Rinv[(size_t)c * ldo + r] = dj * (-0.5 * e);
              Rout[(size_t)c * ldo + r] = (0.5 * e) * diag0[r] * di;
            }
            if (r == c) rdiag[r] = (rtot_mode == 1 ? 1.0 : rdiag[r]) * (1.0 + 0.5 * e) * diag0[r] * di;
          }
        }
      }
      if (tid == 0) {
        status->min_pivot_ratio = 1.0;
        status->gram_dev = sqrt(dev);
        status->shifted = 0;
        status->failed = 0;
        status->tick[0] = tk1 - tk0;
        status->tick[1] = 0;
        status->tick[2] = 0;
        status->tick[3] = clock64() - tk1;
        status->tick[4] = 3;
      }
      return;
    }
  }

  // strictly lower blocks of both factors are zero
#pragma unroll
  for (int t = 0; t < SLOTS; ++t)
    if (sv[t] && sa[t] != sb[t]) {
      const int c = 16 * sa[t] + li;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = 16 * sb[t] + 4 * s + lk;
        if (r < k && c < k) {
          Rout[(size_t)r * ldo + c] = 0.0;
          Rinv[(size_t)r * ldo + c] = 0.0;
        }
      }
    }

  int shifted = 0, failed = 0;
  double shift = 0.0;
  long long tphase[3] = {0, 0, 0};                  // thread 0's view of the phases: diagonal block, block row, trailing update
  for (int attempt = 0; attempt < 2; ++attempt) {
    shift = attempt ? shift_rel * tr : 0.0;
    if (attempt) {
      double d0 = 0.0, d1 = 0.0;
      load_blocks(shift, false, d0, d1);
    }
    if (tid == 0) s_fail = 0;
    for (int i = tid; i < 256; i += NT) sref[i] = i < k ? diag0[i] + shift : 1.0;
    __syncthreads();
    for (int K = 0; K < nb; ++K) {
      const long long ta = clock64();
      int lane = lane0;
      asm volatile("" : "+v"(lane));               // (same: nothing lane-dependent is hoisted out of the block loop)
      const int li = lane & 15, lk = lane >> 4;
      double* pan = panel + (size_t)(K & 1) * nb * 256;
      const int qd = K * (K + 1) / 2 + K;
      const int dw = qd % NW, dt = qd / NW;
      if (wave == dw) {
#pragma unroll
        for (int t = 0; t < SLOTS; ++t)
          if (t == dt) {
#pragma unroll
            for (int s = 0; s < 4; ++s) myscr[(4 * s + lk) * 17 + li] = acc[t][s];
          }
        chol_diag16(myscr, myscr + 272, pan + K * 256, sref, spiv, srd, K, pivot_tol, &s_fail, lane);
      }
      cm_lds_barrier();
      if (s_fail) break;                                   // uniform: every wave reads the word after the same barrier
      const long long tb = clock64();
      // B: block row K
      {
        // R_KK^-1 = Z_KK^T, read transposed from the diagonal wave's scratch (stride 17: no bank conflicts)
        const double* dscr = scratch + dw * CM_SCR;
        d4 xf;
#pragma unroll
        for (int s = 0; s < 4; ++s) xf[s] = dscr[272 + li * 17 + 4 * s + lk];
        if (wave == dw) {
          // the diagonal blocks of both factors, four rows of 128 bytes per store
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int r = 16 * K + 4 * s + lk, c = 16 * K + li;
            if (r < k && c < k) {
              Rout[(size_t)r * ldo + c] = dscr[(4 * s + lk) * 17 + li];
              Rinv[(size_t)r * ldo + c] = xf[s];
            }
          }
        }
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
          if (!sv[t]) continue;
          const int a = cm_opaque(sa[t]), b = cm_opaque(sb[t]);
          const bool rrow = a == K && b > K;               // A_KJ -> R_KJ
          const bool zrow = b == K && a < K;               // W_KJ -> Z_KJ
          if (rrow || zrow) {
            d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) y = MFMA_F64(xf[s], acc[t][s], y);
            const int J = rrow ? b : a;
            double* dst = pan + J * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) dst[64 * s + lane] = y[s];
            if (rrow) {
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                const int r = 16 * K + 4 * s + lk, c = 16 * J + li;
                if (r < k && c < k) Rout[(size_t)r * ldo + c] = y[s];
              }
              acc[t] = d4{0.0, 0.0, 0.0, 0.0};             // the slot now carries W_JK, born in phase C
            } else {
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                const int r = 16 * J + li, c = 16 * K + 4 * s + lk;   // X_JK = Z_KJ^T
                if (r < k && c < k) Rinv[(size_t)r * ldo + c] = y[s];
              }
            }
          }
        }
      }
      cm_lds_barrier();
      const long long tc = clock64();
      // C: everything below block row K
#pragma unroll
      for (int t = 0; t < SLOTS; ++t) {
        const int a = cm_opaque(sa[t]), b = cm_opaque(sb[t]);
        if (!sv[t] || b <= K) continue;
        const bool amode = a > K;
        const double* pa = pan + (amode ? a : b) * 256;
        const double* pb = pan + (amode ? b : a) * 256;
        double fa[4], fb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          fa[s] = -pa[64 * s + lane];
          fb[s] = pb[64 * s + lane];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t] = MFMA_F64(fa[s], fb[s], acc[t]);
      }
      tphase[0] += tb - ta;
      tphase[1] += tc - tb;
      tphase[2] += clock64() - tc;
    }
    __syncthreads();
    if (!s_fail) break;
    if (attempt == 0) shifted = 1;
    else failed = 1;
    __syncthreads();
  }
  if (failed) {
    if (tid == 0) {
      status->min_pivot_ratio = 0.0;
      status->gram_dev = sqrt(dev);
      status->shifted = shifted;
      status->failed = 1;
    }
    return;
  }
  double ratio = 1e300;
  for (int j = tid; j < k; j += NT) ratio = fmin(ratio, spiv[j] / sref[j]);
  const double min_ratio = cm_block_min(ratio, red);
  for (int j = tid; j < k; j += NT) rdiag[j] = (rtot_mode == 1 ? 1.0 : rdiag[j]) * srd[j];
  tk2 = clock64();
  tk3 = tk2;
  if (full_r) {
    __threadfence_block();
    __syncthreads();
    if (rtot_mode == 1) {
      for (int i = wave; i < k; i += NW)
        for (int j = lane; j < k; j += 64) Rtot[(size_t)i * ldo + j] = (j >= i) ? Rout[(size_t)i * ldo + j] : 0.0;
    } else {
      // Rtot <- R Rtot (upper times upper), block (I, J) = sum_{L = I..J} R_IL T_LJ on the MFMA, operands from the global slots
#pragma unroll
      for (int t = 0; t < SLOTS; ++t) {
        if (!sv[t]) continue;
        const int I = sa[t], J = sb[t];
        d4 y = {0.0, 0.0, 0.0, 0.0};
        for (int L = I; L <= J; ++L) {
          double fa[4], fb[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int ar = 16 * I + li, ac = 16 * L + 4 * s + lk;      // A[li][4 s + lk] = R_IL[li][4 s + lk]
            const int br = 16 * L + 4 * s + lk, bc = 16 * J + li;      // B[4 s + lk][li] = T_LJ[4 s + lk][li]
            fa[s] = (ar < k && ac < k) ? Rout[(size_t)ar * ldo + ac] : 0.0;
            fb[s] = (br < k && bc < k) ? Rtot[(size_t)br * ldo + bc] : 0.0;
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) y = MFMA_F64(fa[s], fb[s], y);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = 16 * I + 4 * s + lk, c = 16 * J + li;
          if (r < k && c < k) Rtmp[(size_t)r * ldo + c] = (c >= r) ? y[s] : 0.0;
        }
      }
      __threadfence_block();
      __syncthreads();
      for (int i = wave; i < k; i += NW)
        for (int j = lane; j < k; j += 64) Rtot[(size_t)i * ldo + j] = (j >= i) ? Rtmp[(size_t)i * ldo + j] : 0.0;
    }
  }
  tk4 = clock64();
  if (tid == 0) {
    status->min_pivot_ratio = min_ratio;
    status->gram_dev = sqrt(dev);
    status->shifted = shifted;
    status->failed = 0;
    status->tick[0] = tk1 - tk0;
    status->tick[1] = tk2 - tk1;
    status->tick[2] = tk3 - tk2;
    status->tick[3] = tk4 - tk3;
    status->tick[4] = 0;
    status->tick[5] = tphase[0];
    status->tick[6] = tphase[1];
    status->tick[7] = tphase[2];
  }
}

}  // namespace

int launch_chol_mfma(hfmi_ctx* ctx, int k, int slot_gram, int slot_r, int slot_rinv, int slot_rtot, int rtot_mode, int full_r,
                     double shift_rel, double pivot_tol) {
  if (k < 1 || k > SM_MAXK) HFMI_FAIL(HFMI_ERR_INVALID, "chol_mfma: k=%d out of range", k);
  if (pivot_tol <= 0.0) pivot_tol = 64.0 * k * EPS_D;
  const int nb = (k + 15) / 16;
#define CHOL_MFMA(NWV, SLV)                                                                                                     \
  do {                                                                                                                         \
    auto kern = k_chol_mfma<NWV, SLV>;                                                                                         \
    const size_t shm = (size_t)(32 + 5 * 256 + (NWV) * CM_SCR + 2 * nb * 256) * sizeof(double);                                \
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                     \
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * (NWV)), shm, ctx->stream, sm_ptr(ctx, slot_gram), SM_LD, k, sm_ptr(ctx, slot_r), \
                       sm_ptr(ctx, slot_rinv), sm_ptr(ctx, slot_rtot), sm_ptr(ctx, SM_TMP2), SM_LD, rtot_mode, full_r,         \
                       shift_rel, pivot_tol, sm_ptr(ctx, SM_AUX), sm_ptr(ctx, SM_AUX) + SM_LD, ctx->status_dev);               \
  } while (0)
  // slots = nb (nb + 1) / 2 over the waves
  if (nb <= 5) CHOL_MFMA(8, 2);          // k <= 80: 15 slots
  else if (nb <= 7) CHOL_MFMA(8, 4);     // k <= 112: 28
  else if (nb <= 9) CHOL_MFMA(8, 6);     // k <= 144: 45
  else if (nb <= 12) CHOL_MFMA(8, 10);   // k <= 192: 78
  else CHOL_MFMA(8, 17);                 // k <= 256: 136 (sixteen waves would have 128 registers each: not enough)
#undef CHOL_MFMA
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
