// Rayleigh-Ritz eigensolve of the double pass (np.linalg.eigh(T) inside hippylib's doublePass /
// doublePassG, la.eigh of PODProjector.py:821) for k <= 256: Householder tridiagonalisation, Cuppen divide and conquer
// on the tridiagonal matrix, back-transformation -- the algorithm family of LAPACK's dsyevd, which is what the
// reference's numpy call runs, laid out for one CDNA4 compute unit instead of a CPU core.
//
//   k_tridiag<RI, CJ>   one workgroup, 16 waves.  The k x k matrix lives in REGISTERS (row r on wave r mod 16, column c
//                       on lane c mod 64: RI x CJ doubles per thread), only the Householder vector, the 16 per-wave
//                       partial products of A v and the vector p cross waves, through LDS: three barriers per column.
//                       Finished rows / columns are skipped with wave-uniform tests, so the work per step shrinks with
//                       the trailing block although the ownership map is static.
//   k_dc                one workgroup.  Every coupling e_i is torn off up front (leaves are 1 x 1), the merges run level
//                       by level: rank sort of the poles, deflation scan (dlaed2's rule), the secular equation in the
//                       variable shifted to the nearer pole (G lanes per root), Gu-Eisenstat re-computation of the
//                       rank-one vector, eigenvector update Q <- Q S on the fp64 MFMA.  Q, S live in L2-resident global
//                       scratch (4 k^2 doubles), vectors in LDS.
//   k_dc_back<EL>       many workgroups: the reflectors applied to the eigenvectors, 16 lanes per eigenvector (the dot
//                       products are four DPP steps, no LDS, no barrier), result written to its sorted position.
//
// tests/helpers/dc_eig_twin.py is the numpy twin of these three kernels (same tree, same deflation rule, same secular
// iteration); tests/test_dc_twin.py pins it against numpy.linalg.eigh on the CPU.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "hfmi_dc_common.h"

namespace {
constexpr int DC_THREADS = 1024;
constexpr int DC_WAVES = 16;
constexpr int DC_MAXN = 256;


// ------------------------------------------------------------------------------------------------ tridiagonalisation
// per-phase shader-clock counters of wave 0 (build with -DHFMI_DC_TICKS; every s_memtime read drains the LDS queue, so the
// instrumented kernel is slower than the production one)
#ifdef HFMI_DC_TICKS
#define TRI_TICK_DECL long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tl = clock64()
#define TRI_TICK(slot)              \
  do {                              \
    const long long _t = clock64(); \
    slot += _t - tl;                \
    tl = _t;                        \
  } while (0)
#define TRI_TICK_STORE \
  if (tid == 0) {      \
    ticks[12] = tk0;   \
    ticks[13] = tk1;   \
    ticks[14] = tk2;   \
    ticks[15] = tk3;   \
  }
#else
#define TRI_TICK_DECL
#define TRI_TICK(slot)
#define TRI_TICK_STORE
#endif
// T (row-major, ldt; the symmetric part is used) -> d[0..n), e[0..n-1) of H^T T H scaled by 2^-sexp, the reflectors
// v_j (row j of Vh, v_j[c] = 0 for c <= j, v_j[j+1] = 1) and tau_j, H = H_0 ... H_{n-3}, H_j = I - tau_j v_j v_j^T.
//
// NW = 4 or 8 waves with the matrix in their registers: row r on wave r mod NW, column c on lane c mod 64 (RI x CJ doubles per
// thread).  (16 waves of fewer elements each were measured first: issue-bound by the per-wave bookkeeping, 5000 cycles per column
// at k = 74.)  Per column j:
//   (b) every wave multiplies its rows into a slice of A v (A symmetric: column sums over the wave's rows)        | barrier |
//   (d) every thread adds the NW slices of its own columns: p = tau A v, the dot product p . v is a wave-local sum, and the
//       values v[r], w[r] a row needs are lane reads of those registers (no LDS); rank-two update of the trailing block.  The
//       wave that owns row j + 1 updates the chunk holding that row first, forms reflector j + 1 from it and then finishes
//       its other rows                                                                                             | barrier |
// Rows are processed in chunks of four with ONE wave-uniform test per chunk (finished chunks are skipped, so the work per column
// shrinks with the trailing block; a test per row costs two taken branches, more than the row's arithmetic).
// LR > 0 (192 < k <= 256): 64 k^2 doubles do not fit the registers of one compute unit (it has 64 K of them), so the LAST LR rows
// live in LDS (row r on wave r mod NW as well, 128 KB for LR = 64) and are multiplied / updated from there; the per-row values
// v[r], w[r] are then read where they are used instead of being held (registers), and w[r] comes from a lane read (LDS is full).
template <int RI, int CJ, int NW, int LR = 0>
__global__ __launch_bounds__(64 * NW) void k_tridiag(const double* __restrict__ T, int ldt, int n, double* __restrict__ dvec,
                                                    double* __restrict__ evec, double* __restrict__ Vh, int ldv,
                                                    double* __restrict__ tauv, int* __restrict__ sexp_out,
                                                    long long* __restrict__ ticks) {
  static_assert(RI % 4 == 0, "rows per wave come in chunks of four");
  static_assert(RI * NW + LR <= 64 * CJ, "row indices must fit the column groups");
  static_assert(LR % NW == 0, "LDS rows are dealt to the waves like the register rows");
  constexpr int NCH = RI / 4;
  constexpr int RREG = RI * NW;                 // rows [0, RREG) in registers, [RREG, RREG + LR) in LDS
  constexpr int LI = LR / NW;                   // LDS rows per wave
  constexpr bool PRE = (LR == 0);               // per-row v / w values preloaded into registers
  constexpr int LSTR = 64 * CJ;                 // LDS row stride
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // [LR][LSTR] when LR > 0
  // v_j: one buffer when every wave copies what it needs into registers before the barrier (PRE); two alternating ones when
  // v[r] is read during the update, which overlaps the owner's writing of v_j+1
  __shared__ double s_v2[PRE ? 1 : 2][DC_MAXN];
  __shared__ double s_part[NW][DC_MAXN];
  __shared__ double s_w[PRE ? NW : 1][PRE ? DC_MAXN : 1];
  __shared__ double s_tau;
  __shared__ double s_red[NW];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform for the compiler: scalar branches, scalar lane indices
  double A[RI][CJ];
  double amax = 0.0;
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int r = w + NW * i;
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) {
      const int c = l + 64 * jj;
      double a = 0.0;
      if (r < n && c < n) a = 0.5 * (T[(size_t)r * ldt + c] + T[(size_t)c * ldt + r]);
      A[i][jj] = a;
      amax = fmax(amax, fabs(a));
    }
  }
#pragma unroll
  for (int i2 = 0; i2 < LI; ++i2) {
    const int r = RREG + w + NW * i2;
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) {
      const int c = l + 64 * jj;
      double a = 0.0;
      if (r < n && c < n) a = 0.5 * (T[(size_t)r * ldt + c] + T[(size_t)c * ldt + r]);
      s_dyn[(r - RREG) * LSTR + c] = a;
      amax = fmax(amax, fabs(a));
    }
  }
  // power-of-two scaling to max |entry| in [1, 2): norms cannot overflow / underflow, eigenvalues scale back exactly
  amax = wave_max(amax);
  if (l == 0) s_red[w] = amax;
  __syncthreads();
  amax = 0.0;
  for (int ww = 0; ww < NW; ++ww) amax = fmax(amax, s_red[ww]);
  int sexp = 0;
  if (amax > 0.0 && isfinite(amax)) sexp = ilogb(amax);
  const double sc = ldexp(1.0, -sexp);
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) A[i][jj] *= sc;
#pragma unroll
  for (int i2 = 0; i2 < LI; ++i2)
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) s_dyn[(w + NW * i2) * LSTR + l + 64 * jj] *= sc;      // this wave's own rows
  if (tid == 0) *sexp_out = sexp;

  // reflector j from x = A[j][.] (the owner wave's copy of row j): -> s_v, s_tau now; Vh, tau, e[j], d[j] go to global memory
  // after the next barrier (pend_*): a global store costs ~50 issue cycles and five of them sat on the critical path
  // (the reflector itself is not held: it is flushed in iteration pend_j, where it is the current v the wave has loaded anyway)
  // and its scalars wait in LDS: registers are what this kernel is short of)
  __shared__ double s_pend[3];
  int pend_j = -1;
  auto form_reflector = [&](int j, const double (&x)[CJ]) {
    double xn2 = 0.0, alpha_l = 0.0, d_l = 0.0;
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) {
      const int c = l + 64 * jj;
      if (c == j) d_l = x[jj];
      if (c == j + 1) alpha_l = x[jj];
      if (c > j + 1 && c < n) xn2 = fma(x[jj], x[jj], xn2);
    }
    xn2 = wave_sum(xn2);
    const double alpha = lane_get(alpha_l, (j + 1) & 63);   // only that lane's value is not zero, whichever column group
    double tau = 0.0, beta = alpha, scl = 0.0;
    if (xn2 > 1e-280) {     // entries are scaled to O(1): below this the column is zero to any precision that matters
      const double s2 = fma(alpha, alpha, xn2);
      const double rs = dc_rsqrt(s2);
      const double nrm = s2 * rs;                         // sqrt(alpha^2 + |x|^2)
      beta = -copysign(nrm, alpha);
      tau = fma(fabs(alpha), rs, 1.0);                    // (beta - alpha) / beta
      scl = copysign(dc_rcp(fabs(alpha) + nrm), alpha);   // 1 / (alpha - beta)
    }
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) {
      const int c = l + 64 * jj;
      const double v = (c == j + 1) ? 1.0 : ((c > j + 1 && c < n) ? x[jj] * scl : 0.0);
      s_v2[PRE ? 0 : (j & 1)][c] = v;
    }
    const double dj = lane_get(d_l, j & 63);
    if (l == 0) {
      s_tau = tau;
      s_pend[0] = dj;
      s_pend[1] = beta;
      s_pend[2] = tau;
    }
    pend_j = j;
  };
  auto flush_pending = [&](const double (&vcur)[CJ]) {      // vcur = v_{pend_j}: reflector j is flushed in iteration j
    if (pend_j >= 0) {
      asm volatile("");
#pragma unroll
      for (int jj = 0; jj < CJ; ++jj) {
        const int c = l + 64 * jj;
        if (c < ldv) Vh[(size_t)pend_j * ldv + c] = vcur[jj];
      }
      if (l == 0) {      // (this wave wrote s_pend itself, and nobody rewrites it before the next reflector is formed)
        dvec[pend_j] = s_pend[0];
        evec[pend_j] = s_pend[1];
        tauv[pend_j] = s_pend[2];
      }
      pend_j = -1;
    }
  };
  // row rsel of this wave out of chunk ch
  auto pick_row = [&](int ch, int rsel, double (&x)[CJ]) {
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) x[jj] = 0.0;
    if (LR > 0 && rsel >= RREG) {
#pragma unroll
      for (int jj = 0; jj < CJ; ++jj) x[jj] = s_dyn[(rsel - RREG) * LSTR + l + 64 * jj];
      return;
    }
#pragma unroll
    for (int c4 = 0; c4 < NCH; ++c4) {
      if (c4 == ch) {
        asm volatile("");                      // keep this a branch: as selects it is 4 NCH CJ v_cndmask per column
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if ((w + NW * (4 * c4 + u)) == rsel) {
            asm volatile("");                // (as selects the compiler keeps a copy of the rows in scratch and indexes it)
#pragma unroll
            for (int jj = 0; jj < CJ; ++jj) x[jj] = A[4 * c4 + u][jj];
          }
        }
      }
    }
  };

  if (n > 2 && w == 0) {
    double x[CJ];
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) x[jj] = A[0][jj];
    form_reflector(0, x);
  }
  __syncthreads();
  TRI_TICK_DECL;
  for (int j = 0; j + 2 < n; ++j) {
    const double tau = s_tau;
    const double* s_v = s_v2[PRE ? 0 : (j & 1)];
    const bool next_mine = (j + 3 < n) && (w == ((j + 1) & (NW - 1)));
    const bool next_lds = LR > 0 && j + 1 >= RREG;             // row j + 1 lives in LDS
    const int chn = ((j + 1) / NW) >> 2;       // chunk of row j + 1 on its owner (register rows)
    double vc[CJ];
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) vc[jj] = s_v[l + 64 * jj];
    if (tau != 0.0) {   // uniform over the workgroup
      // (b) this wave's slice of A v.  v[r] of the wave's rows: wave-uniform LDS reads, all issued up front (a lane read of vc
      // costs 40 cycles a row in dependent hazards); finished rows inside a live chunk multiply by v[r] = 0.
      double vr[PRE ? RI : 1];
      if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < RI; ++i) vr[i] = s_v[w + NW * i];
      }
      flush_pending(vc);
      {
        // two accumulators per column group while the groups alone are too few independent chains for the FMA latency
        constexpr int NA = 2;
        double acc[CJ][2];
#pragma unroll
        for (int jj = 0; jj < CJ; ++jj) acc[jj][0] = acc[jj][1] = 0.0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          if (w + NW * (4 * ch + 3) > j) {
            asm volatile("");
            double v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if constexpr (PRE) v4[u] = vr[4 * ch + u];
              else v4[u] = s_v[w + NW * (4 * ch + u)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int i = 4 * ch + u;
#pragma unroll
              for (int jj = 0; jj < CJ; ++jj) acc[jj][u & (NA - 1)] = fma(A[i][jj], v4[u], acc[jj][u & (NA - 1)]);
            }
          }
        }
#pragma unroll
        for (int i2 = 0; i2 < LI; ++i2) {
          const int r = RREG + w + NW * i2;
          if (r > j) {
            asm volatile("");
            const double vri = s_v[r];
#pragma unroll
            for (int jj = 0; jj < CJ; ++jj) acc[jj][i2 & (NA - 1)] = fma(s_dyn[(r - RREG) * LSTR + l + 64 * jj], vri, acc[jj][i2 & (NA - 1)]);
          }
        }
#pragma unroll
        for (int jj = 0; jj < CJ; ++jj) s_part[w][l + 64 * jj] = acc[jj][0] + acc[jj][1];
      }
      __syncthreads();
      TRI_TICK(tk0);
      // (d) p = tau A v on this thread's columns, w = p - (tau/2)(p . v) v, A <- A - v w^T - w v^T
      double wc[CJ];
      {
        double pc[CJ], dl = 0.0;
#pragma unroll
        for (int jj = 0; jj < CJ; ++jj) {
          const int c = l + 64 * jj;
          double q[NW];
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) q[ww] = s_part[ww][c];
#pragma unroll
          for (int st = 1; st < NW; st *= 2)
#pragma unroll
            for (int ww = 0; ww + st < NW; ww += 2 * st) q[ww] += q[ww + st];
          pc[jj] = (c > j && c < n) ? tau * q[0] : 0.0;
          dl = fma(pc[jj], vc[jj], dl);
          if constexpr (!PRE) asm volatile("");       // one column group's slices in flight at a time (registers)
        }
        const double kk = 0.5 * tau * wave_sum(dl);
#pragma unroll
        for (int jj = 0; jj < CJ; ++jj) {
          wc[jj] = fma(-kk, vc[jj], pc[jj]);
          if constexpr (PRE) s_w[w][l + 64 * jj] = wc[jj];
        }
      }
      double wr[PRE ? RI : 1];
      if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < RI; ++i) wr[i] = s_w[w][w + NW * i];
      }
      TRI_TICK(tk1);
      auto update_chunk = [&](int ch) {
        if constexpr (PRE) {
          double v4[4], w4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            v4[u] = vr[4 * ch + u];
            w4[u] = wr[4 * ch + u];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = 4 * ch + u;
#pragma unroll
            for (int jj = 0; jj < CJ; ++jj) A[i][jj] = fma(-w4[u], vc[jj], fma(-v4[u], wc[jj], A[i][jj]));
          }
        } else {
          // two rows at a time: the four (v, w) pairs of a chunk held at once are eight registers this shape does not have
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double v2[2], w2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int i = 4 * ch + 2 * h + u;
              v2[u] = s_v[w + NW * i];
              w2[u] = lane_get(wc[(NW * i) / 64], (w + NW * i) & 63);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int i = 4 * ch + 2 * h + u;
#pragma unroll
              for (int jj = 0; jj < CJ; ++jj) A[i][jj] = fma(-w2[u], vc[jj], fma(-v2[u], wc[jj], A[i][jj]));
            }
            asm volatile("");
          }
        }
      };
      auto update_lds_row = [&](int i2) {
        const int r = RREG + w + NW * i2;
        const double vri = s_v[r];
        const double wri = lane_get(wc[((RREG + NW * i2) / 64) < CJ ? (RREG + NW * i2) / 64 : 0], r & 63);
#pragma unroll
        for (int jj = 0; jj < CJ; ++jj) {
          double* a = &s_dyn[(r - RREG) * LSTR + l + 64 * jj];
          *a = fma(-wri, vc[jj], fma(-vri, wc[jj], *a));
        }
      };
      if (next_mine) {     // the chunk (or LDS row) with row j + 1 first, then the next reflector, then the rest
        if (next_lds) {
#pragma unroll
          for (int i2 = 0; i2 < LI; ++i2)
            if (RREG + w + NW * i2 == j + 1) {
              asm volatile("");
              update_lds_row(i2);
            }
        } else {
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
            if (ch == chn) {
              asm volatile("");
              update_chunk(ch);
            }
        }
        double x[CJ];
        pick_row(chn, j + 1, x);
        form_reflector(j + 1, x);
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        if (w + NW * (4 * ch + 3) > j && !(next_mine && !next_lds && ch == chn)) {
          asm volatile("");
          update_chunk(ch);
        }
      }
#pragma unroll
      for (int i2 = 0; i2 < LI; ++i2) {
        const int r = RREG + w + NW * i2;
        if (r > j && !(next_mine && r == j + 1)) {
          asm volatile("");
          update_lds_row(i2);
        }
      }
    } else {
      // column j is already reduced (tau = 0: diagonal, tridiagonal or decoupled input).  No barrier of the arithmetic path
      // separates the waves here, and form_reflector rewrites s_tau, s_pend and (PRE) the single-buffered s_v2: every wave
      // must have read tau and v_j, and the owner of row j must have flushed reflector j, before the owner of row j + 1 writes
      flush_pending(vc);
      __syncthreads();
      if (next_mine) {
        double x[CJ];
        pick_row(chn, j + 1, x);
        form_reflector(j + 1, x);
      }
    }
    TRI_TICK(tk2);
    __syncthreads();
    TRI_TICK(tk3);
  }
  TRI_TICK_STORE;
  // the last two rows hold d[n-2], e[n-2], d[n-1]
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int r = w + NW * i;
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) {
      const int c = l + 64 * jj;
      if (r < n && r + 2 >= n) {
        if (c == r) dvec[r] = A[i][jj];
        if (r + 2 == n && c == r + 1) evec[r] = A[i][jj];
      }
    }
  }
#pragma unroll
  for (int i2 = 0; i2 < LI; ++i2) {
    const int r = RREG + w + NW * i2;
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj) {
      const int c = l + 64 * jj;
      if (r < n && r + 2 >= n) {
        const double a = s_dyn[(r - RREG) * LSTR + c];
        if (c == r) dvec[r] = a;
        if (r + 2 == n && c == r + 1) evec[r] = a;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ divide and conquer
struct dc_args {
  int n, levels, sort_by_abs;
  const double* dvec;   // n
  const double* evec;   // n - 1
  const int* sexp;
  double* Q;            // n x n column-major (ld = ldq): eigenvectors of the tridiagonal matrix, columns in place
  double* Qt;           // scratch of the same shape
  double* S;            // n x n row-major (ld = ldq): eigenvectors of the rank-one problems of a level
  int ldq;
  double* dvals;        // out: eigenvalues in output order (descending / by |.|), scaled back
  int* perm;            // out: column of Q -> output position
  hfmi_status_words* status;
  long long* ticks;     // 16 counters: shader-clock time per phase, summed over the levels (HFMI_DC_TIMING=1 prints them)
  int split_top;        // 1: the eigenvector update of the LAST merge (half of all the flops of the updates) is left to k_dc_top,
  int* top_meta;        //    which spreads its tiles over the GPU; top_meta = {K, kept columns[n], live flag[n], column[n]}
};

// node i of level L covers columns [i n / 2^L, (i + 1) n / 2^L) (integer division); its children are nodes 2i and 2i + 1 of
// level L + 1, i.e. the split point is (2i + 1) n / 2^(L+1)
__device__ __forceinline__ void dc_node(int n, int L, int t, int& node, int& lo, int& mid, int& hi) {
  node = (((t + 1) << L) - 1) / n;
  lo = (node * n) >> L;
  hi = ((node + 1) * n) >> L;
  mid = ((2 * node + 1) * n) >> (L + 1);
}


template <int G>
__device__ __forceinline__ void dc_body(const dc_args& p) {
  __shared__ double sD[DC_MAXN], sZ[DC_MAXN], sDs[DC_MAXN], sZs[DC_MAXN], sDl[DC_MAXN], sW[DC_MAXN], sTau[DC_MAXN], sZh[DC_MAXN];
  __shared__ double sRc[DC_MAXN], sRs[DC_MAXN];
  __shared__ int sCol[DC_MAXN], sKs[DC_MAXN], sKc[DC_MAXN], sOrg[DC_MAXN], sRa[DC_MAXN], sRb[DC_MAXN];
  __shared__ double nRho[DC_MAXN / 2], nTol[DC_MAXN / 2];
  __shared__ int nK[DC_MAXN / 2], nRot[DC_MAXN / 2], nClose[DC_MAXN / 2];
  __shared__ unsigned long long nDmax[DC_MAXN / 2], nZmax[DC_MAXN / 2];
  __shared__ int sLive[DC_MAXN];
  __shared__ int s_fail, s_anyclose;
#ifdef HFMI_DC_TICKS
  __shared__ int s_evals, s_evmax, s_roots;
  if (threadIdx.x == 0) s_evals = s_evmax = s_roots = 0;
#endif
  const int n = p.n, tid = threadIdx.x, ldq = p.ldq;
  double* __restrict__ Q = p.Q;
  double* __restrict__ S = p.S;
  if (tid == 0) s_fail = 0;
  long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#ifdef HFMI_DC_TICKS
#define DC_TICK(slot)                  \
  do {                                 \
    const long long _t = clock64();    \
    tk[slot] += _t - tlast;            \
    tlast = _t;                        \
  } while (0)
#else
#define DC_TICK(slot)
#endif
  // tearing: every coupling is the split point of exactly one node
  if (tid < n) {
    double d = p.dvec[tid];
    if (tid > 0) d -= fabs(p.evec[tid - 1]);
    if (tid + 1 < n) d -= fabs(p.evec[tid]);
    sD[tid] = d;
  }
  for (int idx = tid; idx < n * n; idx += DC_THREADS) {
    const int c = idx / n, r = idx - c * n;
    Q[(size_t)c * ldq + r] = (r == c) ? 1.0 : 0.0;
    p.Qt[(size_t)c * ldq + r] = 0.0;     // a level writes the rows of its nodes only: the other rows of a column must be zero in BOTH buffers
  }
  __syncthreads();
  DC_TICK(0);

  double* __restrict__ Qin = p.Q;
  double* __restrict__ Qout = p.Qt;
  int Ltop = p.levels - 1;
  // The deepest level only merges 1 x 1 leaves into 2 x 2 blocks: [[d_i, e_i], [e_i, d_i+1]] has its eigen-decomposition in
  // closed form (the arithmetic of LAPACK's dlaev2), one thread per block -- a whole level of the machinery below (about a
  // seventh of this kernel at k = 74) for a few dozen operations.  It works on the untorn diagonal of the pair.
  if (p.levels >= 1) {
    const int L = p.levels - 1;
    if (tid < n) {
      int node, lo, mid, hi;
      dc_node(n, L, tid, node, lo, mid, hi);
      if (hi - lo == 2 && tid == lo) {
        const double e = p.evec[lo];
        const double a = sD[lo] + fabs(e), c = sD[lo + 1] + fabs(e), b = e;     // give the pair its own coupling back
        const double sm = a + c, df = a - c, adf = fabs(df), tb = b + b, ab = fabs(tb);
        const double acmx = fabs(a) > fabs(c) ? a : c, acmn = fabs(a) > fabs(c) ? c : a;
        double rt;
        if (adf > ab) rt = adf * sqrt(1.0 + (ab / adf) * (ab / adf));
        else if (adf < ab) rt = ab * sqrt(1.0 + (adf / ab) * (adf / ab));
        else rt = ab * 1.4142135623730951;
        double rt1, rt2;
        int sgn1;
        if (sm < 0.0) {
          rt1 = 0.5 * (sm - rt);
          sgn1 = -1;
          rt2 = (acmx / rt1) * acmn - (b / rt1) * b;
        } else if (sm > 0.0) {
          rt1 = 0.5 * (sm + rt);
          sgn1 = 1;
          rt2 = (acmx / rt1) * acmn - (b / rt1) * b;
        } else {
          rt1 = 0.5 * rt;
          rt2 = -0.5 * rt;
          sgn1 = 1;
        }
        int sgn2;
        double cs;
        if (df >= 0.0) {
          cs = df + rt;
          sgn2 = 1;
        } else {
          cs = df - rt;
          sgn2 = -1;
        }
        double cs1, sn1;
        if (fabs(cs) > ab) {
          const double ct = -tb / cs;
          sn1 = 1.0 / sqrt(1.0 + ct * ct);
          cs1 = ct * sn1;
        } else if (ab == 0.0) {
          cs1 = 1.0;
          sn1 = 0.0;
        } else {
          const double tn = -cs / tb;
          cs1 = 1.0 / sqrt(1.0 + tn * tn);
          sn1 = tn * cs1;
        }
        if (sgn1 == sgn2) {
          const double tn = cs1;
          cs1 = -sn1;
          sn1 = tn;
        }
        // (cs1, sn1) is the unit eigenvector of rt1, (-sn1, cs1) that of rt2
        sD[lo] = rt1;
        sD[lo + 1] = rt2;
        Qin[(size_t)lo * ldq + lo] = cs1;
        Qin[(size_t)lo * ldq + lo + 1] = sn1;
        Qin[(size_t)(lo + 1) * ldq + lo] = -sn1;
        Qin[(size_t)(lo + 1) * ldq + lo + 1] = cs1;
      }
    }
    __syncthreads();
    Ltop = p.levels - 2;
  }
  for (int L = Ltop; L >= 0; --L) {
    int node = 0, lo = 0, mid = 0, hi = 0;
    bool active = false;
    if (tid < n) {
      dc_node(n, L, tid, node, lo, mid, hi);
      active = hi - lo >= 2;
    }
    if (tid < DC_MAXN / 2) {
      nDmax[tid] = 0ull;
      nZmax[tid] = 0ull;
      nClose[tid] = 0;
    }
    if (tid == 0) s_anyclose = 0;
    __syncthreads();
    // P1: the rank-one vector: last row of the left child's Q, first row of the right child's; node-wide max |d|, max |z|
    // (non-negative doubles order like their bit patterns)
    double zt = 0.0, dt = 0.0, rho = 0.0;
    if (active) {
      const double beta = p.evec[mid - 1];
      rho = 2.0 * fabs(beta);
      const double q = (tid < mid) ? Qin[(size_t)tid * ldq + (mid - 1)] : (beta >= 0.0 ? 1.0 : -1.0) * Qin[(size_t)tid * ldq + mid];
      zt = q * 0.70710678118654752440;
      dt = sD[tid];
      sZ[tid] = zt;
      atomicMax(&nDmax[node], (unsigned long long)__double_as_longlong(fabs(dt)));
      atomicMax(&nZmax[node], (unsigned long long)__double_as_longlong(fabs(zt)));
    }
    __syncthreads();
    DC_TICK(1);
    // P2: ascending rank inside the node; components below the tolerance drop out (dlaed2's first test); position among the
    // survivors.  If no two neighbouring survivors turn out to be close (P3) this IS the deflation: nothing is sequential.
    double tol = 0.0;
    bool small_t = false;
    if (active) {
      const double dmax = __longlong_as_double((long long)nDmax[node]), zmax = __longlong_as_double((long long)nZmax[node]);
      tol = 8.0 * DC_EPS * fmax(dmax, zmax);
      const bool skip = rho * zmax <= tol;
      const double thr = skip ? INFINITY : tol;            // nothing couples: every component counts as small
      small_t = rho * fabs(zt) <= thr;
      int rank = 0, pre = 0, cnt = 0;
      for (int u = lo; u < hi; ++u) {
        const double du = sD[u], zu = sZ[u];
        const bool before = du < dt || (du == dt && u < tid);
        const bool live = !(rho * fabs(zu) <= thr);
        rank += before ? 1 : 0;
        pre += (before && live) ? 1 : 0;
        cnt += live ? 1 : 0;
      }
      sDs[lo + rank] = dt;
      sZs[lo + rank] = zt;
      sCol[lo + rank] = tid;
      sLive[lo + rank] = small_t ? 0 : 1;
      if (!small_t) sKs[lo + pre] = lo + rank;
      if (tid == lo) {
        nRho[node] = rho;
        nTol[node] = tol;
        nK[node] = cnt;
        nRot[node] = 0;
      }
    }
    __syncthreads();
    DC_TICK(2);
    // P3: dlaed2's second test on every pair of neighbouring survivors, in parallel
    if (active) {
      const int s = tid;                                    // sorted position
      if (sLive[s]) {
        int pv = s - 1;
        while (pv >= lo && !sLive[pv]) --pv;
        if (pv >= lo) {
          const double zs = sZs[s], zp = sZs[pv];
          const double t = sDs[s] - sDs[pv];
          // |t c s| with c = zs / tau, s = -zp / tau, tau^2 = zs^2 + zp^2
          if (fabs(t * zs * zp) <= tol * fma(zs, zs, zp * zp)) {
            nClose[node] = 1;
            s_anyclose = 1;
          }
        }
      }
    }
    __syncthreads();
    if (s_anyclose) {     // uniform: some node has close poles -> that node's leader redoes the scan sequentially, with rotations
      if (active && tid == lo && nClose[node]) {
        int K = 0, nrot = 0, pj = -1;
        double zp = 0.0, dp = 0.0;
        for (int s = lo; s < hi; ++s) {
          const double zs = sZs[s];
          const bool live = sLive[s] != 0;
          sLive[s] = 0;
          if (!live) continue;
          const double ds = sDs[s];
          if (pj < 0) {
            pj = s;
            zp = zs;
            dp = ds;
            continue;
          }
          const double tau = sqrt(fma(zs, zs, zp * zp));     // |z| <= 1: no overflow to guard
          const double c = zs / tau, sn = -zp / tau;
          const double t = ds - dp;
          if (fabs(t * c * sn) <= tol) {
            sZs[pj] = 0.0;
            sRa[lo + nrot] = pj;
            sRb[lo + nrot] = s;
            sRc[lo + nrot] = c;
            sRs[lo + nrot] = sn;
            ++nrot;
            sDs[pj] = dp * c * c + ds * sn * sn;
            const double dnew = dp * sn * sn + ds * c * c;
            sDs[s] = dnew;
            sZs[s] = tau;
            pj = s;
            zp = tau;
            dp = dnew;
          } else {
            sLive[pj] = 1;
            sKs[lo + K++] = pj;
            pj = s;
            zp = zs;
            dp = ds;
          }
        }
        if (pj >= 0) {
          sLive[pj] = 1;
          sKs[lo + K++] = pj;
        }
        nK[node] = K;
        nRot[node] = nrot;
      }
      __syncthreads();
    }
    DC_TICK(3);
    // P4: the rotations on the columns of Q (this thread's row), kept poles gathered, eigenvalues of all columns refreshed
    if (active) {
      const int nrot = nRot[node];
      for (int q = 0; q < nrot; ++q) {
        const int ca = sCol[sRa[lo + q]], cb = sCol[sRb[lo + q]];
        const double c = sRc[lo + q], sn = sRs[lo + q];
        const double qa = Qin[(size_t)ca * ldq + tid], qb = Qin[(size_t)cb * ldq + tid];
        Qin[(size_t)ca * ldq + tid] = c * qa + sn * qb;
        Qin[(size_t)cb * ldq + tid] = c * qb - sn * qa;
      }
      const int i = tid - lo;
      if (i < nK[node]) {
        const int pos = sKs[tid];
        sDl[tid] = sDs[pos];
        sW[tid] = sZs[pos];
        sKc[tid] = sCol[pos];
      }
      sD[sCol[tid]] = sDs[tid];
    }
    __syncthreads();
    DC_TICK(4);
    // P5-P7 with GL lanes per root, GL chosen per level: the lanes of a group all execute the ~150 instructions of a round, so
    // small nodes (few poles per root) get few lanes -- at k = 74 the seven levels then keep 2, 2, 3, 3, 5, 10, 10 waves busy
    // instead of 10 each, and a round costs its own issue time instead of three waves' worth per SIMD
    auto secular_phases = [&](auto gtag) {
      constexpr int GL = decltype(gtag)::value;
      const int slot = tid / GL, sub = tid % GL;
      int nd = 0, l2 = 0, m2 = 0, h2 = 0, K = 0;
      if (slot < n) {
        dc_node(n, L, slot, nd, l2, m2, h2);
        K = (h2 - l2 >= 2) ? nK[nd] : 0;
      }
      const int i = slot - l2;
      const bool mine = slot < n && i < K;
      // P5: secular equation
      if (mine) {
        int org;
        double tau;
        int evals;
        const bool ok = secular_root<GL>(sDl + l2, sW + l2, K, i, sub, nRho[nd], org, tau, evals);
        if (sub == 0) {
          sTau[slot] = tau;
          sOrg[slot] = org;
          if (!ok) s_fail = 1;
#ifdef HFMI_DC_TICKS
          atomicAdd(&s_evals, evals);
          atomicMax(&s_evmax, evals);
          atomicAdd(&s_roots, 1);
#endif
        }
      }
      __syncthreads();
      DC_TICK(5);
      // P6: Gu-Eisenstat: the rank-one vector for which the COMPUTED roots are exact
      if (mine) {
        const double* dl = sDl + l2;
        const double di = dl[i];
        double prod = 1.0;
        for (int j = sub; j < K; j += GL) {
          const double num = (dl[sOrg[l2 + j]] - di) + sTau[l2 + j];     // lam_j - dl_i
          prod *= (j == i) ? num : num * dc_rcp(dl[j] - di);
        }
        prod = group_prod<GL>(prod);
        if (sub == 0) sZh[slot] = copysign(sqrt(fabs(prod)), sW[slot]);
      }
      __syncthreads();
      DC_TICK(6);
      // P7: eigenvectors of the rank-one problem, normalised, and the new eigenvalues
      if (mine) {
        const double* dl = sDl + l2;
        const double dorg = dl[sOrg[slot]], tau = sTau[slot];
        double nrm = 0.0;
        for (int q = sub; q < K; q += GL) {
          const double sv = sZh[l2 + q] * dc_rcp((dl[q] - dorg) - tau);
          nrm = fma(sv, sv, nrm);
        }
        nrm = group_sum<GL>(nrm);
        const double inv = dc_rsqrt(nrm);
        for (int q = sub; q < K; q += GL) {
          const double sv = sZh[l2 + q] * dc_rcp((dl[q] - dorg) - tau);
          S[(size_t)(l2 + q) * ldq + (l2 + i)] = sv * inv;
        }
        if (sub == 0) sD[sKc[slot]] = dorg + tau;
      }
      __syncthreads();
      DC_TICK(7);
    };
    {
      const int nnmax = (n + (1 << L) - 1) >> L;
      if (nnmax <= 4) secular_phases(std::integral_constant<int, 1>());
      else if (nnmax <= 12 || G < 4) secular_phases(std::integral_constant<int, 2>());
      else if (nnmax <= 40 || G < 8) secular_phases(std::integral_constant<int, 4>());
      else secular_phases(std::integral_constant<int, 8>());
    }
    // P8: Qout[:, kept] = Qin[:, kept] S on the fp64 MFMA (rows of the node only: Q is block diagonal by construction); every
    // other column moves over unchanged
    if (L == 0 && p.split_top) {
      if (tid == 0) p.top_meta[0] = nK[0];
      if (tid < n) {
        p.top_meta[1 + tid] = sKc[tid];
        p.top_meta[1 + DC_MAXN + tid] = sLive[tid];
        p.top_meta[1 + 2 * DC_MAXN + tid] = sCol[tid];
      }
    } else {
      const int nnmax = (n + (1 << L) - 1) >> L;
      const int tpn = (nnmax + 15) >> 4, tiles = (1 << L) * tpn * tpn;
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, li = l & 15, lk = l >> 4;
      for (int tile = wv; tile < tiles; tile += DC_WAVES) {
        const int nd = tile / (tpn * tpn), rem = tile - nd * tpn * tpn;
        const int tr = rem / tpn, tc = rem - tr * tpn;
        const int l2 = (nd * n) >> L, h2 = ((nd + 1) * n) >> L;
        if (h2 - l2 < 2) continue;
        const int K = nK[nd];
        const int r0 = l2 + tr * 16, j0 = tc * 16;
        if (j0 >= K || r0 >= h2) continue;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const bool colok = j0 + li < K, rowok = r0 + li < h2;
        for (int k0 = 0; k0 < K; k0 += 32) {     // eight k-steps of loads in flight before the first MFMA needs one
          double a[8], b[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int ki = k0 + 4 * u + lk;
            a[u] = 0.0;
            b[u] = 0.0;
            if (ki < K) {
              if (colok) a[u] = S[(size_t)(l2 + ki) * ldq + (l2 + j0 + li)];
              if (rowok) b[u] = Qin[(size_t)sKc[l2 + ki] * ldq + (r0 + li)];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = MFMA_F64(a[u], b[u], acc);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int jc = j0 + lk + 4 * reg;
          if (jc < K && rowok) Qout[(size_t)sKc[l2 + jc] * ldq + (r0 + li)] = acc[reg];
        }
      }
      const int total = n * nnmax;           // (sorted position, row offset)
      for (int idx = tid; idx < total; idx += DC_THREADS) {
        const int sp = idx / nnmax, ro = idx - sp * nnmax;
        int nd, l2, m2, h2;
        dc_node(n, L, sp, nd, l2, m2, h2);
        if (l2 + ro >= h2) continue;
        const bool act = h2 - l2 >= 2;
        if (act && sLive[sp]) continue;        // a kept column: written by the product above
        const size_t off = (size_t)(act ? sCol[sp] : sp) * ldq + (l2 + ro);
        Qout[off] = Qin[off];
      }
    }
    __syncthreads();
    DC_TICK(8);
    {
      double* const t = Qin;
      Qin = Qout;
      Qout = t;
    }
  }
  // output order: descending (by magnitude on request), ties by column
  if (tid < n) {
    const double dt = sD[tid];
    const double key = p.sort_by_abs ? fabs(dt) : dt;
    int rank = 0;
    for (int u = 0; u < n; ++u) {
      const double du = p.sort_by_abs ? fabs(sD[u]) : sD[u];
      rank += (du > key || (du == key && u < tid)) ? 1 : 0;
    }
    p.perm[tid] = rank;
    p.dvals[rank] = ldexp(dt, *p.sexp);
  }
  DC_TICK(10);
  if (tid == 0) {
#ifdef HFMI_DC_TICKS
    tk[11] = ((long long)s_evmax << 40) | ((long long)s_roots << 20) | (long long)s_evals;
#endif
    for (int q = 0; q < 12; ++q) p.ticks[q] = tk[q];
    p.status->failed = s_fail;
    p.status->sweeps = p.levels;
    p.status->offdiag = 0.0;
    p.status->tick[4] = 2;
  }
}

template <int G>
__global__ __launch_bounds__(DC_THREADS) void k_dc(dc_args p) {
  dc_body<G>(p);
}
// the leaves of the whole-GPU eigensolver (hfmi_eig_blocked.hip): one independent tridiagonal problem per workgroup
template <int G>
__global__ __launch_bounds__(DC_THREADS) void k_dc_batch(const dc_args* __restrict__ batch) {
  const dc_args p = batch[blockIdx.x];
  dc_body<G>(p);
}

// ------------------------------------------------------------------------------------------------ last merge, all CUs
// Qout[:, kept] = Qin[:, kept] S for the root node (rows 0 .. n): one wave per 16 x 16 tile, the other columns copied by the
// remaining workgroups.  Inside k_dc the same product is 81 tiles on one CU with every operand an L2 round trip away (k = 138:
// 107 us); here each tile is one latency chain on its own CU.
__global__ __launch_bounds__(64) void k_dc_top(const double* __restrict__ Qin, double* __restrict__ Qout, const double* __restrict__ S,
                                               int ldq, int n, const int* __restrict__ meta) {
  const int K = meta[0];
  const int* __restrict__ kc = meta + 1;
  const int* __restrict__ live = meta + 1 + DC_MAXN;
  const int* __restrict__ colof = meta + 1 + 2 * DC_MAXN;
  const int T = (n + 15) >> 4, l = threadIdx.x, li = l & 15, lk = l >> 4;
  const int b = blockIdx.x;
  if (b < T * T) {
    const int tr = b / T, tc = b - tr * T;
    const int r0 = tr * 16, j0 = tc * 16;
    if (j0 >= K) return;
    const bool colok = j0 + li < K, rowok = r0 + li < n;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 64) {       // sixteen k-steps of loads in flight
      double a[16], bb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int ki = k0 + 4 * u + lk;
        a[u] = 0.0;
        bb[u] = 0.0;
        if (ki < K) {
          if (colok) a[u] = S[(size_t)ki * ldq + (j0 + li)];
          if (rowok) bb[u] = Qin[(size_t)kc[ki] * ldq + (r0 + li)];
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = MFMA_F64(a[u], bb[u], acc);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int jc = j0 + lk + 4 * reg;
      if (jc < K && rowok) Qout[(size_t)kc[jc] * ldq + (r0 + li)] = acc[reg];
    }
  } else {
    // copy duty: sorted positions sp = b - T*T, b - T*T + (copy blocks), ...
    const int nb = gridDim.x - T * T;
    for (int sp = b - T * T; sp < n; sp += nb) {
      if (live[sp]) continue;
      const int c = colof[sp];
      for (int r = l; r < n; r += 64) Qout[(size_t)c * ldq + r] = Qin[(size_t)c * ldq + r];
    }
  }
}

// ------------------------------------------------------------------------------------------------ back-transformation
// V_out[:, perm[c]] = H_0 H_1 ... H_{n-3} Q[:, c].  A wave carries four eigenvectors, 16 lanes each: lane m of a group holds
// rows m, m + 16, ...; the reflector rows are shared by the four groups.
template <int EL>
__global__ __launch_bounds__(64) void k_dc_back(const double* __restrict__ Q, int ldq, int n, const double* __restrict__ Vh, int ldv,
                                                const double* __restrict__ tauv, const int* __restrict__ perm,
                                                double* __restrict__ Vout, int ldo) {
  const int l = threadIdx.x, g = l >> 4, m = l & 15;
  const int col = blockIdx.x * 4 + g;
  const bool live = col < n;
  double z[EL];
#pragma unroll
  for (int e = 0; e < EL; ++e) {
    const int r = m + 16 * e;
    z[e] = (live && r < n) ? Q[(size_t)col * ldq + r] : 0.0;
  }
  // The reflector rows come from L2 (1-2 us away) and one reflector takes ~0.15 us to apply: a ring of PF rows is kept in
  // flight (one row ahead, as first written, the loop waited for memory three quarters of the time: 58 us at k = 138).
  constexpr int PF = EL <= 9 ? 8 : (EL <= 12 ? 6 : 4);       // as deep as the registers of a one-wave workgroup allow
  double vb[PF][EL], tb[PF];
  auto fetch = [&](double (&dst)[EL], double& tdst, int j) {
    const int jj = j >= 0 ? j : 0;                       // past the end: re-read row 0 (never used)
    tdst = tauv[jj];
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const int r = m + 16 * e;
      dst[e] = (r < ldv) ? Vh[(size_t)jj * ldv + r] : 0.0;
    }
  };
  int j = n - 3;
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(vb[u], tb[u], j - u);
  for (; j >= 0; j -= PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int jr = j - u;
      if (jr >= 0) {                                     // uniform
        const double tau = tb[u];
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int e = 0; e < EL; e += 2) {
          d0 = fma(vb[u][e], z[e], d0);
          if (e + 1 < EL) d1 = fma(vb[u][e + 1], z[e + 1], d1);
        }
        const double f = -tau * group_sum<16>(d0 + d1);
#pragma unroll
        for (int e = 0; e < EL; ++e) z[e] = fma(f, vb[u][e], z[e]);
      }
      fetch(vb[u], tb[u], jr - PF);
    }
  }
  if (live) {
    const int pc = perm[col];
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const int r = m + 16 * e;
      if (r < n) Vout[(size_t)r * ldo + pc] = z[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------ leaves of the whole-GPU solver
struct leaf_desc {
  int lo, nl, ldq;
  double* dleaf;
  const double* Qfin;
  const double* dvals;
  const int* perm;
  const hfmi_status_words* status;
};
// the couplings of the upper tree levels are torn before the leaves are solved: a leaf's first / last diagonal entry gives up
// |e| of the coupling to its neighbour (Cuppen's rank-one tearing, the same convention as inside dc_body)
__global__ void k_leaf_prep(const leaf_desc* __restrict__ leaves, int n, const double* __restrict__ dvec, const double* __restrict__ evec) {
  const leaf_desc L = leaves[blockIdx.x];
  for (int i = threadIdx.x; i < L.nl; i += blockDim.x) {
    double d = dvec[L.lo + i];
    if (i == 0 && L.lo > 0) d -= fabs(evec[L.lo - 1]);
    if (i == L.nl - 1 && L.lo + L.nl < n) d -= fabs(evec[L.lo + L.nl - 1]);
    L.dleaf[i] = d;
  }
}
// leaf results into the big problem: eigenvalue of every column, eigenvectors as the diagonal blocks of Q
__global__ void k_leaf_gather(const leaf_desc* __restrict__ leaves, double* __restrict__ Dout, double* __restrict__ Qbig, int64_t ldq,
                              int* __restrict__ fail) {
  const leaf_desc L = leaves[blockIdx.y];
  const int c = blockIdx.x;
  if (c >= L.nl) return;
  if (c == 0 && threadIdx.x == 0 && L.status->failed) atomicOr(fail, 1);
  if (threadIdx.x == 0) Dout[L.lo + c] = L.dvals[L.perm[c]];
  const double* src = L.Qfin + (size_t)c * L.ldq;
  double* dst = Qbig + (size_t)(L.lo + c) * ldq + L.lo;
  for (int r = threadIdx.x; r < L.nl; r += blockDim.x) dst[r] = src[r];
}
}  // namespace

// T (k x k, SM slot, row-major) -> eigenvalues (descending, or by magnitude) into dvals (device, k), eigenvectors into the
// columns of slot_v.  Same contract as launch_jacobi_eig.
int launch_dc_eig(hfmi_ctx* ctx, int k, int slot_t, int slot_v, double* dvals, int sort_by_abs) {
  if (k < 1 || k > SM_MAXK) HFMI_FAIL(HFMI_ERR_INVALID, "dc_eig: k=%d out of range", k);
  const int ldq = (int)round_up(k, 16);
  const size_t mat = (size_t)ldq * ldq;
  // scratch: d, e, tau (3 x 256), reflectors (256 x ldq), Q, Qt, S, perm + scale exponent
  const size_t doubles = 3 * DC_MAXN + (size_t)DC_MAXN * ldq + 3 * mat;
  void* wv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_MISC, doubles * sizeof(double) + (DC_MAXN + 16) * sizeof(int) + 16 * sizeof(long long) + (3 * DC_MAXN + 16) * sizeof(int), &wv));
  double* dvec = (double*)wv;
  double* evec = dvec + DC_MAXN;
  double* tauv = evec + DC_MAXN;
  double* Vh = tauv + DC_MAXN;
  double* Qm = Vh + (size_t)DC_MAXN * ldq;
  double* Qt = Qm + mat;
  double* Sm = Qt + mat;
  int* perm = (int*)(Sm + mat);
  int* sexp = perm + DC_MAXN;
  long long* ticks = (long long*)(sexp + 16);
  int* top_meta = (int*)(ticks + 16);
  const double* T = sm_ptr(ctx, slot_t);
  static int tri_waves = 0;     // HFMI_TRI_WAVES = 4 | 8: A/B of the workgroup shape
  if (!tri_waves) {
    const char* e = getenv("HFMI_TRI_WAVES");
    tri_waves = (e && atoi(e) == 4) ? 4 : 8;
  }
#define TRI(RIV, CJV, NWV)                                                                                                        \
  hipLaunchKernelGGL((k_tridiag<RIV, CJV, NWV>), dim3(1), dim3(64 * NWV), 0, ctx->stream, T, SM_LD, k, dvec, evec, Vh, ldq, tauv, \
                     sexp, ticks)
  if (tri_waves == 4) {
    if (k <= 32) TRI(8, 1, 4);
    else if (k <= 64) TRI(16, 1, 4);
    else if (k <= 80) TRI(20, 2, 4);
    else if (k <= 96) TRI(24, 2, 4);
    else if (k <= 128) TRI(32, 2, 4);
    else if (k <= 144) TRI(36, 3, 4);
    else if (k <= 160) TRI(40, 3, 4);
    else if (k <= 192) TRI(48, 3, 4);
    else TRI(64, 4, 4);
  } else {
    if (k <= 32) TRI(4, 1, 8);
    else if (k <= 64) TRI(8, 1, 8);
    else if (k <= 96) TRI(12, 2, 8);
    else if (k <= 128) TRI(16, 2, 8);
    else if (k <= 160) TRI(20, 3, 8);
    else {
      // 160 < k <= 256: 64 k^2 doubles leave no registers for anything else (the compiler spills 84..99 values per thread
      // and the reduction runs 3x slower), so the last LR rows live in LDS; the shapes below are the ones that do not spill
      // (k <= 192) or spill 5 values (k <= 224); above that 64 rows are all the LDS holds beside the work arrays
#define TRI_LDS(RIV, CJV, LRV)                                                                                                   \
  do {                                                                                                                          \
    auto kern = k_tridiag<RIV, CJV, 8, LRV>;                                                                                    \
    const size_t shm = (size_t)(LRV) * 64 * (CJV) * sizeof(double);                                                             \
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                      \
    hipLaunchKernelGGL(kern, dim3(1), dim3(512), shm, ctx->stream, T, SM_LD, k, dvec, evec, Vh, ldq, tauv, sexp, ticks);        \
  } while (0)
      if (k <= 192) TRI_LDS(20, 3, 32);
      else if (k <= 224) TRI_LDS(20, 4, 64);
      else TRI_LDS(24, 4, 64);
#undef TRI_LDS
    }
  }
#undef TRI
  HIP_TRY(hipGetLastError());
  dc_args a;
  a.n = k;
  a.levels = 0;
  while ((1 << a.levels) < k) ++a.levels;
  a.sort_by_abs = sort_by_abs ? 1 : 0;
  a.dvec = dvec;
  a.evec = evec;
  a.sexp = sexp;
  a.Q = Qm;
  a.Qt = Qt;
  a.S = Sm;
  a.ldq = ldq;
  a.dvals = dvals;
  a.perm = perm;
  a.status = ctx->status_dev;
  a.ticks = ticks;
  static int split_top = -1;    // HFMI_DC_SPLIT_TOP=0: the last merge's product inside k_dc (A/B)
  if (split_top < 0) {
    const char* e = getenv("HFMI_DC_SPLIT_TOP");
    split_top = e ? atoi(e) : 1;
  }
  a.split_top = (split_top && k >= 32) ? 1 : 0;
  a.top_meta = top_meta;
  if (k <= 128) hipLaunchKernelGGL((k_dc<8>), dim3(1), dim3(DC_THREADS), 0, ctx->stream, a);
  else hipLaunchKernelGGL((k_dc<4>), dim3(1), dim3(DC_THREADS), 0, ctx->stream, a);
  HIP_TRY(hipGetLastError());
  if (a.split_top) {
    // level 0 is the last of the ping-pong merges (levels - 1 of them: the deepest level works in place): its input is the
    // buffer after levels - 2 swaps
    const int before = a.levels >= 2 ? a.levels - 2 : 0;
    const double* Qin0 = (before & 1) ? Qt : Qm;
    double* Qout0 = (before & 1) ? Qm : Qt;
    const int T = (k + 15) / 16;
    hipLaunchKernelGGL(k_dc_top, dim3(T * T + 16), dim3(64), 0, ctx->stream, Qin0, Qout0, Sm, ldq, k, top_meta);
    HIP_TRY(hipGetLastError());
  }
  double* Vout = sm_ptr(ctx, slot_v);
  const int blocks = (k + 3) / 4;
  const int swaps = a.levels >= 1 ? a.levels - 1 : 0;  // the deepest level is solved in place (2 x 2 blocks in closed form)
  const double* Qfin = (swaps & 1) ? Qt : Qm;          // the other merges ping-pong between the two buffers
#define BACK(ELV) \
  hipLaunchKernelGGL((k_dc_back<ELV>), dim3(blocks), dim3(64), 0, ctx->stream, Qfin, ldq, k, Vh, ldq, tauv, perm, Vout, SM_LD)
  if (k <= 80) BACK(5);
  else if (k <= 96) BACK(6);
  else if (k <= 128) BACK(8);
  else if (k <= 144) BACK(9);
  else if (k <= 192) BACK(12);
  else BACK(16);
#undef BACK
  HIP_TRY(hipGetLastError());
  static const bool timing = env_flag("HFMI_DC_TIMING");
  if (timing) {
    long long h[16];
    HIP_TRY(hipMemcpyAsync(h, ticks, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    fprintf(stderr, "[hfmi dc k=%d] ticks: init %lld | z %lld | rank %lld | pair test (+ scan) %lld | rot+gather %lld | secular %lld | loewner %lld | S %lld | gemm + copy %lld | - %lld | sort %lld\n",
            k, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
    fprintf(stderr, "[hfmi dc k=%d] secular: %lld roots, %lld evaluations, at most %lld for one root\n", k, (h[11] >> 20) & 0xfffff, h[11] & 0xfffff, h[11] >> 40);
    fprintf(stderr, "[hfmi tridiag k=%d] wave-0 ticks: (b) slices + barrier %lld | p, dot %lld | update (+ next reflector every 4th step) %lld | waiting for the owner %lld\n",
            k, h[12], h[13], h[14], h[15]);
  }
  return HFMI_OK;
}

// tridiag(dvec, evec) of size n, cut at tree level Lf into 2^Lf independent problems of at most 256 rows (node i covers
// [i n / 2^Lf, (i + 1) n / 2^Lf)): Dout[c] = eigenvalue of column c, the eigenvectors go to the diagonal blocks of Qbig (column-major,
// ldq; the caller has zeroed the rest).  *fail (device) is or-ed with 1 if a secular iteration did not converge.
int launch_dc_leaves(hfmi_ctx* ctx, int n, int Lf, const double* dvec, const double* evec, double* Dout, double* Qbig, int64_t ldq,
                     int* fail) {
  const int nleaf = 1 << Lf;
  int nlmax = 0;
  for (int i = 0; i < nleaf; ++i) nlmax = std::max(nlmax, (int)((((int64_t)(i + 1) * n) >> Lf) - (((int64_t)i * n) >> Lf)));
  if (nlmax > DC_MAXN || nlmax < 2) HFMI_FAIL(HFMI_ERR_INVALID, "dc_leaves: leaf size %d out of range", nlmax);
  const int ldl = (int)round_up(nlmax, 16);
  const size_t mat = (size_t)ldl * ldl;
  const size_t per_leaf_d = 2 * DC_MAXN + 3 * mat;                 // dleaf, dvals, Q, Qt, S
  size_t bytes = (size_t)nleaf * per_leaf_d * sizeof(double);
  const size_t off_perm = bytes;
  bytes += (size_t)nleaf * DC_MAXN * sizeof(int);
  const size_t off_ticks = bytes = round_up(bytes, 16);
  bytes += (size_t)nleaf * 16 * sizeof(long long);
  const size_t off_status = bytes = round_up(bytes, 16);
  bytes += (size_t)nleaf * sizeof(hfmi_status_words);
  const size_t off_args = bytes = round_up(bytes, 16);
  bytes += (size_t)nleaf * sizeof(dc_args);
  const size_t off_desc = bytes = round_up(bytes, 16);
  bytes += (size_t)nleaf * sizeof(leaf_desc);
  const size_t off_zero = bytes = round_up(bytes, 16);
  bytes += 16;
  void* wv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_MISC, bytes, &wv));
  char* base = (char*)wv;
  std::vector<dc_args> args(nleaf);
  std::vector<leaf_desc> desc(nleaf);
  for (int i = 0; i < nleaf; ++i) {
    const int lo = (int)(((int64_t)i * n) >> Lf), hi = (int)(((int64_t)(i + 1) * n) >> Lf), nl = hi - lo;
    double* d0 = (double*)base + (size_t)i * per_leaf_d;
    dc_args& a = args[i];
    a.n = nl;
    a.levels = 0;
    while ((1 << a.levels) < nl) ++a.levels;
    a.sort_by_abs = 0;
    a.dvec = d0;
    a.evec = evec + lo;
    a.sexp = (const int*)(base + off_zero);
    a.Q = d0 + 2 * DC_MAXN;
    a.Qt = a.Q + mat;
    a.S = a.Qt + mat;
    a.ldq = ldl;
    a.dvals = d0 + DC_MAXN;
    a.perm = (int*)(base + off_perm) + (size_t)i * DC_MAXN;
    a.status = (hfmi_status_words*)(base + off_status) + i;
    a.ticks = (long long*)(base + off_ticks) + (size_t)i * 16;
    a.split_top = 0;
    a.top_meta = nullptr;
    const int swaps = a.levels >= 1 ? a.levels - 1 : 0;
    leaf_desc& L = desc[i];
    L.lo = lo;
    L.nl = nl;
    L.ldq = ldl;
    L.dleaf = d0;
    L.Qfin = (swaps & 1) ? a.Qt : a.Q;
    L.dvals = a.dvals;
    L.perm = a.perm;
    L.status = a.status;
  }
  HIP_TRY(hipMemsetAsync(base + off_zero, 0, 16, ctx->stream));
  HIP_TRY(hipMemcpyAsync(base + off_args, args.data(), (size_t)nleaf * sizeof(dc_args), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(base + off_desc, desc.data(), (size_t)nleaf * sizeof(leaf_desc), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));      // args / desc are pageable host memory
  const leaf_desc* ddesc = (const leaf_desc*)(base + off_desc);
  hipLaunchKernelGGL(k_leaf_prep, dim3(nleaf), dim3(256), 0, ctx->stream, ddesc, n, dvec, evec);
  if (nlmax <= 128) hipLaunchKernelGGL((k_dc_batch<8>), dim3(nleaf), dim3(DC_THREADS), 0, ctx->stream, (const dc_args*)(base + off_args));
  else hipLaunchKernelGGL((k_dc_batch<4>), dim3(nleaf), dim3(DC_THREADS), 0, ctx->stream, (const dc_args*)(base + off_args));
  hipLaunchKernelGGL(k_leaf_gather, dim3(nlmax, nleaf), dim3(256), 0, ctx->stream, ddesc, Dout, Qbig, ldq, fail);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
