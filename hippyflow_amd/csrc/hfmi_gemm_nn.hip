// tsgemm_nn: Y (N x r) = A S on the fp64 matrix cores -- the long axis is preserved (X^T G, Q R^{-1}, U = Q V).
// Fragment maps and the overall design are described at the top of hfmi_gemm.hip; this half lives in its own
// translation unit so that the two instance families compile in parallel.
#include "hfmi_gemm_common.h"
#include <string.h>

static int g_nn_waves = 0;    // 0 = auto (4 for <= 6 column tiles, else 8)
static int g_nn_hybrid = 1;   // split only the row tiles beyond the last full round of CUs
static int g_nn_tt = 0;       // A/B: force the nn wave-tile height (1 = tallest, 2, 3 = next smaller)
static int g_rem4 = 1;        // last column tile of <= 12 columns in 4-column groups (4x4x4 MFMA)
static int g_nn_res = 1;      // small matrix resident in LDS + persistent workgroups when it fits
static int g_nn_halve_last = 0;   // overlapped rank reduction: last round of a 2-3 round product as two launches of half-height tiles
                                  // (off: on one GPU the shorter tiles cost more than the smaller exposed panel saves, profiles/r04i_halve_last_ab.txt)
static int g_nn_upper = 1;    // Q R^-1: skip the structurally zero column tiles of the upper-triangular small matrix (A/B: "nn_upper")
static int g_nn_res_tt = 0;   // 0: tile height of nn_res by the round count (below); 1: always the table's; 2: always one less (A/B)
int nn_tuning_set(const char* key, int value) {
  if (!strcmp(key, "rem4") && (value == 0 || value == 1)) g_rem4 = value;
  else if (!strcmp(key, "nn_waves") && (value == 0 || value == 4 || value == 8)) g_nn_waves = value;
  else if (!strcmp(key, "nn_tt") && value >= 0 && value <= 3) g_nn_tt = value;
  else if (!strcmp(key, "nn_hybrid") && (value == 0 || value == 1)) g_nn_hybrid = value;
  else if (!strcmp(key, "nn_res") && (value == 0 || value == 1)) g_nn_res = value;
  else if (!strcmp(key, "nn_halve_last") && (value == 0 || value == 1)) g_nn_halve_last = value;
  else if (!strcmp(key, "nn_res_tt") && value >= 0 && value <= 2) g_nn_res_tt = value;
  else if (!strcmp(key, "nn_upper") && (value == 0 || value == 1)) g_nn_upper = value;
  else return 0;
  return 1;
}

// =====================================================================================
// tsgemm_nn
// =====================================================================================
constexpr int NN_KC = 32;  // reduction indices per LDS stage (8 MFMA k-steps)
#ifdef HFMI_NN_NT
#define NN_LOAD_A(p) __builtin_nontemporal_load(p)
#else
#define NN_LOAD_A(p) (*(p))
#endif

// Preconditions: lda multiple of 32 and >= round_up(N,32) (rows beyond N readable); S finite, ld even.
// The reduction axis m may be split over gridDim-many workgroups (msplit > 1): each split writes a raw partial
// block and k_reduce_nn adds them in a fixed order -- this is what balances the grid over the 256 CUs when
// there are only a few row tiles (quantisation), at the price of msplit * N * r * 16 bytes of extra traffic.
// R4 > 0: the last column tile has at most 4 R4 real columns and is computed in R4 groups of 4 columns with the
// 4x4x4 MFMA (see k_tsgemm_tn): the streamed fragment (lane (c16, kk) = long-axis row c16, reduction index kk) is its B
// operand with block g = rows 4g..4g+3, the small-matrix fragment (column 4 q + (lane & 3), the same for every block)
// its A operand, and lane 16 i + 4 g + j receives Y[row 4 g + j = c16][column 4 q + i = 4 q + kk].
template <int TT, int NT, int WAVES, int R4>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_tsgemm_nn(const double* __restrict__ A, int64_t lda, int m,
                                                                    const double* __restrict__ S, int lds_, int r,
                                                                    double* __restrict__ Y, int64_t ldy, int64_t N,
                                                                    int tail_tiles, int msplit, int mchunk, int64_t pstride,
                                                                    int full_tiles, double* __restrict__ Yfull,
                                                                    int64_t ldfull, int full_base, int tail_base) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);  // [2][NN_KC][SLD]
  constexpr int COLS = NT * 16;
  constexpr int SLD = COLS + ((NT % 2 == 0) ? 16 : 0);  // SLD % 32 == 16: conflict-free ds_read_b64
  constexpr int TP = TT / 2;                            // tile pairs fed by one 16-byte load per lane
  constexpr bool ODD = (TT & 1) != 0;                   // plus one single tile fed by an 8-byte load
  constexpr int TPA = TP > 0 ? TP : 1;
  constexpr int NTHR = WAVES * 64;
  constexpr int CH = NN_KC * COLS / 2;            // 16-byte pairs per stage
  constexpr int NQ = (CH + NTHR - 1) / NTHR;      // pairs per thread
  constexpr int NTF = R4 > 0 ? NT - 1 : NT;       // full 16-column tiles
  constexpr int NTA = NTF > 0 ? NTF : 1;
  constexpr int NR4 = R4 > 0 ? R4 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, kk = lane >> 4;
  // Row tiles [0, full_tiles) are whole rounds of the CUs: one workgroup each over the full reduction range, written
  // straight to the result.  The remaining (fewer than one round of) tiles are split msplit-ways over the reduction
  // axis so that they, too, occupy every CU for 1/msplit of a tile's time; only those rows go through partials.
  // A launch covers the whole tiles [full_base, full_base + full_tiles) and the tail tiles [tail_base, tail_base + tail_tiles):
  // the complete product is one launch (full_base = 0, tail_base = full_tiles) or, when the rank all-reduce of the first row
  // panels is to overlap the rest of the product, a few launches over consecutive tile ranges (same arithmetic per element).
  // whole tiles and tail pieces are remapped over the XCDs separately: one contiguous logical range per XCD would
  // put all the short tail pieces on the last XCDs and leave the whole tiles to the others (full_tiles % 8 == 0)
  const bool whole = (int)blockIdx.x < full_tiles;
  const int logical = whole ? xcd_remap(blockIdx.x, full_tiles) : xcd_remap((int)blockIdx.x - full_tiles, tail_tiles * msplit);
  const int split = whole ? 0 : logical / tail_tiles;
  const int tile = whole ? full_base + logical : tail_base + logical % tail_tiles;
  const int64_t t0 = (int64_t)tile * (16 * TT * WAVES) + wave * (16 * TT);
  const int i_begin = whole ? 0 : split * mchunk;
  int i_end = whole ? m : i_begin + mchunk;
  if (i_end > m) i_end = m;
  const int nstages = (i_end - i_begin + NN_KC - 1) / NN_KC;
  const int64_t tmax = round_up_dev(N, 32) - 2;
  double* Yo = whole ? Yfull : Y + (int64_t)split * pstride;
  const int64_t ldout = whole ? ldfull : ldy;

  // streamed operand: lane (c16, kk) fetches rows t0 + tp*32 + 2*c16 + {0,1} of vector i0 + kk
  int64_t toff[TPA];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    int64_t t = t0 + tp * 32 + 2 * c16;
    toff[tp] = t > tmax ? tmax : t;  // rows >= N are never stored; keep the address legal
  }
  int64_t toff1 = t0 + TP * 32 + c16;
  if (toff1 > tmax + 1) toff1 = tmax + 1;
  // S stage: NN_KC rows x COLS cols as 16-byte pairs, NQ per thread
  int s_row[NQ], s_cp[NQ];
#pragma unroll
  for (int qd = 0; qd < NQ; ++qd) {
    int c = tid + NTHR * qd;
    if (c > CH - 1) c = CH - 1;
    s_row[qd] = c / (COLS / 2);
    s_cp[qd] = c % (COLS / 2);
  }

  d4 acc[TT][NTA];
  double acc4[TT][NR4];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
    for (int nt = 0; nt < NTA; ++nt) acc[tt][nt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NR4; ++q) acc4[tt][q] = 0.0;
  }

  struct AFrag {
    d2 p[TPA];
    double s;
  };
  d2 sreg[NQ];
  auto stage_load = [&](int is) {
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
      int row = is + s_row[qd];
      if (row > m - 1) row = m - 1;
      sreg[qd] = *reinterpret_cast<const d2*>(S + (int64_t)row * lds_ + s_cp[qd] * 2);
    }
  };
  auto stage_store = [&](double* L, int is) {
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
      d2 v = sreg[qd];
      if (is + s_row[qd] >= i_end) v = d2{0.0, 0.0};  // rows past this split's range contribute nothing
      if (CH % NTHR == 0 || tid + NTHR * qd < CH) *reinterpret_cast<d2*>(L + s_row[qd] * SLD + s_cp[qd] * 2) = v;
    }
  };
  auto load_a = [&](AFrag& dst, int i0) {
    int col = i0 + kk;
    if (col > m - 1) col = m - 1;
    const double* p = A + (int64_t)col * lda;
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) dst.p[tp] = NN_LOAD_A(reinterpret_cast<const d2*>(p + toff[tp]));
    if (ODD) dst.s = NN_LOAD_A(p + toff1);
  };
  struct SFrag {
    double f[NTA];   // full tiles: column nt*16 + c16
    double g[NR4];   // 4-column groups of the last tile: column NTF*16 + 4 q + (lane & 3)
  };
  auto ldss = [&](SFrag& sf, const double* L, int ks) {
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt) sf.f[nt] = L[(ks * 4 + kk) * SLD + nt * 16 + c16];
    if constexpr (R4 > 0) {
#pragma unroll
      for (int q = 0; q < R4; ++q) sf.g[q] = L[(ks * 4 + kk) * SLD + NTF * 16 + 4 * q + (lane & 3)];
    }
  };
  auto mma = [&](const AFrag& a, const SFrag& sf) {
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
#pragma unroll
      for (int nt = 0; nt < NTF; ++nt) {
        acc[2 * tp][nt] = MFMA_F64(sf.f[nt], a.p[tp].x, acc[2 * tp][nt]);
        acc[2 * tp + 1][nt] = MFMA_F64(sf.f[nt], a.p[tp].y, acc[2 * tp + 1][nt]);
      }
      if constexpr (R4 > 0) {
#pragma unroll
        for (int q = 0; q < R4; ++q) {
          acc4[2 * tp][q] = MFMA_F64_4(sf.g[q], a.p[tp].x, acc4[2 * tp][q]);
          acc4[2 * tp + 1][q] = MFMA_F64_4(sf.g[q], a.p[tp].y, acc4[2 * tp + 1][q]);
        }
      }
    }
    if (ODD) {
#pragma unroll
      for (int nt = 0; nt < NTF; ++nt) acc[TT - 1][nt] = MFMA_F64(sf.f[nt], a.s, acc[TT - 1][nt]);
      if constexpr (R4 > 0) {
#pragma unroll
        for (int q = 0; q < R4; ++q) acc4[TT - 1][q] = MFMA_F64_4(sf.g[q], a.s, acc4[TT - 1][q]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  stage_load(i_begin);
  stage_store(lds, i_begin);
  __syncthreads();
  // register ring of 4 k-steps for the streamed operand (prefetch distance 3), ping-pong LDS fragments
  AFrag a0, a1, a2, a3;
  SFrag sf0, sf1;
  load_a(a0, i_begin);
  load_a(a1, i_begin + 4);
  load_a(a2, i_begin + 8);
  for (int s = 0; s < nstages; ++s) {
    const int is = i_begin + s * NN_KC;
    const bool has_next = s + 1 < nstages;
    if (has_next) stage_load(is + NN_KC);
    const double* L = lds + (s & 1) * NN_KC * SLD;
    ldss(sf0, L, 0);
    load_a(a3, is + 12);
    ldss(sf1, L, 1);
    mma(a0, sf0);
    load_a(a0, is + 16);
    ldss(sf0, L, 2);
    mma(a1, sf1);
    load_a(a1, is + 20);
    ldss(sf1, L, 3);
    mma(a2, sf0);
    load_a(a2, is + 24);
    ldss(sf0, L, 4);
    mma(a3, sf1);
    load_a(a3, is + 28);
    ldss(sf1, L, 5);
    mma(a0, sf0);
    load_a(a0, is + 32);
    ldss(sf0, L, 6);
    mma(a1, sf1);
    load_a(a1, is + 36);
    ldss(sf1, L, 7);
    mma(a2, sf0);
    load_a(a2, is + 40);
    mma(a3, sf1);
    if (has_next) stage_store(lds + ((s + 1) & 1) * NN_KC * SLD, is + NN_KC);
    __syncthreads();
  }

  // Raw accumulator stores only: any VALU arithmetic on the accumulators here makes hipcc keep them in
  // VGPRs across the loop back-edge (256 v_accvgpr copies per stage); scaling/accumulation is done by the caller.
#pragma unroll
  for (int nt = 0; nt < NTF; ++nt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int j = nt * 16 + kk + 4 * rg;
      if (j < r) {
        double* yc = Yo + (int64_t)j * ldout;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          const int64_t t = t0 + tp * 32 + 2 * c16;
          if (t + 1 < N) {
            *reinterpret_cast<d2*>(yc + t) = d2{acc[2 * tp][nt][rg], acc[2 * tp + 1][nt][rg]};
          } else if (t < N) {
            yc[t] = acc[2 * tp][nt][rg];
          }
        }
        if (ODD) {
          const int64_t t = t0 + TP * 32 + c16;
          if (t < N) yc[t] = acc[TT - 1][nt][rg];
        }
      }
    }
  if constexpr (R4 > 0) {
#pragma unroll
    for (int q = 0; q < R4; ++q) {
      const int j = NTF * 16 + 4 * q + kk;
      if (j < r) {
        double* yc = Yo + (int64_t)j * ldout;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          const int64_t t = t0 + tp * 32 + 2 * c16;
          if (t + 1 < N) {
            *reinterpret_cast<d2*>(yc + t) = d2{acc4[2 * tp][q], acc4[2 * tp + 1][q]};
          } else if (t < N) {
            yc[t] = acc4[2 * tp][q];
          }
        }
        if (ODD) {
          const int64_t t = t0 + TP * 32 + c16;
          if (t < N) yc[t] = acc4[TT - 1][q];
        }
      }
    }
  }
}

// k_tsgemm_nn_res: the same product when the small matrix S (m x r) fits LDS whole (m <= ~160: Q R^-1 of the QR
// passes, U = Q V, MvDSmatMult).  The streamed kernel above re-stages S for every row tile and pays its prologue (first
// stage, first fragments) and epilogue (the stores) once per tile with nothing to overlap them -- with m = 138 a tile is
// only five stages long, and config 3's Q R^-1 ran at 0.45 ms where the flops need 0.32.  Here S is staged ONCE per
// workgroup, the workgroups are persistent (grid = number of CUs, row tiles dealt cyclically), there is no barrier
// after the staging, and the streamed operand is prefetched three k-steps ahead ACROSS tile boundaries, so the next
// tile's first fragments are in flight while the current tile's results are being stored.
template <int TT, int NT, int R4, bool UP>
__global__ __launch_bounds__(512, 2) void k_tsgemm_nn_res(const double* __restrict__ A, int64_t lda, int m,
                                                          const double* __restrict__ S, int lds_, int r,
                                                          double* __restrict__ Y, int64_t ldy, int64_t N, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* L = reinterpret_cast<double*>(smem);  // [round_up(m, 4)][SLD]
  constexpr int WAVES = 8;
  constexpr int COLS = NT * 16;
  constexpr int SLD = COLS + ((NT % 2 == 0) ? 16 : 0);  // SLD % 32 == 16: conflict-free ds_read_b64
  constexpr int TP = TT / 2;
  constexpr bool ODD = (TT & 1) != 0;
  constexpr int TPA = TP > 0 ? TP : 1;
  constexpr int NTF = R4 > 0 ? NT - 1 : NT;
  constexpr int NTA = NTF > 0 ? NTF : 1;
  constexpr int NR4 = R4 > 0 ? R4 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, kk = lane >> 4;
  const int mpad = (m + 3) & ~3;
  const int nk = mpad >> 2;  // MFMA k-steps per tile
  // stage S once: rows >= m are zero (the streamed operand is clamped to vector m - 1 there)
  for (int c = tid; c < mpad * (COLS / 2); c += WAVES * 64) {
    const int row = c / (COLS / 2), cp = c - row * (COLS / 2);
    d2 v = d2{0.0, 0.0};
    if (row < m) v = *reinterpret_cast<const d2*>(S + (int64_t)row * lds_ + cp * 2);
    *reinterpret_cast<d2*>(L + row * SLD + cp * 2) = v;
  }
  __syncthreads();

  const int64_t tmax = round_up_dev(N, 32) - 2;
  const int G = gridDim.x;
  int ntl = 0;  // tiles of this workgroup: blockIdx.x, blockIdx.x + G, ...
  if ((int)blockIdx.x < ntiles) ntl = (ntiles - 1 - (int)blockIdx.x) / G + 1;
  const int64_t total = (int64_t)ntl * nk;

  struct AFrag {
    d2 p[TPA];
    double s;
  };
  struct SFrag {
    double f[NTA];
    double g[NR4];
  };
  // prefetch cursor (three k-steps ahead of the MFMAs): tile, k-step, row offsets of that tile
  int ptile = blockIdx.x, pks = 0;
  int64_t ptoff[TPA], ptoff1;
  auto set_ptoff = [&]() {
    const int tl = ptile < ntiles ? ptile : ntiles - 1;   // past the end: re-read the last tile (results unused)
    const int64_t t0 = (int64_t)tl * (16 * TT * WAVES) + wave * (16 * TT);
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
      const int64_t t = t0 + tp * 32 + 2 * c16;
      ptoff[tp] = t > tmax ? tmax : t;
    }
    ptoff1 = t0 + TP * 32 + c16;
    if (ptoff1 > tmax + 1) ptoff1 = tmax + 1;
  };
  set_ptoff();
  auto load_next = [&](AFrag& dst) {
    int col = pks * 4 + kk;
    if (col > m - 1) col = m - 1;
    const double* p = A + (int64_t)col * lda;
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) dst.p[tp] = *reinterpret_cast<const d2*>(p + ptoff[tp]);
    if (ODD) dst.s = p[ptoff1];
    if (++pks == nk) {
      pks = 0;
      ptile += G;
      set_ptoff();
    }
  };
  auto ldss = [&](SFrag& sf, int ks) {
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt) sf.f[nt] = L[(ks * 4 + kk) * SLD + nt * 16 + c16];
    if constexpr (R4 > 0) {
#pragma unroll
      for (int q = 0; q < R4; ++q) sf.g[q] = L[(ks * 4 + kk) * SLD + NTF * 16 + 4 * q + (lane & 3)];
    }
  };

  d4 acc[TT][NTA];
  double acc4[TT][NR4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
      for (int nt = 0; nt < NTA; ++nt) acc[tt][nt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < NR4; ++q) acc4[tt][q] = 0.0;
    }
  };
  // one k-step; the small-matrix fragments are single-buffered: as soon as the MFMAs of column tile nt have issued,
  // its register is refilled with the fragment of k-step ksn and the other tiles' MFMAs cover the LDS latency
  SFrag sf;
  // UP (a template parameter since round 5: as a run-time flag the test cost EVERY instance registers -- U = Q V's <4,4,3> spilled
  // 100 bytes per lane for a branch it never takes): S is upper triangular (S[i][j] = 0 for i > j: R^-1 of the QR).  At reduction step ks (rows 4 ks .. 4 ks + 3 of S)
  // the column tiles nt < ks / 4 hold nothing but zeros: their MFMAs are skipped -- products with exact zeros, so the result
  // has the same bits -- which is 45 % of the matrix instructions at k = 138 (wave-uniform test, one scalar branch per tile).
  auto mma = [&](const AFrag& a, int ksn, int nt0) {
    const double* Ln = L + (ksn * 4 + kk) * SLD;
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt) {
      if (!UP || nt >= nt0) {
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          acc[2 * tp][nt] = MFMA_F64(sf.f[nt], a.p[tp].x, acc[2 * tp][nt]);
          acc[2 * tp + 1][nt] = MFMA_F64(sf.f[nt], a.p[tp].y, acc[2 * tp + 1][nt]);
        }
        if (ODD) acc[TT - 1][nt] = MFMA_F64(sf.f[nt], a.s, acc[TT - 1][nt]);
      }
      sf.f[nt] = Ln[nt * 16 + c16];
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (R4 > 0) {
#pragma unroll
      for (int q = 0; q < R4; ++q) {
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          acc4[2 * tp][q] = MFMA_F64_4(sf.g[q], a.p[tp].x, acc4[2 * tp][q]);
          acc4[2 * tp + 1][q] = MFMA_F64_4(sf.g[q], a.p[tp].y, acc4[2 * tp + 1][q]);
        }
        if (ODD) acc4[TT - 1][q] = MFMA_F64_4(sf.g[q], a.s, acc4[TT - 1][q]);
        sf.g[q] = Ln[NTF * 16 + 4 * q + (lane & 3)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_tile = [&](int tile) {
    const int64_t t0 = (int64_t)tile * (16 * TT * WAVES) + wave * (16 * TT);
    auto put = [&](int j, int tt0, double v0, double v1, bool pair, int64_t t) {
      double* yc = Y + (int64_t)j * ldy;
      if (pair) {
        if (t + 1 < N) *reinterpret_cast<d2*>(yc + t) = d2{v0, v1};
        else if (t < N) yc[t] = v0;
      } else if (t < N) {
        yc[t] = v0;
      }
      (void)tt0;
    };
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int j = nt * 16 + kk + 4 * rg;
        if (j < r) {
#pragma unroll
          for (int tp = 0; tp < TP; ++tp) put(j, tp, acc[2 * tp][nt][rg], acc[2 * tp + 1][nt][rg], true, t0 + tp * 32 + 2 * c16);
          if (ODD) put(j, TT - 1, acc[TT - 1][nt][rg], 0.0, false, t0 + TP * 32 + c16);
        }
      }
    if constexpr (R4 > 0) {
#pragma unroll
      for (int q = 0; q < R4; ++q) {
        const int j = NTF * 16 + 4 * q + kk;
        if (j < r) {
#pragma unroll
          for (int tp = 0; tp < TP; ++tp) put(j, tp, acc4[2 * tp][q], acc4[2 * tp + 1][q], true, t0 + tp * 32 + 2 * c16);
          if (ODD) put(j, TT - 1, acc4[TT - 1][q], 0.0, false, t0 + TP * 32 + c16);
        }
      }
    }
  };

  // prefetch ring of the streamed operand: three k-steps ahead, one for the widest panels (registers)
  constexpr int RD = (TT * NT >= 18) ? 2 : 4;
  AFrag ring[RD];
  zero_acc();
  if (total > 0) {
#pragma unroll
    for (int u = 0; u < RD - 1; ++u) load_next(ring[u]);
    ldss(sf, 0);
  }
  int ctile = blockIdx.x, cks = 0;
  for (int64_t q0 = 0; q0 < total; q0 += RD) {
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      if (q0 + u < total) {   // wave-uniform
        load_next(ring[(u + RD - 1) % RD]);
        mma(ring[u], (cks + 1 == nk) ? 0 : cks + 1, UP ? (cks >> 2) : 0);
        if (++cks == nk) {
          store_tile(ctile);
          zero_acc();
          cks = 0;
          ctile += G;
        }
      }
    }
  }
}

// Y[j][t] = sum_s part[s][j][t]  (fixed order), rows row0 <= t < N (row0 even)
__global__ void k_reduce_nn(const double* __restrict__ part, int msplit, int64_t pstride, int64_t ldp, double* __restrict__ Y,
                            int64_t ldy, int64_t row0, int64_t N, int r) {
  for (int j = blockIdx.y; j < r; j += gridDim.y) {
    const double* p = part + (int64_t)j * ldp;
    double* y = Y + (int64_t)j * ldy;
    for (int64_t t = row0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < N; t += (int64_t)gridDim.x * blockDim.x * 2) {
      if (t + 1 < N) {
        d2 acc = *reinterpret_cast<const d2*>(p + t);
        for (int sp = 1; sp < msplit; ++sp) {
          const d2 v = *reinterpret_cast<const d2*>(p + (int64_t)sp * pstride + t);
          acc.x += v.x;
          acc.y += v.y;
        }
        *reinterpret_cast<d2*>(y + t) = acc;
      } else {
        double acc = p[t];
        for (int sp = 1; sp < msplit; ++sp) acc += p[(int64_t)sp * pstride + t];
        y[t] = acc;
      }
    }
  }
}

// Launch plan of tsgemm_nn for a tile of `tile_rows` rows: how many ways to split the reduction axis m so that the
// grid fills the 256 CUs in (nearly) whole rounds.  Time model (the kernel is MFMA bound, the split partials only
// cost their own HBM round trip in k_reduce_nn): t = flops / (eff * rate) + (msplit + 1) * N * r * 8 / hbm.
static double nn_plan(hfmi_ctx* ctx, int tile_rows, int m, int r, int64_t N, double rate_factor, int* msplit_out) {
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  const int64_t ntiles = (N + tile_rows - 1) / tile_rows;
  const int stages = (m + NN_KC - 1) / NN_KC;
  const double flops = 2.0 * (double)ntiles * tile_rows * (double)m * (double)(((r + 15) / 16) * 16);
  const double rate = 60e12 * rate_factor, hbm = 4.0e12;
  int best = 1;
  double best_t = 1e300;
  for (int ns = 1; ns <= 64; ++ns) {
    if (ns > 1 && stages / ns < 8) break;
    const int64_t blocks = ntiles * ns;
    const int64_t rounds = (blocks + cus - 1) / cus;
    const double eff = (double)blocks / (double)(rounds * cus);
    const double t = flops / (eff * rate) + (ns > 1 ? (ns + 1.0) * (double)N * r * 8.0 / hbm + 3e-6 : 0.0);
    if (t < best_t - 1e-12) {
      best_t = t;
      best = ns;
    }
  }
  *msplit_out = best;
  return best_t;
}

template <int TT, int NT, int WAVES>
static int nn_launch_inst(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r,
                          double* Y, int64_t ldy, int64_t N, int msplit, bool tail_split = false) {
  constexpr int SLD = NT * 16 + ((NT % 2 == 0) ? 16 : 0);
  const size_t shmem = (size_t)2 * NN_KC * SLD * sizeof(double);
  // columns of the last tile: up to 12 are done as 1..3 groups of 4 with the 4x4x4 MFMA
  const int rem = r - (NT - 1) * 16;
  const int r4 = (g_rem4 && rem <= 12) ? (rem + 3) / 4 : 0;
  auto kern = r4 == 1 ? k_tsgemm_nn<TT, NT, WAVES, 1> : r4 == 2 ? k_tsgemm_nn<TT, NT, WAVES, 2>
            : r4 == 3 ? k_tsgemm_nn<TT, NT, WAVES, 3> : k_tsgemm_nn<TT, NT, WAVES, 0>;
  HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const int tile_rows = 16 * TT * WAVES;
  const int ntiles = (int)((N + tile_rows - 1) / tile_rows);
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  // msplit > 1 means the plan found the row tiles badly quantised over the CUs.  With at least one full round of
  // tiles, only the tiles beyond the last full round are split (see the kernel); otherwise every tile is.
  int full_tiles = 0;
  if ((msplit > 1 || tail_split) && ntiles >= cus && g_nn_hybrid) {
    full_tiles = ntiles / cus * cus;
    const int tail = ntiles - full_tiles;
    if (tail == 0) {
      msplit = 1;
    } else {
      const int stages = (m + NN_KC - 1) / NN_KC;
      // split the tail tiles ms ways so that their pieces fill whole rounds of CUs: the tail then costs
      // ceil(tail ms / cus) / ms of a round instead of a whole one (config 3: 162 tail tiles, ms = 3 -> 486 pieces = 2 rounds
      // of a third each = 0.67 of a round; unsplit it was the 8th round of 7.63).  A few tail tiles: one round of short pieces.
      int ms = cus / tail;
      if (ms < 2) {
        double best = 1.0;
        ms = 1;
        for (int c = 2; c <= 8; ++c) {
          const double cost = (double)((tail * c + cus - 1) / cus) / c + 0.01 * c;     // + the partials' round trip
          if (cost < best - 1e-9) {
            best = cost;
            ms = c;
          }
        }
      }
      if (ms > stages / 4) ms = stages / 4;                 // at least four LDS stages per workgroup
      if (ms < 1) ms = 1;
      msplit = ms;
      if (msplit == 1) full_tiles = 0;                      // nothing to split: plain launch
    }
  }
  int mchunk = (int)round_up((m + msplit - 1) / msplit, NN_KC);
  msplit = (m + mchunk - 1) / mchunk;
  double* out = Y;
  int64_t ldo = ldy, pstride = 0;
  if (msplit > 1) {
    ldo = round_up(N, 32);
    pstride = ldo * r;
    void* pv = nullptr;
    HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)msplit * pstride * sizeof(double), &pv));
    out = (double*)pv;
  } else {
    full_tiles = 0;
  }
  const int tail_tiles = ntiles - full_tiles;
  dim3 block(WAVES * 64);
  auto reduce_tail = [&]() -> int {
    const int64_t row0 = (int64_t)full_tiles * tile_rows;   // multiple of 64
    int64_t gx = ((N - row0 + 1) / 2 + 255) / 256;
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_reduce_nn, dim3((unsigned)gx, (unsigned)r), dim3(256), 0, ctx->stream, (const double*)out, msplit,
                       pstride, ldo, Y, ldy, row0, N, r);
    HIP_TRY(hipGetLastError());
    return HFMI_OK;
  };
  // Row panels for an overlapped rank reduction (ctx->nn_hook, set by hfmi_op_apply): whole rounds of tiles per launch, the
  // hook is told which rows are final after each.  Tiles keep the plan of the single launch, so the results are the same bits.
  // A round's tiles finish together, so with R rounds the last panel is 1/R of the block and its reduction is exposed.  When
  // every round already is its own panel and one more panel is allowed, the LAST round is issued as two launches of tiles of
  // HALF the height (same reduction order per row: still the same bits): its first half is final -- and on its way through the
  // fabric -- while the second half is computed, and only a quarter of a two-round product is left exposed.
  if (ctx->nn_hook) {
    const int whole_cnt = msplit > 1 ? full_tiles : ntiles;          // tiles computed in one piece
    const int rounds = whole_cnt / cus;
    if (rounds >= 2) {
      int panels = rounds < ctx->nn_hook_panels ? rounds : ctx->nn_hook_panels;
      if (panels < 1) panels = 1;
      constexpr int TH = TT / 2;
      static const int env_halve = getenv("HFMI_NN_HALVE_LAST") ? atoi(getenv("HFMI_NN_HALVE_LAST")) : -1;   // A/B switch: 1 on, 0 off
      const bool halve = (env_halve >= 0 ? env_halve != 0 : g_nn_halve_last != 0) && (TT % 2 == 0) && TH >= 1 && panels == rounds && panels + 1 <= ctx->nn_hook_panels && panels + 1 <= 8;
      int base = 0;
      for (int p = 0; p < panels; ++p) {
        const bool last = p == panels - 1;
        const int cnt = last ? whole_cnt - base : (rounds / panels + (p < rounds % panels ? 1 : 0)) * cus;
        const int tl = (last && msplit > 1) ? tail_tiles : 0;
        if (last && halve) {
          if constexpr (TT % 2 == 0 && TT >= 2) {
            auto kern_h = r4 == 1 ? k_tsgemm_nn<TH, NT, WAVES, 1> : r4 == 2 ? k_tsgemm_nn<TH, NT, WAVES, 2>
                        : r4 == 3 ? k_tsgemm_nn<TH, NT, WAVES, 3> : k_tsgemm_nn<TH, NT, WAVES, 0>;
            HIP_TRY(hipFuncSetAttribute((const void*)kern_h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            const int cnt_a = cnt / 2, cnt_b = cnt - cnt_a;          // in tiles of the full height
            // first half of the round: 2 cnt_a half-height tiles starting at half-tile 2 base
            hipLaunchKernelGGL(kern_h, dim3((unsigned)(2 * cnt_a)), block, shmem, ctx->stream, A, lda, m, S, lds_, r, Y, ldy, N, 1, 1, mchunk,
                               (int64_t)0, 2 * cnt_a, Y, ldy, 2 * base, 0);
            HIP_TRY(hipGetLastError());
            HFMI_TRY(ctx->nn_hook(ctx->nn_hook_user, Y, ldy, r, (int64_t)base * tile_rows, (int64_t)cnt_a * tile_rows));
            // second half (the last half-height tile may be ragged: rows >= N are never stored), then the split tail tiles
            hipLaunchKernelGGL(kern_h, dim3((unsigned)(2 * cnt_b)), block, shmem, ctx->stream, A, lda, m, S, lds_, r, Y, ldy, N, 1, 1, mchunk,
                               (int64_t)0, 2 * cnt_b, Y, ldy, 2 * (base + cnt_a), 0);
            HIP_TRY(hipGetLastError());
            if (tl > 0) {
              hipLaunchKernelGGL(kern, dim3((unsigned)(tl * msplit)), block, shmem, ctx->stream, A, lda, m, S, lds_, r, out, ldo, N, tl, msplit,
                                 mchunk, pstride, 0, Y, ldy, base, full_tiles);
              HIP_TRY(hipGetLastError());
              HFMI_TRY(reduce_tail());
            }
            const int64_t row0 = (int64_t)(base + cnt_a) * tile_rows;
            HFMI_TRY(ctx->nn_hook(ctx->nn_hook_user, Y, ldy, r, row0, N - row0));
          }
          base += cnt;
          continue;
        }
        if (msplit > 1)
          hipLaunchKernelGGL(kern, dim3((unsigned)(cnt + tl * msplit)), block, shmem, ctx->stream, A, lda, m, S, lds_, r, out, ldo, N,
                             tl > 0 ? tl : 1, msplit, mchunk, pstride, cnt, Y, ldy, base, full_tiles);
        else
          hipLaunchKernelGGL(kern, dim3((unsigned)cnt), block, shmem, ctx->stream, A, lda, m, S, lds_, r, Y, ldy, N, 1, 1, mchunk,
                             (int64_t)0, cnt, Y, ldy, base, 0);
        HIP_TRY(hipGetLastError());
        if (last && msplit > 1) HFMI_TRY(reduce_tail());
        const int64_t row0 = (int64_t)base * tile_rows;
        const int64_t row1 = last ? N : (int64_t)(base + cnt) * tile_rows;
        HFMI_TRY(ctx->nn_hook(ctx->nn_hook_user, Y, ldy, r, row0, row1 - row0));
        base += cnt;
      }
      ctx->nn_hook_called = true;
      return HFMI_OK;
    }
  }
  dim3 grid((unsigned)(full_tiles + tail_tiles * msplit));
  hipLaunchKernelGGL(kern, grid, block, shmem, ctx->stream, A, lda, m, S, lds_, r, out, ldo, N, tail_tiles, msplit, mchunk, pstride,
                     full_tiles, Y, ldy, 0, full_tiles);
  HIP_TRY(hipGetLastError());
  if (msplit > 1) HFMI_TRY(reduce_tail());
  return HFMI_OK;
}

// one-wave-per-SIMD variants: the tile height is chosen among TMAX, TMAX-1, TMAX-2 (16-row tiles per wave) together
// with the reduction split, by the time model above -- a slightly shorter tile often fills the last round of CUs
template <int NT, int TMAX>
static int nn_launch_w4(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double* Y,
                        int64_t ldy, int64_t N) {
  constexpr int T1 = TMAX > 1 ? TMAX - 1 : 1, T2 = TMAX > 2 ? TMAX - 2 : 1;
  int ms0 = 1, ms1 = 1, ms2 = 1;
  const double c0 = nn_plan(ctx, 64 * TMAX, m, r, N, 1.0, &ms0);
  const double c1 = (T1 != TMAX) ? nn_plan(ctx, 64 * T1, m, r, N, 0.98, &ms1) : 1e300;
  const double c2 = (T2 != T1) ? nn_plan(ctx, 64 * T2, m, r, N, 0.96, &ms2) : 1e300;
  // With at least one full round of the tallest tiles the quantisation is handled by splitting only the tail tiles
  // (nn_launch_inst), so the tallest tile -- the best MFMA-to-LDS ratio -- is taken (A/B r01e: config 4 nn 56.5 -> 59.7 TF)
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  if (g_nn_hybrid && g_nn_tt == 0 && (N + 64 * TMAX - 1) / (64 * TMAX) >= cus && m >= 16 * NN_KC)
    return nn_launch_inst<TMAX, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, 1, true);
  if (g_nn_tt == 1) return nn_launch_inst<TMAX, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms0);
  if (g_nn_tt == 2) return nn_launch_inst<T1, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms1);
  if (g_nn_tt == 3) return nn_launch_inst<T2, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms2);
  if (c0 <= c1 && c0 <= c2) return nn_launch_inst<TMAX, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms0);
  if (c1 <= c2) return nn_launch_inst<T1, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms1);
  return nn_launch_inst<T2, NT, 4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms2);
}

template <int NT, int TT>
static int nn_launch_w8(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double* Y,
                        int64_t ldy, int64_t N) {
  int ms = 1;
  nn_plan(ctx, 128 * TT, m, r, N, 1.0, &ms);
  return nn_launch_inst<TT, NT, 8>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, ms);
}

template <int TT, int NT>
static int nn_launch_res(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double* Y,
                         int64_t ldy, int64_t N, size_t shmem) {
  const int rem = r - (NT - 1) * 16;
  const int r4 = (g_rem4 && rem <= 12) ? (rem + 3) / 4 : 0;
  static const bool env_off = getenv("HFMI_NN_UPPER") && atoi(getenv("HFMI_NN_UPPER")) == 0;   // A/B switch
  const bool up = ctx->nn_upper_hint && g_nn_upper && !env_off;
  auto kern = up ? (r4 == 1 ? k_tsgemm_nn_res<TT, NT, 1, true> : r4 == 2 ? k_tsgemm_nn_res<TT, NT, 2, true>
                    : r4 == 3 ? k_tsgemm_nn_res<TT, NT, 3, true> : k_tsgemm_nn_res<TT, NT, 0, true>)
                 : (r4 == 1 ? k_tsgemm_nn_res<TT, NT, 1, false> : r4 == 2 ? k_tsgemm_nn_res<TT, NT, 2, false>
                    : r4 == 3 ? k_tsgemm_nn_res<TT, NT, 3, false> : k_tsgemm_nn_res<TT, NT, 0, false>);
  HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const int tile_rows = 16 * TT * 8;
  const int ntiles = (int)((N + tile_rows - 1) / tile_rows);
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  const int grid = ntiles < cus ? ntiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, ctx->stream, A, lda, m, S, lds_, r, Y, ldy, N, ntiles);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

static int nn_panel(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double* Y,
                    int64_t ldy, int64_t N) {
  const int nt = (r + 15) / 16;
  // S whole in LDS (one workgroup per CU): [round_up(m, 4)][SLD] doubles.  In-place products (Y == A: Q <- Q R^-1) are
  // fine: a workgroup reads the rows of a tile completely before it stores them, and tiles do not overlap.
  if (g_nn_res && nt <= 10 && N >= 4096) {
    const int sld = nt * 16 + ((nt % 2 == 0) ? 16 : 0);
    const size_t shmem = (size_t)((m + 3) & ~3) * sld * sizeof(double);
    if (shmem <= 160 * 1024) {
      // Tile height: the persistent workgroups take whole tiles of 128 TT rows in turn, so the product costs
      // ceil(tiles / CUs) rounds of TT units each.  N = 2e5 with TT = 3 is 521 tiles = 2.03 rounds -> 3 rounds (9 units) where
      // TT = 2 needs 4 rounds of 2 (8 units): the shorter tile is taken when it saves more than the ~5 % its worse
      // MFMA-to-LDS ratio costs.
      const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
      auto units = [&](int tt) {
        const int64_t tiles = (N + 128 * tt - 1) / (128 * tt);
        return (double)((tiles + cus - 1) / cus) * tt;
      };
#define NN_RES(NTV, TTV)                                                                                      \
  case NTV: {                                                                                                 \
    constexpr int TL = TTV > 1 ? TTV - 1 : 1;                                                                 \
    const bool lower = g_nn_res_tt == 2 || (g_nn_res_tt == 0 && TL != TTV && units(TL) * 1.05 < units(TTV)); \
    if (lower) return nn_launch_res<TL, NTV>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, shmem);                   \
    return nn_launch_res<TTV, NTV>(ctx, A, lda, m, S, lds_, r, Y, ldy, N, shmem);                             \
  }
      switch (nt) {
        NN_RES(1, 4) NN_RES(2, 4) NN_RES(3, 4) NN_RES(4, 4) NN_RES(5, 3) NN_RES(6, 2) NN_RES(7, 2) NN_RES(8, 2) NN_RES(9, 2)
        NN_RES(10, 1)
      }
#undef NN_RES
    }
  }
  const int waves = g_nn_waves ? g_nn_waves : (nt >= 7 ? 8 : 4);   // A/B (scripts/nn_waves_ab.py, r01g): one wave per SIMD wins up to 6 column tiles
#define NN_CASE(NTV, TT4, TT8)                                                             \
  case NTV:                                                                                \
    if (waves == 8) return nn_launch_w8<NTV, TT8>(ctx, A, lda, m, S, lds_, r, Y, ldy, N);  \
    return nn_launch_w4<NTV, TT4>(ctx, A, lda, m, S, lds_, r, Y, ldy, N);
  switch (nt) {
    NN_CASE(1, 8, 8) NN_CASE(2, 8, 8) NN_CASE(3, 8, 5) NN_CASE(4, 8, 4) NN_CASE(5, 6, 3) NN_CASE(6, 5, 2)
    NN_CASE(7, 4, 2) NN_CASE(8, 4, 2) NN_CASE(9, 3, 2) NN_CASE(10, 3, 1) NN_CASE(11, 2, 1) NN_CASE(12, 2, 1)
    NN_CASE(13, 2, 1) NN_CASE(14, 2, 1) NN_CASE(15, 2, 1) NN_CASE(16, 2, 1)
  }
#undef NN_CASE
  HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: panel too wide (%d)", r);
}

__global__ void k_small_scale_copy(double* __restrict__ dst, const double* __restrict__ src, int rows, int cols, int ld,
                                   double alpha) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ld) return;
  for (int i = blockIdx.y; i < rows; i += gridDim.y) dst[(int64_t)i * ld + j] = (j < cols) ? alpha * src[(int64_t)i * ld + j] : 0.0;
}

// Y = alpha * A * S + beta * Y.  alpha is folded into a scaled copy of the small matrix; beta goes through a
// scratch block + axpby, so the MFMA kernel itself only ever stores raw accumulators.
int launch_tsgemm_nn(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds_, int r, double alpha,
                     double beta, double* Y, int64_t ldy, int64_t N) {
  if (r <= 0 || N <= 0) return HFMI_OK;
  if (m <= 0) HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: empty reduction");
  if (lds_ % 2 != 0) HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: small-matrix leading dimension must be even");
  if (lda % 32 != 0 || lda < round_up(N, 32))
    HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: leading dimension must be a multiple of 32 and >= round_up(N,32)");
  if (r > 256 && A == Y) HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_nn: in-place update needs r <= 256");
  if (alpha != 1.0) {
    void* sc = nullptr;
    HFMI_TRY(ctx_ws(ctx, WS_MISC, (size_t)m * lds_ * sizeof(double), &sc));
    dim3 block(128), grid((lds_ + 127) / 128, m < 4096 ? m : 4096);
    hipLaunchKernelGGL(k_small_scale_copy, grid, block, 0, ctx->stream, (double*)sc, S, m, r, lds_, alpha);
    HIP_TRY(hipGetLastError());
    S = (const double*)sc;
  }
  double* out = Y;
  int64_t ldo = ldy;
  if (beta != 0.0) {
    void* tmp = nullptr;
    ldo = round_up(N, 32);
    HFMI_TRY(ctx_ws(ctx, WS_STAGE, (size_t)ldo * r * sizeof(double), &tmp));
    out = (double*)tmp;
  }
  for (int r0 = 0; r0 < r; r0 += 256) {
    const int rp = (r - r0 < 256) ? (r - r0) : 256;
    const int pidx = prof_start(ctx, 1, m, rp, N);
    HFMI_TRY(nn_panel(ctx, A, lda, m, S + r0, lds_, rp, out + (int64_t)r0 * ldo, ldo, N));
    prof_stop(ctx, pidx);
  }
  if (beta != 0.0) {
    if (beta != 1.0) HFMI_TRY(launch_scale(ctx, Y, ldy, N, r, beta));
    HFMI_TRY(launch_axpy(ctx, Y, ldy, 1.0, out, ldo, N, r));
  }
  return HFMI_OK;
}
