// tsgemm_ss: C (m x k) = A^T B with BOTH operands skinny (m, k <= 160 columns, m + k <= 288) and the
// reduction axis N long -- the Gram matrices Q^T Q / Q^T B Q, the Rayleigh quotient Q^T (A Q) and the
// first contraction X W of a snapshot/Jacobian operator that holds few rows per rank (SURVEY section 8d:
// "G = X Omega", arithmetic intensity ~ min(m,k)/8 flop/B, HBM-bound below ~55 columns).
//
// In this regime tsgemm_tn (hfmi_gemm.hip) is a poor fit: its streamed operand is read straight from HBM by
// the wave that owns the rows, so with fewer than 16 row tiles most waves idle and the bytes in flight per
// CU are far too few to cover HBM latency.  Here every workgroup owns a contiguous slice of the reduction
// axis and streams ALL columns of both operands through LDS:
//
//   * the whole workgroup (8 waves) issues the global loads of a 32-row stage as 16-byte chunks, 16 lanes per
//     column -> every request is a full 256-byte run of one vector (measured better than 8 columns x 128 B per
//     instruction).  Each byte is read exactly once by exactly one workgroup, so the loads are non-temporal
//     (scripts/stream_probe.hip: +5-10 % HBM throughput for this column-stream pattern).  One or two further
//     stages are in flight in registers while the current one is consumed,
//   * the LDS image of a stage is [16 chunks][columns] in 16-byte elements with NO padding and the column index
//     XOR-ed with (chunk & 7): the fragment read (ds_read_b128, lane (r16, kk) -> chunk it*4+kk, column c0+r16)
//     and the staging write (8 consecutive lanes -> 8 consecutive chunks of one column) are both bank-conflict
//     free (MI355X_MICROARCH.md, LDS lane groups),
//   * the 16x16 output tiles are dealt round-robin to the 8 waves; each wave reads its A and B fragments from
//     LDS (double-buffered in registers under the previous step's MFMAs) -- no operand is privileged, so the
//     kernel does not care which of m, k is smaller,
//   * when A and B are the same block (Q^T Q) it is staged once.
//
// Partial sums of the slices are reduced in a fixed order by k_reduce_partials (deterministic).
#include "hfmi_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
// Column base pointers are typed as GLOBAL pointers: the MFMA-bound variants keep them in an LDS table, and a
// generic pointer read back from LDS would turn the loads into flat_load, which also counts on lgkmcnt and
// serialises against the LDS fragment reads.
typedef const __attribute__((address_space(1))) d2* gd2ptr;
typedef const __attribute__((address_space(1))) double* gdptr;

#ifndef SS_NT
#define SS_NT 1
#endif
#if SS_NT
#define SS_LOAD(p) __builtin_nontemporal_load(p)
#else
#define SS_LOAD(p) (*(p))
#endif
// Timing-only probes of the blocked kernel (round 5, profiles/r05_ssb_ridge_probe.txt; results are garbage; built with
// scripts/build_variant.sh, never in the product library): what would a design gain that streams the LONGER operand HBM -> VGPR
// and keeps only the shorter one in LDS?  SS_PROBE & 1: the staged columns of the second operand are fetched but not written to
// LDS; SS_PROBE & 2: its fragments are not read from LDS either (registers instead) -- together the LDS traffic of that design.
#ifndef SS_PROBE
#define SS_PROBE 0
#endif
constexpr int SS_BK = 32;  // reduction indices per stage (16 chunks of 16 bytes per column)
constexpr int SS_THREADS = 512;

// One 32-row stage of MFMAs for a wave that owns NT tiles.  The A/B fragments of step i+1 are read from LDS
// while the MFMAs of step i run (two register sets); during the last quarter of the steps the wave also
// writes its share of the NEXT stage (held in `r`) to the other LDS buffer, so those stores ride under MFMAs.
// Tile offsets (toa/tob, wave-uniform) live in SGPRs; la / la1 are the per-lane parts for even / odd k-step
// groups.  CT = staged columns.
template <int NT, int TPW, int NQ, bool STORE>
__device__ __forceinline__ void ss_stage(const double* __restrict__ L, double* __restrict__ Lnext, const int la,
                                         const int la1, const int (&toa)[TPW], const int (&tob)[TPW],
                                         d4 (&acc)[TPW], const d2 (&r)[NQ], const int dst0) {
  constexpr int CT = NQ * 32;
  constexpr int S = 4 * NT;                       // steps: it-major, tile-minor
  constexpr int W0 = (S >= 8) ? (3 * S) / 4 : 0;  // first step that carries LDS stores
  constexpr int PER = (NQ + (S - W0) - 1) / (S - W0);
  d2 fa[2], fb[2];
  fa[0] = *reinterpret_cast<const d2*>(L + toa[0] + la);  // step 0: it = 0
  fb[0] = *reinterpret_cast<const d2*>(L + tob[0] + la);
#pragma unroll
  for (int st = 0; st < S; ++st) {
    if (st + 1 < S) {
      const int it = (st + 1) / NT, ti = (st + 1) % NT;
      const int lx = (it & 1) ? la1 : la;  // odd k-step groups sit in chunks 4..7 / 12..15: other XOR pattern
      fa[(st + 1) & 1] = *reinterpret_cast<const d2*>(L + toa[ti] + lx + it * (8 * CT));
      fb[(st + 1) & 1] = *reinterpret_cast<const d2*>(L + tob[ti] + lx + it * (8 * CT));
    }
    if (STORE && st >= W0) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int q = (st - W0) * PER + j;
        if (q < NQ) *reinterpret_cast<d2*>(Lnext + dst0 + q * 64) = r[q];
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // the LDS traffic of this step is issued BEFORE its MFMAs
    const int ti = st % NT;
    acc[ti] = MFMA_F64(fa[st & 1].x, fb[st & 1].x, acc[ti]);
    acc[ti] = MFMA_F64(fa[st & 1].y, fb[st & 1].y, acc[ti]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Column source of staged column v: A columns (clamped to m-1) first, then B columns (clamped to k-1); the
// padding columns v >= ctot repeat the last one and are never read back.
__device__ __forceinline__ gdptr ss_colptr(int v, int ctot, int acols, const double* A, int64_t lda, int m,
                                           const double* B, int64_t ldb, int k) {
  const int vv = v < ctot ? v : ctot - 1;
  const double* p;
  if (vv < acols) {
    const int cc = vv < m ? vv : m - 1;
    p = A + (int64_t)cc * lda;
  } else {
    const int cc = (vv - acols) < k ? (vv - acols) : k - 1;
    p = B + (int64_t)cc * ldb;
  }
  return (gdptr)p;
}

// The stage loop.  PF = global prefetch distance in stages.  PF = 2 (HBM-bound shapes, few tiles per wave): the
// column pointers stay in registers (ptr, chunk offset folded in) and two stages of loads are outstanding per
// workgroup.  PF = 1 (MFMA-bound shapes, whose accumulators need the registers): the column bases are re-read
// from the LDS table every stage and the lane's chunk offset `off` is added.
template <int NT, int TPW, int NQ, int PF>
__device__ __forceinline__ void ss_main(double* __restrict__ lds, const gdptr* __restrict__ ptab,
                                        const gdptr (&ptr)[NQ], const int nstages, const int col0, const int off,
                                        const int dst0, const int la, const int la1, const int (&toa)[TPW],
                                        const int (&tob)[TPW], d4 (&acc)[TPW]) {
  constexpr int BUF = NQ * 32 * SS_BK;  // doubles per LDS stage buffer
  d2 reg[PF][NQ];
  auto stage_load = [&](d2(&r)[NQ], int s) {
    if (s > nstages - 1) s = nstages - 1;  // past the end: re-read the last stage (branch-free, result unused)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if constexpr (PF == 2)
        r[q] = SS_LOAD(reinterpret_cast<gd2ptr>(ptr[q] + (int64_t)s * SS_BK));
      else
        r[q] = SS_LOAD(reinterpret_cast<gd2ptr>(ptab[col0 + 32 * q] + ((int64_t)s * SS_BK + off)));
    }
  };
  if (nstages > 0) {
    stage_load(reg[0], 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<d2*>(lds + dst0 + q * 64) = reg[0][q];
    if constexpr (PF == 2) stage_load(reg[0], 1);
    __syncthreads();
  }
  if constexpr (PF == 1) {
    for (int s = 0; s < nstages; ++s) {
      stage_load(reg[0], s + 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the global loads ahead of the MFMA section
      ss_stage<NT, TPW, NQ, true>(lds + (s & 1) * BUF, lds + ((s + 1) & 1) * BUF, la, la1, toa, tob, acc, reg[0], dst0);
      __syncthreads();
    }
  } else {
    // ONE LDS stage buffer (so that two or three workgroups fit a CU and fill each other's barrier stalls) and two
    // register sets: reg[0] holds stage s+1 (in flight), reg[1] receives stage s+2; two stages per trip keep the
    // names static.  The buffer is rewritten between two barriers once every wave has read its fragments.
    auto stage_store = [&](const d2(&r)[NQ]) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) *reinterpret_cast<d2*>(lds + dst0 + q * 64) = r[q];
    };
    for (int s = 0; s < nstages; s += 2) {
      stage_load(reg[1], s + 2);
      __builtin_amdgcn_sched_barrier(0);
      ss_stage<NT, TPW, NQ, false>(lds, lds, la, la1, toa, tob, acc, reg[0], dst0);
      __syncthreads();
      stage_store(reg[0]);
      __syncthreads();
      if (s + 1 < nstages) {
        stage_load(reg[0], s + 3);
        __builtin_amdgcn_sched_barrier(0);
        ss_stage<NT, TPW, NQ, false>(lds, lds, la, la1, toa, tob, acc, reg[1], dst0);
        __syncthreads();
        stage_store(reg[1]);
        __syncthreads();
      }
    }
  }
}

// TPW: output tiles per wave (compile-time accumulator count), NQ: 16-byte chunks per thread per stage
// (= staged columns / 32), PF: see ss_main.
template <int TPW, int NQ, int PF>
__global__ __launch_bounds__(SS_THREADS, (TPW <= 4 ? 4 : 2)) void k_tsgemm_ss(
    const double* __restrict__ A, int64_t lda, int m, int rt, const double* __restrict__ B, int64_t ldb, int k, int ct,
    int same, int64_t Npad, int64_t chunk, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CT = NQ * 32;                      // staged columns, padded to the load pattern
  constexpr int BUF = CT * SS_BK;                  // doubles per LDS stage buffer
  double* lds = reinterpret_cast<double*>(smem);   // [2][16 chunks][CT] 16-byte elements
  gdptr* ptab = reinterpret_cast<gdptr*>(lds + 2 * BUF);  // [CT] column base pointers (PF == 1 only)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kk = lane >> 4;
  const int acols = rt * 16;
  const int ctot = same ? acols : acols + ct * 16;
  const int bcol0 = same ? 0 : acols;

  const int64_t t_begin = (int64_t)blockIdx.x * chunk;
  int64_t t_end = t_begin + chunk;
  if (t_end > Npad) t_end = Npad;
  const int nstages = t_end > t_begin ? (int)((t_end - t_begin) / SS_BK) : 0;

  // staging map: 16 lanes cover the 16 chunks (256 contiguous bytes) of one column; pass q of this thread:
  // column col0 + 32 q, chunk cq of the stage; LDS element (chunk, column) lives at chunk * CT + (column ^ (chunk & 7))
  const int col0 = tid >> 4;
  const int cq = tid & 15;
  const int dst0 = (cq * CT + (col0 ^ (cq & 7))) * 2;
  gdptr ptr[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) ptr[q] = nullptr;
  if constexpr (PF == 2) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) ptr[q] = ss_colptr(col0 + 32 * q, ctot, acols, A, lda, m, B, ldb, k) + (t_begin + cq * 2);
  } else {
    for (int v = tid; v < CT; v += SS_THREADS) ptab[v] = ss_colptr(v, ctot, acols, A, lda, m, B, ldb, k) + t_begin;
    __syncthreads();
  }

  // this wave's output tiles: linear ids wave, wave+8, ...; only the last one can fall outside the tile list.
  // Two distinct operands: the full rt x ct grid, row-major.  One operand (Gram matrix): only the tiles r <= c,
  // row r holding rt - r of them -- the mirror images are written by the epilogue, so Q^T Q costs rt (rt + 1) / 2
  // instead of rt^2 tiles of MFMAs and is symmetric to the last bit.
  const int ntiles = same ? rt * (rt + 1) / 2 : rt * ct;
  const bool last_valid = wave + 8 * (TPW - 1) < ntiles;
  auto tile_rc = [&](int id, int& r, int& c) {
    if (same) {
      r = 0;
      while (id >= rt - r) {
        id -= rt - r;
        ++r;
      }
      c = r + id;
    } else {
      r = id / ct;
      c = id - r * ct;
    }
  };
  int toa[TPW], tob[TPW];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    int id = wave + 8 * ti;
    if (id > ntiles - 1) id = ntiles - 1;
    int r, c;
    tile_rc(id, r, c);
    toa[ti] = r * 32;
    tob[ti] = (bcol0 + c * 16) * 2;
  }
  const int la = (kk * CT + (r16 ^ kk)) * 2, la1 = (kk * CT + (r16 ^ (kk + 4))) * 2;
  d4 acc[TPW];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) acc[ti] = d4{0.0, 0.0, 0.0, 0.0};

  // the loop is instantiated twice (last tile slot used / unused) OUTSIDE the stage loop, so that the accumulators
  // keep one register assignment throughout
  if (TPW == 1 || last_valid)
    ss_main<TPW, TPW, NQ, PF>(lds, ptab, ptr, nstages, col0, cq * 2, dst0, la, la1, toa, tob, acc);
  else
    ss_main<(TPW > 1 ? TPW - 1 : 1), TPW, NQ, PF>(lds, ptab, ptr, nstages, col0, cq * 2, dst0, la, la1, toa, tob, acc);

  const int kpad = ct * 16;
  double* P = part + (int64_t)blockIdx.x * acols * kpad;
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int id = wave + 8 * ti;
    if (id < ntiles) {
      int r, c;
      tile_rc(id, r, c);
#pragma unroll
      for (int e = 0; e < 4; ++e) P[(int64_t)(r * 16 + kk + 4 * e) * kpad + c * 16 + r16] = acc[ti][e];
      if (same && r != c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) P[(int64_t)(c * 16 + r16) * kpad + r * 16 + kk + 4 * e] = acc[ti][e];
      }
    }
  }
}

// =====================================================================================
// Blocked variant for two distinct operands with at least 2 x 4 output tiles (the MFMA-bound and the transition shapes
// of the north-star kernel point: n = 32 ... 138 snapshots against k = 74 / 84 / 138 probe vectors).
//
// k_tsgemm_ss deals the 16 x 16 output tiles round-robin, so EVERY MFMA pair reads both of its fragments from LDS
// (two ds_read_b128 per 128 matrix-pipe cycles per wave: about half of the LDS pipe).  Here the first
// 4 floor(CT / 4) tile columns are cut into a 2 x 4 grid of rectangular blocks with compile-time shapes, one per wave
// (rows split ceil / floor, columns in four equal groups): a wave with an RB x CB block reads RB + CB fragments per 8
// reduction indices and issues 2 RB CB MFMAs from them (5 x 2 block: 7 reads for 20 MFMAs instead of 20).  Waves w and
// w + 4 share a SIMD and hold the upper and the lower block of one column group, so the four SIMDs carry equal main
// loads; the tiles of the CT mod 4 left-over columns are dealt one at a time to the SIMDs (lower wave first), each
// with its own pair of fragments -- 81 tiles end up as 21 / 20 / 20 / 20 per SIMD.  Staging, LDS image and
// partial-tile output are those of k_tsgemm_ss.
template <int RB, int CB, int EN, int NQ>
__device__ __forceinline__ void ssb_stage(const double* __restrict__ L, double* __restrict__ Lnext, const int la, const int la1,
                                          const int (&toa)[RB], const int (&tob)[CB], const int (&etoa)[EN > 0 ? EN : 1],
                                          const int (&etob)[EN > 0 ? EN : 1], d4 (&acc)[RB][CB], d4 (&eacc)[EN > 0 ? EN : 1],
                                          const d2 (&r)[NQ], const int dst0, const int probe_acols = 0) {
  constexpr int CT = NQ * 32;
  constexpr int EA = EN > 0 ? EN : 1;
  (void)probe_acols;
  d2 fa[2][RB], fb[2][CB], ea[2][EA], eb[2][EA];
#pragma unroll
  for (int i = 0; i < RB; ++i) fa[0][i] = *reinterpret_cast<const d2*>(L + toa[i] + la);
#pragma unroll
  for (int j = 0; j < CB; ++j) fb[0][j] = *reinterpret_cast<const d2*>(L + tob[j] + la);
#pragma unroll
  for (int e = 0; e < EN; ++e) {
    ea[0][e] = *reinterpret_cast<const d2*>(L + etoa[e] + la);
    eb[0][e] = *reinterpret_cast<const d2*>(L + etob[e] + la);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (it + 1 < 4) {
      const int lx = (((it + 1) & 1) ? la1 : la) + (it + 1) * (8 * CT);
#pragma unroll
      for (int i = 0; i < RB; ++i) fa[(it + 1) & 1][i] = *reinterpret_cast<const d2*>(L + toa[i] + lx);
#pragma unroll
      for (int j = 0; j < CB; ++j) {
#if SS_PROBE & 2
        fb[(it + 1) & 1][j] = d2{fb[it & 1][j].y, fb[it & 1][j].x};
#else
        fb[(it + 1) & 1][j] = *reinterpret_cast<const d2*>(L + tob[j] + lx);
#endif
      }
#pragma unroll
      for (int e = 0; e < EN; ++e) {
        ea[(it + 1) & 1][e] = *reinterpret_cast<const d2*>(L + etoa[e] + lx);
        eb[(it + 1) & 1][e] = *reinterpret_cast<const d2*>(L + etob[e] + lx);
      }
    }
    if (it >= 2) {   // the next stage goes to the other LDS buffer under the MFMAs of the last two groups
      constexpr int H = (NQ + 1) / 2;
#pragma unroll
      for (int q = (it - 2) * H; q < ((it - 2) * H + H < NQ ? (it - 2) * H + H : NQ); ++q) {
#if SS_PROBE & 1
        if ((int)(threadIdx.x >> 4) + 32 * q >= probe_acols) {
          asm volatile("" ::"v"(r[q].x), "v"(r[q].y));
          continue;
        }
#endif
        *reinterpret_cast<d2*>(Lnext + dst0 + q * 64) = r[q];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[i][j] = MFMA_F64(fa[it & 1][i].x, fb[it & 1][j].x, acc[i][j]);
#pragma unroll
    for (int e = 0; e < EN; ++e) eacc[e] = MFMA_F64(ea[it & 1][e].x, eb[it & 1][e].x, eacc[e]);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[i][j] = MFMA_F64(fa[it & 1][i].y, fb[it & 1][j].y, acc[i][j]);
#pragma unroll
    for (int e = 0; e < EN; ++e) eacc[e] = MFMA_F64(ea[it & 1][e].y, eb[it & 1][e].y, eacc[e]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Software-pipelined form of ssb_stage (round 3): the stage's barrier sits between the third and the fourth group of eight
// reduction indices instead of after the fourth.  The next stage's LDS image is complete by then (its stores moved to the
// first two groups), so the fragments of the NEXT stage's first group are fetched under the MFMAs of this stage's last
// group, and a stage no longer begins with every wave waiting for its first ds_read_b128s (PMC: the n = 64 ... 138 shapes
// kept the MFMA pipe 75-78 % busy at full clock, i.e. stalled, not power-limited).  fa / fb / ea / eb slot 0 hold the first
// group's fragments on entry and the next stage's on exit.  Reads of the current buffer issued before the barrier are ahead
// of any later store to it in the LDS queue, so the buffer can be refilled by the stage after next without another barrier.
template <int RB, int CB, int EN, int NQ>
__device__ __forceinline__ void ssb_stage_pipe(const double* __restrict__ L, double* __restrict__ Lnext, const int la, const int la1,
                                               const int (&toa)[RB], const int (&tob)[CB], const int (&etoa)[EN > 0 ? EN : 1],
                                               const int (&etob)[EN > 0 ? EN : 1], d4 (&acc)[RB][CB], d4 (&eacc)[EN > 0 ? EN : 1],
                                               const d2 (&r)[NQ], const int dst0, d2 (&fa)[2][RB], d2 (&fb)[2][CB],
                                               d2 (&ea)[2][EN > 0 ? EN : 1], d2 (&eb)[2][EN > 0 ? EN : 1], const int probe_acols = 0) {
  constexpr int CT = NQ * 32;
  (void)probe_acols;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (it == 3) __syncthreads();
    {
      // groups 1..3 of this stage from L; after the barrier the first group of the next stage from Lnext
      const double* src = (it + 1 < 4) ? L : Lnext;
      const int g = (it + 1) & 3;
      const int lx = ((g & 1) ? la1 : la) + g * (8 * CT);
#pragma unroll
      for (int i = 0; i < RB; ++i) fa[(it + 1) & 1][i] = *reinterpret_cast<const d2*>(src + toa[i] + lx);
#pragma unroll
      for (int j = 0; j < CB; ++j) {
#if SS_PROBE & 2
        fb[(it + 1) & 1][j] = d2{fb[it & 1][j].y, fb[it & 1][j].x};
#else
        fb[(it + 1) & 1][j] = *reinterpret_cast<const d2*>(src + tob[j] + lx);
#endif
      }
#pragma unroll
      for (int e = 0; e < EN; ++e) {
        ea[(it + 1) & 1][e] = *reinterpret_cast<const d2*>(src + etoa[e] + lx);
        eb[(it + 1) & 1][e] = *reinterpret_cast<const d2*>(src + etob[e] + lx);
      }
    }
    if (it < 2) {   // the next stage goes to the other LDS buffer under the MFMAs of the FIRST two groups
      constexpr int H = (NQ + 1) / 2;
#pragma unroll
      for (int q = it * H; q < (it * H + H < NQ ? it * H + H : NQ); ++q) {
#if SS_PROBE & 1
        if ((int)(threadIdx.x >> 4) + 32 * q >= probe_acols) {
          asm volatile("" ::"v"(r[q].x), "v"(r[q].y));
          continue;
        }
#endif
        *reinterpret_cast<d2*>(Lnext + dst0 + q * 64) = r[q];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[i][j] = MFMA_F64(fa[it & 1][i].x, fb[it & 1][j].x, acc[i][j]);
#pragma unroll
    for (int e = 0; e < EN; ++e) eacc[e] = MFMA_F64(ea[it & 1][e].x, eb[it & 1][e].x, eacc[e]);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[i][j] = MFMA_F64(fa[it & 1][i].y, fb[it & 1][j].y, acc[i][j]);
#pragma unroll
    for (int e = 0; e < EN; ++e) eacc[e] = MFMA_F64(ea[it & 1][e].y, eb[it & 1][e].y, eacc[e]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ext0 / ext_step: this wave's left-over tiles are the tile numbers ext0, ext0 + ext_step, ... (EN of them) of the
// left-over columns, numbered down the columns: tile e = (row e % RT, column CM + e / RT).
template <int RB, int CB, int EN, int NQ, int PF, int RT, bool PIPE>
__device__ __forceinline__ void ssb_run(double* __restrict__ lds, const gdptr* __restrict__ ptab, const int nstages, const int col0,
                                        const int off, const int dst0, const int la, const int la1, const int row_tile0,
                                        const int col_tile0, const int bcol0, double* __restrict__ P, const int kpad, const int r16,
                                        const int kk, const int ext0, const int ext_col0) {
  constexpr int BUF = NQ * 32 * SS_BK;
  constexpr int EA = EN > 0 ? EN : 1;
  int toa[RB], tob[CB], etoa[EA], etob[EA], er[EA], ec[EA];
#pragma unroll
  for (int i = 0; i < RB; ++i) toa[i] = (row_tile0 + i) * 32;
#pragma unroll
  for (int j = 0; j < CB; ++j) tob[j] = (bcol0 + (col_tile0 + j) * 16) * 2;
#pragma unroll
  for (int e = 0; e < EA; ++e) {
    const int id = ext0 + 8 * e;                  // the same wave is served again two rounds (8 tiles) later
    er[e] = id % RT;
    ec[e] = ext_col0 + id / RT;
    etoa[e] = er[e] * 32;
    etob[e] = (bcol0 + ec[e] * 16) * 2;
  }
  d4 acc[RB][CB], eacc[EA];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int e = 0; e < EA; ++e) eacc[e] = d4{0.0, 0.0, 0.0, 0.0};
  // two register sets where they fit: while stage s is consumed, stage s+1 (already fetched) is written to the other LDS
  // buffer and stage s+2 is on its way -- one stage of MFMAs is shorter than the HBM latency for the small blocks
  d2 reg[PF][NQ];
  auto stage_load = [&](d2(&rr)[NQ], int s) {
    if (s > nstages - 1) s = nstages - 1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) rr[q] = SS_LOAD(reinterpret_cast<gd2ptr>(ptab[col0 + 32 * q] + ((int64_t)s * SS_BK + off)));
  };
  if (nstages > 0) {
    stage_load(reg[0], 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<d2*>(lds + dst0 + q * 64) = reg[0][q];
    if constexpr (PF == 2) stage_load(reg[0], 1);
    __syncthreads();
  }
  if constexpr (PIPE) {
    d2 fa[2][RB], fb[2][CB], ea[2][EA], eb[2][EA];
#pragma unroll
    for (int i = 0; i < RB; ++i) fa[0][i] = *reinterpret_cast<const d2*>(lds + toa[i] + la);
#pragma unroll
    for (int j = 0; j < CB; ++j) fb[0][j] = *reinterpret_cast<const d2*>(lds + tob[j] + la);
#pragma unroll
    for (int e = 0; e < EN; ++e) {
      ea[0][e] = *reinterpret_cast<const d2*>(lds + etoa[e] + la);
      eb[0][e] = *reinterpret_cast<const d2*>(lds + etob[e] + la);
    }
    if constexpr (PF == 2) {
      for (int s = 0; s < nstages; s += 2) {
        stage_load(reg[1], s + 2);
        __builtin_amdgcn_sched_barrier(0);
        ssb_stage_pipe<RB, CB, EN, NQ>(lds, lds + BUF, la, la1, toa, tob, etoa, etob, acc, eacc, reg[0], dst0, fa, fb, ea, eb, RT * 16);
        if (s + 1 < nstages) {
          stage_load(reg[0], s + 3);
          __builtin_amdgcn_sched_barrier(0);
          ssb_stage_pipe<RB, CB, EN, NQ>(lds + BUF, lds, la, la1, toa, tob, etoa, etob, acc, eacc, reg[1], dst0, fa, fb, ea, eb, RT * 16);
        }
      }
    } else {
      for (int s = 0; s < nstages; ++s) {
        stage_load(reg[0], s + 1);
        __builtin_amdgcn_sched_barrier(0);
        ssb_stage_pipe<RB, CB, EN, NQ>(lds + (s & 1) * BUF, lds + ((s + 1) & 1) * BUF, la, la1, toa, tob, etoa, etob, acc, eacc,
                                       reg[0], dst0, fa, fb, ea, eb, RT * 16);
      }
    }
  } else if constexpr (PF == 2) {
    for (int s = 0; s < nstages; s += 2) {
      stage_load(reg[1], s + 2);
      __builtin_amdgcn_sched_barrier(0);
      ssb_stage<RB, CB, EN, NQ>(lds, lds + BUF, la, la1, toa, tob, etoa, etob, acc, eacc, reg[0], dst0, RT * 16);
      __syncthreads();
      if (s + 1 < nstages) {
        stage_load(reg[0], s + 3);
        __builtin_amdgcn_sched_barrier(0);
        ssb_stage<RB, CB, EN, NQ>(lds + BUF, lds, la, la1, toa, tob, etoa, etob, acc, eacc, reg[1], dst0, RT * 16);
        __syncthreads();
      }
    }
  } else {   // large blocks: the accumulators need the registers, one stage in flight
    for (int s = 0; s < nstages; ++s) {
      stage_load(reg[0], s + 1);
      __builtin_amdgcn_sched_barrier(0);
      ssb_stage<RB, CB, EN, NQ>(lds + (s & 1) * BUF, lds + ((s + 1) & 1) * BUF, la, la1, toa, tob, etoa, etob, acc, eacc, reg[0], dst0, RT * 16);
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        P[(int64_t)((row_tile0 + i) * 16 + kk + 4 * e) * kpad + (col_tile0 + j) * 16 + r16] = acc[i][j][e];
#pragma unroll
  for (int x = 0; x < EN; ++x)
#pragma unroll
    for (int e = 0; e < 4; ++e) P[(int64_t)(er[x] * 16 + kk + 4 * e) * kpad + ec[x] * 16 + r16] = eacc[x][e];
}

template <int RT, int CTL, int NQ, bool PIPE>
__global__ __launch_bounds__(SS_THREADS, 2) void k_tsgemm_ssb(const double* __restrict__ A, int64_t lda, int m,
                                                              const double* __restrict__ B, int64_t ldb, int k, int64_t Npad,
                                                              int64_t chunk, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CT = NQ * 32;
  constexpr int BUF = CT * SS_BK;
  constexpr int RB0 = (RT + 1) / 2, RB1 = RT / 2;
  constexpr int CBF = CTL / 4, CM = 4 * CBF, NLEFT = (CTL - CM) * RT;     // main columns, left-over tiles
  static_assert(RB1 >= 1 && CBF >= 1, "blocked variant needs at least 2 x 4 tiles");
  // left-over tile e goes to SIMD e % 4, rounds of four alternate between the lower wave (w >= 4, the smaller block)
  // and the upper one: the lower wave of SIMD i gets e = i, i + 8, ..., the upper wave e = i + 4, i + 12, ...
  constexpr int ENB_HI = (NLEFT + 7) / 8, ENB_LO = NLEFT / 8 + ((NLEFT % 8) > 3 ? 1 : 0);          // lower waves i < / >= NLEFT % 4 ...
  constexpr int ENT_HI = (NLEFT + 3) / 8, ENT_LO = NLEFT > 4 ? (NLEFT - 4) / 8 + (((NLEFT - 4) % 8) > 3 ? 1 : 0) : 0;
  double* lds = reinterpret_cast<double*>(smem);
  gdptr* ptab = reinterpret_cast<gdptr*>(lds + 2 * BUF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kk = lane >> 4;
  constexpr int acols = RT * 16, ctot = (RT + CTL) * 16;
  const int64_t t_begin = (int64_t)blockIdx.x * chunk;
  int64_t t_end = t_begin + chunk;
  if (t_end > Npad) t_end = Npad;
  const int nstages = t_end > t_begin ? (int)((t_end - t_begin) / SS_BK) : 0;
  const int col0 = tid >> 4, cq = tid & 15;
  const int dst0 = (cq * CT + (col0 ^ (cq & 7))) * 2;
  for (int v = tid; v < CT; v += SS_THREADS) ptab[v] = ss_colptr(v, ctot, acols, A, lda, m, B, ldb, k) + t_begin;
  __syncthreads();
  const int la = (kk * CT + (r16 ^ kk)) * 2, la1 = (kk * CT + (r16 ^ (kk + 4))) * 2;
  const int rg = wave >> 2, cg = wave & 3;
  const int row_tile0 = rg ? RB0 : 0;
  const int col_tile0 = cg * CBF;
  const int ext0 = rg ? cg : cg + 4;               // first left-over tile of this wave
  int en = 0;                                      // how many it has: ext0, ext0 + 8, ... < NLEFT
  for (int id = ext0; id < NLEFT; id += 8) ++en;
  constexpr int kpad = CTL * 16;
  double* P = part + (int64_t)blockIdx.x * acols * kpad;
  // two register stages while a stage of MFMAs (tiles per SIMD x 512 cycles) is shorter than about 3.5 us of HBM latency under
  // load; the larger blocks need the registers for their accumulators
  constexpr int PF = (RT * CTL <= 56) ? 2 : 1;
#define SSB_RUN(RBV, ENV) ssb_run<RBV, CBF, ENV, NQ, PF, RT, PIPE>(lds, ptab, nstages, col0, cq * 2, dst0, la, la1, row_tile0, col_tile0, acols, P, kpad, r16, kk, ext0, CM)
  if (rg == 0) {
    if (ENT_HI != ENT_LO && en == ENT_LO) SSB_RUN(RB0, ENT_LO);
    else SSB_RUN(RB0, ENT_HI);
  } else {
    if (ENB_HI != ENB_LO && en == ENB_LO) SSB_RUN(RB1, ENB_LO);
    else SSB_RUN(RB1, ENB_HI);
  }
#undef SSB_RUN
}

static bool ssb_pipe();
template <int RT, int CTL>
static int ssb_launch(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* B, int64_t ldb, int k, int64_t Npad,
                      int64_t chunk, int nsplit, double* part) {
  constexpr int NQ = (RT + CTL + 1) / 2;
  const size_t shmem = 2 * (size_t)NQ * 32 * SS_BK * sizeof(double) + (size_t)NQ * 32 * sizeof(double*);
  // pipelined stages where the fragments fit next to two register stages (measured, scripts/ss_shapes.py: n = 32 / 48 / 64 at
  // k = 138 +4 / +2.5 / +2 %; the 9 x 9 tile shape loses 10 % to the registers the carried fragments cost)
  if (ssb_pipe() && RT * CTL <= 56) {
    auto kern = k_tsgemm_ssb<RT, CTL, NQ, true>;
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kern, dim3(nsplit), dim3(SS_THREADS), shmem, ctx->stream, A, lda, m, B, ldb, k, Npad, chunk, part);
  } else {
    auto kern = k_tsgemm_ssb<RT, CTL, NQ, false>;
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kern, dim3(nsplit), dim3(SS_THREADS), shmem, ctx->stream, A, lda, m, B, ldb, k, Npad, chunk, part);
  }
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}
static int g_ss_blocked = 1;   // A/B knob "ss_blocked": 0 = round-robin kernel, 1 = blocked, 2 = blocked without the stage pipelining
void tsgemm_ss_set_blocked(int v) { g_ss_blocked = v; }
static bool ssb_pipe() { return g_ss_blocked != 2; }
// rt <= ct after the caller's role swap
static bool ssb_has_instance(int rt, int ct) { return rt >= 2 && rt <= ct && (ct == 5 || ct == 6 || ct == 9) && rt + ct <= 18; }
static int ssb_dispatch(hfmi_ctx* ctx, int rt, int ct, const double* A, int64_t lda, int m, const double* B, int64_t ldb, int k,
                        int64_t Npad, int64_t chunk, int nsplit, double* part) {
#define SSB_CASE(R, Cc) \
  if (rt == R && ct == Cc) return ssb_launch<R, Cc>(ctx, A, lda, m, B, ldb, k, Npad, chunk, nsplit, part);
  SSB_CASE(2, 5) SSB_CASE(3, 5) SSB_CASE(4, 5) SSB_CASE(5, 5)
  SSB_CASE(2, 6) SSB_CASE(3, 6) SSB_CASE(4, 6) SSB_CASE(5, 6) SSB_CASE(6, 6)
  SSB_CASE(2, 9) SSB_CASE(3, 9) SSB_CASE(4, 9) SSB_CASE(5, 9) SSB_CASE(6, 9) SSB_CASE(7, 9) SSB_CASE(8, 9) SSB_CASE(9, 9)
#undef SSB_CASE
  HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_ssb: no instance for %d x %d tiles", rt, ct);
}

constexpr int ss_pf(int tpw, int nq) { return (tpw <= 4 && nq <= 6) ? 2 : 1; }

template <int TPW, int NQ>
static int ss_launch(hfmi_ctx* ctx, const double* A, int64_t lda, int m, int rt, const double* B, int64_t ldb, int k,
                     int ct, int same, int64_t Npad, int64_t chunk, int nsplit, double* part, size_t shmem) {
  auto kern = k_tsgemm_ss<TPW, NQ, ss_pf(TPW, NQ)>;
  HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kern, dim3(nsplit), dim3(SS_THREADS), shmem, ctx->stream, A, lda, m, rt, B, ldb, k, ct, same, Npad,
                     chunk, part);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

static int g_ss_percu = 2;  // cap on resident workgroups per CU used to size the grid (A/B knob "ss_percu")
void tsgemm_ss_set_percu(int v) { g_ss_percu = v < 1 ? 1 : v; }

bool tsgemm_ss_applicable(int m, int k, bool same) {
  const int rt = (m + 15) / 16, ct = (k + 15) / 16;
  if (rt < 1 || ct < 1 || rt > 10 || ct > 10) return false;
  const int ctot = same ? rt * 16 : (rt + ct) * 16;
  return ctot <= 288;
}

int launch_tsgemm_ss(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* B, int64_t ldb, int k,
                     int64_t N, double scale, double beta, double* C, int64_t rs, int64_t cs, int nsplit_req) {
  const bool same = (A == B && lda == ldb && m == k);
  const int rt = (m + 15) / 16, ct = (k + 15) / 16;
  const int ctot = same ? rt * 16 : (rt + ct) * 16;
  const int64_t Npad = round_up(N, SS_BK);
  if (lda % 32 != 0 || ldb % 32 != 0 || lda < Npad || ldb < Npad)
    HFMI_FAIL(HFMI_ERR_INVALID, "tsgemm_ss: leading dimensions must be multiples of 32 and >= round_up(N,32)");
  const bool swap = !same && rt > ct;                      // the blocked variant wants rt <= ct: exchange the operand roles
  const bool blocked = !same && g_ss_blocked && ssb_has_instance(swap ? ct : rt, swap ? rt : ct);
  const int tpw = ((same ? rt * (rt + 1) / 2 : rt * ct) + 7) / 8;
  const int nq = (ctot + 31) / 32;  // staged columns are padded to 32 (one 16-byte chunk per thread per 32 columns)
  // unpadded stage buffers: one for the HBM-bound variants (PF = 2), two plus the column pointer table otherwise
  const size_t stage_bytes = (size_t)nq * 32 * SS_BK * sizeof(double);
  const size_t shmem = (ss_pf(tpw, nq) == 2 && !blocked) ? stage_bytes : 2 * stage_bytes + (size_t)nq * 32 * sizeof(double*);
  // workgroups resident per CU: LDS (160 KB) and registers (TPW <= 4 compiles for 4 waves per SIMD = 2 workgroups)
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  int per_cu = (int)((160 * 1024) / shmem);
  const int reg_cap = (tpw <= 4 && !blocked) ? g_ss_percu : 1;
  if (per_cu > reg_cap) per_cu = reg_cap;
  if (per_cu < 1) per_cu = 1;
  const int64_t stages = Npad / SS_BK;
  int nsplit = nsplit_req > 0 ? nsplit_req : cus * per_cu;
  if (nsplit > stages / 2) nsplit = (int)(stages / 2);
  if (nsplit < 1) nsplit = 1;
  int64_t chunk = round_up((Npad + nsplit - 1) / nsplit, SS_BK);
  nsplit = (int)((Npad + chunk - 1) / chunk);
  const int mpad = rt * 16, kpad = ct * 16;
  void* partv = nullptr;
  HFMI_TRY(ctx_ws(ctx, WS_PART, (size_t)nsplit * mpad * kpad * sizeof(double), &partv));
  double* part = (double*)partv;
  const int pidx = prof_start(ctx, 0, m, k, N);
  if (same && pidx >= 0) {   // symmetric output from ONE operand: algorithmic work N k (k + 1) flops, 8 (N k + k^2) bytes
    ctx->prof[pidx].flops = (double)N * k * (k + 1);
    ctx->prof[pidx].bytes = 8.0 * ((double)N * k + (double)k * k);
  }
  if (blocked) {
    if (swap)
      HFMI_TRY(ssb_dispatch(ctx, ct, rt, B, ldb, k, A, lda, m, Npad, chunk, nsplit, part));
    else
      HFMI_TRY(ssb_dispatch(ctx, rt, ct, A, lda, m, B, ldb, k, Npad, chunk, nsplit, part));
    prof_stop(ctx, pidx);
    // swapped roles: the partial tiles hold (A^T B)^T = B^T A, k x m with row stride mpad
    return swap ? launch_reduce_partials(ctx, part, nsplit, (int64_t)mpad * kpad, mpad, true, m, k, scale, beta, C, rs, cs)
                : launch_reduce_partials(ctx, part, nsplit, (int64_t)mpad * kpad, kpad, false, m, k, scale, beta, C, rs, cs);
  }
  int rc = HFMI_ERR_INVALID;
#define SS_CASE(T, Q)                                                                                              \
  if (tpw == T && nq == Q)                                                                                         \
    rc = ss_launch<T, Q>(ctx, A, lda, m, rt, B, ldb, k, ct, same ? 1 : 0, Npad, chunk, nsplit, part, shmem);      \
  else
  // every (tiles per wave, chunks per thread) pair reachable with rt, ct <= 10 and ctot <= 288
  SS_CASE(1, 1) SS_CASE(1, 2) SS_CASE(1, 3) SS_CASE(1, 4) SS_CASE(1, 5) SS_CASE(2, 2) SS_CASE(2, 3) SS_CASE(2, 4)
  SS_CASE(2, 5) SS_CASE(2, 6) SS_CASE(3, 5) SS_CASE(3, 6) SS_CASE(4, 3) SS_CASE(4, 5) SS_CASE(4, 6) SS_CASE(4, 7)
  SS_CASE(5, 3) SS_CASE(5, 6) SS_CASE(5, 7) SS_CASE(6, 7) SS_CASE(7, 4) SS_CASE(7, 7) SS_CASE(7, 8) SS_CASE(8, 4)
  SS_CASE(8, 8) SS_CASE(9, 9) SS_CASE(10, 9) SS_CASE(11, 5) SS_CASE(11, 9) SS_CASE(13, 5)
  // one-operand (Gram) tile lists: rt (rt + 1) / 2 tiles, rt * 16 staged columns
  SS_CASE(3, 3) SS_CASE(4, 4) SS_CASE(5, 4) SS_CASE(6, 5) SS_CASE(7, 5)
  { hfmi_set_error("tsgemm_ss: no instance for tiles/wave=%d chunks/thread=%d", tpw, nq); }
#undef SS_CASE
  HFMI_TRY(rc);
  prof_stop(ctx, pidx);
  return launch_reduce_partials(ctx, part, nsplit, (int64_t)mpad * kpad, kpad, false, m, k, scale, beta, C, rs, cs);
}
