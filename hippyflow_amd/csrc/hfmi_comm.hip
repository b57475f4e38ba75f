// Sample-parallel communicator behind the C ABI (include/hfmi.h "communicator" section).
//
// Stands in for the reference's mpi4py collective (hippyflow/collectives/collective.py:61-71 `_allReduce_array`,
// :98-111 the per-column Allreduce of a MultiVector, :144-152 the per-column Bcast): ONE collective on the whole
// N x k block in HBM, enqueued on the context's stream, with the 1/P of 'avg' fused into the reduction.
//
// Two device transports, both reducing on the GPU:
//   RCCL  -- ncclAllReduce / ncclBroadcast over xGMI (one process per GPU).  librccl is opened at run time
//            (dlopen) so that single-GPU users of libhfmi.so do not depend on it.
//   P2P   -- for ranks that SHARE a GPU (a functional check of the sharded path on a one-GPU box; RCCL refuses
//            duplicate devices) or on request (HFMI_COMM_TRANSPORT=p2p): every rank owns a staging buffer that
//            its peers map through HIP IPC; rank r sums slice r of all P buffers in rank order and writes the
//            result into slice r of all P buffers (reduce-scatter + all-gather by direct loads / stores: the
//            pattern xGMI's point-to-point links favour).  Ranks meet at barriers in a POSIX shared-memory segment.
// Host payloads (the reference's scalar / numpy all-reduces, collective.py:85-93) go through the same segment.
// A communicator made WITHOUT a context is host-only (CPU tests of the launcher / id exchange / barrier logic).
//
// First contact (round 3): the transport is AGREED, never assumed.  Every rank publishes in the segment whether librccl
// loaded, whether ncclCommInitRank succeeded and whether a first all-reduce gave the right sum within a time-out; all ranks
// read the same table and fall back to P2P together if any entry is bad (hfmi_comm_describe reports what happened and why).
// P2P itself is stream-ordered since round 3: arrival / completion counters in the host segment (mapped into every
// rank's GPU with hipHostRegister) are written and polled by tiny kernels on the context's stream, so a collective costs
// no host synchronisation; the polls give up after the communicator's time-out instead of hanging the GPU.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <vector>

#include "hfmi_internal.h"

// ------------------------------------------------------------------ RCCL, opened at run time
namespace {
typedef struct { char internal[128]; } rccl_unique_id;     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
enum { RCCL_DOUBLE = 8, RCCL_INT8 = 0 };                    // ncclDataType_t: ncclInt8 = 0, ncclFloat64 = 8
enum { RCCL_SUM = 0, RCCL_MAX = 2, RCCL_AVG = 4 };          // ncclRedOp_t
struct rccl_api {
  void* handle;
  int (*GetUniqueId)(rccl_unique_id*);
  int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int);
  int (*CommDestroy)(rccl_comm_t);
  int (*CommAbort)(rccl_comm_t);      // optional
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t);
  int (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t);
  const char* (*GetErrorString)(int);
  char path[512];
};
rccl_api g_rccl = {};
bool g_rccl_tried = false;

bool rccl_load() {
  if (g_rccl.handle) return true;
  if (g_rccl_tried) return false;
  g_rccl_tried = true;
  // HFMI_RCCL_LIB names THE library: if it is set and does not load there is no RCCL (the ranks then agree on p2p)
  const char* named = getenv("HFMI_RCCL_LIB");
  const bool only_named = named && *named;
  const char* cands[3] = {only_named ? named : nullptr, only_named ? nullptr : "librccl.so.1",
                          only_named ? nullptr : "/opt/rocm/lib/librccl.so.1"};
  for (const char* c : cands) {
    if (!c || !*c) continue;
    void* h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (!h) continue;
    g_rccl.GetUniqueId = (int (*)(rccl_unique_id*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(rccl_comm_t*, int, rccl_unique_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(rccl_comm_t))dlsym(h, "ncclCommDestroy");
    g_rccl.CommAbort = (int (*)(rccl_comm_t))dlsym(h, "ncclCommAbort");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclBroadcast");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast &&
        g_rccl.GetErrorString) {
      g_rccl.handle = h;
      snprintf(g_rccl.path, sizeof(g_rccl.path), "%s", c);
      return true;
    }
    dlclose(h);
  }
  return false;
}
#define RCCL_TRY(expr)                                                                                  \
  do {                                                                                                  \
    int _r = (expr);                                                                                    \
    if (_r != 0) HFMI_FAIL(HFMI_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));           \
  } while (0)

// ------------------------------------------------------------------ shared control segment
constexpr int COMM_MAX_RANKS = 16;
constexpr int64_t HOST_AREA_DOUBLES = 8192;     // per rank and chunk (64 KB)
struct comm_slot {
  hipIpcMemHandle_t handle;   // this rank's staging buffer
  int64_t bytes;
  char device_id[64];         // PCI bus id; equal strings = ranks sharing a GPU
  int32_t has_device;
  int32_t rccl_ok;            // librccl loaded and the id carries an RCCL id
  int32_t rccl_init;          // 0 = not tried, 1 = ncclCommInitRank succeeded, 2 = failed
  int32_t rccl_first;         // 0 = not tried, 1 = the first all-reduce gave the right sum, 2 = wrong / error / time-out
  int32_t probe_ok;           // p2p_probe_stage: this rank saw every peer's token in its staging buffer
  int32_t pad2;
};
struct alignas(64) comm_flag {
  unsigned long long v;       // written by ONE rank's GPU (system-scope store), polled by the others' GPUs
  char pad[56];
};
struct comm_shared {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> abort_flag;
  uint32_t pad;
  comm_slot slot[COMM_MAX_RANKS];
  comm_flag arrive[COMM_MAX_RANKS];   // sequence number of the last collective whose contribution rank p has staged
  comm_flag done[COMM_MAX_RANKS];     // ... whose slice rank p has reduced and scattered
  double host_area[COMM_MAX_RANKS][HOST_AREA_DOUBLES];
};
struct id_layout {            // HFMI_UNIQUE_ID_BYTES = 256
  rccl_unique_id rccl;        // zeros when librccl could not be opened on the rank that made the id
  unsigned char token[32];    // names the shared-memory segment
  int32_t rccl_valid;
  int32_t version;
  char pad[256 - 128 - 32 - 8];
};
static_assert(sizeof(id_layout) == HFMI_UNIQUE_ID_BYTES, "id layout");

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
double comm_timeout_s() {
  const char* e = getenv("HFMI_COMM_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0 ? v : 300.0;
}
}  // namespace

enum { TRANSPORT_HOST = 0, TRANSPORT_RCCL = 1, TRANSPORT_P2P = 2 };

struct hfmi_comm {
  hfmi_ctx* ctx;             // null: host-only communicator
  int rank, nranks, transport;
  comm_shared* sh;           // null when the segment is not used (HFMI_COMM_TRANSPORT=rccl)
  rccl_comm_t nccl;
  // P2P staging
  double* stage;             // own buffer
  size_t stage_bytes;
  bool stage_fine;           // allocated fine-grained (peer kernels' writes are visible without a kernel boundary)
  double* peer[COMM_MAX_RANKS];   // mapped staging buffers (peer[rank] == stage)
  comm_shared* sh_dev;       // the segment as the GPU sees it (hipHostRegister), null = host-synchronised P2P
  bool sh_registered;
  unsigned long long seq;    // collectives issued on the stream-ordered path (identical on every rank)
  int* dev_err;              // the error word as the GPU sees it: a poll gave up (time-out) or met a peer's poison
  int* err_host;             // the same word from the host (pinned, mapped): read after any stream synchronisation, no copy
  bool distinct_devices;
  char why[256];             // how the transport was chosen (hfmi_comm_describe)
  int probe_rounds;          // p2p_probe_stage: loop-back rounds the last (re)allocation of the staging buffers needed (1 = first try)
  int probe_retries_total;   // ... rounds beyond the first, over the life of the communicator
  int probe_generations;     // sets of staging buffers the last (re)allocation went through until one passed the probe (1 = the first)
  int probe_regenerations_total;
  // device scratch for host payloads on the RCCL-only route
  double* scratch;
  size_t scratch_bytes;
};

// ------------------------------------------------------------------ barrier in the segment
static int shm_barrier(hfmi_comm* c) {
  comm_shared* sh = c->sh;
  if (!sh || c->nranks == 1) return HFMI_OK;
  const uint32_t gen = sh->generation.load(std::memory_order_acquire);
  if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
    sh->arrived.store(0, std::memory_order_relaxed);
    sh->generation.store(gen + 1, std::memory_order_release);
    return HFMI_OK;
  }
  const double t0 = now_s(), limit = comm_timeout_s();
  for (uint64_t spin = 0;; ++spin) {
    if (sh->generation.load(std::memory_order_acquire) != gen) return HFMI_OK;
    if (sh->abort_flag.load(std::memory_order_relaxed)) HFMI_FAIL(HFMI_ERR_COMM, "communicator aborted by a peer rank");
    if (spin > 2000) {
      timespec ts = {0, 20000};
      nanosleep(&ts, nullptr);
      if ((spin & 1023) == 0 && now_s() - t0 > limit) {
        sh->abort_flag.store(1, std::memory_order_relaxed);
        HFMI_FAIL(HFMI_ERR_COMM, "rank %d: barrier timed out after %.0f s (a peer rank died or never arrived)", c->rank, limit);
      }
    }
  }
}

// ------------------------------------------------------------------ unique id
extern "C" int hfmi_comm_unique_id(void* id_out) {
  if (!id_out) HFMI_FAIL(HFMI_ERR_INVALID, "comm_unique_id: null argument");
  id_layout id;
  memset(&id, 0, sizeof(id));
  id.version = 1;
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0 || read(fd, id.token, sizeof(id.token)) != (ssize_t)sizeof(id.token)) {
    if (fd >= 0) close(fd);
    HFMI_FAIL(HFMI_ERR_COMM, "comm_unique_id: cannot read /dev/urandom");
  }
  close(fd);
  const char* tr = getenv("HFMI_COMM_TRANSPORT");
  const bool want_rccl = !(tr && strcmp(tr, "p2p") == 0) && !(tr && strcmp(tr, "host") == 0);
  int ndev = 0;
  if (want_rccl && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && rccl_load()) {
    if (g_rccl.GetUniqueId(&id.rccl) == 0) id.rccl_valid = 1;
  } else {
    (void)hipGetLastError();
  }
  memcpy(id_out, &id, sizeof(id));
  return HFMI_OK;
}

// rank 0 makes the id and publishes it in `path` (O_EXCL temporary + rename: readers never see a partial file and nobody
// else's file is followed or overwritten); the other ranks wait for the file and accept it only if it is a regular file of
// this user with mode 0600 whose token they have not joined before (a stale file of an earlier communicator of the same
// launch is skipped, not joined).  Rank 0 removes the file as soon as every rank has it (first barrier of init) and no rank
// returns before that (last barrier), so two communicators made back to back cannot meet in the wrong segment.
// The reference's counterpart is mpi4py's own bootstrap; a host that has MPI can instead broadcast the bytes of
// hfmi_comm_unique_id itself and call hfmi_comm_init_rank.
static unsigned char g_last_token[32];
static bool g_have_last_token = false;
static int comm_init_impl(hfmi_ctx* ctx, const void* id_bytes, int nranks, int rank, const char* unlink_path, hfmi_comm** out);

extern "C" int hfmi_comm_init_from_file(hfmi_ctx* ctx, const char* path, int nranks, int rank, hfmi_comm** out) {
  if (!path || !out) HFMI_FAIL(HFMI_ERR_INVALID, "comm_init_from_file: null argument");
  unsigned char id[HFMI_UNIQUE_ID_BYTES];
  if (rank == 0) {
    HFMI_TRY(hfmi_comm_unique_id(id));
    char tmp[1024];
    snprintf(tmp, sizeof(tmp), "%s.tmp.%d", path, (int)getpid());
    (void)unlink(tmp);
    const int fd = open(tmp, O_CREAT | O_EXCL | O_NOFOLLOW | O_WRONLY | O_CLOEXEC, 0600);
    if (fd < 0) HFMI_FAIL(HFMI_ERR_COMM, "comm_init_from_file: cannot create %s: %s", tmp, strerror(errno));
    const ssize_t w = write(fd, id, sizeof(id));
    close(fd);
    if (w != (ssize_t)sizeof(id) || rename(tmp, path) != 0) {
      (void)unlink(tmp);
      HFMI_FAIL(HFMI_ERR_COMM, "comm_init_from_file: cannot publish %s: %s", path, strerror(errno));
    }
  } else {
    const double t0 = now_s(), limit = comm_timeout_s();
    for (;;) {
      const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
      if (fd >= 0) {
        struct stat st;
        bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 077) == 0 &&
                  st.st_size == (off_t)sizeof(id);
        if (ok) ok = read(fd, id, sizeof(id)) == (ssize_t)sizeof(id);
        close(fd);
        if (ok) {
          id_layout probe;
          memcpy(&probe, id, sizeof(probe));
          if (!(g_have_last_token && memcmp(probe.token, g_last_token, sizeof(g_last_token)) == 0)) break;
        }
      }
      if (now_s() - t0 > limit) HFMI_FAIL(HFMI_ERR_COMM, "rank %d: no (fresh, own, mode 0600) communicator id in %s after %.0f s", rank, path, limit);
      timespec ts = {0, 2000000};
      nanosleep(&ts, nullptr);
    }
  }
  return comm_init_impl(ctx, id, nranks, rank, rank == 0 ? path : nullptr, out);
}

// ------------------------------------------------------------------ P2P staging buffers
static int p2p_close_peers(hfmi_comm* c) {
  for (int p = 0; p < c->nranks; ++p) {
    if (p != c->rank && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
    c->peer[p] = nullptr;
  }
  return HFMI_OK;
}
// the staging buffer: fine-grained device memory when the runtime grants it AND exports it (peer GPUs' kernel writes are then
// visible to this GPU without relying on a kernel boundary flushing a remote L2), else ordinary device memory
static int p2p_alloc_stage(hfmi_comm* c, size_t want, bool force_coarse = false) {
  static const bool coarse = env_flag("HFMI_P2P_COARSE");
  c->stage_fine = false;
  if (!coarse && !force_coarse) {
    void* q = nullptr;
    if (hipExtMallocWithFlags(&q, want, hipDeviceMallocFinegrained) == hipSuccess) {
      hipIpcMemHandle_t h;
      if (hipIpcGetMemHandle(&h, q) == hipSuccess) {
        c->stage = (double*)q;
        c->stage_fine = true;
        return HFMI_OK;
      }
      (void)hipFree(q);
    }
    (void)hipGetLastError();
  }
  HIP_TRY(hipMalloc((void**)&c->stage, want));
  return HFMI_OK;
}
// Loop-back check of freshly mapped staging buffers (round 5).  The determinism soak (tests/test_gpu_comm.py, scripts/p2p_soak.py)
// showed the FIRST collective after the staging buffers had been allocated and mapped returning, in ~5 % of the runs, this rank's own
// contribution in the three quarters of the block its peers reduce: their stores through the new HIP-IPC mappings had not reached
// what this rank then read, although their completion flags had (kernel copies and copy commands alike; never once the mappings had
// been used: thousands of later collectives, no case).  So a new set of mappings is exercised before it carries data: every rank
// fills a probe area of its own buffer with its own pattern, every rank then stores a round-specific token into its slot of every
// PEER's probe area through the mapping (the store path of the reduction), and every rank reads its own area back through a kernel
// copy (the load path of the copy-out) and checks all P tokens.  A set of mappings gets at most two rounds (the count is in
// hfmi_comm_describe: "p2p_probe_rounds"); when both fail the caller replaces the buffers, up to three generations -- six failed
// rounds in all are an error, not silent data.  A HIP failure on ONE rank is published in the shared table as probe_ok = -1 and
// that rank still takes every barrier of the round, so that all ranks return the error together instead of its peers waiting for
// the communicator's time-out.
__global__ void __launch_bounds__(256) k_p2p_copy(double* __restrict__ dst, const double* __restrict__ src, int64_t count) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int64_t n2 = count >> 1, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride)
    reinterpret_cast<d2*>(dst)[i] = reinterpret_cast<const d2*>(src)[i];
  if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[count - 1] = src[count - 1];
}
constexpr int PROBE_SLOT = 32;      // doubles per rank in the probe area (256 bytes: two 128-byte lines)
__global__ void k_p2p_probe_fill(double* stage, int nranks, double value) {
  for (int i = threadIdx.x; i < nranks * PROBE_SLOT; i += blockDim.x) stage[i] = value;
}
struct p2p_probe_ptrs {
  double* p[COMM_MAX_RANKS];
};
__global__ void k_p2p_probe_store(p2p_probe_ptrs bufs, int nranks, int rank, double token) {
  const int p = blockIdx.x;
  if (p < nranks && threadIdx.x < PROBE_SLOT) bufs.p[p][rank * PROBE_SLOT + threadIdx.x] = token;
}
static int p2p_probe_stage(hfmi_comm* c, bool* verdict) {
  hfmi_ctx* ctx = c->ctx;
  hipStream_t st = ctx->stream;
  const int P = c->nranks, n = P * PROBE_SLOT;
  double* scratch = nullptr;
  HIP_TRY(hipMalloc((void**)&scratch, (size_t)n * sizeof(double)));
  std::vector<double> host(n);
  p2p_probe_ptrs bufs;
  for (int p = 0; p < COMM_MAX_RANKS; ++p) bufs.p[p] = p < P ? c->peer[p] : nullptr;
  int rounds = 0, rc = HFMI_OK;
  bool all_ok = false;
  bool hip_bad = false, peer_bad = false;       // a HIP call failed here / on some rank
  for (int round = 0; round < 2 && !all_ok && rc == HFMI_OK && !peer_bad; ++round) {
    ++rounds;
    const double own = -1.0 - c->rank - 100.0 * round;
    if (!hip_bad) {
      hipLaunchKernelGGL(k_p2p_probe_fill, dim3(1), dim3(256), 0, st, c->stage, P, own);
      if (hipStreamSynchronize(st) != hipSuccess) hip_bad = true;
    }
    rc = shm_barrier(c);                                         // every rank's own pattern is in place
    if (rc != HFMI_OK) break;
    if (!hip_bad) {
      hipLaunchKernelGGL(k_p2p_probe_store, dim3(P), dim3(64), 0, st, bufs, P, c->rank, 1000.0 * (round + 1) + c->rank);
      if (hipStreamSynchronize(st) != hipSuccess) hip_bad = true;
    }
    rc = shm_barrier(c);                                         // every rank's tokens have been stored (kernels complete)
    if (rc != HFMI_OK) break;
    if (!hip_bad) {
      hipLaunchKernelGGL(k_p2p_copy, dim3(1), dim3(256), 0, st, scratch, (const double*)c->stage, (int64_t)n);
      if (hipMemcpyAsync(host.data(), scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess)
        hip_bad = true;
    }
    bool ok = !hip_bad;
    for (int p = 0; p < P && ok; ++p)
      for (int i = 0; i < PROBE_SLOT; ++i)
        if (host[p * PROBE_SLOT + i] != 1000.0 * (round + 1) + p) {
          ok = false;
          break;
        }
    c->sh->slot[c->rank].probe_ok = hip_bad ? -1 : (ok ? 1 : 0);
    rc = shm_barrier(c);
    if (rc != HFMI_OK) break;
    all_ok = true;
    for (int p = 0; p < P; ++p) {
      all_ok = all_ok && c->sh->slot[p].probe_ok == 1;
      peer_bad = peer_bad || c->sh->slot[p].probe_ok < 0;
    }
    rc = shm_barrier(c);                                         // everybody has read the table before the next round rewrites it
  }
  if (rc == HFMI_OK && (hip_bad || peer_bad)) rc = HFMI_ERR_HIP;
  (void)hipFree(scratch);
  if (rc == HFMI_ERR_HIP)
    hfmi_set_error("p2p staging probe: a HIP call failed on %s: %s", hip_bad ? "this rank" : "another rank", hipGetErrorString(hipGetLastError()));
  if (rc != HFMI_OK) return rc;
  c->probe_rounds = rounds;
  c->probe_retries_total += rounds - 1;
  *verdict = all_ok;
  return HFMI_OK;
}

// Collective: every rank calls it with the same `bytes`.
// A set of staging buffers + mappings whose loop-back probe fails is torn down and replaced (new allocations of another size; the
// third generation as ordinary instead of fine-grained device memory): in the soak runs where the probe failed it failed in EVERY
// round on those mappings (6 process groups in 100), while the buffers a later, larger collective allocated always worked.
static int p2p_ensure_stage(hfmi_comm* c, size_t bytes) {
  if (bytes <= c->stage_bytes) return HFMI_OK;
  // this rank's last copy out of the old buffer is complete -- on BOTH streams collectives run on (row panels of an operator
  // application are reduced on the auxiliary stream)
  HIP_TRY(hipStreamSynchronize(c->ctx->stream));
  HIP_TRY(hipStreamSynchronize(c->ctx->aux_stream));
  static const bool no_probe = env_flag("HFMI_P2P_NO_PROBE");    // A/B: the behaviour up to round 4
  const size_t want = round_up((int64_t)(bytes + bytes / 8), 1 << 20);
  comm_slot& me = c->sh->slot[c->rank];
  for (int gen = 0; gen < 3; ++gen) {
    HFMI_TRY(shm_barrier(c));                       // nobody is still reading the old buffers
    HFMI_TRY(p2p_close_peers(c));
    HFMI_TRY(shm_barrier(c));                       // every mapping of the old buffer is closed before it is freed
    if (c->stage) HIP_TRY(hipFree(c->stage));
    c->stage = nullptr;
    c->stage_bytes = 0;
    // The export of a fresh allocation was seen to fail once with "invalid argument" (rank 0 of four ranks sharing the GPU, right
    // after another four-rank job had ended; an identical run a minute later was fine): a failed export is retried on a new
    // allocation of a slightly different size before it becomes an error.  Every rank publishes the AGREED size `want`.
    hipError_t exp_err = hipSuccess;
    for (int attempt = 0; attempt < 4; ++attempt) {
      const size_t actual = want + ((size_t)(4 * gen + attempt) << 21);
      HFMI_TRY(p2p_alloc_stage(c, actual, gen == 2));
      // the capacity the growth decision above compares with must be the same number on every rank: `attempt` is rank-local
      // (an export retried on some ranks only), so it stays out of it -- the allocation is merely up to 6 MB larger than this
      c->stage_bytes = want + ((size_t)(4 * gen) << 21);
      exp_err = hipIpcGetMemHandle(&me.handle, c->stage);
      {      // test hook: HFMI_P2P_INJECT_EXPORT_FAIL=<rank> makes the FIRST export of that rank fail once (the retry path above runs on
             // one rank only: tests/test_gpu_comm.py checks that the ranks still agree on when the buffers have to grow)
        static const char* inj = getenv("HFMI_P2P_INJECT_EXPORT_FAIL");
        static bool injected = false;
        if (inj && !injected && atoi(inj) == c->rank && exp_err == hipSuccess) {
          injected = true;
          exp_err = hipErrorInvalidValue;
        }
      }
      if (exp_err == hipSuccess) break;
      (void)hipGetLastError();
      (void)hipFree(c->stage);
      c->stage = nullptr;
      c->stage_bytes = 0;
    }
    if (exp_err != hipSuccess) {
      hfmi_set_error("hipIpcGetMemHandle of the p2p staging buffer failed four times: %s", hipGetErrorString(exp_err));
      return HFMI_ERR_HIP;
    }
    me.bytes = (int64_t)want;
    HFMI_TRY(shm_barrier(c));
    for (int p = 0; p < c->nranks; ++p) {
      if (p == c->rank) {
        c->peer[p] = c->stage;
        continue;
      }
      if (c->sh->slot[p].bytes != (int64_t)want)
        HFMI_FAIL(HFMI_ERR_COMM, "rank %d: rank %d staged %lld bytes, expected %lld (collective called with different sizes)",
                  c->rank, p, (long long)c->sh->slot[p].bytes, (long long)want);
      void* q = nullptr;
      HIP_TRY(hipIpcOpenMemHandle(&q, c->sh->slot[p].handle, hipIpcMemLazyEnablePeerAccess));
      c->peer[p] = (double*)q;
    }
    HFMI_TRY(shm_barrier(c));
    if (no_probe) return HFMI_OK;
    bool ok = false;
    HFMI_TRY(p2p_probe_stage(c, &ok));              // the verdict is the same on every rank (read from the shared table)
    c->probe_generations = gen + 1;
    if (ok) return HFMI_OK;
    c->probe_regenerations_total += 1;
  }
  HFMI_FAIL(HFMI_ERR_COMM, "rank %d: stores through the HIP-IPC mappings of the p2p staging buffers did not become visible on three "
            "successive sets of buffers: the transport cannot be trusted on this system", c->rank);
}
struct p2p_ptrs {
  double* p[COMM_MAX_RANKS];
};
// rank r owns the d2 elements [lo, hi): sum over ranks in rank order (every rank ends up with identical bits),
// scale, and store into the same slice of every rank's buffer.
__global__ void __launch_bounds__(256) k_p2p_reduce(p2p_ptrs bufs, int nranks, int64_t lo, int64_t hi, double scale, int op,
                                                    const int* __restrict__ dev_err) {
  // a poll gave up: the peers' data may be missing, leave the buffers alone.  The error word lives in pinned HOST memory: one
  // thread per workgroup reads it over PCIe and hands it on through LDS (every thread reading it was 256 reads per workgroup)
  __shared__ int s_err;
  if (threadIdx.x == 0) s_err = dev_err ? *(volatile const int*)dev_err : 0;
  __syncthreads();
  if (s_err) return;
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride) {
    d2 acc = reinterpret_cast<const d2*>(bufs.p[0])[i];
    for (int p = 1; p < nranks; ++p) {
      const d2 v = reinterpret_cast<const d2*>(bufs.p[p])[i];
      if (op == HFMI_REDUCE_MAX) {
        acc.x = fmax(acc.x, v.x);
        acc.y = fmax(acc.y, v.y);
      } else {
        acc += v;
      }
    }
    acc *= scale;
    for (int p = 0; p < nranks; ++p) reinterpret_cast<d2*>(bufs.p[p])[i] = acc;
  }
}
// stream-ordered hand-shake through the host segment: one thread publishes this rank's sequence number -- or, once this
// rank has given up on a collective, the POISON value: every peer that waits on it raises its own error word, so a
// time-out on one rank can never leave another rank with a silently unreduced block ...
constexpr unsigned long long P2P_POISON = ~0ull;
__global__ void k_p2p_signal(comm_flag* flag, unsigned long long seq, const int* dev_err) {
  __threadfence_system();
  const unsigned long long v = (*(volatile const int*)dev_err) ? P2P_POISON : seq;
  __hip_atomic_store(&flag->v, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and one wave waits until every rank has published at least `seq` (lane p polls rank p).  wall_clock64 runs at a
// constant 100 MHz: the poll gives up after `ticks` and raises the error word instead of occupying the GPU for ever.
__global__ void k_p2p_wait(const comm_flag* flags, int nranks, unsigned long long seq, long long ticks, int* dev_err) {
  const int p = threadIdx.x;
  if (p >= nranks) return;
  const long long t0 = wall_clock64();
  unsigned long long v;
  while ((v = __hip_atomic_load(&flags[p].v, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < seq) {
    if (wall_clock64() - t0 > ticks || *(volatile int*)dev_err) {
      __hip_atomic_store(dev_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // a plain store: the word lives in host memory, no PCIe atomic needed
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  if (v == P2P_POISON) __hip_atomic_store(dev_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // a plain store: the word lives in host memory, no PCIe atomic needed
}

// Copies between a block and the staging buffer.  Up to round 4 these were hipMemcpyAsync (device to device): a copy COMMAND, which
// the runtime may hand to a blit kernel or to an SDMA engine.  The determinism soak of round 5 (tests/test_gpu_comm.py, 200
// all-reduces x 40 runs) caught the FIRST collective on a freshly allocated staging buffer returning, in 2 runs of 40, this rank's
// own contribution in the three quarters of the block its peers had reduced -- the copy-out had not seen the peers' stores although
// their completion flags had been observed.  As plain kernels on the stream the two copies are ordered and made visible by the same
// kernel-boundary acquire / release as the signal, wait and reduce kernels around them.  HFMI_P2P_COPY=memcpy restores the copy
// commands (A/B).
static int p2p_copy(hfmi_comm* c, double* dst, const double* src, int64_t count, hipStream_t stream) {
  static const bool use_memcpy = [] {
    const char* e = getenv("HFMI_P2P_COPY");
    return e && !strcmp(e, "memcpy");
  }();
  if (use_memcpy || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) {
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, stream));
    return HFMI_OK;
  }
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((count / 2 + 255) / 256, (int64_t)c->ctx->num_cus * 8));
  hipLaunchKernelGGL(k_p2p_copy, dim3(blocks), dim3(256), 0, stream, dst, src, count);
  HIP_TRY(hipGetLastError());
  return HFMI_OK;
}

static int p2p_allreduce_dev(hfmi_comm* c, double* data, int64_t count, int op, hipStream_t stream) {
  hfmi_ctx* ctx = c->ctx;
  const int64_t padded = round_up(count, 2);
  const size_t bytes = (size_t)padded * sizeof(double);
  HFMI_TRY(p2p_ensure_stage(c, bytes));
  if (padded != count) HIP_TRY(hipMemsetAsync(c->stage + count, 0, sizeof(double), stream));
  HFMI_TRY(p2p_copy(c, c->stage, data, count, stream));
  const int64_t n2 = padded / 2;
  const int64_t lo = n2 * c->rank / c->nranks, hi = n2 * (c->rank + 1) / c->nranks;
  p2p_ptrs bufs;
  for (int p = 0; p < COMM_MAX_RANKS; ++p) bufs.p[p] = p < c->nranks ? c->peer[p] : nullptr;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((hi - lo + 255) / 256, (int64_t)ctx->num_cus * 8));
  const double scale = (op == HFMI_REDUCE_AVG) ? 1.0 / c->nranks : 1.0;
  if (c->sh_dev) {
    // stream-ordered: no host synchronisation, no host barrier
    const unsigned long long seq = ++c->seq;
    const long long ticks = (long long)(comm_timeout_s() * 1e8);
    hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(1), 0, stream, &c->sh_dev->arrive[c->rank], seq, (const int*)c->dev_err);
    hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, stream, c->sh_dev->arrive, c->nranks, seq, ticks, c->dev_err);
    if (hi > lo) hipLaunchKernelGGL(k_p2p_reduce, dim3(blocks), dim3(256), 0, stream, bufs, c->nranks, lo, hi, scale, op, c->dev_err);
    hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(1), 0, stream, &c->sh_dev->done[c->rank], seq, (const int*)c->dev_err);
    hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, stream, c->sh_dev->done, c->nranks, seq, ticks, c->dev_err);
    HIP_TRY(hipGetLastError());
    HFMI_TRY(p2p_copy(c, data, c->stage, count, stream));
    // the next collective's copy-in is ordered behind this copy-out on the stream, and no peer touches this rank's buffer
    // before this rank has published the next sequence number
    return HFMI_OK;
  }
  HIP_TRY(hipStreamSynchronize(stream));
  HFMI_TRY(shm_barrier(c));                       // every rank's contribution is in its staging buffer
  if (hi > lo) {
    hipLaunchKernelGGL(k_p2p_reduce, dim3(blocks), dim3(256), 0, stream, bufs, c->nranks, lo, hi, scale, op, (const int*)nullptr);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(stream));
  HFMI_TRY(shm_barrier(c));                       // every slice of every buffer is final
  HFMI_TRY(p2p_copy(c, data, c->stage, count, stream));
  // the staging buffer may be overwritten by the next collective only after this copy: the next collective
  // starts with a stream synchronise of its own copy-in, which is ordered behind this copy on the same stream
  return HFMI_OK;
}
// A poll that gave up (or met a peer's poison) leaves a flag in pinned host memory.  Every host synchronisation that can
// follow a collective looks at it (ctx_check_comm: read_back, block download, context synchronise, the barrier); the first
// rank to see it also raises the segment's abort flag, which ends every later barrier of the peers.  The communicator is
// unusable afterwards: the word is deliberately sticky.
int comm_check_error(hfmi_comm* c) {
  if (!c || !c->err_host) return HFMI_OK;
  if (*(volatile int*)c->err_host) {
    if (c->sh) c->sh->abort_flag.store(1, std::memory_order_relaxed);
    HFMI_FAIL(HFMI_ERR_COMM, "rank %d: a peer rank did not reach a collective within %.0f s, or gave up on one (stream-ordered p2p transport); "
              "the block it returned is NOT reduced and the communicator is unusable", c->rank, comm_timeout_s());
  }
  return HFMI_OK;
}
static int p2p_check_err(hfmi_comm* c) { return comm_check_error(c); }
void comm_forget_ctx(hfmi_comm* c) {
  if (c) c->ctx = nullptr;
}

static int p2p_bcast_dev(hfmi_comm* c, void* data, size_t bytes, int root) {
  hfmi_ctx* ctx = c->ctx;
  HFMI_TRY(p2p_ensure_stage(c, bytes));
  if (c->rank == root) {
    HIP_TRY(hipMemcpyAsync(c->stage, data, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  HFMI_TRY(shm_barrier(c));
  if (c->rank != root) {
    HIP_TRY(hipMemcpyAsync(data, c->peer[root], bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return shm_barrier(c);                          // the root may reuse its staging buffer
}

// ------------------------------------------------------------------ transport decision
// One pure function of the table every rank reads from the segment, so that all ranks decide alike.
// force: 0 = auto, 2 = p2p requested.  reason (>= 160 bytes) says why in words.
static int decide_transport(int nranks, const comm_slot* slot, int force, char* reason) {
  bool all_dev = true, any_dev = false, all_rccl = true, distinct = true;
  int first_no_rccl = -1;
  for (int p = 0; p < nranks; ++p) {
    all_dev = all_dev && slot[p].has_device;
    any_dev = any_dev || slot[p].has_device;
    if (!slot[p].rccl_ok) {
      all_rccl = false;
      if (first_no_rccl < 0) first_no_rccl = p;
    }
    for (int q = 0; q < p; ++q)
      if (slot[p].has_device && slot[q].has_device && strcmp(slot[p].device_id, slot[q].device_id) == 0) distinct = false;
  }
  if (!all_dev) {
    if (any_dev) {
      snprintf(reason, 160, "some ranks passed a device context and some did not");
      return -1;
    }
    snprintf(reason, 160, "no rank has a device context: host payloads only");
    return TRANSPORT_HOST;
  }
  if (force == 2) {
    snprintf(reason, 160, "p2p requested (HFMI_COMM_TRANSPORT=p2p)");
    return TRANSPORT_P2P;
  }
  if (!distinct) {
    snprintf(reason, 160, "ranks share a GPU (RCCL refuses duplicate devices)");
    return TRANSPORT_P2P;
  }
  if (!all_rccl) {
    snprintf(reason, 160, "librccl not usable on rank %d (HFMI_RCCL_LIB / librccl.so.1 did not load, or the id carries no RCCL id)", first_no_rccl);
    return TRANSPORT_P2P;
  }
  snprintf(reason, 160, "every rank has its own GPU and librccl");
  return TRANSPORT_RCCL;
}
// test hook (CPU suite): the decision for a hand-made table.  has_device / rccl_ok: nranks ints; device_ids: nranks strings.
extern "C" int hfmi_comm_decide_transport(int nranks, const int* has_device, const int* rccl_ok, const char* const* device_ids,
                                          int force_p2p, int* transport, char* reason, int reason_len) {
  if (nranks < 1 || nranks > COMM_MAX_RANKS || !has_device || !rccl_ok || !device_ids || !transport)
    HFMI_FAIL(HFMI_ERR_INVALID, "comm_decide_transport: bad argument");
  comm_slot slot[COMM_MAX_RANKS];
  memset(slot, 0, sizeof(slot));
  for (int p = 0; p < nranks; ++p) {
    slot[p].has_device = has_device[p];
    slot[p].rccl_ok = rccl_ok[p];
    snprintf(slot[p].device_id, sizeof(slot[p].device_id), "%s", device_ids[p] ? device_ids[p] : "");
  }
  char why[160];
  *transport = decide_transport(nranks, slot, force_p2p ? 2 : 0, why);
  if (reason && reason_len > 0) snprintf(reason, (size_t)reason_len, "%s", why);
  return HFMI_OK;
}

// ------------------------------------------------------------------ init / destroy
static void comm_free(hfmi_comm* c) {
  if (!c) return;
  if (c->ctx) ctx_unwatch_comm(c->ctx, c);
  if (c->sh_registered) (void)hipHostUnregister(c->sh);
  if (c->err_host) (void)hipHostFree(c->err_host);
  if (c->sh) munmap(c->sh, sizeof(comm_shared));
  delete c;
}
// stream-ordered P2P needs the segment mapped into the GPU's address space and a device error word
static int p2p_enable_stream_order(hfmi_comm* c) {
  static const char* mode = getenv("HFMI_P2P_SYNC");      // "host" | "stream" | unset = stream when the GPUs are distinct
  const bool want = mode ? strcmp(mode, "stream") == 0 : c->distinct_devices;
  if (!want || c->nranks == 1) return HFMI_OK;
  if (hipHostRegister(c->sh, sizeof(comm_shared), hipHostRegisterMapped) != hipSuccess) {
    (void)hipGetLastError();
    return HFMI_OK;                                         // stays host-synchronised
  }
  c->sh_registered = true;
  void *dp = nullptr, *ep = nullptr;
  if (hipHostGetDevicePointer(&dp, c->sh, 0) != hipSuccess || hipHostMalloc((void**)&c->err_host, 64, hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    return HFMI_OK;
  }
  *c->err_host = 0;
  if (hipHostGetDevicePointer(&ep, c->err_host, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(c->err_host);
    c->err_host = nullptr;
    return HFMI_OK;
  }
  c->dev_err = (int*)ep;
  c->sh_dev = (comm_shared*)dp;
  ctx_watch_comm(c->ctx, c);
  return HFMI_OK;
}
// the first all-reduce on a fresh RCCL communicator, with a time-out: 1 = right sum, 2 = anything else
static int rccl_first_contact(hfmi_comm* c) {
  const int n = 1024;
  double* buf = nullptr;
  if (hipMalloc((void**)&buf, n * sizeof(double)) != hipSuccess) {
    (void)hipGetLastError();
    return 2;
  }
  std::vector<double> h(n, (double)(c->rank + 1));
  int verdict = 2;
  hipStream_t st = c->ctx->stream;
  if (hipMemcpyAsync(buf, h.data(), n * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess &&
      hipStreamSynchronize(st) == hipSuccess &&
      g_rccl.AllReduce(buf, buf, (size_t)n, RCCL_DOUBLE, RCCL_SUM, c->nccl, st) == 0) {
    const double t0 = now_s(), limit = std::min(comm_timeout_s(), 120.0);
    hipError_t q;
    while ((q = hipStreamQuery(st)) == hipErrorNotReady && now_s() - t0 < limit) {
      timespec ts = {0, 200000};
      nanosleep(&ts, nullptr);
    }
    if (q == hipSuccess && hipMemcpy(h.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) {
      const double want = 0.5 * c->nranks * (c->nranks + 1.0);
      verdict = 1;
      for (int i = 0; i < n; ++i)
        if (h[i] != want) verdict = 2;
    }
  }
  (void)hipGetLastError();
  if (verdict == 1) (void)hipFree(buf);   // after a time-out the buffer may still be in use by a stuck kernel: leak it
  return verdict;
}

extern "C" int hfmi_comm_init_rank(hfmi_ctx* ctx, const void* id_bytes, int nranks, int rank, hfmi_comm** out) {
  return comm_init_impl(ctx, id_bytes, nranks, rank, nullptr, out);
}
static int comm_init_impl(hfmi_ctx* ctx, const void* id_bytes, int nranks, int rank, const char* unlink_path, hfmi_comm** out) {
  if (!id_bytes || !out) HFMI_FAIL(HFMI_ERR_INVALID, "comm_init_rank: null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) HFMI_FAIL(HFMI_ERR_INVALID, "comm_init_rank: rank %d of %d", rank, nranks);
  id_layout id;
  memcpy(&id, id_bytes, sizeof(id));
  if (id.version != 1) HFMI_FAIL(HFMI_ERR_INVALID, "comm_init_rank: not an id made by hfmi_comm_unique_id");
  const char* tr = getenv("HFMI_COMM_TRANSPORT");
  const bool force_rccl = tr && strcmp(tr, "rccl") == 0;
  const bool force_p2p = tr && strcmp(tr, "p2p") == 0;
  if (tr && *tr && !force_rccl && !force_p2p && strcmp(tr, "auto") != 0 && strcmp(tr, "host") != 0)
    HFMI_FAIL(HFMI_ERR_INVALID, "HFMI_COMM_TRANSPORT=%s: expected auto, rccl, p2p or host", tr);
  if (!force_rccl && nranks > COMM_MAX_RANKS)
    HFMI_FAIL(HFMI_ERR_INVALID, "comm_init_rank: at most %d ranks share a node segment (set HFMI_COMM_TRANSPORT=rccl beyond)", COMM_MAX_RANKS);
  hfmi_comm* c = new (std::nothrow) hfmi_comm();
  if (!c) HFMI_FAIL(HFMI_ERR_INVALID, "out of host memory");
  memset((void*)c, 0, sizeof(*c));
  c->ctx = ctx;
  c->rank = rank;
  c->nranks = nranks;
  c->transport = TRANSPORT_HOST;
  if (ctx && hipSetDevice(ctx->device) != hipSuccess) {
    comm_free(c);
    HFMI_FAIL(HFMI_ERR_HIP, "comm_init_rank: hipSetDevice(%d) failed", ctx->device);
  }
  memcpy(g_last_token, id.token, sizeof(g_last_token));
  g_have_last_token = true;

  if (!force_rccl) {
    // node-local control segment named by the id's token; a fresh segment is zero-filled = initial barrier state
    char name[80] = "/hfmi-";
    for (int i = 0; i < 16; ++i) snprintf(name + 6 + 2 * i, 3, "%02x", id.token[i]);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) {
      comm_free(c);
      HFMI_FAIL(HFMI_ERR_COMM, "comm_init_rank: shm_open(%s) failed: %s", name, strerror(errno));
    }
    if (ftruncate(fd, sizeof(comm_shared)) != 0) {
      close(fd);
      comm_free(c);
      HFMI_FAIL(HFMI_ERR_COMM, "comm_init_rank: ftruncate failed: %s", strerror(errno));
    }
    void* m = mmap(nullptr, sizeof(comm_shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
      comm_free(c);
      HFMI_FAIL(HFMI_ERR_COMM, "comm_init_rank: mmap failed: %s", strerror(errno));
    }
    c->sh = (comm_shared*)m;
    comm_slot& me = c->sh->slot[rank];
    memset(&me, 0, sizeof(me));
    me.has_device = ctx ? 1 : 0;
    if (ctx) {
      if (hipDeviceGetPCIBusId(me.device_id, sizeof(me.device_id), ctx->device) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(me.device_id, sizeof(me.device_id), "device-%d", ctx->device);
      }
      me.rccl_ok = (id.rccl_valid && rccl_load()) ? 1 : 0;
    }
    int s = shm_barrier(c);
    // every rank has the segment mapped and (file bootstrap) has read the id file: nothing is left behind in /dev/shm or in
    // the temp directory, and a later communicator cannot pick this one's file up (its ranks leave only after the last barrier)
    if (rank == 0) {
      (void)shm_unlink(name);
      if (unlink_path) (void)unlink(unlink_path);
    }
    if (s != HFMI_OK) {
      comm_free(c);
      return s;
    }
    // the same decision on every rank, from the same table
    c->distinct_devices = true;
    for (int p = 0; p < nranks; ++p)
      for (int q = 0; q < p; ++q)
        if (strcmp(c->sh->slot[p].device_id, c->sh->slot[q].device_id) == 0) c->distinct_devices = false;
    const int t = decide_transport(nranks, c->sh->slot, force_p2p ? 2 : 0, c->why);
    if (t < 0) {
      char msg[200];
      snprintf(msg, sizeof(msg), "%s", c->why);
      comm_free(c);
      HFMI_FAIL(HFMI_ERR_INVALID, "comm_init_rank: %s", msg);
    }
    c->transport = t;
  } else {
    if (!ctx) {
      comm_free(c);
      HFMI_FAIL(HFMI_ERR_INVALID, "HFMI_COMM_TRANSPORT=rccl needs a device context");
    }
    c->transport = TRANSPORT_RCCL;
    snprintf(c->why, sizeof(c->why), "rccl requested (HFMI_COMM_TRANSPORT=rccl): no node segment, no fallback");
    // the id file stays until ncclCommInitRank -- a rendezvous of all ranks -- has returned: every peer has read it by then
  }

  if (c->transport == TRANSPORT_RCCL) {
    if (!id.rccl_valid || !rccl_load()) {
      comm_free(c);
      HFMI_FAIL(HFMI_ERR_COMM, "comm_init_rank: librccl is not available (HFMI_RCCL_LIB / librccl.so.1)");
    }
    static const char* inject = getenv("HFMI_COMM_INJECT");     // test hook: "init" / "first" make THIS path fail on every rank
    int r = (inject && !strcmp(inject, "init")) ? -1 : g_rccl.CommInitRank(&c->nccl, nranks, id.rccl, rank);
    if (!c->sh) {
      if (rank == 0 && unlink_path) (void)unlink(unlink_path);
      if (r != 0) {
        comm_free(c);
        HFMI_FAIL(HFMI_ERR_COMM, "ncclCommInitRank failed: %s", r > 0 ? g_rccl.GetErrorString(r) : "injected failure");
      }
    } else {
      // agree on the outcome: one rank's failure sends EVERY rank to the p2p transport
      comm_slot& me = c->sh->slot[rank];
      me.rccl_init = (r == 0) ? 1 : 2;
      int s = shm_barrier(c);
      bool all_ok = s == HFMI_OK;
      int bad = -1;
      for (int p = 0; p < nranks && s == HFMI_OK; ++p)
        if (c->sh->slot[p].rccl_init != 1) {
          all_ok = false;
          if (bad < 0) bad = p;
        }
      const char* stage = "ncclCommInitRank";
      if (all_ok) {
        me.rccl_first = (inject && !strcmp(inject, "first")) ? 2 : rccl_first_contact(c);
        s = shm_barrier(c);
        all_ok = s == HFMI_OK;
        for (int p = 0; p < nranks && s == HFMI_OK; ++p)
          if (c->sh->slot[p].rccl_first != 1) {
            all_ok = false;
            if (bad < 0) bad = p;
          }
        stage = "the first ncclAllReduce";
      }
      if (s != HFMI_OK) {
        comm_free(c);
        return s;
      }
      if (!all_ok) {
        if (c->nccl) {
          // a communicator that failed its first collective may hang in destroy: abort it if the library can, else leak it
          if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->nccl);
          else if (me.rccl_first != 2) (void)g_rccl.CommDestroy(c->nccl);
          c->nccl = nullptr;
        }
        c->transport = TRANSPORT_P2P;
        snprintf(c->why, sizeof(c->why), "fell back from rccl: %s failed on rank %d; all ranks agreed on p2p", stage, bad);
      }
    }
  }
  if (c->transport == TRANSPORT_P2P) {
    const int s = p2p_enable_stream_order(c);
    if (s != HFMI_OK) {
      comm_free(c);
      return s;
    }
    // every rank must take the same path: stream-ordered only if all of them could map the segment
    c->sh->slot[rank].rccl_first = c->sh_dev ? 11 : 12;
    int s2 = shm_barrier(c);
    if (s2 != HFMI_OK) {
      comm_free(c);
      return s2;
    }
    for (int p = 0; p < nranks; ++p)
      if (c->sh->slot[p].rccl_first != 11) c->sh_dev = nullptr;
  }
  const int s = shm_barrier(c);
  if (s != HFMI_OK) {
    comm_free(c);
    return s;
  }
  *out = c;
  return HFMI_OK;
}

extern "C" int hfmi_comm_destroy(hfmi_comm* c) {
  if (!c) return HFMI_OK;
  if (c->ctx) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
  }
  if (c->transport == TRANSPORT_P2P && c->sh) {
    // peers may still be reading this rank's buffer: meet first, unmap, meet again, then free
    const bool ok = shm_barrier(c) == HFMI_OK;
    (void)p2p_close_peers(c);
    if (ok) (void)shm_barrier(c);
    if (c->stage) (void)hipFree(c->stage);
  }
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  if (c->scratch) (void)hipFree(c->scratch);
  comm_free(c);
  return HFMI_OK;
}

// what was chosen and why, as one line of JSON (bench.py puts it into its record): transport, library, the PCI bus id of
// every rank, whether the p2p path is stream-ordered
extern "C" int hfmi_comm_describe(const hfmi_comm* c, char* buf, int len) {
  if (!c || !buf || len < 2) HFMI_FAIL(HFMI_ERR_INVALID, "comm_describe: bad argument");
  static const char* names[3] = {"host", "rccl", "p2p"};
  int o = snprintf(buf, (size_t)len, "{\"transport\": \"%s\", \"nranks\": %d, \"rank\": %d, \"why\": \"%s\", \"rccl_library\": \"%s\", "
                   "\"p2p_sync\": \"%s\", \"p2p_stage\": \"%s\", \"devices\": [",
                   names[c->transport], c->nranks, c->rank, c->why, g_rccl.handle ? g_rccl.path : "",
                   c->transport != TRANSPORT_P2P ? "" : (c->sh_dev ? "stream" : "host"),
                   c->transport != TRANSPORT_P2P || !c->stage ? "" : (c->stage_fine ? "fine-grained" : "coarse-grained"));
  for (int p = 0; p < c->nranks && o > 0 && o < len; ++p)
    o += snprintf(buf + o, (size_t)(len - o), "%s\"%s\"", p ? ", " : "", c->sh ? c->sh->slot[p].device_id : "");
  if (o > 0 && o < len) snprintf(buf + o, (size_t)(len - o), "], \"p2p_probe_rounds\": %d, \"p2p_probe_retries_total\": %d, \"p2p_probe_generations\": %d, "
                                 "\"p2p_regenerations_total\": %d}", c->probe_rounds, c->probe_retries_total, c->probe_generations,
                                 c->probe_regenerations_total);
  return HFMI_OK;
}

int comm_transport(const hfmi_comm* c) { return c ? c->transport : 0; }
int comm_reserve_stage(hfmi_comm* c, size_t bytes) {
  if (!c || c->transport != TRANSPORT_P2P || c->nranks == 1) return HFMI_OK;
  return p2p_ensure_stage(c, bytes);
}
extern "C" int hfmi_comm_info(const hfmi_comm* c, int* nranks, int* rank, int* transport) {
  if (!c) HFMI_FAIL(HFMI_ERR_INVALID, "null communicator");
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  if (transport) *transport = c->transport;
  return HFMI_OK;
}

// ------------------------------------------------------------------ collectives on device memory
int comm_allreduce_device(hfmi_comm* c, double* data, int64_t count, int op) {
  return comm_allreduce_device_on(c, data, count, op, nullptr);
}
// the same on another stream of the context (the row panels of an operator application are reduced on the auxiliary stream
// while the next panel is being computed, hfmi_api.hip)
int comm_allreduce_device_on(hfmi_comm* c, double* data, int64_t count, int op, void* hip_stream) {
  if (!c || !data) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce: null argument");
  if (op != HFMI_REDUCE_SUM && op != HFMI_REDUCE_AVG && op != HFMI_REDUCE_MAX) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce: unknown operation %d", op);
  if (!c->ctx) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce: a host-only communicator cannot reduce device memory");
  HIP_TRY(hipSetDevice(c->ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->ctx->stream;
  if (c->transport == TRANSPORT_RCCL) {
    const int rop = op == HFMI_REDUCE_SUM ? RCCL_SUM : op == HFMI_REDUCE_AVG ? RCCL_AVG : RCCL_MAX;
    RCCL_TRY(g_rccl.AllReduce(data, data, (size_t)count, RCCL_DOUBLE, rop, c->nccl, st));
    return HFMI_OK;
  }
  if (c->nranks == 1) return HFMI_OK;
  return p2p_allreduce_dev(c, data, count, op, st);
}
static int comm_bcast_device(hfmi_comm* c, void* data, size_t bytes, int root) {
  if (!c->ctx) HFMI_FAIL(HFMI_ERR_INVALID, "bcast: a host-only communicator cannot broadcast device memory");
  HIP_TRY(hipSetDevice(c->ctx->device));
  if (c->transport == TRANSPORT_RCCL) {
    RCCL_TRY(g_rccl.Broadcast(data, data, bytes, RCCL_INT8, root, c->nccl, c->ctx->stream));
    return HFMI_OK;
  }
  if (c->nranks == 1) return HFMI_OK;
  return p2p_bcast_dev(c, data, bytes, root);
}

// hp.MultiVector all-reduce of collective.py:98-111 (k Allreduce calls of length N there; one call on the whole
// block here, padding rows included: they are zero on every rank).  In place, on the context's stream.
extern "C" int hfmi_allreduce(hfmi_comm* c, hfmi_block* Y, int op) {
  if (!c || !Y) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce: null argument");
  if (c->ctx && Y->ctx != c->ctx) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce: block and communicator belong to different contexts");
  return comm_allreduce_device(c, Y->p, Y->ld * (int64_t)Y->nvec, op);
}
// collective.py:144-152 (k Bcast calls of length N there).
extern "C" int hfmi_bcast(hfmi_comm* c, hfmi_block* Y, int root) {
  if (!c || !Y) HFMI_FAIL(HFMI_ERR_INVALID, "bcast: null argument");
  if (root < 0 || root >= c->nranks) HFMI_FAIL(HFMI_ERR_INVALID, "bcast: root %d of %d ranks", root, c->nranks);
  if (c->ctx && Y->ctx != c->ctx) HFMI_FAIL(HFMI_ERR_INVALID, "bcast: block and communicator belong to different contexts");
  return comm_bcast_device(c, Y->p, (size_t)Y->ld * Y->nvec * sizeof(double), root);
}

// ------------------------------------------------------------------ collectives on host payloads
static int comm_scratch(hfmi_comm* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return HFMI_OK;
  HIP_TRY(hipStreamSynchronize(c->ctx->stream));
  if (c->scratch) HIP_TRY(hipFree(c->scratch));
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  HIP_TRY(hipMalloc((void**)&c->scratch, bytes + 4096));
  c->scratch_bytes = bytes + 4096;
  return HFMI_OK;
}
// collective.py:61-71 `_allReduce_array` (numpy arrays; floats and ints are wrapped into arrays by the caller, :85-93)
extern "C" int hfmi_allreduce_host(hfmi_comm* c, double* v, int64_t count, int op) {
  if (!c || (!v && count > 0)) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce_host: null argument");
  if (op != HFMI_REDUCE_SUM && op != HFMI_REDUCE_AVG && op != HFMI_REDUCE_MAX) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce_host: unknown operation %d", op);
  if (count < 0) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce_host: negative count");
  if (c->sh) {
    if (c->nranks == 1) return HFMI_OK;
    for (int64_t o = 0; o < count || o == 0; o += HOST_AREA_DOUBLES) {
      const int64_t n = std::max<int64_t>(0, std::min<int64_t>(HOST_AREA_DOUBLES, count - o));
      memcpy(c->sh->host_area[c->rank], v + o, (size_t)n * sizeof(double));
      HFMI_TRY(shm_barrier(c));
      for (int64_t i = 0; i < n; ++i) {            // rank order: identical bits on every rank
        double acc = c->sh->host_area[0][i];
        for (int p = 1; p < c->nranks; ++p) {
          const double w = c->sh->host_area[p][i];
          acc = (op == HFMI_REDUCE_MAX) ? (w > acc ? w : acc) : acc + w;
        }
        v[o + i] = (op == HFMI_REDUCE_AVG) ? acc * (1.0 / c->nranks) : acc;
      }
      HFMI_TRY(shm_barrier(c));
      if (count == 0) break;
    }
    return HFMI_OK;
  }
  // no node segment (HFMI_COMM_TRANSPORT=rccl): stage through device memory
  HIP_TRY(hipSetDevice(c->ctx->device));
  HFMI_TRY(comm_scratch(c, (size_t)std::max<int64_t>(count, 1) * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(c->scratch, v, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->ctx->stream));
  HFMI_TRY(comm_allreduce_device(c, c->scratch, count, op));
  HIP_TRY(hipMemcpyAsync(v, c->scratch, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream));
  HIP_TRY(hipStreamSynchronize(c->ctx->stream));
  return HFMI_OK;
}
extern "C" int hfmi_bcast_host(hfmi_comm* c, void* v, int64_t nbytes, int root) {
  if (!c || (!v && nbytes > 0)) HFMI_FAIL(HFMI_ERR_INVALID, "bcast_host: null argument");
  if (root < 0 || root >= c->nranks) HFMI_FAIL(HFMI_ERR_INVALID, "bcast_host: root %d of %d ranks", root, c->nranks);
  if (nbytes < 0) HFMI_FAIL(HFMI_ERR_INVALID, "bcast_host: negative size");
  if (c->sh) {
    if (c->nranks == 1) return HFMI_OK;
    const int64_t chunk = HOST_AREA_DOUBLES * (int64_t)sizeof(double);
    for (int64_t o = 0; o < nbytes; o += chunk) {
      const int64_t n = std::min<int64_t>(chunk, nbytes - o);
      if (c->rank == root) memcpy(c->sh->host_area[root], (char*)v + o, (size_t)n);
      HFMI_TRY(shm_barrier(c));
      if (c->rank != root) memcpy((char*)v + o, c->sh->host_area[root], (size_t)n);
      HFMI_TRY(shm_barrier(c));
    }
    return HFMI_OK;
  }
  HIP_TRY(hipSetDevice(c->ctx->device));
  HFMI_TRY(comm_scratch(c, (size_t)std::max<int64_t>(nbytes, 1)));
  HIP_TRY(hipMemcpyAsync(c->scratch, v, (size_t)nbytes, hipMemcpyHostToDevice, c->ctx->stream));
  HFMI_TRY(comm_bcast_device(c, c->scratch, (size_t)nbytes, root));
  HIP_TRY(hipMemcpyAsync(v, c->scratch, (size_t)nbytes, hipMemcpyDeviceToHost, c->ctx->stream));
  HIP_TRY(hipStreamSynchronize(c->ctx->stream));
  return HFMI_OK;
}

// Every rank's queued device work is complete and every rank has arrived.
extern "C" int hfmi_comm_barrier(hfmi_comm* c) {
  if (!c) HFMI_FAIL(HFMI_ERR_INVALID, "null communicator");
  if (c->ctx) {
    HIP_TRY(hipSetDevice(c->ctx->device));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    HFMI_TRY(p2p_check_err(c));
  }
  if (c->sh) return shm_barrier(c);
  double one = 1.0;
  return hfmi_allreduce_host(c, &one, 1, HFMI_REDUCE_SUM);
}
