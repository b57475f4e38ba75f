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
  const char* cands[3] = {getenv("HFMI_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* c : cands) {
    if (!c || !*c) continue;
    void* h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (!h) continue;
    g_rccl.GetUniqueId = (int (*)(rccl_unique_id*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(rccl_comm_t*, int, rccl_unique_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(rccl_comm_t))dlsym(h, "ncclCommDestroy");
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
  int32_t rccl_ok;
};
struct comm_shared {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> abort_flag;
  uint32_t pad;
  comm_slot slot[COMM_MAX_RANKS];
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
  double* peer[COMM_MAX_RANKS];   // mapped staging buffers (peer[rank] == stage)
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

// rank 0 makes the id and publishes it in `path` (write + rename: readers never see a partial file); the other
// ranks wait for the file.  The reference's counterpart is mpi4py's own bootstrap; a host that has MPI can instead
// broadcast the bytes of hfmi_comm_unique_id itself and call hfmi_comm_init_rank.
extern "C" int hfmi_comm_init_from_file(hfmi_ctx* ctx, const char* path, int nranks, int rank, hfmi_comm** out) {
  if (!path || !out) HFMI_FAIL(HFMI_ERR_INVALID, "comm_init_from_file: null argument");
  unsigned char id[HFMI_UNIQUE_ID_BYTES];
  if (rank == 0) {
    HFMI_TRY(hfmi_comm_unique_id(id));
    char tmp[1024];
    snprintf(tmp, sizeof(tmp), "%s.tmp.%d", path, (int)getpid());
    FILE* f = fopen(tmp, "wb");
    if (!f) HFMI_FAIL(HFMI_ERR_COMM, "comm_init_from_file: cannot create %s: %s", tmp, strerror(errno));
    const size_t w = fwrite(id, 1, sizeof(id), f);
    fclose(f);
    if (w != sizeof(id) || rename(tmp, path) != 0) HFMI_FAIL(HFMI_ERR_COMM, "comm_init_from_file: cannot publish %s", path);
  } else {
    const double t0 = now_s(), limit = comm_timeout_s();
    for (;;) {
      FILE* f = fopen(path, "rb");
      if (f) {
        const size_t r = fread(id, 1, sizeof(id), f);
        fclose(f);
        if (r == sizeof(id)) break;
      }
      if (now_s() - t0 > limit) HFMI_FAIL(HFMI_ERR_COMM, "rank %d: no communicator id in %s after %.0f s", rank, path, limit);
      timespec ts = {0, 2000000};
      nanosleep(&ts, nullptr);
    }
  }
  HFMI_TRY(hfmi_comm_init_rank(ctx, id, nranks, rank, out));
  if (rank == 0) (void)unlink(path);      // every rank has read it: init_rank ends with a barrier
  return HFMI_OK;
}

// ------------------------------------------------------------------ P2P staging buffers
static int p2p_close_peers(hfmi_comm* c) {
  for (int p = 0; p < c->nranks; ++p) {
    if (p != c->rank && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
    c->peer[p] = nullptr;
  }
  return HFMI_OK;
}
// Collective: every rank calls it with the same `bytes`.
static int p2p_ensure_stage(hfmi_comm* c, size_t bytes) {
  if (bytes <= c->stage_bytes) return HFMI_OK;
  HIP_TRY(hipStreamSynchronize(c->ctx->stream));  // this rank's last copy out of the old buffer is complete
  HFMI_TRY(shm_barrier(c));                       // nobody is still reading the old buffers
  HFMI_TRY(p2p_close_peers(c));
  HFMI_TRY(shm_barrier(c));                       // every mapping of the old buffer is closed before it is freed
  if (c->stage) HIP_TRY(hipFree(c->stage));
  c->stage = nullptr;
  c->stage_bytes = 0;
  const size_t want = round_up((int64_t)(bytes + bytes / 8), 1 << 20);
  HIP_TRY(hipMalloc((void**)&c->stage, want));
  c->stage_bytes = want;
  comm_slot& me = c->sh->slot[c->rank];
  HIP_TRY(hipIpcGetMemHandle(&me.handle, c->stage));
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
  return shm_barrier(c);
}

struct p2p_ptrs {
  double* p[COMM_MAX_RANKS];
};
// rank r owns the d2 elements [lo, hi): sum over ranks in rank order (every rank ends up with identical bits),
// scale, and store into the same slice of every rank's buffer.
__global__ void __launch_bounds__(256) k_p2p_reduce(p2p_ptrs bufs, int nranks, int64_t lo, int64_t hi, double scale, int op) {
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

static int p2p_allreduce_dev(hfmi_comm* c, double* data, int64_t count, int op) {
  hfmi_ctx* ctx = c->ctx;
  const int64_t padded = round_up(count, 2);
  const size_t bytes = (size_t)padded * sizeof(double);
  HFMI_TRY(p2p_ensure_stage(c, bytes));
  if (padded != count) HIP_TRY(hipMemsetAsync(c->stage + count, 0, sizeof(double), ctx->stream));
  HIP_TRY(hipMemcpyAsync(c->stage, data, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HFMI_TRY(shm_barrier(c));                       // every rank's contribution is in its staging buffer
  const int64_t n2 = padded / 2;
  const int64_t lo = n2 * c->rank / c->nranks, hi = n2 * (c->rank + 1) / c->nranks;
  if (hi > lo) {
    p2p_ptrs bufs;
    for (int p = 0; p < COMM_MAX_RANKS; ++p) bufs.p[p] = p < c->nranks ? c->peer[p] : nullptr;
    const int blocks = (int)std::min<int64_t>((hi - lo + 255) / 256, (int64_t)ctx->num_cus * 8);
    const double scale = (op == HFMI_REDUCE_AVG) ? 1.0 / c->nranks : 1.0;
    hipLaunchKernelGGL(k_p2p_reduce, dim3(blocks), dim3(256), 0, ctx->stream, bufs, c->nranks, lo, hi, scale, op);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HFMI_TRY(shm_barrier(c));                       // every slice of every buffer is final
  HIP_TRY(hipMemcpyAsync(data, c->stage, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  // the staging buffer may be overwritten by the next collective only after this copy: the next collective
  // starts with a stream synchronise of its own copy-in, which is ordered behind this copy on the same stream
  return HFMI_OK;
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

// ------------------------------------------------------------------ init / destroy
extern "C" int hfmi_comm_init_rank(hfmi_ctx* ctx, const void* id_bytes, int nranks, int rank, hfmi_comm** out) {
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
  if (ctx) HIP_TRY(hipSetDevice(ctx->device));

  if (!force_rccl) {
    // node-local control segment named by the id's token; a fresh segment is zero-filled = initial barrier state
    char name[80] = "/hfmi-";
    for (int i = 0; i < 16; ++i) snprintf(name + 6 + 2 * i, 3, "%02x", id.token[i]);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) {
      delete c;
      HFMI_FAIL(HFMI_ERR_COMM, "comm_init_rank: shm_open(%s) failed: %s", name, strerror(errno));
    }
    if (ftruncate(fd, sizeof(comm_shared)) != 0) {
      close(fd);
      delete c;
      HFMI_FAIL(HFMI_ERR_COMM, "comm_init_rank: ftruncate failed: %s", strerror(errno));
    }
    void* m = mmap(nullptr, sizeof(comm_shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
      delete c;
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
    if (s == HFMI_OK && rank == 0) (void)shm_unlink(name);    // every rank has it mapped: nothing is left in /dev/shm
    if (s != HFMI_OK) {
      if (rank == 0) (void)shm_unlink(name);
      munmap(c->sh, sizeof(comm_shared));
      delete c;
      return s;
    }
    // the same decision on every rank, from the same table
    bool all_dev = true, all_rccl = true, distinct = true;
    for (int p = 0; p < nranks; ++p) {
      all_dev = all_dev && c->sh->slot[p].has_device;
      all_rccl = all_rccl && c->sh->slot[p].rccl_ok;
      for (int q = 0; q < p; ++q)
        if (strcmp(c->sh->slot[p].device_id, c->sh->slot[q].device_id) == 0) distinct = false;
    }
    if (!all_dev) {
      bool any = false;
      for (int p = 0; p < nranks; ++p) any = any || c->sh->slot[p].has_device;
      if (any) {
        munmap(c->sh, sizeof(comm_shared));
        delete c;
        HFMI_FAIL(HFMI_ERR_INVALID, "comm_init_rank: some ranks passed a context and some did not");
      }
      c->transport = TRANSPORT_HOST;
    } else if (force_p2p || !distinct || !all_rccl) {
      c->transport = TRANSPORT_P2P;
    } else {
      c->transport = TRANSPORT_RCCL;
    }
  } else {
    if (!ctx) {
      delete c;
      HFMI_FAIL(HFMI_ERR_INVALID, "HFMI_COMM_TRANSPORT=rccl needs a device context");
    }
    c->transport = TRANSPORT_RCCL;
  }

  if (c->transport == TRANSPORT_RCCL) {
    if (!id.rccl_valid || !rccl_load()) {
      if (c->sh) munmap(c->sh, sizeof(comm_shared));
      delete c;
      HFMI_FAIL(HFMI_ERR_COMM, "comm_init_rank: librccl is not available (HFMI_RCCL_LIB / librccl.so.1)");
    }
    const int r = g_rccl.CommInitRank(&c->nccl, nranks, id.rccl, rank);
    if (r != 0) {
      if (c->sh) munmap(c->sh, sizeof(comm_shared));
      delete c;
      HFMI_FAIL(HFMI_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
  }
  *out = c;
  const int s = shm_barrier(c);
  if (s != HFMI_OK) return s;
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
  if (c->sh) munmap(c->sh, sizeof(comm_shared));
  delete c;
  return HFMI_OK;
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
  if (!c || !data) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce: null argument");
  if (op != HFMI_REDUCE_SUM && op != HFMI_REDUCE_AVG && op != HFMI_REDUCE_MAX) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce: unknown operation %d", op);
  if (!c->ctx) HFMI_FAIL(HFMI_ERR_INVALID, "allreduce: a host-only communicator cannot reduce device memory");
  HIP_TRY(hipSetDevice(c->ctx->device));
  if (c->transport == TRANSPORT_RCCL) {
    const int rop = op == HFMI_REDUCE_SUM ? RCCL_SUM : op == HFMI_REDUCE_AVG ? RCCL_AVG : RCCL_MAX;
    RCCL_TRY(g_rccl.AllReduce(data, data, (size_t)count, RCCL_DOUBLE, rop, c->nccl, c->ctx->stream));
    return HFMI_OK;
  }
  if (c->nranks == 1) return HFMI_OK;
  return p2p_allreduce_dev(c, data, count, op);
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
  }
  if (c->sh) return shm_barrier(c);
  double one = 1.0;
  return hfmi_allreduce_host(c, &one, 1, HFMI_REDUCE_SUM);
}
