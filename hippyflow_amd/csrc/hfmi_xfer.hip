// Large host <-> device transfers of the C ABI's host-in / host-out calls (the n x n matrix of hfmi_sym_eig_small and its n x n
// eigenvector matrix: 512 MB each way at n = 8192; PODProjector.py:812-833 hands numpy arrays over and takes numpy arrays back).
//
// The caller's arrays are pageable.  hipMemcpyAsync on pageable memory goes through the runtime's own staging / page-locking
// path: measured (profiles/r05_eig_large.txt) 16-17 GB/s device -> host at every size, and 24-78 ms stalls on one call in four
// in the phase that uploads the matrix.  Here the transfer is pipelined through a ring of pinned chunks owned by the context:
//   device -> host: copy engine fills chunk c + 1 ... c + 3 while a few host threads move chunk c out of the ring into the caller's
//                   array (the first touch of a fresh numpy array's pages is spread over those threads as well);
//   host -> device: the threads fill chunk c + 1 ... while chunk c travels (not the default: see xfer_h2d).
// Both are stream-ordered on the context's stream like the copies they replace.  xfer_d2h returns when the host array is
// complete; xfer_h2d returns when the host array has been read and every chunk has left the ring.
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "hfmi_internal.h"

namespace {
constexpr size_t XF_CHUNK_MAX = (size_t)16 << 20;   // a ring slot: eight 2 MB pages, one per worker (see for_slices)
constexpr size_t XF_CHUNK = XF_CHUNK_MAX;
constexpr int XF_NBUF = 3;
constexpr size_t XF_DIRECT = (size_t)8 << 20;   // below this the plain copy is as good

struct pool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::function<void(int)> job;
  uint64_t generation = 0;
  int pending = 0;
  bool stop = false;
  void start(int n) {
    for (int t = 0; t < n; ++t)
      threads.emplace_back([this, t] {
        uint64_t seen = 0;
        for (;;) {
          std::function<void(int)> fn;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_job.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            fn = job;
          }
          fn(t);
          {
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
          }
        }
      });
  }
  void launch(const std::function<void(int)>& fn) {
    std::lock_guard<std::mutex> lk(mu);
    job = fn;
    pending = (int)threads.size();
    ++generation;
    cv_job.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_job.notify_all();
    for (auto& th : threads) th.join();
  }
};
}  // namespace

// device -> pinned host memory by a kernel: the shader cores write across PCIe themselves.  The copy engine the runtime uses for
// hipMemcpyAsync delivered 21-26 GB/s device -> host on these boxes (49 GB/s host -> device: profiles/r06a_eig_large.txt)
typedef double xd2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_xfer_out(xd2* __restrict__ dst, const xd2* __restrict__ src, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

struct xfer_state {
  char* pin_dev = nullptr;        // the ring as the device sees it
  char* pin = nullptr;
  hipEvent_t ev[XF_NBUF];
  bool have_events = false;
  pool workers;
  int nthreads = 0;
};

static int xfer_get(hfmi_ctx* ctx, xfer_state** out) {
  if (!ctx->xfer) {
    xfer_state* s = new (std::nothrow) xfer_state();
    if (!s) HFMI_FAIL(HFMI_ERR_INVALID, "out of host memory");
    ctx->xfer = s;
    HIP_TRY(hipHostMalloc((void**)&s->pin, XF_NBUF * XF_CHUNK, hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer((void**)&s->pin_dev, s->pin, 0));
    for (int i = 0; i < XF_NBUF; ++i) HIP_TRY(hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming));
    s->have_events = true;
    const char* e = getenv("HFMI_XFER_THREADS");
    int nt = e ? atoi(e) : 0;
    if (nt <= 0) nt = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 4));
    s->nthreads = std::min(nt, 32);
    s->workers.start(s->nthreads);
  }
  *out = ctx->xfer;
  return HFMI_OK;
}
void xfer_destroy(hfmi_ctx* ctx) {
  xfer_state* s = ctx->xfer;
  if (!s) return;
  if (s->have_events)
    for (int i = 0; i < XF_NBUF; ++i) (void)hipEventDestroy(s->ev[i]);
  if (s->pin) (void)hipHostFree(s->pin);
  delete s;
  ctx->xfer = nullptr;
}

// Thread t's share of a chunk of `len` bytes that lands at host address `dst`: the pieces between consecutive 2 MB boundaries OF THE
// DESTINATION, dealt round-robin.  A fresh numpy array is backed by transparent huge pages that are zero-filled on first touch,
// one page per fault, by ONE thread while every other thread touching the same page waits: with 4 KB slices of a 4 MB chunk eight
// threads queued on two page faults per chunk and the 512 MB output of n = 8192 took 24-27 ms (19 GB/s); with one huge page
// per thread the faults run side by side.  Calls fn(offset, bytes) for every piece of thread t.
template <class F>
static inline void for_slices(const void* dst, size_t len, int t, int T, F fn) {
  constexpr uintptr_t HP = (uintptr_t)2 << 20;
  const uintptr_t a0 = (uintptr_t)dst, a1 = a0 + len;
  uintptr_t piece = a0 & ~(HP - 1);
  for (int k = 0; piece < a1; piece += HP, ++k) {
    if (k % T != t) continue;
    const uintptr_t lo = std::max(piece, a0), hi = std::min(piece + HP, a1);
    if (hi > lo) fn((size_t)(lo - a0), (size_t)(hi - lo));
  }
}
static inline void slice_of(size_t len, int t, int T, size_t& lo, size_t& hi) {      // (the upload ring: 4 KB pages, contiguous shares)
  const size_t pages = (len + 4095) / 4096, per = (pages + T - 1) / T;
  lo = std::min(len, (size_t)t * per * 4096);
  hi = std::min(len, (size_t)(t + 1) * per * 4096);
}

int xfer_d2h(hfmi_ctx* ctx, void* host, const void* dev, size_t bytes) {
  static const bool plain = env_flag("HFMI_XFER_PLAIN");       // A/B: the runtime's pageable path
  hipStream_t st = ctx->stream;
  if (bytes < XF_DIRECT || plain) {
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return HFMI_OK;
  }
  xfer_state* s = nullptr;
  HFMI_TRY(xfer_get(ctx, &s));
  // chunk: an eighth of the transfer in whole 2 MB pages, 8 ... 16 MB (a 32 MB output: four chunks of four huge pages)
  const size_t XF_CHUNK = std::min(XF_CHUNK_MAX, std::max((size_t)8 << 20, ((bytes / 8) >> 21) << 21));
  const int nch = (int)((bytes + XF_CHUNK - 1) / XF_CHUNK), T = s->nthreads;
  // hand-over between this thread and the workers through ONE mutex + condition variable: nobody spins (a spinning helper thread
  // competes with the thread that feeds the GPU for the cores the process may use -- see DESIGN section 8 item 3)
  std::mutex mu;
  std::condition_variable cv;
  int ready = 0;                    // chunks whose data is in the ring
  bool abort_flag = false;
  std::vector<int> done(nch, 0);    // workers that have moved chunk c out
  auto chunk_len = [&](int c) { return std::min(XF_CHUNK, bytes - (size_t)c * XF_CHUNK); };
  s->workers.launch([&, T](int t) {
    for (int c = 0; c < nch; ++c) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ready > c || abort_flag; });
        if (abort_flag) return;
      }
      char* dst = (char*)host + (size_t)c * XF_CHUNK;
      const char* src = s->pin + (size_t)(c % XF_NBUF) * XF_CHUNK_MAX;
      for_slices(dst, chunk_len(c), t, T, [&](size_t off, size_t nb) { memcpy(dst + off, src + off, nb); });
      {
        std::lock_guard<std::mutex> lk(mu);
        if (++done[c] == T) cv.notify_all();
      }
    }
  });
  hipError_t err = hipSuccess;
  static const bool engine = env_flag("HFMI_XFER_D2H_ENGINE");      // A/B: the runtime's copy engine instead of the copy kernel
  const bool aligned = (((uintptr_t)dev) & 15) == 0;
  auto issue = [&](int c) {
    const size_t len = chunk_len(c);
    if (engine || !aligned || (len & 15)) {
      err = hipMemcpyAsync(s->pin + (size_t)(c % XF_NBUF) * XF_CHUNK_MAX, (const char*)dev + (size_t)c * XF_CHUNK, len, hipMemcpyDeviceToHost, st);
    } else {
      hipLaunchKernelGGL(k_xfer_out, dim3(128), dim3(256), 0, st, (xd2*)(s->pin_dev + (size_t)(c % XF_NBUF) * XF_CHUNK_MAX),
                         (const xd2*)((const char*)dev + (size_t)c * XF_CHUNK), len / 16);
      err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipEventRecord(s->ev[c % XF_NBUF], st);
  };
  for (int c = 0; c < std::min(nch, XF_NBUF) && err == hipSuccess; ++c) issue(c);
  for (int c = 0; c < nch && err == hipSuccess; ++c) {
    err = hipEventSynchronize(s->ev[c % XF_NBUF]);
    if (err != hipSuccess) break;
    {
      std::unique_lock<std::mutex> lk(mu);
      ready = c + 1;
      cv.notify_all();
      if (c + XF_NBUF < nch) cv.wait(lk, [&] { return done[c] == T; });       // the ring slot is free again
    }
    if (c + XF_NBUF < nch) issue(c + XF_NBUF);
  }
  if (err != hipSuccess) {
    std::lock_guard<std::mutex> lk(mu);
    abort_flag = true;
    cv.notify_all();
  }
  s->workers.wait();
  if (err != hipSuccess) {
    (void)hipStreamSynchronize(st);
    hfmi_set_error("device -> host transfer of %zu bytes failed: %s", bytes, hipGetErrorString(err));
    return HFMI_ERR_HIP;
  }
  return HFMI_OK;
}

int xfer_h2d(hfmi_ctx* ctx, void* dev, const void* host, size_t bytes) {
  // Upload: the runtime's own pageable path is the faster one here (it pins the caller's pages and lets the copy engine read them:
  // 49 GB/s at 512 MB against 34 GB/s through the ring, whose threads have to read AND write every byte first: profiles/r06a), so
  // the ring is the A/B partner (HFMI_XFER_H2D_RING=1), not the default.
  static const bool plain = !env_flag("HFMI_XFER_H2D_RING") || env_flag("HFMI_XFER_PLAIN");
  hipStream_t st = ctx->stream;
  if (bytes < XF_DIRECT || plain) {
    HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
    return HFMI_OK;      // (pageable source: the call returns once the runtime has staged it)
  }
  xfer_state* s = nullptr;
  HFMI_TRY(xfer_get(ctx, &s));
  const int nch = (int)((bytes + XF_CHUNK - 1) / XF_CHUNK), T = s->nthreads;
  std::atomic<int> freed{0};          // chunks whose copy has left the ring
  std::atomic<bool> abort_flag{false};
  std::vector<std::atomic<int>> filled(nch);
  for (auto& d : filled) d.store(0, std::memory_order_relaxed);
  auto chunk_len = [&](int c) { return std::min(XF_CHUNK, bytes - (size_t)c * XF_CHUNK); };
  s->workers.launch([&, T](int t) {
    for (int c = 0; c < nch; ++c) {
      while (c >= freed.load(std::memory_order_acquire) + XF_NBUF) {
        if (abort_flag.load(std::memory_order_relaxed)) return;
        std::this_thread::yield();
      }
      size_t lo, hi;
      slice_of(chunk_len(c), t, T, lo, hi);
      if (hi > lo) memcpy(s->pin + (size_t)(c % XF_NBUF) * XF_CHUNK + lo, (const char*)host + (size_t)c * XF_CHUNK + lo, hi - lo);
      filled[c].fetch_add(1, std::memory_order_release);
    }
  });
  hipError_t err = hipSuccess;
  for (int c = 0; c < nch && err == hipSuccess; ++c) {
    while (filled[c].load(std::memory_order_acquire) < T) std::this_thread::yield();
    err = hipMemcpyAsync((char*)dev + (size_t)c * XF_CHUNK, s->pin + (size_t)(c % XF_NBUF) * XF_CHUNK, chunk_len(c), hipMemcpyHostToDevice, st);
    if (err == hipSuccess) err = hipEventRecord(s->ev[c % XF_NBUF], st);
    if (err == hipSuccess && c + 1 >= XF_NBUF - 1) {
      // keep XF_NBUF - 1 copies in flight: the oldest of them must have left its slot before the threads refill it
      const int oldest = c + 1 - (XF_NBUF - 1);
      err = hipEventSynchronize(s->ev[oldest % XF_NBUF]);
      freed.store(oldest + 1, std::memory_order_release);
    }
  }
  if (err != hipSuccess) abort_flag.store(true);
  s->workers.wait();
  if (err == hipSuccess) err = hipStreamSynchronize(st);      // the ring is idle when the call returns
  if (err != hipSuccess) {
    hfmi_set_error("host -> device transfer of %zu bytes failed: %s", bytes, hipGetErrorString(err));
    return HFMI_ERR_HIP;
  }
  return HFMI_OK;
}
