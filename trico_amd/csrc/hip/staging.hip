// staging.hip — large copies between PAGEABLE host memory and the device through a ring of pinned chunks (gfx950 host side).
//
// Every caller of the reference passes host pointers (tools/trico_encoder/main.c:253-323, trico.c:215-262, 323-378), so the plain API
// begins with 1.8 GB of host-to-device copy for the benchmark mesh.  hipMemcpyAsync from pageable memory stages through pinned buffers
// of the runtime with ONE host thread: measured in round 5 (rocprofv3 --memory-copy-trace over tools/perf_host_pointers.py), the DMA
// engines need 15 ms for that mesh and the call 42 ms.  Here the staging is done by a few host threads: the source is cut into
// chunks, every chunk is copied into one of RING pinned buffers by all threads at once, handed to the copy engine
// (hipMemcpyAsync on a stream of its own) and the next chunk is staged while it flies.  The kernels' stream waits for the last copy
// with an event.  The other direction (payloads into a host archive) the same way round.
// One pipe per process, one user at a time (a mutex): concurrent callers take turns, small or already pinned transfers do not come
// here at all (shim.hip: stage_in, trico_hip_copy).
#include "common.hpp"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <pthread.h>
#include <string.h>
#include <thread>
#include <vector>

namespace trico {

namespace {

constexpr size_t CHUNK = 16u << 20;               // bytes per pinned buffer
constexpr int RING = 4;

// a few host threads that copy slices of one chunk at a time
struct CopyPool
  {
  struct Slice { char* d; const char* s; size_t n; };
  std::mutex m;
  std::condition_variable wake, done;
  std::deque<Slice> q;
  int busy = 0;
  int nthreads = 0;

  void start(int n)
    {
    nthreads = n;
    for (int i = 0; i < n; ++i)
      std::thread([this] { run(); }).detach();    // (they live as long as the process: the library is never unloaded by its users)
    }

  void run()
    {
    std::unique_lock<std::mutex> lk(m);
    for (;;)
      {
      wake.wait(lk, [this] { return !q.empty(); });
      const Slice s = q.front();
      q.pop_front();
      lk.unlock();
      memcpy(s.d, s.s, s.n);
      lk.lock();
      if (--busy == 0)
        done.notify_all();
      }
    }

  // d[0..n) = s[0..n), by the pool's threads and the caller; returns when all of it is there
  void copy(char* d, const char* s, size_t n)
    {
    const int parts = nthreads + 1;
    size_t per = (n / (size_t)parts + 4095) & ~(size_t)4095;
    if (per == 0 || nthreads == 0)
      {
      memcpy(d, s, n);
      return;
      }
    size_t mine = per < n ? per : n, off = mine;
      {
      std::lock_guard<std::mutex> lk(m);
      while (off < n)
        {
        const size_t len = n - off < per ? n - off : per;
        q.push_back(Slice{ d + off, s + off, len });
        ++busy;
        off += len;
        }
      }
    wake.notify_all();
    memcpy(d, s, mine);
    std::unique_lock<std::mutex> lk(m);
    done.wait(lk, [this] { return busy == 0; });
    }
  };

struct Pipe
  {
  std::mutex m;                                   // one transfer at a time
  bool tried = false, ok = false;
  int device = -1;
  char* pin[RING] = { nullptr, nullptr, nullptr, nullptr };
  hipEvent_t ev[RING];
  bool used[RING] = { false, false, false, false };
  hipEvent_t edge;                                // between the caller's stream and the copy stream
  hipStream_t copies = nullptr;
  CopyPool pool;

  bool init()
    {
    if (tried)
      return ok;
    tried = true;
    int threads = 0;
    if (const char* e = getenv("TRICO_HIP_STAGE_THREADS"))
      threads = atoi(e);
    else
      {
      // + the calling thread; with 3 the runtime's own staging is faster, 15 gain nothing (profiles/r05_host_pointers.txt).  The ranks
      // of a node share its cores: each takes its share of them (LOCAL_WORLD_SIZE, as torchrun and mpirun export it)
      unsigned hw = std::thread::hardware_concurrency();
      if (const char* lw = getenv("LOCAL_WORLD_SIZE"))
        if (atoi(lw) > 1)
          hw /= (unsigned)atoi(lw);
      threads = hw >= 16 ? 7 : 0;
      }
    if (threads > 32)
      threads = 32;                               // (an upper bound whatever the variable says)
    if (threads <= 0)
      return false;                               // (0: the runtime's own staging, as before)
    if (hipGetDevice(&device) != hipSuccess)
      return false;
    for (int i = 0; i < RING; ++i)
      if (hipHostMalloc((void**)&pin[i], CHUNK, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess)
        {
        (void)hipGetLastError();
        return false;
        }
    if (hipEventCreateWithFlags(&edge, hipEventDisableTiming) != hipSuccess || hipStreamCreateWithFlags(&copies, hipStreamNonBlocking) != hipSuccess)
      {
      (void)hipGetLastError();
      return false;
      }
    pool.start(threads);
    ok = true;
    return true;
    }

  bool usable()
    {
    int dev = -1;
    return init() && hipGetDevice(&dev) == hipSuccess && dev == device;       // (the buffers and the stream belong to one device)
    }
  };

// (on the heap and never destroyed: the pool's threads wait on its condition variable for as long as the process lives, and the
// destructor of a condition variable waits for its waiters - a global would hang every exit)
// A forked child has none of the pool's threads (fork copies the calling thread only) and may not touch the parent's HIP objects:
// its pipe is a new one that has not been tried (what the parent's object held stays allocated; the child never looks at it).
static std::atomic<Pipe*> g_pipe{ nullptr };
static void pipe_forget_in_child() { g_pipe.store(nullptr); }

Pipe& the_pipe()
  {
  static const int hooked = pthread_atfork(nullptr, nullptr, pipe_forget_in_child);
  (void)hooked;
  Pipe* p = g_pipe.load();
  if (!p)
    {
    Pipe* fresh = new Pipe;
    if (g_pipe.compare_exchange_strong(p, fresh))
      p = fresh;
    else
      delete fresh;
    }
  return *p;
  }

} // namespace

static bool pointer_is_pageable_host(const void* p)
  {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess)
    {
    (void)hipGetLastError();
    return true;                                  // not known to the runtime at all: ordinary host memory
    }
  return attr.type == hipMemoryTypeUnregistered;
  }

// bytes of pageable host memory to the device; kernels launched on `consumer` afterwards see them.  false: not done (no pipe on this
// device, or an error - the caller then copies the plain way)
static bool staged_upload(void* d_dst, const void* h_src, size_t bytes, hipStream_t consumer)
  {
  Pipe& P = the_pipe();
  std::lock_guard<std::mutex> lk(P.m);
  if (!P.usable())
    return false;
  // what the consumer stream has in flight may still read the destination buffer (it is reused between calls)
  if (hipEventRecord(P.edge, consumer) != hipSuccess || hipStreamWaitEvent(P.copies, P.edge, 0) != hipSuccess)
    return false;
  const char* s = (const char*)h_src;
  char* d = (char*)d_dst;
  int last = -1;
  size_t k = 0;
  for (size_t off = 0; off < bytes; off += CHUNK, ++k)
    {
    const int slot = (int)(k % RING);
    const size_t len = bytes - off < CHUNK ? bytes - off : CHUNK;
    if (P.used[slot] && hipEventSynchronize(P.ev[slot]) != hipSuccess)
      return false;
    P.pool.copy(P.pin[slot], s + off, len);
    if (hipMemcpyAsync(d + off, P.pin[slot], len, hipMemcpyHostToDevice, P.copies) != hipSuccess || hipEventRecord(P.ev[slot], P.copies) != hipSuccess)
      return false;
    P.used[slot] = true;
    last = slot;
    }
  if (last >= 0 && hipStreamWaitEvent(consumer, P.ev[last], 0) != hipSuccess)
    return false;
  return true;
  }

// bytes from the device (produced on `producer`) into pageable host memory; complete on return
static bool staged_download(void* h_dst, const void* d_src, size_t bytes, hipStream_t producer)
  {
  Pipe& P = the_pipe();
  std::lock_guard<std::mutex> lk(P.m);
  if (!P.usable())
    return false;
  if (hipEventRecord(P.edge, producer) != hipSuccess || hipStreamWaitEvent(P.copies, P.edge, 0) != hipSuccess)
    return false;
  char* d = (char*)h_dst;
  const char* s = (const char*)d_src;
  const size_t chunks = (bytes + CHUNK - 1) / CHUNK;
  auto issue = [&](size_t k) -> bool
    {
    const int slot = (int)(k % RING);
    const size_t off = k * CHUNK, len = bytes - off < CHUNK ? bytes - off : CHUNK;
    if (P.used[slot] && hipEventSynchronize(P.ev[slot]) != hipSuccess)
      return false;
    if (hipMemcpyAsync(P.pin[slot], s + off, len, hipMemcpyDeviceToHost, P.copies) != hipSuccess || hipEventRecord(P.ev[slot], P.copies) != hipSuccess)
      return false;
    P.used[slot] = true;
    return true;
    };
  // RING - 1 chunks in flight; a chunk is copied out by the host threads while the next ones arrive
  for (size_t k = 0; k < chunks && k < (size_t)(RING - 1); ++k)
    if (!issue(k))
      return false;
  for (size_t k = 0; k < chunks; ++k)
    {
    const int slot = (int)(k % RING);
    const size_t off = k * CHUNK, len = bytes - off < CHUNK ? bytes - off : CHUNK;
    if (hipEventSynchronize(P.ev[slot]) != hipSuccess)
      return false;
    if (k + RING - 1 < chunks && !issue(k + RING - 1))
      return false;
    P.pool.copy(d + off, P.pin[slot], len);
    }
  return true;
  }

constexpr size_t STAGED_FROM = 8u << 20;           // smaller transfers: the runtime's own path (its fixed costs are lower)

bool upload_bytes(void* d_dst, const void* h_src, size_t bytes, hipStream_t st)
  {
  if (bytes == 0)
    return true;
  if (bytes >= STAGED_FROM && pointer_is_pageable_host(h_src) && staged_upload(d_dst, h_src, bytes, st))
    return true;
  (void)hipGetLastError();
  return hip_ok(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st), "H2D copy");
  }

bool download_bytes(void* h_dst, const void* d_src, size_t bytes, hipStream_t st, bool wait)
  {
  if (bytes == 0)
    return true;
  if (bytes >= STAGED_FROM && pointer_is_pageable_host(h_dst) && staged_download(h_dst, d_src, bytes, st))
    return true;
  (void)hipGetLastError();
  if (!hip_ok(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st), "D2H copy"))
    return false;
  return !wait || hip_ok(hipStreamSynchronize(st), "D2H copy (wait)");
  }

} // namespace trico
