// engine.hip — the decode engine: every stream of a batch of decode jobs, ONE launch per chain kernel.
//
// What it replaces: the reference reads an archive stream by stream, component by component (trico.c:943-1668, one
// trico_decompress / LZ4_decompress_safe after the other), and round 2 of this library gave every stream of every archive a
// context, a HIP stream and a kernel launch of its own.  The format leaves one serial chain per floating-point component, so the
// number of chains in flight is the decode throughput, and that made the throughput a function of the number of hardware queues
// of the process (and, with more busy queues than the GPU has slots, made the hardware scheduler save and restore chain waves,
// which the chains' scalar-cache tables do not survive: shim.hip, "the chain decoders and their self-check").
//
// Here a batch is a table of jobs (trico_hip_decode_job: one stream each, from one archive or from many).  All float chains of
// the batch are ONE grid of k_fpc32_decode_batch, all double chains one grid of k_fpc64_decode_batch, the integer streams run one
// after the other on a third HIP stream next to them, and the self-checks (re-encode + compare) follow the chain grids on their
// streams, sharing one encoder workspace.  Three HIP streams whatever the batch holds; one status read-back at the end.  A job
// whose chain decode does not check out (or timed out, or needs table shapes the chain kernels do not have) is repeated through
// the single-stream path of shim.hip with its ladder.
//
// Workspaces belong to the engine and are kept between calls (grow-only), so a second batch of the same shape allocates nothing;
// trico_hip_decode_jobs_reserve takes the allocation out of the first one as well.  The scratch of the chain kernels (tables
// behind the scalar data cache) is memory of its own that is never handed to anything else.
#include "common.hpp"

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace trico {

namespace {

struct Engine
  {
  std::mutex mu;
  bool ready = false;
  hipStream_t s32 = nullptr, s64 = nullptr, sint = nullptr;
  hipEvent_t ev_user = nullptr, ev32 = nullptr, ev64 = nullptr, evint = nullptr;
  DevBuf jobs32, jobs64, status, in, parked, vws32, vws64, lz4ws, planes;
  uint8_t* scratch32 = nullptr; size_t scratch32_cap = 0;     // chain scratch: dedicated allocations, never pooled
  uint8_t* chain64 = nullptr; size_t chain64_cap = 0;
  uint8_t* h_tab = nullptr; size_t h_tab_cap = 0;             // pinned: job tables up, status words down
  };

// One engine per device: streams, events, chain scratch and workspaces belong to the device that was current when they were made,
// and so do the calls that may be combined into one batch (a process whose threads drive different GPUs decodes on each of them).
constexpr int MAX_DEVICES = 32;
Engine g_engines[MAX_DEVICES];

int current_device_index()
  {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES)
    {
    (void)hipGetLastError();
    dev = 0;
    }
  return dev;
  }

// the engine of the calling thread's current device (every entry point below asks once and passes it on through this reference)
thread_local Engine* t_engine = nullptr;
#define E (*t_engine)
struct EngineScope
  {
  Engine* prev;
  EngineScope() : prev(t_engine) { t_engine = &g_engines[current_device_index()]; }
  ~EngineScope() { t_engine = prev; }
  };

constexpr uint32_t STATUS_WORDS = 16;          // per job: [0] status, [4..9] payload sizes + flags of the self-check's re-encode

bool dedicated_reserve(uint8_t*& p, size_t& cap, size_t bytes)
  {
  if (bytes <= cap)
    return true;
  if (p)
    {
    (void)hipDeviceSynchronize();
    (void)hipFree(p);
    p = nullptr;
    cap = 0;
    }
  void* np = nullptr;
  const size_t want = align_up(bytes + bytes / 4, 1 << 16);
  if (!hip_ok(hipMalloc(&np, want), "hipMalloc(chain scratch)"))
    {
    trim_pool();
    if (!hip_ok(hipMalloc(&np, want), "hipMalloc(chain scratch)"))
      return false;
    }
  // the chain kernels zero what they use through the scalar cache; nothing else ever lives here
  p = (uint8_t*)np;
  cap = want;
  return true;
  }

bool pinned_reserve(size_t bytes)
  {
  if (bytes <= E.h_tab_cap)
    return true;
  if (E.h_tab)
    (void)hipHostFree(E.h_tab);
  E.h_tab = nullptr;
  E.h_tab_cap = 0;
  void* hp = nullptr;
  const size_t want = align_up(2 * bytes, 4096);
  if (!hip_ok(hipHostMalloc(&hp, want, hipHostMallocDefault), "hipHostMalloc(job tables)"))
    return false;
  E.h_tab = (uint8_t*)hp;
  E.h_tab_cap = want;
  return true;
  }

bool engine_init()
  {
  if (E.ready)
    return true;
  // The three streams must be three hardware queues (the chain grids run for seconds; whatever shares their queue waits).  The
  // runtime spreads streams of ONE priority over at most GPU_MAX_HW_QUEUES queues in an order the library does not control -
  // with 4 queues two of these streams were seen on one (32 archives: 4.1 s instead of 1.9 s) - but it keeps a pool of queues per
  // priority level, so the streams get three different levels: float chains highest (two waves per chain, latency is everything),
  // double chains normal, integer streams lowest (wide throughput kernels that fill what the chains leave).
  int prio_least = 0, prio_greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess)
    {
    (void)hipGetLastError();
    prio_least = prio_greatest = 0;
    }
  const int prio_mid = (prio_least + prio_greatest) / 2;
  if (!hip_ok(hipStreamCreateWithPriority(&E.s32, hipStreamNonBlocking, prio_greatest), "hipStreamCreate") ||
      !hip_ok(hipStreamCreateWithPriority(&E.s64, hipStreamNonBlocking, prio_mid), "hipStreamCreate") ||
      !hip_ok(hipStreamCreateWithPriority(&E.sint, hipStreamNonBlocking, prio_least), "hipStreamCreate") ||
      !hip_ok(hipEventCreateWithFlags(&E.ev_user, hipEventDisableTiming), "hipEventCreate") ||
      !hip_ok(hipEventCreateWithFlags(&E.ev32, hipEventDisableTiming), "hipEventCreate") ||
      !hip_ok(hipEventCreateWithFlags(&E.ev64, hipEventDisableTiming), "hipEventCreate") ||
      !hip_ok(hipEventCreateWithFlags(&E.evint, hipEventDisableTiming), "hipEventCreate"))
    return false;
  E.ready = true;
  return true;
  }

// Float chains per workgroup (= per compute unit, the workgroups claim most of a unit's LDS).  Measured on the MI355X with 24 /
// 96 / 192 chains of 50 M values (profiles/r03_batch_decode_experiments.txt): two chains on one unit, on different SIMDs with
// their parsers on the other two, cost each other 1.5 % (1.797 -> 1.825 s) - what round 2 saw as "two chains on one unit
// halve each other" was two chain waves on ONE SIMD, not the shared scalar cache.  Four chains per unit do not fit the 16 KiB
// scalar cache any more (4 x (4 KiB table + 2 KiB ring)): 1.4-2.7 x slower.  So: one chain per unit while the batch leaves
// half of the GPU to the integer streams decoding beside it, two beyond that; TRICO_FPC32_CHAINS_PER_CU overrides.
int chains_per_group(uint32_t nchains)
  {
  static const int v = [] { const char* e = getenv("TRICO_FPC32_CHAINS_PER_CU"); const int x = e ? atoi(e) : 0; return x >= 4 ? 4 : (x >= 2 ? 2 : (x == 1 ? 1 : 0)); }();
  return v ? v : (nchains > 128u ? 2 : 1);
  }

enum Kind { K_SKIP = 0, K_FP32, K_FP64, K_INT, K_SINGLE, K_BAD };   // K_SINGLE: straight to the single-stream path

struct Plan                                      // sizes of everything a batch needs
  {
  size_t in_bytes = 0, parked_bytes = 0, vws32 = 0, vws64 = 0, lz4ws = 0, planes = 0;
  uint32_t chains32 = 0, chains64 = 0;
  };

Kind classify(const trico_hip_decode_job& j)
  {
  if (!j.dst)
    return K_SKIP;
  if (j.n == 0)
    return K_SINGLE;                              // an empty stream still has a payload to look at (header + pad group)
  if (j.is_int)
    {
    if ((j.width != 1 && j.width != 2 && j.width != 4 && j.width != 8) || j.n > 0x7E000000u)
      return K_BAD;
    for (int c = 0; c < j.width; ++c)
      if (!j.payloads[c] || j.sizes[c] < 1)
        return K_BAD;
    return K_INT;
    }
  if (j.arity < 1 || j.arity > 3 || (j.width != 4 && j.width != 8))
    return K_BAD;
  for (int c = 0; c < j.arity; ++c)
    if (!j.payloads[c] || j.sizes[c] < 5)
      return K_BAD;
  return j.width == 4 ? K_FP32 : K_FP64;
  }

size_t job_out_bytes(const trico_hip_decode_job& j)
  {
  return (size_t)j.n * (size_t)j.width * (size_t)(j.is_int ? 1 : j.arity);
  }

Plan make_plan(const trico_hip_decode_job* jobs, int count, const Kind* kind)
  {
  Plan p;
  for (int i = 0; i < count; ++i)
    {
    const trico_hip_decode_job& j = jobs[i];
    if (kind[i] == K_SKIP || kind[i] == K_BAD || kind[i] == K_SINGLE)
      continue;
    const int units = j.is_int ? j.width : j.arity;
    for (int c = 0; c < units; ++c)
      if (!trico_hip_pointer_is_device(j.payloads[c]))
        p.in_bytes += align_up((size_t)j.sizes[c] + 16, 256);
    if (!trico_hip_pointer_is_device(j.dst))
      p.parked_bytes += align_up(job_out_bytes(j) + 16, 256);
    if (kind[i] == K_FP32)
      {
      p.chains32 += (uint32_t)j.arity;
      if (decode_check_enabled())
        {
        const size_t ws = fpc32_encode_workspace(j.n, j.arity);
        p.vws32 = ws > p.vws32 ? ws : p.vws32;
        }
      }
    else if (kind[i] == K_FP64)
      {
      p.chains64 += (uint32_t)j.arity;
      if (decode_check_enabled())
        {
        const size_t stride = align_up(fpc_bound(j.n, 8), 256);
        const bool sorted = j.n >= fpc64_sorted_threshold() && j.n <= 0x7fffffffu;
        const size_t ws = stride * j.arity + (sorted ? fpc64_sorted_workspace(j.n, j.arity) : (size_t)j.arity * 2 * ((size_t)1 << 20) * 8) + 256;
        p.vws64 = ws > p.vws64 ? ws : p.vws64;
        }
      }
    else
      {
      const size_t plane_stride = align_up((size_t)j.n + 16, 256);
      if (j.width > 1)
        p.planes = plane_stride * j.width > p.planes ? plane_stride * j.width : p.planes;
      if (j.n >= lz4_pdecode_threshold())
        {
        const size_t ws = lz4_pdecode_workspace(j.n, j.sizes, j.width);
        p.lz4ws = ws > p.lz4ws ? ws : p.lz4ws;
        }
      }
    }
  return p;
  }

bool reserve_all(const Plan& p, int count)
  {
  const size_t tab = (size_t)(p.chains32 + p.chains64) * sizeof(Fpc32ChainJob) + (size_t)count * STATUS_WORDS * 4 + 1024;
  return pinned_reserve(tab) &&
         E.jobs32.reserve((size_t)p.chains32 * sizeof(Fpc32ChainJob) + 256) &&
         E.jobs64.reserve((size_t)p.chains64 * sizeof(Fpc64ChainJob) + 256) &&
         E.status.reserve((size_t)count * STATUS_WORDS * 4 + 256) &&
         E.in.reserve(p.in_bytes + 256) && E.parked.reserve(p.parked_bytes + 256) &&
         E.vws32.reserve(p.vws32 + 256) && E.vws64.reserve(p.vws64 + 256) &&
         E.lz4ws.reserve(p.lz4ws + 256) && E.planes.reserve(p.planes + 256) &&
         dedicated_reserve(E.scratch32, E.scratch32_cap, (size_t)p.chains32 * FPC32_DECODE_TABLE_BYTES + 256) &&
         dedicated_reserve(E.chain64, E.chain64_cap, (size_t)p.chains64 * FPC64_DECODE_CHAIN_BYTES + 256);
  }

// the single-stream path with its ladder (shim.hip), on a context of its own: for jobs the batch could not settle
int redo_single(trico_hip_decode_job& j, const uint8_t* const d_pay[8], void* d_dst, int first_attempt)
  {
  trico_hip_ctx* ctx = trico_hip_ctx_create();
  if (!ctx)
    return 0;
  set_first_attempt(first_attempt);
  const int ok = j.is_int ? trico_hip_int_decode(ctx, d_pay, j.sizes, j.width, j.n, d_dst)
                          : trico_hip_fpc_decode(ctx, d_pay, j.sizes, j.arity, j.width, j.n, d_dst);
  set_first_attempt(0);
  j.other_writer = trico_hip_ctx_other_writer_streams(ctx) ? 1 : 0;
  trico_hip_ctx_destroy(ctx);
  return ok;
  }

} // namespace

} // namespace trico

using namespace trico;

extern "C" {

void trico_hip_release_workspaces(void)
  {
  if (!device_ready())
    return;
  EngineScope scope;
  std::lock_guard<std::mutex> lock(E.mu);
  (void)hipDeviceSynchronize();
  DevBuf* bufs[] = { &E.jobs32, &E.jobs64, &E.status, &E.in, &E.parked, &E.vws32, &E.vws64, &E.lz4ws, &E.planes };
  for (DevBuf* b : bufs)
    b->release();                                   // to the pool ...
  trim_pool();                                      // ... which is freed as a whole
  if (E.scratch32) (void)hipFree(E.scratch32);
  if (E.chain64) (void)hipFree(E.chain64);
  E.scratch32 = E.chain64 = nullptr;
  E.scratch32_cap = E.chain64_cap = 0;
  }

int trico_hip_decode_jobs_reserve(const trico_hip_decode_job* jobs, int count)
  {
  if (!device_ready() || !jobs || count < 0)
    return 0;
  if (count == 0)
    return 1;
  EngineScope scope;
  Kind* kind = (Kind*)malloc(sizeof(Kind) * (size_t)count);
  if (!kind)
    return 0;
  for (int i = 0; i < count; ++i)
    kind[i] = classify(jobs[i]);
  std::lock_guard<std::mutex> lock(E.mu);
  const int ok = engine_init() && reserve_all(make_plan(jobs, count, kind), count) ? 1 : 0;
  free(kind);
  return ok;
  }

// one batch, start to finish (the callers' streams have been drained by their own threads)
static int run_batch(trico_hip_decode_job* jobs, int count)
  {
  Kind* kind = (Kind*)malloc(sizeof(Kind) * (size_t)count);
  const uint8_t* (*d_pay)[8] = (const uint8_t* (*)[8])calloc((size_t)count, sizeof(const uint8_t*[8]));
  void** d_dst = (void**)calloc((size_t)count, sizeof(void*));
  if (!kind || !d_pay || !d_dst)
    {
    free(kind); free((void*)d_pay); free(d_dst);
    set_error("trico_hip_decode_jobs: out of memory");
    return 0;
    }
  int all_ok = 1;
  {
  std::lock_guard<std::mutex> lock(E.mu);
  for (int i = 0; i < count; ++i)
    {
    kind[i] = classify(jobs[i]);
    jobs[i].ok = kind[i] == K_SKIP ? 1 : 0;
    jobs[i].other_writer = 0;
    }
  const Plan plan = make_plan(jobs, count, kind);
  hipStream_t user = current_stream();
  bool launched = engine_init() && reserve_all(plan, count);
  uint32_t* d_status = (uint32_t*)E.status.p;
  if (launched)
    launched = hip_ok(hipMemsetAsync(d_status, 0, (size_t)count * STATUS_WORDS * 4, E.s32), "memset(status)") &&
               hip_ok(hipEventRecord(E.ev32, E.s32), "hipEventRecord") &&
               hip_ok(hipStreamWaitEvent(E.s64, E.ev32, 0), "hipStreamWaitEvent") &&
               hip_ok(hipStreamWaitEvent(E.sint, E.ev32, 0), "hipStreamWaitEvent");
  // ---- staging: host payloads up (on the stream that reads them), device homes for host destinations -------------------------
  size_t in_off = 0, parked_off = 0;
  for (int i = 0; launched && i < count; ++i)
    {
    if (kind[i] == K_SKIP || kind[i] == K_BAD || kind[i] == K_SINGLE)
      continue;
    const trico_hip_decode_job& j = jobs[i];
    set_current_stream(kind[i] == K_FP32 ? E.s32 : (kind[i] == K_FP64 ? E.s64 : E.sint));
    const int units = j.is_int ? j.width : j.arity;
    for (int c = 0; launched && c < units; ++c)
      {
      d_pay[i][c] = (const uint8_t*)stage_in(E.in, j.payloads[c], j.sizes[c], in_off);
      if (!d_pay[i][c])
        launched = false;
      if (d_pay[i][c] != j.payloads[c])
        in_off += align_up((size_t)j.sizes[c] + 16, 256);
      }
    if (trico_hip_pointer_is_device(j.dst))
      d_dst[i] = j.dst;
    else
      {
      d_dst[i] = E.parked.p + parked_off;
      parked_off += align_up(job_out_bytes(j) + 16, 256);
      }
    }
  // ---- float chains: one grid, then the checks --------------------------------------------------------------------------------
  Fpc32ChainJob* h32 = (Fpc32ChainJob*)E.h_tab;
  Fpc64ChainJob* h64 = (Fpc64ChainJob*)(E.h_tab + (size_t)plan.chains32 * sizeof(Fpc32ChainJob));
  uint32_t* h_status = (uint32_t*)(E.h_tab + (size_t)(plan.chains32 + plan.chains64) * sizeof(Fpc32ChainJob));
  if (launched && plan.chains32)
    {
    set_current_stream(E.s32);
    uint32_t k = 0;
    for (int i = 0; i < count; ++i)
      if (kind[i] == K_FP32)
        for (int c = 0; c < jobs[i].arity; ++c)
          h32[k++] = Fpc32ChainJob{ d_pay[i][c], (uint32_t*)d_dst[i] + c, d_status + (size_t)i * STATUS_WORDS, jobs[i].sizes[c], jobs[i].n, (uint32_t)jobs[i].arity, 0u };
    launched = hip_ok(hipMemcpyAsync(E.jobs32.p, h32, (size_t)k * sizeof(Fpc32ChainJob), hipMemcpyHostToDevice, E.s32), "H2D(job table)");
    if (launched)
      {
      ProfSpan span(TRICO_HIP_K_FPC32_DECODE);
      launched = (decode_robust_first() ? launch_fpc32_decode_robust((const Fpc32ChainJob*)E.jobs32.p, k)
                                        : launch_fpc32_decode_batch((const Fpc32ChainJob*)E.jobs32.p, k, (uint32_t*)E.scratch32, chains_per_group(k))) != 0;
      for (int i = 0; launched && i < count; ++i)
        if (kind[i] == K_FP32)
          launched = decode_sabotage(0, d_dst[i], jobs[i].n, jobs[i].arity, 4) &&
                     fpc_selfcheck_launch(d_dst[i], jobs[i].n, jobs[i].arity, 4, d_pay[i], jobs[i].sizes, E.vws32,
                                          d_status + (size_t)i * STATUS_WORDS + 4, d_status + (size_t)i * STATUS_WORDS) != 0;
      }
    }
  if (launched && plan.chains64)
    {
    set_current_stream(E.s64);
    uint32_t k = 0;
    for (int i = 0; i < count; ++i)
      if (kind[i] == K_FP64)
        for (int c = 0; c < jobs[i].arity; ++c)
          h64[k++] = Fpc64ChainJob{ d_pay[i][c], (uint64_t*)d_dst[i] + c, d_status + (size_t)i * STATUS_WORDS, jobs[i].sizes[c], jobs[i].n, (uint32_t)jobs[i].arity, 0u };
    launched = hip_ok(hipMemcpyAsync(E.jobs64.p, h64, (size_t)k * sizeof(Fpc64ChainJob), hipMemcpyHostToDevice, E.s64), "H2D(job table)");
    if (launched)
      {
      ProfSpan span(TRICO_HIP_K_FPC64_DECODE);
      launched = launch_fpc64_decode_batch((const Fpc64ChainJob*)E.jobs64.p, k, E.chain64) != 0;
      for (int i = 0; launched && i < count; ++i)
        if (kind[i] == K_FP64)
          launched = decode_sabotage(0, d_dst[i], jobs[i].n, jobs[i].arity, 8) &&
                     fpc_selfcheck_launch(d_dst[i], jobs[i].n, jobs[i].arity, 8, d_pay[i], jobs[i].sizes, E.vws64,
                                          d_status + (size_t)i * STATUS_WORDS + 4, d_status + (size_t)i * STATUS_WORDS) != 0;
      }
    }
  // ---- integer streams: one after the other beside the chains ---------------------------------------------------------------
  if (launched)
    {
    set_current_stream(E.sint);
    for (int i = 0; launched && i < count; ++i)
      {
      if (kind[i] != K_INT)
        continue;
      const trico_hip_decode_job& j = jobs[i];
      uint32_t* st = d_status + (size_t)i * STATUS_WORDS;
      const size_t plane_stride = align_up((size_t)j.n + 16, 256);
      uint8_t* d_planes = j.width > 1 ? E.planes.p : (uint8_t*)d_dst[i];
      {
      ProfSpan span(TRICO_HIP_K_LZ4_DECODE);
      if (force_serial_stage(8))
        launched = launch_lz4_decode_serial(d_pay[i], j.sizes, j.width, d_planes, plane_stride, j.n, st) != 0;
      else if (j.n >= lz4_pdecode_threshold())
        launched = launch_lz4_decode_parallel(d_pay[i], j.sizes, j.width, d_planes, plane_stride, j.n, st, E.lz4ws.p, E.lz4ws.cap) != 0;
      else
        launched = launch_lz4_decode_lds(d_pay[i], j.sizes, j.width, d_planes, plane_stride, j.n, st) != 0;
      }
      if (launched && j.width > 1)
        {
        ProfSpan span(TRICO_HIP_K_PLANES_MERGE);
        launched = launch_planes_merge(d_planes, plane_stride, j.n, j.width, d_dst[i]) != 0;
        }
      }
    }
  set_current_stream(user);
  // ---- wait, read the status words back, settle what did not check out ------------------------------------------------------
  bool synced = hip_ok(hipStreamSynchronize(E.s32), "sync(float chains)");
  synced = hip_ok(hipStreamSynchronize(E.s64), "sync(double chains)") && synced;
  synced = hip_ok(hipStreamSynchronize(E.sint), "sync(integer streams)") && synced;
  if (launched && synced)
    launched = hip_ok(hipMemcpy(h_status, d_status, (size_t)count * STATUS_WORDS * 4, hipMemcpyDeviceToHost), "D2H(status)");
  for (int i = 0; i < count; ++i)
    {
    if (kind[i] == K_SKIP)
      continue;
    trico_hip_decode_job& j = jobs[i];
    if (kind[i] == K_SINGLE)
      {
      d_dst[i] = j.dst;
      j.ok = redo_single(j, j.payloads, j.dst, 0);
      if (!j.ok)
        all_ok = 0;
      continue;
      }
    if (kind[i] == K_BAD || !launched || !synced)
      {
      if (kind[i] == K_BAD)
        set_error("trico_hip_decode_jobs: bad job");
      else
        j.ok = -1;                                // not attempted (workspaces, a launch or a copy failed): nothing is known about the stream
      all_ok = 0;
      continue;
      }
    const uint32_t st = h_status[(size_t)i * STATUS_WORDS];
    if (st == 0)
      {
      j.ok = 1;
      continue;
      }
    // The batch could not settle this stream: its values did not code back to the payload or its waves lost each other (repeat,
    // shim.hip's ladder), its tables have another shape than the chain kernels' (reference-order kernel), or it is malformed
    // (the single-stream path says how).
    if (!j.is_int && (st & ~(FPC_STATUS_CHECK | FPC_STATUS_TIMEOUT)) == 0)
      {
      stats_count_repeat();
      if (getenv("TRICO_HIP_DEBUG"))
        fprintf(stderr, "trico_hip: batch decode of job %d did not check out (status %#x), repeating it\n", i, st);
      }
    // (a chain decode of the batch that did not check out was attempt 0 of the ladder)
    j.ok = redo_single(j, d_pay[i], d_dst[i], (!j.is_int && (st & ~(FPC_STATUS_CHECK | FPC_STATUS_TIMEOUT)) == 0) ? 1 : 0);
    if (!j.ok)
      all_ok = 0;
    }
  // ---- results for host destinations -------------------------------------------------------------------------------------------
  for (int i = 0; i < count; ++i)
    if (jobs[i].ok && kind[i] != K_SKIP && d_dst[i] != jobs[i].dst)
      if (!download_bytes(jobs[i].dst, d_dst[i], job_out_bytes(jobs[i]), current_stream(), true))
        {
        jobs[i].ok = 0;
        all_ok = 0;
        }
  }
  free(kind);
  free((void*)d_pay);
  free(d_dst);
  return all_ok;
  }

// ---- callers from several threads become ONE batch -----------------------------------------------------------------------
// The reference's threading model is one archive handle per thread, so an application that decodes N archives
// does it from N threads.  A batch keeps the engine for seconds; calls that simply queued behind each other would run one
// after the other (8 threads: 8 x 1.8 s), and calls with engines of their own would put N chain grids on however many hardware
// queues the process has (round 2).  So the calls are combined: the first thread to arrive leads - it waits a moment for
// others that were started together, takes every job that has been handed in, runs ONE batch and hands the results back;
// threads that arrive while a batch is running form the next one.
namespace {
struct Waiter { trico_hip_decode_job* jobs; int count; bool done; int result; Waiter* next; };
// (one queue per device: only calls for the same device can share a batch)
struct Queue
  {
  std::mutex mu;
  std::condition_variable cv;
  Waiter* head = nullptr;
  Waiter* tail = nullptr;
  bool leading = false;
  };
Queue g_queues[MAX_DEVICES];
}

int trico_hip_decode_jobs(trico_hip_decode_job* jobs, int count)
  {
  if (!device_ready() || !jobs || count < 0)
    return 0;
  if (count == 0)
    return 1;
  // whatever produced the payloads on this thread's stream comes first (the batch may be launched by another thread)
  if (!hip_ok(hipStreamSynchronize(current_stream()), "hipStreamSynchronize"))
    return 0;
  EngineScope scope;
  Queue& Q = g_queues[current_device_index()];
  Waiter me = { jobs, count, false, 0, nullptr };
  std::unique_lock<std::mutex> lk(Q.mu);
  if (Q.tail) Q.tail->next = &me; else Q.head = &me;
  Q.tail = &me;
  while (Q.leading && !me.done)
    Q.cv.wait(lk);
  if (me.done)
    {
    if (!me.result)
      set_error("trico_hip_decode_jobs: a stream of this call's jobs failed (decoded in a batch led by another thread)");
    return me.result;
    }
  Q.leading = true;
  {
  // Threads started together should land in this batch: wait while calls keep arriving, for a time that is small against what the
  // batch itself will take (the longest float / double chain at ~35 ns per value decides that): at most 1 % of it, at most 10 ms.
  double est_us = 0.0;
  for (int i = 0; i < count; ++i)
    if (!jobs[i].is_int && jobs[i].dst)
      est_us = est_us > 0.035 * jobs[i].n ? est_us : 0.035 * jobs[i].n;
  long window = (long)(est_us / 100.0);
  window = window < 40 ? 40 : (window > 10000 ? 10000 : window);
  const auto t_start = std::chrono::steady_clock::now();
  for (;;)
    {
    const Waiter* seen = Q.tail;
    Q.cv.wait_for(lk, std::chrono::microseconds(window / 4 + 10), [] { return false; });
    const long waited = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_start).count();
    if (waited >= window || (Q.tail == seen && waited >= window / 2))
      break;
    }
  }
  Waiter* batch = Q.head;
  Q.head = Q.tail = nullptr;
  lk.unlock();
  int total = 0;
  for (Waiter* w = batch; w; w = w->next)
    total += w->count;
  int result_all;
  if (batch == &me && !me.next)
    result_all = run_batch(jobs, count);                                          // the common case: nobody else
  else
    {
    trico_hip_decode_job* all = (trico_hip_decode_job*)malloc(sizeof(trico_hip_decode_job) * (size_t)total);
    if (!all)
      result_all = -1;
    else
      {
      int at = 0;
      for (Waiter* w = batch; w; w = w->next)
        {
        memcpy(all + at, w->jobs, sizeof(trico_hip_decode_job) * (size_t)w->count);
        at += w->count;
        }
      result_all = run_batch(all, total);
      at = 0;
      for (Waiter* w = batch; w; w = w->next)
        {
        bool attempted = true;
        for (int i = 0; i < w->count; ++i)
          {
          w->jobs[i].ok = all[at + i].ok;
          w->jobs[i].other_writer = all[at + i].other_writer;
          attempted = attempted && all[at + i].ok >= 0;
          }
        // the combined batch could not be launched (its workspaces are the sum of everybody's): one caller's resource problem is
        // not the others' - every caller's jobs run again as a batch of their own
        if (!attempted)
          (void)run_batch(w->jobs, w->count);
        at += w->count;
        }
      free(all);
      }
    }
  lk.lock();
  for (Waiter* w = batch; w;)
    {
    Waiter* nx = w->next;                       // (a follower's Waiter lives on its stack: gone once it is woken with done set)
    int r = result_all < 0 ? 0 : 1;
    for (int i = 0; r && i < w->count; ++i)
      r = w->jobs[i].ok > 0 ? 1 : 0;
    w->result = r;
    w->done = true;
    w = nx;
    }
  Q.leading = false;
  Q.cv.notify_all();
  return me.result;
  }


} // extern "C"
