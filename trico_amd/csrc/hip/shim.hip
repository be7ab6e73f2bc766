// shim.hip — the extern "C" surface of include/trico/trico_hip.h: device/context management,
// host<->device staging and dispatch to the gfx950 kernels.  No compute happens on the host.
#include "common.hpp"

#include <atomic>
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace trico {

static thread_local char g_err[256] = "";
static thread_local hipStream_t g_stream = nullptr;

void set_error(const char* msg)
  {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  }

bool hip_ok(hipError_t e, const char* what)
  {
  if (e == hipSuccess)
    return true;
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  (void)hipGetLastError();
  return false;
  }

hipStream_t current_stream() { return g_stream; }

// TRICO_HIP_SERIAL: bit mask routing a stage through the reference-order kernels of k_serial.hip
// (A/B debugging): 1 = float encode, 2 = float decode, 4 = LZ4 encode, 8 = LZ4 decode; "1" alone = all.
static int serial_mask()
  {
  static int v = -1;
  if (v < 0)
    {
    const char* e = getenv("TRICO_HIP_SERIAL");
    v = e ? atoi(e) : 0;
    if (v == 1) v = 15;
    }
  return v;
  }
bool force_serial() { return serial_mask() == 15; }
bool force_serial_stage(int bit) { return (serial_mask() & bit) != 0; }

// Device workspaces are recycled across contexts (one context per archive handle): hipMalloc / hipFree of
// multi-GB buffers cost hundreds of milliseconds and an implicit device sync each.
struct PoolEntry { uint8_t* p; size_t cap; int kind; };
static PoolEntry g_pool[1024];  // eight archives read at once park ~8 x 2 streams x 5 buffers
static int g_pool_n = 0;
static std::mutex g_pool_mutex;

bool DevBuf::reserve(size_t bytes)
  {
  if (bytes <= cap)
    return true;
  release();
  {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  int best = -1;
  for (int i = 0; i < g_pool_n; ++i)
    if (g_pool[i].kind == kind && g_pool[i].cap >= bytes && (best < 0 || g_pool[i].cap < g_pool[best].cap))
      best = i;
  if (best >= 0 && g_pool[best].cap <= 4 * bytes + (64u << 20))
    {
    p = g_pool[best].p;
    cap = g_pool[best].cap;
    g_pool[best] = g_pool[--g_pool_n];
    return true;
    }
  }
  const size_t want = align_up(bytes + bytes / 8, 4096);
  void* np = nullptr;
  if (!hip_ok(hipMalloc(&np, want), "hipMalloc(workspace)"))
    {
    // out of memory: give the pooled buffers back and retry once
    trim_pool();
    if (!hip_ok(hipMalloc(&np, want), "hipMalloc(workspace)"))
      return false;
    }
  p = (uint8_t*)np;
  cap = want;
  return true;
  }

void DevBuf::release()
  {
  if (p)
    {
    // Another context (another thread, another stream) may take the buffer out of the pool right away, so no work of this
    // thread's stream may still be using it: e.g. the gather of a float stream into a device archive returns with its
    // kernel reading the segment slots.  Buffers only grow, so this wait is rare.
    (void)hipStreamSynchronize(current_stream());
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_pool_n < (int)(sizeof(g_pool) / sizeof(g_pool[0])))
      g_pool[g_pool_n++] = PoolEntry{ p, cap, kind };
    else
      (void)hipFree(p);
    }
  p = nullptr;
  cap = 0;
  }

void trim_pool()
  {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  for (int i = 0; i < g_pool_n; ++i)
    (void)hipFree(g_pool[i].p);
  g_pool_n = 0;
  }

// ---- profiling ---------------------------------------------------------------------------------
struct ProfRec { hipEvent_t e0, e1; int k; bool counted; };
static bool g_prof_on = false;
static ProfRec* g_prof = nullptr;
static size_t g_prof_n = 0, g_prof_cap = 0;
static double g_prof_ms[TRICO_HIP_K_COUNT];
static uint64_t g_prof_spans[TRICO_HIP_K_COUNT];

static void prof_drain()
  {
  for (size_t i = 0; i < g_prof_n; ++i)
    {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof[i].e1) == hipSuccess && hipEventElapsedTime(&ms, g_prof[i].e0, g_prof[i].e1) == hipSuccess)
      {
      g_prof_ms[g_prof[i].k] += ms;
      g_prof_spans[g_prof[i].k] += g_prof[i].counted ? 1 : 0;
      }
    (void)hipEventDestroy(g_prof[i].e0);
    (void)hipEventDestroy(g_prof[i].e1);
    }
  g_prof_n = 0;
  }

ProfSpan::ProfSpan(int kernel_id, bool count_launch) : k(kernel_id), active(false), counted(count_launch)
  {
  if (!g_prof_on)
    return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
    return;
  active = true;
  (void)hipEventRecord(e0, current_stream());
  }

ProfSpan::~ProfSpan() { stop(); }

void ProfSpan::stop()
  {
  if (!active)
    return;
  active = false;
  (void)hipEventRecord(e1, current_stream());
  if (g_prof_n == g_prof_cap)
    {
    const size_t nc = g_prof_cap ? g_prof_cap * 2 : 256;
    ProfRec* np = (ProfRec*)realloc(g_prof, nc * sizeof(ProfRec));
    if (!np)
      return;
    g_prof = np;
    g_prof_cap = nc;
    }
  g_prof[g_prof_n++] = ProfRec{ e0, e1, k, counted };
  }

static int g_device_state = 0;   // 0 unknown, 1 ok, -1 none
static thread_local uint32_t g_stats[4] = { 0, 0, 0, 0 };      // words 0, 1: the last trico_hip_int_encode of this thread
static std::atomic<uint32_t> g_repeats{ 0 }, g_other_writer{ 0 };   // words 2, 3: process-wide (a batch may be led by another thread)
static std::atomic<uint32_t> g_recoded_order{ 0 }, g_recoded_sentinel{ 0 }, g_recoded_scan{ 0 };   // trico_hip_encode_stats, trico_hip_encode_scan_recodes
static std::atomic<int> g_strict{ -1 };                     // trico_hip_set_strict: -1 = what TRICO_HIP_STRICT says
static std::atomic<int> g_verify{ -1 };                     // trico_hip_set_encode_verify: -1 = what TRICO_HIP_ENCODE_VERIFY says
static std::atomic<uint64_t> g_verified_streams{ 0 }, g_verified_values{ 0 }, g_verify_mismatch{ 0 };

static bool strict_reader()
  {
  const int s = g_strict.load();
  if (s >= 0)
    return s != 0;
  static const bool env = [] { const char* e = getenv("TRICO_HIP_STRICT"); return e && e[0] == '1'; }();
  return env;
  }

static bool encode_verify_on()
  {
  const int s = g_verify.load();
  if (s >= 0)
    return s != 0;
  static const bool env = [] { const char* e = getenv("TRICO_HIP_ENCODE_VERIFY"); return e && e[0] == '1'; }();
  return env;
  }

void stats_count_repeat() { g_repeats += 1; }
void set_current_stream(hipStream_t s) { g_stream = s; }

bool device_ready()
  {
  if (g_device_state == 0)
    {
    // (No process setting is touched here.  Round 2 asked for GPU_MAX_HW_QUEUES=16 because every stream of an archive was a
    // kernel launch of its own on a HIP stream of its own; the decode engine (engine.hip) launches all chains of a batch as one
    // grid and needs three HIP streams whatever the batch holds.)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
      {
      (void)hipGetLastError();
      g_device_state = -1;
      }
    else
      g_device_state = 1;
    }
  if (g_device_state < 0)
    set_error("no HIP device available: the trico hot path has no CPU fallback");
  return g_device_state > 0;
  }

// stage `bytes` from src (host or device) so that kernels can read it; returns a device pointer
const void* stage_in(DevBuf& buf, const void* src, size_t bytes, size_t offset)
  {
  if (trico_hip_pointer_is_device(src))
    return src;
  if (!upload_bytes(buf.p + offset, src, bytes, current_stream()))
    return nullptr;
  return buf.p + offset;
  }

// the six words k_fpc32_offsets mirrored into the pinned buffer (launch_fpc32_encode's h_sizes)
constexpr int MIRROR_AT = 32;
static int read_mirrored_sizes(trico_hip_ctx* ctx, uint32_t six[6])
  {
  TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
  memcpy(six, ctx->h_pinned + MIRROR_AT, 6 * sizeof(uint32_t));
  return 1;
  }

static int read_back_words(trico_hip_ctx* ctx, const uint32_t* d_words, int count, uint32_t* host)
  {
  TRICO_HIP_TRY(hipMemcpyAsync(ctx->h_pinned, d_words, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, current_stream()));
  TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
  memcpy(host, ctx->h_pinned, sizeof(uint32_t) * count);
  return 1;
  }

} // namespace trico

using namespace trico;

extern "C" {

int trico_hip_available(void) { return device_ready() ? 1 : 0; }

const char* trico_hip_last_error(void) { return g_err; }

trico_hip_ctx* trico_hip_ctx_create(void)
  {
  if (!device_ready())
    return nullptr;
  trico_hip_ctx* ctx = new (std::nothrow) trico_hip_ctx();
  if (!ctx)
    return nullptr;
  void* hp = nullptr;
  if (!hip_ok(hipHostMalloc(&hp, 64 * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc") ||
      !ctx->aux.reserve(1 << 16))
    {
    if (hp) (void)hipHostFree(hp);
    delete ctx;
    return nullptr;
    }
  ctx->h_pinned = (uint32_t*)hp;
  return ctx;
  }

void trico_hip_ctx_destroy(trico_hip_ctx* ctx)
  {
  if (!ctx)
    return;
  (void)hipStreamSynchronize(current_stream());
  ctx->in.release();
  ctx->out.release();
  ctx->tmp.release();
  ctx->aux.release();
  ctx->ws.release();
  ctx->unit.release();
  ctx->vws.release();
  ctx->chain.release();
  if (ctx->h_pinned)
    (void)hipHostFree(ctx->h_pinned);
  delete ctx;
  }

void trico_hip_set_stream(void* hip_stream) { g_stream = (hipStream_t)hip_stream; }

int trico_hip_synchronize(void)
  {
  if (!device_ready())
    return 0;
  TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
  return 1;
  }

int trico_hip_pointer_is_device(const void* p)
  {
  if (!p || g_device_state < 0)
    return 0;
  if (!device_ready())
    return 0;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess)
    {
    (void)hipGetLastError();
    return 0;
    }
  return attr.type == hipMemoryTypeDevice ? 1 : 0;
  }

uint64_t trico_hip_device_free_bytes(void)
  {
  if (!device_ready())
    return 0;
  size_t fr = 0, total = 0;
  if (hipMemGetInfo(&fr, &total) != hipSuccess)
    {
    (void)hipGetLastError();
    return 0;
    }
  return (uint64_t)fr;
  }

// device allocations handed to the host code (archive buffers) come from the same pool as the workspaces
struct LiveAlloc { void* p; size_t cap; };
static LiveAlloc g_live[64];
static int g_live_n = 0;
static std::mutex g_live_mutex;

void* trico_hip_device_alloc(size_t bytes)
  {
  if (!device_ready())
    return nullptr;
  DevBuf b;
  if (!b.reserve(bytes ? bytes : 1))
    return nullptr;
  std::lock_guard<std::mutex> lock(g_live_mutex);
  if (g_live_n == (int)(sizeof(g_live) / sizeof(g_live[0])))
    {
    (void)hipFree(b.p);              // table full: untracked allocations are simply freed on release
    void* p = nullptr;
    return hip_ok(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc") ? p : nullptr;
    }
  g_live[g_live_n++] = LiveAlloc{ b.p, b.cap };
  return b.p;
  }

void trico_hip_device_free(void* p)
  {
  if (!p)
    return;
  {
  std::lock_guard<std::mutex> lock(g_live_mutex);
  for (int i = 0; i < g_live_n; ++i)
    if (g_live[i].p == p)
      {
      DevBuf b;
      b.p = (uint8_t*)p;
      b.cap = g_live[i].cap;
      g_live[i] = g_live[--g_live_n];
      b.release();                   // back to the pool
      return;
      }
  }
  (void)hipFree(p);
  }

int trico_hip_copy(void* dst, const void* src, size_t bytes)
  {
  if (bytes == 0)
    return 1;
  if (!device_ready())
    return 0;
  const bool dst_dev = trico_hip_pointer_is_device(dst) != 0, src_dev = trico_hip_pointer_is_device(src) != 0;
  if (dst_dev && !src_dev)
    {
    if (!upload_bytes(dst, src, bytes, current_stream()))
      return 0;
    }
  else if (src_dev && !dst_dev)
    return download_bytes(dst, src, bytes, current_stream(), true) ? 1 : 0;
  else
    TRICO_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, current_stream()));
  TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
  return 1;
  }

// ---- framing of a device-resident archive -----------------------------------------------------------
struct NcompTable { uint8_t n[21]; };

__global__ void k_walk_frames(const uint8_t* __restrict__ data, uint64_t size, uint64_t pos, NcompTable nc, trico_hip_frame_bytes* __restrict__ out,
                              int cap, uint32_t* __restrict__ tail)
  {
  // one thread: a frame is found only through the size fields of the frames before it
  uint8_t* head = (uint8_t*)(tail + 1);
  for (uint32_t k = 0; k < 8u; ++k)
    head[k] = k < size ? data[k] : 0u;
  int n = 0;
  while (n < cap && pos < size)
    {
    trico_hip_frame_bytes f;
    f.tpos = pos;
    f.ncomp = 0;
    for (int c = 0; c < 8; ++c) { f.size_pos[c] = 0; f.size_valid[c] = 0; }
    for (int k = 0; k < 40; ++k) f.bytes[k] = 0;
    f.nbytes_head = size - pos < 5u ? (uint32_t)(size - pos) : 5u;
    for (uint32_t k = 0; k < f.nbytes_head; ++k)
      f.bytes[k] = data[pos + k];
    const uint32_t type = f.bytes[0];
    const uint32_t want = (f.nbytes_head == 5u && type <= 20u) ? nc.n[type] : 0u;
    uint64_t q = pos + 5u;
    bool whole = want != 0u;
    for (uint32_t c = 0; c < want; ++c)
      {
      if (q + 4u > size) { whole = false; break; }
      uint32_t sz = 0;
      for (uint32_t k = 0; k < 4u; ++k)
        {
        f.bytes[5u + 4u * c + k] = data[q + k];
        sz |= (uint32_t)data[q + k] << (8u * k);
        }
      f.size_pos[c] = q;
      f.size_valid[c] = 1;
      f.ncomp = c + 1u;
      q += 4u + (uint64_t)sz;
      if (q > size) { whole = false; break; }
      }
    out[n++] = f;
    if (!whole)
      break;
    pos = q;
    }
  tail[0] = (uint32_t)n;
  }

int trico_hip_walk_frames(const uint8_t* d_data, uint64_t size, uint64_t pos, const uint8_t ncomp_of_type[21],
                          trico_hip_frame_bytes* out, int cap, uint8_t head8[8])
  {
  if (!device_ready() || !d_data || !ncomp_of_type || !out || cap < 1 || cap > 256)
    return -1;
  static std::mutex mu;
  static uint8_t* d_bufs[32] = { nullptr };   // per device: 256 frame records + count + header bytes
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32)
    dev = 0;
  uint8_t*& d_buf = d_bufs[dev];
  const size_t rec = sizeof(trico_hip_frame_bytes) * 256;
  if (!d_buf && !hip_ok(hipMalloc((void**)&d_buf, rec + 64), "hipMalloc(frame records)"))
    return -1;
  NcompTable nc;
  memcpy(nc.n, ncomp_of_type, 21);
  uint32_t* d_tail = (uint32_t*)(d_buf + rec);
  hipStream_t st = current_stream();
  hipLaunchKernelGGL(k_walk_frames, dim3(1), dim3(1), 0, st, d_data, size, pos, nc, (trico_hip_frame_bytes*)d_buf, cap, d_tail);
  if (!hip_ok(hipGetLastError(), "k_walk_frames"))
    return -1;
  // records and tail are adjacent when cap == 256; otherwise two ranges -> copy the used records and the tail together through
  // one staging image: the tail is copied to sit right behind the `cap` records
  static uint8_t h_img[sizeof(trico_hip_frame_bytes) * 256 + 64];
  if (!hip_ok(hipMemcpyAsync(h_img, d_buf, rec + 64, hipMemcpyDeviceToHost, st), "D2H(frame records)") ||
      !hip_ok(hipStreamSynchronize(st), "D2H(frame records)"))
    return -1;
  const uint32_t n = *(const uint32_t*)(h_img + rec);
  if (head8)
    memcpy(head8, h_img + rec + 4, 8);
  const int got = (int)(n > (uint32_t)cap ? (uint32_t)cap : n);
  memcpy(out, h_img, sizeof(trico_hip_frame_bytes) * (size_t)got);
  return got;
  }

// ---- floating point -----------------------------------------------------------------------------

int trico_hip_fpc_encode(trico_hip_ctx* ctx, const void* src, uint32_t n, int arity, int width, uint32_t sizes[3])
  {
  return trico_hip_fpc_encode_ex(ctx, src, n, arity, width, width == 4 ? 4u : 20u, width == 4 ? 10u : 20u, sizes);
  }

// One queue of launches from the values to the framed stream body in a device buffer: encode, then the gather placed by the sizes in
// device memory (`u32 bytes, payload` per component from d_first on); the host waits once, at the end, for the sizes.  1 = done,
// 0 = error, -1 = not this way (not a float stream of the API's table sizes, empty, destination not in device memory, full
// verification asked for, or a flag of the one-sweep coder was raised): the caller takes trico_hip_fpc_encode + fetch_payloads, which
// deals with every case; nothing the caller may rely on has been written.
int trico_hip_fpc_encode_place(trico_hip_ctx* ctx, const void* src, uint32_t n, int arity, int width, void* d_first, uint32_t sizes[3])
  {
  if (!ctx || !device_ready())
    return 0;
  if (width != 4 || n == 0 || arity < 1 || arity > 3 || !src || !d_first || force_serial_stage(1) || encode_verify_on() ||
      !trico_hip_pointer_is_device(d_first))
    return -1;
  const size_t in_bytes = (size_t)n * arity * 4u;
  if (!ctx->in.reserve(in_bytes + 16))
    return 0;
  ctx->out_count = 0;
  ctx->out_in_slots = false;
  const void* d_src = stage_in(ctx->in, src, in_bytes);
  if (!d_src)
    return 0;
  uint32_t* d_sizes = (uint32_t*)ctx->aux.p;
  const size_t ws = fpc32_encode_workspace(n, arity);
  if (!ctx->tmp.reserve(ws))
    return 0;
  {
  ProfSpan span(TRICO_HIP_K_FPC32_ENCODE);
  if (!launch_fpc32_encode(d_src, n, arity, nullptr, 0, d_sizes, ctx->tmp.p, ctx->tmp.cap, FPC32_CODER_AUTO, ctx->h_pinned + MIRROR_AT) ||
      !launch_fpc32_gather_framed(n, arity, ctx->tmp.p, (uint8_t*)d_first, d_sizes))
    return 0;
  }
  uint32_t six[6] = { 0, 0, 0, 0, 0, 0 };
  if (!read_mirrored_sizes(ctx, six))
    return 0;
  for (int c = 0; c < arity; ++c)
    if (six[3 + c] != 0)
      return -1;                    // (what was placed is not the stream; the caller's other way codes it again and counts the flag)
  for (int c = 0; c < arity; ++c)
    sizes[c] = six[c];
  return 1;
  }

int trico_hip_fpc_encode_ex(trico_hip_ctx* ctx, const void* src, uint32_t n, int arity, int width, uint32_t e1, uint32_t e2,
                            uint32_t sizes[3])
  {
  if (!ctx || !device_ready())
    return 0;
  if (arity < 1 || arity > 3 || (width != 4 && width != 8) || (n != 0 && !src))
    {
    set_error("trico_hip_fpc_encode: bad arguments");
    return 0;
    }
  const uint32_t emax1 = width == 4 ? 4u : 20u, emax2 = width == 4 ? 10u : 20u;
  // the reference's normalisation (fpsc.c:88-93, 578-583): odd exponents are rounded down, everything above 30 is 30
  e1 &= ~1u; e2 &= ~1u;
  if (e1 > 30u) e1 = 30u;
  if (e2 > 30u) e2 = 30u;
  const bool defaults = e1 == emax1 && e2 == emax2;
  const size_t in_bytes = (size_t)n * arity * width;
  const size_t stride = align_up(fpc_bound(n, width), 256);
  if (!ctx->in.reserve(in_bytes + 16) || !ctx->out.reserve(stride * arity))
    return 0;
  ctx->out_stride = stride;
  ctx->out_count = 0;
  ctx->out_in_slots = false;
  const void* d_src = n ? stage_in(ctx->in, src, in_bytes) : ctx->in.p;
  if (!d_src)
    return 0;
  uint32_t* d_sizes = (uint32_t*)ctx->aux.p;
  {
  ProfSpan span(width == 4 ? TRICO_HIP_K_FPC32_ENCODE : TRICO_HIP_K_FPC64_ENCODE);
  uint64_t* d_tables = nullptr;
  const bool sorted64 = width == 8 && defaults && !force_serial_stage(1) && n >= fpc64_sorted_threshold() && n <= 0x7fffffffu;
  const bool lds_tables = width == 4 && e1 <= 4u && e2 <= 10u;        // the reference-order kernel keeps such float tables in LDS
  if (!(width == 4 && defaults && !force_serial_stage(1)) && !sorted64 && !lds_tables)
    {
    // tables of 2^e1 + 2^e2 entries per component (fpsc.c:96-97, 592-593), zeroed per call.  The API's own (20,20) doubles
    // need 16 MiB per component; exponent 30 means 8 / 16 GiB per component, which the device has.
    const size_t tb = (size_t)arity * (((size_t)1 << e1) + ((size_t)1 << e2)) * (size_t)width;
    if (!ctx->tmp.reserve(tb))
      return 0;
    TRICO_HIP_TRY(hipMemsetAsync(ctx->tmp.p, 0, tb, current_stream()));
    d_tables = (uint64_t*)ctx->tmp.p;
    }
  if (width == 4 && defaults && !force_serial_stage(1))
    {
    const size_t ws = fpc32_encode_workspace(n, arity);
    if (!ctx->tmp.reserve(ws))
      return 0;
    if (!launch_fpc32_encode(d_src, n, arity, ctx->out.p, stride, d_sizes, ctx->tmp.p, ctx->tmp.cap, FPC32_CODER_AUTO, ctx->h_pinned + MIRROR_AT))
      return 0;
    span.stop();                    // (the kernels, not the host's wait for them)
    ctx->out_in_slots = n != 0;
    ctx->slots_n = n;
    ctx->slots_arity = arity;
    for (int c = 0; c < 3; ++c)
      ctx->out_materialized[c] = false;
    }
  else if (sorted64)
    {
    // large streams: the table lookups become sorts (k_fpc64_sort.hip); the tables themselves are not needed
    const size_t ws = fpc64_sorted_workspace(n, arity);
    if (!ctx->ws.reserve(ws))
      return 0;
    if (!launch_fpc64_encode_sorted(d_src, n, arity, ctx->out.p, stride, d_sizes, ctx->ws.p, ctx->ws.cap))
      return 0;
    }
  else if (width == 8 && defaults && !force_serial_stage(1))
    {
    if (!launch_fpc64_encode(d_src, n, arity, ctx->out.p, stride, d_sizes, d_tables))
      return 0;
    }
  else if (!launch_fpc_encode_serial(d_src, n, arity, width, ctx->out.p, stride, d_sizes, d_tables, e1, e2))
    return 0;
  }
  if (width == 4 && defaults && !force_serial_stage(1) && n != 0)
    {
    // sizes + the flags of the one-sweep coder's waves (k_fpc32_sweep.hip): a sampled step whose LDS exchange was not in lane
    // order, or a payload equal to the coder's "never written" mark.  Raised, the payloads are not to be trusted: the stream is
    // coded again by the two-sweep coder with ballots, which depends on neither; a device that showed the first is not asked again.
    uint32_t six[6] = { 0, 0, 0, 0, 0, 0 };
    if (!read_mirrored_sizes(ctx, six))
      return 0;
    uint32_t raised = 0;
    for (int c = 0; c < arity; ++c)
      raised |= six[3 + c];
#ifdef TRICO_HIP_TEST_HOOKS
    if (raised != 0 && getenv("TRICO_HIP_ENCODE_KEEP_FLAGGED"))
      {
      // test hook: count the flags, keep what the one-sweep coder wrote (tools/dbg_sweep.py compares it with the oracle)
      if (raised & FPC32_FLAG_ORDER) g_recoded_order += 1;
      if (raised & FPC32_FLAG_SENTINEL) g_recoded_sentinel += 1;
      raised = 0;
      }
#endif
    if (raised != 0)
      {
      if (raised & FPC32_FLAG_ORDER)
        {
        fpc32_distrust_lane_order();
        g_recoded_order += 1;
        }
      if (raised & FPC32_FLAG_SENTINEL)
        g_recoded_sentinel += 1;
      if (raised & FPC32_FLAG_SCAN)
        g_recoded_scan += 1;
      if (getenv("TRICO_HIP_DEBUG"))
        fprintf(stderr, "trico_hip: float encoder flags 0x%x (1: LDS exchange out of lane order, 2: payload equals the table mark, "
                        "4: a wait in the scan kernel ran out); coding the stream again with the ballot coder\n", raised);
      raised = 0;
      if (!launch_fpc32_encode(d_src, n, arity, ctx->out.p, stride, d_sizes, ctx->tmp.p, ctx->tmp.cap, FPC32_CODER_BALLOT) ||
          !read_back_words(ctx, d_sizes, 6, six))
        return 0;
      for (int c = 0; c < arity; ++c)
        raised |= six[3 + c];
      if (raised != 0)
        {
        set_error("float encoder: the ballot coder raised a flag");
        return 0;
        }
      }
    for (int c = 0; c < arity; ++c)
      ctx->out_sizes[c] = six[c];
    if (encode_verify_on() && fpc32_code_sweep_mode() != 0)
      {
      // Opt-in full verification (trico_hip_set_encode_verify): the write-side guard above samples 16 segments x 64 steps of the
      // one-sweep coder; here EVERY value is coded again by the two-sweep coder with ballots - which rests neither on the order the
      // LDS unit applies an exchange in nor on the table mark - in a workspace of its own, and the payloads are compared byte for
      // byte on the device.  A difference is counted, reported on stderr, and the ballot coder's payload is what the stream gets.
      uint8_t* pay[3] = { nullptr, nullptr, nullptr };
      const uint8_t* cpay[3] = { nullptr, nullptr, nullptr };
      for (int c = 0; c < arity; ++c)
        cpay[c] = pay[c] = ctx->out.p + (size_t)c * stride;
      if (!launch_fpc32_gather_all(n, arity, ctx->tmp.p, pay))
        return 0;
      for (int c = 0; c < arity; ++c)
        ctx->out_materialized[c] = true;
      uint32_t* d_vstatus = (uint32_t*)ctx->aux.p + 32;
      uint32_t* d_vsizes = (uint32_t*)ctx->aux.p + 64;
      TRICO_HIP_TRY(hipMemsetAsync(d_vstatus, 0, 4, current_stream()));
      const size_t vws = fpc32_encode_workspace(n, arity);
      uint32_t verdict = 0;
      if (!ctx->vws.reserve(vws) ||
          !launch_fpc32_encode(d_src, n, arity, nullptr, 0, d_vsizes, ctx->vws.p, ctx->vws.cap, FPC32_CODER_BALLOT) ||
          !launch_fpc32_compare(n, arity, ctx->vws.p, d_vsizes, cpay, ctx->out_sizes, d_vstatus, 0x100u) ||
          !read_back_words(ctx, d_vstatus, 1, &verdict))
        return 0;
      g_verified_streams += 1;
      g_verified_values += (uint64_t)n * (uint64_t)arity;
      if (verdict != 0)
        {
        g_verify_mismatch += 1;
        fprintf(stderr, "trico_hip: ENCODE VERIFICATION FAILED (components 0x%x of a stream of %u x %d floats): the one-sweep coder and the ballot "
                        "coder disagree; writing the ballot coder's payload\n", verdict >> 8, n, arity);
        fpc32_distrust_lane_order();
        if (!launch_fpc32_encode(d_src, n, arity, ctx->out.p, stride, d_sizes, ctx->tmp.p, ctx->tmp.cap, FPC32_CODER_BALLOT) ||
            !read_back_words(ctx, d_sizes, 6, six))
          return 0;
        for (int c = 0; c < arity; ++c)
          {
          ctx->out_sizes[c] = six[c];
          ctx->out_materialized[c] = false;
          }
        }
      }
    }
  else if (!read_back_words(ctx, d_sizes, arity, ctx->out_sizes))
    return 0;
  ctx->out_count = arity;
  for (int c = 0; c < arity; ++c)
    sizes[c] = ctx->out_sizes[c];
  return 1;
  }

// ---- the chain decoders and their self-check -----------------------------------------------------------------------------
// k_fpc32_decode / k_fpc64_decode keep their predictor tables behind the scalar data cache, as dirty lines (scalar stores).  That
// state belongs to the compute unit the wave runs on, and it does not travel: when a process has more hardware queues in use
// than the GPU has queue slots (measured on the MI355X: all of GPU_MAX_HW_QUEUES=24 or 32 busy; 16 busy queues are fine), the
// hardware scheduler time-slices the queues, long-running waves are saved and restored, and a chain that comes back on another
// compute unit reads its table from memory while newer entries still sit in the cache it left.  The decode then goes wrong
// from some value on — silently: the payload is well-formed.  (Twelve and more archives decoded at once with 32 queues: 20-40 %
// of the noisy streams; eight archives, or sixteen queues: none in hundreds of runs; write-through stores (glc) cure it at eight
// times the time per value.)  Therefore:
//   * the library itself no longer creates that situation: all chains of a batch are ONE kernel launch on one of three HIP
//     streams (engine.hip), whatever the number of archives, so it needs no queue setting and makes none (round 2 set
//     GPU_MAX_HW_QUEUES=16 here);
//   * the two waves of a chain wait for each other with a bound (k_fpc32_decode.hip: SPIN_LIMIT_*): counters that went stale
//     end the kernel with FPC_STATUS_TIMEOUT instead of hanging it, and the host repeats the stream;
//   * every decode of these kernels is CHECKED: the decoded values are coded again with the throughput encoder (1 ms for 50 M
//     float vertices, 25 ms for doubles) and the bytes compared with the payload on the device.  The coder is a deterministic
//     function of the values and the decoder of the payload, so equal payloads mean the values are the payload's values.  A
//     stream that fails is decoded again: twice more by the same kernel, then by the reference-order kernel of k_serial.hip,
//     which does not use the scalar cache (20x slower; trico_hip_last_stats word 2 counts the repeats).  That last rung is
//     checked as well; values of it that do not code back mean a payload the reference's encoder did not write (word 3).
//     TRICO_HIP_DECODE_CHECK=0 switches the check off (measurements only).

} // extern "C"

namespace trico {

bool decode_check_enabled()
  {
  static const bool on = [] { const char* e = tune_env("TRICO_HIP_DECODE_CHECK"); return !(e && e[0] == '0'); }();
  return on;
  }

// status bits 0x100 << c: component c of the re-encode differs from the payload
constexpr uint32_t CHECK_BITS = FPC_STATUS_CHECK;

// Codes the n x arity values at d_vals again (throughput encoders, workspace `vws`, payload sizes to d_vsizes[0..2]) and compares
// the bytes with the payloads the values were decoded from: bit 0x100 << c of *d_status is set where component c differs.
// Everything is queued on current_stream(); nothing waits.
int fpc_selfcheck_launch(const void* d_vals, uint32_t n, int arity, int width, const uint8_t* const d_pay[3], const uint32_t sizes[3],
                         DevBuf& vws, uint32_t* d_vsizes, uint32_t* d_status)
  {
  if (n == 0 || !decode_check_enabled())
    return 1;
  if (width == 4)
    {
    const size_t ws = fpc32_encode_workspace(n, arity);
    if (!vws.reserve(ws))
      return 0;
    // (always the two-sweep coder with ballots: the check depends neither on the order the LDS unit applies an exchange in nor on
    // anything else the coder that wrote the archive rests on)
    return launch_fpc32_encode(d_vals, n, arity, nullptr, 0, d_vsizes, vws.p, vws.cap, FPC32_CODER_BALLOT) &&
           launch_fpc32_compare(n, arity, vws.p, d_vsizes, d_pay, sizes, d_status, 0x100u);
    }
  const size_t stride = align_up(fpc_bound(n, 8), 256);
  const bool sorted = n >= fpc64_sorted_threshold() && n <= 0x7fffffffu;
  const size_t wsb = sorted ? fpc64_sorted_workspace(n, arity) : (size_t)arity * 2 * ((size_t)1 << 20) * 8;
  if (!vws.reserve(stride * arity + wsb + 256))
    return 0;
  uint8_t* out = vws.p;
  uint8_t* wsp = vws.p + stride * arity;
  if (sorted)
    {
    if (!launch_fpc64_encode_sorted(d_vals, n, arity, out, stride, d_vsizes, wsp, wsb))
      return 0;
    }
  else
    {
    TRICO_HIP_TRY(hipMemsetAsync(wsp, 0, wsb, current_stream()));
    if (!launch_fpc64_encode(d_vals, n, arity, out, stride, d_vsizes, (uint64_t*)wsp))
      return 0;
    }
  for (int c = 0; c < arity; ++c)
    if (!launch_bytes_compare(out + (size_t)c * stride, d_pay[c], sizes[c], d_vsizes + c, d_status, 0x100u << c))
      return 0;
  return 1;
  }

} // namespace trico

extern "C" {

static int fpc_check_launch(trico_hip_ctx* ctx, uint32_t* d_status)
  {
  uint32_t* d_vsizes = (uint32_t*)ctx->aux.p + 64;                 // the status words are at the start of aux
  return fpc_selfcheck_launch(ctx->chk_dst, ctx->chk_n, ctx->chk_arity, ctx->chk_width, ctx->chk_pay, ctx->chk_sizes, ctx->vws, d_vsizes, d_status);
  }

} // extern "C"

namespace trico {

// Test hook, compiled only into libtrico_testhooks.so (-DTRICO_HIP_TEST_HOOKS, trico_amd/build.py): TRICO_HIP_DECODE_SABOTAGE=k
// damages the output of the first k chain decodes of every stream (one bit of the last value) before the check sees it, so that
// the repeat path can be exercised on purpose (tests/test_gpu_selfcheck.py).  The product library has no such switch.
#ifdef TRICO_HIP_TEST_HOOKS
__global__ void k_flip_bit(uint8_t* p) { p[0] ^= 1u; }
int decode_sabotage(int attempt, void* d_vals, uint32_t n, int arity, int width)
  {
  static const int count = [] { const char* e = getenv("TRICO_HIP_DECODE_SABOTAGE"); return e ? atoi(e) : 0; }();
  if (n && attempt < count)
    hipLaunchKernelGGL(k_flip_bit, dim3(1), dim3(1), 0, current_stream(), (uint8_t*)d_vals + ((size_t)n * arity - 1) * width);
  return 1;
  }
#else
int decode_sabotage(int, void*, uint32_t, int, int) { return 1; }
#endif

// Attempt number of the chain decode the single-stream path starts with: 0, or 1 when the decode engine (engine.hip) has
// already made attempt 0 as part of a batch and hands the stream over to be repeated.
static thread_local int g_first_attempt = 0;
void set_first_attempt(int a) { g_first_attempt = a; }

bool decode_robust_first()
  {
  static const bool on = [] { const char* e = getenv("TRICO_HIP_DECODE_ROBUST"); return e && e[0] == '1'; }();
  return on;
  }

} // namespace trico

extern "C" {

// launches the chain decoder for the stream remembered in ctx->chk_* and its check (attempt 0, 1, 2)
static int fpc_chain_decode(trico_hip_ctx* ctx, int attempt)
  {
  const int arity = ctx->chk_arity, width = ctx->chk_width;
  const uint32_t n = ctx->chk_n;
  uint32_t* d_status = (uint32_t*)ctx->aux.p;
  // tables and rings of the chain kernels: memory that is only ever chain scratch (a dirty line written back late from a compute
  // unit the wave has left can then only land in another chain's scratch, whose decode is checked)
  const size_t tb = width == 8 ? (size_t)arity * 2 * ((size_t)1 << 20) * 8 + 3 * FPC64_DECODE_SCRATCH_BYTES : 3 * FPC32_DECODE_TABLE_BYTES;
  if (!ctx->chain.reserve(tb))
    return 0;
  int ok;
  if (width == 4 && (attempt >= 2 || decode_robust_first()))
    {
    // float streams, third attempt (or TRICO_HIP_DECODE_ROBUST=1): the decoder that keeps everything in LDS and registers
    Fpc32ChainJob h[3];
    for (int c = 0; c < arity; ++c)
      h[c] = Fpc32ChainJob{ ctx->chk_pay[c], (uint32_t*)ctx->chk_dst + c, d_status, ctx->chk_sizes[c], n, (uint32_t)arity, 0u };
    Fpc32ChainJob* d_jobs = (Fpc32ChainJob*)((uint8_t*)ctx->aux.p + 1024);
    TRICO_HIP_TRY(hipMemcpyAsync(d_jobs, h, sizeof(Fpc32ChainJob) * (size_t)arity, hipMemcpyHostToDevice, current_stream()));
    TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));          // h lives on this stack
    ok = launch_fpc32_decode_robust(d_jobs, (uint32_t)arity);
    }
  else
    ok = width == 4 ? launch_fpc32_decode(ctx->chk_pay, ctx->chk_sizes, arity, n, ctx->chk_dst, d_status, (uint32_t*)ctx->chain.p)
                    : launch_fpc64_decode(ctx->chk_pay, ctx->chk_sizes, arity, n, ctx->chk_dst, (uint64_t*)ctx->chain.p, d_status);
  return ok && decode_sabotage(attempt, ctx->chk_dst, n, arity, width) && fpc_check_launch(ctx, d_status);
  }

// Stages the payloads, launches the decode into d_dst (ctx->out when NULL) on current_stream() and queues the
// read-back of the status word; nothing waits.  decode_complete() below does.
static int fpc_decode_launch(trico_hip_ctx* ctx, const uint8_t* const payloads[3], const uint32_t sizes[3],
                             int arity, int width, uint32_t n, void* d_dst)
  {
  // stage host payloads back to back (256-byte aligned) in ctx->in
  size_t total = 0, offs[3] = { 0, 0, 0 };
  for (int c = 0; c < arity; ++c)
    {
    if (sizes[c] < 5 || !payloads[c])
      {
      set_error("trico_hip_fpc_decode: payload too short");
      return 0;
      }
    offs[c] = total;
    total += align_up((size_t)sizes[c] + 16, 256);
    }
  const uint8_t* d_pay[3] = { nullptr, nullptr, nullptr };
  bool any_host = false;
  for (int c = 0; c < arity; ++c)
    any_host |= !trico_hip_pointer_is_device(payloads[c]);
  if (any_host && !ctx->in.reserve(total))
    return 0;
  for (int c = 0; c < arity; ++c)
    {
    d_pay[c] = (const uint8_t*)stage_in(ctx->in, payloads[c], sizes[c], offs[c]);
    if (!d_pay[c])
      return 0;
    }
  const size_t out_bytes = (size_t)n * arity * width;
  if (!d_dst)
    {
    if (!ctx->out.reserve(out_bytes + 16))
      return 0;
    ctx->out_count = 0;
    d_dst = ctx->out.p;
    }
  uint32_t* d_status = (uint32_t*)ctx->aux.p;
  TRICO_HIP_TRY(hipMemsetAsync(d_status, 0, 64, current_stream()));
  {
  ProfSpan span(width == 4 ? TRICO_HIP_K_FPC32_DECODE : TRICO_HIP_K_FPC64_DECODE);
  // The table sizes come from the payload's hash_info byte (fpsc.c:214-217, 806-809).  The archive API always writes (4,10)
  // for floats and (20,20) for doubles, which the throughput kernels are built for; any other shape the reference can
  // write (even exponents up to 30) goes through the reference-order kernel with tables sized from the header.
  uint32_t e1max = 0, e2max = 0;
  bool standard = true;
  for (int c = 0; c < arity; ++c)
    {
    uint8_t hi = 0;
    if (trico_hip_pointer_is_device(payloads[c]))
      {
      TRICO_HIP_TRY(hipMemcpyAsync(ctx->h_pinned, d_pay[c], 1, hipMemcpyDeviceToHost, current_stream()));
      TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
      hi = *(const uint8_t*)ctx->h_pinned;
      }
    else
      hi = payloads[c][0];
    const uint32_t e1 = (uint32_t)(hi >> 4) << 1, e2 = (uint32_t)(hi & 15u) << 1;
    standard = standard && (width == 4 ? (e1 == 4u && e2 == 10u) : (e1 == 20u && e2 == 20u));
    e1max = e1 > e1max ? e1 : e1max;
    e2max = e2 > e2max ? e2 : e2max;
    }
  if (e1max > 30u || e2max > 30u)
    {
    set_error("trico_hip_fpc_decode: table exponents above 30");
    return 0;
    }
  const size_t table_stride = ((size_t)1 << e1max) + ((size_t)1 << e2max);     // entries per component
  const bool lds_tables = width == 4 && e1max <= 4u && e2max <= 10u;
  uint64_t* d_tables = nullptr;
  if (width == 8 || !lds_tables)
    {
    const size_t tb = (size_t)arity * table_stride * (size_t)width;
    // the exponents come from a byte of the payload: a stream may ask for up to 48 GiB of tables.  The reference would calloc them;
    // here a request beyond half of the free device memory is refused instead of being tried
    if (!standard && tb > (size_t)(trico_hip_device_free_bytes() / 2))
      {
      set_error("trico_hip_fpc_decode: the stream's table exponents need more device memory than is free");
      return 0;
      }
    if (!standard || force_serial_stage(2))          // (the chain decoders bring their own scratch: fpc_chain_decode)
      {
      if (!ctx->tmp.reserve(tb + 256))
        return 0;
      TRICO_HIP_TRY(hipMemsetAsync(ctx->tmp.p, 0, tb, current_stream()));
      d_tables = (uint64_t*)ctx->tmp.p;
      }
    }
  ctx->chk_active = false;
  if (standard && !force_serial_stage(2))
    {
    // the chain decoders: checked by coding what they decoded (fpc_chain_decode), repeated by decode_complete if that fails
    ctx->chk_active = true;
    for (int c = 0; c < 3; ++c)
      {
      ctx->chk_pay[c] = c < arity ? d_pay[c] : nullptr;
      ctx->chk_sizes[c] = c < arity ? sizes[c] : 0u;
      }
    ctx->chk_arity = arity;
    ctx->chk_width = width;
    ctx->chk_n = n;
    ctx->chk_dst = d_dst;
    if (!fpc_chain_decode(ctx, g_first_attempt))
      return 0;
    }
  else if (!launch_fpc_decode_serial(d_pay, sizes, arity, width, n, d_dst, d_tables, table_stride, d_status))
    return 0;
  }
  TRICO_HIP_TRY(hipMemcpyAsync(ctx->h_pinned, d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, current_stream()));
  return 1;
  }

// waits for the decode launched on current_stream() and checks its status word
static int decode_complete(trico_hip_ctx* ctx, const char* what)
  {
  TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
  uint32_t st = ctx->h_pinned[0];
  for (int attempt = g_first_attempt + 1; ctx->chk_active && st != 0 && (st & ~(CHECK_BITS | FPC_STATUS_TIMEOUT)) == 0 && attempt <= 3; ++attempt)
    {
    // An archive that has already shown a stream of another writer (below) will show more: its streams fail the check every time,
    // by construction, not because a chain lost its tables - they go straight to the last rung instead of through two more chain
    // decodes each.
    if (ctx->other_writer_seen && attempt < 3 && !(st & FPC_STATUS_TIMEOUT))
      attempt = 3;
    // the payload parsed but the values do not code back to it: the chain went wrong (see fpc_chain_decode).  Twice more, then
    // in reference order without the scalar cache (20x slower).
    g_repeats += 1;
    if (getenv("TRICO_HIP_DEBUG"))
      fprintf(stderr, "trico_hip: decode self-check failed (status %#x, %d x %u values of %d bytes), attempt %d\n", st, ctx->chk_arity, ctx->chk_n,
              ctx->chk_width, attempt + 1);
    uint32_t* d_status = (uint32_t*)ctx->aux.p;
    TRICO_HIP_TRY(hipMemsetAsync(d_status, 0, 64, current_stream()));
    if (attempt <= 2)
      {
      if (!fpc_chain_decode(ctx, attempt))
        return 0;
      }
    else
      {
      const int arity = ctx->chk_arity, width = ctx->chk_width;
      uint64_t* d_tables = nullptr;
      size_t table_stride = 16 + 1024;
      if (width == 8)
        {
        table_stride = (size_t)2 << 20;
        const size_t tb = (size_t)arity * table_stride * 8;
        if (!ctx->tmp.reserve(tb))
          return 0;
        TRICO_HIP_TRY(hipMemsetAsync(ctx->tmp.p, 0, tb, current_stream()));
        d_tables = (uint64_t*)ctx->tmp.p;
        }
      // (this rung is checked like the others; see below for what a mismatch means here)
      if (!launch_fpc_decode_serial(ctx->chk_pay, ctx->chk_sizes, arity, width, ctx->chk_n, ctx->chk_dst, d_tables, table_stride, d_status) ||
          !fpc_check_launch(ctx, d_status))
        return 0;
      }
    TRICO_HIP_TRY(hipMemcpyAsync(ctx->h_pinned, d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, current_stream()));
    TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
    st = ctx->h_pinned[0];
    if (attempt == 3 && st != 0 && (st & ~CHECK_BITS) == 0)
      {
      // The reference-order kernel keeps no state in the scalar cache: what it decoded IS what the payload says (its loop is the
      // reference's, fpsc.c:308-326).  If that does not code back to the payload either, the payload was not written by the
      // reference's encoder (another writer may choose other, equally decodable codes): the values stand, and word 3 of
      // trico_hip_last_stats counts the stream so that a caller can see it.
      g_other_writer += 1;
      ctx->other_writer_seen = true;
      ctx->other_writer_streams += 1;
      if (strict_reader())
        {
        // trico_hip_set_strict(1) / TRICO_HIP_STRICT=1: a caller that only ever reads archives of the reference's own writer (or this
        // library's) wants to hear about it instead of getting values
        ctx->chk_active = false;
        set_error("trico decode: the payload decodes, but the reference's encoder would not have written it (the values, decoded in "
                  "reference order, do not code back to it); refused because the reader is strict (trico_hip_set_strict / TRICO_HIP_STRICT=1)");
        return 0;
        }
      st = 0;
      }
    }
  ctx->chk_active = false;
  if (st != 0)
    {
    set_error((st & (CHECK_BITS | FPC_STATUS_TIMEOUT)) && (st & ~(CHECK_BITS | FPC_STATUS_TIMEOUT)) == 0
                ? "trico decode: the decoded values do not code back to the payload" : what);
    return 0;
    }
  return 1;
  }

int trico_hip_fpc_decode(trico_hip_ctx* ctx, const uint8_t* const payloads[3], const uint32_t sizes[3],
                         int arity, int width, uint32_t n, void* dst)
  {
  if (!ctx || !device_ready())
    return 0;
  if (arity < 1 || arity > 3 || (width != 4 && width != 8))
    {
    set_error("trico_hip_fpc_decode: bad arguments");
    return 0;
    }
  if (!dst)
    return 1;
  const bool dst_dev = trico_hip_pointer_is_device(dst) != 0;
  if (!fpc_decode_launch(ctx, payloads, sizes, arity, width, n, dst_dev ? dst : nullptr) ||
      !decode_complete(ctx, "trico_hip_fpc_decode: malformed payload"))
    return 0;
  const size_t out_bytes = (size_t)n * arity * width;
  if (!dst_dev && out_bytes)
    {
    if (!download_bytes(dst, ctx->out.p, out_bytes, current_stream(), true))
      return 0;
    }
  return 1;
  }

// ---- integers -----------------------------------------------------------------------------------

int trico_hip_int_encode(trico_hip_ctx* ctx, const void* src, uint32_t count, int width, uint32_t sizes[8])
  {
  if (!ctx || !device_ready())
    return 0;
  if ((width != 1 && width != 2 && width != 4 && width != 8) || (count != 0 && !src) || count > 0x7E000000u)
    {
    set_error("trico_hip_int_encode: bad arguments");
    return 0;
    }
  const size_t in_bytes = (size_t)count * width;
  const size_t stride = align_up(lz4_bound(count), 256);
  const size_t plane_stride = align_up((size_t)count + 16, 256);
  if (!ctx->in.reserve(in_bytes + 16) || !ctx->out.reserve(stride * width) || !ctx->tmp.reserve(plane_stride * width))
    return 0;
  ctx->out_stride = stride;
  ctx->out_count = 0;
  ctx->out_in_slots = false;
  const void* d_src = count ? stage_in(ctx->in, src, in_bytes) : ctx->in.p;
  if (!d_src)
    return 0;
  const uint8_t* d_planes = (const uint8_t*)d_src;
  if (width > 1)
    {
    ProfSpan span(TRICO_HIP_K_PLANES_SPLIT);
    if (!launch_planes_split(d_src, count, width, ctx->tmp.p, plane_stride))
      return 0;
    d_planes = ctx->tmp.p;
    }
  uint32_t* d_sizes = (uint32_t*)ctx->aux.p;
  uint32_t* d_status = d_sizes + 16;
  TRICO_HIP_TRY(hipMemsetAsync(d_status, 0, 16, current_stream()));
  {
  ProfSpan span(TRICO_HIP_K_LZ4_ENCODE);
  if (force_serial_stage(4))
    {
    if (!launch_lz4_encode_serial(d_planes, plane_stride, count, width, ctx->out.p, stride, d_sizes))
      return 0;
    }
  else if (count >= lz4_chunked_threshold())
    {
    const size_t ws = lz4_chunked_workspace(count, width, plane_stride);
    if (!ctx->ws.reserve(ws))
      return 0;
    if (!launch_lz4_encode_chunked(d_planes, plane_stride, count, width, ctx->out.p, stride, d_sizes, ctx->ws.p, ctx->ws.cap, d_status))
      return 0;
    }
  else if (!launch_lz4_encode_wave(d_planes, plane_stride, count, width, ctx->out.p, stride, d_sizes))
    return 0;
  }
  if (!read_back_words(ctx, d_sizes, 20, ctx->out_sizes_raw))
    return 0;
  if (ctx->out_sizes_raw[16] != 0)
    {
    set_error("trico_hip_int_encode: internal LZ4 stitch error");
    return 0;
    }
  for (int c = 0; c < width; ++c)
    ctx->out_sizes[c] = ctx->out_sizes_raw[c];
  g_stats[0] = ctx->out_sizes_raw[17];      // chunks accepted by the LZ4 stitch pass (beyond chunk 0)
  g_stats[1] = ctx->out_sizes_raw[18];      // of those, re-parsed because the speculation was not provably equivalent
  ctx->out_count = width;
  for (int c = 0; c < width; ++c)
    sizes[c] = ctx->out_sizes[c];
  return 1;
  }

static int int_decode_launch(trico_hip_ctx* ctx, const uint8_t* const payloads[8], const uint32_t sizes[8],
                             int width, uint32_t count, void* d_dst)
  {
  ctx->chk_active = false;
  if (count > 0x7E000000u)
    {
    // LZ4_MAX_INPUT_SIZE (lz4.h:170): the reference cannot have written a larger block, and the data-parallel decoder tags its
    // source words in bit 31
    set_error("trico_hip_int_decode: plane larger than LZ4_MAX_INPUT_SIZE");
    return 0;
    }
  size_t total = 0, offs[8];
  bool any_host = false;
  for (int c = 0; c < width; ++c)
    {
    if (sizes[c] < 1 || !payloads[c])
      {
      set_error("trico_hip_int_decode: empty payload");
      return 0;
      }
    offs[c] = total;
    total += align_up((size_t)sizes[c] + 16, 256);
    any_host |= !trico_hip_pointer_is_device(payloads[c]);
    }
  if (any_host && !ctx->in.reserve(total))
    return 0;
  const uint8_t* d_pay[8];
  for (int c = 0; c < width; ++c)
    {
    d_pay[c] = (const uint8_t*)stage_in(ctx->in, payloads[c], sizes[c], offs[c]);
    if (!d_pay[c])
      return 0;
    }
  const size_t out_bytes = (size_t)count * width;
  if (!d_dst)
    {
    if (!ctx->out.reserve(out_bytes + 16))
      return 0;
    ctx->out_count = 0;
    d_dst = ctx->out.p;
    }
  uint8_t* d_planes = (uint8_t*)d_dst;
  const size_t plane_stride = align_up((size_t)count + 16, 256);
  if (width > 1)
    {
    if (!ctx->tmp.reserve(plane_stride * width))
      return 0;
    d_planes = ctx->tmp.p;
    }
  uint32_t* d_status = (uint32_t*)ctx->aux.p;
  TRICO_HIP_TRY(hipMemsetAsync(d_status, 0, 64, current_stream()));
  {
  ProfSpan span(TRICO_HIP_K_LZ4_DECODE);
  if (force_serial_stage(8))
    {
    if (!launch_lz4_decode_serial(d_pay, sizes, width, d_planes, plane_stride, count, d_status))
      return 0;
    }
  else if (count >= lz4_pdecode_threshold() && !getenv("TRICO_LZ4_DECODE_LDS"))
    {
    const size_t ws = lz4_pdecode_workspace(count, sizes, width);
    if (!ctx->ws.reserve(ws))
      return 0;
    if (!launch_lz4_decode_parallel(d_pay, sizes, width, d_planes, plane_stride, count, d_status, ctx->ws.p, ctx->ws.cap))
      return 0;
    }
  else if (!launch_lz4_decode_lds(d_pay, sizes, width, d_planes, plane_stride, count, d_status))
    return 0;
  }
  if (width > 1)
    {
    ProfSpan span(TRICO_HIP_K_PLANES_MERGE);
    if (!launch_planes_merge(d_planes, plane_stride, count, width, d_dst))
      return 0;
    }
  TRICO_HIP_TRY(hipMemcpyAsync(ctx->h_pinned, d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, current_stream()));
  return 1;
  }

int trico_hip_int_decode(trico_hip_ctx* ctx, const uint8_t* const payloads[8], const uint32_t sizes[8],
                         int width, uint32_t count, void* dst)
  {
  if (!ctx || !device_ready())
    return 0;
  if (width != 1 && width != 2 && width != 4 && width != 8)
    {
    set_error("trico_hip_int_decode: bad arguments");
    return 0;
    }
  if (!dst)
    return 1;
  const bool dst_dev = trico_hip_pointer_is_device(dst) != 0;
  if (!int_decode_launch(ctx, payloads, sizes, width, count, dst_dev ? dst : nullptr) ||
      !decode_complete(ctx, "trico_hip_int_decode: malformed LZ4 block"))
    return 0;
  const size_t out_bytes = (size_t)count * width;
  if (!dst_dev && out_bytes)
    {
    if (!download_bytes(dst, ctx->out.p, out_bytes, current_stream(), true))
      return 0;
    }
  return 1;
  }

int trico_hip_lz4_decoded_size(trico_hip_ctx* ctx, const void* payload, uint32_t size, uint32_t capacity, uint32_t* out_size)
  {
  if (!ctx || !device_ready() || !payload || !out_size || size == 0 || capacity > 0x7E000000u)
    {
    if (ctx)
      set_error("trico_hip_lz4_decoded_size: bad arguments");
    return 0;
    }
  if (!trico_hip_pointer_is_device(payload) && !ctx->in.reserve((size_t)size + 16))
    return 0;
  const uint8_t* d_pay = (const uint8_t*)stage_in(ctx->in, payload, size);
  if (!d_pay)
    return 0;
  uint32_t* d_out = (uint32_t*)ctx->aux.p;
  uint32_t two[2] = { 0u, 1u };
  if (!launch_lz4_measure(d_pay, size, capacity, d_out) || !read_back_words(ctx, d_out, 2, two))
    return 0;
  if (two[1])
    {
    set_error("trico_hip_lz4_decoded_size: malformed LZ4 block, or one that decodes to more than the capacity");
    return 0;
    }
  *out_size = two[0];
  return 1;
  }

// ---- stand-alone transposes (the reference's transpose_aos_to_soa.h) ------------------------------
// The coders above fuse the transposes; these exist for callers of the reference's low-level API.

int trico_hip_split_components(trico_hip_ctx* ctx, const void* aos, uint32_t n, int arity, int width, void* const* comps)
  {
  if (!ctx || !device_ready() || arity < 1 || arity > 8 || (width != 1 && width != 4 && width != 8) || (n && !aos))
    return 0;
  // width 1: `arity` byte planes of n arity-byte integers; width 4 / 8: components of interleaved reals
  const size_t elem = width == 1 ? (size_t)arity : (size_t)arity * width;
  const size_t comp_bytes = (size_t)n * (width == 1 ? 1 : width);
  const size_t stride = align_up(comp_bytes + 16, 256);
  if (!ctx->in.reserve((size_t)n * elem + 16) || !ctx->tmp.reserve(stride * arity))
    return 0;
  if (n == 0)
    return 1;
  const void* d_src = stage_in(ctx->in, aos, (size_t)n * elem);
  if (!d_src)
    return 0;
  const int ok = width == 1 ? launch_planes_split(d_src, n, arity, ctx->tmp.p, stride)
                            : launch_deinterleave(d_src, n, arity, width, ctx->tmp.p, stride);
  if (!ok)
    return 0;
  for (int c = 0; c < arity; ++c)
    TRICO_HIP_TRY(hipMemcpyAsync(comps[c], ctx->tmp.p + (size_t)c * stride, comp_bytes, hipMemcpyDefault, current_stream()));
  TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
  return 1;
  }

int trico_hip_merge_components(trico_hip_ctx* ctx, const void* const* comps, uint32_t n, int arity, int width, void* aos)
  {
  if (!ctx || !device_ready() || arity < 1 || arity > 8 || (width != 1 && width != 4 && width != 8) || (n && !aos))
    return 0;
  const size_t elem = width == 1 ? (size_t)arity : (size_t)arity * width;
  const size_t comp_bytes = (size_t)n * (width == 1 ? 1 : width);
  const size_t stride = align_up(comp_bytes + 16, 256);
  if (!ctx->tmp.reserve(stride * arity) || !ctx->out.reserve((size_t)n * elem + 16))
    return 0;
  ctx->out_count = 0;
  if (n == 0)
    return 1;
  for (int c = 0; c < arity; ++c)
    TRICO_HIP_TRY(hipMemcpyAsync(ctx->tmp.p + (size_t)c * stride, comps[c], comp_bytes, hipMemcpyDefault, current_stream()));
  const bool dst_dev = trico_hip_pointer_is_device(aos) != 0;
  void* d_dst = dst_dev ? aos : (void*)ctx->out.p;
  const int ok = width == 1 ? launch_planes_merge(ctx->tmp.p, stride, n, arity, d_dst)
                            : launch_interleave(ctx->tmp.p, stride, n, arity, width, d_dst);
  if (!ok)
    return 0;
  if (!dst_dev)
    TRICO_HIP_TRY(hipMemcpyAsync(aos, d_dst, (size_t)n * elem, hipMemcpyDeviceToHost, current_stream()));
  TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
  return 1;
  }

// ---- vertex welding for the STL reader ---------------------------------------------------------------
// corners: 3 * ntri positions (xyz floats, host or device).  On success returns 1: *nr_of_vertices unique positions in
// vertices (capacity 3 * ntri positions) in lexicographic order and 3 * ntri indices in triangles.  Returns 2 when the
// input holds -0.0 or NaN (the tie rule of the reference's quicksort then matters; the caller welds on the host), 0 on
// error.
int trico_hip_weld_vertices(trico_hip_ctx* ctx, const float* corners, uint32_t ntri, float* vertices, uint32_t* triangles,
                            uint32_t* nr_of_vertices)
  {
  if (!ctx || !device_ready() || !corners || !vertices || !triangles || !nr_of_vertices || ntri == 0 || ntri > 0x2aaaaaaau)
    return 0;
  const uint32_t n = 3u * ntri;
  const size_t pos_bytes = (size_t)n * 12;
  const size_t ws = weld_workspace(n);
  if (!ctx->in.reserve(pos_bytes + 16) || !ctx->out.reserve(pos_bytes + (size_t)n * 4 + 512) || !ctx->ws.reserve(ws))
    return 0;
  ctx->out_count = 0;
  const void* d_pos = stage_in(ctx->in, corners, pos_bytes);
  if (!d_pos)
    return 0;
  uint32_t* d_out_pos = (uint32_t*)ctx->out.p;
  uint32_t* d_out_tri = (uint32_t*)(ctx->out.p + align_up(pos_bytes, 256));
  uint32_t* d_result = (uint32_t*)ctx->aux.p;
  if (!launch_weld((const uint32_t*)d_pos, n, d_out_pos, d_out_tri, ctx->ws.p, ctx->ws.cap, d_result))
    return 0;
  uint32_t res[2] = { 0, 0 };
  if (!read_back_words(ctx, d_result, 2, res))
    return 0;
  if (res[1] != 0)
    return 2;
  *nr_of_vertices = res[0];
  TRICO_HIP_TRY(hipMemcpyAsync(vertices, d_out_pos, (size_t)res[0] * 12, hipMemcpyDefault, current_stream()));
  TRICO_HIP_TRY(hipMemcpyAsync(triangles, d_out_tri, (size_t)n * 4, hipMemcpyDefault, current_stream()));
  TRICO_HIP_TRY(hipStreamSynchronize(current_stream()));
  return 1;
  }

// float payloads of the throughput encoder live in segment slots: gather component c into ctx->out
static int materialize(trico_hip_ctx* ctx, int c)
  {
  if (!ctx->out_in_slots || ctx->out_materialized[c])
    return 1;
  ProfSpan span(TRICO_HIP_K_FPC32_ENCODE, false);
  if (!launch_fpc32_gather(ctx->slots_n, ctx->slots_arity, c, ctx->tmp.p, ctx->out.p + (size_t)c * ctx->out_stride))
    return 0;
  ctx->out_materialized[c] = true;
  return 1;
  }

int trico_hip_fetch_payload(trico_hip_ctx* ctx, int c, void* dst)
  {
  if (!ctx || c < 0 || c >= ctx->out_count)
    {
    set_error("trico_hip_fetch_payload: no such payload");
    return 0;
    }
  if (ctx->out_in_slots && !ctx->out_materialized[c] && trico_hip_pointer_is_device(dst))
    {
    // straight from the segment slots into the destination (e.g. the device-resident archive buffer)
    ProfSpan span(TRICO_HIP_K_FPC32_ENCODE, false);
    if (!launch_fpc32_gather(ctx->slots_n, ctx->slots_arity, c, ctx->tmp.p, (uint8_t*)dst))
      return 0;
    return 1;
    }
  if (!materialize(ctx, c))
    return 0;
  return trico_hip_copy(dst, ctx->out.p + (size_t)c * ctx->out_stride, ctx->out_sizes[c]);
  }

int trico_hip_fetch_payloads(trico_hip_ctx* ctx, int count, void* const* dsts)
  {
  if (!ctx || count != ctx->out_count || count < 1 || count > 8)
    {
    set_error("trico_hip_fetch_payloads: payload count mismatch");
    return 0;
    }
  bool fused = ctx->out_in_slots && count <= 3;
  for (int c = 0; c < count && fused; ++c)
    fused = !ctx->out_materialized[c] && trico_hip_pointer_is_device(dsts[c]);
  if (fused)
    {
    // float payloads still in their segment slots, device destinations: one gather launch for all components
    uint8_t* d[3] = { (uint8_t*)dsts[0], count > 1 ? (uint8_t*)dsts[1] : nullptr, count > 2 ? (uint8_t*)dsts[2] : nullptr };
    ProfSpan span(TRICO_HIP_K_FPC32_ENCODE, false);
    return launch_fpc32_gather_all(ctx->slots_n, ctx->slots_arity, ctx->tmp.p, d);
    }
  for (int c = 0; c < count; ++c)
    if (!trico_hip_fetch_payload(ctx, c, dsts[c]))
      return 0;
  return 1;
  }

const uint8_t* trico_hip_payload_device_pointer(trico_hip_ctx* ctx, int c)
  {
  if (!ctx || c < 0 || c >= ctx->out_count)
    return nullptr;
  if (!materialize(ctx, c))
    return nullptr;
  return ctx->out.p + (size_t)c * ctx->out_stride;
  }

// ---- profiling ----------------------------------------------------------------------------------

int trico_hip_fpc32_code_sweep(void)
  {
  if (!device_ready())
    return 0;
  return fpc32_code_sweep_mode();
  }

void trico_hip_last_stats(uint32_t out[4])
  {
  out[0] = g_stats[0];
  out[1] = g_stats[1];
  out[2] = g_repeats.load();
  out[3] = g_other_writer.load();
  }

void trico_hip_encode_stats(uint32_t out[2])
  {
  out[0] = g_recoded_order.load();
  out[1] = g_recoded_sentinel.load();
  }

uint32_t trico_hip_encode_scan_recodes(void) { return g_recoded_scan.load(); }

void trico_hip_set_strict(int on) { g_strict.store(on < 0 ? -1 : on != 0); }
uint32_t trico_hip_ctx_other_writer_streams(const trico_hip_ctx* ctx) { return ctx ? ctx->other_writer_streams : 0u; }
void trico_hip_set_encode_verify(int on) { g_verify.store(on < 0 ? -1 : on != 0); }
void trico_hip_encode_verify_stats(uint64_t out[3])
  {
  out[0] = g_verified_streams.load();
  out[1] = g_verified_values.load();
  out[2] = g_verify_mismatch.load();
  }

void trico_hip_profile_enable(int on) { g_prof_on = on != 0; }

void trico_hip_profile_reset(void)
  {
  prof_drain();
  for (int k = 0; k < TRICO_HIP_K_COUNT; ++k)
    {
    g_prof_ms[k] = 0.0;
    g_prof_spans[k] = 0;
    }
  }

double trico_hip_profile_ms(int k, uint64_t* spans)
  {
  if (k < 0 || k >= TRICO_HIP_K_COUNT)
    return 0.0;
  prof_drain();
  if (spans)
    *spans = g_prof_spans[k];
  return g_prof_ms[k];
  }

} // extern "C"
