// dist.hip — what the multi-GPU layer needs behind the C-ABI (include/trico/trico_hip.h, "stream-sharded encoding" and
// "RCCL exchange"): encoders for ONE unit of a stream (a component of a real stream, a byte plane of an integer stream),
// and a thin wrapper over RCCL for the one exchange step the format has: collecting variable-size byte strings on a root.
//
// The reference compresses the units of a stream one after the other on one core (trico.c:229-260, 346-368); they are
// independent, so the ranks of a job can each take some.  RCCL is loaded with dlopen when the first communicator call is
// made: a process that never exchanges anything does not pay for loading it, and libtrico.so does not link it.
#include "common.hpp"

#include <dlfcn.h>
#include <mutex>
#include <string.h>

using namespace trico;

// ---- unit encoders ---------------------------------------------------------------------------------------------------------
int trico_hip_fpc_encode_component(trico_hip_ctx* ctx, const void* src, uint32_t n, int arity, int width, int comp, uint32_t* size)
  {
  if (!ctx || !size || arity < 1 || arity > 3 || comp < 0 || comp >= arity || (width != 4 && width != 8) || (n && !src))
    {
    set_error("trico_hip_fpc_encode_component: bad arguments");
    return 0;
    }
  if (!trico_hip_available())
    return 0;
  uint32_t sizes[3] = { 0, 0, 0 };
  if (arity == 1)
    {
    if (!trico_hip_fpc_encode(ctx, src, n, 1, width, sizes))
      return 0;
    *size = sizes[0];
    return 1;
    }
  // split the interleaved array once (a pass at HBM speed), then code the wanted component as a scalar stream
  const size_t comp_bytes = (size_t)n * width, stride = align_up(comp_bytes + 16, 256);
  if (!ctx->unit.reserve(stride * arity + 16))
    return 0;
  const void* d_src = src;
  if (!trico_hip_pointer_is_device(src))
    {
    if (!ctx->in.reserve((size_t)n * arity * width + 16))
      return 0;
    if (!upload_bytes(ctx->in.p, src, (size_t)n * arity * width, current_stream()))
      return 0;
    d_src = ctx->in.p;
    }
  if (n && !launch_deinterleave(d_src, n, arity, width, ctx->unit.p, stride))
    return 0;
  if (!trico_hip_fpc_encode(ctx, ctx->unit.p + (size_t)comp * stride, n, 1, width, sizes))
    return 0;
  *size = sizes[0];
  return 1;
  }

int trico_hip_int_encode_plane(trico_hip_ctx* ctx, const void* src, uint32_t count, int width, int plane, uint32_t* size)
  {
  if (!ctx || !size || (width != 1 && width != 2 && width != 4 && width != 8) || plane < 0 || plane >= width || (count && !src))
    {
    set_error("trico_hip_int_encode_plane: bad arguments");
    return 0;
    }
  if (!trico_hip_available())
    return 0;
  uint32_t sizes[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  if (width == 1)
    {
    if (!trico_hip_int_encode(ctx, src, count, 1, sizes))
      return 0;
    *size = sizes[0];
    return 1;
    }
  const size_t stride = align_up((size_t)count + 16, 256);
  if (!ctx->unit.reserve(stride * width + 16))
    return 0;
  const void* d_src = src;
  if (!trico_hip_pointer_is_device(src))
    {
    if (!ctx->in.reserve((size_t)count * width + 16))
      return 0;
    if (!upload_bytes(ctx->in.p, src, (size_t)count * width, current_stream()))
      return 0;
    d_src = ctx->in.p;
    }
  if (count && !launch_planes_split(d_src, count, width, ctx->unit.p, stride))
    return 0;
  if (!trico_hip_int_encode(ctx, ctx->unit.p + (size_t)plane * stride, count, 1, sizes))
    return 0;
  *size = sizes[0];
  return 1;
  }

// ---- RCCL, loaded on demand ------------------------------------------------------------------------------------------------
namespace {

// the part of rccl.h this file uses (ABI of RCCL 2.x: ncclUniqueId is 128 opaque bytes passed by value)
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
enum { RCCL_UINT8 = 1, RCCL_UINT64 = 5 };

struct Rccl
  {
  int (*GetUniqueId)(rcclUniqueId*);
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int);
  int (*CommDestroy)(rcclComm_t);
  int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t);
  int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t);
  int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  const char* (*GetErrorString)(int);
  };

Rccl g_rccl;
int g_rccl_state = 0;   // 0 unknown, 1 loaded, -1 unavailable
std::mutex g_rccl_mutex;

bool rccl_ready()
  {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl_state == 0)
    {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h)
      h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    bool ok = h != nullptr;
    auto sym = [&](const char* name) -> void* { void* p = ok ? dlsym(h, name) : nullptr; ok = ok && p; return p; };
    g_rccl.GetUniqueId = (int (*)(rcclUniqueId*))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(rcclComm_t*, int, rcclUniqueId, int))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(rcclComm_t))sym("ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, rcclComm_t, hipStream_t))sym("ncclAllGather");
    g_rccl.Send = (int (*)(const void*, size_t, int, int, rcclComm_t, hipStream_t))sym("ncclSend");
    g_rccl.Recv = (int (*)(void*, size_t, int, int, rcclComm_t, hipStream_t))sym("ncclRecv");
    g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
    g_rccl.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    // the declarations above are the ABI of RCCL / NCCL 2.x (128-byte unique id by value, these enum values): refuse anything else
    int (*get_version)(int*) = (int (*)(int*))sym("ncclGetVersion");
    int version = 0;
    if (ok && (get_version(&version) != 0 || version < 20000 || version >= 30000))
      ok = false;
    g_rccl_state = ok ? 1 : -1;
    }
  if (g_rccl_state < 0)
    set_error("RCCL (librccl.so, version 2.x) could not be loaded");
  return g_rccl_state > 0;
  }

bool rccl_ok(int r, const char* what)
  {
  if (r == 0)
    return true;
  static thread_local char msg[256];
  snprintf(msg, sizeof(msg), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  set_error(msg);
  return false;
  }

} // namespace

#ifdef TRICO_HIP_TEST_HOOKS
// Test transport (libtrico_testhooks.so only): the ranks of a "job" are THREADS of one process on one GPU, and the four RCCL
// calls trico_hip_comm_gather makes are played by a rendezvous in host memory plus hipMemcpy between the ranks' device buffers.
// It exists so that the offset arithmetic and the call pattern of the gather (sizes, verdict of the root, one transfer per
// non-empty rank, empty ranks, a root that is not rank 0) can be tested with more than one rank on a box with one GPU.
#include <condition_variable>
namespace {
struct FakeGroup
  {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, arrived = 0, generation = 0;
  const void* gather_src[64];
  struct Msg { const void* src; size_t bytes; bool posted; } send[64];      // one outstanding send per source rank
  };
FakeGroup g_fake;
struct FakeComm { int rank; };

void fake_barrier(std::unique_lock<std::mutex>& lk)
  {
  const int gen = g_fake.generation;
  if (++g_fake.arrived == g_fake.world)
    {
    g_fake.arrived = 0;
    ++g_fake.generation;
    g_fake.cv.notify_all();
    }
  else
    g_fake.cv.wait(lk, [&] { return g_fake.generation != gen; });
  }

int fake_all_gather(const void* send, void* recv, size_t count, int type, rcclComm_t comm, hipStream_t st)
  {
  const size_t esz = type == RCCL_UINT64 ? 8 : 1;
  FakeComm* fc = (FakeComm*)comm;
  (void)hipStreamSynchronize(st);
  std::unique_lock<std::mutex> lk(g_fake.mu);
  g_fake.gather_src[fc->rank] = send;
  fake_barrier(lk);
  for (int r = 0; r < g_fake.world; ++r)
    if (hipMemcpy((uint8_t*)recv + (size_t)r * count * esz, g_fake.gather_src[r], count * esz, hipMemcpyDeviceToDevice) != hipSuccess)
      return 1;
  fake_barrier(lk);                                   // nobody changes its send buffer before everybody has read it
  return 0;
  }
int fake_send(const void* src, size_t count, int, int, rcclComm_t comm, hipStream_t st)
  {
  FakeComm* fc = (FakeComm*)comm;
  (void)hipStreamSynchronize(st);
  std::unique_lock<std::mutex> lk(g_fake.mu);
  g_fake.send[fc->rank] = FakeGroup::Msg{ src, count, true };
  g_fake.cv.notify_all();
  g_fake.cv.wait(lk, [&] { return !g_fake.send[fc->rank].posted; });       // until the root has taken it
  return 0;
  }
int fake_recv(void* dst, size_t count, int, int peer, rcclComm_t, hipStream_t st)
  {
  (void)hipStreamSynchronize(st);
  std::unique_lock<std::mutex> lk(g_fake.mu);
  g_fake.cv.wait(lk, [&] { return g_fake.send[peer].posted; });
  const FakeGroup::Msg m = g_fake.send[peer];
  const int rc = (m.bytes == count && hipMemcpy(dst, m.src, count, hipMemcpyDeviceToDevice) == hipSuccess) ? 0 : 1;
  g_fake.send[peer].posted = false;
  g_fake.cv.notify_all();
  return rc;
  }
int fake_ok() { return 0; }
int fake_destroy(rcclComm_t comm) { delete (FakeComm*)comm; return 0; }
const char* fake_error(int) { return "fake transport error"; }
} // namespace
#endif

struct trico_hip_comm
  {
  rcclComm_t comm = nullptr;
  int rank = 0, world = 1;
  uint64_t* d_sizes = nullptr;     // world + 1 words: [0] mine, [1..] everybody's
  Rccl tr;                         // the transport of this communicator: RCCL, or the test transport
  };

#ifdef TRICO_HIP_TEST_HOOKS
extern "C" TRICO_API trico_hip_comm* trico_hip_comm_create_fake(int rank, int world)
  {
  if (world < 1 || world > 64 || rank < 0 || rank >= world || !trico_hip_available())
    return nullptr;
  {
  std::lock_guard<std::mutex> lk(g_fake.mu);
  g_fake.world = world;
  g_fake.send[rank].posted = false;
  }
  trico_hip_comm* c = new trico_hip_comm;
  c->rank = rank;
  c->world = world;
  c->comm = new FakeComm{ rank };
  c->tr = Rccl{ nullptr, nullptr, fake_destroy, fake_all_gather, fake_send, fake_recv, fake_ok, fake_ok, fake_error };
  if (!hip_ok(hipMalloc((void**)&c->d_sizes, sizeof(uint64_t) * (size_t)(world + 1)), "hipMalloc(comm sizes)"))
    {
    delete (FakeComm*)c->comm;
    delete c;
    return nullptr;
    }
  return c;
  }
#endif

int trico_hip_comm_unique_id(uint8_t id[128])
  {
  if (!id || !trico_hip_available() || !rccl_ready())
    return 0;
  rcclUniqueId u;
  if (!rccl_ok(g_rccl.GetUniqueId(&u), "ncclGetUniqueId"))
    return 0;
  memcpy(id, u.internal, 128);
  return 1;
  }

trico_hip_comm* trico_hip_comm_create(const uint8_t id[128], int rank, int world)
  {
  if (!id || world < 1 || rank < 0 || rank >= world || !trico_hip_available() || !rccl_ready())
    return nullptr;
  trico_hip_comm* c = new trico_hip_comm;
  c->rank = rank;
  c->world = world;
  c->tr = g_rccl;
  rcclUniqueId u;
  memcpy(u.internal, id, 128);
  if (!rccl_ok(g_rccl.CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank") ||
      !hip_ok(hipMalloc((void**)&c->d_sizes, sizeof(uint64_t) * (size_t)(world + 1)), "hipMalloc(comm sizes)"))
    {
    if (c->comm)
      (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return nullptr;
    }
  return c;
  }

void trico_hip_comm_destroy(trico_hip_comm* c)
  {
  if (!c)
    return;
  (void)hipStreamSynchronize(current_stream());
  if (c->comm)
    (void)c->tr.CommDestroy(c->comm);
  if (c->d_sizes)
    (void)hipFree(c->d_sizes);
  delete c;
  }

int trico_hip_comm_gather(trico_hip_comm* c, const void* d_local, uint64_t local_bytes, int root, void* d_root, uint64_t root_capacity,
                          uint64_t* sizes)
  {
  if (!c || !sizes || root < 0 || root >= c->world || (local_bytes && !d_local))
    {
    set_error("trico_hip_comm_gather: bad arguments");
    return 0;
    }
  hipStream_t st = current_stream();
  // 1. everybody learns everybody's size (one all-gather of the compressed sizes)
  TRICO_HIP_TRY(hipMemcpyAsync(c->d_sizes, &local_bytes, sizeof(uint64_t), hipMemcpyHostToDevice, st));
  if (!rccl_ok(c->tr.AllGather(c->d_sizes, c->d_sizes + 1, 1, RCCL_UINT64, c->comm, st), "ncclAllGather(sizes)"))
    return 0;
  TRICO_HIP_TRY(hipMemcpyAsync(sizes, c->d_sizes + 1, sizeof(uint64_t) * (size_t)c->world, hipMemcpyDeviceToHost, st));
  TRICO_HIP_TRY(hipStreamSynchronize(st));
  uint64_t total = 0;
  for (int r = 0; r < c->world; ++r)
    total += sizes[r];
  if (c->rank == root && (total > root_capacity || (total && !d_root)))
    set_error("trico_hip_comm_gather: root buffer too small");
  // every rank can evaluate the same condition only if it knows the capacity: the root's verdict travels as a second tiny
  // all-gather, so that nobody posts a transfer the root will not match
  uint64_t verdict = (c->rank == root && (total > root_capacity || (total && !d_root))) ? 1u : 0u;
  TRICO_HIP_TRY(hipMemcpyAsync(c->d_sizes, &verdict, sizeof(uint64_t), hipMemcpyHostToDevice, st));
  if (!rccl_ok(c->tr.AllGather(c->d_sizes, c->d_sizes + 1, 1, RCCL_UINT64, c->comm, st), "ncclAllGather(verdict)"))
    return 0;
  uint64_t root_verdict = 0;
  TRICO_HIP_TRY(hipMemcpyAsync(&root_verdict, c->d_sizes + 1 + root, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  TRICO_HIP_TRY(hipStreamSynchronize(st));
  if (root_verdict)
    return 0;
  // 2. payloads straight to their final offsets (point-to-point: over xGMI every sender has its own link to the root)
  if (c->rank == root)
    {
    uint64_t off = 0;
    if (!rccl_ok(c->tr.GroupStart(), "ncclGroupStart"))
      return 0;
    bool ok = true;
    for (int r = 0; r < c->world; ++r)
      {
      if (r != root && sizes[r])
        ok = ok && rccl_ok(c->tr.Recv((uint8_t*)d_root + off, (size_t)sizes[r], RCCL_UINT8, r, c->comm, st), "ncclRecv");
      off += sizes[r];
      }
    if (!rccl_ok(c->tr.GroupEnd(), "ncclGroupEnd") || !ok)
      return 0;
    uint64_t mine = 0;
    for (int r = 0; r < root; ++r)
      mine += sizes[r];
    if (local_bytes)
      TRICO_HIP_TRY(hipMemcpyAsync((uint8_t*)d_root + mine, d_local, (size_t)local_bytes, hipMemcpyDeviceToDevice, st));
    }
  else if (local_bytes)
    {
    if (!rccl_ok(c->tr.Send(d_local, (size_t)local_bytes, RCCL_UINT8, root, c->comm, st), "ncclSend"))
      return 0;
    }
  TRICO_HIP_TRY(hipStreamSynchronize(st));
  return 1;
  }
