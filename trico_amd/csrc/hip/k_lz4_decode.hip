// k_lz4_decode.hip — LZ4 block decompressor, one 1024-thread workgroup per byte plane.
//
// Replaces LZ4_decompress_safe (lz4.c:2078-2083 -> LZ4_decompress_generic lz4.c:1657-2072) as called per
// plane by the readers (trico.c:1100-1129), with the same acceptance rules for well-formed blocks and a
// clean failure (status word) for malformed ones.
//
// A block is one chain of sequences: token -> literal run -> offset -> match, every match possibly
// reading what the previous sequence just wrote.  A lone wave issues an instruction every ~5-9 cycles and a
// global store needs ~1 us to become readable again, so the design keeps the chain out of global memory:
//   * the last 128 KiB of output live in an LDS ring (matches reach back at most 65535 bytes): short
//     match copies are LDS -> LDS (in-order, no waits) plus fire-and-forget global stores;
//   * the compressed bytes are staged through an 8 KiB LDS window: token, length and offset bytes and
//     short literal runs come from LDS;
//   * overlapping matches are produced directly as a periodic pattern (dst[k] = period[k % offset]);
//   * copies of 4 KiB and more go to all 16 waves (job + barrier): literals global -> global, matches as
//     a periodic fill from the LDS ring; both refresh the ring's last 64 KiB.
// Latency-bound on the chain for short sequences, HBM-bound on long runs.
#include "common.hpp"

namespace trico {

namespace {

constexpr int WG = 1024;
constexpr uint32_t ORING = 128u << 10;           // output ring bytes (power of two, > 2 * 65535)
constexpr uint32_t OMASK = ORING - 1u;
constexpr uint32_t CWIN = 8192;                  // compressed window bytes
constexpr uint32_t BULK = 4096;                  // copies from this size go to the whole workgroup

struct Lz4DecArgs
  {
  const uint8_t* pay[8];
  uint32_t size[8];
  };

enum { JOB_EXIT = 0, JOB_LIT = 1, JOB_MATCH = 2 };
struct Job
  {
  uint32_t kind, n, op, off;       // bytes, output position, match offset
  const uint8_t* src;              // literal source (global)
  };

__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) u128u { u32x4 v; };

// whole workgroup: n literal bytes from global src to dst[op..] (global) and into the ring (last 64 KiB)
__device__ __forceinline__ void wg_literals(uint8_t* __restrict__ dst, uint8_t* ring, const uint8_t* __restrict__ src, uint32_t op, uint32_t n, int tid)
  {
  const uint32_t keep = n > 65536u ? n - 65536u : 0u;            // only the tail has to stay in the ring
  // bulk part [0, keep): 16-byte copies, destination aligned
  if (keep)
    {
    uint8_t* d = dst + op;
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)d & 15u)) & 15u);
    const uint32_t h = head < keep ? head : keep;
    if ((uint32_t)tid < h)
      d[tid] = src[tid];
    uint32_t i = h + 16u * (uint32_t)tid;
    for (; i + 16u <= keep; i += WG * 16u)
      *(u32x4*)(d + i) = ((const u128u*)(src + i))->v;
    if (i < keep && keep - i < 16u)
      for (uint32_t k = i; k < keep; ++k)
        d[k] = src[k];
    }
  for (uint32_t k = keep + (uint32_t)tid; k < n; k += WG)
    {
    const uint8_t b = src[k];
    dst[op + k] = b;
    ring[(op + k) & OMASK] = b;
    }
  }

// whole workgroup: match of length n at offset off, overlapping or not: dst[op + k] = out[op - off + (k % off)]
// (the period [op - off, op) is complete in the ring).  The ring is updated for the last 64 KiB after a barrier
// because a long match may overwrite its own period in the ring.
__device__ __forceinline__ void wg_match(uint8_t* __restrict__ dst, uint8_t* ring, uint32_t op, uint32_t off, uint32_t n, int tid)
  {
  const uint32_t per = (uint32_t)tid * 16u;
  // pass 1: global only, source from the ring period
  for (uint32_t base = per; base < n; base += WG * 16u)
    {
    uint32_t r = base % off;
    const uint32_t e = base + 16u < n ? base + 16u : n;
    for (uint32_t k = base; k < e; ++k)
      {
      dst[op + k] = ring[(op - off + r) & OMASK];
      if (++r == off) r = 0;
      }
    }
  __syncthreads();
  // pass 2: refresh the ring tail from the same period definition, reading the period bytes first into registers is
  // not possible for long periods, so re-read them from global memory (they are at least one barrier old)
  const uint32_t keep = n > 65536u ? n - 65536u : 0u;
  __builtin_amdgcn_s_waitcnt(0);
  for (uint32_t k = keep + (uint32_t)tid; k < n; k += WG)
    ring[(op + k) & OMASK] = __hip_atomic_load(&dst[op + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

__device__ __forceinline__ void helper_loop(Job* job, uint8_t* __restrict__ dst, uint8_t* ring, int tid)
  {
  for (;;)
    {
    __syncthreads();
    const uint32_t kind = job->kind;
    if (kind == JOB_EXIT)
      return;
    if (kind == JOB_LIT)
      wg_literals(dst, ring, job->src, job->op, job->n, tid);
    else
      wg_match(dst, ring, job->op, job->off, job->n, tid);
    __syncthreads();
    }
  }

__global__ void __launch_bounds__(WG) k_lz4_decode_lds(Lz4DecArgs a, uint8_t* __restrict__ planes, size_t plane_stride, uint32_t cap,
                                                       uint32_t* __restrict__ status)
  {
  extern __shared__ uint8_t lds[];
  uint8_t* ring = lds;                       // ORING bytes
  uint8_t* cw = lds + ORING;                 // CWIN bytes
  __shared__ Job job;
  const int tid = threadIdx.x;
  const uint8_t* src = a.pay[blockIdx.x];
  const uint32_t n = a.size[blockIdx.x];
  uint8_t* dst = planes + (size_t)blockIdx.x * plane_stride;
  if (tid >= 64)
    {
    helper_loop(&job, dst, ring, tid);
    return;
    }
  const int lane = tid;
  uint32_t ip = 0, op = 0;
  uint32_t cw0 = 0;                          // stream position of cw[0]
  auto refill = [&](uint32_t from)
    {
    cw0 = from;
    for (uint32_t k = (uint32_t)lane; k < CWIN; k += 64u)
      cw[k] = (from + k < n) ? src[from + k] : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
  refill(0);
  bool bad = (n == 0);
  while (!bad)
    {
    if (ip + 64u > cw0 + CWIN)               // token + extension bytes + offset stay inside the window
      refill(ip);
    if (ip >= n) { bad = true; break; }
    const uint32_t tok = uni(cw[ip - cw0]);
    ++ip;
    uint32_t lit = tok >> 4;
    if (lit == 15u)
      {
      uint32_t bb;
      do
        {
        if (ip >= n) { bad = true; break; }
        if (ip >= cw0 + CWIN) refill(ip);
        bb = uni(cw[ip - cw0]);
        ++ip;
        lit += bb;
        }
      while (bb == 255u);
      if (bad) break;
      }
    if (lit > n - ip || lit > cap - op) { bad = true; break; }
    // ---- literals ----
    if (lit >= BULK)
      {
      if (lane == 0) { job.kind = JOB_LIT; job.n = lit; job.op = op; job.src = src + ip; }
      __syncthreads();
      wg_literals(dst, ring, src + ip, op, lit, tid);
      __syncthreads();
      }
    else if (lit)
      {
      if (ip + lit > cw0 + CWIN)
        refill(ip);
      for (uint32_t k = (uint32_t)lane; k < lit; k += 64u)
        {
        const uint8_t b = cw[ip - cw0 + k];
        ring[(op + k) & OMASK] = b;
        dst[op + k] = b;
        }
      }
    ip += lit; op += lit;
    if (ip == n) break;                                   // last sequence: literals only
    if (n - ip < 2u) { bad = true; break; }
    if (ip + 64u > cw0 + CWIN)
      refill(ip);
    const uint32_t off = uni((uint32_t)cw[ip - cw0] | ((uint32_t)cw[ip - cw0 + 1u] << 8));
    ip += 2;
    if (off == 0u || off > op) { bad = true; break; }
    uint32_t ml = tok & 15u;
    if (ml == 15u)
      {
      uint32_t bb;
      do
        {
        if (ip >= n) { bad = true; break; }
        if (ip >= cw0 + CWIN) refill(ip);
        bb = uni(cw[ip - cw0]);
        ++ip;
        ml += bb;
        }
      while (bb == 255u);
      if (bad) break;
      }
    ml += 4u;
    if (ml > cap - op) { bad = true; break; }
    // ---- match ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (ml >= BULK)
      {
      if (lane == 0) { job.kind = JOB_MATCH; job.n = ml; job.op = op; job.off = off; }
      __syncthreads();
      wg_match(dst, ring, op, off, ml, tid);
      __syncthreads();
      }
    else if (off >= ml)
      {
      // no overlap: plain copy through the ring, 64 bytes per iteration
      for (uint32_t k = (uint32_t)lane; k < ml; k += 64u)
        {
        const uint8_t b = ring[(op - off + k) & OMASK];
        ring[(op + k) & OMASK] = b;
        dst[op + k] = b;
        }
      }
    else
      {
      // overlap: periodic pattern from the period [op - off, op), which this match never overwrites in the ring
      // while k < 4096 + ... (ml < BULK <= ORING - 65535)
      for (uint32_t k = (uint32_t)lane; k < ml; k += 64u)
        {
        const uint8_t b = ring[(op - off + (k % off)) & OMASK];
        ring[(op + k) & OMASK] = b;
        dst[op + k] = b;
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    op += ml;
    }
  if ((bad || op != cap) && lane == 0)
    atomicOr(status, 8u);
  if (lane == 0)
    job.kind = JOB_EXIT;
  __syncthreads();
  }

} // namespace

int launch_lz4_decode_lds(const uint8_t* const d_payloads[8], const uint32_t sizes[8], int nplanes,
                          uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, uint32_t* d_status)
  {
  Lz4DecArgs a;
  for (int c = 0; c < 8; ++c)
    {
    a.pay[c] = c < nplanes ? d_payloads[c] : nullptr;
    a.size[c] = c < nplanes ? sizes[c] : 0;
    }
  static bool attr_set = false;
  if (!attr_set)
    {
    if (!hip_ok(hipFuncSetAttribute((const void*)k_lz4_decode_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ORING + CWIN)),
                "hipFuncSetAttribute(k_lz4_decode_lds)"))
      return 0;
    attr_set = true;
    }
  hipLaunchKernelGGL(k_lz4_decode_lds, dim3(nplanes), dim3(WG), ORING + CWIN, current_stream(),
                     a, d_planes, plane_stride, plane_bytes, d_status);
  return hip_ok(hipGetLastError(), "k_lz4_decode_lds") ? 1 : 0;
  }

} // namespace trico
