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
#include <atomic>
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

// inclusive prefix maximum over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_prefix_max(uint32_t v)
  {
  uint32_t t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); v = v > t ? v : t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); v = v > t ? v : t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); v = v > t ? v : t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); v = v > t ? v : t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;
  return v;
  }

// inclusive prefix sum over the 64 lanes of a wave (DPP row shifts + row broadcasts)
__device__ __forceinline__ uint32_t wave_prefix_add(uint32_t v)
  {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);                   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);                   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);                   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);                   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);                  // row_bcast:15 into rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);                  // row_bcast:31 into rows 2, 3
  return v;
  }

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
  __shared__ uint32_t own[64];                // short-sequence batches: owner (lane + 1) of each output byte of a pass
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
  uint32_t skip = 0;
  while (!bad)
    {
    // ---- short sequences: the tokens inside the next 64 compressed bytes are parsed at once ---------------------
    // Every lane reads the byte at ip + lane as if it were a token; hopping from token to token (v_readlane of the
    // per-lane "next token" distance) marks the real ones.  Offsets, output positions (wave prefix sum) and the
    // validity checks are then done for all of them together, and each sequence is one vector copy: lanes below
    // the literal length read the compressed window, the others the output ring (or the sequence's own literals
    // when the match reaches into them).  Tokens with extended lengths end the batch and take the path below.
    if (skip)
      --skip;                                // the last tokens had extended lengths: long sequences, straight to the path below
    else if (ip + 82u <= n)                  // no sequence starting in the batch can be the final, literal-only one
      {
      if (ip + 96u > cw0 + CWIN)
        refill(ip);
      const uint32_t rel = ip - cw0;
      const uint32_t t = cw[rel + (uint32_t)lane];
      const uint32_t lit = t >> 4, mlc = t & 15u;
      const uint64_t cx = __ballot(lit == 15u || mlc == 15u);
      const uint32_t nxt = (uint32_t)lane + 3u + lit;
      uint64_t R = 0;
      uint32_t pos = 0, consumed = 0;
      for (;;)
        {
        if ((cx >> pos) & 1ull) { consumed = pos; break; }
        R |= 1ull << pos;
        const uint32_t nx = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)pos);
        if (nx >= 64u) { consumed = nx; break; }
        pos = nx;
        }
      if (R)
        {
        const bool in_r = ((R >> lane) & 1ull) != 0ull;
        const uint32_t offp = rel + (uint32_t)lane + 1u + lit;
        const uint32_t off = in_r ? ((uint32_t)cw[offp] | ((uint32_t)cw[offp + 1u] << 8)) : 1u;
        const uint32_t outlen = in_r ? lit + mlc + 4u : 0u;
        const uint32_t incl = wave_prefix_add(outlen);
        const uint32_t o = incl - outlen;                                     // output offset of the sequence inside the batch
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total > cap - op || __ballot(in_r && (off == 0u || off > op + o + lit)))
          {
          bad = true;
          break;
          }
        // The batch is copied 64 output bytes at a time, all its sequences together.  The owner of an output byte
        // is the last sequence that starts at or before it (sequence lanes drop their lane number at their first
        // byte, prefix maximum over the lanes).  A literal byte comes from the compressed window; a match byte is
        // the output byte `offset` positions earlier.  When that byte belongs to the same pass, its lane's source
        // is taken over instead (pointer jumping: at most 6 rounds for 64 lanes), which also covers overlapping
        // matches; in the end every lane holds the LDS address of a byte that existed before the pass.
        uint32_t carry = 0;
        for (uint32_t base = 0; base < total; base += 64u)
          {
          own[lane] = 0u;
          if (in_r && o - base < 64u)
            own[o - base] = (uint32_t)lane + 1u;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          uint32_t w = wave_prefix_max(own[lane]);
          w = w > carry ? w : carry;
          carry = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);
          const uint32_t ow = w - 1u;
          const uint32_t oo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ow << 2), (int)o);
          const uint32_t ol = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ow << 2), (int)lit);
          const uint32_t of = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ow << 2), (int)off);
          const uint32_t b = base + (uint32_t)lane, k = b - oo;
          const bool live = b < total;
          uint32_t addr = ORING + rel + ow + 1u + k;            // literal: LDS index of its byte in the window
          int from = -1;                                         // lane of this pass that produces my source byte, -1: none
          if (k >= ol)
            {
            const uint32_t src = op + b - of;                     // output position of the source byte
            addr = src & OMASK;
            if (src >= op + base)
              from = (int)(src - (op + base));
            }
          if (!live)
            from = -1;
          while (__ballot(from >= 0))
            {
            const uint32_t ta = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)addr);
            const int tf = __builtin_amdgcn_ds_bpermute(from << 2, from);
            if (from >= 0)
              {
              addr = ta;
              from = tf;
              }
            }
          if (live)
            {
            const uint8_t v = lds[addr];
            ring[(op + b) & OMASK] = v;
            dst[op + b] = v;
            }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          }
        ip += consumed;
        op += total;
        continue;
        }
      skip = 8;
      }
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

// The decoded size of a block, for callers that only know an upper bound (LZ4_decompress_safe's dstCapacity, lz4.h:153-158): one
// thread follows the sequences as the safe decoder does (lz4.c:1700-1850: token, literal length, literals, offset, match length)
// without moving a byte.  out[0] = size, out[1] = 0; or out[1] = 1 for a block that leaves its input, decodes to more than
// `capacity` bytes or has a match reaching before the output's start.  Slow (a memory round trip per sequence) and rare: the archive
// format always knows the exact size (trico.c:1100-1129), so the decoders are tried with the capacity first.
__global__ void k_lz4_measure(const uint8_t* __restrict__ src, uint32_t size, uint32_t capacity, uint32_t* __restrict__ out)
  {
  uint64_t op = 0;
  uint32_t ip = 0;
  uint32_t bad = 0;
  if (size == 0)
    bad = 1;
  while (!bad)
    {
    const uint32_t token = src[ip++];
    uint64_t lit = token >> 4;
    if (lit == 15)
      {
      uint32_t b;
      do
        {
        if (ip >= size) { bad = 1; break; }
        b = src[ip++];
        lit += b;
        } while (b == 255);
      }
    if (bad || lit > (uint64_t)(size - ip)) { bad = 1; break; }
    ip += (uint32_t)lit;
    op += lit;
    if (op > capacity) { bad = 1; break; }
    if (ip == size)
      break;                                       // the last sequence: literals only
    if (size - ip < 2) { bad = 1; break; }
    const uint32_t off = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8);
    ip += 2;
    if (off == 0 || off > op) { bad = 1; break; }
    uint64_t ml = token & 15;
    if (ml == 15)
      {
      uint32_t b;
      do
        {
        if (ip >= size) { bad = 1; break; }
        b = src[ip++];
        ml += b;
        } while (b == 255);
      }
    if (bad) break;
    op += ml + 4;
    if (op > capacity || ip >= size) { bad = 1; break; }      // (a block never ends with a match: the last five bytes are literals, lz4.c:826)
    }
  out[0] = bad ? 0u : (uint32_t)op;
  out[1] = bad;
  }

} // namespace

int launch_lz4_measure(const uint8_t* d_payload, uint32_t size, uint32_t capacity, uint32_t* d_out)
  {
  hipLaunchKernelGGL(k_lz4_measure, dim3(1), dim3(1), 0, current_stream(), d_payload, size, capacity, d_out);
  return hip_ok(hipGetLastError(), "k_lz4_measure") ? 1 : 0;
  }

int launch_lz4_decode_lds(const uint8_t* const d_payloads[8], const uint32_t sizes[8], int nplanes,
                          uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, uint32_t* d_status)
  {
  Lz4DecArgs a;
  for (int c = 0; c < 8; ++c)
    {
    a.pay[c] = c < nplanes ? d_payloads[c] : nullptr;
    a.size[c] = c < nplanes ? sizes[c] : 0;
    }
  // (the attribute belongs to the function ON A DEVICE: claimed once per device of the process)
  static std::atomic<int> attr_set[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16)
    dev = -1;
  if (dev < 0 || !attr_set[dev].load())
    {
    if (!hip_ok(hipFuncSetAttribute((const void*)k_lz4_decode_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ORING + CWIN)),
                "hipFuncSetAttribute(k_lz4_decode_lds)"))
      return 0;
    if (dev >= 0)
      attr_set[dev].store(1);
    }
  hipLaunchKernelGGL(k_lz4_decode_lds, dim3(nplanes), dim3(WG), ORING + CWIN, current_stream(),
                     a, d_planes, plane_stride, plane_bytes, d_status);
  return hip_ok(hipGetLastError(), "k_lz4_decode_lds") ? 1 : 0;
  }

} // namespace trico
