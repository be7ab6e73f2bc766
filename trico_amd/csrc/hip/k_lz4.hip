// k_lz4.hip — LZ4 block compressor for planes below 256 KiB (TRICO_LZ4_CHUNKED_MIN; 4 MiB until round 5), one 1024-thread workgroup per byte plane
// (larger planes: k_lz4_chunked.hip; decompressor: k_lz4_decode.hip).
//
// Compressor: byte-exact with LZ4 1.9.2's LZ4_compress_default as Trico calls it (trico.c:343-368 ->
// lz4.c:1271 -> 1184 -> LZ4_compress_generic lz4.c:793-1181, notLimited, byU16 below 65547 input
// bytes else byU32, noDict, acceleration 1).  The greedy parse is one dependent chain per plane (the
// hash-table content depends on the whole parse history), so wave 0 runs the control flow
// wave-uniformly with the 16 KiB table in LDS; a lone wave issues one instruction per ~5-9 cycles on
// gfx950 (tools/ubench/issue.hip), so everything data-parallel is taken off that wave:
//   * short match counts / literal copies are done by wave 0's 64 lanes (512 B / 1 KiB per iteration);
//   * long ones (>= BULK_MIN bytes) are posted as a job to the 15 helper waves parked on a barrier and
//     done by all 1024 threads, 64 KiB per iteration (LZ4_count, lz4.c:539-563, is a first-mismatch
//     search: per-wave ballot + LDS atomicMin).
//
// Roofline: HBM-bound only on long matches / long literal runs; otherwise latency-bound on the
// dependent chain.  Algorithmic bytes per plane byte: 1 read + its share of the block written.
#include "common.hpp"

namespace trico {

namespace {

constexpr int WG = 1024;
constexpr uint32_t BULK_MIN = 8192;      // bytes from which a count/copy is worth two barriers

struct __attribute__((packed, aligned(1))) u32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) u64u { uint64_t v; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) u128u { u32x4 v; };

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return ((const u32u*)p)->v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { return ((const u64u*)p)->v; }
__device__ __forceinline__ u32x4 ld128(const uint8_t* p) { return ((const u128u*)p)->v; }
__device__ __forceinline__ void st128(uint8_t* p, u32x4 v) { ((u128u*)p)->v = v; }

// wave-uniform value from lane 0 (keeps scalar state in SGPRs)
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// ---- jobs for the whole workgroup ---------------------------------------------------------------------
enum { JOB_EXIT = 0, JOB_COUNT = 1, JOB_COPY = 2 };
struct Job
  {
  uint32_t kind;
  uint32_t n;               // bytes to count / copy
  const uint8_t* a;         // count: first stream;  copy: source
  const uint8_t* b;         // count: second stream; copy: destination (cast)
  uint32_t result;          // count: number of equal bytes
  };

// all WG threads: equal-byte count of a[] and b[] up to n, result in job->result (<= n)
__device__ __forceinline__ void wg_count(Job* job, int tid)
  {
  const uint8_t* a = job->a;
  const uint8_t* b = job->b;
  const uint32_t n = job->n;
  for (uint32_t base = 0; base < n; base += WG * 64u)
    {
    uint32_t first = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      {
      const uint32_t o = base + (uint32_t)q * WG * 16u + 16u * (uint32_t)tid;
      if (o + 16u <= n)
        {
        const u32x4 x = ld128(a + o), y = ld128(b + o);
        const uint64_t d0 = ((uint64_t)(x.y ^ y.y) << 32) | (x.x ^ y.x), d1 = ((uint64_t)(x.w ^ y.w) << 32) | (x.z ^ y.z);
        if (d0 | d1)
          {
          const uint32_t e = d0 ? (uint32_t)__builtin_ctzll(d0) >> 3 : 8u + ((uint32_t)__builtin_ctzll(d1) >> 3);
          first = min(first, o + e);
          }
        }
      else if (o < n)
        {
        for (uint32_t k = o; k < n; ++k)
          if (a[k] != b[k]) { first = min(first, k); break; }
        }
      }
    if (first != 0xffffffffu)
      atomicMin(&job->result, first);
    __syncthreads();
    const uint32_t r = job->result;          // read between two barriers: wave 0 may republish right after
    __syncthreads();
    if (r < n)
      return;
    }
  }

// all WG threads: b[0..n) = a[0..n) (non-overlapping)
__device__ __forceinline__ void wg_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, int tid)
  {
  const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
  const uint32_t h = head < n ? head : n;
  if ((uint32_t)tid < h)
    dst[tid] = src[tid];
  uint32_t i = h + 16u * (uint32_t)tid;
  for (; i + 16u <= n; i += WG * 16u)
    *(u32x4*)(dst + i) = ld128(src + i);
  // tail: fewer than 16 bytes left for exactly one thread
  if (i < n && n - i < 16u)
    for (uint32_t k = i; k < n; ++k)
      dst[k] = src[k];
  }

// helper waves: park on the barrier, run jobs until JOB_EXIT
__device__ __forceinline__ void helper_loop(Job* job, int tid)
  {
  for (;;)
    {
    __syncthreads();                        // job published
    const uint32_t kind = job->kind;
    if (kind == JOB_EXIT)
      return;
    if (kind == JOB_COUNT)
      wg_count(job, tid);
    else
      {
      wg_copy((uint8_t*)job->b, job->a, job->n, tid);
      __syncthreads();
      }
    }
  }

// wave 0: run a job with the whole workgroup
__device__ __forceinline__ uint32_t run_count(Job* job, const uint8_t* a, const uint8_t* b, uint32_t n, int tid)
  {
  if (tid == 0)
    {
    job->kind = JOB_COUNT; job->a = a; job->b = b; job->n = n; job->result = n;
    }
  __syncthreads();
  wg_count(job, tid);
  // wg_count leaves every thread after a barrier at which job->result is final
  return uni(job->result);
  }

__device__ __forceinline__ void run_copy(Job* job, uint8_t* dst, const uint8_t* src, uint32_t n, int tid)
  {
  __builtin_amdgcn_s_waitcnt(0);
  if (tid == 0)
    {
    job->kind = JOB_COPY; job->a = src; job->b = (const uint8_t*)dst; job->n = n;
    }
  __syncthreads();
  wg_copy(dst, src, n, tid);
  __syncthreads();
  }

__device__ __forceinline__ void run_exit(Job* job, int tid)
  {
  if (tid == 0)
    job->kind = JOB_EXIT;
  __syncthreads();
  }

// ---- wave-level primitives (wave 0 only) ----------------------------------------------------------------

// number of equal bytes of a[] and b[], at most `limit`
__device__ __forceinline__ uint32_t wave_count(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t limit, int lane,
                                               uint32_t max_bytes)
  {
  uint32_t done = 0;
  const uint32_t lim = limit < max_bytes ? limit : max_bytes;
  while (done < lim)
    {
    const uint32_t o = done + 8u * (uint32_t)lane;
    uint64_t x = 0;
    uint32_t valid = 0;                      // bytes this lane may compare
    if (o < limit)
      {
      valid = limit - o < 8u ? limit - o : 8u;
      if (valid == 8u)
        x = ld64(a + o) ^ ld64(b + o);
      else
        for (uint32_t k = 0; k < valid; ++k)
          x |= (uint64_t)(a[o + k] ^ b[o + k]) << (8u * k);
      }
    const uint32_t eq = x ? (uint32_t)__builtin_ctzll(x) >> 3 : valid;     // equal bytes in this lane's window
    const uint64_t stop = __ballot(eq < 8u);                                // lanes where the run ends (mismatch or limit)
    if (stop)
      {
      const int first = __builtin_ctzll(stop);
      const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)eq, first);
      return done + 8u * (uint32_t)first + e;
      }
    done += 512u;
    }
  return done;          // all of the first `done` bytes are equal (done >= lim)
  }

// dst[0..n) = src[0..n), non-overlapping, any alignment
__device__ __forceinline__ void wave_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, int lane)
  {
  uint32_t i = 0;
  if (n >= 1024u)
    {
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);      // align the stores
    if ((uint32_t)lane < head)
      dst[lane] = src[lane];
    i = head;
    for (; i + 1024u <= n; i += 1024u)
      *(u32x4*)(dst + i + 16u * lane) = ld128(src + i + 16u * lane);
    }
  for (; i < n; i += 64u)
    if (i + lane < n)
      dst[i + lane] = src[i + lane];
  }

__device__ __forceinline__ uint32_t any_count(Job* job, const uint8_t* a, const uint8_t* b, uint32_t limit, int tid)
  {
  // probe the first BULK_MIN bytes with the wave; continue with the workgroup if they are all equal
  uint32_t m = wave_count(a, b, limit, tid, BULK_MIN);
  if (m >= BULK_MIN && m < limit)
    m += run_count(job, a + m, b + m, limit - m, tid);
  return m;
  }

__device__ __forceinline__ void any_copy(Job* job, uint8_t* dst, const uint8_t* src, uint32_t n, int tid)
  {
  if (n >= BULK_MIN)
    run_copy(job, dst, src, n, tid);
  else
    wave_copy(dst, src, n, tid);
  }

__device__ __forceinline__ uint8_t* put_len(uint8_t* op, uint32_t len, int lane)
  {
  // length extension bytes after a 15 nibble: floor(len/255) x 0xff then len % 255
  const uint32_t q = len / 255u;
  for (uint32_t i = (uint32_t)lane; i < q; i += 64u)
    op[i] = 255;
  if (lane == 0)
    op[q] = (uint8_t)(len - q * 255u);
  return op + q + 1u;
  }

// ---- compressor -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_lz4_encode(const uint8_t* __restrict__ planes, size_t plane_stride, uint32_t n,
                                                   uint8_t* __restrict__ out_base, size_t out_stride, uint32_t* __restrict__ sizes)
  {
  __shared__ uint32_t tab[4096];   // u32[4096] or, for n < 65547, u16[8192] in the same 16 KiB
  __shared__ Job job;
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += WG)
    tab[i] = 0;
  __syncthreads();
  if (tid >= 64)
    {
    helper_loop(&job, tid);
    return;
    }
  const int lane = tid;
  const uint8_t* src = planes + (size_t)blockIdx.x * plane_stride;
  uint8_t* dst = out_base + (size_t)blockIdx.x * out_stride;
  uint16_t* tab16 = (uint16_t*)tab;
  const bool small = n < 65547u;                                                   // lz4.c:570,1190
#define LZ_HASH(p) (small ? ((ld32(p) * 2654435761u) >> 19) : (uint32_t)(((ld64(p) << 24) * 889523592379ull) >> 52))
#define LZ_GET(h) (small ? (uint32_t)tab16[h] : tab[h])
#define LZ_SET(h, v) do { if (lane == 0) { if (small) tab16[h] = (uint16_t)(v); else tab[h] = (v); } } while (0)
  uint8_t* op = dst;
  uint32_t anchor = 0;
  if (n >= 13u)                                                                    // lz4.c:863
    {
    const uint32_t mfl1 = n - 11u, mlim = n - 5u;                                  // lz4.c:825-826
    LZ_SET(uni(LZ_HASH(src)), 0u);
    uint32_t ip = 1;
    uint32_t fh = uni(LZ_HASH(src + 1));
    bool done = false;
    while (!done)
      {
      uint32_t cand = 0, fwd = ip, step = 1, nb = 64;
      for (;;)                                                                     // lz4.c:898-956
        {
        const uint32_t h = fh, cur = fwd;
        cand = uni(LZ_GET(h));
        ip = fwd;
        fwd += step;
        step = nb++ >> 6;
        if (fwd > mfl1) { done = true; break; }
        fh = uni(LZ_HASH(src + fwd));
        LZ_SET(h, cur);
        if (!small && cand + 65535u < cur) continue;
        if (uni(ld32(src + cand)) == uni(ld32(src + ip))) break;
        }
      if (done) break;
      // catch up (lz4.c:960-961): extend the match backwards over equal bytes
      {
      const uint32_t maxback = ip - anchor < cand ? ip - anchor : cand;
      uint32_t back = 0;
      while (back < maxback)
        {
        const uint32_t k = back + (uint32_t)lane + 1u;
        const bool eq = k <= maxback && src[ip - k] == src[cand - k];
        const uint64_t ne = ~__ballot(eq);
        if (ne)
          {
          back += (uint32_t)__builtin_ctzll(ne);
          break;
          }
        back += 64u;
        }
      if (back > maxback) back = maxback;
      ip -= back;
      cand -= back;
      }
      const uint32_t lit = ip - anchor;
      uint8_t* token = op++;
      uint32_t tok = lit >= 15u ? 0xf0u : (lit << 4);
      if (lit >= 15u) op = put_len(op, lit - 15u, lane);
      any_copy(&job, op, src + anchor, lit, tid);
      op += lit;
      for (;;)
        {
        // next_match (lz4.c:1007-1077): offset, match length
        if (lane == 0)
          {
          op[0] = (uint8_t)(ip - cand);
          op[1] = (uint8_t)((ip - cand) >> 8);
          }
        op += 2;
        const uint32_t room = mlim > ip + 4u ? mlim - (ip + 4u) : 0u;
        const uint32_t m = any_count(&job, src + ip + 4u, src + cand + 4u, room, tid);
        ip += m + 4u;
        if (m >= 15u)
          {
          tok |= 15u;
          if (lane == 0) *token = (uint8_t)tok;
          op = put_len(op, m - 15u, lane);
          }
        else if (lane == 0)
          *token = (uint8_t)(tok | m);
        anchor = ip;
        if (ip >= mfl1) { done = true; break; }
        LZ_SET(uni(LZ_HASH(src + ip - 2)), ip - 2u);                               // lz4.c:1088
        const uint32_t h = uni(LZ_HASH(src + ip));
        cand = uni(LZ_GET(h));
        LZ_SET(h, ip);
        if ((small || cand + 65535u >= ip) && uni(ld32(src + cand)) == uni(ld32(src + ip)))
          {
          token = op++;                                                            // lz4.c:1101-1138
          tok = 0;
          continue;
          }
        break;
        }
      if (done) break;
      fh = uni(LZ_HASH(src + (++ip)));
      }
    }
  {
  const uint32_t run = n - anchor;                                                 // lz4.c:1146-1172
  if (lane == 0)
    *op = run >= 15u ? 0xf0 : (uint8_t)(run << 4);
  op += 1;
  if (run >= 15u) op = put_len(op, run - 15u, lane);
  any_copy(&job, op, src + anchor, run, tid);
  op += run;
  }
  if (lane == 0)
    sizes[blockIdx.x] = (uint32_t)(op - dst);
  run_exit(&job, tid);
#undef LZ_HASH
#undef LZ_GET
#undef LZ_SET
  }

} // namespace

int launch_lz4_encode_wave(const uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, int nplanes, uint8_t* d_out,
                           size_t out_stride, uint32_t* d_sizes)
  {
  hipLaunchKernelGGL(k_lz4_encode, dim3(nplanes), dim3(WG), 0, current_stream(),
                     d_planes, plane_stride, plane_bytes, d_out, out_stride, d_sizes);
  return hip_ok(hipGetLastError(), "k_lz4_encode") ? 1 : 0;
  }

} // namespace trico
