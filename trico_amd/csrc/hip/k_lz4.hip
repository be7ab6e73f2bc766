// k_lz4.hip — wave-cooperative LZ4 block compressor / decompressor, one wave per byte plane.
//
// Compressor: byte-exact with LZ4 1.9.2's LZ4_compress_default as Trico calls it (trico.c:343-368 ->
// lz4.c:1271 -> 1184 -> LZ4_compress_generic lz4.c:793-1181, notLimited, byU16 below 65547 input
// bytes else byU32, noDict, acceleration 1).  The greedy parse is one dependent chain per plane (the
// table content depends on the whole parse history), so the control flow is wave-uniform and only the
// data-parallel parts are spread over the 64 lanes: match-length counting (LZ4_count, lz4.c:539-563)
// compares 512 bytes per iteration, literal runs are copied 64 x 16 bytes per iteration.  The hash
// table (16 KiB) lives in LDS.  Decompressor: LZ4_decompress_safe semantics (lz4.c:1657-2072) with
// wide literal copies and wide (period-aware) match copies.
//
// Roofline: HBM-bound only on long matches / long literal runs; otherwise latency-bound on the
// dependent chain.  Algorithmic bytes per plane byte: 1 read + its share of the block written.
#include "common.hpp"

namespace trico {

namespace {

struct __attribute__((packed, aligned(1))) u32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) u64u { uint64_t v; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) u128u { u32x4 v; };

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return ((const u32u*)p)->v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { return ((const u64u*)p)->v; }

// wave-uniform value from lane 0 (keeps scalar state in SGPRs)
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// number of equal bytes of src[a..] and src[b..], at most `limit` (wave-cooperative LZ4_count)
__device__ __forceinline__ uint32_t wave_count(const uint8_t* __restrict__ src, uint32_t a, uint32_t b, uint32_t limit, int lane)
  {
  uint32_t done = 0;
  while (done < limit)
    {
    const uint32_t o = done + 8u * (uint32_t)lane;
    uint64_t x = 0;
    uint32_t valid = 0;                      // bytes this lane may compare
    if (o < limit)
      {
      valid = limit - o < 8u ? limit - o : 8u;
      if (valid == 8u)
        x = ld64(src + a + o) ^ ld64(src + b + o);
      else
        for (uint32_t k = 0; k < valid; ++k)
          x |= (uint64_t)(src[a + o + k] ^ src[b + o + k]) << (8u * k);
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
  return limit;
  }

// dst[0..n) = src[0..n), non-overlapping, any alignment, all lanes participate
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
      {
      const u32x4 v = ((const u128u*)(src + i + 16u * lane))->v;
      *(u32x4*)(dst + i + 16u * lane) = v;
      }
    }
  for (; i < n; i += 64u)
    if (i + lane < n)
      dst[i + lane] = src[i + lane];
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
__global__ void __launch_bounds__(64) k_lz4_encode_wave(const uint8_t* __restrict__ planes, size_t plane_stride, uint32_t n,
                                                        uint8_t* __restrict__ out_base, size_t out_stride, uint32_t* __restrict__ sizes)
  {
  __shared__ uint32_t tab[4096];   // u32[4096] or, for n < 65547, u16[8192] in the same 16 KiB
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64)
    tab[i] = 0;
  __syncthreads();
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
      uint32_t maxback = ip - anchor < cand ? ip - anchor : cand;
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
      wave_copy(op, src + anchor, lit, lane);
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
        const uint32_t m = wave_count(src, ip + 4u, cand + 4u, room, lane);
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
  wave_copy(op, src + anchor, run, lane);
  op += run;
  }
  if (lane == 0)
    sizes[blockIdx.x] = (uint32_t)(op - dst);
#undef LZ_HASH
#undef LZ_GET
#undef LZ_SET
  }

// ---- decompressor ---------------------------------------------------------------------------------------
struct Lz4DecArgs
  {
  const uint8_t* pay[8];
  uint32_t size[8];
  };

// dst[0..n) = dst[-off..), the LZ77 overlap-aware copy.  With overlap (off < n) the result is periodic
// with period `off`: the first P bytes (P = smallest multiple of off >= 4096) are produced with a modulo
// fill from the bytes before dst, the rest 4 KiB per iteration from P bytes back (already written).
__device__ __forceinline__ void wave_match_copy(uint8_t* __restrict__ dst, uint32_t off, uint32_t n, int lane)
  {
  if (off >= n)
    {
    wave_copy(dst, dst - off, n, lane);
    return;
    }
  const uint8_t* period = dst - off;
  const uint32_t P = off >= 4096u ? off : off * ((4096u + off - 1u) / off);
  const uint32_t first = n < P ? n : P;
  if (off >= 64u)
    {
    // chunks of 64 bytes only read bytes at least `off` >= 64 back: written by earlier iterations
    for (uint32_t i = 0; i < first; i += 64u)
      {
      if (i + lane < first)
        dst[i + lane] = period[i + lane];        // (not dst[i + lane - off]: that index wraps as uint32)
      __builtin_amdgcn_s_waitcnt(0);
      }
    }
  else
    for (uint32_t i = 0; i < first; i += 64u)
      if (i + lane < first)
        dst[i + lane] = period[(i + lane) % off];
  __builtin_amdgcn_s_waitcnt(0);
  uint32_t i = first;
  for (; i + 4096u <= n; i += 4096u)
    {
    u32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = ((const u128u*)(dst + i + 1024u * q + 16u * lane - P))->v;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ((u128u*)(dst + i + 1024u * q + 16u * lane))->v = v[q];
    __builtin_amdgcn_s_waitcnt(0);
    }
  for (; i < n; i += 64u)
    if (i + lane < n)
      dst[i + lane] = dst[i + lane - P];
  }

__global__ void __launch_bounds__(64) k_lz4_decode_wave(Lz4DecArgs a, uint8_t* __restrict__ planes, size_t plane_stride, uint32_t cap,
                                                        uint32_t* __restrict__ status)
  {
  const int lane = threadIdx.x;
  const uint8_t* src = a.pay[blockIdx.x];
  const uint32_t n = a.size[blockIdx.x];
  uint8_t* dst = planes + (size_t)blockIdx.x * plane_stride;
  uint32_t ip = 0, op = 0;
  bool bad = (n == 0);
  while (!bad)
    {
    if (ip >= n) { bad = true; break; }
    const uint32_t tok = uni(src[ip]);
    ++ip;
    uint32_t lit = tok >> 4;
    if (lit == 15u)
      {
      uint32_t bb;
      do { if (ip >= n) { bad = true; break; } bb = uni(src[ip]); ++ip; lit += bb; } while (bb == 255u);
      if (bad) break;
      }
    if (lit > n - ip || lit > cap - op) { bad = true; break; }
    wave_copy(dst + op, src + ip, lit, lane);
    ip += lit; op += lit;
    if (ip == n) break;                                   // last sequence: literals only
    if (n - ip < 2u) { bad = true; break; }
    const uint32_t off = uni((uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8));
    ip += 2;
    if (off == 0u || off > op) { bad = true; break; }
    uint32_t ml = tok & 15u;
    if (ml == 15u)
      {
      uint32_t bb;
      do { if (ip >= n) { bad = true; break; } bb = uni(src[ip]); ++ip; ml += bb; } while (bb == 255u);
      if (bad) break;
      }
    ml += 4u;
    if (ml > cap - op) { bad = true; break; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);                        // earlier stores of this wave must land before they are re-read
    wave_match_copy(dst + op, off, ml, lane);
    op += ml;
    }
  if ((bad || op != cap) && lane == 0)
    atomicOr(status, 8u);
  }

} // namespace

int launch_lz4_encode_wave(const uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, int nplanes, uint8_t* d_out,
                           size_t out_stride, uint32_t* d_sizes)
  {
  hipLaunchKernelGGL(k_lz4_encode_wave, dim3(nplanes), dim3(64), 0, current_stream(),
                     d_planes, plane_stride, plane_bytes, d_out, out_stride, d_sizes);
  return hip_ok(hipGetLastError(), "k_lz4_encode_wave") ? 1 : 0;
  }

int launch_lz4_decode_wave(const uint8_t* const d_payloads[8], const uint32_t sizes[8], int nplanes,
                           uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, uint32_t* d_status)
  {
  Lz4DecArgs a;
  for (int c = 0; c < 8; ++c)
    {
    a.pay[c] = c < nplanes ? d_payloads[c] : nullptr;
    a.size[c] = c < nplanes ? sizes[c] : 0;
    }
  hipLaunchKernelGGL(k_lz4_decode_wave, dim3(nplanes), dim3(64), 0, current_stream(),
                     a, d_planes, plane_stride, plane_bytes, d_status);
  return hip_ok(hipGetLastError(), "k_lz4_decode_wave") ? 1 : 0;
  }

} // namespace trico
