// k_fpc32_decode.hip — decoder for 32-bit floating-point streams, one wave per component stream.
//
// Replaces trico_decompress (fpsc.c:212-417) + trico_transpose_*_soa_to_aos (transpose_aos_to_soa.c:18-26,
// 58-66): the decoded component is written straight into its slot of the interleaved output.
//
// The format leaves no parallelism inside a stream: value i is xor_i ^ prediction_i and the table keys
// for prediction_{i+1} come from the decoded value i (fpsc.c:308-326).  So the design minimises the
// latency of that one dependent chain instead:
//   * the compressed bytes are staged through LDS in 8 KiB windows with coalesced loads;
//   * per group of 8 values, lanes 0..7 parse the 3-byte header, locate and byte-swap their residuals in
//     parallel (off the chain);
//   * the chain itself runs wave-uniform on the scalar unit: the FCM table (16 entries) lives in one
//     VGPR across lanes 0..15 (v_readlane / v_writelane with scalar index), the DFCM table (1024) in LDS
//     and its read is only waited for when the next code actually uses it;
//   * decoded values are collected in a VGPR (one per lane) and stored 64 at a time.
// Generic table exponents up to (4,10) are honoured (hash_info byte); the archive API always writes (4,10).
//
// This kernel is latency-bound by construction; algorithmic bytes per value:
// its payload share read + 4 written.
#include "common.hpp"

namespace trico {

namespace {

constexpr int WINW = 2048;             // staging window, dwords (8 KiB)
constexpr int WIN_LOW = 64;            // refill when fewer than this many bytes remain (a group needs <= 35)
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

struct DecodeArgs
  {
  const uint8_t* pay[3];
  uint32_t size[3];
  };

__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

__global__ void __launch_bounds__(64) k_fpc32_decode(DecodeArgs args, int arity, uint32_t n, uint32_t* __restrict__ dst,
                                                     uint32_t* __restrict__ status)
  {
  __shared__ uint32_t win[WINW + 4];
  const int lane = threadIdx.x;
  const int c = blockIdx.x;
  const uint8_t* in = args.pay[c];
  const uint32_t len = args.size[c];
  if (len < 5u)
    {
    if (lane == 0) atomicOr(status, 1u);
    return;
    }
  const uint32_t e1 = (uint32_t)(in[0] >> 4) << 1, e2 = (uint32_t)(in[0] & 15) << 1;
  const uint32_t cnt = ((uint32_t)in[1] << 24) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 8) | in[4];
  if (cnt != n || e1 == 0u || e2 == 0u || e1 > 4u || e2 > 10u)
    {
    if (lane == 0) atomicOr(status, 2u);
    return;
    }
  // DFCM table (up to 1024 entries) in 16 VGPRs: entry e lives in register e & 15, lane e >> 4
  u32x16 T2 = (u32x16)(0u);
  const uint32_t m1 = (1u << e1) - 1u, m2 = (1u << e2) - 1u, sh1 = 32u - e1, sh2 = 32u - e2, e2h = e2 >> 1;
  // window over the payload, in units of aligned dwords of the underlying buffer
  const uint32_t al = (uint32_t)((uintptr_t)in & 3u);
  const uint32_t* abase = (const uint32_t*)(in - al);
  const uint32_t total_q = len + al;                     // payload end in aligned-byte coordinates
  uint32_t wd = 0;                                       // first dword of the window
  uint32_t q = 5u + al;                                  // read cursor, aligned-byte coordinates
  auto refill = [&](uint32_t from_q)
    {
    wd = from_q >> 2;
    const uint32_t ndw = (total_q + 3u) >> 2;
    for (uint32_t i = (uint32_t)lane; i < (uint32_t)WINW + 4u; i += 64u)
      win[i] = (wd + i < ndw) ? abase[wd + i] : 0u;
    __syncthreads();
    };
  refill(q);
  uint32_t T1 = 0;                  // FCM table: entry h lives in lane h
  uint32_t h1 = 0, h2 = 0, p1 = 0, p2 = 0, last = 0;
  uint32_t outv = 0;                // lane j holds value (i0 + j) of the current batch of 64
  bool bad = false;
  for (uint32_t i = 0; i < n; i += 8u)
    {
    if (q + WIN_LOW > 4u * (wd + (uint32_t)WINW))
      {
      __syncthreads();
      refill(q);
      }
    // ---- parallel part: lanes 0..7 fetch their residuals --------------------------------------------
    const uint32_t lq = q - 4u * wd;                       // cursor inside the window (bytes)
    const uint8_t* wb = (const uint8_t*)win;
    const uint32_t bc = ((uint32_t)wb[lq] << 16) | ((uint32_t)wb[lq + 1u] << 8) | wb[lq + 2u];
    const uint32_t j = (uint32_t)lane & 7u;
    const uint32_t code = (bc >> (3u * j)) & 7u;
    const uint32_t nb = code <= 4u ? code : code - 4u;
    // bytes of the residuals before mine: sum of lengths of codes 0..j-1
    uint32_t before = 0;
#pragma unroll
    for (uint32_t t = 0; t < 7u; ++t)
      {
      const uint32_t ct = (bc >> (3u * t)) & 7u;
      before += (t < j) ? (ct <= 4u ? ct : ct - 4u) : 0u;
      }
    const uint32_t rp = lq + 3u + before;                  // my residual starts here
    const uint32_t lo = win[rp >> 2], hi = win[(rp >> 2) + 1u];
    const uint32_t raw = __builtin_amdgcn_alignbyte(hi, lo, rp & 3u);         // 4 stream bytes, first in the low byte
    const uint32_t be = __builtin_bswap32(raw);                                // first stream byte on top
    const uint32_t xr = nb ? be >> (8u * (4u - nb)) : 0u;
    // total bytes of the group = 3 + sum of all 8 lengths (lane 7 knows: before + nb)
    const uint32_t gbytes = 3u + rfl((uint32_t)__builtin_amdgcn_readlane((int)(before + nb), 7));
    const uint32_t m = (n - i < 8u) ? (n - i) : 8u;        // values in this group (tail: fpsc.c:329-414)
    if (q + gbytes > total_q)
      {
      bad = true;
      break;
      }
    q += gbytes;
    // ---- the dependent chain (wave-uniform) ----------------------------------------------------------
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k)
      {
      if (k < m)
        {
        const uint32_t ck = (bc >> (3u * k)) & 7u;
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xr, (int)k);
        uint32_t p = p1;
        if (ck > 4u)
          p = p2 + (uint32_t)__builtin_amdgcn_readlane((int)T2[h2 & 15u], (int)(h2 >> 4));   // value + stride (fpsc.c:323)
        const uint32_t v = x ^ p;
        T1 = ((uint32_t)lane == h1) ? v : T1;             // v_writelane semantics via compare + select
        h1 = ((h1 << e1) ^ (v >> sh1)) & m1;
        p1 = (uint32_t)__builtin_amdgcn_readlane((int)T1, (int)h1);
        const uint32_t s = v - last;
        {
        const uint32_t r = h2 & 15u;
        T2[r] = ((uint32_t)lane == (h2 >> 4)) ? s : T2[r];
        }
        h2 = ((h2 << e2h) ^ (s >> sh2)) & m2;
        p2 = v;
        last = v;
        outv = ((uint32_t)lane == ((i + k) & 63u)) ? v : outv;
        }
      }
    if (((i + 8u) & 63u) == 0u || i + 8u >= n)
      {
      const uint32_t i0 = i & ~63u;
      const uint32_t idx = i0 + (uint32_t)lane;
      if (idx < n)
        dst[(size_t)idx * arity + c] = outv;
      }
    }
  if (bad && lane == 0)
    atomicOr(status, 4u);
  }

} // namespace

int launch_fpc32_decode(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, uint32_t n, void* d_dst,
                        uint32_t* d_status)
  {
  DecodeArgs a;
  for (int c = 0; c < 3; ++c)
    {
    a.pay[c] = c < arity ? d_payloads[c] : nullptr;
    a.size[c] = c < arity ? sizes[c] : 0;
    }
  hipLaunchKernelGGL(k_fpc32_decode, dim3(arity), dim3(64), 0, current_stream(), a, arity, n, (uint32_t*)d_dst, d_status);
  return hip_ok(hipGetLastError(), "k_fpc32_decode") ? 1 : 0;
  }

} // namespace trico
