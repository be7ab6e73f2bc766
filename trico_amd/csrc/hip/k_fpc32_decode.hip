// k_fpc32_decode.hip — decoder for 32-bit floating-point streams, one wave per component stream.
//
// Replaces trico_decompress (fpsc.c:212-417) + trico_transpose_*_soa_to_aos (transpose_aos_to_soa.c:18-26,
// 58-66): the decoded component is written straight into its slot of the interleaved output.
//
// The format leaves no parallelism inside a stream: value i is xor_i ^ prediction_i and the table keys
// for prediction_{i+1} come from the decoded value i (fpsc.c:308-326).  A lone wave on gfx950 issues
// one instruction per ~5-9 cycles (tools/ubench/issue.hip), so the kernel is organised to put as few
// instructions as possible on that one chain:
//   * compressed bytes are staged through LDS in 8 KiB windows (coalesced loads);
//   * values are handled in batches of 64 = 8 groups.  The 8 group positions are found by a short
//     scalar walk over the 3-byte headers (residual lengths from bit tricks, no per-code loop); then all
//     64 lanes locate, align and byte-swap their own residual at once, and one ballot says which values
//     use the DFCM prediction;
//   * the chain itself is fully unrolled and wave-uniform: residual by v_readlane with a constant lane,
//     FCM table (16 entries) in one VGPR across lanes (compare/select to write, v_readlane to read), DFCM
//     table (1024 entries) in LDS, read only when the value's code asks for it; ~17 instructions per value;
//   * decoded values are dropped into a VGPR with v_writelane and stored 64 at a time.
// Streams with table exponents other than the (4,10) the archive API writes use the generic loop below.
//
// Latency-bound by construction; algorithmic bytes per value: its payload share
// read + 4 written.
#include "common.hpp"
#include <stdlib.h>

// M0 carries the lane select of v_writelane (gfx9 allows one SGPR on the constant bus); the compiler only ever
// sets M0 right before its own uses, so clobbering it inside the asm statement is safe.
#pragma clang diagnostic ignored "-Winline-asm"

namespace trico {

namespace {

constexpr int WINW = 2048;             // staging window, dwords (8 KiB)
constexpr uint32_t BATCH_BYTES = 8 * 35;   // a batch of 8 groups needs at most this many payload bytes

struct DecodeArgs
  {
  const uint8_t* pay[3];
  uint32_t size[3];
  };

__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// sum of the residual lengths of the 3-bit codes packed in x (codes 0..4 -> 0..4 bytes, 5..7 -> 1..3 bytes)
__device__ __forceinline__ uint32_t lens_sum(uint32_t x)
  {
  const uint32_t b0 = x & 0x249249u, b1 = (x >> 1) & 0x249249u, b2 = (x >> 2) & 0x249249u;
  const uint32_t hi = b2 & (b1 | b0);                            // codes 5, 6, 7
  return (uint32_t)__popc(b0) + 2u * (uint32_t)__popc(b1) + 4u * ((uint32_t)__popc(b2) - (uint32_t)__popc(hi));
  }

// Chain state (all wave-uniform except the tables).  Both predictor tables live in registers and are touched only
// when the value's class changes:
//   FCM  table (16 entries): one VGPR, entry h in lane h.  The entry of the current hash is cached in p1: as long
//        as the hash does not change (top bits of consecutive values equal) the table update is `p1 = value`.
//   DFCM table (1024 entries): 16 VGPRs, entry h in lane h & 63 of register h >> 6 (dynamic register index via
//        s_set_gpr_idx, lane via v_readlane / v_writelane).  The entry of the current hash is cached in t2c and
//        `row` holds its register; only a hash change writes the row back and fetches the new one.
// A smooth stream therefore runs on ~15 scalar instructions per value with no table traffic; a noisy one pays
// ~9 more for the register-file table, still without any LDS round trip on the chain.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

struct Chain
  {
  uint32_t h1, h2, p1, t2c, last;   // wave-uniform
  uint32_t T1;                      // FCM table
  u32x16 T2;                        // DFCM table
  uint32_t row;                     // copy of T2[h2 >> 6]
  uint32_t outv;                    // lane k: value k of the current batch
  };

struct Exps { uint32_t e1, e2h, sh1, sh2, m1, m2; };

__device__ __forceinline__ uint32_t chain_value(Chain& c, uint32_t x, bool dfcm, const Exps& e, int lane)
  {
  const uint32_t p = dfcm ? c.last + c.t2c : c.p1;                 // decoder keeps value + stride (fpsc.c:310-311, 323)
  const uint32_t v = x ^ p;
  const uint32_t h1n = ((c.h1 << e.e1) ^ (v >> e.sh1)) & e.m1;     // fpsc.c:76-79
  if (__builtin_expect(h1n != c.h1, 0))
    {
    c.T1 = ((uint32_t)lane == c.h1) ? v : c.T1;                    // hash_table_1[hash1] = value
    c.h1 = h1n;
    c.p1 = (uint32_t)__builtin_amdgcn_readlane((int)c.T1, (int)h1n);
    }
  else
    c.p1 = v;
  const uint32_t s = v - c.last;
  const uint32_t h2n = ((c.h2 << e.e2h) ^ (s >> e.sh2)) & e.m2;    // fpsc.c:81-84
  if (h2n != c.h2)
    {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(c.row) : "s"(s), "s"(c.h2) : "m0");      // hash_table_2[hash2] = stride
    c.T2[c.h2 >> 6] = c.row;
    c.h2 = h2n;
    c.row = c.T2[h2n >> 6];
    c.t2c = (uint32_t)__builtin_amdgcn_readlane((int)c.row, (int)h2n);
    }
  else
    c.t2c = s;
  c.last = v;
  return v;
  }

template <int K>
__device__ __forceinline__ void chain_step(Chain& c, uint32_t xr, uint64_t dfcm, const Exps& e, int lane)
  {
  const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xr, K);
  const uint32_t v = chain_value(c, x, ((dfcm >> K) & 1ull) != 0ull, e, lane);
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(c.outv) : "s"(v), "n"(K));
  }

template <int K, int N> struct Unroll
  {
  static __device__ __forceinline__ void run(Chain& c, uint32_t xr, uint64_t dfcm, const Exps& e, int lane)
    {
    chain_step<K>(c, xr, dfcm, e, lane);
    Unroll<K + 1, N>::run(c, xr, dfcm, e, lane);
    }
  };
template <int N> struct Unroll<N, N>
  {
  static __device__ __forceinline__ void run(Chain&, uint32_t, uint64_t, const Exps&, int) {}
  };

// ---- batches with few DFCM-coded values: prefix scan over the FCM-coded ones --------------------------------
// While the top four bits of the values do not change, an FCM-coded value is predicted by its predecessor
// (fpsc.c:308-309: the table entry of the current hash is the value just decoded), i.e. value = residual ^ previous
// value: a run of FCM-coded values is a prefix XOR of its residuals, which the wave computes at once.  Only the
// DFCM-coded values remain serial points.  At each of them the strides of all earlier values of the batch are
// known, and so are the hashes under which those values stored their strides (fpsc.c:81-84, 323-326), so the
// table read is: the stride of the latest earlier value of the batch with the same hash, else the table as it
// was before the batch.  The table is brought up to date once per batch (last writer per hash wins, found with
// a ds_max of lane numbers).  Random-walk like streams (a handful of DFCM-coded values per 64) decode several
// times faster this way; the assumption (top bits constant over the batch, FCM entry == last value) is checked
// before anything is committed, and a batch that violates it, or has many DFCM-coded values, takes the chain.
constexpr uint32_t SCAN_MAX = 12;    // break-even against the chain: ~1.4 us per batch + ~0.3 us per DFCM-coded value vs 4.2-6 us

__device__ __forceinline__ uint32_t dpp_shr1(uint32_t carry, uint32_t v)
  {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);       // lane l <- lane l-1, lane 0 <- carry
  }

__device__ __forceinline__ uint32_t wave_prefix_xor(uint32_t v)
  {
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);                   // row_shr:1
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);                   // row_shr:2
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);                   // row_shr:4
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);                   // row_shr:8
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);                  // row_bcast:15 into rows 1, 3
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);                  // row_bcast:31 into rows 2, 3
  return v;
  }

__device__ __forceinline__ bool scan_batch(Chain& c, uint32_t xr, uint64_t dfcm, uint32_t* __restrict__ idxtab, int lane)
  {
  if (c.p1 != c.last)
    return false;                                                  // the FCM entry of the current hash is not the last value
  // the DFCM entry cached in registers goes back into the table: the lookups below read the table itself
  asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(c.row) : "s"(c.t2c), "s"(c.h2) : "m0");
  c.T2[c.h2 >> 6] = c.row;
  const bool isd = ((dfcm >> lane) & 1ull) != 0ull;
  const uint32_t P = wave_prefix_xor(isd ? 0u : xr);               // XOR of the FCM residuals up to and including my lane
  uint64_t todo = dfcm;
  uint32_t base = c.last;                                          // (value before the current run) ^ (P before the run)
  int prev = -1;                                                   // lanes up to prev are final
  uint32_t vv = 0, S = 0, Kw = 0, vm1 = 0;
  for (;;)
    {
    vv = (lane > prev) ? (base ^ P) : vv;                          // final below the next serial point, provisional above
    vm1 = dpp_shr1(c.last, vv);                                    // value l-1
    S = vv - vm1;                                                  // stride of value l
    const uint32_t G = S >> 22;
    const uint32_t G1 = dpp_shr1(c.h2 & 31u, G);                   // hash part of stride l-1 (its low five bits survive in h2)
    const uint32_t G2 = dpp_shr1(0u, G1);                          // ... of stride l-2
    Kw = lane == 0 ? c.h2 : (((G2 & 31u) << 5) ^ G1);              // hash under which value l reads and then stores its stride
    if (todo == 0ull)
      break;
    const int b = __builtin_ctzll(todo);
    const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)Kw, b);
    const uint64_t m = __ballot(Kw == key) & ((1ull << b) - 1ull);
    uint32_t stride;
    if (m)
      stride = (uint32_t)__builtin_amdgcn_readlane((int)S, 63 - __builtin_clzll(m));
    else
      {
      const uint32_t r = c.T2[key >> 6];
      stride = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)key);
      }
    const uint32_t vb = (uint32_t)__builtin_amdgcn_readlane((int)xr, b) ^ ((uint32_t)__builtin_amdgcn_readlane((int)vm1, b) + stride);
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vv) : "s"(vb), "s"(b) : "m0");
    base = vb ^ (uint32_t)__builtin_amdgcn_readlane((int)P, b);
    prev = b;
    todo &= todo - 1ull;
    }
  if (__ballot((vv >> 28) != c.h1))
    return false;                                                  // the FCM hash changes inside the batch
  // ---- commit -----------------------------------------------------------------------------------------------
  c.outv = vv;
  atomicMax(&idxtab[Kw], (uint32_t)lane + 1u);                     // last writer (lane + 1) per hash
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t from[16];
#pragma unroll
  for (int r = 0; r < 16; ++r)
    from[r] = idxtab[64 * r + lane];                               // who writes entry (register r, my lane)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  idxtab[Kw] = 0u;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    {
    const uint32_t sv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((from[r] - 1u) << 2), (int)S);
    c.T2[r] = from[r] ? sv : c.T2[r];
    }
  const uint32_t s62 = (uint32_t)__builtin_amdgcn_readlane((int)S, 62), s63 = (uint32_t)__builtin_amdgcn_readlane((int)S, 63);
  c.last = (uint32_t)__builtin_amdgcn_readlane((int)vv, 63);
  c.p1 = c.last;
  c.h2 = (((s62 >> 22) & 31u) << 5) ^ (s63 >> 22);
  c.row = c.T2[c.h2 >> 6];
  c.t2c = (uint32_t)__builtin_amdgcn_readlane((int)c.row, (int)c.h2);
  return true;
  }

__global__ void __launch_bounds__(64) k_fpc32_decode_v1(DecodeArgs args, int arity, uint32_t n, uint32_t* __restrict__ dst,
                                                     uint32_t* __restrict__ status)
  {
  __shared__ uint32_t win[WINW + 4];
  __shared__ uint32_t idxtab[1024];                      // scan path: last writer per DFCM hash, zero between batches
  const int lane = threadIdx.x;
  const int comp = blockIdx.x;
  for (int i = lane; i < 1024; i += 64)
    idxtab[i] = 0u;
  const uint8_t* in = args.pay[comp];
  const uint32_t len = args.size[comp];
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
  // window over the payload, in units of aligned dwords of the underlying buffer
  const uint32_t al = (uint32_t)((uintptr_t)in & 3u);
  const uint32_t* abase = (const uint32_t*)(in - al);
  const uint32_t total_q = len + al;                     // payload end in aligned-byte coordinates
  const uint32_t ndw = (total_q + 3u) >> 2;
  uint32_t wd = 0;                                       // first dword of the window
  uint32_t q = 5u + al;                                  // read cursor, aligned-byte coordinates
  auto refill = [&](uint32_t from_q)
    {
    __syncthreads();
    wd = from_q >> 2;
    for (uint32_t i = (uint32_t)lane; i < (uint32_t)WINW + 4u; i += 64u)
      win[i] = (wd + i < ndw) ? abase[wd + i] : 0u;
    __syncthreads();
    };
  refill(q);
  Chain c;
  c.h1 = 0; c.h2 = 0; c.p1 = 0; c.t2c = 0; c.last = 0; c.T1 = 0; c.row = 0; c.outv = 0;
  c.T2 = (u32x16)(0u);
  const bool standard = (e1 == 4u && e2 == 10u);
  uint32_t i0 = 0;
  bool bad = false;
  if (standard)
    {
    for (; i0 + 64u <= n; i0 += 64u)
      {
      if (q + BATCH_BYTES + 8u > 4u * (wd + (uint32_t)WINW))
        refill(q);
      // ---- positions of the 8 groups: scalar walk over the headers -----------------------------------
      uint32_t lq = q - 4u * wd;
      uint32_t bcv = 0, myq = 0;
#pragma unroll
      for (uint32_t g = 0; g < 8u; ++g)
        {
        const uint32_t w = rfl(__builtin_amdgcn_alignbyte(win[(lq >> 2) + 1u], win[lq >> 2], lq & 3u));
        const uint32_t bc = __builtin_bswap32(w) >> 8;              // 3 header bytes, big-endian (fpsc.c:245-247)
        if (((uint32_t)lane >> 3) == g)
          {
          bcv = bc;
          myq = lq;
          }
        lq += 3u + lens_sum(bc);
        }
      const uint32_t qend = 4u * wd + lq;
      if (qend > total_q)
        {
        bad = true;
        break;
        }
      q = qend;
      // ---- all 64 lanes fetch their residual ------------------------------------------------------------
      const uint32_t j3 = 3u * ((uint32_t)lane & 7u);
      const uint32_t code = (bcv >> j3) & 7u;
      const uint32_t nb = code <= 4u ? code : code - 4u;
      const uint32_t rp = myq + 3u + lens_sum(bcv & ((1u << j3) - 1u));
      const uint32_t raw = __builtin_amdgcn_alignbyte(win[(rp >> 2) + 1u], win[rp >> 2], rp & 3u);
      const uint32_t xr = nb ? __builtin_bswap32(raw) >> (8u * (4u - nb)) : 0u;
      const uint64_t dfcm = __ballot(code > 4u);
      // ---- the dependent chain ---------------------------------------------------------------------------
      if ((uint32_t)__popcll(dfcm) > SCAN_MAX || !scan_batch(c, xr, dfcm, idxtab, lane))
        {
        const Exps es = { 4u, 5u, 28u, 22u, 15u, 1023u };
        Unroll<0, 64>::run(c, xr, dfcm, es, lane);
        }
      dst[(size_t)(i0 + (uint32_t)lane) * arity + comp] = c.outv;
      }
    }
  if (!bad && i0 < n)
    {
    // generic loop: tail of the stream (fewer than 64 values, fpsc.c:329-414) or non-standard exponents
    const Exps eg = { e1, e2 >> 1, 32u - e1, 32u - e2, (1u << e1) - 1u, (1u << e2) - 1u };
    for (uint32_t i = i0; i < n; i += 8u)
      {
      if (q + 64u > 4u * (wd + (uint32_t)WINW))
        refill(q);
      const uint32_t lq = q - 4u * wd;
      const uint32_t w = rfl(__builtin_amdgcn_alignbyte(win[(lq >> 2) + 1u], win[lq >> 2], lq & 3u));
      const uint32_t bc = __builtin_bswap32(w) >> 8;
      const uint32_t j3 = 3u * ((uint32_t)lane & 7u);
      const uint32_t code = (bc >> j3) & 7u;
      const uint32_t nb = code <= 4u ? code : code - 4u;
      const uint32_t rp = lq + 3u + lens_sum(bc & ((1u << j3) - 1u));
      const uint32_t raw = __builtin_amdgcn_alignbyte(win[(rp >> 2) + 1u], win[rp >> 2], rp & 3u);
      const uint32_t xr = nb ? __builtin_bswap32(raw) >> (8u * (4u - nb)) : 0u;
      const uint32_t gbytes = 3u + lens_sum(bc);
      if (q + gbytes > total_q)
        {
        bad = true;
        break;
        }
      q += gbytes;
      const uint32_t m = (n - i < 8u) ? (n - i) : 8u;
      for (uint32_t k = 0; k < m; ++k)
        {
        const uint32_t ck = (bc >> (3u * k)) & 7u;
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xr, (int)k);
        const uint32_t v = chain_value(c, x, ck > 4u, eg, lane);
        if (lane == 0)
          dst[(size_t)(i + k) * arity + comp] = v;
        }
      }
    }
  if (bad && lane == 0)
    atomicOr(status, 4u);
  }

// =====================================================================================================================
// v2: the chain on the vector unit with both predictor tables in LDS, a second wave doing everything that is not the chain
// =====================================================================================================================
// One wave alone issues one instruction per 4 cycles whatever it is, and a dependent scalar instruction costs two issue
// rounds; the v1 chain above (scalar unit, tables in registers) runs at 165-225 cycles per value.  Here the chain is
// branch-free vector code, identical for every code / stream kind:
//     value  = residual ^ (dfcm ? last + T2[a2] : T1[a1])                       add, bfi, xor
//     stride = value - last;  T2[a2] = stride;  T1[a1] = value                  sub, 2 ds_write
//     a2' = ((stride & 0xffc00000) ^ P) >> 20;  P' = (stride & 0xffc00000) << 5  and, xor, shr, shl     (fpsc.c:81-84)
//     a1' = (value >> 26) & 0x3c                                                shr, and                 (fpsc.c:76-79)
//     issue the reads of T2[a2'] and T1[a1'] for the next value                 2 ds_read
// (a1, a2 are LDS byte offsets of the current table entries; the hash of the strides is kept in the top ten bits, where the
// shift by five drops the old bits and the masked low bits are zero, so no further masking is needed).  The only long
// latency on the chain is one LDS round trip per value; the other instructions of the step issue underneath it.
// The second wave of the workgroup walks the group headers, extracts the 64 residuals of the next batch and stores the
// values of the previous batch, exchanging them through double-buffered LDS slots with one barrier per batch.
struct Slot2
  {
  uint32_t xr[64];
  uint32_t dlo, dhi, pad0, pad1;
  };

struct Chain2 { uint32_t last, a1, a2, P, t1, t2; };

template <int K>
__device__ __forceinline__ void chain2_step(Chain2& c, uint32_t vx, uint32_t dlo, uint32_t dhi, uint8_t* __restrict__ T1b,
                                            uint8_t* __restrict__ T2b, uint32_t* __restrict__ ob)
  {
  // residual and code class of value K: independent of the chain, issued while the table reads are in flight
  const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)vx, K);
  uint32_t m;                                                      // all ones if the value is DFCM-coded
  asm volatile("s_bfe_i32 %0, %1, %2" : "=s"(m) : "s"(K < 32 ? dlo : dhi), "n"((K & 31) | 0x10000) : "scc");
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" : "+v"(c.t1), "+v"(c.t2));                       // opaque: keeps the chain on the vector unit (see above)
  const uint32_t p2 = c.last + c.t2;                               // decoder keeps value + stride (fpsc.c:310-311, 323)
  const uint32_t p = (m & p2) | (~m & c.t1);
  const uint32_t v = x ^ p;
  const uint32_t s = v - c.last;
  *(uint32_t*)(T2b + c.a2) = s;                                    // hash_table_2[hash2] = stride
  const uint32_t S22 = s & 0xffc00000u;
  const uint32_t a2n = (S22 ^ c.P) >> 20;
  c.t2 = *(const uint32_t*)(T2b + a2n);
  __builtin_amdgcn_sched_barrier(0);
  c.P = S22 << 5;
  *(uint32_t*)(T1b + c.a1) = v;                                    // hash_table_1[hash1] = value
  const uint32_t a1n = (v >> 26) & 0x3cu;
  c.t1 = *(const uint32_t*)(T1b + a1n);
  ob[K] = v;
  c.a1 = a1n;
  c.a2 = a2n;
  c.last = v;
  }

template <int K, int N> struct Unroll2
  {
  static __device__ __forceinline__ void run(Chain2& c, uint32_t vx, uint32_t dlo, uint32_t dhi, uint8_t* T1b, uint8_t* T2b, uint32_t* ob)
    {
    chain2_step<K>(c, vx, dlo, dhi, T1b, T2b, ob);
    Unroll2<K + 1, N>::run(c, vx, dlo, dhi, T1b, T2b, ob);
    }
  };
template <int N> struct Unroll2<N, N>
  {
  static __device__ __forceinline__ void run(Chain2&, uint32_t, uint32_t, uint32_t, uint8_t*, uint8_t*, uint32_t*) {}
  };

// v3 of the step.  Measured on gfx950 (tools/ubench/lat.hip, lat2.hip): one wave issues one instruction per 4 cycles, dependent
// or not; an LDS read returns after ~48 cycles; but a read of an address whose WRITE is still in flight returns only after
// ~120.  The step above writes T1[a1] and reads T1[a1'] with a1' == a1 almost always (the top bits of consecutive values),
// and the same for T2 on smooth streams, so it ran at 154 cycles per value whatever the stream.  Here
//   * T2 is read BEFORE the stride of this value is written; if both addresses are equal the stride is forwarded;
//   * T1 lives in a register: entry h in lanes 4h..4h+3, written with a lane predicate (computed one value earlier), read
//     with ds_bpermute whose lane address v >> 24 needs no masking (bits 0-1 of the address are ignored, bits 2-3 select
//     one of the four copies).
struct Chain3 { uint32_t last, a2, P, t1, t2raw, s, T1r; bool eq2, pred; };

template <int K, int ABL>
__device__ __forceinline__ void chain3_step(Chain3& c, uint32_t vx, uint32_t dlo, uint32_t dhi, uint32_t L26, uint8_t* __restrict__ T2b,
                                            uint32_t* __restrict__ ob)
  {
  uint32_t x, m;                                                   // m: all ones if the value is DFCM-coded
  if (ABL & 2) { x = dlo; m = dhi; }
  else
    {
    x = (uint32_t)__builtin_amdgcn_readlane((int)vx, K);
    asm volatile("s_bfe_i32 %0, %1, %2" : "=s"(m) : "s"(K < 32 ? dlo : dhi), "n"((K & 31) | 0x10000) : "scc");
    }
  const uint32_t t2 = c.eq2 ? c.s : c.t2raw;
  const uint32_t p2 = c.last + t2;                                 // decoder keeps value + stride (fpsc.c:310-311, 323)
  const uint32_t p = (m & p2) | (~m & c.t1);
  const uint32_t v = x ^ p;
  const uint32_t s = v - c.last;
  const uint32_t S22 = s & 0xffc00000u;
  const uint32_t a2n = (S22 ^ c.P) >> 20;
  if (ABL & 16) c.t2raw = s ^ a2n; else
  c.t2raw = *(const uint32_t*)(T2b + a2n);                         // hash_table_2[new hash2], possibly before ...
  if (!(ABL & 8))
  *(uint32_t*)(T2b + c.a2) = s;                                    // ... hash_table_2[hash2] = stride lands
  c.eq2 = a2n == c.a2;
  c.P = S22 << 5;
  if (ABL & 4) c.t1 = v; else
    {
  c.T1r = c.pred ? v : c.T1r;                                      // hash_table_1[hash1] = value
  c.t1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(v >> 24), (int)c.T1r);
  c.pred = (v ^ L26) < 0x10000000u;                                // my lane holds the entry of this value's top four bits
    }
  if (!(ABL & 1))
  ob[K] = v;
  c.a2 = a2n;
  c.s = s;
  c.last = v;
  }

template <int K, int N, int ABL> struct Unroll3
  {
  static __device__ __forceinline__ void run(Chain3& c, uint32_t vx, uint32_t dlo, uint32_t dhi, uint32_t L26, uint8_t* T2b, uint32_t* ob)
    {
    chain3_step<K, ABL>(c, vx, dlo, dhi, L26, T2b, ob);
    Unroll3<K + 1, N, ABL>::run(c, vx, dlo, dhi, L26, T2b, ob);
    }
  };
template <int N, int ABL> struct Unroll3<N, N, ABL>
  {
  static __device__ __forceinline__ void run(Chain3&, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*) {}
  };

// =====================================================================================================================
// v4: the chain on the SCALAR unit, both predictor tables in global memory behind the scalar data cache
// =====================================================================================================================
// What one wave of gfx950 pays per instruction (tools/ubench/lat3.hip, smem.hip, smem2.hip): any ALU instruction 4 cycles,
// dependent or not; an LDS instruction 13-17 cycles of issue (plus ~48 of latency); a scalar load or store 5-6 cycles of
// issue, a scalar-cache hit ~37 cycles of latency; a branch ~26.  gfx950 still executes scalar STORES (s_store_dword), a
// scalar load issued after a scalar store to the same address returns the stored value, the two low address bits are
// ignored, and dirty lines survive other kernels being dispatched (smem2.hip: 24 chains x 8M operations against the host
// while 10^5 other kernels were launched).  So the tables go where the cheap instructions can reach them: a 4 KiB + 64 B
// scratch per stream in global memory that only this wave touches, zeroed here with scalar stores and written back
// (s_dcache_wb) before the kernel ends so that no dirty line outlives the buffer.  Per value, all on the scalar unit:
//     lm = dfcm ? last : 0                        s_bitcmp1, s_cselect        (while the loads are in flight)
//     q = dfcm ? T2 entry : T1 entry              s_cselect                   (same SCC: s_waitcnt does not touch it)
//     v = x ^ (q + lm); s = v - last              s_add, s_xor, s_sub         (fpsc.c:308-311, 323)
//     T2[a2] = s; a2 = ((s ^ P) >> 20); load T2[a2]   s_store, s_xor, s_lshr, s_load    (fpsc.c:81-84, 324-326; P = (s & 0xffc00000) << 5
//                                                 of the previous value: the hash lives in the top ten bits, the shift drops
//                                                 the old bits, and bits 0-1 of the address are junk the hardware ignores)
//     T1[a1] = v; a1 = v >> 26; load T1[a1]       s_store, s_lshr, s_load     (fpsc.c:76-79, 312-314)
//     P = (s & 0xffc00000) << 5                   s_and, s_lshl
//     out lane K = v; x = residual K + 1          v_writelane, v_readlane
// 18 instructions, ~75 cycles per value whatever the stream (v1: 165-225, the LDS variants above: 146).
// One value.  Registers alternate between consecutive values (value / last, stride / previous stride, hash address /
// previous hash address) so that nothing is copied.  The T2 load of the new hash is issued BEFORE the store of this
// value's stride under the old hash; when both addresses are equal the load has read the entry too early and the next
// value takes the stride from the register instead (`g` = the DFCM mask of the coming value, or 0 if forwarding).  A
// scalar load issued right after a scalar store to the same address is NOT reliably ordered behind it when the line
// misses (smem2.hip under cache pressure; a parity test caught it too); every other store is complete before the
// next value starts, because each value begins with s_waitcnt lgkmcnt(0).
// v_readlane costs ~21 cycles in a scalar instruction stream (tools/ubench/chain4.hip), so the residuals do not come from
// a VGPR: the parser wave puts them into global memory with scalar stores (same scalar cache, same CU) and the chain
// loads eight at a time with s_load_dwordx8.  Fixed registers (clobbered by the statement):
//   s[52:53] = {stride, value} of the previous value on even steps, s[54:55] on odd steps
//   s[56:57] = {T1 entry of the current hash, 0}    s[58:59] = {cand, lm}: one s_cselect_b64 picks {stride, value} or {T1 entry, 0}
//   s[60:67], s[68:75] residuals of the current / next eight values      s[84:99] the FCM table (s_movrels / s_movreld, M0 = hash)
//   D: mask word of value K, DN: mask word of value K + 1
#define CH4_XLOAD(K) "s_load_dwordx8 s[60 + (((" #K ") + 8) & 15) : 67 + (((" #K ") + 8) & 15)], %[Xb], 4 * ((" #K ") + 8)\n"
#define CH4_NOX(K) ""
#define CH4_STEP(K, D, DN, PIN, PINV, POUTS, POUTV, AO, AN, XL)       \
  "s_bitcmp1_b32 %[" D "], (" #K ") & 31\n"                          \
  "s_cselect_b64 s[58:59], " PIN ", s[56:57]\n"                      \
  "s_bitcmp1_b32 %[g], (" #K ") & 31\n"                              \
  "s_waitcnt lgkmcnt(0)\n"                                           \
  "s_cselect_b32 %[q], %[t2], s58\n"                                 \
  "s_add_u32 %[q], %[q], s59\n"                                      \
  "s_xor_b32 " POUTV ", s[60 + ((" #K ") & 15)], %[q]\n"             \
  "s_sub_u32 " POUTS ", " POUTV ", " PINV "\n"                       \
  "s_and_b32 %[h], " POUTS ", 0xffc00000\n"                          \
  "s_xor_b32 %[q], %[h], %[P]\n"                                     \
  "s_lshr_b32 %[" AN "], %[q], 20\n"                                 \
  "s_load_dword %[t2], %[T2b], %[" AN "]\n"                          \
  "s_store_dword " POUTS ", %[T2b], %[" AO "]\n"                     \
  XL(K)                                                               \
  "s_movreld_b32 s84, " POUTV "\n"                                   \
  "s_lshr_b32 m0, " POUTV ", 28\n"                                   \
  "s_lshl_b32 %[P], %[h], 5\n"                                       \
  "s_movrels_b32 s56, s84\n"                                         \
  "s_cmp_lg_u32 %[" AN "], %[" AO "]\n"                              \
  "s_cselect_b32 %[g], %[" DN "], 0\n"                               \
  "v_writelane_b32 %[outv], " POUTV ", " #K "\n"
#define CH4_EVEN(K, D, DN, XL) CH4_STEP(K, D, DN, "s[52:53]", "s53", "s54", "s55", "a2a", "a2b", XL)
#define CH4_ODD(K, D, DN, XL) CH4_STEP(K, D, DN, "s[54:55]", "s55", "s52", "s53", "a2b", "a2a", XL)
#define CH4_OCT(B, D, DN, XL) CH4_EVEN(B + 0, D, D, XL) CH4_ODD(B + 1, D, D, CH4_NOX) CH4_EVEN(B + 2, D, D, CH4_NOX) CH4_ODD(B + 3, D, D, CH4_NOX) \
                              CH4_EVEN(B + 4, D, D, CH4_NOX) CH4_ODD(B + 5, D, D, CH4_NOX) CH4_EVEN(B + 6, D, D, CH4_NOX) CH4_ODD(B + 7, D, DN, CH4_NOX)

// wave-uniform chain state (SGPRs); the FCM table is parked in the stream's scratch between batches
struct Chain4 { uint32_t last, sprev, a2, P, t2, fwd; };

// one batch of 64 values; T2b: the stream's scratch in global memory (DFCM table, then 64 B for the FCM table); Xb: the 64
// residuals of the batch (written by the parser wave with scalar stores).  Returns the 64 values (lane K = value K).
__device__ __forceinline__ uint32_t chain4_batch(Chain4& c, uint32_t dlo, uint32_t dhi, const uint32_t* T2b, const uint32_t* Xb)
  {
  uint32_t outv = 0, a2b, q, h, g;
  asm volatile(
    "s_load_dwordx8 s[60:67], %[Xb], 0x0\n"
    "s_load_dwordx4 s[84:87], %[T2b], 0x1000\n"
    "s_load_dwordx4 s[88:91], %[T2b], 0x1010\n"
    "s_load_dwordx4 s[92:95], %[T2b], 0x1020\n"
    "s_load_dwordx4 s[96:99], %[T2b], 0x1030\n"
    "s_mov_b32 s52, %[sprev]\n"
    "s_mov_b32 s53, %[last]\n"
    "s_mov_b32 s57, 0\n"
    "s_lshr_b32 m0, %[last], 28\n"
    "s_cmp_eq_u32 %[fwd], 0\n"
    "s_cselect_b32 %[g], %[dlo], 0\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_movrels_b32 s56, s84\n"
    CH4_OCT(0, "dlo", "dlo", CH4_XLOAD) CH4_OCT(8, "dlo", "dlo", CH4_XLOAD) CH4_OCT(16, "dlo", "dlo", CH4_XLOAD) CH4_OCT(24, "dlo", "dhi", CH4_XLOAD)
    CH4_OCT(32, "dhi", "dhi", CH4_XLOAD) CH4_OCT(40, "dhi", "dhi", CH4_XLOAD) CH4_OCT(48, "dhi", "dhi", CH4_XLOAD) CH4_OCT(56, "dhi", "dhi", CH4_NOX)
    "s_cselect_b32 %[fwd], 0, 1\n"                     /* SCC still says whether the last load address differed from the last store address */
    "s_mov_b32 %[sprev], s52\n"
    "s_mov_b32 %[last], s53\n"
    "s_store_dwordx4 s[84:87], %[T2b], 0x1000\n"
    "s_store_dwordx4 s[88:91], %[T2b], 0x1010\n"
    "s_store_dwordx4 s[92:95], %[T2b], 0x1020\n"
    "s_store_dwordx4 s[96:99], %[T2b], 0x1030\n"
    "s_waitcnt lgkmcnt(0)\n"
    : [last] "+s"(c.last), [sprev] "+s"(c.sprev), [a2a] "+s"(c.a2), [a2b] "=&s"(a2b), [P] "+s"(c.P), [t2] "+s"(c.t2), [fwd] "+s"(c.fwd),
      [outv] "+v"(outv), [q] "=&s"(q), [h] "=&s"(h), [g] "=&s"(g)
    : [T2b] "s"(T2b), [Xb] "s"(Xb), [dlo] "s"(dlo), [dhi] "s"(dhi)
    : "scc", "memory", "m0", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
      "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94",
      "s95", "s96", "s97", "s98", "s99");
  return outv;
  }

// parser wave: the 64 residuals of a batch (lane K = residual K) go to `Xb` with scalar stores, through the scalar cache the chain reads
__device__ __forceinline__ void chain4_put_residuals(uint32_t xr, const uint32_t* Xb)
  {
#define CH4_PUT4(J, R0, R1, R2, R3) \
  "v_readlane_b32 s" #R0 ", %[xr], 4 * " #J "\n v_readlane_b32 s" #R1 ", %[xr], 4 * " #J " + 1\n" \
  "v_readlane_b32 s" #R2 ", %[xr], 4 * " #J " + 2\n v_readlane_b32 s" #R3 ", %[xr], 4 * " #J " + 3\n" \
  "s_nop 0\n s_store_dwordx4 s[" #R0 ":" #R3 "], %[Xb], 16 * " #J "\n"
  asm volatile(
    CH4_PUT4(0, 52, 53, 54, 55) CH4_PUT4(1, 56, 57, 58, 59) CH4_PUT4(2, 60, 61, 62, 63) CH4_PUT4(3, 64, 65, 66, 67)
    CH4_PUT4(4, 52, 53, 54, 55) CH4_PUT4(5, 56, 57, 58, 59) CH4_PUT4(6, 60, 61, 62, 63) CH4_PUT4(7, 64, 65, 66, 67)
    CH4_PUT4(8, 52, 53, 54, 55) CH4_PUT4(9, 56, 57, 58, 59) CH4_PUT4(10, 60, 61, 62, 63) CH4_PUT4(11, 64, 65, 66, 67)
    CH4_PUT4(12, 52, 53, 54, 55) CH4_PUT4(13, 56, 57, 58, 59) CH4_PUT4(14, 60, 61, 62, 63) CH4_PUT4(15, 64, 65, 66, 67)
    "s_waitcnt lgkmcnt(0)\n"
    :: [xr] "v"(xr), [Xb] "s"(Xb)
    : "memory", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
  }

// reference-order loop of one lane over LDS tables: stream tails (< 64 values) and table exponents below the API's (4,10)
__device__ void serial_values(const uint8_t* __restrict__ in, uint32_t len, uint32_t& pos, uint32_t i0, uint32_t n, uint32_t e1,
                              uint32_t e2, uint32_t& h1, uint32_t& h2, uint32_t& last, uint32_t* T1, uint32_t* T2,
                              uint32_t* __restrict__ dst, int arity, int comp, bool& bad)
  {
  const uint32_t m1 = (1u << e1) - 1u, m2 = (1u << e2) - 1u;
  for (uint32_t i = i0; i < n; i += 8u)
    {
    if (pos + 3u > len) { bad = true; return; }
    const uint32_t bc = ((uint32_t)in[pos] << 16) | ((uint32_t)in[pos + 1] << 8) | in[pos + 2];
    pos += 3u;
    const uint32_t m = (n - i < 8u) ? (n - i) : 8u;
    for (uint32_t k = 0; k < m; ++k)
      {
      const uint32_t code = (bc >> (3u * k)) & 7u;
      const uint32_t nb = code <= 4u ? code : code - 4u;
      if (pos + nb > len) { bad = true; return; }
      uint32_t x = 0;
      for (uint32_t b = 0; b < nb; ++b) x = (x << 8) | in[pos++];
      const uint32_t p = code > 4u ? last + T2[h2] : T1[h1];
      const uint32_t v = x ^ p;
      T1[h1] = v;
      h1 = ((h1 << e1) ^ (v >> (32u - e1))) & m1;
      const uint32_t s = v - last;
      T2[h2] = s;
      h2 = ((h2 << (e2 >> 1)) ^ (s >> (32u - e2))) & m2;
      last = v;
      dst[(size_t)(i + k) * arity + comp] = v;
      }
    }
  }

} // namespace
__device__ unsigned long long g_dec_prof[16];
namespace {

template <int V, int ABL>
__global__ void __launch_bounds__(128) k_fpc32_decode(DecodeArgs args, int arity, uint32_t n, uint32_t* __restrict__ dst,
                                                      uint32_t* __restrict__ status, uint32_t* __restrict__ tables)
  {
  __shared__ uint32_t win[WINW + 4];
  __shared__ uint32_t T2[1024];
  __shared__ uint32_t T1[16];
  __shared__ Slot2 slot[2];
  __shared__ uint32_t outb[2][64];
  __shared__ uint32_t sh_bad, sh_q;
  const int lane = threadIdx.x & 63;
  const int wave = (int)rfl(threadIdx.x >> 6);
  const int comp = blockIdx.x;
  for (int i = threadIdx.x; i < 1024; i += 128)
    T2[i] = 0u;
  if (threadIdx.x < 16)
    T1[threadIdx.x] = 0u;
  if (threadIdx.x == 0)
    {
    sh_bad = 0u;
    sh_q = 5u;
    }
  const uint8_t* in = args.pay[comp];
  const uint32_t len = args.size[comp];
  if (len < 5u)
    {
    if (threadIdx.x == 0) atomicOr(status, 1u);
    return;
    }
  const uint32_t e1 = (uint32_t)(in[0] >> 4) << 1, e2 = (uint32_t)(in[0] & 15) << 1;
  const uint32_t cnt = ((uint32_t)in[1] << 24) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 8) | in[4];
  if (cnt != n || e1 == 0u || e2 == 0u || e1 > 4u || e2 > 10u)
    {
    if (threadIdx.x == 0) atomicOr(status, 2u);
    return;
    }
  __syncthreads();
  const bool standard = (e1 == 4u && e2 == 10u);
  const uint32_t nb = standard ? n / 64u : 0u;
  Chain2 c = { 0u, 0u, 0u, 0u, 0u, 0u };
  Chain3 c3 = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, true, (uint32_t)lane < 4u };
  const uint32_t L26 = (uint32_t)lane << 26;
  Chain4 c4 = { 0u, 0u, 0u, 0u, 0u, 1u };
  const uint32_t* T2g = tables + 2048u * (uint32_t)comp;   // 8 KiB per stream: T2 (4 KiB), T1 (64 B), at 4352: two slots of 64 residuals
  const uint32_t* Xg = T2g + 1088;
  if (V == 4 && wave == 0 && nb)
    {
    for (uint32_t off = 0; off < 4096u + 64u; off += 16u)
      asm volatile("s_mov_b64 s[40:41], 0\n s_mov_b64 s[42:43], 0\n s_store_dwordx4 s[40:43], %0, %1" :: "s"(T2g), "s"(off) : "s40", "s41", "s42", "s43", "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  // ---- parser state (wave 1): window over the payload, in units of aligned dwords of the underlying buffer ----------
  const uint32_t al = (uint32_t)((uintptr_t)in & 3u);
  const uint32_t* abase = (const uint32_t*)(in - al);
  const uint32_t total_q = len + al;                     // payload end in aligned-byte coordinates
  const uint32_t ndw = (total_q + 3u) >> 2;
  uint32_t wd = 0;                                       // first dword of the window
  uint32_t q = 5u + al;                                  // read cursor, aligned-byte coordinates
  auto refill = [&](uint32_t from_q)
    {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    wd = from_q >> 2;
    for (uint32_t i = (uint32_t)lane; i < (uint32_t)WINW + 4u; i += 64u)
      win[i] = (wd + i < ndw) ? abase[wd + i] : 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
  if (wave == 1 && nb)
    refill(q);
  unsigned long long pw_work = 0, pw_wait = 0, cw_work = 0, cw_wait = 0, realt0 = __builtin_amdgcn_s_memrealtime(), cyc0 = __builtin_amdgcn_s_memtime();
  for (uint32_t t = 0; t < nb + 2u; ++t)
    {
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
    if (wave == 1)
      {
      if (t < nb)
        {
        if (q + BATCH_BYTES + 8u > 4u * (wd + (uint32_t)WINW))
          refill(q);
        // ---- positions of the 8 groups: scalar walk over the headers -----------------------------------
        uint32_t lq = q - 4u * wd;
        uint32_t bcv = 0, myq = 0;
#pragma unroll
        for (uint32_t g = 0; g < 8u; ++g)
          {
          const uint32_t w = rfl(__builtin_amdgcn_alignbyte(win[(lq >> 2) + 1u], win[lq >> 2], lq & 3u));
          const uint32_t bc = __builtin_bswap32(w) >> 8;              // 3 header bytes, big-endian (fpsc.c:245-247)
          if (((uint32_t)lane >> 3) == g)
            {
            bcv = bc;
            myq = lq;
            }
          lq += 3u + lens_sum(bc);
          }
        const uint32_t qend = 4u * wd + lq;
        if (qend > total_q)
          {
          if (lane == 0) sh_bad = 1u;
          }
        else
          {
          q = qend;
          // ---- all 64 lanes fetch their residual ------------------------------------------------------------
          const uint32_t j3 = 3u * ((uint32_t)lane & 7u);
          const uint32_t code = (bcv >> j3) & 7u;
          const uint32_t nbytes = code <= 4u ? code : code - 4u;
          const uint32_t rp = myq + 3u + lens_sum(bcv & ((1u << j3) - 1u));
          const uint32_t raw = __builtin_amdgcn_alignbyte(win[(rp >> 2) + 1u], win[rp >> 2], rp & 3u);
          const uint32_t xr = nbytes ? __builtin_bswap32(raw) >> (8u * (4u - nbytes)) : 0u;
          const uint64_t dfcm = __ballot(code > 4u);
          Slot2& sl = slot[t & 1u];
          if (V == 4)
            chain4_put_residuals(xr, Xg + 64u * (t & 1u));
          else
            sl.xr[lane] = xr;
          if (lane == 0)
            {
            sl.dlo = (uint32_t)dfcm;
            sl.dhi = (uint32_t)(dfcm >> 32);
            sh_q = q - al;
            }
          }
        }
      if (t >= 2u)
        dst[(size_t)(64u * (t - 2u) + (uint32_t)lane) * arity + comp] = outb[t & 1u][lane];
      }
    else if (t >= 1u && t <= nb)
      {
      const Slot2& sl = slot[(t - 1u) & 1u];
      const uint32_t vx = sl.xr[lane];
      const uint32_t dlo = rfl(sl.dlo), dhi = rfl(sl.dhi);
      uint32_t* ob = outb[(t - 1u) & 1u];
      if (V == 4)
        ob[lane] = chain4_batch(c4, dlo, dhi, T2g, Xg + 64u * ((t - 1u) & 1u));
      else if (V == 3)
        Unroll3<0, 64, ABL>::run(c3, vx, dlo, dhi, L26, (uint8_t*)T2, ob);
      else
        Unroll2<0, 64>::run(c, vx, dlo, dhi, (uint8_t*)T1, (uint8_t*)T2, ob);
      }
    const unsigned long long tb = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long tc = __builtin_amdgcn_s_memtime();
    if (wave == 1) { pw_work += tb - ta; pw_wait += tc - tb; } else { cw_work += tb - ta; cw_wait += tc - tb; }
    if (sh_bad)
      break;
    }
  if (lane == 0 && comp == 0)
    {
    if (wave == 1) { g_dec_prof[0] = pw_work; g_dec_prof[1] = pw_wait; }
    else { g_dec_prof[2] = cw_work; g_dec_prof[3] = cw_wait; g_dec_prof[4] = __builtin_amdgcn_s_memrealtime() - realt0; g_dec_prof[5] = __builtin_amdgcn_s_memtime() - cyc0; g_dec_prof[6] = nb; }
    }
  bool bad = sh_bad != 0u;
  const uint32_t i0 = 64u * nb;
  if (V == 3)
    {
    if (wave == 0 && (lane & 3) == 0)
      T1[lane >> 2] = c3.T1r;                            // the register copy of the FCM table goes to LDS for the tail loop
    __syncthreads();
    }
  if (V == 4 && nb)
    {
    // no dirty line of the scalar cache may outlive the table buffer; the tail loop below works on LDS copies of the tables
    if (wave == 0)
      asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (i0 < n && !bad)
      {
      for (int i = threadIdx.x; i < 1024; i += 128)
        T2[i] = __builtin_nontemporal_load(T2g + i);
      if (threadIdx.x < 16)
        T1[threadIdx.x] = __builtin_nontemporal_load(T2g + 1024 + threadIdx.x);
      }
    __syncthreads();
    }
  if (!bad && i0 < n && threadIdx.x == 0)
    {
    // tail of the stream (fewer than 64 values, fpsc.c:329-414), or a stream with smaller tables than the API's
    uint32_t pos = sh_q, h1 = c.a1 >> 2, h2 = c.a2 >> 2, last = c.last;
    if (V == 3)
      {
      h1 = c3.last >> 28;
      h2 = c3.a2 >> 2;
      last = c3.last;
      }
    if (V == 4)
      {
      h1 = c4.last >> 28;
      h2 = c4.a2 >> 2;
      last = c4.last;
      }
    serial_values(in, len, pos, i0, n, e1, e2, h1, h2, last, T1, T2, dst, arity, comp, bad);
    }
  if (bad && lane == 0)
    atomicOr(status, 4u);
  }

} // namespace

int launch_fpc32_decode(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, uint32_t n, void* d_dst,
                        uint32_t* d_status, uint32_t* d_tables)
  {
  DecodeArgs a;
  for (int c = 0; c < 3; ++c)
    {
    a.pay[c] = c < arity ? d_payloads[c] : nullptr;
    a.size[c] = c < arity ? sizes[c] : 0;
    }
  static const int variant = getenv("TRICO_FPC32_DEC") ? atoi(getenv("TRICO_FPC32_DEC")) : 4;
  if (variant == 1)
    hipLaunchKernelGGL(k_fpc32_decode_v1, dim3(arity), dim3(64), 0, current_stream(), a, arity, n, (uint32_t*)d_dst, d_status);
  else if (variant == 3)
    {
    static const int abl = getenv("TRICO_FPC32_ABL") ? atoi(getenv("TRICO_FPC32_ABL")) : 0;
#define ABL_CASE(A) case A: hipLaunchKernelGGL((k_fpc32_decode<3, A>), dim3(arity), dim3(128), 0, current_stream(), a, arity, n, (uint32_t*)d_dst, d_status, d_tables); break;
    switch (abl)
      {
      ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(7) ABL_CASE(8) ABL_CASE(15) ABL_CASE(16) ABL_CASE(31) ABL_CASE(24) ABL_CASE(28)
      default: hipLaunchKernelGGL((k_fpc32_decode<3, 0>), dim3(arity), dim3(128), 0, current_stream(), a, arity, n, (uint32_t*)d_dst, d_status, d_tables);
      }
    }
  else if (variant == 4)
    hipLaunchKernelGGL((k_fpc32_decode<4, 0>), dim3(arity), dim3(128), 0, current_stream(), a, arity, n, (uint32_t*)d_dst, d_status, d_tables);
  else
    hipLaunchKernelGGL((k_fpc32_decode<2, 0>), dim3(arity), dim3(128), 0, current_stream(), a, arity, n, (uint32_t*)d_dst, d_status, d_tables);
  return hip_ok(hipGetLastError(), "k_fpc32_decode") ? 1 : 0;
  }

} // namespace trico

extern "C" TRICO_API void trico_hip_debug_dec_prof(unsigned long long out[16])
  {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(trico::g_dec_prof), sizeof(unsigned long long) * 16);
  }
