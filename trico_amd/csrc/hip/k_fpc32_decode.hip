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

__global__ void __launch_bounds__(64) k_fpc32_decode(DecodeArgs args, int arity, uint32_t n, uint32_t* __restrict__ dst,
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
