// k_fpc64.hip — double-precision FCM/DFCM coder: wave-wide encoder and batch-parsing decoder.
//
// Replaces trico_compress_double_precision / trico_decompress_double_precision (fpsc.c:576-800 / 803-1164)
// with exponents (20,20) as the archive API passes them (trico.c:396), fused with the double AoS<->SoA
// transposes (transpose_aos_to_soa.c:28-46, 68-82).
//
// The two tables have 2^20 u64 entries each (8 MiB): they cannot live in LDS, and per-segment copies as in
// the float encoder are impossible, so one wave owns one component stream and walks it in order with the
// tables in global memory (zeroed per call, resident in L2 / Infinity Cache):
//   encoder: 64 values per step.  Classes come from the input alone: FCM class = top 20
//     bits of v[i-1], DFCM class = f(v[i-3..i-1]).  Inside a step the latest earlier value of a class is
//     found with ballots over the distinct classes present (runs of equal class are resolved by the
//     previous lane), across steps by one gather per table; the last lane of every class scatters its
//     payload.  4-bit codes, two values per header byte, residual bytes MSB first; bytes are staged in an
//     LDS ring and flushed as aligned dwords straight to their final place (the wave writes sequentially).
//     One global round trip per step bounds it: ~1 us per 64 values per stream.
//   decoder: two waves.  The parser stages the compressed bytes through LDS and hands batches of 64 residuals (a scalar walk
//     over the 32 header bytes finds the group positions, all lanes fetch / align / byte-swap their residual) to the chain
//     through a ring in scalar memory; the chain is hand-written branch-free scalar code with the tables behind the scalar
//     data cache: one table entry per value - only the table the value's code asks for, requested as soon as its hash is
//     known, and replaced by the value / stride just stored when the hash did not change (the common case on smooth data);
//     on noisy data the DFCM entry is a dependent miss into an 8 MiB table.
// Latency-bound by construction; algorithmic bytes per value: 8 + its payload share.
#include "common.hpp"

namespace trico {

namespace {

typedef unsigned long long u64;

constexpr uint32_t E = 20;                 // table exponent (both tables)
constexpr uint32_t TSIZE = 1u << E;
constexpr uint32_t RING = 4096;            // bytes, encoder output staging (a step emits at most 32 + 512 bytes)
constexpr int PF = 4;

__device__ __forceinline__ uint32_t dpp_shr1(uint32_t carry, uint32_t v)
  {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
  }
__device__ __forceinline__ uint32_t dpp_shl1(uint32_t carry, uint32_t v)
  {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x130, 0xf, 0xf, false);
  }
__device__ __forceinline__ u64 dpp_shr1_64(u64 carry, u64 v)
  {
  return ((u64)dpp_shr1((uint32_t)(carry >> 32), (uint32_t)(v >> 32)) << 32) | dpp_shr1((uint32_t)carry, (uint32_t)v);
  }
__device__ __forceinline__ u64 readlane64(u64 v, int l)
  {
  return ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
  }
__device__ __forceinline__ u64 bpermute64(int src_lane, u64 v)
  {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(v >> 32));
  return ((u64)hi << 32) | lo;
  }
__device__ __forceinline__ uint32_t popc_below(uint64_t mask)
  {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  }
__device__ __forceinline__ uint32_t blen64(u64 x) { return x ? (71u - (uint32_t)__builtin_clzll(x)) >> 3 : 0u; }

// coherent table access (the same wave re-reads entries it wrote a step earlier: bypass the vector L1)
__device__ __forceinline__ u64 tab_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tab_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// src = nearest lower active lane with the same key (-1 if none), last = no higher active lane has it.
// Runs of equal keys are resolved by neighbours.  Whether a class has several runs in this step is probed
// through a 4096-entry lane-id table in LDS (run starts write their lane at LID[key & 4095] and read it
// back: a lane that does not read itself shares the slot with another run); only such classes enter the
// ballot loop.  On data with 64 different classes per step the loop does not run at all.
__device__ __forceinline__ void wave_pred(uint32_t key, bool act, uint8_t* __restrict__ LID, uint64_t lt, int lane,
                                          bool& start, int& src, bool& last)
  {
  const uint32_t kp = dpp_shr1(0xfffffffeu, key), kn = dpp_shl1(0xfffffffeu, key);
  start = act && key != kp;
  const bool end = act && key != kn;
  src = start ? -1 : lane - 1;
  last = end;
  const uint32_t slot = key & 4095u;
  if (start)
    LID[slot] = (uint8_t)lane;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t w = (uint32_t)lane;
  if (start)
    w = LID[slot];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint64_t todo = __ballot(start && w != (uint32_t)lane);
  while (todo)
    {
    const int leader = __builtin_ctzll(todo);
    const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
    const bool mine = key == kk;
    const uint64_t mS = __ballot(mine && start);
    if (mS & (mS - 1ull))                                  // several runs of this class
      {
      const uint64_t mE = __ballot(mine && end);
      if (mine)
        {
        const uint64_t lower = mE & lt;
        if (start)
          src = lower ? 63 - __builtin_clzll(lower) : -1;
        last = end && (mE >> lane) == 1ull;
        }
      }
    todo &= ~mS;
    }
  }

struct Carry64 { u64 m1, m2, m3; };

// ---- encoder -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fpc64_encode(const u64* __restrict__ src, uint32_t n, int arity, uint8_t* __restrict__ out_base,
                                                     size_t out_stride, uint32_t* __restrict__ sizes, u64* __restrict__ tables)
  {
  __shared__ uint32_t ringw[RING / 4];
  __shared__ uint8_t LID[4096];
  uint8_t* ring = (uint8_t*)ringw;
  const int lane = threadIdx.x;
  const int c = blockIdx.x;
  u64* T1 = tables + (size_t)c * 2 * TSIZE;
  u64* T2 = T1 + TSIZE;
  uint8_t* out = out_base + (size_t)c * out_stride;
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t pos = 5, flushed = 0;
  if (lane == 0)
    {
    ring[0] = 0xaa;                          // (20/2) << 4 | (20/2), fpsc.c:609
    ring[1] = (uint8_t)(n >> 24); ring[2] = (uint8_t)(n >> 16); ring[3] = (uint8_t)(n >> 8); ring[4] = (uint8_t)n;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  Carry64 cy = { 0, 0, 0 };
  const uint32_t n2 = (n + 1u) & ~1u;        // slots incl. the tail padding slot (fpsc.c:789-794)
  u64 cur[PF], nxt[PF];
#pragma unroll
  for (int pu = 0; pu < PF; ++pu)
    {
    const uint32_t i = 64u * pu + lane;
    cur[pu] = i < n ? src[(size_t)i * arity + c] : 0ull;
    }
  for (uint32_t ib = 0; ib < n || ib == 0; ib += 64u * PF)
    {
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      {
      const uint32_t i = ib + 64u * PF + 64u * pu + lane;
      nxt[pu] = i < n ? src[(size_t)i * arity + c] : 0ull;
      }
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      {
      const uint32_t i0 = ib + 64u * pu;
      if (i0 >= n && !(n == 0 && i0 == 0))
        break;
      const uint32_t i = i0 + lane;
      const bool act = i < n;
      const u64 v = cur[pu];
      const u64 a = dpp_shr1_64(cy.m1, v), b = dpp_shr1_64(cy.m2, a), d = dpp_shr1_64(cy.m3, b);
      const u64 s = v - a, s1 = a - b, s2 = b - d;
      uint32_t k1 = (uint32_t)(a >> 44);                                                     // fpsc.c:565-568, e1 = 20
      uint32_t k2 = ((((uint32_t)(s2 >> 44)) & 1023u) << 10) ^ (uint32_t)(s1 >> 44);       // fpsc.c:570-573, e2 = 20
      if (!act)
        k1 = k2 = 0xffffffffu;
      int src1, src2;
      bool st1, st2, last1, last2;
      wave_pred(k1, act, LID, lt, lane, st1, src1, last1);
      wave_pred(k2, act, LID, lt, lane, st2, src2, last2);
      u64 p1 = a, p2 = s1;                   // inside a run: previous lane's value / stride
      const bool t1 = st1 && src1 < 0, t2 = st2 && src2 < 0;
      u64 tv1 = 0, tv2 = 0;
      if (__ballot(t1 || t2))
        __builtin_amdgcn_s_waitcnt(0);       // scatters of earlier steps must have reached L2 before they are re-read
      if (t1) tv1 = tab_load(&T1[k1]);
      if (t2) tv2 = tab_load(&T2[k2]);
      if (__ballot((st1 && src1 >= 0) || (st2 && src2 >= 0)))
        {
        const u64 q1 = bpermute64(src1, v), q2 = bpermute64(src2, s);
        if (st1) p1 = q1;
        if (st2) p2 = q2;
        }
      if (t1) p1 = tv1;
      if (t2) p2 = tv2;
      if (last1) tab_store(&T1[k1], v);
      if (last2) tab_store(&T2[k2], s);
      // code selection (fpsc.c:635-782)
      const u64 x1 = v ^ p1, x2 = v ^ (a + p2);
      const uint32_t n1 = blen64(x1);
      uint32_t nn2 = blen64(x2);
      nn2 = nn2 ? nn2 : 1u;
      const bool use2 = n1 > 1u && nn2 < n1;
      uint32_t len = use2 ? nn2 : n1;
      uint32_t code = use2 ? 8u + nn2 : n1;
      u64 x = use2 ? x2 : x1;
      const bool slot = act || (i < n2) || (n == 0 && i < 2u);
      if (!act)
        {
        code = slot ? 1u : 0u;
        len = slot ? 1u : 0u;
        x = 0;
        }
      // layout of the step: [hdr g0][res 0][res 1][hdr g1][res 2][res 3]...
      const uint64_t b0 = __ballot(len & 1u), b1 = __ballot(len & 2u), b2 = __ballot(len & 4u), b3 = __ballot(len & 8u);
      const uint32_t pre = popc_below(b0) + 2u * popc_below(b1) + 4u * popc_below(b2) + 8u * popc_below(b3);
      const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)code, 0xB1, 0xf, 0xf, true);    // partner lane's code
      const uint32_t grp = (uint32_t)lane >> 1;
      const uint32_t rpos = pos + (grp + 1u) + pre;
      for (uint32_t kb = 0; kb < len; ++kb)
        ring[(rpos + kb) & (RING - 1)] = (uint8_t)(x >> (8u * (len - 1u - kb)));
      if (slot && (lane & 1) == 0)
        ring[(pos + grp + pre) & (RING - 1)] = (uint8_t)((other << 4) | code);
      const uint32_t nslots = (uint32_t)__popcll(__ballot(slot));
      pos += (nslots >> 1) + (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2) + 8u * (uint32_t)__popcll(b3);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      while (pos - flushed >= 256u)
        {
        const uint32_t off = flushed + 4u * lane;
        *(uint32_t*)(out + off) = ringw[(off & (RING - 1)) >> 2];
        flushed += 256u;
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      cy.m1 = readlane64(v, 63);
      cy.m2 = readlane64(v, 62);
      cy.m3 = readlane64(v, 61);
      }
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      cur[pu] = nxt[pu];
    if (n == 0)
      break;
    }
  while (flushed < pos)
    {
    const uint32_t off = flushed + 4u * lane;
    if (off < pos)
      {
      const uint32_t w = ringw[(off & (RING - 1)) >> 2];
      if (off + 4u <= pos)
        *(uint32_t*)(out + off) = w;
      else
        for (uint32_t bb = 0; off + bb < pos; ++bb)
          out[off + bb] = (uint8_t)(w >> (8u * bb));
      }
    flushed += 256u;
    }
  if (lane == 0)
    sizes[c] = pos;
  }

// ---- decoder -------------------------------------------------------------------------------------------
constexpr int WINW = 4096;                 // staging window, dwords (16 KiB)
constexpr uint32_t BATCH_BYTES = 32 * 17;  // 32 groups of at most 1 + 16 bytes

struct DecodeArgs
  {
  const uint8_t* pay[3];
  uint32_t size[3];
  };

__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t nib_len(uint32_t c) { return c <= 8u ? c : c - 8u; }

// Two waves: wave 1 parses the group headers and the residuals of the next batches; wave 0 runs the chain and stores the values.
// They are coupled through a ring of RING64 batches and two counters in the stream's scratch, all of it scalar memory
// (k_fpc32_decode.hip has the float version of the same design and what was measured about scalar loads and stores).
constexpr uint32_t RING64 = 4;                                   // batches the parser may run ahead
constexpr uint32_t REC64_DWORDS = 16, SLOT64_DWORDS = 18 * REC64_DWORDS;      // ring slot: 16 quad records of 64 bytes and the one that ends the batch
constexpr uint32_t SCR64_DWORDS = 2048;                          // scratch per component (FPC64_DECODE_SCRATCH_BYTES)
constexpr uint32_t SCR64_PRODUCED = RING64 * SLOT64_DWORDS, SCR64_CONSUMED = SCR64_PRODUCED + 16, SCR64_CODE = SCR64_CONSUMED + 16;
constexpr uint32_t SCR64_USED = SCR64_CODE + 16;
static_assert(SLOT64_DWORDS * 4 == 0x480 && SCR64_PRODUCED * 4 == 0x1200 && SCR64_CONSUMED * 4 == 0x1240 && SCR64_CODE * 4 == 0x1280 &&
              SCR64_USED <= SCR64_DWORDS, "offsets are spelled out in the chain");
constexpr uint32_t ABORT64 = 0xffffffffu;                        // `produced` when the parser gives up

// The 16 quad records of a batch -> ring slot, with scalar stores; complete on return.  w = the residuals (lane K = value K); lane q
// of tlo / thi = address of quad q's body; off = byte offset of the slot from the scratch base; first_d = the batch's first value is
// DFCM-coded (the chain requests that value's table entry itself).
// Record: dwords 0-7 four residuals, 8-9 address of the body, 10 bit of the quad's lane, 11 offset of the next record, 12 first_d.
__device__ __forceinline__ void put_quads64(u64 w, uint32_t tlo, uint32_t thi, const uint32_t* slot, uint32_t off, uint32_t first_d)
  {
  const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
#define P64_Q(Q, R) \
  "v_readlane_b32 s[" #R "], %[lo], 4 * (" #Q ")\n v_readlane_b32 s[" #R " + 1], %[hi], 4 * (" #Q ")\n" \
  "v_readlane_b32 s[" #R " + 2], %[lo], 4 * (" #Q ") + 1\n v_readlane_b32 s[" #R " + 3], %[hi], 4 * (" #Q ") + 1\n" \
  "v_readlane_b32 s[" #R " + 4], %[lo], 4 * (" #Q ") + 2\n v_readlane_b32 s[" #R " + 5], %[hi], 4 * (" #Q ") + 2\n" \
  "v_readlane_b32 s[" #R " + 6], %[lo], 4 * (" #Q ") + 3\n v_readlane_b32 s[" #R " + 7], %[hi], 4 * (" #Q ") + 3\n" \
  "v_readlane_b32 s[" #R " + 8], %[tlo], " #Q "\n v_readlane_b32 s[" #R " + 9], %[thi], " #Q "\n" \
  "s_mov_b32 s[" #R " + 10], 1 << (" #Q ")\n s_add_u32 s[" #R " + 11], %[off], 64 * ((" #Q ") + 1)\n" \
  "s_store_dwordx4 s[" #R ":" #R " + 3], %[slot], 64 * (" #Q ")\n s_store_dwordx4 s[" #R " + 4:" #R " + 7], %[slot], 64 * (" #Q ") + 16\n" \
  "s_store_dwordx4 s[" #R " + 8:" #R " + 11], %[slot], 64 * (" #Q ") + 32\n"
  asm volatile("s_store_dword %[fd], %[slot], 48\n"
               P64_Q(0, 52) P64_Q(1, 64) P64_Q(2, 52) P64_Q(3, 64) P64_Q(4, 52) P64_Q(5, 64) P64_Q(6, 52) P64_Q(7, 64)
               P64_Q(8, 52) P64_Q(9, 64) P64_Q(10, 52) P64_Q(11, 64) P64_Q(12, 52) P64_Q(13, 64) P64_Q(14, 52) P64_Q(15, 64)
               "s_waitcnt lgkmcnt(0)\n"
               :: [lo] "v"(lo), [hi] "v"(hi), [tlo] "v"(tlo), [thi] "v"(thi), [slot] "s"(slot), [off] "s"(off), [fd] "s"(first_d)
               : "scc", "memory", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
                 "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75");
  }

// the record that ends a batch: only the address in it counts
__device__ __forceinline__ void put_end64(uint64_t target, const uint32_t* slot)
  {
  asm volatile("s_store_dwordx2 %[t], %[slot], 64 * 16 + 32\n s_waitcnt lgkmcnt(0)\n" :: [t] "s"(target), [slot] "s"(slot) : "memory");
  }

__device__ __forceinline__ uint32_t counter_load64(const uint32_t* base, uint32_t dword)
  {
  uint32_t r;
  asm volatile("s_load_dword %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(base), "s"(4u * dword) : "memory");
  return r;
  }
__device__ __forceinline__ void counter_store64(const uint32_t* base, uint32_t dword, uint32_t v)
  {
  asm volatile("s_store_dword %0, %1, %2\n s_waitcnt lgkmcnt(0)" :: "s"(v), "s"(base), "s"(4u * dword) : "memory");
  }

// ---- the chain (wave 0) ------------------------------------------------------------------------------------------------
// Per value ONE table entry is needed: the DFCM entry if the value is DFCM-coded, else the FCM entry (fpsc.c:977-978); both
// tables are written for every value (fpsc.c:980-995).  With the API's exponents (20, 20) the FCM hash is the top 20 bits of the
// previous value and the DFCM hash ((h2 << 10) ^ top 20 bits of the stride) & 0xfffff; both are kept as byte offsets (<< 3).
// Round 2's chain ran the same 30 branch-free scalar instructions for every value: the entry of the next value was requested
// behind both hashes, both stores and five selects (54 ns per value on smooth streams, 112 on noisy ones).  Since round 4 the
// parser hands the values over in quads with the address of straight-line code for their pattern of kinds and the kind of the
// value behind them (tools/gen_chain64.py -> chain64_bodies.inc; the float chain's construction, k_fpc32_decode.hip):
//     wait; e = forwarded ? what the value before stored : loaded entry          s_waitcnt, s_cmp_lg, s_cselect_b64
//     D: v = x ^ (e + last)     F: v = x ^ e                                      (s_add, s_addc,) s_xor_b64     (fpsc.c:977-981)
//     next is F: o1' = (v.hi >> 9) & 0x7ffff8; load T1[o1']; forwarded' = (o1' == o1)      2 + 1 + 2 instructions behind v
//     next is D: s = v - last; o2' = ((o2 << 10) ^ (s.hi >> 9)) & 0x7ffff8; load T2[o2']; forwarded' = (o2' == o2)
//     then the rest: s, T1[o1] = v, T2[o2] = s, the other hash                    s_store_dwordx2 x 2, ...      (fpsc.c:982-995)
//     lane of the quad in output register pair j = v                              2 v_mov (EXEC = that lane)
// 21-24 instructions per value, the load 5-9 instructions behind the value instead of ~20.  A scalar load is not ordered behind a
// scalar store to the same address that is still in flight: the entry a value needs is taken from the register when the value
// before stored to the same offset (`forwarded`), and every value begins with s_waitcnt lgkmcnt(0), which completes all older stores.
// On noisy doubles the DFCM entry is a miss into an 8 MiB table on top of that.
#include "chain64_bodies.inc"

// The whole chain of a stream: batches 0 .. nb-1 of 64 values (the last one may hold fewer: `last_mask` has a bit per value of
// it; the quads beyond run too, nothing reads the tables or the state after them).  Per batch: wait until the parser has
// published it, load its first record, request the entry of its first value (nothing of the batch before is still in flight
// then), 16 quads, eight vector stores of 16 lanes straight to their place in the interleaved output (lane q of register pair j
// holds value 4 q + j), publish `consumed`.
//   s59 / s[62:63] offset / table of the entry a batch begins with   s[60:61] output address of the batch   s96 scratch   s97 batch counter
constexpr uint32_t SPIN64_LIMIT_CHAIN = 1u << 25, SPIN64_LIMIT_PARSER = 1u << 23;

// returns 1 if the wait for the parser ran into its bound
__device__ __forceinline__ uint32_t chain64_run(const u64* T1, const u64* T2, const uint32_t* ring, uint32_t nb, uint64_t last_mask, u64* out0,
                                            uint32_t lane, uint32_t arity, uint32_t out_step)
  {
  const uint64_t ob = (uint64_t)(uintptr_t)out0;
  const uint32_t olo = (uint32_t)ob, ohi = (uint32_t)(ob >> 32);
  // value 4 q + j of a batch: byte offset in the output, and which values of the last batch exist
  const uint32_t va0 = 32u * lane * arity, va1 = va0 + 8u * arity, va2 = va0 + 16u * arity, va3 = va0 + 24u * arity;
  uint32_t m[4] = { 0, 0, 0, 0 };
  for (uint32_t q = 0; q < 16u; ++q)
    for (uint32_t j = 0; j < 4u; ++j)
      m[j] |= (uint32_t)((last_mask >> (4u * q + j)) & 1ull) << q;
  uint32_t o0lo, o0hi, o1lo, o1hi, o2lo, o2hi, o3lo, o3hi, spin, tmo;
  asm volatile(
    "s_mov_b64 s[36:37], 0\n s_mov_b64 s[38:39], 0\n s_mov_b64 s[40:41], 0\n s_mov_b64 s[42:43], 0\n"
    "s_mov_b64 s[44:45], 0\n"
    "s_mov_b32 s52, 0\n s_mov_b32 s53, 0\n s_mov_b32 s54, 0\n s_mov_b32 s55, 0\n s_mov_b32 s56, 0\n"
    "s_mov_b64 s[46:47], %[T1]\n s_mov_b64 s[48:49], %[T2]\n s_mov_b64 s[98:99], %[ringp]\n"
    "s_mov_b32 s97, 0\n s_mov_b32 s60, %[olo]\n s_mov_b32 s61, %[ohi]\n s_mov_b32 exec_hi, 0\n"
    "s_mov_b32 %[spin], 0\n s_mov_b32 %[tmo], 0\n"
    /* where the bodies are: the parser puts their addresses into the records */
    "s_getpc_b64 s[62:63]\n"
    ".Lc64_pc_%=:\n"
    "s_add_u32 s62, s62, .Lc64_body_%= - .Lc64_pc_%=\n"
    "s_addc_u32 s63, s63, 0\n"
    "s_store_dwordx2 s[62:63], s[98:99], 0x1280\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_mov_b32 s96, 1\n"
    "s_store_dword s96, s[98:99], 0x1288\n"
    "s_cmp_lt_u32 s97, %[nb]\n"
    "s_cbranch_scc0 .Lc64_done_%=\n"
    "s_branch .Lc64_poll_%=\n"
    CH64_BODIES
    ".Lc64_end_%=:\n"
    "s_add_u32 s97, s97, 1\n"
    "s_cmp_eq_u32 s97, %[nb]\n"                        /* the last batch may hold fewer than 64 values */
    "s_cselect_b32 s57, %[m0], 0xffff\n s_mov_b32 exec_lo, s57\n"
    "global_store_dword %[va0], %[o0lo], s[60:61]\n global_store_dword %[va0], %[o0hi], s[60:61] offset:4\n"
    "s_cselect_b32 s57, %[m1], 0xffff\n s_mov_b32 exec_lo, s57\n"
    "global_store_dword %[va1], %[o1lo], s[60:61]\n global_store_dword %[va1], %[o1hi], s[60:61] offset:4\n"
    "s_cselect_b32 s57, %[m2], 0xffff\n s_mov_b32 exec_lo, s57\n"
    "global_store_dword %[va2], %[o2lo], s[60:61]\n global_store_dword %[va2], %[o2hi], s[60:61] offset:4\n"
    "s_cselect_b32 s57, %[m3], 0xffff\n s_mov_b32 exec_lo, s57\n"
    "global_store_dword %[va3], %[o3lo], s[60:61]\n global_store_dword %[va3], %[o3hi], s[60:61] offset:4\n"
    "s_add_u32 s60, s60, %[ostep]\n"
    "s_addc_u32 s61, s61, 0\n"
    "s_store_dword s97, s[98:99], 0x1240\n"
    "s_cmp_lt_u32 s97, %[nb]\n"
    "s_cbranch_scc0 .Lc64_done_%=\n"
    ".Lc64_poll_%=:\n"
    "s_load_dword s96, s[98:99], 0x1200\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_cmp_gt_u32 s96, s97\n"
    "s_cbranch_scc1 .Lc64_go_%=\n"
    /* bounded wait (see SPIN_LIMIT_CHAIN in k_fpc32_decode.hip): seconds without a published batch = the waves lost each other */
    "s_add_u32 %[spin], %[spin], 1\n"
    "s_cmp_lt_u32 %[spin], %[limit]\n"
    "s_cbranch_scc0 .Lc64_tmo_%=\n"
    "s_sleep 1\n"
    "s_branch .Lc64_poll_%=\n"
    ".Lc64_tmo_%=:\n"
    "s_mov_b32 %[tmo], 1\n"
    "s_mov_b32 s96, -1\n"
    "s_store_dword s96, s[98:99], 0x1240\n"          /* consumed = ABORT: the parser stops too */
    "s_branch .Lc64_done_%=\n"
    ".Lc64_go_%=:\n"
    "s_mov_b32 %[spin], 0\n"
    "s_cmp_eq_u32 s96, -1\n"
    "s_cbranch_scc1 .Lc64_done_%=\n"
    "s_and_b32 s96, s97, 3\n"
    "s_mul_i32 s96, s96, 0x480\n"
    "s_load_dwordx16 s[64:79], s[98:99], s96\n"
    "s_waitcnt lgkmcnt(0)\n"
    /* the entry of the batch's first value: the stores of the batch before are complete (the poll waited for them) */
    "s_cmp_lg_u32 s76, 0\n"
    "s_cselect_b32 s59, s54, s52\n"
    "s_cselect_b64 s[62:63], s[48:49], s[46:47]\n"
    "s_load_dwordx2 s[44:45], s[62:63], s59\n"
    "s_mov_b32 s56, 0\n"
    "s_setpc_b64 s[72:73]\n"
    ".Lc64_done_%=:\n"
    "s_mov_b64 exec, -1\n"
    "s_waitcnt lgkmcnt(0)\n"
    : [o0lo] "=&v"(o0lo), [o0hi] "=&v"(o0hi), [o1lo] "=&v"(o1lo), [o1hi] "=&v"(o1hi), [o2lo] "=&v"(o2lo), [o2hi] "=&v"(o2hi),
      [o3lo] "=&v"(o3lo), [o3hi] "=&v"(o3hi), [spin] "=&s"(spin), [tmo] "=&s"(tmo)
    : [T1] "s"(T1), [T2] "s"(T2), [ringp] "s"(ring), [olo] "s"(olo), [ohi] "s"(ohi), [va0] "v"(va0), [va1] "v"(va1), [va2] "v"(va2), [va3] "v"(va3),
      [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]), [ostep] "s"(out_step), [nb] "s"(nb), [limit] "s"(SPIN64_LIMIT_CHAIN)
    : "scc", "memory", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",
      "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68",
      "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89",
      "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
  return tmo;
  }

__device__ __forceinline__ uint64_t rfl64(uint64_t x) { return ((uint64_t)rfl((uint32_t)(x >> 32)) << 32) | rfl((uint32_t)x); }

// one chain = one workgroup of two waves; `tables`: 2 x TSIZE entries of this chain, `scratch`: its ring and counters
__device__ __forceinline__ void decode64_pair(const Fpc64ChainJob& job, u64* __restrict__ tables, uint32_t* __restrict__ scratch)
  {
  __shared__ uint32_t win[WINW + 8];
  __shared__ uint32_t sh_bad;
  const int lane = threadIdx.x & 63;
  const int wave = (int)rfl(threadIdx.x >> 6);
  const uint8_t* in = job.pay;
  const uint32_t n = job.n;
  const int arity = (int)job.stride;
  u64* dst = (u64*)job.dst;
  uint32_t* status = job.status;
  if (threadIdx.x == 0)
    sh_bad = 0u;
  const uint32_t len = job.size;
  if (len < 5u)
    {
    if (threadIdx.x == 0) atomicOr(status, FPC_STATUS_SHORT);
    return;
    }
  const uint32_t e1 = rfl((uint32_t)(in[0] >> 4) << 1), e2 = rfl((uint32_t)(in[0] & 15) << 1);
  const uint32_t cnt = rfl(((uint32_t)in[1] << 24) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 8) | in[4]);
  if (cnt != n || e1 != E || e2 != E)                                    // other table shapes: k_serial.hip (shim.hip routes them)
    {
    if (threadIdx.x == 0) atomicOr(status, FPC_STATUS_HEADER);
    return;
    }
  const u64* T1 = tables;
  const u64* T2 = T1 + TSIZE;
  const uint32_t* ring = scratch;                                          // RING64 slots, then the two counters
  const uint32_t nb = (n + 63u) / 64u;                                     // the last batch may be partial
  if (wave == 0)
    {
    // ring and counters start at zero; scalar stores of whole lines, so that every line is in the scalar cache whatever it held
    for (uint32_t off = 0; off < 4u * SCR64_USED; off += 16u)
      asm volatile("s_mov_b64 s[40:41], 0\n s_mov_b64 s[42:43], 0\n s_store_dwordx4 s[40:43], %0, %1" :: "s"(ring), "s"(off) : "s40", "s41", "s42", "s43", "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  __syncthreads();
  if (wave == 1)
    {
    // ---- parser: group headers and residuals of batch t ------------------------------------------------------------
    const uint32_t al = (uint32_t)((uintptr_t)in & 3u);
    const uint32_t* abase = (const uint32_t*)(in - al);
    const uint32_t total_q = len + al;
    const uint32_t ndw = (total_q + 3u) >> 2;
    uint32_t wd = 0, q = 5u + al;
    auto refill = [&](uint32_t from_q)
      {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      wd = from_q >> 2;
      for (uint32_t i = (uint32_t)lane; i < (uint32_t)WINW + 8u; i += 64u)
        win[i] = (wd + i < ndw) ? abase[wd + i] : 0u;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };
    refill(q);
    const uint8_t* wb = (const uint8_t*)win;
    uint32_t t = 0;
    uint32_t failed = 0;                                 // 1: malformed payload, 2: the chain stopped answering
    uint32_t spins = 0;
    // where the chain's bodies are (it publishes that first thing, behind the zeroing of its tables), and the records that end the batches
    while (counter_load64(ring, SCR64_CODE + 2u) == 0u)
      {
      if (++spins > SPIN64_LIMIT_PARSER)
        {
        failed = 2u;
        break;
        }
      __builtin_amdgcn_s_sleep(4);
      }
    const uint64_t code_base = ((uint64_t)counter_load64(ring, SCR64_CODE + 1u) << 32) | counter_load64(ring, SCR64_CODE);
    for (uint32_t sl = 0; sl < RING64 && !failed; ++sl)
      put_end64(code_base + (uint64_t)CH64_SLOT_END * (uint64_t)CH64_STRIDE, ring + SLOT64_DWORDS * sl);
    spins = 0;
    while (t < nb && !failed)
      {
      const uint32_t cons = counter_load64(ring, SCR64_CONSUMED);
      if (cons == ABORT64)                               // the chain gave up waiting (it reports the timeout itself)
        break;
      if (t >= cons + RING64)
        {
        if (++spins > SPIN64_LIMIT_PARSER)
          {
          failed = 2u;
          break;
          }
        __builtin_amdgcn_s_sleep(4);
        continue;
        }
      spins = 0;
      if (q + BATCH_BYTES + 16u > 4u * (wd + (uint32_t)WINW))
        refill(q);
      const uint32_t i0 = 64u * t;
      const uint32_t nvals = n - i0 < 64u ? n - i0 : 64u;
      const uint32_t ngroups = (nvals + 1u) >> 1;
      // positions of the groups: scalar walk over the header bytes
      uint32_t lq = q - 4u * wd;
      uint32_t myhdr = 0, myq = 0;
      for (uint32_t g = 0; g < ngroups; ++g)
        {
        const uint32_t hdr = rfl((uint32_t)wb[lq]);
        if (((uint32_t)lane >> 1) == g)
          {
          myhdr = hdr;
          myq = lq;
          }
        lq += 1u + nib_len(hdr & 15u) + nib_len(hdr >> 4);
        }
      const uint32_t qend = 4u * wd + lq;
      if (qend > total_q)
        {
        failed = 1u;
        break;
        }
      q = qend;
      // all lanes fetch their residual
      const uint32_t code = (lane & 1) ? (myhdr >> 4) : (myhdr & 15u);
      const uint32_t nbytes = nib_len(code);
      const uint32_t rp = myq + 1u + ((lane & 1) ? nib_len(myhdr & 15u) : 0u);
      const uint32_t w0 = win[rp >> 2], w1 = win[(rp >> 2) + 1u], w2 = win[(rp >> 2) + 2u];
      const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, rp & 3u), hi = __builtin_amdgcn_alignbyte(w2, w1, rp & 3u);
      const u64 be = ((u64)__builtin_bswap32(lo) << 32) | __builtin_bswap32(hi);      // first stream byte on top
      const u64 xr = nbytes ? be >> (8u * (8u - nbytes)) : 0ull;
      const uint64_t dfcm = __ballot(code > 8u);
      // the body of quad q (lanes 0..15): by the parity of q, the kinds of its four values and the kind of the value behind it
      // (the last quad of a batch takes the bodies that assume an F there: the chain requests the first entry of a batch itself)
      const uint32_t slot_idx = (((uint32_t)lane & 1u) << 5) | ((uint32_t)(dfcm >> (4u * ((uint32_t)lane & 15u))) & 31u);
      const uint64_t target = code_base + (uint64_t)slot_idx * (uint64_t)CH64_STRIDE;
      const uint32_t sl = t % RING64;
      put_quads64(xr, (uint32_t)target, (uint32_t)(target >> 32), ring + SLOT64_DWORDS * sl, 4u * SLOT64_DWORDS * sl, (uint32_t)(dfcm & 1ull));
      ++t;
      counter_store64(ring, SCR64_PRODUCED, t);
      }
    if (failed)
      {
      counter_store64(ring, SCR64_PRODUCED, ABORT64);
      if (lane == 0)
        atomicOr(&sh_bad, failed);
      }
    }
  else
    {
    // ---- chain (wave-uniform, scalar unit) -------------------------------------------------------------------------
    // The tables start at zero (fpsc.c:822-833).  They are zeroed HERE, through the scalar cache this wave will read them
    // through, so that no stale line of an earlier kernel that used the same buffer can be hit (2 x 8 MiB = 1 M stores of 16
    // bytes, ~3 ms).
    for (uint32_t off = 0; off < 2u * TSIZE * 8u; off += 64u)
      asm volatile("s_mov_b64 s[40:41], 0\n s_mov_b64 s[42:43], 0\n"
                   "s_store_dwordx4 s[40:43], %0, %1\n s_store_dwordx4 s[40:43], %0, %2\n"
                   "s_store_dwordx4 s[40:43], %0, %3\n s_store_dwordx4 s[40:43], %0, %4"
                   :: "s"(T1), "s"(off), "s"(off + 16u), "s"(off + 32u), "s"(off + 48u) : "s40", "s41", "s42", "s43", "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(3);                        // the chain owns its SIMD's issue slots whenever it can issue
    const uint32_t nlast = n - 64u * (nb - 1u);           // values of the last batch, 1 .. 64
    const uint64_t last_mask = nlast >= 64u ? ~0ull : (1ull << nlast) - 1ull;
    const uint32_t tmo = chain64_run(T1, T2, ring, nb, last_mask, dst, (uint32_t)lane, (uint32_t)arity, 512u * (uint32_t)arity);
    __builtin_amdgcn_s_setprio(0);
    if (tmo && lane == 0)
      atomicOr(&sh_bad, 2u);
    // no dirty line of the scalar cache may outlive the table buffer
    asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  __syncthreads();
  if (sh_bad && threadIdx.x == 0)
    atomicOr(status, ((sh_bad & 1u) ? FPC_STATUS_MALFORMED : 0u) | ((sh_bad & 2u) ? FPC_STATUS_TIMEOUT : 0u));
  }

struct ChainJobs3d { Fpc64ChainJob j[3]; };

__device__ __forceinline__ Fpc64ChainJob uniform_job(const Fpc64ChainJob* jp)
  {
  Fpc64ChainJob job;
  job.pay = (const uint8_t*)rfl64((uint64_t)(uintptr_t)jp->pay);
  job.dst = (uint64_t*)rfl64((uint64_t)(uintptr_t)jp->dst);
  job.status = (uint32_t*)rfl64((uint64_t)(uintptr_t)jp->status);
  job.size = rfl(jp->size);
  job.n = rfl(jp->n);
  job.stride = rfl(jp->stride);
  job.pad = 0u;
  return job;
  }

// single stream: tables of component c at c * 2 * TSIZE entries, the rings behind all tables (launch_fpc64_decode)
__global__ void __launch_bounds__(128) k_fpc64_decode(ChainJobs3d args, u64* __restrict__ tables, uint32_t* __restrict__ scratch)
  {
  const uint32_t c = blockIdx.x;
  decode64_pair(uniform_job(&args.j[c]), tables + (size_t)c * 2 * TSIZE, scratch + SCR64_DWORDS * c);
  }

// batch: FPC64_DECODE_CHAIN_BYTES per chain = its two tables, then its ring
__global__ void __launch_bounds__(128) k_fpc64_decode_batch(const Fpc64ChainJob* __restrict__ jobs, uint8_t* __restrict__ chain_mem)
  {
  const uint32_t c = blockIdx.x;
  uint8_t* mem = (uint8_t*)rfl64((uint64_t)(uintptr_t)(chain_mem + (size_t)c * FPC64_DECODE_CHAIN_BYTES));
  decode64_pair(uniform_job(jobs + c), (u64*)mem, (uint32_t*)(mem + 2 * (size_t)TSIZE * 8));
  }

} // namespace

int launch_fpc64_encode(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes, uint64_t* d_tables)
  {
  hipLaunchKernelGGL(k_fpc64_encode, dim3(arity), dim3(64), 0, current_stream(),
                     (const u64*)d_src, n, arity, d_out, out_stride, d_sizes, (u64*)d_tables);
  return hip_ok(hipGetLastError(), "k_fpc64_encode") ? 1 : 0;
  }

// one chain per CU (see launch_fpc32_decode): the workgroup claims more than half of the CU's LDS
constexpr size_t CLAIM64 = 72u << 10;

int launch_fpc64_decode(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, uint32_t n, void* d_dst,
                        uint64_t* d_tables, uint32_t* d_status)
  {
  ChainJobs3d a;
  for (int c = 0; c < 3; ++c)
    a.j[c] = Fpc64ChainJob{ c < arity ? d_payloads[c] : nullptr, (uint64_t*)d_dst + c, d_status, c < arity ? sizes[c] : 0u, n, (uint32_t)arity, 0u };
  static const bool claimed = hipFuncSetAttribute((const void*)k_fpc64_decode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLAIM64) == hipSuccess;
  // the rings of the two waves live behind the tables: FPC64_DECODE_SCRATCH_BYTES per component
  uint32_t* scratch = (uint32_t*)(d_tables + (size_t)arity * 2 * TSIZE);
  hipLaunchKernelGGL(k_fpc64_decode, dim3(arity), dim3(128), claimed ? CLAIM64 : 0, current_stream(), a, (u64*)d_tables, scratch);
  return hip_ok(hipGetLastError(), "k_fpc64_decode") ? 1 : 0;
  }

int launch_fpc64_decode_batch(const Fpc64ChainJob* d_jobs, uint32_t njobs, uint8_t* d_tables)
  {
  if (njobs == 0)
    return 1;
  static_assert(FPC64_DECODE_CHAIN_BYTES == 2 * (size_t)TSIZE * 8 + SCR64_DWORDS * 4, "chain memory layout");
  static const bool claimed = hipFuncSetAttribute((const void*)k_fpc64_decode_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLAIM64) == hipSuccess;
  hipLaunchKernelGGL(k_fpc64_decode_batch, dim3(njobs), dim3(128), claimed ? CLAIM64 : 0, current_stream(), d_jobs, d_tables);
  return hip_ok(hipGetLastError(), "k_fpc64_decode_batch") ? 1 : 0;
  }

} // namespace trico
