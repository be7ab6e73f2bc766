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
//   decoder: compressed bytes staged through LDS; batches of 64 values: a scalar walk over the 32 header
//     bytes finds the group positions, all lanes fetch/align/byte-swap their residual, then the dependent
//     chain runs wave-uniform on the scalar unit with the tables behind the scalar data cache.  A table read is
//     skipped when the key did not change (then the entry is the value just written) — the common case on
//     smooth data — and only the table the value's code asks for is read; otherwise it is a dependent miss.
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

// Table accesses of the chain go through the SCALAR data cache (s_load / s_store; see k_fpc32_decode.hip for what was
// measured about them on gfx950): a scalar load costs ~6 cycles of issue against the ~16 of a vector load plus the
// VGPR -> SGPR hop, it leaves the result where the wave-uniform chain wants it, and a store is fire-and-forget.  Every
// access is preceded by s_waitcnt lgkmcnt(0), so a load is never in flight together with a store (a scalar load is not
// reliably ordered behind an earlier scalar store to the same address while that store is still in flight).
__device__ __forceinline__ u64 table_load(const u64* base, uint32_t byte_offset)
  {
  u64 r;
  asm volatile("s_waitcnt lgkmcnt(0)\n s_load_dwordx2 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(base), "s"(byte_offset) : "memory");
  return r;
  }
__device__ __forceinline__ void table_store(const u64* base, uint32_t byte_offset, u64 v)
  {
  asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(v), "s"(base), "s"(byte_offset) : "memory");
  }

// Two waves: wave 1 parses the group headers and the residuals of the next batches and stores the values of the finished ones;
// wave 0 runs the chain.  They are coupled through two LDS counters and two rings in the stream's scratch (residuals in,
// values out) that both waves access through the scalar cache (k_fpc32_decode.hip has the float version of the same design).
constexpr uint32_t RING64 = 4;                                   // batches the parser may run ahead
constexpr uint32_t SCR64_DWORDS = 2048;                          // scratch per component: RING64 x 512 B of residuals, then of values

// lane l's 64-bit word -> dwords 2l, 2l + 1 of `slot` (512 bytes), with scalar stores
__device__ __forceinline__ void put_words64(u64 w, const uint32_t* slot)
  {
  const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
#define P64_PUT2(J, R0, R1, R2, R3) \
  "v_readlane_b32 s" #R0 ", %[lo], 2 * (" #J ")\n v_readlane_b32 s" #R1 ", %[hi], 2 * (" #J ")\n" \
  "v_readlane_b32 s" #R2 ", %[lo], 2 * (" #J ") + 1\n v_readlane_b32 s" #R3 ", %[hi], 2 * (" #J ") + 1\n" \
  "s_nop 0\n s_store_dwordx4 s[" #R0 ":" #R3 "], %[slot], 16 * (" #J ")\n"
#define P64_4(J) P64_PUT2(J, 52, 53, 54, 55) P64_PUT2(J + 1, 56, 57, 58, 59) P64_PUT2(J + 2, 60, 61, 62, 63) P64_PUT2(J + 3, 64, 65, 66, 67)
  asm volatile(P64_4(0) P64_4(4) P64_4(8) P64_4(12) P64_4(16) P64_4(20) P64_4(24) P64_4(28)
               "s_waitcnt lgkmcnt(0)\n"
               :: [lo] "v"(lo), [hi] "v"(hi), [slot] "s"(slot)
               : "memory", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
  }

// dwords 2l, 2l + 1 of `slot` -> lane l, with scalar loads (the values were stored through the scalar cache)
__device__ __forceinline__ u64 get_words64(const uint32_t* slot)
  {
  uint32_t lo = 0, hi = 0;
#define G64_8(Q) \
  "s_load_dwordx16 s[52:67], %[slot], 64 * (" #Q ")\n s_waitcnt lgkmcnt(0)\n" \
  "v_writelane_b32 %[lo], s52, 8 * (" #Q ") + 0\n v_writelane_b32 %[hi], s53, 8 * (" #Q ") + 0\n" \
  "v_writelane_b32 %[lo], s54, 8 * (" #Q ") + 1\n v_writelane_b32 %[hi], s55, 8 * (" #Q ") + 1\n" \
  "v_writelane_b32 %[lo], s56, 8 * (" #Q ") + 2\n v_writelane_b32 %[hi], s57, 8 * (" #Q ") + 2\n" \
  "v_writelane_b32 %[lo], s58, 8 * (" #Q ") + 3\n v_writelane_b32 %[hi], s59, 8 * (" #Q ") + 3\n" \
  "v_writelane_b32 %[lo], s60, 8 * (" #Q ") + 4\n v_writelane_b32 %[hi], s61, 8 * (" #Q ") + 4\n" \
  "v_writelane_b32 %[lo], s62, 8 * (" #Q ") + 5\n v_writelane_b32 %[hi], s63, 8 * (" #Q ") + 5\n" \
  "v_writelane_b32 %[lo], s64, 8 * (" #Q ") + 6\n v_writelane_b32 %[hi], s65, 8 * (" #Q ") + 6\n" \
  "v_writelane_b32 %[lo], s66, 8 * (" #Q ") + 7\n v_writelane_b32 %[hi], s67, 8 * (" #Q ") + 7\n"
  asm volatile(G64_8(0) G64_8(1) G64_8(2) G64_8(3) G64_8(4) G64_8(5) G64_8(6) G64_8(7)
               : [lo] "+v"(lo), [hi] "+v"(hi) : [slot] "s"(slot)
               : "memory", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
  return ((u64)hi << 32) | lo;
  }

struct Oct64 { u64 v[8]; };
#define OCT_REGS "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83"
// eight 64-bit words at byte offset `off` of `slot`, through the scalar cache
__device__ __forceinline__ void load_oct(const uint32_t* slot, uint32_t off, Oct64& o)
  {
  asm volatile("s_load_dwordx16 s[68:83], %8, %9\n s_waitcnt lgkmcnt(0)\n"
               "s_mov_b64 %0, s[68:69]\n s_mov_b64 %1, s[70:71]\n s_mov_b64 %2, s[72:73]\n s_mov_b64 %3, s[74:75]\n"
               "s_mov_b64 %4, s[76:77]\n s_mov_b64 %5, s[78:79]\n s_mov_b64 %6, s[80:81]\n s_mov_b64 %7, s[82:83]"
               : "=&s"(o.v[0]), "=&s"(o.v[1]), "=&s"(o.v[2]), "=&s"(o.v[3]), "=&s"(o.v[4]), "=&s"(o.v[5]), "=&s"(o.v[6]), "=&s"(o.v[7])
               : "s"(slot), "s"(off) : "memory", OCT_REGS);
  }
__device__ __forceinline__ void store_oct(const uint32_t* slot, uint32_t off, const Oct64& o)
  {
  asm volatile("s_mov_b64 s[68:69], %0\n s_mov_b64 s[70:71], %1\n s_mov_b64 s[72:73], %2\n s_mov_b64 s[74:75], %3\n"
               "s_mov_b64 s[76:77], %4\n s_mov_b64 s[78:79], %5\n s_mov_b64 s[80:81], %6\n s_mov_b64 s[82:83], %7\n"
               "s_store_dwordx4 s[68:71], %8, %9\n s_store_dwordx4 s[72:75], %8, %10\n"
               "s_store_dwordx4 s[76:79], %8, %11\n s_store_dwordx4 s[80:83], %8, %12\n"
               "s_waitcnt lgkmcnt(0)"
               :: "s"(o.v[0]), "s"(o.v[1]), "s"(o.v[2]), "s"(o.v[3]), "s"(o.v[4]), "s"(o.v[5]), "s"(o.v[6]), "s"(o.v[7]),
                  "s"(slot), "s"(off), "s"(off + 16u), "s"(off + 32u), "s"(off + 48u) : "memory", OCT_REGS);
  }

__global__ void __launch_bounds__(128) k_fpc64_decode(DecodeArgs args, int arity, uint32_t n, u64* __restrict__ dst, u64* __restrict__ tables,
                                                      uint32_t* __restrict__ scratch, uint32_t* __restrict__ status)
  {
  __shared__ uint32_t win[WINW + 8];
  __shared__ uint32_t dmask[RING64][2];
  __shared__ uint32_t sh_bad, produced, consumed;
  const int lane = threadIdx.x & 63;
  const int wave = (int)rfl(threadIdx.x >> 6);
  const int comp = blockIdx.x;
  const uint8_t* in = args.pay[comp];
  if (threadIdx.x == 0)
    {
    sh_bad = 0u;
    produced = 0u;
    consumed = 0u;
    }
  const uint32_t len = args.size[comp];
  if (len < 5u)
    {
    if (threadIdx.x == 0) atomicOr(status, 1u);
    return;
    }
  const uint32_t e1 = (uint32_t)(in[0] >> 4) << 1, e2 = (uint32_t)(in[0] & 15) << 1;
  const uint32_t cnt = ((uint32_t)in[1] << 24) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 8) | in[4];
  if (cnt != n || e1 == 0u || e2 == 0u || e1 > E || e2 > E)
    {
    if (threadIdx.x == 0) atomicOr(status, 2u);
    return;
    }
  __syncthreads();
  const u64* T1 = tables + (size_t)comp * 2 * TSIZE;
  const u64* T2 = T1 + TSIZE;
  const uint32_t* xring = scratch + SCR64_DWORDS * (uint32_t)comp;        // RING64 slots of 64 residuals
  const uint32_t* oring = xring + 128u * RING64;                           // RING64 slots of 64 values
  const uint32_t nb = (n + 63u) / 64u;                                     // the last batch may be partial
  if (wave == 1)
    {
    // ---- parser: group headers and residuals of batch t, values of the finished batches to memory ----------------
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
    uint32_t t = 0, st = 0;
    while (st < nb)
      {
      const uint32_t cdone = rfl(__hip_atomic_load(&consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
      bool progress = false;
      for (; st < cdone; ++st)
        {
        const u64 v = get_words64(oring + 128u * (st % RING64));
        const uint32_t idx = 64u * st + (uint32_t)lane;
        if (idx < n)
          dst[(size_t)idx * arity + comp] = v;
        progress = true;
        }
      if (t < nb && t < st + RING64)
        {
        progress = true;
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
          if (lane == 0)
            __hip_atomic_store(&sh_bad, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        put_words64(xr, xring + 128u * (t % RING64));
        if (lane == 0)
          {
          dmask[t % RING64][0] = (uint32_t)dfcm;
          dmask[t % RING64][1] = (uint32_t)(dfcm >> 32);
          __hip_atomic_store(&produced, t + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        ++t;
        }
      if (!progress)
        __builtin_amdgcn_s_sleep(4);
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
    const u64 m1 = (1ull << e1) - 1ull, m2 = (1ull << e2) - 1ull;
    const uint32_t sh1 = 64u - e1, sh2 = 64u - e2, e2h = e2 >> 1;
    // fwd1 / fwd2: the hash did not change with the last value, so the entry of the current hash is the value / stride just
    // stored and is taken from the register (p1 / t2v) instead of being loaded
    uint32_t h1 = 0, h2 = 0;
    u64 p1 = 0, last = 0, t2v = 0;
    bool fwd1 = true, fwd2 = true;                        // zeroed tables: the entries of hash 0 are 0
    for (uint32_t t = 0; t < nb; ++t)
      {
      bool stop = false;
      while (rfl(__hip_atomic_load(&produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) <= t)
        {
        if (rfl(__hip_atomic_load(&sh_bad, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)))
          {
          stop = true;
          break;
          }
        __builtin_amdgcn_s_sleep(1);
        }
      if (stop)
        break;
      const uint32_t b = t % RING64;
      const uint64_t dfcm = ((uint64_t)rfl(dmask[b][1]) << 32) | rfl(dmask[b][0]);
      const uint32_t nvals = n - 64u * t < 64u ? n - 64u * t : 64u;
      // Per value ONE table entry is needed: the DFCM entry if the value is DFCM-coded, else the FCM entry
      // (fpsc.c:977-978), and only if the hash changed with the previous value; both tables are written for every value
      // (fpsc.c:980-995), fire and forget.  On noisy doubles that is one dependent miss into an 8 MiB table (Infinity
      // Cache, ~230 ns) for the DFCM-coded values and a scalar-cache / L2 hit for the others.
      for (uint32_t j = 0; 8u * j < nvals; ++j)
        {
        Oct64 xs, vs;
        load_oct(xring + 128u * b, 64u * j, xs);
#pragma unroll
        for (uint32_t k8 = 0; k8 < 8u; ++k8)
          {
          const uint32_t k = 8u * j + k8;
          u64 p;
          if ((dfcm >> k) & 1ull)
            {
            if (!fwd2)
              t2v = table_load(T2, h2 << 3);
            p = last + t2v;                                         // prediction2 = value + table entry
            }
          else
            {
            if (!fwd1)
              p1 = table_load(T1, h1 << 3);
            p = p1;
            }
          const u64 v = xs.v[k8] ^ p;
          const u64 s = v - last;
          // (the padding slots of a partial last batch run too: nothing reads the tables or the state after them)
          table_store(T1, h1 << 3, v);                              // hash_table_1[hash1] = value
          table_store(T2, h2 << 3, s);                              // hash_table_2[hash2] = stride
          const uint32_t nh1 = (uint32_t)((((u64)h1 << e1) ^ (v >> sh1)) & m1);
          const uint32_t nh2 = (uint32_t)((((u64)h2 << e2h) ^ (s >> sh2)) & m2);
          fwd1 = nh1 == h1;
          fwd2 = nh2 == h2;
          h1 = nh1;
          h2 = nh2;
          p1 = v;
          t2v = s;
          last = v;
          vs.v[k8] = v;
          }
        store_oct(oring + 128u * b, 64u * j, vs);
        }
      if (lane == 0)
        __hip_atomic_store(&consumed, t + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    // no dirty line of the scalar cache may outlive the table buffer
    asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  __syncthreads();
  if (sh_bad && threadIdx.x == 0)
    atomicOr(status, 4u);
  }

} // namespace

int launch_fpc64_encode(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes, uint64_t* d_tables)
  {
  hipLaunchKernelGGL(k_fpc64_encode, dim3(arity), dim3(64), 0, current_stream(),
                     (const u64*)d_src, n, arity, d_out, out_stride, d_sizes, (u64*)d_tables);
  return hip_ok(hipGetLastError(), "k_fpc64_encode") ? 1 : 0;
  }

int launch_fpc64_decode(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, uint32_t n, void* d_dst,
                        uint64_t* d_tables, uint32_t* d_status)
  {
  DecodeArgs a;
  for (int c = 0; c < 3; ++c)
    {
    a.pay[c] = c < arity ? d_payloads[c] : nullptr;
    a.size[c] = c < arity ? sizes[c] : 0;
    }
  // one chain per CU (see launch_fpc32_decode): the workgroup claims more than half of the CU's LDS
  constexpr size_t CLAIM = 72u << 10;
  static const bool claimed = hipFuncSetAttribute((const void*)k_fpc64_decode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLAIM) == hipSuccess;
  // the rings of the two waves live behind the tables: FPC64_DECODE_SCRATCH_BYTES per component
  uint32_t* scratch = (uint32_t*)(d_tables + (size_t)arity * 2 * TSIZE);
  hipLaunchKernelGGL(k_fpc64_decode, dim3(arity), dim3(128), claimed ? CLAIM : 0, current_stream(), a, arity, n, (u64*)d_dst, (u64*)d_tables,
                     scratch, d_status);
  return hip_ok(hipGetLastError(), "k_fpc64_decode") ? 1 : 0;
  }

} // namespace trico
