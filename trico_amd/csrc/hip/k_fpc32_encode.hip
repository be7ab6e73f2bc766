// k_fpc32_encode.hip — throughput encoder for 32-bit floating-point streams (gfx950, wave64).
//
// Replaces, fused: trico_transpose_xyz/uv_aos_to_soa (transpose_aos_to_soa.c:8-16, 48-56) and
// trico_compress(..., 4, 10) (fpsc.c:86-210) for every component of a vec3 / vec2 / scalar stream.
//
// Why this can be parallel and still bit-exact: the FCM hash of value
// i is a pure function of v[i-1] (top 4 bits) and the DFCM hash a pure function of v[i-3..i-1], so
// every value belongs to a *class* known from the input alone, and the reference's table read for
// value i returns the payload (value / stride) of the latest earlier value of the same class, or 0.
//
// Structure: each component stream is cut into S contiguous segments of L values (L % 64 == 0);
// one wave owns one (segment, component) and walks it 64 values per step.
//   sweep A  (k_fpc32_index):  classes only.  The run-end lane of every class run does an LDS
//            ds_max of its value index into a 16+1024 entry table -> "last writer index per class"
//            of the segment.
//   scan     (k_fpc32_scan_*): prefix-max over segments per class = the table every segment starts
//            with, as value indices (0 = never written = the reference's zeroed table).
//   sweep C  (k_fpc32_code):   loads the incoming table (payloads gathered from the input by index),
//            then per step: the latest earlier value of my class is the previous lane inside a run
//            of equal classes (runs span steps through the carry); the lanes where a run starts or ends
//            do ONE exchange on the wave-private payload table per predictor (resolve_xchg: the LDS unit
//            applies the lanes of an instruction in lane order, which is the reference's read-then-write,
//            value after value; tested on the device before use, ballots otherwise: resolve), only in
//            steps that have any (see the comment block above code_step).  Codes, residual lengths, wave
//            prefix sums (DPP scan), 3-byte group headers (DPP or-reduce), bytes staged in a linear LDS
//            buffer and flushed as aligned dwords into the segment's slot.
//   offsets  (k_fpc32_offsets): exclusive scan of the segment byte counts per component.
//   gather   (k_fpc32_gather): slot -> final position (this is the copy the reference does with
//            memcpy into the archive, trico.c:57-63; it targets the archive buffer directly).
//
// Tile variants of both sweeps (k_fpc32_index_t / k_fpc32_code_t, TRICO_FPC32_TILE): one wave walks all components of whole
// vertices, so the interleaved array is read once per sweep; less traffic, more time.
//
// HBM traffic: 2 x raw input + 2 x payload bytes + table traffic (measured 3.9 x algorithmic, DESIGN.md
// 4.1: the component waves drift apart and re-fetch lines).  No MFMA: integer bit-twiddling; algorithmic bytes per
// value = 4 + its payload share.  The bound is the vector ALU (80-100 instructions per 64-value step at 4 cycles
// each, 30 waves per CU), not HBM.
#include "common.hpp"
#include <stdlib.h>
#include <mutex>

namespace trico {

namespace {

constexpr int TAB = 1040;      // 16 FCM entries followed by 1024 DFCM entries
constexpr int ROW = 1040;      // words per (segment, component) row in the global index tables
constexpr int CH = 32;         // segments per chunk in the cross-segment scan
constexpr int LDSW_A = TAB;                        // per-wave LDS words, sweep A
constexpr int PF = 8;          // steps (of 64 values) whose loads are kept in flight per wave
#ifndef TRICO_PFC
#define TRICO_PFC 8
#endif
constexpr int PFC = TRICO_PFC; // ... in the two-sweep code sweep

__device__ __forceinline__ uint32_t dpp_shr1(uint32_t carry, uint32_t v)
  {
  // lane l <- lane l-1, lane 0 <- carry   (DPP wave_shr:1)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
  }

__device__ __forceinline__ uint32_t dpp_shl1(uint32_t carry, uint32_t v)
  {
  // lane l <- lane l+1, lane 63 <- carry   (DPP wave_shl:1)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x130, 0xf, 0xf, false);
  }

__device__ __forceinline__ uint32_t popc_below(uint64_t mask)
  {
  // number of set bits of `mask` strictly below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  }

__device__ __forceinline__ uint32_t blen(uint32_t x) { return (39u - (uint32_t)__clz((int)x)) >> 3; }

// residual selection (fpsc.c:146-189): returns code, sets len and the residual to emit
__device__ __forceinline__ uint32_t pick(uint32_t x1, uint32_t x2, uint32_t& len, uint32_t& x)
  {
  const uint32_t n1 = blen(x1);
  uint32_t n2 = blen(x2);
  n2 = n2 ? n2 : 1u;
  const bool use2 = (n1 > 1u) && (n2 < n1);
  len = use2 ? n2 : n1;
  x = use2 ? x2 : x1;
  return use2 ? 4u + n2 : n1;
  }

struct Carry { uint32_t m1, m2, m3; };

// loads PF steps of this wave's component starting at value index i0 (0 beyond i_end)
template <int P>
__device__ __forceinline__ void load_block(uint32_t (&r)[P], const uint32_t* __restrict__ src, uint32_t i0, uint32_t i_end,
                                           int arity, int c, int lane)
  {
#pragma unroll
  for (int pu = 0; pu < P; ++pu)
    {
    const uint32_t i = i0 + 64u * pu + lane;
    r[pu] = (i0 < i_end && i < i_end) ? src[(size_t)i * arity + c] : 0u;
    }
  }

__device__ __forceinline__ Carry load_carry(const uint32_t* __restrict__ src, uint32_t i_begin, int arity, int c)
  {
  Carry k;
  k.m1 = i_begin >= 1u ? src[(size_t)(i_begin - 1u) * arity + c] : 0u;
  k.m2 = i_begin >= 2u ? src[(size_t)(i_begin - 2u) * arity + c] : 0u;
  k.m3 = i_begin >= 3u ? src[(size_t)(i_begin - 3u) * arity + c] : 0u;
  return k;
  }

// classes of the 64 values of a step: k1 in [0,16) (FCM), k2 in [16,1040) (DFCM); a = v[i-1], b = v[i-2]
__device__ __forceinline__ void classes(uint32_t v, const Carry& cy, bool act, uint32_t& a, uint32_t& b, uint32_t& k1, uint32_t& k2)
  {
  a = dpp_shr1(cy.m1, v);
  b = dpp_shr1(cy.m2, a);
  const uint32_t d = dpp_shr1(cy.m3, b);     // v[i-3]
  const uint32_t s1 = a - b, s2 = b - d;     // strides of values i-1, i-2
  k1 = a >> 28;                                                    // fpsc.c:76-79 with e1 = 4
  k2 = 16u + ((((s2 >> 22) & 31u) << 5) ^ (s1 >> 22));            // fpsc.c:81-84 with e2 = 10
  if (!act)
    k1 = k2 = 0xffffffffu;
  }

__device__ __forceinline__ void next_carry(Carry& cy, uint32_t v)
  {
  cy.m1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  cy.m2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 62);
  cy.m3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 61);
  }

// ---- cooperative AoS staging ----------------------------------------------------------------------------
// The component waves of a workgroup walk the same vertices.  If each of them loads its own component straight
// from the interleaved array, every cache line is requested once per component, and the per-CU miss queue (not
// HBM) limits the sweep to ~2.5 TB/s.  Instead the workgroup fetches a block of 64 * PF vertices with coalesced
// loads (16 bytes per lane when the array is 16-byte aligned), parks it in LDS, and every wave picks its
// component from there (stride `arity` dwords: conflict-free for arity 1..3).  The next block is in flight in
// registers while the current one is processed.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int BLOCK_V = 64 * PF;               // vertices per staged block

struct BlockRegs { u32x4 q[2]; };              // 8 dwords per thread = 64 * PF * arity dwords per workgroup

template <bool X4>
__device__ __forceinline__ void block_fetch(BlockRegs& r, const uint32_t* __restrict__ src, uint64_t first_dword, uint64_t total_dwords,
                                            uint32_t threads, uint32_t tid)
  {
#pragma unroll
  for (int j = 0; j < 2; ++j)
    {
    u32x4 q = { 0u, 0u, 0u, 0u };
    if (X4)
      {
      const uint64_t e = first_dword + 4ull * ((uint64_t)j * threads + tid);
      if (e + 4ull <= total_dwords)
        q = *(const u32x4*)(src + e);
      else
        for (int k = 0; k < 4; ++k)
          if (e + (uint64_t)k < total_dwords)
            q[k] = src[e + (uint64_t)k];
      }
    else
      for (int k = 0; k < 4; ++k)
        {
        const uint64_t e = first_dword + ((uint64_t)(4 * j + k) * threads + tid);
        if (e < total_dwords)
          q[k] = src[e];
        }
    r.q[j] = q;
    }
  }

template <bool X4>
__device__ __forceinline__ void block_park(const BlockRegs& r, uint32_t* __restrict__ stage, uint32_t threads, uint32_t tid)
  {
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (X4)
      *(u32x4*)(stage + 4u * ((uint32_t)j * threads + tid)) = r.q[j];
    else
      for (int k = 0; k < 4; ++k)
        stage[(uint32_t)(4 * j + k) * threads + tid] = r.q[j][k];
  }

// ---- sweep A: last writer index (+1) per class of every segment ---------------------------------------
// "Last writer" is a maximum over value indices, so a segment may be swept by several waves at once:
// every segment is cut into ISPLIT sub-ranges (grid.y), each with its own LDS table, combined into the
// segment's row with global atomicMax (the rows are zeroed before the launch).
constexpr uint32_t ISPLIT = 2;

template <bool X4>
__global__ void __launch_bounds__(192) k_fpc32_index(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L,
                                                     uint32_t* __restrict__ summ)
  {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x;
  uint32_t* stage = lds;                                 // [BLOCK_V * arity]
  uint32_t* T = lds + BLOCK_V * arity + c * LDSW_A;
  for (int i = lane; i < TAB; i += 64)
    T[i] = 0u;
  const uint32_t sub = L / ISPLIT;                       // L is a multiple of 64 * ISPLIT
  const uint32_t seg_end = (n - g * L < L) ? n : g * L + L;
  const uint32_t i_begin = g * L + blockIdx.y * sub;
  if (i_begin >= seg_end)
    return;
  const uint32_t i_end = (seg_end - i_begin < sub) ? seg_end : i_begin + sub;
  const uint64_t total = (uint64_t)n * (uint64_t)arity;
  const uint32_t threads = blockDim.x, tid = threadIdx.x;
  Carry cy = load_carry(src, i_begin, arity, c);
  BlockRegs regs;
  block_fetch<X4>(regs, src, (uint64_t)i_begin * arity, total, threads, tid);
  block_park<X4>(regs, stage, threads, tid);
  __syncthreads();
  for (uint32_t ib = i_begin; ib < i_end; ib += BLOCK_V)
    {
    const bool more = ib + BLOCK_V < i_end;
    if (more)
      block_fetch<X4>(regs, src, (uint64_t)(ib + BLOCK_V) * arity, total, threads, tid);
    // the wave's values of the whole block leave the staging area together (one LDS round trip per block, not per step)
    uint32_t vv[PF];
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      vv[pu] = stage[(uint32_t)(64 * pu + lane) * (uint32_t)arity + (uint32_t)c];
    if (ib + BLOCK_V <= i_end)
      {
      // every value of the block is inside the range: no activity masks
#pragma unroll
      for (int pu = 0; pu < PF; ++pu)
        {
        const uint32_t i = ib + 64u * pu + lane;
        const uint32_t v = vv[pu];
        uint32_t a, b, k1, k2;
        classes(v, cy, true, a, b, k1, k2);
        // only the last lane of a run of equal classes can be the class's last writer in this step
        const uint32_t kn1 = dpp_shl1(0xfffffffeu, k1), kn2 = dpp_shl1(0xfffffffeu, k2);
        if (k1 != kn1) atomicMax(&T[k1], i + 1u);
        if (k2 != kn2) atomicMax(&T[k2], i + 1u);
        next_carry(cy, v);
        }
      }
    else
      {
#pragma unroll 1
      for (int pu = 0; pu < PF; ++pu)
        {
        const uint32_t i0 = ib + 64u * pu;
        if (i0 >= i_end)
          break;
        const uint32_t i = i0 + lane;
        const bool act = i < i_end;
        uint32_t v = vv[0];
#pragma unroll
        for (int q = 1; q < PF; ++q)
          v = pu == q ? vv[q] : v;
        uint32_t a, b, k1, k2;
        classes(v, cy, act, a, b, k1, k2);
        const uint32_t kn1 = dpp_shl1(0xfffffffeu, k1), kn2 = dpp_shl1(0xfffffffeu, k2);
        if (act && k1 != kn1) atomicMax(&T[k1], i + 1u);
        if (act && k2 != kn2) atomicMax(&T[k2], i + 1u);
        next_carry(cy, v);
        }
      }
    if (more)
      {
      __syncthreads();                                   // everybody is done with the parked block
      block_park<X4>(regs, stage, threads, tid);
      __syncthreads();
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t* row = summ + ((size_t)g * arity + c) * ROW;
  for (int i = lane; i < TAB; i += 64)
    {
    const uint32_t t = T[i];
    if (t)
      atomicMax(&row[i], t);
    }
  }

// ---- tile variant of sweep A: one wave owns a segment with ALL its components ---------------------------------------------
// Lane l loads vertex i0 + l whole (A consecutive dwords: the 64 lanes of a load cover 64 * A * 4 contiguous bytes), so every
// cache line of the interleaved array is requested once per sweep, by one wave, with no staging through LDS and no barrier.
// The wave walks the components of a tile one after the other; their tables sit side by side in its LDS.  No sub-ranges
// (the whole GPU is covered by S one-wave workgroups), so the rows are written directly and need no zeroing.
template <int A> struct __attribute__((packed, aligned(4))) VertexT { uint32_t w[A]; };
constexpr int PFT = 2;         // tiles (of 64 vertices) whose loads are kept in flight beside the block being worked on

constexpr int PFI = 8;         // ... in sweep A, which has nothing but its loads to wait for

template <int A, int P>
__device__ __forceinline__ void load_tiles(uint32_t (&r)[P][A], const uint32_t* __restrict__ src, uint32_t i0, uint32_t i_end, int lane)
  {
#pragma unroll
  for (int pu = 0; pu < P; ++pu)
    {
    const uint32_t i = i0 + 64u * pu + lane;
    VertexT<A> t = {};
    if (i0 < i_end && i < i_end)
      t = *(const VertexT<A>*)(src + (size_t)i * A);
#pragma unroll
    for (int c = 0; c < A; ++c)
      r[pu][c] = t.w[c];
    }
  }

template <int A>
__global__ void __launch_bounds__(64) k_fpc32_index_t(const uint32_t* __restrict__ src, uint32_t n, uint32_t L, uint32_t* __restrict__ summ)
  {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  const uint32_t g = blockIdx.x;
  for (int i = lane; i < A * TAB; i += 64)
    lds[i] = 0u;
  const uint32_t i_begin = g * L;
  const uint32_t i_end = (n - i_begin < L) ? n : i_begin + L;
  Carry cy[A];
#pragma unroll
  for (int c = 0; c < A; ++c)
    cy[c] = load_carry(src, i_begin, A, c);
  uint32_t cur[PFI][A], nxt[PFI][A];
  load_tiles<A, PFI>(cur, src, i_begin, i_end, lane);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (uint32_t ib = i_begin; ib < i_end; ib += 64u * PFI)
    {
    load_tiles<A, PFI>(nxt, src, ib + 64u * PFI, i_end, lane);
#pragma unroll
    for (int pu = 0; pu < PFI; ++pu)
      {
      const uint32_t i0 = ib + 64u * pu;
      if (i0 < i_end)
        {
        const uint32_t i = i0 + lane;
        const bool act = i < i_end;
#pragma unroll
        for (int c = 0; c < A; ++c)
          {
          uint32_t* T = lds + c * TAB;
          const uint32_t v = cur[pu][c];
          uint32_t a, b, k1, k2;
          classes(v, cy[c], act, a, b, k1, k2);
          const uint32_t kn1 = dpp_shl1(0xfffffffeu, k1), kn2 = dpp_shl1(0xfffffffeu, k2);
          if (act && k1 != kn1) atomicMax(&T[k1], i + 1u);
          if (act && k2 != kn2) atomicMax(&T[k2], i + 1u);
          next_carry(cy[c], v);
          }
        }
      }
#pragma unroll
    for (int pu = 0; pu < PFI; ++pu)
#pragma unroll
      for (int c = 0; c < A; ++c)
        cur[pu][c] = nxt[pu][c];
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t* row = summ + (size_t)g * A * ROW;
#pragma unroll
  for (int c = 0; c < A; ++c)
    for (int i = lane; i < TAB; i += 64)
      row[c * ROW + i] = lds[c * TAB + i];
  }

// ---- scan: incoming index table of segment g = max over earlier segments -----------------------------
__global__ void __launch_bounds__(256) k_fpc32_scan_a(const uint32_t* __restrict__ summ, uint32_t S, int arity,
                                                      uint32_t* __restrict__ chmax, uint32_t* __restrict__ flags, uint32_t* __restrict__ nrec)
  {
  if (blockIdx.x == 0 && blockIdx.y == 0)
    {
    if (threadIdx.x == 0)
      flags[0] = 0u;                                         // the code sweep's "LDS order violated" word (a memset of 4 bytes costs 7 us)
    for (uint32_t r = threadIdx.x; r < S * (uint32_t)arity; r += 256u)
      nrec[r] = 0u;                                          // two sweeps: no deferred values for the gather to skip around
    }
  const uint32_t col = blockIdx.x * 256u + threadIdx.x;      // (component, class)
  const uint32_t ncol = (uint32_t)arity * TAB;
  if (col >= ncol)
    return;
  const uint32_t c = col / TAB, k = col % TAB;
  const uint32_t g0 = blockIdx.y * CH, g1 = (g0 + CH < S) ? g0 + CH : S;
  uint32_t m = 0;
#pragma unroll 8
  for (uint32_t g = g0; g < g1; ++g)
    m = max(m, summ[((size_t)g * arity + c) * ROW + k]);
  chmax[(size_t)blockIdx.y * ncol + col] = m;
  }

__global__ void __launch_bounds__(256) k_fpc32_scan_b(const uint32_t* __restrict__ summ, uint32_t S, int arity,
                                                      const uint32_t* __restrict__ chmax, uint32_t* __restrict__ inc)
  {
  const uint32_t col = blockIdx.x * 256u + threadIdx.x;
  const uint32_t ncol = (uint32_t)arity * TAB;
  if (col >= ncol)
    return;
  const uint32_t c = col / TAB, k = col % TAB;
  uint32_t carry = 0;
  for (uint32_t j = 0; j < blockIdx.y; ++j)
    carry = max(carry, chmax[(size_t)j * ncol + col]);
  const uint32_t g0 = blockIdx.y * CH, g1 = (g0 + CH < S) ? g0 + CH : S;
#pragma unroll 8
  for (uint32_t g = g0; g < g1; ++g)
    {
    const size_t r = ((size_t)g * arity + c) * ROW + k;
    inc[r] = carry;
    carry = max(carry, summ[r]);
    }
  }

// ---- sweep C -------------------------------------------------------------------------------------------
//
// Instruction count is what bounds this sweep (a wave64 VALU instruction occupies its 16-lane SIMD for 4
// cycles; at ~100+ instructions per 64 values the VALU time exceeds the HBM time by far), so the step is
// organised around doing nothing that the data does not ask for:
//   * lane 0 compares its class with the class of the previous step's last value (carried in SGPRs), so a
//     run of equal classes can span any number of steps.  A step without a run start in a predictor needs
//     no table access for it at all: every value is predicted from the previous lane (FCM: previous value,
//     DFCM: previous stride), and the table write of the step's last value stays *pending* in the carry;
//   * only when a predictor has a run start in the step, its pending write is flushed and the run starts
//     are resolved (mask table M + payload table T, below), for that predictor alone or for both at once
//     so that their LDS round trips overlap;
//   * the packed bytes of a step go to a linear LDS staging area, so byte addresses are base + immediate;
//     a residual is stored as the 4 big-endian bytes that END at its last byte: the leading zero bytes
//     land on bytes of earlier lanes / headers of the same step, all of which are written by a later
//     instruction (proof in DESIGN.md), so there is no per-byte predicate;
//   * full steps (64 values, the normal case) are a separate instantiation without activity masks.
//
// Run starts: lanes form runs of equal class.  A run START needs the nearest lower lane of its class, which
// is the END lane of an earlier run.  Every lane gets the set of lanes of its class from one ballot per class
// bit (match_any: 4 ballots for the FCM class, 10 for the DFCM class; round 1 kept a u64 lane mask per class in
// LDS, 8.3 KB per wave, which held the sweep at 12 waves per CU and three dependent LDS round trips per step).
// The nearest lower set bit is the source lane (ds_bpermute); no bit below means the table T holds the latest
// earlier value (from an earlier step or the segment's incoming table).  The highest lane of a class owns the
// table write.

constexpr int STAGE_LIVE = 544;                   // < 256 unflushed + <= 280 of the step, rounded
constexpr int STAGE = STAGE_LIVE + 256;           // + 4 dump bytes per lane
constexpr int LDSW_C = TAB + STAGE / 4;           // per-wave LDS words, sweep C (4,960 B: 10 x 3 waves per CU)
constexpr int LDSW_CA = 2 * TAB + STAGE / 4;      // ... of the variant whose table entries carry a tag (9,120 B: 5 x 3 waves per CU)

struct LaneK                                      // per-lane constants
  {
  int lane;
  uint64_t lt, bit;
  uint32_t sh3, grp3, dumpw, dumpq;
  bool lead;
  };

struct Sweep                                      // wave-uniform running state
  {
  Carry cy;
  uint32_t kc1, kc2;                              // classes of the previous step's last value
  bool pend1, pend2;                              // its table writes are still pending
  uint32_t posl, flushed;                         // bytes staged in LDS / bytes already in the slot
  uint32_t tag;                                   // resolve_atomic: number of the resolving step, << 6
  uint32_t viol;                                  // resolve_atomic: the LDS unit applied an atomic out of lane order
  uint32_t fl_nb, fl_off;                         // flush in flight: 256-byte blocks (0 = none) and their offset in the slot
  uint32_t fw0, fw1, ft;                          // ... per lane: its words of the blocks and of what moves to the front
  };

// second half of a flush (see the end of code_step): the words read from the staging area a step ago go to the slot, the
// unflushed rest moves to the front.  Must run before the next byte is staged.
__device__ __forceinline__ void flush_end(Sweep& sw, uint8_t* __restrict__ stage, uint8_t* __restrict__ gbase, const LaneK& lk)
  {
  if (sw.fl_nb)
    {
    // (streaming stores: the slot is read again only by the gather, and the lines should not push the input out of the L2)
    __builtin_nontemporal_store(sw.fw0, (uint32_t*)(gbase + sw.fl_off + 4u * (uint32_t)lk.lane));
    if (sw.fl_nb == 2u)
      __builtin_nontemporal_store(sw.fw1, (uint32_t*)(gbase + sw.fl_off + 256u + 4u * (uint32_t)lk.lane));
    ((uint32_t*)stage)[lk.lane] = sw.ft;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    sw.fl_nb = 0u;
    }
  }

// lanes whose class equals mine, for B-bit classes (B even): one ballot per class bit.  `diff` collects the lanes that differ
// from me in some bit: diff |= ballot(bit) ^ mybit, one v_bitop3 per half; the caller takes live & ~diff.  Four vector
// instructions per bit, written out because the kernel is bound by the vector ALU (hipcc spent six); two bits per
// statement so that the wait states between a v_cmp writing an SGPR pair and the v_bitop3 reading it are filled.
template <int B0>
__device__ __forceinline__ void match_bits2(uint32_t k, uint32_t& dlo, uint32_t& dhi)
  {
  uint32_t p0, p1;
  asm volatile("v_bfe_i32 %[p0], %[k], %[b0], 1\n"
               "v_bfe_i32 %[p1], %[k], %[b1], 1\n"
               "v_cmp_ne_u32_e64 s[40:41], 0, %[p0]\n"
               "v_cmp_ne_u32_e64 s[42:43], 0, %[p1]\n"
               "s_nop 0\n"
               "v_bitop3_b32 %[dlo], %[dlo], s40, %[p0] bitop3:0xf6\n"       // dlo | (ballot ^ p0)
               "v_bitop3_b32 %[dhi], %[dhi], s41, %[p0] bitop3:0xf6\n"
               "v_bitop3_b32 %[dlo], %[dlo], s42, %[p1] bitop3:0xf6\n"
               "v_bitop3_b32 %[dhi], %[dhi], s43, %[p1] bitop3:0xf6\n"
               : [dlo] "+v"(dlo), [dhi] "+v"(dhi), [p0] "=&v"(p0), [p1] "=&v"(p1)
               : [k] "v"(k), [b0] "n"(B0), [b1] "n"(B0 + 1)
               : "s40", "s41", "s42", "s43");
  }

template <int B>
__device__ __forceinline__ void match_any(uint32_t k, uint32_t& same_lo, uint32_t& same_hi)
  {
  uint32_t dlo = 0, dhi = 0;
  match_bits2<0>(k, dlo, dhi);
  match_bits2<2>(k, dlo, dhi);
  if (B > 4)
    {
    match_bits2<4>(k, dlo, dhi);
    match_bits2<6>(k, dlo, dhi);
    match_bits2<8>(k, dlo, dhi);
    }
  same_lo &= ~dlo;
  same_hi &= ~dhi;
  }

template <bool FULL, bool D1, bool D2>
__device__ __forceinline__ void resolve(uint32_t k1, uint32_t k2, bool st1, bool st2, bool act, uint32_t v, uint32_t s,
                                        uint32_t& p1, uint32_t& p2, uint32_t* __restrict__ T, Sweep& sw, const LaneK& lk)
  {
  // pending writes of the previous step's last value (lane 0 writes, the others hit their dump word)
  if (D1 && sw.pend1) T[lk.lane == 0 ? sw.kc1 : lk.dumpw] = sw.cy.m1;
  if (D2 && sw.pend2) T[lk.lane == 0 ? sw.kc2 : lk.dumpw] = sw.cy.m1 - sw.cy.m2;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // table entries of the run starts that have no earlier value of their class in this step
  const uint32_t r1 = FULL ? k1 : (act ? k1 : 0u), r2 = FULL ? k2 : (act ? k2 : 16u);
  uint32_t tv1 = 0, tv2 = 0;
  if (D1) tv1 = T[r1];
  if (D2) tv2 = T[r2];
  // the lanes of my class (no LDS: ballots), while those reads are in flight
  const uint64_t live = FULL ? ~0ull : __ballot(act);
  uint32_t s1lo = (uint32_t)live, s1hi = (uint32_t)(live >> 32), s2lo = s1lo, s2hi = s1hi;
  if (D1) match_any<4>(k1, s1lo, s1hi);
  if (D2) match_any<10>(k2 - 16u, s2lo, s2hi);
  const uint64_t same1 = ((uint64_t)s1hi << 32) | s1lo, same2 = ((uint64_t)s2hi << 32) | s2lo;
  // a run START takes the payload of the nearest lower lane of its class (the END of an earlier run), else the table's
  const uint64_t lo1 = same1 & lk.lt, lo2 = same2 & lk.lt;
  const bool hit1 = D1 && st1 && lo1 != 0ull, hit2 = D2 && st2 && lo2 != 0ull;
  if (D1) p1 = st1 ? tv1 : p1;
  if (D2) p2 = st2 ? tv2 : p2;
  if (__ballot(hit1 || hit2))
    {
    if (D1)
      {
      const uint32_t q = (uint32_t)__builtin_amdgcn_ds_bpermute((63 - __builtin_clzll(lo1 | 1ull)) << 2, (int)v);
      p1 = hit1 ? q : p1;
      }
    if (D2)
      {
      const uint32_t q = (uint32_t)__builtin_amdgcn_ds_bpermute((63 - __builtin_clzll(lo2 | 1ull)) << 2, (int)s);
      p2 = hit2 ? q : p2;
      }
    }
  // table writes by the highest lane of every class (after the reads: LDS operations of a wave stay in order)
  const bool own1 = (FULL || act) && (same1 >> lk.lane) == 1ull, own2 = (FULL || act) && (same2 >> lk.lane) == 1ull;
  if (D1) T[own1 ? k1 : lk.dumpw] = v;
  if (D2) T[own2 ? k2 : lk.dumpw] = s;
  if (D1) { sw.kc1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 63); sw.pend1 = false; }
  if (D2) { sw.kc2 = (uint32_t)__builtin_amdgcn_readlane((int)k2, 63); sw.pend2 = false; }
  }

// The same lookup as ONE LDS instruction per predictor.  Table entries are 64 bits: {tag, payload}, tag = (number of the resolving
// step << 6 | lane) of the value that wrote the entry (0 for what the segment came in with).  Every run END does
// ds_max_rtn_u64(entry of its class, {its tag, its payload}); a run START that is not an end does the same with {0, 0}, which changes
// nothing.  What comes back is the entry as it was when the lane's turn came: on gfx950 the LDS unit applies the lanes of one atomic
// instruction in increasing lane order (tools/ubench/lds_atomic_order.hip: 7.7 M instructions, key sets from 1 to 1024 keys, not one
// lane out of order), so a start lane gets the payload of the nearest lower lane of its class if this step has one - necessarily a
// run end - and otherwise what earlier steps left, i.e. exactly what the reference's table holds when it codes the value
// (fpsc.c:133-143).  The table is up to date afterwards too: the maximum is the highest lane's entry.  The order is not documented,
// so it is CHECKED in every step: in any other order some lane sees a tag of this step that is not below its own; such a step
// raises `viol`, the host throws the encode away and repeats it with the ballot kernel (never seen to happen).
template <bool FULL, bool D1, bool D2>
__device__ __forceinline__ void resolve_atomic(uint32_t k1, uint32_t k2, bool st1, bool st2, bool act, uint32_t v, uint32_t s,
                                               uint32_t& p1, uint32_t& p2, unsigned long long* __restrict__ T, Sweep& sw, const LaneK& lk)
  {
  sw.tag += 64u;
  const uint32_t mytag = sw.tag | (uint32_t)lk.lane;
  // pending writes of the previous step's last value: it is later than everything the table holds and earlier than this step
  if (lk.lane == 0)
    {
    if (D1 && sw.pend1) T[sw.kc1] = ((unsigned long long)(sw.tag - 1u) << 32) | sw.cy.m1;
    if (D2 && sw.pend2) T[sw.kc2] = ((unsigned long long)(sw.tag - 1u) << 32) | (uint32_t)(sw.cy.m1 - sw.cy.m2);
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  bool bad = false;
  if (D1)
    {
    const bool en = (FULL || act) && k1 != dpp_shl1(0xfffffffeu, k1);           // last lane of a run (lane 63 always)
    if (st1 || en)
      {
      const unsigned long long mine = en ? (((unsigned long long)mytag << 32) | v) : 0ull;
      const unsigned long long old = atomicMax(&T[k1], mine);
      const uint32_t ot = (uint32_t)(old >> 32);
      bad = (ot >> 6) == (sw.tag >> 6) && (ot & 63u) >= (uint32_t)lk.lane;
      p1 = st1 ? (uint32_t)old : p1;
      }
    }
  if (D2)
    {
    const bool en = (FULL || act) && k2 != dpp_shl1(0xfffffffeu, k2);
    if (st2 || en)
      {
      const unsigned long long mine = en ? (((unsigned long long)mytag << 32) | s) : 0ull;
      const unsigned long long old = atomicMax(&T[k2], mine);
      const uint32_t ot = (uint32_t)(old >> 32);
      bad = bad || ((ot >> 6) == (sw.tag >> 6) && (ot & 63u) >= (uint32_t)lk.lane);
      p2 = st2 ? (uint32_t)old : p2;
      }
    }
  if (__ballot(bad))
    sw.viol = 1u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (D1) { sw.kc1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 63); sw.pend1 = false; }
  if (D2) { sw.kc2 = (uint32_t)__builtin_amdgcn_readlane((int)k2, 63); sw.pend2 = false; }
  }

// The same lookup with 32-bit entries and NO tags: ds_wrxchg_rtn_b32.  The reference codes a value by reading the entry of its class
// and then writing its own payload there (fpsc.c:133-143), value after value; an exchange is exactly that pair, and the LDS unit of
// gfx950 applies the active lanes of one exchange instruction in increasing lane order (tools/ubench/lds_xchg_order.hip: 131 M
// instructions, 1 to 1024 keys, random exec masks, four waves per workgroup on the same LDS: every lane got the payload of the nearest
// lower active lane of its key, or what earlier steps had left, and the entry ended with the highest lane's payload).  Only the lanes
// where a run of equal classes starts or ends take part (inside a run the previous lane is the predecessor, DPP): a start takes what
// comes back; a start that is not an end leaves its payload there for a moment, and the end lane of its run - a higher lane of the
// same instruction - replaces it.  Ten ballots, a table read, a ds_bpermute and a table write become one LDS instruction, and the
// table stays at 4 bytes per entry (30 waves per CU, unlike resolve_atomic).  The order is not documented, so the library tests it on
// the device before the first encode (fpc32_xchg_usable(), below) and falls back to the ballot kernel if the test fails; the decoders'
// self-check always re-encodes with the ballot kernel, which does not depend on it.
template <bool FULL, bool D1, bool D2>
__device__ __forceinline__ void resolve_xchg(uint32_t k1, uint32_t k2, bool st1, bool st2, bool act, uint32_t v, uint32_t s,
                                             uint32_t& p1, uint32_t& p2, uint32_t* __restrict__ T, Sweep& sw, const LaneK& lk)
  {
  // pending writes of the previous step's last value
  if (lk.lane == 0)
    {
    if (D1 && sw.pend1) T[sw.kc1] = sw.cy.m1;
    if (D2 && sw.pend2) T[sw.kc2] = sw.cy.m1 - sw.cy.m2;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (D1)
    {
    const bool en = (FULL || act) && k1 != dpp_shl1(0xfffffffeu, k1);           // last lane of a run (lane 63 always)
    if (st1 || en)
      {
      const uint32_t old = __hip_atomic_exchange(&T[k1], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      p1 = st1 ? old : p1;
      }
    }
  if (D2)
    {
    const bool en = (FULL || act) && k2 != dpp_shl1(0xfffffffeu, k2);
    if (st2 || en)
      {
      const uint32_t old = __hip_atomic_exchange(&T[k2], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      p2 = st2 ? old : p2;
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (D1) { sw.kc1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 63); sw.pend1 = false; }
  if (D2) { sw.kc2 = (uint32_t)__builtin_amdgcn_readlane((int)k2, 63); sw.pend2 = false; }
  }

// resolve_xchg for the one-sweep encoder (k_fpc32_sweep1): the segment's incoming table is not known while it is coded.  The table
// starts filled with a sentinel and a bitmap of the classes written so far; a run start that gets the sentinel back for a class
// not in the bitmap has met the incoming entry (ft = first touch): its value is coded later (k_fpc32_fixup).  A payload that
// happens to equal the sentinel is told apart by the bitmap.
constexpr uint32_t SENT = 0x7fc0dead;

template <bool FULL, bool D1, bool D2>
__device__ __forceinline__ void resolve_xchg_h(uint32_t k1, uint32_t k2, bool st1, bool st2, bool act, uint32_t v, uint32_t s,
                                               uint32_t& p1, uint32_t& p2, bool& ft1, bool& ft2, uint32_t* __restrict__ T,
                                               uint32_t* __restrict__ seen, Sweep& sw, const LaneK& lk)
  {
  if (lk.lane == 0)
    {
    if (D1 && sw.pend1) T[sw.kc1] = sw.cy.m1;
    if (D2 && sw.pend2) T[sw.kc2] = sw.cy.m1 - sw.cy.m2;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  bool c1 = false, c2 = false;
  if (D1)
    {
    const bool en = (FULL || act) && k1 != dpp_shl1(0xfffffffeu, k1);
    if (st1 || en)
      {
      const uint32_t old = __hip_atomic_exchange(&T[k1], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      p1 = st1 ? old : p1;
      c1 = st1 && old == SENT;
      }
    }
  if (D2)
    {
    const bool en = (FULL || act) && k2 != dpp_shl1(0xfffffffeu, k2);
    if (st2 || en)
      {
      const uint32_t old = __hip_atomic_exchange(&T[k2], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      p2 = st2 ? old : p2;
      c2 = st2 && old == SENT;
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (__ballot(c1 || c2))
    {
    // lane order again: of two starts of one class in this step the lower one finds the bit clear
    if (c1)
      {
      const uint32_t bit = 1u << (k1 & 31u);
      ft1 = (__hip_atomic_fetch_or(&seen[k1 >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) & bit) == 0u;
      }
    if (c2)
      {
      const uint32_t bit = 1u << (k2 & 31u);
      ft2 = (__hip_atomic_fetch_or(&seen[k2 >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) & bit) == 0u;
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  if (D1) { sw.kc1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 63); sw.pend1 = false; }
  if (D2) { sw.kc2 = (uint32_t)__builtin_amdgcn_readlane((int)k2, 63); sw.pend2 = false; }
  }

// store the bytes of staged word `w` (byte offset off inside the slot, multiple of 4) that lie below hi
__device__ __forceinline__ void store_span(uint8_t* __restrict__ gbase, uint32_t off, uint32_t w, uint32_t hi)
  {
  if (off + 4u <= hi)
    *(uint32_t*)(gbase + off) = w;
  else
    for (uint32_t bb = 0; bb < 4u; ++bb)
      if (off + bb < hi)
        gbase[off + bb] = (uint8_t)(w >> (8u * bb));
  }

// inclusive prefix sum over the wave (DPP: four steps inside the rows of 16 lanes, two row broadcasts)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
  {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);     // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);     // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);     // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);     // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);     // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);     // row_bcast:31 into rows 2 and 3
  return x;
  }

// one step: 64 values starting at index i0 (FULL: all of them inside the segment)
// MODE: how run starts find their predecessor - 0 ballots (resolve), 1 tagged 64-bit entries (resolve_atomic), 2 exchange (resolve_xchg)
constexpr int M_BALLOT = 0, M_TAGGED = 1, M_XCHG = 2;

// The step in phases, so that a wave that walks several components (tile_step) can run the same phase of all of them back to back:
// the phases of different components are independent instruction streams inside one basic block, and each covers the others' waits.
struct StepRegs                                  // per-lane values a step carries from phase to phase
  {
  uint32_t v, a, s, k1, k2, p1, p2;
  bool act, st1, st2, any1, any2;
  bool ft1, ft2;                                 // one-sweep encoder: the prediction is what the segment came in with, not known yet
  };

// classes (fpsc.c:76-84 with e1 = 4, e2 = 10): k1 from v[i-1], k2 from the strides of v[i-1] and v[i-2]; run starts
template <bool FULL>
__device__ __forceinline__ void step_head(StepRegs& r, uint32_t v, uint32_t i, uint32_t i_end, const Sweep& sw)
  {
  r.v = v;
  r.act = FULL || i < i_end;
  r.a = dpp_shr1(sw.cy.m1, v);                              // v[i-1]
  const uint32_t b = dpp_shr1(sw.cy.m2, r.a);               // v[i-2]
  const uint32_t s1 = r.a - b;                              // stride of v[i-1]
  const uint32_t s2 = dpp_shr1(sw.cy.m2 - sw.cy.m3, s1);    // stride of v[i-2]
  r.k1 = r.a >> 28;
  r.k2 = 16u + ((((s2 >> 22) & 31u) << 5) ^ (s1 >> 22));
  if (!FULL && !r.act)
    r.k1 = r.k2 = 0xffffffffu;
  r.st1 = r.k1 != dpp_shr1(sw.kc1, r.k1);
  r.st2 = r.k2 != dpp_shr1(sw.kc2, r.k2);
  if (!FULL) { r.st1 = r.st1 && r.act; r.st2 = r.st2 && r.act; }
  r.any1 = __ballot(r.st1) != 0ull;
  r.any2 = __ballot(r.st2) != 0ull;
  r.s = v - r.a;
  r.p1 = r.a;                                               // inside a run: previous value / previous stride
  r.p2 = s1;
  r.ft1 = r.ft2 = false;
  }

// predictions of the run starts (only in steps that have any), MODE as above
template <bool FULL, int MODE>
__device__ __forceinline__ void step_resolve(StepRegs& r, uint32_t* __restrict__ T, Sweep& sw, const LaneK& lk)
  {
  if (MODE == M_XCHG)
    {
    if (r.any1 && r.any2)
      resolve_xchg<FULL, true, true>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T, sw, lk);
    else if (r.any1)
      {
      resolve_xchg<FULL, true, false>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T, sw, lk);
      sw.pend2 = true;
      }
    else if (r.any2)
      {
      resolve_xchg<FULL, false, true>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T, sw, lk);
      sw.pend1 = true;
      }
    else
      sw.pend1 = sw.pend2 = true;
    }
  else if (MODE == M_TAGGED)
    {
    unsigned long long* T64 = (unsigned long long*)T;
    if (r.any1 && r.any2)
      resolve_atomic<FULL, true, true>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T64, sw, lk);
    else if (r.any1)
      {
      resolve_atomic<FULL, true, false>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T64, sw, lk);
      sw.pend2 = true;
      }
    else if (r.any2)
      {
      resolve_atomic<FULL, false, true>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T64, sw, lk);
      sw.pend1 = true;
      }
    else
      sw.pend1 = sw.pend2 = true;
    }
  else if (r.any1 && r.any2)
    resolve<FULL, true, true>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T, sw, lk);
  else if (r.any1)
    {
    resolve<FULL, true, false>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T, sw, lk);
    sw.pend2 = true;
    }
  else if (r.any2)
    {
    resolve<FULL, false, true>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, T, sw, lk);
    sw.pend1 = true;
    }
  else
    sw.pend1 = sw.pend2 = true;
  }

// residual selection, byte layout of the step, bytes into the staging area (flush_end must have run)
struct RecSink { uint32_t* recs; uint32_t off, count; }; // one-sweep encoder: all record lists, word offset of this wave's, its length (4 words each)

template <bool FULL, bool HOLES = false>
__device__ __forceinline__ void step_tail(const StepRegs& r, uint32_t i, uint32_t i_end, uint32_t n, uint8_t* __restrict__ stage, Sweep& sw,
                                          const LaneK& lk, RecSink* sink = nullptr)
  {
  // residual selection (fpsc.c:146-189)
  const uint32_t x1 = r.v ^ r.p1, x2 = r.v ^ (r.a + r.p2);
  const uint32_t n1 = (39u - (uint32_t)__clz((int)x1)) >> 3;
  const uint32_t n2 = (39u - (uint32_t)__clz((int)(x2 | 1u))) >> 3;        // DFCM residuals take at least one byte
  // n2 >= 1, so n2 < n1 already says n1 > 1: one comparison decides, the length is the minimum, and 4 + n2 = 4 | n2 (n2 <= 3 here)
  const bool use2 = n2 < n1;
  uint32_t len = min(n1, n2);
  uint32_t x = use2 ? x2 : x1;
  uint32_t code = use2 ? (n2 | 4u) : n1;
  bool slot = true;
  if (!FULL)
    {
    slot = r.act || (i_end == n && i < ((n + 7u) & ~7u));    // value or tail padding slot (fpsc.c:196-204)
    if (!r.act)
      {
      code = slot ? 1u : 0u;
      len = code;
      x = 0u;
      }
    }
  const bool hole = HOLES && (r.ft1 || r.ft2);
  if (hole)
    {
    // four zero bytes and code 0 for now; k_fpc32_fixup writes the residual (front of the four bytes) and ORs the code in,
    // the gather drops what the residual does not need
    len = 4u;
    code = 0u;
    x = 0u;
    }
  // byte layout of the step: [hdr g0][residuals 0..7][hdr g1][residuals 8..15]...
  // bytes of the residuals below my lane: one DPP scan (six adds) instead of three ballots and their six mbcnt
  const uint32_t incl = wave_scan_incl(len);
  const uint32_t pre = incl - len;
  uint32_t bc = code << lk.sh3;
  bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
  bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
  bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0x141, 0xf, 0xf, true);    // row_half_mirror
  const uint32_t hq = sw.posl + lk.grp3 + pre;               // my group's header (if I lead it), my residual starts at hq + 3
  {
  const uint32_t re = len ? hq + len : lk.dumpq;             // residual end - 3
  // most significant byte first, in this order (a zero byte of lane l must not overtake the byte its owner writes);
  // volatile keeps four byte stores in program order (merged into one dword store, lanes would race)
  typedef __attribute__((address_space(3))) volatile uint8_t lds_vu8;
  lds_vu8* vs = (lds_vu8*)stage;
  vs[re - 1u] = (uint8_t)(x >> 24);
  vs[re] = (uint8_t)(x >> 16);
  vs[re + 1u] = (uint8_t)(x >> 8);
  vs[re + 2u] = (uint8_t)x;
  }
  {
  const bool lead = FULL ? lk.lead : (lk.lead && slot);
  const uint32_t ha = lead ? hq : lk.dumpq;
  stage[ha] = (uint8_t)(bc >> 16);
  stage[ha + 1u] = (uint8_t)(bc >> 8);
  stage[ha + 2u] = (uint8_t)bc;
  }
  if (HOLES)
    {
    const uint64_t hm = __ballot(hole);
    if (hm)
      {
      // record: where the four bytes are (offset in the slot) and how far behind its group header, which value, which classes are
      // open, the prediction that is known if only one is open
      const uint32_t pre_lead = (uint32_t)__builtin_amdgcn_ds_bpermute((lk.lane & ~7) << 2, (int)pre);
      const uint32_t pos = sw.flushed + hq + 3u, dh = 3u + pre - pre_lead;
      const uint32_t idx = sink->count + popc_below(hm);
      if (hole)
        {
        u32x4 w;
        w[0] = pos | (dh << 27);
        w[1] = i;
        w[2] = r.k1 | ((r.k2 - 16u) << 4) | ((uint32_t)r.ft1 << 14) | ((uint32_t)r.ft2 << 15) | (((uint32_t)lk.lane & 7u) << 16);
        w[3] = r.ft1 ? (r.ft2 ? 0u : r.p2) : r.p1;
        *(u32x4*)(sink->recs + sink->off + 4u * idx) = w;
        }
      sink->count += (uint32_t)__popcll(hm);
      }
    }
  const uint32_t hdr = FULL ? 24u : 3u * ((uint32_t)__popcll(__ballot(slot)) >> 3);
  sw.posl += hdr + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }

// first half of a flush: full 256-byte blocks go to the slot as aligned dwords, the rest moves to the front of the staging area.  Only
// the LDS reads are issued here; their data is used by flush_end() in the next step, right before its first byte is staged (LDS
// operations of a wave execute in order), so the wave never waits for the round trip.
__device__ __forceinline__ void flush_begin(Sweep& sw, const uint8_t* __restrict__ stage, const LaneK& lk)
  {
  if (sw.posl >= 256u)
    {
    const uint32_t* stw = (const uint32_t*)stage;
    const uint32_t nb = sw.posl >> 8;                        // 1 or 2
    sw.fw0 = stw[lk.lane];
    sw.fw1 = stw[64 + lk.lane];
    sw.ft = stw[nb * 64u + (uint32_t)lk.lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    sw.fl_nb = nb;
    sw.fl_off = sw.flushed;
    sw.flushed += nb << 8;
    sw.posl &= 255u;
    }
  }

// one step of one component: 64 values starting at index i0 (FULL: all of them inside the segment)
template <bool FULL, int MODE>
__device__ __forceinline__ void code_step(uint32_t v, uint32_t i0, uint32_t i_end, uint32_t n, uint32_t* __restrict__ T,
                                          uint8_t* __restrict__ stage, uint8_t* __restrict__ gbase,
                                          Sweep& sw, const LaneK& lk)
  {
  const uint32_t i = i0 + (uint32_t)lk.lane;
  StepRegs r;
  step_head<FULL>(r, v, i, i_end, sw);
  step_resolve<FULL, MODE>(r, T, sw, lk);
  flush_end(sw, stage, gbase, lk);
  step_tail<FULL>(r, i, i_end, n, stage, sw, lk);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  flush_begin(sw, stage, lk);
  next_carry(sw.cy, v);
  }

// one step of all A components of a tile of 64 vertices (k_fpc32_code_t), phase by phase
template <int A, bool FULL, int MODE, int LW, int TW>
__device__ __forceinline__ void tile_step(const uint32_t (&v)[A], uint32_t i0, uint32_t i_end, uint32_t n, uint32_t* __restrict__ lds,
                                          uint8_t* __restrict__ slots, size_t slot_stride, size_t slot_off, Sweep (&sw)[A], const LaneK& lk)
  {
  const uint32_t i = i0 + (uint32_t)lk.lane;
  StepRegs r[A];
#pragma unroll
  for (int c = 0; c < A; ++c)
    step_head<FULL>(r[c], v[c], i, i_end, sw[c]);
  bool any = false, anyfl = false;
#pragma unroll
  for (int c = 0; c < A; ++c)
    {
    any = any || r[c].any1 || r[c].any2;
    anyfl = anyfl || sw[c].fl_nb != 0u;
    }
  if (any)
    {
#pragma unroll
    for (int c = 0; c < A; ++c)
      step_resolve<FULL, MODE>(r[c], lds + c * LW, sw[c], lk);
    }
  else
    {
#pragma unroll
    for (int c = 0; c < A; ++c)
      sw[c].pend1 = sw[c].pend2 = true;
    }
  if (anyfl)
    {
#pragma unroll
    for (int c = 0; c < A; ++c)
      flush_end(sw[c], (uint8_t*)(lds + c * LW + TW), slots + (size_t)c * slot_stride + slot_off, lk);
    }
#pragma unroll
  for (int c = 0; c < A; ++c)
    step_tail<FULL>(r[c], i, i_end, n, (uint8_t*)(lds + c * LW + TW), sw[c], lk);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int c = 0; c < A; ++c)
    flush_begin(sw[c], (const uint8_t*)(lds + c * LW + TW), lk);
#pragma unroll
  for (int c = 0; c < A; ++c)
    next_carry(sw[c].cy, v[c]);
  }

// MODE M_TAGGED: table entries of 64 bits and resolve_atomic; else 32-bit entries and resolve (ballots) or resolve_xchg.
// `flags`: word 0 is raised when a tagged step found the LDS unit out of lane order.
template <int MODE>
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(8, 8))) k_fpc32_code(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L, uint32_t S,
                                                    const uint32_t* __restrict__ inc, uint8_t* __restrict__ slots, size_t slot_stride,
                                                    uint32_t segcap, uint32_t* __restrict__ segbytes, uint32_t* __restrict__ flags,
                                                    uint32_t prio_mode)
  {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x;
  constexpr bool ATOMIC = MODE == M_TAGGED;
  constexpr int TW = ATOMIC ? 2 * TAB : TAB;           // table words
  volatile uint32_t* prog = lds + arity * (ATOMIC ? LDSW_CA : LDSW_C);   // [4] progress of the component waves (prio_mode 8)
  if (prio_mode >= 1u && prio_mode <= 3u && (uint32_t)c == prio_mode - 1u)
    __builtin_amdgcn_s_setprio(3);
  uint32_t* T = lds + c * (ATOMIC ? LDSW_CA : LDSW_C); // [TAB] payload table (ATOMIC: {payload, tag} pairs)
  uint8_t* stage = (uint8_t*)(T + TW);                 // [STAGE] packed bytes of the steps not yet flushed + dump
  // incoming table: payload of the last writer of every class before this segment (0 if none).  Every wave of the sweep is here at the
  // same time, so nobody covers anybody's latency: the index loads of a batch of entries are issued together, then the value loads
  // they point at (two round trips per batch of nine instead of two per entry).
  const uint32_t* row = inc + ((size_t)g * arity + c) * ROW;
  constexpr int TB = 9;                                // 17 entries per lane = 9 + 8
#pragma unroll 1
  for (int k0 = lane; k0 < TAB; k0 += 64 * TB)
    {
    uint32_t idx[TB], vi[TB], vp[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j)
      {
      const int k = k0 + 64 * j;
      idx[j] = k < TAB ? row[k] : 0u;
      }
#pragma unroll
    for (int j = 0; j < TB; ++j)
      {
      vi[j] = idx[j] ? src[(size_t)(idx[j] - 1u) * arity + c] : 0u;
      vp[j] = idx[j] >= 2u ? src[(size_t)(idx[j] - 2u) * arity + c] : 0u;
      }
#pragma unroll
    for (int j = 0; j < TB; ++j)
      {
      const int k = k0 + 64 * j;
      const uint32_t pay = idx[j] ? (k < 16 ? vi[j] : vi[j] - vp[j]) : 0u;
      if (k < TAB)
        {
        if (ATOMIC)
          ((unsigned long long*)T)[k] = pay;           // tag 0: older than every step of this segment
        else
          T[k] = pay;
        }
      }
    }
  LaneK lk;
  lk.lane = lane;
  lk.lt = (1ull << lane) - 1ull;
  lk.bit = 1ull << lane;
  lk.sh3 = 3u * ((uint32_t)lane & 7u);
  lk.grp3 = 3u * ((uint32_t)lane >> 3);
  lk.dumpw = (uint32_t)(TW + STAGE_LIVE / 4 + lane);
  lk.dumpq = (uint32_t)(STAGE_LIVE + 4 * lane + 1);
  lk.lead = (lane & 7) == 0;
  const uint32_t i_begin = g * L;
  const uint32_t i_end = (n - i_begin < L) ? n : i_begin + L;
  uint8_t* gbase = slots + (size_t)c * slot_stride + (size_t)g * segcap;
  Sweep sw;
  sw.kc1 = sw.kc2 = 0xfffffffeu;                       // the first value of a segment always looks at the table
  sw.pend1 = sw.pend2 = false;
  sw.posl = 0;
  sw.flushed = 0;
  sw.tag = 0;
  sw.viol = 0;
  sw.fl_nb = 0;
  if (g == 0)
    {
    if (lane == 0)
      {
      stage[0] = 0x25;                      // (4/2) << 4 | (10/2), fpsc.c:120
      stage[1] = (uint8_t)(n >> 24); stage[2] = (uint8_t)(n >> 16); stage[3] = (uint8_t)(n >> 8); stage[4] = (uint8_t)n;
      }
    sw.posl = 5u;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  sw.cy = load_carry(src, i_begin, arity, c);
  uint32_t cur[PFC];                                   // the next PFC steps' values: a slot is loaded again as soon as its step begins
  load_block(cur, src, i_begin, i_end, arity, c, lane);
  const uint32_t lag = prio_mode >> 8;                 // TRICO_FPC32_LAG: blocks a component wave may run ahead of the slowest (0 = any)
  prio_mode &= 255u;
  if (prio_mode == 8u && lane == 0)
    prog[c] = i_begin;
  if (lag)
    __syncthreads();                                   // everybody's progress word is this workgroup's before anybody compares
  for (uint32_t ib = i_begin; ib < i_end; ib += 64u * PFC)
    {
    if (prio_mode == 8u)
      {
      // the component that is behind gets the issue slots first: the waves of a workgroup hold their LDS until the last of them
      // is done, and the sweep ends when the slowest component does
      if (lane == 0)
        prog[c] = ib;
      uint32_t ahead = 0;
      for (int o = 0; o < arity; ++o)
        ahead = max(ahead, (uint32_t)__builtin_amdgcn_readfirstlane((int)prog[o]) + 1u);      // (a finished wave's 0xffffffff counts as 0)
      if (ib + 1u + 64u * PFC <= ahead)
        __builtin_amdgcn_s_setprio(3);
      else
        __builtin_amdgcn_s_setprio(0);
      // ... and with a lag the ones in front wait for it (bounded), so that the three waves read the same cache lines at about the
      // same time and the interleaved array comes over HBM once
      for (uint32_t spin = 0; lag && spin < 4096u; ++spin)
        {
        uint32_t lo = 0xffffffffu;
        for (int o = 0; o < arity; ++o)
          lo = min(lo, (uint32_t)__builtin_amdgcn_readfirstlane((int)prog[o]));
        if (lo == 0xffffffffu || ib <= lo + lag * 64u * PFC)
          break;
        __builtin_amdgcn_s_sleep(2);
        }
      }
#pragma unroll
    for (int pu = 0; pu < PFC; ++pu)
      {
      const uint32_t i0 = ib + 64u * pu;
      const uint32_t v = cur[pu];
      {
      // the value this slot holds PFC steps from now: always PFC steps in flight, never more (a whole block fetched ahead doubled what
      // a workgroup keeps in the L2, and 320 workgroups per XCD share 4 MB)
      const uint32_t in = i0 + 64u * PFC + (uint32_t)lane;
      cur[pu] = in < i_end ? src[(size_t)in * arity + c] : 0u;
      }
      if (i0 + 64u <= i_end)
        code_step<true, MODE>(v, i0, i_end, n, T, stage, gbase, sw, lk);
      else if (i0 < i_end)
        code_step<false, MODE>(v, i0, i_end, n, T, stage, gbase, sw, lk);
      }
    }
  // what is left in the staging area (< 256 bytes)
  flush_end(sw, stage, gbase, lk);
  {
  const uint32_t off = 4u * (uint32_t)lane;
  if (off < sw.posl)
    store_span(gbase + sw.flushed, off, ((const uint32_t*)stage)[lane], sw.posl);
  }
  if (prio_mode == 8u && lane == 0)
    prog[c] = 0xffffffffu;                             // done: nobody is behind me any more, nobody waits for me
  if (lane == 0)
    segbytes[(size_t)c * S + g] = sw.flushed + sw.posl;
  if (ATOMIC && sw.viol && lane == 0)
    atomicOr(flags, 1u);
  }

// ---- ONE sweep: code every segment without knowing what it comes in with --------------------------------------------------------
// The two-sweep scheme reads the input twice because a segment's first lookup of a class needs the latest writer of that class in
// everything before it.  Those lookups are few (one per class the segment touches: a handful for a smooth coordinate, a few hundred of
// 19,584 values for a noisy one), and nothing else depends on them - a value's prediction only decides ITS residual and code.  So:
//   k_fpc32_sweep1  the code sweep with the exchange resolve (resolve_xchg_h), tables starting as "unknown".  A value whose prediction
//                   would come from the incoming table gets four zero bytes and code 0 and a 16-byte record; at the end the wave
//                   publishes its table (= what the segment leaves behind, per class) and the bitmap of the classes it wrote.
//   k_fpc32_pscan_* incoming payload of every (segment, class) = the entry of the nearest earlier segment that wrote the class.
//   k_fpc32_fixup   per record: residual, length and code from the incoming entry, and where the gather will find them in its output
//                   (nothing is written to the slot: scattered stores on a million fields cost more than the sweep saves); the
//                   unused bytes of the fields are counted per segment.
//   k_fpc32_offsets, k_fpc32_gather (which drops the unused bytes of the fields and ORs residuals and codes into the vectors on their
//                   way through its registers).
// The input is read once (+ 8 bytes per record), the index sweep and its zeroing are gone.
constexpr int SEENW = 36;                               // words of the per-wave class bitmap (1040 bits, padded)
constexpr int LDSW_1 = TAB + STAGE / 4 + SEENW;         // per-wave LDS words of k_fpc32_sweep1 (5,104 B)
constexpr uint32_t RCAP = 1040;                         // records per (segment, component): a class is met first at most once

__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_fpc32_sweep1(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L, uint32_t S, uint32_t* __restrict__ outT,
               uint32_t* __restrict__ outSeen, uint8_t* __restrict__ slots, size_t slot_stride, uint32_t segcap,
               uint32_t* __restrict__ segbytes, uint32_t* __restrict__ nrec, uint32_t* __restrict__ recs, uint32_t prio_mode)
  {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x;
  prio_mode &= 255u;                                   // (the upper bits carry the two-sweep code sweep's lag)
  volatile uint32_t* prog = lds + arity * LDSW_1;
  uint32_t* T = lds + c * LDSW_1;
  uint8_t* stage = (uint8_t*)(T + TAB);
  uint32_t* seen = T + TAB + STAGE / 4;
  for (int k = lane; k < TAB; k += 64)
    T[k] = SENT;
  if (lane < SEENW)
    seen[lane] = 0u;
  LaneK lk;
  lk.lane = lane;
  lk.lt = (1ull << lane) - 1ull;
  lk.bit = 1ull << lane;
  lk.sh3 = 3u * ((uint32_t)lane & 7u);
  lk.grp3 = 3u * ((uint32_t)lane >> 3);
  lk.dumpw = (uint32_t)(TAB + STAGE_LIVE / 4 + lane);
  lk.dumpq = (uint32_t)(STAGE_LIVE + 4 * lane + 1);
  lk.lead = (lane & 7) == 0;
  const uint32_t i_begin = g * L;
  const uint32_t i_end = (n - i_begin < L) ? n : i_begin + L;
  uint8_t* gbase = slots + (size_t)c * slot_stride + (size_t)g * segcap;
  const size_t rowi = (size_t)g * arity + c;
  RecSink sink = { recs, (uint32_t)(rowi * RCAP * 4u), 0u };
  Sweep sw;
  sw.kc1 = sw.kc2 = 0xfffffffeu;
  sw.pend1 = sw.pend2 = false;
  sw.posl = 0;
  sw.flushed = 0;
  sw.tag = 0;
  sw.viol = 0;
  sw.fl_nb = 0;
  if (g == 0)
    {
    if (lane == 0)
      {
      stage[0] = 0x25;                      // (4/2) << 4 | (10/2), fpsc.c:120
      stage[1] = (uint8_t)(n >> 24); stage[2] = (uint8_t)(n >> 16); stage[3] = (uint8_t)(n >> 8); stage[4] = (uint8_t)n;
      }
    sw.posl = 5u;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t cur[PF];                                    // rolling prefetch as in k_fpc32_code
  sw.cy = load_carry(src, i_begin, arity, c);
  load_block(cur, src, i_begin, i_end, arity, c, lane);
  if (prio_mode == 8u && lane == 0)
    prog[c] = i_begin;
  for (uint32_t ib = i_begin; ib < i_end; ib += 64u * PF)
    {
    if (prio_mode == 8u)
      {
      if (lane == 0)
        prog[c] = ib;
      uint32_t ahead = 0;
      for (int o = 0; o < arity; ++o)
        ahead = max(ahead, (uint32_t)__builtin_amdgcn_readfirstlane((int)prog[o]) + 1u);
      if (ib + 1u + 64u * PF <= ahead)
        __builtin_amdgcn_s_setprio(3);
      else
        __builtin_amdgcn_s_setprio(0);
      }
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      {
      const uint32_t i0 = ib + 64u * pu;
      if (i0 >= i_end)
        break;
      const uint32_t i = i0 + (uint32_t)lane;
      const uint32_t vcur = cur[pu];
      {
      const uint32_t in = i0 + 64u * PF + (uint32_t)lane;
      cur[pu] = in < i_end ? src[(size_t)in * arity + c] : 0u;
      }
      StepRegs r;
      if (i0 + 64u <= i_end)
        {
        step_head<true>(r, vcur, i, i_end, sw);
        if (r.any1 && r.any2)
          resolve_xchg_h<true, true, true>(r.k1, r.k2, r.st1, r.st2, true, r.v, r.s, r.p1, r.p2, r.ft1, r.ft2, T, seen, sw, lk);
        else if (r.any1)
          {
          resolve_xchg_h<true, true, false>(r.k1, r.k2, r.st1, r.st2, true, r.v, r.s, r.p1, r.p2, r.ft1, r.ft2, T, seen, sw, lk);
          sw.pend2 = true;
          }
        else if (r.any2)
          {
          resolve_xchg_h<true, false, true>(r.k1, r.k2, r.st1, r.st2, true, r.v, r.s, r.p1, r.p2, r.ft1, r.ft2, T, seen, sw, lk);
          sw.pend1 = true;
          }
        else
          sw.pend1 = sw.pend2 = true;
        flush_end(sw, stage, gbase, lk);
        step_tail<true, true>(r, i, i_end, n, stage, sw, lk, &sink);
        }
      else
        {
        step_head<false>(r, vcur, i, i_end, sw);
        if (r.any1 || r.any2)
          resolve_xchg_h<false, true, true>(r.k1, r.k2, r.st1, r.st2, r.act, r.v, r.s, r.p1, r.p2, r.ft1, r.ft2, T, seen, sw, lk);
        else
          sw.pend1 = sw.pend2 = true;
        flush_end(sw, stage, gbase, lk);
        step_tail<false, true>(r, i, i_end, n, stage, sw, lk, &sink);
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      flush_begin(sw, stage, lk);
      next_carry(sw.cy, vcur);
      }
    }
  flush_end(sw, stage, gbase, lk);
  {
  const uint32_t off = 4u * (uint32_t)lane;
  if (off < sw.posl)
    store_span(gbase + sw.flushed, off, ((const uint32_t*)stage)[lane], sw.posl);
  }
  if (prio_mode == 8u && lane == 0)
    prog[c] = 0u;
  // what the segment leaves behind: the table with the last value's writes applied, and which classes it wrote at all
  if (lane == 0)
    {
    if (sw.pend1) T[sw.kc1] = sw.cy.m1;
    if (sw.pend2) T[sw.kc2] = sw.cy.m1 - sw.cy.m2;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int k = lane; k < TAB; k += 64)
    outT[rowi * ROW + k] = T[k];
  if (lane < SEENW)
    outSeen[rowi * SEENW + lane] = seen[lane];
  if (lane == 0)
    {
    segbytes[(size_t)c * S + g] = sw.flushed + sw.posl;
    nrec[rowi] = sink.count;
    }
  }

// incoming payload of (segment, class): the published entry of the nearest earlier segment whose bitmap has the class, else 0
__global__ void __launch_bounds__(256) k_fpc32_pscan_a(const uint32_t* __restrict__ outT, const uint32_t* __restrict__ outSeen, uint32_t S, int arity,
                                                       uint32_t* __restrict__ chlast, uint32_t* __restrict__ chhas, uint32_t* __restrict__ flags)
  {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    flags[0] = 0u;                                           // read back with the sizes (k_fpc32_offsets); only the tagged sweep raises it
  const uint32_t col = blockIdx.x * 256u + threadIdx.x;
  const uint32_t ncol = (uint32_t)arity * TAB;
  if (col >= ncol)
    return;
  const uint32_t c = col / TAB, k = col % TAB;
  const uint32_t g0 = blockIdx.y * CH, g1 = (g0 + CH < S) ? g0 + CH : S;
  uint32_t last = 0, has = 0;
#pragma unroll 8
  for (uint32_t g = g0; g < g1; ++g)
    {
    const size_t r = (size_t)g * arity + c;
    const uint32_t w = outSeen[r * SEENW + (k >> 5)], t = outT[r * ROW + k];
    if ((w >> (k & 31u)) & 1u) { last = t; has = 1u; }
    }
  chlast[(size_t)blockIdx.y * ncol + col] = last;
  chhas[(size_t)blockIdx.y * ncol + col] = has;
  }

__global__ void __launch_bounds__(256) k_fpc32_pscan_b(const uint32_t* __restrict__ outT, const uint32_t* __restrict__ outSeen, uint32_t S, int arity,
                                                       const uint32_t* __restrict__ chlast, const uint32_t* __restrict__ chhas,
                                                       uint32_t* __restrict__ inc)
  {
  const uint32_t col = blockIdx.x * 256u + threadIdx.x;
  const uint32_t ncol = (uint32_t)arity * TAB;
  if (col >= ncol)
    return;
  const uint32_t c = col / TAB, k = col % TAB;
  uint32_t carry = 0;
  for (uint32_t j = 0; j < blockIdx.y; ++j)
    if (chhas[(size_t)j * ncol + col])
      carry = chlast[(size_t)j * ncol + col];
  const uint32_t g0 = blockIdx.y * CH, g1 = (g0 + CH < S) ? g0 + CH : S;
#pragma unroll 8
  for (uint32_t g = g0; g < g1; ++g)
    {
    const size_t r = (size_t)g * arity + c;
    const uint32_t w = outSeen[r * SEENW + (k >> 5)], t = outT[r * ROW + k];
    inc[r * ROW + k] = carry;
    if ((w >> (k & 31u)) & 1u) carry = t;
    }
  }

// the deferred values: residual, length and code from the incoming entries (fpsc.c:133-189 for one value).  Nothing is written to the
// slot here (scattered byte stores and atomics on 1 M fields cost 0.13 ms): the records take the result, in the form the gather wants -
// it ORs residuals and codes into the bytes on their way through its registers and drops the unused bytes of the fields.  Record
// afterwards: w0 = e (output position at which the unused rest of the field would start: the residual is the `length` output bytes
// right before it), w1 = residual, w2 = code | length << 4 | index in the group << 8 | (residual start - header, in output bytes) << 12,
// w3 = unused field bytes up to and including this record.
__global__ void __launch_bounds__(256) k_fpc32_fixup(const uint32_t* __restrict__ src, int arity, uint32_t S, const uint32_t* __restrict__ inc,
                                                     uint32_t* __restrict__ segbytes, const uint32_t* __restrict__ nrec, uint32_t* __restrict__ recs)
  {
  __shared__ uint32_t rp[RCAP], cum[RCAP], part[4];
  const uint32_t g = blockIdx.x, c = blockIdx.y;
  const size_t rowi = (size_t)g * arity + c;
  const uint32_t H = nrec[rowi];
  if (H == 0u)
    return;
  uint32_t* list = recs + rowi * RCAP * 4u;
  const uint32_t* row = inc + rowi * ROW;
  // consecutive records per thread, so that the running sum of unused bytes is a scan over threads
  const uint32_t per = (H + 255u) / 256u;
  const uint32_t j0 = threadIdx.x * per < H ? threadIdx.x * per : H, j1 = (j0 + per < H) ? j0 + per : H;
  uint32_t sum = 0;
  for (uint32_t j = j0; j < j1; ++j)
    {
    const u32x4 w = *(const u32x4*)(list + 4u * j);
    const uint32_t i = w[1];
    const uint32_t k1 = w[2] & 15u, k2 = 16u + ((w[2] >> 4) & 1023u), gi = (w[2] >> 16) & 7u;
    const bool ft1 = (w[2] >> 14) & 1u, ft2 = (w[2] >> 15) & 1u;
    const uint32_t v = src[(size_t)i * arity + c];
    const uint32_t a = i ? src[(size_t)(i - 1u) * arity + c] : 0u;
    const uint32_t p1 = ft1 ? row[k1] : w[3];
    const uint32_t p2 = ft2 ? row[k2] : w[3];
    const uint32_t x1 = v ^ p1, x2 = v ^ (a + p2);
    const uint32_t n1 = blen(x1);
    uint32_t n2 = blen(x2);
    n2 = n2 ? n2 : 1u;
    const bool use2 = n2 < n1;
    const uint32_t len = use2 ? n2 : n1, x = use2 ? x2 : x1, code = use2 ? (n2 | 4u) : n1;
    rp[j] = w[0];
    list[4u * j + 1u] = x;
    list[4u * j + 2u] = code | (len << 4) | (gi << 8);
    sum += 4u - len;
    cum[j] = 4u - len;                                    // for now: this record's unused bytes
    }
  const uint32_t incl = wave_scan_incl(sum);
  if ((threadIdx.x & 63u) == 63u)
    part[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t run = incl - sum;
  for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv)
    run += part[wv];
  for (uint32_t j = j0; j < j1; ++j)
    {
    run += cum[j];
    cum[j] = run;
    }
  __syncthreads();
  for (uint32_t j = j0; j < j1; ++j)
    {
    const uint32_t pos = rp[j] & 0x7ffffffu, hdr = pos - (rp[j] >> 27);
    const uint32_t before = j ? cum[j - 1u] : 0u, u = cum[j] - before;
    // unused bytes before the group header: those of the fields that END at or before it (the fields of the same group between the
    // header and this record lie behind the header)
    uint32_t jj = j, hshift = 0;
    while (jj > 0u)
      {
      --jj;
      if ((rp[jj] & 0x7ffffffu) + 4u <= hdr) { hshift = cum[jj]; break; }
      }
    const uint32_t rstart = pos - before;                 // output position of the field's first byte
    list[4u * j] = rstart + 4u - u;                       // e
    list[4u * j + 2u] |= (rstart - (hdr - hshift)) << 12;
    list[4u * j + 3u] = cum[j];
    }
  if (threadIdx.x == 255u)
    segbytes[(size_t)c * S + g] -= run;                   // thread 255 ends with the total
  }

// ---- tile variant of sweep C: one wave codes ALL components of its segment (see k_fpc32_index_t) --------------------------
// Same steps, same tables and staging areas per component (side by side in the wave's LDS), same slots: only who walks them
// differs.  The interleaved array is read once, 64 whole vertices per load.
template <int A, int MODE>
__global__ void __launch_bounds__(64) k_fpc32_code_t(const uint32_t* __restrict__ src, uint32_t n, uint32_t L, uint32_t S,
                                                     const uint32_t* __restrict__ inc, uint8_t* __restrict__ slots, size_t slot_stride,
                                                     uint32_t segcap, uint32_t* __restrict__ segbytes, uint32_t* __restrict__ flags)
  {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  const uint32_t g = blockIdx.x;
  constexpr bool ATOMIC = MODE == M_TAGGED;
  constexpr int TW = ATOMIC ? 2 * TAB : TAB;
  constexpr int LW = ATOMIC ? LDSW_CA : LDSW_C;
  // incoming tables: payload of the last writer of every class before this segment (0 if none); the loads of all components
  // of an iteration are independent
  const uint32_t* row = inc + (size_t)g * A * ROW;
  for (int k = lane; k < TAB; k += 64)
    {
    uint32_t idx[A], vi[A], vp[A];
#pragma unroll
    for (int c = 0; c < A; ++c)
      idx[c] = row[c * ROW + k];
#pragma unroll
    for (int c = 0; c < A; ++c)
      {
      vi[c] = idx[c] ? src[(size_t)(idx[c] - 1u) * A + c] : 0u;
      vp[c] = idx[c] >= 2u ? src[(size_t)(idx[c] - 2u) * A + c] : 0u;
      }
#pragma unroll
    for (int c = 0; c < A; ++c)
      {
      const uint32_t pay = idx[c] ? (k < 16 ? vi[c] : vi[c] - vp[c]) : 0u;
      if (ATOMIC)
        ((unsigned long long*)(lds + c * LW))[k] = pay;
      else
        lds[c * LW + k] = pay;
      }
    }
  LaneK lk;
  lk.lane = lane;
  lk.lt = (1ull << lane) - 1ull;
  lk.bit = 1ull << lane;
  lk.sh3 = 3u * ((uint32_t)lane & 7u);
  lk.grp3 = 3u * ((uint32_t)lane >> 3);
  lk.dumpw = (uint32_t)(TW + STAGE_LIVE / 4 + lane);
  lk.dumpq = (uint32_t)(STAGE_LIVE + 4 * lane + 1);
  lk.lead = (lane & 7) == 0;
  const uint32_t i_begin = g * L;
  const uint32_t i_end = (n - i_begin < L) ? n : i_begin + L;
  Sweep sw[A];
#pragma unroll
  for (int c = 0; c < A; ++c)
    {
    sw[c].kc1 = sw[c].kc2 = 0xfffffffeu;
    sw[c].pend1 = sw[c].pend2 = false;
    sw[c].posl = 0;
    sw[c].flushed = 0;
    sw[c].tag = 0;
    sw[c].viol = 0;
    sw[c].fl_nb = 0;
    if (g == 0)
      {
      uint8_t* stage = (uint8_t*)(lds + c * LW + TW);
      if (lane == 0)
        {
        stage[0] = 0x25;
        stage[1] = (uint8_t)(n >> 24); stage[2] = (uint8_t)(n >> 16); stage[3] = (uint8_t)(n >> 8); stage[4] = (uint8_t)n;
        }
      sw[c].posl = 5u;
      }
    sw[c].cy = load_carry(src, i_begin, A, c);
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t cur[PFT][A], nxt[PFT][A];
  load_tiles<A, PFT>(cur, src, i_begin, i_end, lane);
  uint32_t ib = i_begin;
  for (; ib + 64u * PFT <= i_end; ib += 64u * PFT)
    {
    load_tiles<A, PFT>(nxt, src, ib + 64u * PFT, i_end, lane);
#pragma unroll
    for (int pu = 0; pu < PFT; ++pu)
      tile_step<A, true, MODE, LW, TW>(cur[pu], ib + 64u * pu, i_end, n, lds, slots, slot_stride, (size_t)g * segcap, sw, lk);
#pragma unroll
    for (int pu = 0; pu < PFT; ++pu)
#pragma unroll
      for (int c = 0; c < A; ++c)
        cur[pu][c] = nxt[pu][c];
    }
  // fewer than PFT tiles are left; only the last one of a stream can be partial
  for (; ib < i_end; ib += 64u)
    {
    const uint32_t i = ib + (uint32_t)lane;
    VertexT<A> t = {};
    if (i < i_end)
      t = *(const VertexT<A>*)(src + (size_t)i * A);
    if (ib + 64u <= i_end)
      tile_step<A, true, MODE, LW, TW>(t.w, ib, i_end, n, lds, slots, slot_stride, (size_t)g * segcap, sw, lk);
    else
      tile_step<A, false, MODE, LW, TW>(t.w, ib, i_end, n, lds, slots, slot_stride, (size_t)g * segcap, sw, lk);
    }
  uint32_t viol = 0;
#pragma unroll
  for (int c = 0; c < A; ++c)
    {
    uint8_t* gb = slots + (size_t)c * slot_stride + (size_t)g * segcap;
    flush_end(sw[c], (uint8_t*)(lds + c * LW + TW), gb, lk);
    const uint32_t off = 4u * (uint32_t)lane;
    if (off < sw[c].posl)
      store_span(gb + sw[c].flushed, off, (lds + c * LW + TW)[lane], sw[c].posl);
    if (lane == 0)
      segbytes[(size_t)c * S + g] = sw[c].flushed + sw[c].posl;
    viol |= sw[c].viol;
    }
  if (ATOMIC && viol && lane == 0)
    atomicOr(flags, 1u);
  }

// ---- offsets: exclusive scan of segment sizes per component (one workgroup per component) -------------
__global__ void __launch_bounds__(1024) k_fpc32_offsets(const uint32_t* __restrict__ segbytes, uint32_t S, uint32_t* __restrict__ segoff,
                                                        uint32_t* __restrict__ sizes, const uint32_t* __restrict__ flags)
  {
  if (blockIdx.x == 0 && threadIdx.x == 0)
    sizes[3] = flags[0];                           // read back with the sizes: "the code sweep distrusts its LDS atomics"
  __shared__ uint32_t part[1024];
  const uint32_t c = blockIdx.x;
  const uint32_t per = (S + 1023u) / 1024u;
  const uint32_t g0 = threadIdx.x * per, g1 = (g0 + per < S) ? g0 + per : S;
  uint32_t sum = 0;
  for (uint32_t g = g0; g < g1; ++g)
    sum += segbytes[(size_t)c * S + g];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t o = 1; o < 1024u; o <<= 1)
    {
    const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
    }
  uint32_t run = part[threadIdx.x] - sum;
  for (uint32_t g = g0; g < g1; ++g)
    {
    segoff[(size_t)c * S + g] = run;
    run += segbytes[(size_t)c * S + g];
    }
  if (threadIdx.x == 1023u)
    sizes[c] = part[1023];
  }

// ---- gather: segment slots -> contiguous payload -------------------------------------------------------
// grid (S, arity); each workgroup moves one segment.  The destination is written as aligned 16-byte vectors;
// the source (a 256-byte aligned slot) is read as 4 + 1 dwords per vector and re-aligned with v_alignbyte.
constexpr uint32_t GBLK = 512;                    // gather, one-sweep path: search index per 256 output bytes (segments up to 128 KiB)
constexpr uint32_t GDIRTY = 4096;                 // ... vectors near a record that wait for the second pass (power of two)
struct GatherDst { uint8_t* p[3]; };             // destination of every component (grid.y)

__global__ void __launch_bounds__(256) k_fpc32_gather(const uint8_t* __restrict__ slots, size_t slot_stride, uint32_t segcap, uint32_t S,
                                                      const uint32_t* __restrict__ segbytes, const uint32_t* __restrict__ segoff,
                                                      GatherDst dst, const uint32_t* __restrict__ nrec, const uint32_t* __restrict__ recs,
                                                      int arity, int c0)
  {
  __shared__ uint32_t e[RCAP], cum[RCAP], rx[RCAP], rm[RCAP], bidx[GBLK], ndirty;      // (RCAP >= 256: one speculative record per thread)
  __shared__ uint16_t dirty[GDIRTY];
  const uint32_t g = blockIdx.x, c = blockIdx.y;
  const uint32_t len = segbytes[(size_t)c * S + g];
  const uint8_t* s = slots + (size_t)c * slot_stride + (size_t)g * segcap;       // 256-byte aligned, segcap has 280 bytes of slack
  uint8_t* d = dst.p[c] + segoff[(size_t)c * S + g];
  const uint32_t head = (uint32_t)((16u - ((uintptr_t)d & 15u)) & 15u);           // bytes until d is 16-byte aligned
  const uint32_t h = head < len ? head : len;
  const uint32_t body = (len - h) >> 4;                                           // aligned destination vectors
  u32x4* dd = (u32x4*)(d + h);
  const uint32_t done = h + 16u * body;
  const size_t rowi = (size_t)g * arity + (size_t)(c0 + (int)c);
  const uint32_t* list = recs + rowi * RCAP * 4u;
  // the thread's record before the number of records is known (garbage beyond it): one round trip less on the way to the first copy
  const u32x4 spec = *(const u32x4*)(list + 4u * threadIdx.x);
  const uint32_t H = nrec[rowi];
  if (H == 0u)
    {
    if (threadIdx.x < h)
      d[threadIdx.x] = s[threadIdx.x];
    const uint32_t* ss = (const uint32_t*)s + (h >> 2);
    const uint32_t sh = h & 3u;
    // destination vector t holds source bytes h + 16t .. h + 16t + 15
    for (uint32_t t = threadIdx.x; t < body; t += 256u)
      {
      const u32x4 lo = *(const u32x4*)(ss + 4u * t);                                // 4-byte aligned 16-byte load
      const uint32_t hi = ss[4u * t + 4u];
      u32x4 o;
      o[0] = __builtin_amdgcn_alignbyte(lo[1], lo[0], sh);
      o[1] = __builtin_amdgcn_alignbyte(lo[2], lo[1], sh);
      o[2] = __builtin_amdgcn_alignbyte(lo[3], lo[2], sh);
      o[3] = __builtin_amdgcn_alignbyte(hi, lo[3], sh);
      dd[t] = o;
      }
    if (threadIdx.x < len - done)
      d[done + threadIdx.x] = s[done + threadIdx.x];
    return;
    }
  // One-sweep encoder: the slot holds H reserved fields of four zero bytes (k_fpc32_sweep1); the records say what belongs there
  // (k_fpc32_fixup).  Output byte o is slot byte o + (unused field bytes before it); e[j] = output position at which the unused rest
  // of field j would start - so the residual of record j is the output bytes right before e[j] - and cum[j] = unused bytes up to and
  // including field j; e is non-decreasing (fields do not overlap), so the shift of an output position is a search in e.  Residual
  // bytes and code bits are ORed into the vectors on their way through the registers (everything a record touches was left zero by
  // the sweep); a record can only touch a vector that begins less than 51 bytes before its e (4 residual bytes, at most 31 from the
  // group header to the field).
  uint32_t bs = 8;
  while ((len >> bs) >= GBLK - 1u)
    ++bs;
  if (threadIdx.x == 0)
    ndirty = 0u;
  e[threadIdx.x] = spec[0];
  rx[threadIdx.x] = spec[1];
  rm[threadIdx.x] = spec[2];
  cum[threadIdx.x] = spec[3];
  for (uint32_t j = threadIdx.x + 256u; j < H; j += 256u)
    {
    const u32x4 w = *(const u32x4*)(list + 4u * j);
    e[j] = w[0];
    rx[j] = w[1];
    rm[j] = w[2];
    cum[j] = w[3];
    }
  __syncthreads();
  // first record with e > o: by bisection for a handful of records; for more, from the number of unused ranges that start at or
  // before every 256th output position (coarser for segments beyond 128 KiB), which costs a pass and a barrier
  const bool indexed = H > 16u;
  if (indexed)
    {
    for (uint32_t bq = threadIdx.x; (bq << bs) <= len; bq += 256u)
      {
      const uint32_t o = bq << bs;
      uint32_t lo = 0, hi = H;
      while (lo < hi)
        {
        const uint32_t mid = (lo + hi) >> 1;
        if (e[mid] <= o) lo = mid + 1u; else hi = mid;
        }
      bidx[bq] = lo;
      }
    __syncthreads();
    }
  auto first_after = [&](uint32_t o) -> uint32_t
    {
    if (indexed)
      {
      uint32_t m = bidx[o >> bs];
      while (m < H && e[m] <= o) ++m;
      return m;
      }
    uint32_t lo = 0, hi = H;
    while (lo < hi)
      {
      const uint32_t mid = (lo + hi) >> 1;
      if (e[mid] <= o) lo = mid + 1u; else hi = mid;
      }
    return lo;
    };
  // ORs what the records say into the 16 output bytes that begin at o (m = first record with e > o)
  auto patch = [&](u32x4& out, uint32_t o, uint32_t m)
    {
    for (uint32_t j = m; j < H && e[j] < o + 51u; ++j)
      {
      const uint32_t x = rx[j], rmj = rm[j];
      const uint32_t ln = (rmj >> 4) & 7u, h24 = (rmj & 7u) << (3u * ((rmj >> 8) & 7u)), oh = e[j] - ln - (rmj >> 12);
      for (uint32_t bb = 0; bb < ln; ++bb)
        {
        const uint32_t q = e[j] - ln + bb - o;                                    // wraps to a huge number if before o
        if (q < 16u)
          out[q >> 2] |= ((x >> (8u * (ln - 1u - bb))) & 255u) << (8u * (q & 3u));
        }
      for (uint32_t bb = 0; bb < 3u; ++bb)
        {
        const uint32_t q = oh + bb - o;
        if (q < 16u)
          out[q >> 2] |= ((h24 >> (8u * (2u - bb))) & 255u) << (8u * (q & 3u));
        }
      }
    };
  // one output byte the slow way (head and tail bytes of the segment)
  auto out_byte = [&](uint32_t o) -> uint8_t
    {
    const uint32_t m = first_after(o);
    u32x4 v = { (uint32_t)s[o + (m ? cum[m - 1u] : 0u)], 0u, 0u, 0u };
    patch(v, o, m);
    return (uint8_t)v[0];
    };
  if (threadIdx.x < h)
    d[threadIdx.x] = out_byte(threadIdx.x);
  // Pass 1: the vectors no record is near are a plain copy with a shift; the others are only noted.  Pass 2 takes those, densely packed
  // over the lanes: with a few records per KiB every wave of a single pass would walk the slow code for a handful of its lanes.
  for (uint32_t t = threadIdx.x; t < body; t += 256u)
    {
    const uint32_t o = h + 16u * t;
    const uint32_t m = first_after(o);
    if (m == H || e[m] >= o + 51u)
      {
      const uint32_t so = o + (m ? cum[m - 1u] : 0u), sh = so & 3u;
      const uint32_t* ss = (const uint32_t*)(s + (so & ~3u));
      const u32x4 lo = *(const u32x4*)ss;
      const uint32_t hi = ss[4];
      u32x4 out;
      out[0] = __builtin_amdgcn_alignbyte(lo[1], lo[0], sh);
      out[1] = __builtin_amdgcn_alignbyte(lo[2], lo[1], sh);
      out[2] = __builtin_amdgcn_alignbyte(lo[3], lo[2], sh);
      out[3] = __builtin_amdgcn_alignbyte(hi, lo[3], sh);
      dd[t] = out;
      }
    else
      dirty[atomicAdd(&ndirty, 1u) & (GDIRTY - 1u)] = (uint16_t)t;
    }
  __syncthreads();
  const uint32_t nd = body > 65535u ? GDIRTY + 1u : ndirty;      // (16-bit vector numbers: segments beyond 1 MiB take every vector again)
  // (more than GDIRTY of them: the list has wrapped and is useless, every vector is taken again)
  for (uint32_t q = threadIdx.x; q < (nd <= GDIRTY ? nd : body); q += 256u)
    {
    const uint32_t t = nd <= GDIRTY ? (uint32_t)dirty[q] : q;
    const uint32_t o = h + 16u * t;
    const uint32_t m = first_after(o);
    uint32_t shift = m ? cum[m - 1u] : 0u;
    u32x4 out;
    if (m == H || e[m] >= o + 16u)
      {
      const uint32_t so = o + shift, sh = so & 3u;
      const uint32_t* ss = (const uint32_t*)(s + (so & ~3u));
      const u32x4 lo = *(const u32x4*)ss;
      const uint32_t hi = ss[4];
      out[0] = __builtin_amdgcn_alignbyte(lo[1], lo[0], sh);
      out[1] = __builtin_amdgcn_alignbyte(lo[2], lo[1], sh);
      out[2] = __builtin_amdgcn_alignbyte(lo[3], lo[2], sh);
      out[3] = __builtin_amdgcn_alignbyte(hi, lo[3], sh);
      }
    else
      {
      // unused ranges inside the vector: where every byte comes from first (LDS only), then the sixteen loads together
      uint32_t mm = m, so[16];
#pragma unroll
      for (int qq = 0; qq < 16; ++qq)
        {
        const uint32_t ob = o + (uint32_t)qq;
        while (mm < H && e[mm] <= ob) { shift = cum[mm]; ++mm; }
        so[qq] = ob + shift;
        }
      uint8_t by[16];
#pragma unroll
      for (int qq = 0; qq < 16; ++qq)
        by[qq] = s[so[qq]];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        out[qq] = (uint32_t)by[4 * qq] | ((uint32_t)by[4 * qq + 1] << 8) | ((uint32_t)by[4 * qq + 2] << 16) | ((uint32_t)by[4 * qq + 3] << 24);
      }
    patch(out, o, m);
    dd[t] = out;
    }
  if (threadIdx.x < len - done)
    d[done + threadIdx.x] = out_byte(done + threadIdx.x);
  }

// ---- compare: segment slots against an existing payload -------------------------------------------------
// The decoders check themselves by coding what they decoded and comparing with what they were given (shim.hip): the coder is a
// deterministic function of the values, so equal payloads mean equal values.  Same walk as the gather, reading both sides.
struct ComparePay { const uint8_t* p[3]; uint32_t size[3]; };

__global__ void __launch_bounds__(256) k_fpc32_compare(const uint8_t* __restrict__ slots, size_t slot_stride, uint32_t segcap, uint32_t S,
                                                       const uint32_t* __restrict__ segbytes, const uint32_t* __restrict__ segoff,
                                                       const uint32_t* __restrict__ sizes, ComparePay pay, uint32_t* __restrict__ status,
                                                       uint32_t flag)
  {
  const uint32_t g = blockIdx.x, c = blockIdx.y;
  if (sizes[c] != pay.size[c])
    {
    if (g == 0 && threadIdx.x == 0) atomicOr(status, flag << c);
    return;
    }
  const uint32_t len = segbytes[(size_t)c * S + g];
  const uint8_t* a = slots + (size_t)c * slot_stride + (size_t)g * segcap;       // 256-byte aligned
  const uint8_t* b = pay.p[c] + segoff[(size_t)c * S + g];
  bool diff = false;
  const uint32_t words = len >> 2;
  const uint32_t sh = (uint32_t)((uintptr_t)b & 3u);
  const uint32_t* bw = (const uint32_t*)(b - sh);                                // aligned dwords around b
  for (uint32_t t = threadIdx.x; t < words; t += 256u)
    {
    const uint32_t x = ((const uint32_t*)a)[t];
    // the high dword is only read when its first byte belongs to this segment (sh != 0): never a whole dword past the payload
    const uint32_t y = __builtin_amdgcn_alignbyte(sh ? bw[t + 1u] : 0u, bw[t], sh);
    diff = diff || x != y;
    }
  for (uint32_t t = 4u * words + threadIdx.x; t < len; t += 256u)
    diff = diff || a[t] != b[t];
  if (diff)
    atomicOr(status, flag << c);
  }

// n == 0: undefined in the reference; defined as header + one full pad group
__global__ void k_fpc32_empty(uint8_t* out, size_t out_stride, uint32_t* sizes)
  {
  uint8_t* o = out + (size_t)blockIdx.x * out_stride;
  const uint8_t bts[16] = { 0x25, 0, 0, 0, 0, 0x24, 0x92, 0x49, 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int i = 0; i < 16; ++i) o[i] = bts[i];
  sizes[blockIdx.x] = 16;
  if (blockIdx.x == 0)
    sizes[3] = 0;
  }

struct Plan { uint32_t L, S, segcap, nch; size_t rows, slot_stride, off_summ, off_inc, off_chmax, off_chhas, off_seen, off_nrec, off_recs, off_segbytes, off_segoff, off_flags, off_slots, total; };

// Before resolve_xchg is trusted on a device, the device shows that its LDS unit applies the active lanes of one ds_wrxchg_rtn_b32
// in increasing lane order (the property the kernel rests on; see resolve_xchg): 1024 waves x 96 exchanges with random keys (1 to
// 1024 distinct ones, per workgroup), random exec masks and four waves per workgroup, every returned value and every final entry
// compared with what ballots say it has to be.  ~0.3 ms, once per device and process.
__global__ void __launch_bounds__(256) k_fpc32_xchg_selftest(uint32_t rounds, uint32_t* __restrict__ bad)
  {
  __shared__ uint32_t T[4][1024];
  __shared__ uint32_t shadow[4][1024];
  __shared__ uint32_t bits[4][32], sbits[4][32];       // the one-sweep encoder's "class written" bitmap: ds_or_rtn_b32, same question
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  for (uint32_t i = lane; i < 1024u; i += 64u) { T[w][i] = 0u; shadow[w][i] = 0u; }
  if (lane < 32u) { bits[w][lane] = 0u; sbits[w][lane] = 0u; }
  __syncthreads();
  const uint32_t nkeys = 1u << (blockIdx.x % 11u);
  uint32_t x = (blockIdx.x * 0x9E3779B9u) ^ (threadIdx.x * 0x85EBCA6Bu) ^ 0x2545F491u;
  uint32_t wrong = 0;
  for (uint32_t r = 0; r < rounds; ++r)
    {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t k = (x >> 8) & (nkeys - 1u);
    const bool active = ((x >> 3) & 7u) != 0u || (r & 15u) == 0u;
    const uint32_t val = ((r + 1u) << 6) | lane;
    uint32_t expect = shadow[w][k];
    bool last = true, lower = false;
    for (uint32_t j = 0; j < 64u; ++j)
      {
      const uint32_t kj = (uint32_t)__shfl((int)k, (int)j, 64), vj = (uint32_t)__shfl((int)val, (int)j, 64);
      const bool aj = __shfl((int)active, (int)j, 64) != 0;
      if (aj && kj == k) { if (j < lane) { expect = vj; lower = true; } if (j > lane) last = false; }
      }
    const uint32_t bit = 1u << (k & 31u);
    const bool expect_bit = lower || (sbits[w][k >> 5] & bit) != 0u;
    uint32_t old = 0, oldbits = 0;
    if (active)
      {
      old = __hip_atomic_exchange(&T[w][k], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      oldbits = __hip_atomic_fetch_or(&bits[w][k >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    if (active && (old != expect || ((oldbits & bit) != 0u) != expect_bit)) ++wrong;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (active && last) shadow[w][k] = val;
    if (active) atomicOr(&sbits[w][k >> 5], bit);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (active && last && T[w][k] != val) ++wrong;
    }
  if (wrong)
    atomicAdd(bad, wrong);
  }

// Which code sweep?  TRICO_FPC32_ATOMIC=1: tagged 64-bit entries (resolve_atomic; checks the lane order in every step, given up for
// good once a step has seen a violation).  Otherwise the exchange sweep (resolve_xchg) if this device passes the test above, unless
// TRICO_FPC32_XCHG=0; else ballots (resolve).  Measured on the MI355X (profiles/r03_fpc32_encode_experiments.txt): tagged entries make
// a noisy component alone 15 % faster but halve the waves per CU (9,120 B of LDS per wave), which costs the smooth components more.
static bool g_atomic_distrusted = false;
bool fpc32_use_atomic()
  {
  static const bool env_atomic = [] { const char* e = getenv("TRICO_FPC32_ATOMIC"); return e && e[0] == '1'; }();
  return env_atomic && !g_atomic_distrusted;
  }

bool lane_order_tested()
  {
  static std::mutex mu;
  static int state[32] = { 0 };                  // per device: 0 not tested, 1 passed, 2 failed
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32)
    return false;
  std::lock_guard<std::mutex> lock(mu);
  if (state[dev] == 0)
    {
    uint32_t* d_bad = nullptr;
    uint32_t h_bad = 1;
    state[dev] = 2;
    if (hipMalloc(&d_bad, 4) == hipSuccess)
      {
      hipStream_t st = current_stream();
      if (hipMemsetAsync(d_bad, 0, 4, st) == hipSuccess)
        {
        hipLaunchKernelGGL(k_fpc32_xchg_selftest, dim3(256), dim3(256), 0, st, 96u, d_bad);
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess && h_bad == 0u)
          state[dev] = 1;
        }
      (void)hipFree(d_bad);
      }
    }
  return state[dev] == 1;
  }

bool fpc32_xchg_usable()
  {
  static const bool env_off = [] { const char* e = getenv("TRICO_FPC32_XCHG"); return e && e[0] == '0'; }();
  return !env_off && lane_order_tested();
  }

Plan make_plan(uint32_t n, int arity)
  {
  static int waves = 0;
  if (!waves)
    {
    const char* e = getenv("TRICO_FPC32_WAVES");      // tuning knob: waves per sweep
    waves = e ? atoi(e) : (fpc32_use_atomic() ? 3840 : 7680);   // 15 (tagged tables: 9,120 B of LDS per wave) or 30 waves per CU (4,960 B)
    if (waves < 3) waves = 3;
    }
  Plan p;
  const uint32_t target = (uint32_t)waves / (uint32_t)arity;
  uint64_t L = ((uint64_t)n + target - 1) / target;
  L = (L + 127) / 128 * 128;                    // multiple of 64 * ISPLIT
  if (L < 1024) L = 1024;
  p.L = (uint32_t)L;
  p.S = (uint32_t)(((uint64_t)n + L - 1) / L);
  if (p.S == 0) p.S = 1;
  p.segcap = (uint32_t)align_up(5 + 4 * (size_t)L + 3 * ((size_t)L / 8) + 16 + 280, 256);
  p.nch = (p.S + CH - 1) / CH;
  p.rows = (size_t)p.S * arity;
  p.slot_stride = (size_t)p.S * p.segcap;
  size_t o = 0;
  p.off_summ = o;      o += align_up(p.rows * ROW * 4, 256);
  p.off_inc = o;       o += align_up(p.rows * ROW * 4, 256);
  p.off_chmax = o;     o += align_up((size_t)p.nch * arity * TAB * 4, 256);
  p.off_chhas = o;     o += align_up((size_t)p.nch * arity * TAB * 4, 256);
  p.off_seen = o;      o += align_up(p.rows * SEENW * 4, 256);
  p.off_nrec = o;      o += align_up(p.rows * 4, 256);
  p.off_recs = o;      o += align_up(p.rows * RCAP * 16, 256);
  p.off_segbytes = o;  o += align_up(p.rows * 4, 256);
  p.off_segoff = o;    o += align_up(p.rows * 4, 256);
  p.off_flags = o;     o += 256;
  p.off_slots = o;     o += p.slot_stride * arity;
  p.total = o + 256;
  return p;
  }

} // namespace

size_t fpc32_encode_workspace(uint32_t n, int arity)
  {
  return make_plan(n, arity).total;
  }

void fpc32_distrust_atomic() { g_atomic_distrusted = true; }

bool lds_lane_order_ok() { return lane_order_tested(); }

int fpc32_code_sweep_mode() { return fpc32_use_atomic() ? M_TAGGED : (fpc32_xchg_usable() ? M_XCHG : M_BALLOT); }

int launch_fpc32_encode(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                        uint8_t* d_ws, size_t ws_bytes, bool allow_atomic)
  {
  hipStream_t st = current_stream();
  if (n == 0)
    {
    hipLaunchKernelGGL(k_fpc32_empty, dim3(arity), dim3(1), 0, st, d_out, out_stride, d_sizes);
    return hip_ok(hipGetLastError(), "k_fpc32_empty") ? 1 : 0;
    }
  const Plan p = make_plan(n, arity);
  if (p.total > ws_bytes)
    {
    set_error("fpc32 encode: workspace too small");
    return 0;
    }
  uint32_t* summ = (uint32_t*)(d_ws + p.off_summ);
  uint32_t* inc = (uint32_t*)(d_ws + p.off_inc);
  uint32_t* chmax = (uint32_t*)(d_ws + p.off_chmax);
  uint32_t* segbytes = (uint32_t*)(d_ws + p.off_segbytes);
  uint32_t* segoff = (uint32_t*)(d_ws + p.off_segoff);
  uint8_t* slots = d_ws + p.off_slots;
  const uint32_t* src = (const uint32_t*)d_src;
  const unsigned threads = 64u * (unsigned)arity;
  uint32_t* nrec = (uint32_t*)(d_ws + p.off_nrec);
  uint32_t* flags = (uint32_t*)(d_ws + p.off_flags);
  static const int tile = [] { const char* e = getenv("TRICO_FPC32_TILE"); return e ? atoi(e) : 0; }();
  static const int sweeps = [] { const char* e = getenv("TRICO_FPC32_SWEEPS"); return e ? atoi(e) : 2; }();
  static const uint32_t prio_mode = [] {
    const char* e = getenv("TRICO_FPC32_PRIO"), * l = getenv("TRICO_FPC32_LAG");
    return (e ? (uint32_t)atoi(e) & 255u : 8u) | ((l ? (uint32_t)atoi(l) & 255u : 1u) << 8);
  }();
  const int mode = allow_atomic ? fpc32_code_sweep_mode() : M_BALLOT;
  if (sweeps == 1 && mode == M_XCHG && !tile)
    {
    // one sweep (see k_fpc32_sweep1): the input is read once, the values that depend on what a segment comes in with are coded afterwards
    uint32_t* outSeen = (uint32_t*)(d_ws + p.off_seen);
    uint32_t* chhas = (uint32_t*)(d_ws + p.off_chhas);
    uint32_t* recs = (uint32_t*)(d_ws + p.off_recs);
    hipLaunchKernelGGL(k_fpc32_sweep1, dim3(p.S), dim3(threads), (size_t)arity * LDSW_1 * 4 + 16, st, src, n, arity, p.L, p.S, summ, outSeen,
                       slots, p.slot_stride, p.segcap, segbytes, nrec, recs, prio_mode);
    const unsigned colblocks = ((unsigned)arity * TAB + 255u) / 256u;
    hipLaunchKernelGGL(k_fpc32_pscan_a, dim3(colblocks, p.nch), dim3(256), 0, st, summ, outSeen, p.S, arity, chmax, chhas, flags);
    hipLaunchKernelGGL(k_fpc32_pscan_b, dim3(colblocks, p.nch), dim3(256), 0, st, summ, outSeen, p.S, arity, chmax, chhas, inc);
    hipLaunchKernelGGL(k_fpc32_fixup, dim3(p.S, arity), dim3(256), 0, st, src, arity, p.S, inc, segbytes, nrec, recs);
    hipLaunchKernelGGL(k_fpc32_offsets, dim3(arity), dim3(1024), 0, st, segbytes, p.S, segoff, d_sizes, flags);
    return hip_ok(hipGetLastError(), "fpc32 encode kernels (one sweep)") ? 1 : 0;
    }
  if (tile && arity == 3)
    {
    // tile variants (bit 0: sweep A, bit 1: sweep C): one wave per segment walks all components (the interleaved array is read
    // once per sweep)
    if (tile & 1)
      hipLaunchKernelGGL(k_fpc32_index_t<3>, dim3(p.S), dim3(64), (size_t)3 * TAB * 4, st, src, n, p.L, summ);
    else
      {
      if (!hip_ok(hipMemsetAsync(summ, 0, p.rows * ROW * 4, st), "memset(summ)"))
        return 0;
      hipLaunchKernelGGL(k_fpc32_index<false>, dim3(p.S, ISPLIT), dim3(192), ((size_t)BLOCK_V * 3 + (size_t)3 * LDSW_A) * 4, st, src, n, arity, p.L, summ);
      }
    const unsigned colblocks = (3u * TAB + 255u) / 256u;
    hipLaunchKernelGGL(k_fpc32_scan_a, dim3(colblocks, p.nch), dim3(256), 0, st, summ, p.S, arity, chmax, flags, nrec);
    hipLaunchKernelGGL(k_fpc32_scan_b, dim3(colblocks, p.nch), dim3(256), 0, st, summ, p.S, arity, chmax, inc);
    if (!(tile & 2))
      hipLaunchKernelGGL(k_fpc32_code<M_BALLOT>, dim3(p.S), dim3(192), (size_t)3 * LDSW_C * 4 + 16, st,
                         src, n, arity, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, flags, 8u);
    else if (mode == M_TAGGED)
      hipLaunchKernelGGL((k_fpc32_code_t<3, M_TAGGED>), dim3(p.S), dim3(64), (size_t)3 * LDSW_CA * 4, st,
                         src, n, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, flags);
    else if (mode == M_XCHG)
      hipLaunchKernelGGL((k_fpc32_code_t<3, M_XCHG>), dim3(p.S), dim3(64), (size_t)3 * LDSW_C * 4, st,
                         src, n, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, flags);
    else
      hipLaunchKernelGGL((k_fpc32_code_t<3, M_BALLOT>), dim3(p.S), dim3(64), (size_t)3 * LDSW_C * 4, st,
                         src, n, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, flags);
    hipLaunchKernelGGL(k_fpc32_offsets, dim3(arity), dim3(1024), 0, st, segbytes, p.S, segoff, d_sizes, flags);
    return hip_ok(hipGetLastError(), "fpc32 encode kernels (tile)") ? 1 : 0;
    }
  if (!hip_ok(hipMemsetAsync(summ, 0, p.rows * ROW * 4, st), "memset(summ)"))
    return 0;
  const size_t lds_a = ((size_t)BLOCK_V * arity + (size_t)arity * LDSW_A) * 4;
  if (((uintptr_t)src & 15u) == 0)
    hipLaunchKernelGGL(k_fpc32_index<true>, dim3(p.S, ISPLIT), dim3(threads), lds_a, st, src, n, arity, p.L, summ);
  else
    hipLaunchKernelGGL(k_fpc32_index<false>, dim3(p.S, ISPLIT), dim3(threads), lds_a, st, src, n, arity, p.L, summ);
  const unsigned colblocks = ((unsigned)arity * TAB + 255u) / 256u;
  hipLaunchKernelGGL(k_fpc32_scan_a, dim3(colblocks, p.nch), dim3(256), 0, st, summ, p.S, arity, chmax, flags, nrec);
  hipLaunchKernelGGL(k_fpc32_scan_b, dim3(colblocks, p.nch), dim3(256), 0, st, summ, p.S, arity, chmax, inc);
  if (mode == M_TAGGED)
    hipLaunchKernelGGL(k_fpc32_code<M_TAGGED>, dim3(p.S), dim3(threads), (size_t)arity * LDSW_CA * 4 + 16, st,
                       src, n, arity, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, flags, prio_mode);
  else if (mode == M_XCHG)
    hipLaunchKernelGGL(k_fpc32_code<M_XCHG>, dim3(p.S), dim3(threads), (size_t)arity * LDSW_C * 4 + 16, st,
                       src, n, arity, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, flags, prio_mode);
  else
    hipLaunchKernelGGL(k_fpc32_code<M_BALLOT>, dim3(p.S), dim3(threads), (size_t)arity * LDSW_C * 4 + 16, st,
                       src, n, arity, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, flags, prio_mode);
  hipLaunchKernelGGL(k_fpc32_offsets, dim3(arity), dim3(1024), 0, st, segbytes, p.S, segoff, d_sizes, flags);
  return hip_ok(hipGetLastError(), "fpc32 encode kernels") ? 1 : 0;
  }

// Moves component `c` of the last launch_fpc32_encode (same n, arity, workspace) from its segment slots to
// `d_dst`, contiguous.  This is the one copy the payload needs to reach its place in the archive.
int launch_fpc32_gather(uint32_t n, int arity, int c, const uint8_t* d_ws, uint8_t* d_dst)
  {
  if (n == 0)
    return 1;           // the empty-stream kernel wrote the payload in place
  const Plan p = make_plan(n, arity);
  const uint32_t* segbytes = (const uint32_t*)(d_ws + p.off_segbytes);
  const uint32_t* segoff = (const uint32_t*)(d_ws + p.off_segoff);
  const uint8_t* slots = d_ws + p.off_slots;
  GatherDst dst = { { d_dst, nullptr, nullptr } };
  hipLaunchKernelGGL(k_fpc32_gather, dim3(p.S, 1), dim3(256), 0, current_stream(), slots + (size_t)c * p.slot_stride, (size_t)0,
                     p.segcap, p.S, segbytes + (size_t)c * p.S, segoff + (size_t)c * p.S, dst, (const uint32_t*)(d_ws + p.off_nrec),
                     (const uint32_t*)(d_ws + p.off_recs), arity, c);
  return hip_ok(hipGetLastError(), "k_fpc32_gather") ? 1 : 0;
  }

// Compares the payloads of the last launch_fpc32_encode (same n, arity, workspace) with `d_pay` / `sizes`, without moving them:
// bit (flag << c) of *d_status is set if component c differs in size or bytes.
int launch_fpc32_compare(uint32_t n, int arity, const uint8_t* d_ws, const uint32_t* d_sizes, const uint8_t* const d_pay[3],
                         const uint32_t sizes[3], uint32_t* d_status, uint32_t flag)
  {
  if (n == 0)
    return 1;
  const Plan p = make_plan(n, arity);
  const uint32_t* segbytes = (const uint32_t*)(d_ws + p.off_segbytes);
  const uint32_t* segoff = (const uint32_t*)(d_ws + p.off_segoff);
  ComparePay pay;
  for (int c = 0; c < 3; ++c)
    {
    pay.p[c] = c < arity ? d_pay[c] : nullptr;
    pay.size[c] = c < arity ? sizes[c] : 0u;
    }
  hipLaunchKernelGGL(k_fpc32_compare, dim3(p.S, arity), dim3(256), 0, current_stream(), d_ws + p.off_slots, p.slot_stride, p.segcap, p.S,
                     segbytes, segoff, d_sizes, pay, d_status, flag);
  return hip_ok(hipGetLastError(), "k_fpc32_compare") ? 1 : 0;
  }

// All components in one launch, each to its own destination (the archive writer knows all of them up front).
int launch_fpc32_gather_all(uint32_t n, int arity, const uint8_t* d_ws, uint8_t* const d_dst[3])
  {
  if (n == 0)
    return 1;
  const Plan p = make_plan(n, arity);
  const uint32_t* segbytes = (const uint32_t*)(d_ws + p.off_segbytes);
  const uint32_t* segoff = (const uint32_t*)(d_ws + p.off_segoff);
  GatherDst dst = { { d_dst[0], arity > 1 ? d_dst[1] : nullptr, arity > 2 ? d_dst[2] : nullptr } };
  hipLaunchKernelGGL(k_fpc32_gather, dim3(p.S, arity), dim3(256), 0, current_stream(), d_ws + p.off_slots, p.slot_stride,
                     p.segcap, p.S, segbytes, segoff, dst, (const uint32_t*)(d_ws + p.off_nrec), (const uint32_t*)(d_ws + p.off_recs), arity, 0);
  return hip_ok(hipGetLastError(), "k_fpc32_gather") ? 1 : 0;
  }

} // namespace trico
