// k_fpc32_encode.hip — the TWO-sweep coder for 32-bit floating-point streams (gfx950, wave64) and the launchers of the float encoder.
//
// Same result as k_fpc32_sweep.hip - trico_transpose_*_aos_to_soa (transpose_aos_to_soa.c:8-16, 48-56) fused with
// trico_compress(..., 4, 10) (fpsc.c:86-210) - by a formulation that shares nothing with it where it matters: the tables every
// segment comes in with are computed first, and run starts find their predecessor with ballots (resolve), not with a lane-ordered
// LDS exchange.  It is
//   * what the decoders' self-check re-encodes with (shim.hip: fpc_selfcheck_launch), so the two ways of finding a predecessor
//     check each other on every decode;
//   * what a stream is coded with again when the one-sweep coder raised a flag (a sampled step out of lane order, a payload equal
//     to its sentinel), and what codes everything on a device that fails the lane-order test.
//
// Structure: each component stream is cut into S contiguous segments of L values (L % 128 == 0); one wave owns one
// (segment, component) and walks it 64 values per step.
//   sweep A  (k_fpc32_index):  classes only.  The run-end lane of every class run does an LDS ds_max of its value index into a
//            16+1024 entry table -> "last writer index per class" of the segment.
//   scan     (k_fpc32_scan_*): prefix-max over segments per class = the table every segment starts with, as value indices
//            (0 = never written = the reference's zeroed table).
//   sweep C  (k_fpc32_code):   loads the incoming table (payloads gathered from the input by index), then per step: the latest
//            earlier value of my class is the previous lane inside a run of equal classes (runs span steps through the carry);
//            run starts are resolved with one ballot per class bit, only in steps that have any (see the comment block above
//            code_step).  Codes, residual lengths, wave prefix sums (DPP scan), 3-byte group headers (DPP or-reduce), bytes staged
//            in a linear LDS buffer and flushed as aligned dwords into the segment's slot.
//   offsets  (k_fpc32_offsets): exclusive scan of the segment byte counts per component; the flags of the sweep.
//   gather   (k_fpc32_sweep.hip): slot -> final position.
// TRICO_FPC32_SWEEPS=2 makes this the coder of every stream (measurements; with the exchange in place of the ballots where the
// device passes the test: that is round 3's default encoder).
#include "fpc32_common.hpp"
#include <mutex>

namespace trico {

using namespace fpc32;

namespace {

constexpr int LDSW_A = TAB;                        // per-wave LDS words, sweep A
constexpr int PF = 8;          // steps (of 64 values) whose loads are kept in flight per wave
#ifndef TRICO_PFC
#define TRICO_PFC 8
#endif
constexpr int PFC = TRICO_PFC; // ... in the code sweep

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


// ---- scan: incoming index table of segment g = max over earlier segments -----------------------------
__global__ void __launch_bounds__(256) k_fpc32_scan_a(const uint32_t* __restrict__ summ, uint32_t S, int arity,
                                                      uint32_t* __restrict__ chmax, uint32_t* __restrict__ nrec)
  {
  if (blockIdx.x == 0 && blockIdx.y == 0)
    for (uint32_t r = threadIdx.x; r < S * (uint32_t)arity; r += 256u)
      nrec[r] = 0u;                                          // two sweeps: no deferred values, no flags
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


// The same lookup with 32-bit entries and NO tags: ds_wrxchg_rtn_b32.  The reference codes a value by reading the entry of its class
// and then writing its own payload there (fpsc.c:133-143), value after value; an exchange is exactly that pair, and the LDS unit of
// gfx950 applies the active lanes of one exchange instruction in increasing lane order (tools/ubench/lds_xchg_order.hip: 131 M
// instructions, 1 to 1024 keys, random exec masks, four waves per workgroup on the same LDS: every lane got the payload of the nearest
// lower active lane of its key, or what earlier steps had left, and the entry ended with the highest lane's payload).  Only the lanes
// where a run of equal classes starts or ends take part (inside a run the previous lane is the predecessor, DPP): a start takes what
// comes back; a start that is not an end leaves its payload there for a moment, and the end lane of its run - a higher lane of the
// same instruction - replaces it.  Ten ballots, a table read, a ds_bpermute and a table write become one LDS instruction, and the
// table stays at 4 bytes per entry (30 waves per CU).  The order is not documented, so the library tests it on
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


// one step: 64 values starting at index i0 (FULL: all of them inside the segment)
// MODE: how run starts find their predecessor - 0 ballots (resolve), 2 exchange (resolve_xchg)
constexpr int M_BALLOT = 0, M_XCHG = 2, M_SWEEP = 3;

struct StepRegs                                  // per-lane values a step carries from phase to phase
  {
  uint32_t v, a, s, k1, k2, p1, p2;
  bool act, st1, st2, any1, any2;
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
template <bool FULL>
__device__ __forceinline__ void step_tail(const StepRegs& r, uint32_t i, uint32_t i_end, uint32_t n, uint8_t* __restrict__ stage, Sweep& sw,
                                          const LaneK& lk)
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


// MODE: resolve (ballots) or resolve_xchg
template <int MODE>
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(8, 8))) k_fpc32_code(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L, uint32_t S,
                                                    const uint32_t* __restrict__ inc, uint8_t* __restrict__ slots, size_t slot_stride,
                                                    uint32_t segcap, uint32_t* __restrict__ segbytes, uint32_t prio_mode)
  {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x;
  constexpr int TW = TAB;                              // table words
  volatile uint32_t* prog = lds + arity * LDSW_C;      // [4] progress of the component waves (prio_mode 8)
  if (prio_mode >= 1u && prio_mode <= 3u && (uint32_t)c == prio_mode - 1u)
    __builtin_amdgcn_s_setprio(3);
  uint32_t* T = lds + c * LDSW_C;                      // [TAB] payload table
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
        T[k] = pay;
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
  }


// ---- offsets: exclusive scan of segment sizes per component (one workgroup per component) -------------
// sizes[c] = payload bytes of component c; sizes[3 + c] = the flags its waves raised (FLAG_*; they travel in the upper half of the
// record counts, so that nothing has to be zeroed before a sweep), read back with the sizes.
// rectot[c] = the records (deferred values) of component c: the gather takes the components in the order of these.
__global__ void __launch_bounds__(1024) k_fpc32_offsets(const uint32_t* __restrict__ segbytes, uint32_t S, int arity, uint32_t* __restrict__ segoff,
                                                        uint32_t* __restrict__ sizes, const uint32_t* __restrict__ nrec, uint32_t* __restrict__ rectot,
                                                        uint32_t* __restrict__ mirror)
  {
  __shared__ uint32_t wsum[16], wfl[16], wrec[16];
  const uint32_t c = blockIdx.x;
  const uint32_t per = (S + 1023u) / 1024u;
  const uint32_t g0 = threadIdx.x * per < S ? threadIdx.x * per : S, g1 = (g0 + per < S) ? g0 + per : S;
  uint32_t sum = 0, f = 0, h = 0;
  for (uint32_t g = g0; g < g1; ++g)
    {
    sum += segbytes[(size_t)c * S + g];
    const uint32_t w = nrec[(size_t)g * arity + c];
    f |= w >> 16;
    h += w & 0xffffu;
    }
  // (one scan inside the wave, sixteen partial sums through LDS: two barriers instead of the twenty of a ladder over 1024 threads)
  const uint32_t incl = wave_scan_incl(sum), hin = wave_scan_incl(h);
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint64_t fany = __ballot(f != 0u);
  uint32_t fw = 0;
  if (fany)
    for (uint32_t b = 0; b < 16u; ++b)
      fw |= __ballot((f >> b) & 1u) ? (1u << b) : 0u;
  if (lane == 63u)
    {
    wsum[wv] = incl;
    wrec[wv] = hin;
    wfl[wv] = fw;
    }
  __syncthreads();
  uint32_t before = 0, total = 0, fl = 0, recs = 0;
  for (uint32_t w = 0; w < 16u; ++w)
    {
    before += w < wv ? wsum[w] : 0u;
    total += wsum[w];
    fl |= wfl[w];
    recs += wrec[w];
    }
  uint32_t run = before + incl - sum;
  for (uint32_t g = g0; g < g1; ++g)
    {
    segoff[(size_t)c * S + g] = run;
    run += segbytes[(size_t)c * S + g];
    }
  if (threadIdx.x == 1023u)
    {
    sizes[c] = total;
    sizes[3u + c] = fl;
    rectot[c] = recs;
    if (mirror)
      {
      // (pinned host memory: the caller reads the sizes there when the stream is done - no copy launch behind this kernel)
      mirror[c] = total;
      mirror[3u + c] = fl;
      }
    }
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
__global__ void k_fpc32_empty(uint8_t* out, size_t out_stride, uint32_t* sizes, uint32_t* mirror)
  {
  uint8_t* o = out + (size_t)blockIdx.x * out_stride;
  const uint8_t bts[16] = { 0x25, 0, 0, 0, 0, 0x24, 0x92, 0x49, 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int i = 0; i < 16; ++i) o[i] = bts[i];
  sizes[blockIdx.x] = 16;
  sizes[3u + blockIdx.x] = 0;
  if (mirror)
    {
    mirror[blockIdx.x] = 16;
    mirror[3u + blockIdx.x] = 0;
    }
  }


// Before resolve_xchg is trusted on a device, the device shows that its LDS unit applies the active lanes of one ds_wrxchg_rtn_b32
// in increasing lane order (the property the kernel rests on; see resolve_xchg): 1024 waves x 96 exchanges with random keys (1 to
// 1024 distinct ones, per workgroup), random exec masks and four waves per workgroup, every returned value and every final entry
// compared with what ballots say it has to be.  ~0.3 ms, once per device and process.
__global__ void __launch_bounds__(256) k_fpc32_xchg_selftest(uint32_t rounds, uint32_t* __restrict__ bad)
  {
  __shared__ uint32_t T[4][1024];
  __shared__ uint32_t shadow[4][1024];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  for (uint32_t i = lane; i < 1024u; i += 64u) { T[w][i] = 0u; shadow[w][i] = 0u; }
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
    bool last = true;
    for (uint32_t j = 0; j < 64u; ++j)
      {
      const uint32_t kj = (uint32_t)__shfl((int)k, (int)j, 64), vj = (uint32_t)__shfl((int)val, (int)j, 64);
      const bool aj = __shfl((int)active, (int)j, 64) != 0;
      if (aj && kj == k) { if (j < lane) expect = vj; if (j > lane) last = false; }
      }
    uint32_t old = 0;
    if (active)
      old = __hip_atomic_exchange(&T[w][k], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (active && old != expect) ++wrong;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (active && last) shadow[w][k] = val;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (active && last && T[w][k] != val) ++wrong;
    }
  if (wrong)
    atomicAdd(bad, wrong);
  }

// Per device: 0 not tested, 1 the exchange is lane ordered, 2 it is not (or a sampled step of k_fpc32_sweep said so later).
static std::mutex g_order_mu;
static int g_order_state[32] = { 0 };

bool lane_order_tested()
  {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32)
    return false;
  std::lock_guard<std::mutex> lock(g_order_mu);
  int* state = g_order_state;
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
  static const bool env_off = [] { const char* e = tune_env("TRICO_FPC32_XCHG"); return e && e[0] == '0'; }();
  return !env_off && lane_order_tested();
  }

} // namespace

size_t fpc32_encode_workspace(uint32_t n, int arity)
  {
  return make_plan(n, arity).total;
  }

void fpc32_distrust_lane_order()
  {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32)
    return;
  std::lock_guard<std::mutex> lock(g_order_mu);
  g_order_state[dev] = 2;
  }

bool lds_lane_order_ok() { return lane_order_tested(); }

// What FPC32_CODER_AUTO runs: the one-sweep coder (3) on a device whose LDS exchange is lane ordered, else two sweeps with ballots (0).
// TRICO_FPC32_SWEEPS=2 (measurements): two sweeps, with the exchange where it is usable (2) - round 3's encoder.
int fpc32_code_sweep_mode()
  {
  static const int sweeps = [] { const char* e = tune_env("TRICO_FPC32_SWEEPS"); return e ? atoi(e) : 1; }();
  if (!fpc32_xchg_usable())
    return M_BALLOT;
  return sweeps == 2 ? M_XCHG : M_SWEEP;
  }

int launch_fpc32_encode(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                        uint8_t* d_ws, size_t ws_bytes, int coder, uint32_t* h_sizes)
  {
  hipStream_t st = current_stream();
  if (n == 0)
    {
    hipLaunchKernelGGL(k_fpc32_empty, dim3(arity), dim3(1), 0, st, d_out, out_stride, d_sizes, h_sizes);
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
  const int mode = coder == FPC32_CODER_BALLOT ? M_BALLOT : fpc32_code_sweep_mode();
  if (mode == M_SWEEP)
    {
    if (!launch_fpc32_sweep(src, n, arity, p, d_ws))
      return 0;
    hipLaunchKernelGGL(k_fpc32_offsets, dim3(arity), dim3(1024), 0, st, segbytes, p.S, arity, segoff, d_sizes, nrec, (uint32_t*)(d_ws + p.off_diag + 256), h_sizes);
    return hip_ok(hipGetLastError(), "k_fpc32_offsets") ? 1 : 0;
    }
  const uint32_t prio_mode = 8u | (1u << 8);      // priority by progress, at most one block of lag between the component waves (measured in round 3)
  if (!hip_ok(hipMemsetAsync(summ, 0, p.rows * ROW * 4, st), "memset(summ)"))
    return 0;
  const size_t lds_a = ((size_t)BLOCK_V * arity + (size_t)arity * LDSW_A) * 4;
  if (((uintptr_t)src & 15u) == 0)
    hipLaunchKernelGGL(k_fpc32_index<true>, dim3(p.S, ISPLIT), dim3(threads), lds_a, st, src, n, arity, p.L, summ);
  else
    hipLaunchKernelGGL(k_fpc32_index<false>, dim3(p.S, ISPLIT), dim3(threads), lds_a, st, src, n, arity, p.L, summ);
  const unsigned colblocks = ((unsigned)arity * TAB + 255u) / 256u;
  hipLaunchKernelGGL(k_fpc32_scan_a, dim3(colblocks, p.nch), dim3(256), 0, st, summ, p.S, arity, chmax, nrec);
  hipLaunchKernelGGL(k_fpc32_scan_b, dim3(colblocks, p.nch), dim3(256), 0, st, summ, p.S, arity, chmax, inc);
  if (mode == M_XCHG)
    hipLaunchKernelGGL(k_fpc32_code<M_XCHG>, dim3(p.S), dim3(threads), (size_t)arity * LDSW_C * 4 + 16, st,
                       src, n, arity, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, prio_mode);
  else
    hipLaunchKernelGGL(k_fpc32_code<M_BALLOT>, dim3(p.S), dim3(threads), (size_t)arity * LDSW_C * 4 + 16, st,
                       src, n, arity, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes, prio_mode);
  hipLaunchKernelGGL(k_fpc32_offsets, dim3(arity), dim3(1024), 0, st, segbytes, p.S, arity, segoff, d_sizes, nrec, (uint32_t*)(d_ws + p.off_diag + 256), h_sizes);
  return hip_ok(hipGetLastError(), "fpc32 encode kernels") ? 1 : 0;
  }

// Moves component `c` of the last launch_fpc32_encode (same n, arity, workspace) from its segment slots to
// `d_dst`, contiguous.  This is the one copy the payload needs to reach its place in the archive.
int launch_fpc32_gather(uint32_t n, int arity, int c, const uint8_t* d_ws, uint8_t* d_dst)
  {
  if (n == 0)
    return 1;           // the empty-stream kernel wrote the payload in place
  uint8_t* const dst[3] = { d_dst, nullptr, nullptr };
  return launch_fpc32_gather_rec(make_plan(n, arity), arity, c, 1, d_ws, dst);
  }

// Compares the payloads of the last launch_fpc32_encode with coder FPC32_CODER_BALLOT (same n, arity, workspace) with `d_pay` /
// `sizes`, without moving them: bit (flag << c) of *d_status is set if component c differs in size or bytes.
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

// All components in one launch, framed for the container (`u32 bytes, payload` per component) from d_first on, placed on the device by
// the sizes launch_fpc32_encode left in d_sizes: nothing here needs the host to have read them.
int launch_fpc32_gather_framed(uint32_t n, int arity, const uint8_t* d_ws, uint8_t* d_first, const uint32_t* d_sizes)
  {
  if (n == 0)
    return 0;
  uint8_t* const dst[3] = { d_first, nullptr, nullptr };
  return launch_fpc32_gather_rec(make_plan(n, arity), arity, 0, arity, d_ws, dst, d_sizes);
  }

// All components in one launch, each to its own destination (the archive writer knows all of them up front).
int launch_fpc32_gather_all(uint32_t n, int arity, const uint8_t* d_ws, uint8_t* const d_dst[3])
  {
  if (n == 0)
    return 1;
  return launch_fpc32_gather_rec(make_plan(n, arity), arity, 0, arity, d_ws, d_dst);
  }

} // namespace trico
