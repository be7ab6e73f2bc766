// k_fpc32_sweep.hip — the throughput encoder for 32-bit floating-point streams: ONE sweep over the input (gfx950, wave64).
//
// Replaces, fused: trico_transpose_xyz/uv_aos_to_soa (transpose_aos_to_soa.c:8-16, 48-56) and trico_compress(..., 4, 10)
// (fpsc.c:86-210) for every component of a vec3 / vec2 / scalar stream.
//
// Why this can be parallel and still bit-exact: the FCM hash of value i is a pure function of v[i-1]
// (top 4 bits) and the DFCM hash a pure function of the strides of v[i-1] and v[i-2], so every value belongs to a *class* known
// from the input alone, and the reference's table read for value i (fpsc.c:133-143) returns the payload (value / stride) of the
// latest earlier value of the same class, or 0.
//
// Structure: each component stream is cut into S contiguous segments of L values; one wave owns one (segment, component) and walks
// it 64 values per step; the component waves of a segment share a workgroup (they read the same cache lines).
//   k_fpc32_sweep   codes every segment WITHOUT knowing the tables it comes in with.  Inside a run of equal classes the
//                   predecessor is the previous lane (DPP; runs span steps through the registers the previous step left).  The
//                   lanes where a run starts or ends do ONE ds_wrxchg_rtn_b32 on the wave's payload table per predictor: the LDS
//                   unit applies the lanes of an instruction in lane order, which is the reference's read-then-write, value after
//                   value (tested on the device before use and checked again in sampled steps against ballots, see guard_*).
//                   A run start that gets the sentinel back has met the entry the segment came in with: its value is *deferred* -
//                   four zero bytes, code 0 and a record - and nothing else depends on it (a prediction only decides its own
//                   value's residual).  Steps whose 64 values all continue their runs with an exact prediction - the x / y of a
//                   grid, the flat parts of a scan - leave a constant 24- or 88-byte pattern with one LDS store and skip the
//                   byte layout altogether.  At its end the wave publishes its tables (= what the segment leaves behind).
//   k_fpc32_scanfix incoming payload of every (segment, class) = the entry of the nearest earlier segment that wrote the class; then
//                   per record: residual, length and code from the incoming entry; the segment's final size.
//   k_fpc32_offsets (k_fpc32_encode.hip) exclusive scan of the segment sizes.
//   k_fpc32_gather  slot -> final position: a plain copy for segments without records, else a pass through LDS that drops the
//                   unused bytes of the reserved fields and ORs residuals and codes in.
// HBM traffic: the input once (+ what the component waves re-fetch when they drift apart), the payload three times (slot write,
// gather read + write), tables and records.  No MFMA: integer bit-twiddling; algorithmic bytes per value = 4 + its payload share.
#include "fpc32_common.hpp"
#include <atomic>

namespace trico {
namespace fpc32 {

namespace {

// (sabotage switches for the tests exist in libtrico_testhooks.so only)
#ifdef TRICO_HIP_TEST_HOOKS
constexpr bool SWEEP_HOOK = true;
#else
constexpr bool SWEEP_HOOK = false;
#endif

constexpr int FLB = 512;                          // the staging area leaves in blocks of this many bytes (8 per lane)
constexpr int STAGE_LIVE = FLB + 280;             // < 512 unflushed + <= 280 of the step
constexpr int STAGE = STAGE_LIVE + 256;           // + 4 dump bytes per lane (compiled step only)
constexpr int LDSW = 1312;                        // per-wave LDS words: TAB + STAGE / 4 = 1302, rounded so that every wave's table starts
                                                  // at a multiple of 64 bytes (5,248 B: 10 workgroups of 3 waves per CU)
static_assert(LDSW >= TAB + STAGE / 4 && (LDSW * 4) % 64 == 0 && (TAB * 4) % 8 == 0, "LDS layout of a wave");

constexpr int STAGGER_DEFAULT = 300;                // Stagger (fpc32_common.hpp): first class + 30 %, last class - 30 % (measured: profiles/r06_sweep_stagger.txt)
constexpr uint32_t GUARD_STEPS = 64;              // steps of a sampled segment the guard codes again
constexpr uint32_t GUARD_CAP = GUARD_SLOT;        // bytes they can produce (a step: 24 header + 256 residual bytes), rounded

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) volatile uint8_t lds_vu8;

struct LaneK                                      // per-lane constants
  {
  uint32_t lane, lane4;
  uint32_t sh3, grp3, c4sh;                       // 3 * (lane % 8), 3 * (lane / 8), 4 << sh3
  uint32_t pat5;                                  // this lane's dword of the 88-byte pattern of a step of 64 exact DFCM hits
  bool lead;
  };

struct Sweep                                      // running state of a wave between steps
  {
  uint32_t vp, sp, s1p, a1p, a2p;                 // per lane, of the previous step: value, stride, stride of the lane below, table addresses
  uint32_t pend1, pend2;                          // 1: the table writes of the previous step's last value are still pending (compiled step only)
  uint32_t posl, flushed;                         // bytes staged in LDS / bytes already in the slot
  uint32_t nrec;                                  // records written so far
  uint32_t flags;                                 // FLAG_* raised by this wave
  uint64_t sent;                                  // lanes that stored the sentinel as a payload, in any step of sweep_blocks_asm
  };

constexpr uint32_t DUMP = STAGE_LIVE + 1;          // a lane's four dump bytes: stage + DUMP - 1 + 4 * lane

// Global memory goes through buffer descriptors: scalar base + scalar offset + the lane's constant offset, so that neither the
// rolling prefetch nor the flush costs a vector instruction for its address, and loads past the end of the array return zeros
// instead of needing bounds code.  (Word 3 as for every raw 32-bit buffer on gfx9.)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint64_t bytes)
  {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes), 0x00020000);
  }
// the same descriptor as four words in scalar registers (what sweep_blocks_asm takes as an operand)
__device__ __forceinline__ u32x4 make_desc(const void* base, uint64_t bytes)
  {
  const uint64_t b = (uint64_t)(uintptr_t)base;
  u32x4 d;
  d[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  d[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
  d[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes));
  d[3] = 0x00020000u;
  return d;
  }

// A full block of the staging area (512 bytes, 8 per lane) goes to the slot, what is behind it (< 280 bytes) moves to the front.
// The compiled step does this on the spot; sweep_blocks_asm issues the LDS reads at the end of a step and uses them in the next.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void flush_sync(Sweep& sw, uint8_t* __restrict__ stage, rsrc_t slot, const LaneK& lk)
  {
  if (sw.posl >= (uint32_t)FLB)
    {
    uint32_t* stw = (uint32_t*)stage;
    const u32x2 w = *(const u32x2*)(stw + 2u * lk.lane);
    const uint32_t t0 = stw[128u + lk.lane], t1 = stw[192u + lk.lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // (streaming stores: the slot is read again only by the gather, and the lines should not push the input out of the L2)
    __builtin_amdgcn_raw_buffer_store_b64(w, slot, 8u * lk.lane, sw.flushed, 2);
    stw[lk.lane] = t0;
    stw[64u + lk.lane] = t1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    sw.flushed += (uint32_t)FLB;
    sw.posl -= (uint32_t)FLB;
    }
  }

// ---- run starts ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ lds_u32* lds_at(uint32_t address) { return (lds_u32*)(uintptr_t)address; }

// What the reference's table read hands a lane of `part` (the lanes where a run starts or ends) for the entry at LDS address `ad`
// (class k, B bits): the payload of the nearest lower lane of `part` with the same class, else what the table holds.  Also tells
// the lane whether it is the highest of its class (its payload is the table's afterwards).  The lanes of my class: one ballot per
// class bit.  This is how the guard workgroups resolve run starts - no LDS exchange, no assumption about the order of anything.
template <int B>
__device__ __forceinline__ uint32_t ballot_lookup(uint32_t ad, uint32_t k, bool part, uint32_t pay, uint32_t lane, bool& owner)
  {
  const uint32_t tv = part ? *lds_at(ad) : 0u;
  uint64_t same = __ballot(part);
#pragma unroll 1
  for (int b = 0; b < B; ++b)
    {
    const bool bit = (k >> b) & 1u;
    const uint64_t m = __ballot(bit);
    same &= bit ? m : ~m;
    }
  const uint64_t lo = same & ((1ull << lane) - 1ull);
  const uint32_t q = (uint32_t)__builtin_amdgcn_ds_bpermute((63 - __builtin_clzll(lo | 1ull)) << 2, (int)pay);
  owner = part && (same >> lane) == 1ull;
  return lo ? q : tv;
  }

// The reference codes a value by reading the entry of its class and then writing its own payload there (fpsc.c:133-143), value
// after value; an exchange is exactly that pair, and the LDS unit of gfx950 applies the active lanes of one ds_wrxchg_rtn_b32 in
// increasing lane order (tools/ubench/lds_xchg_order.hip: 131 M instructions, 1 to 1024 keys, random exec masks, four waves per
// workgroup on the same LDS).  Only the lanes where a run of equal classes starts or ends take part: a start takes what comes back;
// a start that is not an end leaves its payload there for the moment it takes the end lane of its run - a higher lane of the same
// instruction - to replace it.  The table starts as SENT everywhere: a start that gets SENT back has met what the segment came in
// with (ft*).  The order is not documented, hence: the device test before first use (k_fpc32_xchg_selftest), and in every encode a
// few workgroups code the beginning of sampled segments again with BALLOT = true - ballots instead of the exchange - for the
// fix-up kernel to compare (FLAG_ORDER -> the host codes the stream again with the ballot coder).
// a1 / a2 are the LDS addresses of the values' table entries.
template <bool FULL, bool D1, bool D2, bool BALLOT, bool HOOK>
__device__ __forceinline__ void resolve_h(uint32_t a1, uint32_t a2, bool st1, bool st2, bool act, uint32_t v, uint32_t s,
                                          uint32_t& p1, uint32_t& p2, bool& ft1, bool& ft2, uint32_t t1abs,
                                          Sweep& sw, const LaneK& lk, uint32_t sabotage)
  {
  // pending writes of the previous step's last value (its lane 63 still holds address and payload)
  if (lk.lane == 63u)
    {
    if (D1 && sw.pend1) *lds_at(sw.a1p) = sw.vp;
    if (D2 && sw.pend2) *lds_at(sw.a2p) = sw.sp;
    }
  uint64_t sent = 0;
  if (D1 && sw.pend1) sent |= __ballot(lk.lane == 63u && sw.vp == SENT);
  if (D2 && sw.pend2) sent |= __ballot(lk.lane == 63u && sw.sp == SENT);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const bool en1 = D1 && (FULL || act) && a1 != dpp_shl1(0xffffffffu, a1);     // last lane of a run (lane 63 always)
  const bool en2 = D2 && (FULL || act) && a2 != dpp_shl1(0xffffffffu, a2);
  const bool part1 = D1 && (st1 || en1), part2 = D2 && (st2 || en2);
  uint32_t old1 = 0, old2 = 0;
  if (BALLOT)
    {
    bool own1 = false, own2 = false;
    if (D1) old1 = ballot_lookup<4>(a1, (a1 >> 2) & 15u, part1, v, lk.lane, own1);
    if (D2) old2 = ballot_lookup<10>(a2, (a2 - (t1abs + 64u)) >> 2, part2, s, lk.lane, own2);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (own1) *lds_at(a1) = v;
    if (own2) *lds_at(a2) = s;
    }
  else
    {
    if (part1)
      old1 = __hip_atomic_exchange(lds_at(a1), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (part2)
      old2 = __hip_atomic_exchange(lds_at(a2), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (HOOK && !BALLOT && sabotage)
    {
    // test hook (libtrico_testhooks.so only): pretend the LDS unit served the lanes in another order - a start in the upper half
    // of the wave gets something else than its predecessor left
    if (D1 && st1 && lk.lane >= 32u) old1 ^= 0x100u;
    if (D2 && st2 && lk.lane >= 32u) old2 ^= 0x100u;
    }
  if (D1) { p1 = st1 ? old1 : p1; ft1 = st1 && old1 == SENT; sent |= __ballot(v == SENT); }
  if (D2) { p2 = st2 ? old2 : p2; ft2 = st2 && old2 == SENT; sent |= __ballot(s == SENT); }
  if (sent)
    sw.flags |= FLAG_SENTINEL;
  if (D1) sw.pend1 = 0u;
  if (D2) sw.pend2 = 0u;
  }

struct RecSink { uint32_t* recs; };               // this wave's record list (RECW words each)

// residual selection, byte layout of the step, bytes into the staging area, records of the deferred values (flush_end must have run)
template <bool FULL>
__device__ __forceinline__ void step_tail(uint32_t v, uint32_t a, uint32_t a1, uint32_t a2, uint32_t p1, uint32_t p2, bool ft1, bool ft2,
                                          bool act, uint32_t i, uint32_t i_end, uint32_t n, uint32_t t1abs, uint32_t sbase, Sweep& sw,
                                          const LaneK& lk, const RecSink& sink)
  {
  // residual selection (fpsc.c:146-189).  With m = leading zero BYTES (m1 in 0..4, m2 in 0..3: a DFCM residual takes at least
  // one byte): n = 4 - m, the DFCM residual is chosen iff n2 < n1 (n2 >= 1, so this already says n1 > 1), the length is the
  // smaller n
  const uint32_t x1 = v ^ p1, x2 = v ^ (a + p2);
  const uint32_t m1 = (uint32_t)__clz((int)x1) >> 3;
  const uint32_t m2 = (uint32_t)__clz((int)(x2 | 1u)) >> 3;
  const bool use2 = m2 > m1;
  uint32_t len = 4u - max(m1, m2);
  uint32_t x = use2 ? x2 : x1;
  uint32_t code = len + (use2 ? 4u : 0u);
  bool slot = true;
  if (!FULL)
    {
    slot = act || (i_end == n && i < ((n + 7u) & ~7u));      // value or tail padding slot (fpsc.c:196-204)
    if (!act)
      {
      code = slot ? 1u : 0u;
      len = code;
      x = 0u;
      }
    }
  const bool hole = ft1 || ft2;
  const uint64_t hm = __ballot(hole);
  if (hm)
    {
    // four zero bytes and code 0 for now; the fix-up codes the value when the incoming entry is known, the gather drops what the
    // residual does not need
    len = hole ? 4u : len;
    code = hole ? 0u : code;
    x = hole ? 0u : x;
    }
  // byte layout of the step: [hdr g0][residuals 0..7][hdr g1][residuals 8..15]...
  const uint32_t incl = wave_scan_incl(len);
  const uint32_t pre = incl - len;
  uint32_t bc = code << lk.sh3;
  bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
  bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
  bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0x141, 0xf, 0xf, true);    // row_half_mirror
  const uint32_t hq = (sbase + sw.posl) + lk.grp3 + pre;     // LDS address of my group's header (if I lead it), my residual starts at hq + 3
  {
  // A residual is stored as the 4 big-endian bytes that END at its last byte, most significant first, in this order: the leading
  // zero bytes of lane l land on bytes owned by lower lanes or on group headers of the same step, all of which are written by a
  // LATER instruction (proof in DESIGN.md), so there is no per-byte predicate.  volatile keeps four byte stores in program order
  // (merged into one dword store, lanes would race).
  const uint32_t re = len ? hq + len : (sbase + DUMP) + lk.lane4;      // residual end - 3
  lds_vu8* vs = (lds_vu8*)(uintptr_t)re;
  vs[-1] = (uint8_t)(x >> 24);
  vs[0] = (uint8_t)(x >> 16);
  vs[1] = (uint8_t)(x >> 8);
  vs[2] = (uint8_t)x;
  }
  {
  const bool lead = FULL ? lk.lead : (lk.lead && slot);
  lds_vu8* hs = (lds_vu8*)(uintptr_t)(lead ? hq : (sbase + DUMP) + lk.lane4);
  hs[0] = (uint8_t)(bc >> 16);
  hs[1] = (uint8_t)(bc >> 8);
  hs[2] = (uint8_t)bc;
  }
  if (hm)
    {
    // record (fpc32_common.hpp): where the four bytes and the group's header are in the slot, the value and its predecessor,
    // which classes are open, the prediction that is known if only one is open
    const uint32_t hq_lead = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lk.lane & ~7u) << 2), (int)hq);
    const uint32_t idx = sw.nrec + popc_below(hm);
    if (hole)
      {
      u32x4 w;
      w[0] = (sw.flushed + (hq - sbase) + 3u) | ((lk.lane & 7u) << REC_GI_SHIFT) | (ft1 ? REC_FT1 : 0u) | (ft2 ? REC_FT2 : 0u);
      w[1] = sw.flushed + (hq_lead - sbase);
      w[2] = v;
      w[3] = a;
      uint32_t* r = sink.recs + RECW * idx;
      *(u32x4*)r = w;
      r[4] = ft1 ? (ft2 ? 0u : p2) : p1;
      r[5] = a2 - (t1abs + 64u);
      }
    sw.nrec += (uint32_t)__popcll(hm);
    }
  const uint32_t hdr = FULL ? 24u : 3u * ((uint32_t)__popcll(__ballot(slot)) >> 3);
  sw.posl += hdr + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }

// resolve + tail, compiled from C++
template <bool FULL, bool BALLOT, bool HOOK>
__device__ __forceinline__ void step_general(uint32_t v, uint32_t a, uint32_t s, uint32_t s1, uint32_t a1, uint32_t a2, bool st1, bool st2,
                                             bool any1, bool any2, bool act, uint32_t i, uint32_t i_end, uint32_t n, uint32_t t1abs,
                                             uint32_t sbase, Sweep& sw, const LaneK& lk, const RecSink& sink, uint32_t sabotage)
  {
  uint32_t p1 = a, p2 = s1;                                  // inside a run: previous value / previous stride
  bool ft1 = false, ft2 = false;
  if (any1 && any2)
    resolve_h<FULL, true, true, BALLOT, HOOK>(a1, a2, st1, st2, act, v, s, p1, p2, ft1, ft2, t1abs, sw, lk, sabotage);
  else if (any1)
    {
    resolve_h<FULL, true, false, BALLOT, HOOK>(a1, a2, st1, st2, act, v, s, p1, p2, ft1, ft2, t1abs, sw, lk, sabotage);
    sw.pend2 = 1u;
    }
  else if (any2)
    {
    resolve_h<FULL, false, true, BALLOT, HOOK>(a1, a2, st1, st2, act, v, s, p1, p2, ft1, ft2, t1abs, sw, lk, sabotage);
    sw.pend1 = 1u;
    }
  else
    sw.pend1 = sw.pend2 = 1u;
  step_tail<FULL>(v, a, a1, a2, p1, p2, ft1, ft2, act, i, i_end, n, t1abs, sbase, sw, lk, sink);
  }

// one step of one component, compiled from C++ throughout: 64 values starting at index i0 (FULL: all of them inside the segment
// and the stream).  The last step of a stream takes this path, the guard workgroups (BALLOT), and every step under TRICO_FPC32_ASM=0.
template <bool FULL, bool BALLOT, bool HOOK>
__device__ __forceinline__ void code_step(uint32_t v, uint32_t i0, uint32_t i_end, uint32_t n, uint32_t t1abs,
                                          uint8_t* __restrict__ stage, rsrc_t slot, Sweep& sw, const LaneK& lk,
                                          const RecSink& sink, uint32_t sabotage)
  {
  const uint32_t i = i0 + lk.lane;
  const bool act = FULL || i < i_end;
  // classes (fpsc.c:76-84 with e1 = 4, e2 = 10) as table addresses: FCM from v[i-1], DFCM from the strides of v[i-1] and v[i-2]
  const uint32_t a = shr1_across(sw.vp, v);                  // v[i-1]
  const uint32_t s = v - a;                                  // stride of v[i]
  const uint32_t s1 = shr1_across(sw.sp, s);                 // stride of v[i-1]
  const uint32_t s2 = shr1_across(sw.s1p, s1);               // stride of v[i-2]
  uint32_t a1 = ((a >> 26) & 0x3cu) | t1abs;                 // (the table starts at a multiple of 64 bytes)
  uint32_t a2 = (((((s2 >> 22) & 31u) << 5) ^ (s1 >> 22)) << 2) + (t1abs + 64u);
  if (!FULL && !act)
    a1 = a2 = 0xffffffffu;
  // run starts: my class differs from the class of the value before me (lane 0: the previous step's lane 63)
  bool st1 = a1 != shr1_across(sw.a1p, a1);
  bool st2 = a2 != shr1_across(sw.a2p, a2);
  if (!FULL) { st1 = st1 && act; st2 = st2 && act; }
  const bool any1 = __ballot(st1) != 0ull, any2 = __ballot(st2) != 0ull;
  const uint32_t sbase = (uint32_t)(uintptr_t)(lds_u8*)stage;        // LDS address of the staging area (uniform)
  // Every value continues its runs and every prediction is exact: the 64 values are a constant - eight groups of a zero header
  // (FCM hit: code 0, no byte), or of header b6 db 6d and eight zero bytes (DFCM hit: code 5, residual byte 0x00; it is the choice
  // only if the FCM residual needs more than one byte, fpsc.c:146-189).  One LDS store, no byte layout.
  bool u0 = false, u5 = false;
  if (FULL && !any1 && !any2)
    {
    const uint32_t x1 = v ^ a, x2 = v ^ (a + s1);
    u0 = __ballot(x1 != 0u) == 0ull;
    u5 = !u0 && (__ballot(x2 != 0u) | __ballot(x1 < 256u)) == 0ull;
    }
  if (u0 || u5)
    {
    const uint32_t ad = (sbase + sw.posl) + lk.lane4;
    if (lk.lane < (u5 ? 22u : 6u))
      asm volatile("ds_write_b32 %0, %1" :: "v"(ad), "v"(u5 ? lk.pat5 : 0u) : "memory");         // (not dword aligned: gfx950 executes it)
    sw.posl += u5 ? 88u : 24u;
    sw.pend1 = sw.pend2 = 1u;
    }
  else
    step_general<FULL, BALLOT, HOOK>(v, a, s, s1, a1, a2, st1, st2, any1, any2, act, i, i_end, n, t1abs, sbase, sw, lk, sink, sabotage);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  flush_sync(sw, stage, slot, lk);
  sw.vp = v; sw.sp = s; sw.s1p = s1; sw.a1p = a1; sw.a2p = a2;
  }

// Replay of one full step in front of a segment: the table writes of its 64 values, nothing else (see SW_RSTEP).  The compiled
// form: the guard workgroups (BALLOT) and TRICO_FPC32_ASM=0.
template <bool BALLOT>
__device__ __forceinline__ void replay_step(uint32_t v, uint32_t t1abs, Sweep& sw, const LaneK& lk)
  {
  const uint32_t a = shr1_across(sw.vp, v);
  const uint32_t s = v - a;
  const uint32_t s1 = shr1_across(sw.sp, s);
  const uint32_t s2 = shr1_across(sw.s1p, s1);
  const uint32_t a1 = ((a >> 26) & 0x3cu) | t1abs;
  const uint32_t a2 = (((((s2 >> 22) & 31u) << 5) ^ (s1 >> 22)) << 2) + (t1abs + 64u);
  const bool st1 = a1 != shr1_across(sw.a1p, a1);
  const bool st2 = a2 != shr1_across(sw.a2p, a2);
  uint32_t p1 = a, p2 = s1;
  bool ft1 = false, ft2 = false;
  // (both predictors every time: a step without a run start writes its last value's entries, nothing stays pending)
  resolve_h<true, true, true, BALLOT, false>(a1, a2, st1, st2, true, v, s, p1, p2, ft1, ft2, t1abs, sw, lk, 0u);
  sw.vp = v; sw.sp = s; sw.s1p = s1; sw.a1p = a1; sw.a2p = a2;
  }

__device__ __forceinline__ LaneK lane_constants(uint32_t lane)
  {
  LaneK lk;
  lk.lane = lane;
  lk.lane4 = 4u * lane;
  lk.sh3 = 3u * (lane & 7u);
  lk.grp3 = 3u * (lane >> 3);
  lk.c4sh = 4u << lk.sh3;
  lk.lead = (lane & 7u) == 0u;
  uint32_t w = 0;
  for (uint32_t b = 0; b < 4u; ++b)
    {
    const uint32_t q = (4u * lane + b) % 11u;
    const uint32_t by = q == 0u ? 0xb6u : q == 1u ? 0xdbu : q == 2u ? 0x6du : 0u;      // eight codes 5 = 0xb6db6d, then eight zero bytes
    w |= by << (8u * b);
    }
  lk.pat5 = w;
  return lk;
  }

// (the compiler takes what an asm statement returns for divergent, scalar registers or not: these tell it otherwise)
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uni(uint64_t x) { return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | uni((uint32_t)x); }

// ---- the full steps, written out: one asm statement for all whole blocks of a segment --------------------------------------------
// Why: the kernel is bound by what ONE wave can issue.  A wave issues at most one instruction per ~4 cycles, scalar or vector,
// branches cost several of those, and the step the compiler makes of C++ around asm pieces (round 4) spent more scalar and branch
// instructions on glue than vector instructions on the coder (SQ counters: 128 M scalar against 102 M vector per sweep of the
// benchmark mesh; ~12 branches per step) - and waited for memory it had asked for six steps ahead, because the flush and record
// stores sit in the same in-order queue as the prefetch loads and the compiler's vmcnt knows nothing of stores inside asm.
// So the loop is one piece of hand-scheduled code with these rules:
//   * a step is general or uniform.  The GENERAL step always resolves both predictors with the LDS exchange (exec = the lanes where a
//     run of equal classes starts or ends; lane 63 always ends one): there is no "pending table write" state, no variant without
//     FCM run starts, and deferred values cost three selects, not a branch;
//   * the UNIFORM step (the step before had no run start and only exact hits: the rows of a grid, flat parts of a scan) checks
//     that this one is the same (9 vector instructions) and stores a constant pattern; the table is brought up to date when such
//     a stretch ends (SW_LEAVE).  A general step without run starts looks whether the next one may try (SW_ENTRY);
//   * the staging area leaves in blocks of 512 bytes - at most one per step, so there is no second case.  Whether a block is full
//     is a scalar compare at the end of the step (the staging position is kept as an LDS address); only then are the two LDS reads
//     issued and the bookkeeping done, and the stores follow at the start of the next step under exec = "a block is pending";
//   * loads and stores share one in-order queue.  EVERY step issues its prefetch load and the block store (exec = 0 when no block is
//     pending; the four-step piece issues four loads and four stores), so when step j waits, at least twelve memory instructions are
//     younger than the load of six steps ago that it needs: s_waitcnt vmcnt(12) guarantees that load.  The two record stores are
//     issued only where there is a record; they make the wait a little longer than necessary.  (The variant in which every step
//     issued them too, with exec = 0, and waited for exactly vmcnt(24), was slower: 326 against 276 us.)
//   * eight value registers rotate (value of the step, of the step before, six loads in flight): the loop body is eight steps;
//   * a record is 15 vector instructions (the header's position comes from two DPP moves, not from a cross-lane read).
// Hazards the assembler does not see (gfx950): a DPP source written by a vector instruction needs two instructions in between,
// v_readlane one, a vector instruction reading an SGPR pair / VCC written by a vector instruction two.  EXEC is all ones between
// the pieces.  Fixed registers: v[54:55] / v[56:57] the words of the staging area read for the next step's flush, v[58:63] the
// record being written (also the four progress words in the lag check).
#define SW_DPPF " row_mask:0xf bank_mask:0xf\n"
#define SW_DPPB " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

// what the previous step left to flush: block to the slot, tail to the front (exec = pend ? all : none)
#define SW_FLUSH_STORES \
  "buffer_store_dwordx2 v[56:57], %[lane8], %[slr], %[floff] offen nt\n" \
  "ds_write2st64_b32 %[stw], v54, v55 offset1:1\n" \
  "s_mov_b32 %[pend], 0\n"

// a stretch of uniform steps ends (or the loop does): the registers of its last step that were not kept - stride, stride before it,
// table addresses - and the table writes of its last value (lane 63), which are the only ones of the stretch that last
#define SW_LEAVE(VP) \
  "s_cmp_eq_u32 %[ust], 88\n" \
  "s_cselect_b32 %[tms], %[ustr], 0\n" \
  "v_mov_b32 %[t3], 0\n" \
  "v_mov_b32 %[sp], %[tms]\n" \
  "v_mov_b32 %[s1p], %[tms]\n" \
  "v_mov_b32_dpp %[t3], " VP " wave_shr:1" SW_DPPF \
  "s_lshr_b32 %[cand], %[tms], 22\n" \
  "s_and_b32 %[total], %[cand], 31\n" \
  "v_lshrrev_b32 %[t3], 26, %[t3]\n" \
  "s_lshl_b32 %[total], %[total], 5\n" \
  "v_and_or_b32 %[a1p], %[t3], 60, %[t1s]\n" \
  "s_xor_b32 %[total], %[total], %[cand]\n" \
  "s_lshl_b32 %[total], %[total], 2\n" \
  "s_add_u32 %[total], %[total], %[t2s]\n" \
  "v_mov_b32 %[a2p], %[total]\n" \
  "s_mov_b32 exec_lo, 0\n" \
  "s_brev_b32 exec_hi, 1\n" \
  "ds_write_b32 %[a1p], " VP "\n" \
  "ds_write_b32 %[a2p], %[sp]\n" \
  "v_cmp_eq_u32_e32 vcc, %[ksent], " VP "\n" \
  "s_or_b64 %[sent], %[sent], vcc\n" \
  "v_cmp_eq_u32_e32 vcc, %[ksent], %[sp]\n" \
  "s_mov_b64 exec, -1\n" \
  "s_or_b64 %[sent], %[sent], vcc\n" \
  "s_mov_b32 %[ust], 0\n" \
  "s_nop 1\n"

// the flips of the test hook (libtrico_testhooks.so only): a run start in the upper half of the wave gets another prediction
#define SW_HOOK \
  "s_and_b64 %[tm64], %[st1], %[hk]\n" \
  "v_xor_b32 %[t3], 0x100, %[t1]\n" \
  "s_nop 1\n" \
  "v_cndmask_b32_e64 %[t1], %[t1], %[t3], %[tm64]\n" \
  "s_and_b64 %[tm64], %[st2], %[hk]\n" \
  "v_xor_b32 %[t3], 0x100, %[t2v]\n" \
  "s_nop 1\n" \
  "v_cndmask_b32_e64 %[t2v], %[t2v], %[t3], %[tm64]\n"

// one step: J label suffix, V its values, VP the values of the step before, LD the register the load of the step six ahead goes to.
// Temporaries: A = v[i-1]; t0 s2 / x1 / reversed residual; t1 class before me / exchange result / FCM prediction; t2v the same
// for DFCM; t3, t4 scratch / x2 / scan; t5 leading bytes / length; t6 leading bytes / header bits; t7 residual; t8 byte address;
// t9 my group's header - 1.
#define SW_RESIDUAL_STORES \
  "s_mov_b64 exec, %[tm64]\n" \
  "ds_write_b8 %[t8], %[t0]\n"                                       /* most significant byte first, in this order (see step_tail) */ \
  "ds_write_b8_d16_hi %[t8], %[t7] offset:1\n" \
  "ds_write_b8_d16_hi %[t8], %[t0] offset:2\n" \
  "ds_write_b8 %[t8], %[t7] offset:3\n"
#define SW_QUAD_TRY(J) \
  "s_cmp_lg_u32 %[ust], 0\n" \
  "s_cbranch_scc1 .Lq" J "_%=\n" \
  ".Lt" J "_%=:\n"
#define SW_STEP(J, V, VP, LD, HOOKTXT) \
  "buffer_load_dword " LD ", %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "s_waitcnt vmcnt(12)\n" \
  "s_cmp_lg_u32 %[ust], 0\n" \
  "s_cbranch_scc1 .Lu" J "_%=\n" \
  ".Lg" J "_%=:\n" \
  /* classes and run starts */ \
  "v_mov_b32_dpp v61, " VP " wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t0], %[s1p] wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[s1p], %[sp] wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t1], %[a1p] wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t2v], %[a2p] wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp v61, " V " wave_shr:1" SW_DPPF                    /* A = v[i-1] */ \
  "v_sub_u32 %[sp], " V ", v61\n"                                   /* stride of v[i] */ \
  "v_lshrrev_b32 %[t3], 26, v61\n" \
  "v_and_or_b32 %[a1p], %[t3], 60, %[t1s]\n"                         /* FCM entry: table + 4 * (top four bits of v[i-1]) */ \
  "v_mov_b32_dpp %[s1p], %[sp] wave_shr:1" SW_DPPF                   /* stride of v[i-1] */ \
  "s_nop 0\n" \
  "v_mov_b32_dpp %[t1], %[a1p] wave_shr:1" SW_DPPF \
  "v_lshrrev_b32 %[t4], 22, %[s1p]\n" \
  "v_mov_b32_dpp %[t0], %[s1p] wave_shr:1" SW_DPPF                   /* stride of v[i-2] */ \
  "v_cmp_ne_u32_e64 %[st1], %[a1p], %[t1]\n" \
  "v_lshrrev_b32 %[t3], 17, %[t0]\n" \
  "v_bitop3_b32 %[t3], %[t3], %[t4], %[k3e0] bitop3:0x6c\n"          /* ((s2 >> 17) & 0x3e0) ^ (s1 >> 22): the DFCM class */ \
  "v_lshl_add_u32 %[a2p], %[t3], 2, %[t2s]\n" \
  "s_lshr_b64 vcc, %[st1], 1\n"                                      /* a run ends where the next lane starts one ... */ \
  "s_bitset1_b32 vcc_hi, 31\n"                                       /* (... lane 63 always; also the second wait state of the move below) */ \
  "v_mov_b32_dpp %[t2v], %[a2p] wave_shr:1" SW_DPPF \
  "v_cmp_ne_u32_e64 %[st2], %[a2p], %[t2v]\n" \
  /* run starts: exchange by the lanes where a run starts or ends */ \
  "s_or_b64 exec, vcc, %[st1]\n" \
  "ds_wrxchg_rtn_b32 %[t1], %[a1p], " V "\n" \
  "v_cmp_eq_u32_e32 vcc, %[ksent], " V "\n" \
  "s_or_b64 %[sent], %[sent], vcc\n" \
  "s_lshr_b64 vcc, %[st2], 1\n" \
  "s_bitset1_b32 vcc_hi, 31\n" \
  "s_or_b64 exec, vcc, %[st2]\n" \
  "ds_wrxchg_rtn_b32 %[t2v], %[a2p], %[sp]\n" \
  "v_cmp_eq_u32_e32 vcc, %[ksent], %[sp]\n" \
  "s_or_b64 %[sent], %[sent], vcc\n" \
  "s_cmp_lg_u32 %[pend], 0\n" \
  "s_cselect_b64 exec, -1, 0\n" \
  "s_waitcnt lgkmcnt(0)\n"                                           /* ONE wait per step: the exchanges, and the staging words read at the end of the step before */ \
  SW_FLUSH_STORES \
  "s_mov_b64 exec, -1\n" \
  "v_cndmask_b32_e64 %[t1], v61, %[t1], %[st1]\n"                   /* FCM prediction: inside a run the previous value */ \
  "v_cndmask_b32_e64 %[t2v], %[s1p], %[t2v], %[st2]\n"               /* DFCM prediction: inside a run the previous stride */ \
  "v_cmp_eq_u32_e32 vcc, %[ksent], %[t1]\n" \
  "s_and_b64 %[ft1], vcc, %[st1]\n"                                  /* starts that met an entry nobody wrote in this segment */ \
  "v_cmp_eq_u32_e32 vcc, %[ksent], %[t2v]\n" \
  "s_and_b64 %[ft2], vcc, %[st2]\n" \
  "s_or_b64 %[hole], %[ft1], %[ft2]\n" \
  HOOKTXT \
  /* residual selection (fpsc.c:146-189): DFCM iff its residual is shorter; byte layout; byte stores */ \
  "v_xor_b32 %[t0], " V ", %[t1]\n" \
  "v_add_u32 %[t3], v61, %[t2v]\n" \
  "v_xor_b32 %[t4], " V ", %[t3]\n" \
  "v_ffbh_u32 %[t5], %[t0]\n" \
  "v_ffbh_u32 %[t6], %[t4]\n" \
  "v_min_u32 %[t5], 32, %[t5]\n" \
  "v_bfe_u32 %[t6], %[t6], 3, 2\n"                                   /* leading zero bytes 0..3 (a DFCM residual takes at least one byte) */ \
  "v_lshrrev_b32 %[t5], 3, %[t5]\n"                                  /* leading zero bytes 0..4 */ \
  "v_cmp_gt_u32_e32 vcc, %[t6], %[t5]\n" \
  "v_max_u32 %[t3], %[t5], %[t6]\n" \
  "v_sub_u32 %[t5], 4, %[t3]\n"                                      /* length */ \
  "v_cndmask_b32_e32 %[t7], %[t0], %[t4], vcc\n" \
  "v_cndmask_b32_e32 %[t3], 0, %[c4sh], vcc\n"                       /* code = length, + 4 for DFCM: already at the code's place in the header */ \
  "v_lshl_or_b32 %[t6], %[t5], %[sh3], %[t3]\n" \
  "v_cndmask_b32_e64 %[t5], %[t5], 4, %[hole]\n"                     /* deferred values: four zero bytes, code 0 */ \
  "v_cndmask_b32_e64 %[t6], %[t6], 0, %[hole]\n" \
  "v_cndmask_b32_e64 %[t7], %[t7], 0, %[hole]\n" \
  "v_perm_b32 %[t0], %[t7], %[t7], %[kswap]\n"                       /* the residual, bytes reversed */ \
  "v_or_b32_dpp %[t6], %[t6], %[t6] quad_perm:[1,0,3,2]" SW_DPPB \
  "v_add_u32_dpp %[t4], %[t5], %[t5] row_shr:1" SW_DPPB \
  "v_cmp_ne_u32_e64 %[tm64], 0, %[t5]\n" \
  "v_or_b32_dpp %[t6], %[t6], %[t6] quad_perm:[2,3,0,1]" SW_DPPB \
  "v_add_u32_dpp %[t4], %[t4], %[t4] row_shr:2" SW_DPPB \
  "s_nop 0\n" \
  "v_or_b32_dpp %[t6], %[t6], %[t6] row_half_mirror" SW_DPPB \
  "v_add_u32_dpp %[t4], %[t4], %[t4] row_shr:4" SW_DPPB \
  "s_or_b64 vcc, %[st1], %[st2]\n" \
  "s_cselect_b32 %[cand], 0, 1\n"                                    /* no run start at all: the next step may be a uniform one */ \
  "v_add_u32_dpp %[t4], %[t4], %[t4] row_shr:8" SW_DPPB \
  "v_lshrrev_b32 %[t3], 8, %[t6]\n" \
  "s_nop 0\n" \
  "v_add_u32_dpp %[t4], %[t4], %[t4] row_bcast:15 row_mask:0xa bank_mask:0xf\n" \
  "s_nop 1\n" \
  "v_add_u32_dpp %[t4], %[t4], %[t4] row_bcast:31 row_mask:0xc bank_mask:0xf\n"     /* inclusive sum of the lengths */ \
  "v_add3_u32 %[t8], %[pa], %[grp3], %[t4]\n"                       /* the four bytes that END with my residual's last byte */ \
  "v_sub_u32 %[t9], %[t8], %[t5]\n"                                  /* my group's header - 1 (my residual begins 4 behind it) */ \
  "v_readlane_b32 %[total], %[t4], 63\n" \
  SW_RESIDUAL_STORES \
  "s_mov_b64 exec, %[lead]\n" \
  "ds_write_b8_d16_hi %[t9], %[t6] offset:1\n"                       /* three header bytes, big-endian, by the lanes that lead a group */ \
  "ds_write_b8 %[t9], %[t3] offset:2\n" \
  "ds_write_b8 %[t9], %[t6] offset:3\n" \
  "s_mov_b64 exec, -1\n" \
  "s_add_u32 %[total], %[total], 24\n" \
  /* records of the deferred values (fpc32_common.hpp) */ \
  "s_cmp_eq_u64 %[hole], 0\n" \
  "s_cbranch_scc1 .Lnr" J "_%=\n" \
  "s_sub_u32 %[tms], %[flushed], %[sbase]\n" \
  "s_mov_b64 vcc, %[hole]\n" \
  "v_mov_b32_dpp %[t3], %[t9] quad_perm:[0,0,0,0]" SW_DPPF           /* the group's header - 1: from its first lane */ \
  "v_cndmask_b32_e64 %[t4], 0, 1, %[ft1]\n" \
  "v_cndmask_b32_e64 %[t5], 0, 2, %[ft2]\n" \
  "v_mov_b32_dpp %[t3], %[t3] row_shr:4 row_mask:0xf bank_mask:0xa\n" \
  "v_or_b32 %[t4], %[t4], %[t5]\n" \
  "v_lshl_or_b32 %[t4], %[t4], 28, %[lanegi]\n" \
  "s_add_u32 %[tms], %[tms], 4\n" \
  "v_add3_u32 v58, %[t9], %[tms], %[t4]\n"                           /* w0: slot offset of the four bytes | gi | ft1 | ft2 */ \
  "s_sub_u32 %[tms], %[tms], 3\n" \
  "v_add_u32 v59, %[tms], %[t3]\n"                                   /* w1: slot offset of the group's header */ \
  "v_mov_b32 v60, " V "\n" \
  "v_cndmask_b32_e64 %[t5], %[t2v], 0, %[ft2]\n" \
  "v_cndmask_b32_e64 v62, %[t1], %[t5], %[ft1]\n"                    /* w4: the prediction that is known if only one is open */ \
  "v_subrev_u32 v63, %[t2s], %[a2p]\n"                               /* w5: 4 * DFCM class */ \
  "v_mbcnt_lo_u32_b32 %[t4], vcc_lo, 0\n" \
  "v_mbcnt_hi_u32_b32 %[t4], vcc_hi, %[t4]\n" \
  "v_add_lshl_u32 %[t4], %[t4], %[nrec], 5\n" \
  "s_bcnt1_i32_b64 %[tms], %[hole]\n" \
  "s_add_u32 %[nrec], %[nrec], %[tms]\n" \
  "s_mov_b64 exec, %[hole]\n" \
  "buffer_store_dwordx4 v[58:61], %[t4], %[rcr], 0 offen\n" \
  "buffer_store_dwordx2 v[62:63], %[t4], %[rcr], 0 offen offset:16\n" \
  "s_mov_b64 exec, -1\n" \
  ".Lnr" J "_%=:\n" \
  "s_cmp_lg_u32 %[cand], 0\n" \
  "s_cbranch_scc1 .Lce" J "_%=\n" \
  /* end of the step: bytes staged; the reads of the next step's flush; is a block full? */ \
  ".Lend" J "_%=:\n" \
  "s_add_u32 %[pa], %[pa], %[total]\n" \
  "s_cmp_ge_u32 %[pa], %[lim]\n"                                     /* 512 bytes or more are staged */ \
  "s_cbranch_scc0 .Lnf" J "_%=\n" \
  "ds_read_b64 v[56:57], %[stw8]\n" \
  "ds_read2st64_b32 v[54:55], %[stw] offset0:2 offset1:3\n" \
  "s_mov_b32 %[floff], %[flushed]\n" \
  "s_mov_b32 %[pend], 1\n" \
  "s_add_u32 %[flushed], %[flushed], 512\n" \
  "s_sub_u32 %[pa], %[pa], 512\n" \
  ".Lnf" J "_%=:\n"

// Four uniform steps at once (the steps J .. J + 3 of a block, J = 0 or 4, when the step before was uniform): the same tests as the
// uniform step, for the four value registers, one verdict; 352 bytes (or 96) of pattern with two stores; the four loads of the
// steps six ahead and four stores (one flush, three of nothing: the prefetch wait of later steps counts two memory instructions
// per step) - about 20 instructions per step instead of 45.  Nothing is changed before the verdict: a no goes to the ordinary
// step J.  A yes ends at the end-of-step code of step J + 3.
#define SW_QUAD_STEP88(V, AR) \
  "v_sub_u32 %[t3], " V ", " AR "\n"                                 /* stride */ \
  "v_subrev_u32 %[t4], %[ustr], " AR "\n"                            /* v[i-2], if the strides are what they were */ \
  "v_xor_b32 %[t3], %[ustr], %[t3]\n" \
  "v_xor_b32 %[t4], %[t4], " AR "\n" \
  "v_xor_b32 %[t6], " V ", " AR "\n" \
  "v_and_or_b32 %[t3], %[t4], %[kf0], %[t3]\n"                       /* another stride, or v[i-2] and v[i-1] in different FCM classes */ \
  "v_cmp_gt_u32_e32 vcc, 0x100, %[t6]\n"                            /* an FCM residual of one byte would win */ \
  "v_or_b32 %[t5], %[t5], %[t3]\n" \
  "s_or_b64 %[tm64], %[tm64], vcc\n"
#define SW_QUAD(J, JE, V0, V1, V2, V3, VPREV, LD0, LD1, LD2, LD3) \
  ".Lq" J "_%=:\n" \
  "s_waitcnt vmcnt(2)\n" \
  "v_mov_b32_dpp v61, " VPREV " wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t0], " V0 " wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t1], " V1 " wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t2v], " V2 " wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp v61, " V0 " wave_shr:1" SW_DPPF                    /* v[i-1] of the four steps */ \
  "v_mov_b32_dpp %[t0], " V1 " wave_shr:1" SW_DPPF \
  "v_mov_b32_dpp %[t1], " V2 " wave_shr:1" SW_DPPF \
  "v_mov_b32_dpp %[t2v], " V3 " wave_shr:1" SW_DPPF \
  "s_cmp_eq_u32 %[ust], 24\n" \
  "s_cbranch_scc1 .Lqz" J "_%=\n" \
  "v_mov_b32 %[t5], 0\n" \
  "s_mov_b64 %[tm64], 0\n" \
  SW_QUAD_STEP88(V0, "v61") \
  SW_QUAD_STEP88(V1, "%[t0]") \
  SW_QUAD_STEP88(V2, "%[t1]") \
  SW_QUAD_STEP88(V3, "%[t2v]") \
  "v_cmp_ne_u32_e32 vcc, 0, %[t5]\n" \
  "s_or_b64 %[tm64], %[tm64], vcc\n" \
  "s_cmp_lg_u64 %[tm64], 0\n" \
  "s_cbranch_scc1 .Lt" J "_%=\n" \
  "s_mov_b32 %[total], 352\n" \
  "s_mov_b64 %[tm64], 0xffffff\n"                                   /* dwords 64 .. 87 of the pattern */ \
  "v_mov_b32 %[t4], %[pat5]\n" \
  "v_mov_b32 %[t5], %[pat5b]\n" \
  "s_branch .Lqc" J "_%=\n" \
  ".Lqz" J "_%=:\n"                                                 /* behind exact FCM hits: every value equals the one before */ \
  "v_xor_b32 %[t3], " V0 ", v61\n" \
  "v_xor_b32 %[t4], " V1 ", %[t0]\n" \
  "v_xor_b32 %[t5], " V2 ", %[t1]\n" \
  "v_xor_b32 %[t6], " V3 ", %[t2v]\n" \
  "v_or3_b32 %[t3], %[t3], %[t4], %[t5]\n" \
  "v_or_b32 %[t4], %[t3], %[t6]\n" \
  "v_cmp_ne_u32_e32 vcc, 0, %[t4]\n" \
  "s_cmp_lg_u64 vcc, 0\n" \
  "s_cbranch_scc1 .Lt" J "_%=\n" \
  "s_mov_b32 %[total], 96\n"                                        /* 24 dwords of zero (t4 is zero in every lane) */ \
  "s_mov_b64 %[tm64], 0\n" \
  ".Lqc" J "_%=:\n" \
  /* yes: what the step before left to flush, the pattern, the loads, the stores */ \
  "s_waitcnt lgkmcnt(0)\n" \
  "s_cmp_lg_u32 %[pend], 0\n" \
  "s_cselect_b64 exec, -1, 0\n" \
  SW_FLUSH_STORES \
  "s_mov_b64 exec, -1\n" \
  "s_cmp_eq_u32 %[total], 96\n" \
  "s_cselect_b64 exec, 0xffffff, -1\n" \
  "v_add_u32 %[t3], %[pa], %[lane4p1]\n" \
  "ds_write_b32 %[t3], %[t4]\n"                                     /* (not dword aligned: gfx950 executes it) */ \
  "s_mov_b64 exec, %[tm64]\n" \
  "ds_write_b32 %[t3], %[t5] offset:256\n" \
  "s_mov_b64 exec, -1\n" \
  "buffer_load_dword " LD0 ", %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "buffer_load_dword " LD1 ", %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "buffer_load_dword " LD2 ", %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "buffer_load_dword " LD3 ", %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "s_mov_b64 exec, 0\n" \
  "buffer_store_dword %[t3], %[lane8], %[rcr], 0 offen\n" \
  "buffer_store_dword %[t3], %[lane8], %[rcr], 0 offen\n" \
  "buffer_store_dword %[t3], %[lane8], %[rcr], 0 offen\n" \
  "s_mov_b64 exec, -1\n" \
  "s_branch .Lend" JE "_%=\n"

// Replay: in front of its segment a wave runs over the last blocks of the segment before - classes, run starts and the exchange on the
// tables, no coding - so that a class written there has its true entry when the segment begins and its first value in the segment
// is not deferred.  (An entry comes from the latest earlier value of its class: if that one lies in the window, replaying the
// window gives exactly it; a class nobody wrote in the window stays "never written" and is deferred as before.  The table the
// segment publishes then also holds what the window wrote, which is still the latest payload of those classes at the segment's
// end.)  On the benchmark mesh 32 steps in front of 342 halve the deferred values of the noisy component (393 -> 180 per
// segment), and with them the fix-up and the slow path of the gather.  The loads keep the rhythm of the main loop (a store of
// nothing stands for its flush store).
#define SW_RSTEP(V, VP, LD) \
  "buffer_load_dword " LD ", %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "s_waitcnt vmcnt(12) lgkmcnt(0)\n" \
  "v_mov_b32_dpp v61, " VP " wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t0], %[s1p] wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[s1p], %[sp] wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t1], %[a1p] wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp %[t2v], %[a2p] wave_ror:1" SW_DPPF \
  "v_mov_b32_dpp v61, " V " wave_shr:1" SW_DPPF \
  "v_sub_u32 %[sp], " V ", v61\n" \
  "v_lshrrev_b32 %[t3], 26, v61\n" \
  "v_and_or_b32 %[a1p], %[t3], 60, %[t1s]\n" \
  "v_mov_b32_dpp %[s1p], %[sp] wave_shr:1" SW_DPPF \
  "s_nop 0\n" \
  "v_mov_b32_dpp %[t1], %[a1p] wave_shr:1" SW_DPPF \
  "v_lshrrev_b32 %[t4], 22, %[s1p]\n" \
  "v_mov_b32_dpp %[t0], %[s1p] wave_shr:1" SW_DPPF \
  "v_cmp_ne_u32_e64 %[st1], %[a1p], %[t1]\n" \
  "v_lshrrev_b32 %[t3], 17, %[t0]\n" \
  "v_bitop3_b32 %[t3], %[t3], %[t4], %[k3e0] bitop3:0x6c\n" \
  "v_lshl_add_u32 %[a2p], %[t3], 2, %[t2s]\n" \
  "s_lshr_b64 vcc, %[st1], 1\n" \
  "s_bitset1_b32 vcc_hi, 31\n" \
  "v_mov_b32_dpp %[t2v], %[a2p] wave_shr:1" SW_DPPF \
  "v_cmp_ne_u32_e64 %[st2], %[a2p], %[t2v]\n" \
  "s_or_b64 exec, vcc, %[st1]\n" \
  "ds_wrxchg_rtn_b32 %[t8], %[a1p], " V "\n" \
  "v_cmp_eq_u32_e32 vcc, %[ksent], " V "\n" \
  "s_or_b64 %[sent], %[sent], vcc\n" \
  "s_lshr_b64 vcc, %[st2], 1\n" \
  "s_bitset1_b32 vcc_hi, 31\n" \
  "s_or_b64 exec, vcc, %[st2]\n" \
  "ds_wrxchg_rtn_b32 %[t9], %[a2p], %[sp]\n" \
  "v_cmp_eq_u32_e32 vcc, %[ksent], %[sp]\n" \
  "s_mov_b64 exec, 0\n" \
  "buffer_store_dword %[t3], %[lane8], %[rcr], 0 offen\n" \
  "s_mov_b64 exec, -1\n" \
  "s_or_b64 %[sent], %[sent], vcc\n"

// the out-of-line parts of step J: the uniform step, the end of a uniform stretch, the look at a general step without run starts
#define SW_STEP_OOL(J, V, VP) \
  ".Lu" J "_%=:\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "s_cmp_lg_u32 %[pend], 0\n" \
  "s_cselect_b64 exec, -1, 0\n" \
  SW_FLUSH_STORES \
  "s_mov_b64 exec, -1\n" \
  "v_mov_b32_dpp v61, " VP " wave_ror:1" SW_DPPF \
  "s_cmp_eq_u32 %[ust], 24\n" \
  "s_nop 0\n" \
  "v_mov_b32_dpp v61, " V " wave_shr:1" SW_DPPF                     /* A = v[i-1] */ \
  "v_xor_b32 %[t0], " V ", v61\n" \
  "s_cbranch_scc1 .Luz" J "_%=\n" \
  /* behind 64 exact DFCM hits of stride S: again iff stride == S, v[i-2] = v[i-1] - S in the FCM class of v[i-1], v ^ v[i-1] >= 256 */ \
  "v_sub_u32 %[t3], " V ", v61\n" \
  "v_cmp_ne_u32_e64 %[tm64], %[ustr], %[t3]\n" \
  "v_subrev_u32 %[t3], %[ustr], v61\n" \
  "v_cmp_gt_u32_e32 vcc, 0x100, %[t0]\n" \
  "v_xor_b32 %[t3], %[t3], v61\n" \
  "s_or_b64 %[tm64], %[tm64], vcc\n" \
  "v_cmp_lt_u32_e32 vcc, 0xfffffff, %[t3]\n" \
  "s_or_b64 %[tm64], %[tm64], vcc\n" \
  "s_cmp_lg_u64 %[tm64], 0\n" \
  "s_cbranch_scc1 .Lul" J "_%=\n" \
  "s_mov_b64 exec, 0x3fffff\n"                                       /* 88 bytes: eight times (header of eight codes 5, eight zero bytes) */ \
  "v_add_u32 %[t3], %[pa], %[lane4p1]\n" \
  "s_mov_b32 %[total], 88\n" \
  "ds_write_b32 %[t3], %[pat5]\n"                                    /* (not dword aligned: gfx950 executes it) */ \
  "s_mov_b64 exec, -1\n" \
  "s_branch .Lend" J "_%=\n" \
  ".Luz" J "_%=:\n"                                                  /* behind 64 exact FCM hits: again iff every value equals the one before */ \
  "v_cmp_ne_u32_e32 vcc, 0, %[t0]\n" \
  "s_cmp_lg_u64 vcc, 0\n" \
  "s_cbranch_scc1 .Lul" J "_%=\n" \
  "s_mov_b64 exec, 0x3f\n"                                           /* 24 zero bytes (t0 is zero in every lane) */ \
  "v_add_u32 %[t3], %[pa], %[lane4p1]\n" \
  "s_mov_b32 %[total], 24\n" \
  "ds_write_b32 %[t3], %[t0]\n" \
  "s_mov_b64 exec, -1\n" \
  "s_branch .Lend" J "_%=\n" \
  ".Lul" J "_%=:\n" \
  SW_LEAVE(VP) \
  "s_branch .Lg" J "_%=\n" \
  /* a general step without a run start: 64 exact FCM hits (24 bytes), or 64 exact DFCM hits whose FCM residual is longer (88)? */ \
  ".Lce" J "_%=:\n" \
  "v_xor_b32 %[t0], " V ", v61\n" \
  "v_add_u32 %[t3], v61, %[s1p]\n" \
  "v_cmp_ne_u32_e64 %[tm64], 0, %[t0]\n" \
  "v_xor_b32 %[t4], " V ", %[t3]\n" \
  "v_cmp_gt_u32_e32 vcc, 0x100, %[t0]\n" \
  "s_cmp_eq_u64 %[tm64], 0\n" \
  "s_cselect_b32 %[ust], 24, 0\n" \
  "v_cmp_ne_u32_e64 %[tm64], 0, %[t4]\n" \
  "v_readlane_b32 %[ustr], %[sp], 63\n" \
  "s_or_b64 vcc, vcc, %[tm64]\n" \
  "s_cmp_eq_u64 vcc, 0\n" \
  "s_cselect_b32 %[tms], 88, 0\n" \
  "s_or_b32 %[ust], %[ust], %[tms]\n" \
  "s_branch .Lend" J "_%=\n"

// Once per block of eight steps: my progress, everybody's; the components in front wait for the slowest (bounded), so that the component
// waves read the same cache lines at about the same time and the interleaved array comes over HBM once, and the wave that did not
// have to wait - the slow component - runs the next block at priority 3 (the waves of a workgroup hold their LDS until the last of
// them is done, and the sweep ends when the slowest component does).  A wave that is done has 0xffffffff there.
// (Measured and dropped in round 5, profiles/r05_sweep_experiments.txt: balancing the WORKGROUPS of a compute unit against the
// hardware's oldest-first issue order - priorities by age, priorities from a table of progress per unit, sleeping while ahead of the
// unit's slowest.  Each does what it says - with the last one every segment of the mesh takes the same time to 2 % - and none
// changes when the kernel ends: the SIMDs issue about 1.3 instructions per 4 cycles whoever runs.)
#define SW_PROG_READ \
  "ds_read_b128 v[58:61], %[prog0]\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "v_readfirstlane_b32 %[pg0], v58\n" \
  "v_readfirstlane_b32 %[pg1], v59\n" \
  "v_readfirstlane_b32 %[pg2], v60\n"
#define SW_LAG \
  "v_mov_b32 %[t3], %[ib]\n" \
  "s_mov_b64 exec, 1\n" \
  "ds_write_b32 %[progc], %[t3]\n" \
  "s_mov_b64 exec, -1\n" \
  SW_PROG_READ \
  "s_mov_b32 %[cand], 0\n" \
  ".Lsp_%=:\n" \
  "s_min_u32 %[tms], %[pg0], %[pg1]\n" \
  "s_min_u32 %[tms], %[tms], %[pg2]\n" \
  "s_cmp_ge_u32 %[tms], %[ib]\n"                                     /* nobody is behind me (done waves count as ahead) */ \
  "s_cbranch_scc1 .Lgo_%=\n" \
  "s_sleep 12\n"                                                     /* (768 cycles: the wave that is waited for needs the issue slots) */ \
  "s_add_u32 %[cand], %[cand], 1\n" \
  "s_cmp_lt_u32 %[cand], 4096\n" \
  "s_cbranch_scc0 .Lgo_%=\n" \
  SW_PROG_READ \
  "s_branch .Lsp_%=\n" \
  ".Lgo_%=:\n" \
  "s_cmp_eq_u32 %[cand], 0\n" \
  "s_cbranch_scc1 .Lp3_%=\n" \
  "s_setprio 0\n" \
  "s_branch .Lp_%=\n" \
  ".Lp3_%=:\n" \
  "s_setprio 3\n" \
  ".Lp_%=:\n"

#define SW_LOOP(HOOKTXT) \
  "s_add_u32 %[pa], %[pa], %[sbase]\n"                               /* bytes staged -> LDS address of the first free staging byte - 1 */ \
  "s_sub_u32 %[pa], %[pa], 1\n" \
  "s_mov_b32 %[ust], 0\n" \
  "s_mov_b32 %[ustr], 0\n" \
  "s_mov_b32 %[pend], 0\n" \
  "s_mov_b32 %[floff], 0\n" \
  "buffer_load_dword %[c0], %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "buffer_load_dword %[c1], %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "buffer_load_dword %[c2], %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "buffer_load_dword %[c3], %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "buffer_load_dword %[c4], %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "buffer_load_dword %[c5], %[voff], %[inr], %[soff] offen\n" \
  "s_add_u32 %[soff], %[soff], %[stepb]\n" \
  "s_waitcnt vmcnt(0)\n" \
  "s_cmp_eq_u32 %[nrb], 0\n" \
  "s_cbranch_scc1 .Lblk_%=\n" \
  ".Lrb_%=:\n" \
  SW_RSTEP("%[c0]", "%[c7]", "%[c6]") \
  SW_RSTEP("%[c1]", "%[c0]", "%[c7]") \
  SW_RSTEP("%[c2]", "%[c1]", "%[c0]") \
  SW_RSTEP("%[c3]", "%[c2]", "%[c1]") \
  SW_RSTEP("%[c4]", "%[c3]", "%[c2]") \
  SW_RSTEP("%[c5]", "%[c4]", "%[c3]") \
  SW_RSTEP("%[c6]", "%[c5]", "%[c4]") \
  SW_RSTEP("%[c7]", "%[c6]", "%[c5]") \
  "s_sub_u32 %[nrb], %[nrb], 1\n" \
  "s_cmp_lg_u32 %[nrb], 0\n" \
  "s_cbranch_scc1 .Lrb_%=\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  ".Lblk_%=:\n" \
  SW_LAG \
  SW_QUAD_TRY("0") \
  SW_STEP("0", "%[c0]", "%[c7]", "%[c6]", HOOKTXT) \
  SW_STEP("1", "%[c1]", "%[c0]", "%[c7]", HOOKTXT) \
  SW_STEP("2", "%[c2]", "%[c1]", "%[c0]", HOOKTXT) \
  SW_STEP("3", "%[c3]", "%[c2]", "%[c1]", HOOKTXT) \
  SW_QUAD_TRY("4") \
  SW_STEP("4", "%[c4]", "%[c3]", "%[c2]", HOOKTXT) \
  SW_STEP("5", "%[c5]", "%[c4]", "%[c3]", HOOKTXT) \
  SW_STEP("6", "%[c6]", "%[c5]", "%[c4]", HOOKTXT) \
  SW_STEP("7", "%[c7]", "%[c6]", "%[c5]", HOOKTXT) \
  "s_add_u32 %[ib], %[ib], 512\n" \
  "s_sub_u32 %[nblk], %[nblk], 1\n" \
  "s_cmp_lg_u32 %[nblk], 0\n" \
  "s_cbranch_scc1 .Lblk_%=\n" \
  /* the end: the table and the registers as a general step leaves them, nothing left to flush, nothing in flight */ \
  "s_cmp_eq_u32 %[ust], 0\n" \
  "s_cbranch_scc1 .Lxg_%=\n" \
  SW_LEAVE("%[c7]") \
  ".Lxg_%=:\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "s_cmp_lg_u32 %[pend], 0\n" \
  "s_cselect_b64 exec, -1, 0\n" \
  SW_FLUSH_STORES \
  "s_mov_b64 exec, -1\n" \
  "s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
  "s_add_u32 %[pa], %[pa], 1\n" \
  "s_sub_u32 %[pa], %[pa], %[sbase]\n" \
  "s_branch .Lout_%=\n" \
  SW_QUAD("0", "3", "%[c0]", "%[c1]", "%[c2]", "%[c3]", "%[c7]", "%[c6]", "%[c7]", "%[c0]", "%[c1]") \
  SW_QUAD("4", "7", "%[c4]", "%[c5]", "%[c6]", "%[c7]", "%[c3]", "%[c2]", "%[c3]", "%[c4]", "%[c5]") \
  SW_STEP_OOL("0", "%[c0]", "%[c7]") \
  SW_STEP_OOL("1", "%[c1]", "%[c0]") \
  SW_STEP_OOL("2", "%[c2]", "%[c1]") \
  SW_STEP_OOL("3", "%[c3]", "%[c2]") \
  SW_STEP_OOL("4", "%[c4]", "%[c3]") \
  SW_STEP_OOL("5", "%[c5]", "%[c4]") \
  SW_STEP_OOL("6", "%[c6]", "%[c5]") \
  SW_STEP_OOL("7", "%[c7]", "%[c6]") \
  ".Lout_%=:\n"

// What the loop is given besides the running state
struct LoopK
  {
  u32x4 inr, slr, rcr;                            // descriptors: the input from the segment's first value on, the slot, the record list
  uint32_t voff, stepb;                           // the lane's offset in the interleaved array, bytes of it a step covers
  uint32_t t1abs, sbase;                          // LDS addresses of the wave's tables and of its staging area
  uint32_t prog0, progc;                          // LDS addresses of the workgroup's progress words and of this wave's
  };

// nblk whole blocks of eight full steps, starting at value index i_begin of the stream
template <bool HOOK>
__device__ __forceinline__ void sweep_blocks_asm(Sweep& sw, uint32_t nrb, uint32_t nblk, uint32_t i_begin, const LoopK& k, const LaneK& lk, uint32_t sabotage)
  {
  nrb = uni(nrb);
  uint32_t c0, c1, c2, c3, c4, c5, c6 = 0u, c7 = sw.vp;
  uint32_t sp = sw.sp, s1p = sw.s1p, a1p = sw.a1p, a2p = sw.a2p;
  uint32_t posl = uni(sw.posl), flushed = uni(sw.flushed), nrec = uni(sw.nrec), soff = 0u, ib = uni(i_begin);
  uint64_t sent = uni(sw.sent);
  nblk = uni(nblk);
  uint32_t t0, t1, t2v, t3, t4, t5, t6, t7, t8, t9;
  uint32_t ust, ustr, pend, floff, cand, total, tms, pg0, pg1, pg2, pg3;
  uint64_t tm64, st1, st2, ft1, ft2, hole;
  const uint32_t lane8 = 8u * lk.lane, stw = k.sbase + lk.lane4, stw8 = k.sbase + lane8, lanegi = (lk.lane & 7u) << REC_GI_SHIFT;
  // the lane's dword of the pattern of exact DFCM hits (LaneK::pat5) 64 dwords on: four uniform steps are 88 dwords
  uint32_t pat5b = 0;
  for (uint32_t b = 0; b < 4u; ++b)
    {
    const uint32_t q = (256u + 4u * lk.lane + b) % 11u;
    pat5b |= (q == 0u ? 0xb6u : q == 1u ? 0xdbu : q == 2u ? 0x6du : 0u) << (8u * b);
    }
  const uint64_t hk = (HOOK && sabotage) ? 0xffffffff00000000ull : 0ull;
  // (every read-write operand is early-clobber: without the "&" the compiler may put an input that has the same value on entry -
  // a constant 0, say - into the same register, and the loop changes it under that input's feet)
#define SW_OPERANDS \
    : [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3), [c4] "=&v"(c4), [c5] "=&v"(c5), [c6] "+&v"(c6), [c7] "+&v"(c7), \
      [sp] "+&v"(sp), [s1p] "+&v"(s1p), [a1p] "+&v"(a1p), [a2p] "+&v"(a2p), \
      [pa] "+&s"(posl), [flushed] "+&s"(flushed), [nrec] "+&s"(nrec), [soff] "+&s"(soff), [ib] "+&s"(ib), [nblk] "+&s"(nblk), [nrb] "+&s"(nrb), [sent] "+&s"(sent), \
      [t0] "=&v"(t0), [t1] "=&v"(t1), [t2v] "=&v"(t2v), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), \
      [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9), \
      [ust] "=&s"(ust), [ustr] "=&s"(ustr), [pend] "=&s"(pend), [floff] "=&s"(floff), [cand] "=&s"(cand), [total] "=&s"(total), [tms] "=&s"(tms), \
      [pg0] "=&s"(pg0), [pg1] "=&s"(pg1), [pg2] "=&s"(pg2), [pg3] "=&s"(pg3), \
      [tm64] "=&s"(tm64), [st1] "=&s"(st1), [st2] "=&s"(st2), [ft1] "=&s"(ft1), [ft2] "=&s"(ft2), [hole] "=&s"(hole) \
    : [voff] "v"(k.voff), [lane8] "v"(lane8), [stw] "v"(stw), [stw8] "v"(stw8), [grp3] "v"(lk.grp3), [sh3] "v"(lk.sh3), [c4sh] "v"(lk.c4sh), \
      [lanegi] "v"(lanegi), [pat5] "v"(lk.pat5), [progc] "v"(k.progc), [prog0] "v"(k.prog0), [pat5b] "v"(pat5b), [lane4p1] "v"(lk.lane4 + 1u), \
      [inr] "s"(k.inr), [slr] "s"(k.slr), [rcr] "s"(k.rcr), [stepb] "s"(k.stepb), [t1s] "s"(k.t1abs), [t2s] "s"(k.t1abs + 64u), [sbase] "s"(k.sbase), \
      [ksent] "s"(SENT), [k3e0] "s"(0x3e0u), [kswap] "s"(0x00010203u), [lead] "s"(0x0101010101010101ull), [hk] "s"(hk), [kf0] "s"(0xf0000000u), [lim] "s"(k.sbase + 511u) \
    : "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc", "scc", "memory"
  if (HOOK)
    asm volatile(SW_LOOP(SW_HOOK) SW_OPERANDS);
  else
    asm volatile(SW_LOOP("") SW_OPERANDS);
#undef SW_OPERANDS
  sw.vp = c7; sw.sp = sp; sw.s1p = s1p; sw.a1p = a1p; sw.a2p = a2p;
  sw.posl = uni(posl); sw.flushed = uni(flushed); sw.nrec = uni(nrec); sw.sent = uni(sent);
  sw.pend1 = sw.pend2 = 0u;
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

// ---- the sweep ---------------------------------------------------------------------------------------------------------------
// what a wave is given to start a segment with
__device__ __forceinline__ void sweep_begin(Sweep& sw, const uint32_t* __restrict__ src, uint32_t n, uint32_t arity, uint32_t c, uint32_t g,
                                            uint32_t i_begin, uint32_t* __restrict__ T, uint8_t* __restrict__ stage, uint32_t lane)
  {
  for (uint32_t k = lane; k < (uint32_t)TAB; k += 64u)
    T[k] = SENT;
  sw.pend1 = sw.pend2 = 0u;
  sw.posl = 0;
  sw.flushed = 0;
  sw.nrec = 0;
  sw.flags = 0;
  sw.sent = 0;
  sw.a1p = sw.a2p = 0xfffffffeu;                       // the first value of a segment always looks at the table
  // the three values before the segment (0 before the stream: the reference starts from zeroed state, fpsc.c:104-116)
  const uint32_t m1 = i_begin >= 1u ? src[(size_t)(i_begin - 1u) * arity + c] : 0u;
  const uint32_t m2 = i_begin >= 2u ? src[(size_t)(i_begin - 2u) * arity + c] : 0u;
  const uint32_t m3 = i_begin >= 3u ? src[(size_t)(i_begin - 3u) * arity + c] : 0u;
  sw.vp = m1;
  sw.sp = m1 - m2;
  sw.s1p = m2 - m3;
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
  }

// what is left in the staging area (< 512 bytes) goes to the slot; returns the bytes the wave produced
__device__ __forceinline__ uint32_t sweep_end(Sweep& sw, uint8_t* __restrict__ stage, uint8_t* __restrict__ gbase, const LaneK& lk)
  {
  for (uint32_t off = lk.lane4; off < sw.posl; off += 256u)
    store_span(gbase + sw.flushed, off, ((const uint32_t*)stage)[off >> 2], sw.posl);
  return sw.flushed + sw.posl;
  }

// The write-side guard (workgroups S .. S + G - 1 of the sweep): workgroup j codes the first GUARD_STEPS steps of a sampled segment
// again - compiled step, run starts resolved with BALLOTS, nothing that depends on the order in which the LDS unit applies an
// exchange - into a slot and a record list of its own.  The fix-up kernel compares them with what the sweep made of that segment
// (k_fpc32_scanfix, guard_compare): same bytes, same records, or FLAG_ORDER.
struct GuardMeta { uint32_t seg, bytes, nrec, pad; };

template <bool HOOK>
__device__ __forceinline__ void guard_segment(const uint32_t* __restrict__ src, uint32_t n, uint32_t arity, uint32_t L, const Stagger& sg, uint32_t S, uint32_t j,
                                              uint32_t c, uint32_t lane, uint32_t seed, uint32_t* __restrict__ lds, uint8_t* __restrict__ gslots,
                                              uint32_t* __restrict__ grecs, GuardMeta* __restrict__ gmeta, uint32_t rblocks)
  {
  const uint32_t g = (uint32_t)(((uint64_t)j * 0x9E3779B1ull + seed) % S);        // which segment: from a counter of the encodes
  uint32_t* T = lds + c * LDSW;
  uint8_t* stage = (uint8_t*)(T + TAB);
  const uint32_t t1abs = (uint32_t)(uintptr_t)(lds_u8*)T;
  const LaneK lk = lane_constants(lane);
  uint32_t i_begin, Lg;
  segment_range(sg, L, g, i_begin, Lg);
  const uint32_t seg_end = (n - i_begin < Lg) ? n : i_begin + Lg;
  const uint32_t i_end = (seg_end - i_begin < 64u * GUARD_STEPS) ? seg_end : i_begin + 64u * GUARD_STEPS;
  const size_t row = (size_t)j * arity + c;
  uint8_t* gbase = gslots + row * GUARD_CAP;
  const rsrc_t slot = make_rsrc(gbase, GUARD_CAP);
  const RecSink sink = { grecs + row * RCAP * RECW };
  Sweep sw;
  const uint32_t ws = i_begin - 512u * (i_begin / 512u < rblocks ? i_begin / 512u : rblocks);      // the window the sweep replays in front of the segment
  sweep_begin(sw, src, n, arity, c, g, ws, T, stage, lane);
#pragma unroll 1
  for (uint32_t i0 = ws; i0 < i_begin; i0 += 64u)
    replay_step<true>(src[(size_t)(i0 + lane) * arity + c], t1abs, sw, lk);
#pragma unroll 1
  for (uint32_t i0 = i_begin; i0 < i_end; i0 += 64u)
    {
    const uint32_t i = i0 + lane;
    const uint32_t v = i < seg_end ? src[(size_t)i * arity + c] : 0u;
    if (i0 + 64u <= seg_end)
      code_step<true, true, HOOK>(v, i0, seg_end, n, t1abs, stage, slot, sw, lk, sink, 0u);
    else
      code_step<false, true, HOOK>(v, i0, seg_end, n, t1abs, stage, slot, sw, lk, sink, 0u);
    }
  const uint32_t bytes = sweep_end(sw, stage, gbase, lk);
  if (lane == 0)
    gmeta[row] = GuardMeta{ g, bytes, sw.nrec, 0u };
  }

// ASM = false: every step through the compiled code_step (TRICO_FPC32_ASM=0, for A/B runs)
template <bool HOOK, bool ASM>
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_fpc32_sweep(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L, const Stagger sg, uint32_t S, uint32_t* __restrict__ outT,
              uint8_t* __restrict__ slots, size_t slot_stride, uint32_t segcap, uint32_t* __restrict__ segbytes, uint32_t* __restrict__ rawbytes,
              uint32_t* __restrict__ nrec, uint32_t* __restrict__ recs, uint32_t sabotage, uint32_t seed,
              uint8_t* __restrict__ gslots, uint32_t* __restrict__ grecs, GuardMeta* __restrict__ gmeta, uint64_t* __restrict__ diag,
              uint32_t rblocks, uint64_t* __restrict__ agg, uint32_t agg_words)
  {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  // the ready words and counters of the scan that follows (k_fpc32_scanfix): a word per thread
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < agg_words; q += gridDim.x * blockDim.x)
    agg[q] = 0ull;
  const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;      // (c in a scalar register)
  const uint32_t g = blockIdx.x;
#ifdef TRICO_SWEEP_DIAG
  const uint64_t dg_w0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (g >= S)
    {
    guard_segment<HOOK>(src, n, (uint32_t)arity, L, sg, S, g - S, c, lane, seed, lds, gslots, grecs, gmeta, rblocks);
#ifdef TRICO_SWEEP_DIAG
    if (lane == 0)
      {
      uint64_t* r = diag + 64 + ((size_t)g * arity + c) * 8;
      r[0] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); r[1] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
      r[2] = dg_w0; r[3] = 0; r[4] = 0; r[5] = __builtin_amdgcn_s_memrealtime(); r[6] = 0; r[7] = 0;
      }
#endif
    return;
    }
  volatile uint32_t* prog = lds + arity * LDSW;        // [4] progress of the component waves (value index; 0xffffffff: done, or no such wave)
  uint32_t* T = lds + c * LDSW;
  uint8_t* stage = (uint8_t*)(T + TAB);
  const uint32_t t1abs = (uint32_t)(uintptr_t)(lds_u8*)T;                  // LDS address of the wave's tables (a multiple of 64)
  const LaneK lk = lane_constants(lane);
  uint32_t i_begin, Lg;
  segment_range(sg, L, g, i_begin, Lg);
  const uint32_t i_end = (n - i_begin < Lg) ? n : i_begin + Lg;
  uint8_t* gbase = slots + (size_t)c * slot_stride + (size_t)g * segcap;
  const rsrc_t slot = make_rsrc(gbase, segcap);
  const size_t rowi = (size_t)g * arity + c;
  const RecSink sink = { recs + rowi * RCAP * RECW };
  Sweep sw;
  // the blocks of the segment before that are replayed (SW_RSTEP): the wave starts there, with empty tables
  const uint32_t rblk = i_begin / 512u < rblocks ? i_begin / 512u : rblocks;
  const uint32_t ws = i_begin - 512u * rblk;
  sweep_begin(sw, src, n, (uint32_t)arity, c, g, ws, T, stage, lane);
  // The descriptor of the input begins at the window and ends with the array: what a load beyond the segment fetches belongs to
  // the next one and is not used, beyond the array it is zero.
  const uint8_t* seg_src = (const uint8_t*)(src + (size_t)ws * arity);
  const uint64_t seg_bytes = ((uint64_t)n - ws) * (uint64_t)arity * 4u;
  const rsrc_t in = make_rsrc(seg_src, seg_bytes);
  const uint32_t voff = (lane * (uint32_t)arity + c) * 4u;
  const uint32_t stepb = 256u * (uint32_t)arity;       // bytes of the interleaved array a step covers
  if (lane == 0)
    prog[c] = i_begin;
  if (c == 0 && lane >= (uint32_t)arity && lane < 4u)
    prog[lane] = 0xffffffffu;
  __syncthreads();                                     // everybody's progress word is this workgroup's before anybody compares
#ifdef TRICO_SWEEP_DIAG
  const uint64_t dg_t0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // whole blocks of eight full steps: the hand-written loop; what is left (fewer than eight full steps and the last, partial step of
  // a stream) goes through the compiled step
  uint32_t i0 = i_begin;
  const uint32_t nblk = ASM ? (i_end - i_begin) / 512u : 0u;
  if (nblk == 0u)
    {
#pragma unroll 1
    for (uint32_t r0 = ws; r0 < i_begin; r0 += 64u)
      replay_step<false>(__builtin_amdgcn_raw_buffer_load_b32(in, voff, ((r0 - ws) >> 6) * stepb, 0), t1abs, sw, lk);
    }
  else
    {
      {
      LoopK k;
      k.inr = make_desc(seg_src, seg_bytes);
      k.slr = make_desc(gbase, segcap);
      k.rcr = make_desc(sink.recs, (uint64_t)RCAP * RECW * 4u);
      k.voff = voff;
      k.stepb = stepb;
      k.t1abs = t1abs;
      k.sbase = (uint32_t)(uintptr_t)(lds_u8*)stage;
      k.prog0 = (uint32_t)(uintptr_t)(lds_u8*)(uint32_t*)(lds + arity * LDSW);
      k.progc = k.prog0 + 4u * c;
      sweep_blocks_asm<HOOK>(sw, rblk, nblk, i_begin, k, lk, sabotage);
      i0 += 512u * nblk;
      }
    }
  if (lane == 0)
    prog[c] = 0xffffffffu;                             // done with the part that takes time: nobody waits for me any more
#ifdef TRICO_SWEEP_DIAG
  const uint64_t dg_r1 = __builtin_amdgcn_s_memrealtime(), dg_t1 = __builtin_amdgcn_s_memtime();
  if (g == S / 2u && lane == 0)
    {
    // clocks of one wave per component: shader clock and 100 MHz reference clock over the loop, steps it took
    diag[4u * c + 0u] = dg_t1 - dg_t0;
    diag[4u * c + 1u] = dg_r1 - dg_r0;
    diag[4u * c + 2u] = (i0 - i_begin) / 64u;
    diag[4u * c + 3u] = sw.nrec;
    }
#endif
#pragma unroll 1
  for (; i0 < i_end; i0 += 64u)
    {
    // (the values beyond the stream come back as zeros, or as the next segment's: the partial step masks them by index)
    const uint32_t vcur = __builtin_amdgcn_raw_buffer_load_b32(in, voff, ((i0 - ws) >> 6) * stepb, 0);
    if (i0 + 64u <= i_end)
      code_step<true, false, HOOK>(vcur, i0, i_end, n, t1abs, stage, slot, sw, lk, sink, sabotage);
    else
      code_step<false, false, HOOK>(vcur, i0, i_end, n, t1abs, stage, slot, sw, lk, sink, sabotage);
    }
  const uint32_t bytes = sweep_end(sw, stage, gbase, lk);
  // what the segment leaves behind: the tables with the last value's writes applied (SENT = not written here)
  if (lane == 63u)
    {
    if (sw.pend1) *lds_at(sw.a1p) = sw.vp;
    if (sw.pend2) *lds_at(sw.a2p) = sw.sp;
    }
  if ((sw.pend1 && __ballot(lane == 63u && sw.vp == SENT)) || (sw.pend2 && __ballot(lane == 63u && sw.sp == SENT)) || sw.sent != 0ull)
    sw.flags |= FLAG_SENTINEL;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (uint32_t k = lane; k < (uint32_t)TAB; k += 64u)
    outT[rowi * ROW + k] = T[k];
  if (lane == 0)
    {
    segbytes[(size_t)c * S + g] = bytes;                 // (what the fix-up kernel takes the unused bytes of the deferred values off)
    rawbytes[(size_t)c * S + g] = bytes;                 // what the slot holds
    nrec[rowi] = sw.nrec | (sw.flags << 16);
    }
#ifdef TRICO_SWEEP_DIAG
  if (lane == 0)
    {
    // per wave: where it ran (HW_ID, XCC_ID), when it started, when its loop began and ended, when it was done (100 MHz clock)
    uint64_t* r = diag + 64 + ((size_t)g * arity + c) * 8;
    r[0] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); r[1] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    r[2] = dg_w0; r[3] = dg_r0; r[4] = dg_r1; r[5] = __builtin_amdgcn_s_memrealtime(); r[6] = dg_t1 - dg_t0; r[7] = sw.nrec;
    }
#endif
  }

// ---- cross-segment scan + the deferred values, ONE launch -------------------------------------------------------------------------
// Incoming payload of (segment, class) = the published entry of the nearest earlier segment that wrote the class, else 0 (the
// reference's zeroed tables, fpsc.c:104-105); then, per record of a deferred value: residual, length and code from the incoming entry
// (fpsc.c:133-189 for one value; the record keeps them for the gather: w6 = residual, w7 = length | code << 4) and the segment's size
// without the unused bytes of its reserved fields (rawbytes keeps what the slot holds).
// Until round 6 these were three launches - k_fpc32_pscan_a (latest entry per chunk of CH segments), k_fpc32_pscan_b (the same rows
// read again, the incoming rows written: 28 MB), k_fpc32_fixup (a wave per row, looking its classes up in those rows) - of 6 + 12 +
// 20 us with ~4.6 us of nothing between two launches.  Now ONE workgroup of 1024 threads owns (chunk y, component c): thread k reads
// column k of the chunk's CH rows, leaves for every row "the latest entry written earlier IN THIS CHUNK, else SENT" in LDS (CH x 1040
// words: 133 KB), publishes the chunk's latest as one 64-bit word "ready | entry" (an agent-scope store: a word that carries its own
// flag needs no fence), takes the words of the chunks before it as they appear (what comes into the chunk: a 1040-word row of LDS),
// and then the workgroup's waves take the chunk's rows, a wave per row: a look-up is an LDS read, and SENT there means "what came
// into the chunk".  The incoming rows never exist in memory.  The sweep zeroes the ready words.  Workgroups are dispatched in the
// order of their index and a workgroup waits only for lower indices: no wait needs a workgroup the device has not started.  The
// waits are bounded all the same (SPIN_MAX polls): FLAG_SCAN sends the stream to the two-sweep coder.
// The write-side guard's comparison rides along: what guard workgroup j of the sweep coded with ballots (guard_segment) against what
// the sweep left for that segment - the bytes, the number of records up to there and the records themselves (the words the sweep
// wrote).  A difference raises FLAG_ORDER for the component (it travels with the segment's record count to k_fpc32_offsets and the
// host).  The LAST workgroups do it, between publishing and looking back: they have the longest wait in front of them.
constexpr uint64_t AGG_READY = 1ull << 32;
constexpr uint32_t SPIN_MAX = 1u << 21;             // x (a round of loads + s_sleep 2): seconds
constexpr uint32_t SF_THREADS = 1024;
constexpr uint32_t SF_LDS_WORDS = (uint32_t)CH * TAB + TAB + CH;
static_assert(TAB - 1024 == 16 && SF_LDS_WORDS * 4u <= 160u * 1024u, "thread k: DFCM... column k, the first 16 threads one more; LDS of a compute unit");

__device__ __forceinline__ uint64_t agg_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// What comes into chunk y for column k of component c: the latest entry of the chunks before it, else 0.  One round of look_back_round
// asks for B words at once, nearest chunk first (a round trip of such a load is ~2 us: no cache may serve it), and a wave all of whose
// lanes have met a writer stops looking.
template <int B>
__device__ __forceinline__ void look_back_round(const uint64_t* __restrict__ agg, uint32_t j1, uint32_t arity, uint32_t c, uint32_t k,
                                                uint32_t& carry, bool& found, bool& late, uint32_t spin_max)
  {
  uint64_t w[B];
  uint32_t polls = 0;
  for (;;)
    {
    bool ok = true;
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)B; ++q)
      {
      w[q] = q < j1 ? agg_load(&agg[((size_t)(j1 - 1u - q) * arity + c) * TAB + k]) : (AGG_READY | SENT);      // chunk j1 - 1 first
      ok = ok && (w[q] >> 32) != 0ull;
      }
    if (ok)
      break;
    if (++polls > spin_max)
      {
      late = true;
      break;
      }
    __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
  for (uint32_t q = 0; q < (uint32_t)B; ++q)
    {
    const uint32_t v = (uint32_t)w[q];
    if (!found && v != SENT)
      {
      carry = v;
      found = true;
      }
    }
  }

// Two levels: the chunks of my group of GROUPC first (one round); the last chunk of every group publishes the group's latest entry
// when it has done that, and whoever is still looking takes the groups before its own (one more round for up to eight groups) - at most
// 7 + 8 words per column and two or three round trips.  (One level - the eight nearest chunks, then 24 at a time, up to 71 words and four
// round trips: 34.2 us on the benchmark mesh, 26.2 us on the walk mesh, against 31.7 and 17.3.)
constexpr uint32_t GROUPC = 8;
__device__ __forceinline__ uint32_t look_back(uint64_t* __restrict__ agg, uint32_t nch, uint32_t y, uint32_t arity, uint32_t c, uint32_t k,
                                              uint32_t own, bool& late, uint32_t spin_max)
  {
  uint32_t carry = 0;
  bool found = false;
  const uint32_t gp = y / GROUPC, j = y % GROUPC;
  if (j > 0u)
    look_back_round<(int)GROUPC>(agg + (size_t)gp * GROUPC * arity * TAB, j, arity, c, k, carry, found, late, spin_max);
  uint64_t* ga = agg + (size_t)nch * arity * TAB;
  if (j == GROUPC - 1u)
    __hip_atomic_store(&ga[((size_t)gp * arity + c) * TAB + k], AGG_READY | (own != SENT ? own : found ? carry : SENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t g1 = gp; g1 > 0u && __ballot(!found) != 0ull; g1 = g1 > GROUPC ? g1 - GROUPC : 0u)
    look_back_round<(int)GROUPC>(ga, g1, arity, c, k, carry, found, late, spin_max);
  return carry;
  }

// one (guard row, component) by the first 256 threads of a workgroup
__device__ __forceinline__ void guard_compare(int arity, uint32_t S, const uint32_t* __restrict__ rawbytes, uint32_t* __restrict__ nrec,
                                              const uint32_t* __restrict__ recs, const uint8_t* __restrict__ slots, size_t slot_stride, uint32_t segcap,
                                              const uint8_t* __restrict__ gslots, const uint32_t* __restrict__ grecs,
                                              const GuardMeta* __restrict__ gmeta, uint32_t j, uint32_t c, uint32_t lane)
  {
  const size_t row = (size_t)j * arity + c;
  const GuardMeta m = gmeta[row];
  const size_t rowi = (size_t)m.seg * arity + c;
  const uint32_t* mine = (const uint32_t*)(slots + (size_t)c * slot_stride + (size_t)m.seg * segcap);
  const uint32_t* theirs = (const uint32_t*)(gslots + row * GUARD_CAP);
  const uint32_t H = nrec[rowi] & 0xffffu;
  bool diff = H < m.nrec || rawbytes[(size_t)c * S + m.seg] < m.bytes;
  // (differences are ORed together, not tested one by one: the loads of all rounds are then independent of each other - with
  // `diff = diff || ...` the up to 18 rounds of a workgroup were 18 dependent pairs of round trips)
  uint32_t acc = 0;
  const uint32_t nw = (m.bytes + 3u) >> 2;
#pragma unroll 4
  for (uint32_t t = lane; t < nw; t += 256u)
    {
    const uint32_t keep = 4u * t + 4u <= m.bytes ? 0xffffffffu : (1u << (8u * (m.bytes & 3u))) - 1u;
    acc |= (mine[t] ^ theirs[t]) & keep;
    }
  const uint32_t* ra = recs + rowi * RCAP * RECW, * rb = grecs + row * RCAP * RECW;
  const uint32_t ncmp = H < m.nrec ? 0u : REC_CMPW * m.nrec;
#pragma unroll 4
  for (uint32_t t = lane; t < ncmp; t += 256u)
    acc |= ra[RECW * (t / REC_CMPW) + t % REC_CMPW] ^ rb[RECW * (t / REC_CMPW) + t % REC_CMPW];
  diff = diff || acc != 0u;
  if (lane == 0 && H > m.nrec)
    diff = diff || (ra[RECW * m.nrec] & REC_POS) < m.bytes;      // a record of the sweep inside the compared bytes that the guard does not have
  if (diff)
    atomicOr(&nrec[rowi], FLAG_ORDER << 16);
  }

// The deferred values of ONE (segment, component) row, by one wave.  What a record needs is its words, the incoming entry of its
// class (LDS: `local` = the row of entries written earlier in the chunk, `cin` = what came into the chunk), the store of the result,
// and nothing else, so a lane takes up to FIX_B records at once: their loads are issued together, then their stores.
constexpr uint32_t FIX_B = 8;
struct FixRecs { u32x4 w[FIX_B]; uint32_t known[FIX_B], k2w[FIX_B]; };

// records base + 64 i + lane (i < FIX_B) of a row's list: the words the sweep wrote
__device__ __forceinline__ void fix_load(FixRecs& r, const uint32_t* __restrict__ list, uint32_t H, uint32_t base, uint32_t lane)
  {
#pragma unroll
  for (uint32_t i = 0; i < FIX_B; ++i)
    {
    const uint32_t j = base + 64u * i + lane;
    r.w[i] = u32x4{ 0u, 0u, 0u, 0u };
    r.known[i] = r.k2w[i] = 0u;
    if (j < H)
      {
      r.w[i] = *(const u32x4*)(list + RECW * j);
      const u32x2 t = *(const u32x2*)(list + RECW * j + 4u);
      r.known[i] = t[0];
      r.k2w[i] = t[1];
      }
    }
  }

// ... coded from the incoming entries; returns the lane's unused bytes
__device__ __forceinline__ uint32_t fix_code(const FixRecs& r, uint32_t* __restrict__ list, uint32_t H, uint32_t base, uint32_t lane,
                                             const uint32_t* __restrict__ local, const uint32_t* __restrict__ cin)
  {
  uint32_t unused = 0;
#pragma unroll
  for (uint32_t i = 0; i < FIX_B; ++i)
    {
    const uint32_t j = base + 64u * i + lane;
    if (j < H)
      {
      const bool ft1 = (r.w[i][0] & REC_FT1) != 0u, ft2 = (r.w[i][0] & REC_FT2) != 0u;
      uint32_t p1 = r.known[i], p2 = r.known[i];
      // (fpsc.c:96-97: the FCM class is the predecessor's top four bits)
      if (ft1)
        {
        const uint32_t k = r.w[i][3] >> 28;
        p1 = local[k];
        p1 = p1 == SENT ? cin[k] : p1;
        }
      if (ft2)
        {
        const uint32_t k = 16u + (r.k2w[i] >> 2);
        p2 = local[k];
        p2 = p2 == SENT ? cin[k] : p2;
        }
      const uint32_t v = r.w[i][2], a = r.w[i][3];
      const uint32_t x1 = v ^ p1, x2 = v ^ (a + p2);
      const uint32_t n1 = (39u - (uint32_t)__clz((int)x1)) >> 3;
      const uint32_t n2 = (39u - (uint32_t)__clz((int)(x2 | 1u))) >> 3;
      const bool use2 = n2 < n1;
      const uint32_t len = use2 ? n2 : n1, x = use2 ? x2 : x1, code = use2 ? (n2 | 4u) : n1;
      *(u32x2*)(list + RECW * j + 6u) = u32x2{ x, len | (code << 4) };
      unused += 4u - len;
      }
    }
  return unused;
  }

// the rest of a row whose first 64 * FIX_B records `first` holds, and the row's size
__device__ __forceinline__ void fixup_row_wave(const FixRecs& first, const uint32_t* __restrict__ local, const uint32_t* __restrict__ cin, uint32_t H,
                                               uint32_t* __restrict__ segbytes, const uint32_t* __restrict__ rawbytes,
                                               uint32_t* __restrict__ recs, uint32_t S, uint32_t arity, uint32_t rowi, uint32_t lane)
  {
  if (H == 0u)
    return;
  uint32_t* list = recs + (size_t)rowi * RCAP * RECW;
  uint32_t unused = fix_code(first, list, H, 0u, lane, local, cin);
  for (uint32_t base = 64u * FIX_B; base < H; base += 64u * FIX_B)
    {
    FixRecs more;
    fix_load(more, list, H, base, lane);
    unused += fix_code(more, list, H, base, lane, local, cin);
    }
  const uint32_t incl = wave_scan_incl(unused);
  if (lane == 63u)
    {
    const uint32_t c = rowi % arity, g = rowi / arity;
    segbytes[(size_t)c * S + g] = rawbytes[(size_t)c * S + g] - incl;
    }
  }

template <bool HOOK>
__global__ void __launch_bounds__(1024) k_fpc32_scanfix(const uint32_t* __restrict__ outT, uint32_t S, int arity, uint32_t nch,
                                                        uint64_t* __restrict__ agg, uint32_t* __restrict__ segbytes,
                                                        const uint32_t* __restrict__ rawbytes, uint32_t* __restrict__ nrec,
                                                        uint32_t* __restrict__ recs, const uint8_t* __restrict__ slots, size_t slot_stride, uint32_t segcap,
                                                        const uint8_t* __restrict__ gslots, const uint32_t* __restrict__ grecs,
                                                        const GuardMeta* __restrict__ gmeta, uint32_t guard_rows, uint32_t sabotage)
  {
  extern __shared__ __attribute__((aligned(16))) uint32_t sf_lds[];
  uint32_t* M = sf_lds;                              // [CH][TAB]: entries written earlier in the chunk
  uint32_t* cin = sf_lds + (uint32_t)CH * TAB;       // [TAB]: what comes into the chunk
  uint32_t* Hs = cin + TAB;                          // [CH]: records of the chunk's rows
  const uint32_t y = blockIdx.x / (uint32_t)arity, c = blockIdx.x - y * (uint32_t)arity;
  const uint32_t tid = threadIdx.x;
  const bool two = tid < (uint32_t)TAB - SF_THREADS;                 // the first sixteen threads: a second column
  const uint32_t k0 = tid, k1 = two ? SF_THREADS + tid : k0;
  const uint32_t g0 = y * CH, cnt = (S - g0 < (uint32_t)CH) ? S - g0 : (uint32_t)CH;
  if (tid < (uint32_t)CH)
    Hs[tid] = tid < cnt ? nrec[(size_t)(g0 + tid) * arity + c] & 0xffffu : 0u;
  uint32_t last0_kept = SENT, last1_kept = SENT;         // the chunk's own latest entries
  {
  uint32_t t0[CH], t1[CH];
#pragma unroll
  for (uint32_t i = 0; i < (uint32_t)CH; ++i)
    {
    const uint32_t* row = outT + ((size_t)(g0 + i) * arity + c) * ROW;
    t0[i] = i < cnt ? row[k0] : SENT;
    t1[i] = (two && i < cnt) ? row[k1] : SENT;
    }
  // first what the others wait for: the chunk's latest entries
  uint32_t last0 = SENT, last1 = SENT;
#pragma unroll
  for (uint32_t i = 0; i < (uint32_t)CH; ++i)
    {
    last0 = t0[i] != SENT ? t0[i] : last0;
    last1 = t1[i] != SENT ? t1[i] : last1;
    }
  last0_kept = last0;
  last1_kept = last1;
  uint64_t* mine = agg + ((size_t)y * arity + c) * TAB;
  // (test hook, libtrico_testhooks.so only: the first workgroup never publishes - everybody behind it in its component has to give up)
  const bool mute = HOOK && (sabotage & 2u) != 0u && blockIdx.x == 0u;
  if (!mute)
    {
    __hip_atomic_store(&mine[k0], AGG_READY | last0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (two)
      __hip_atomic_store(&mine[k1], AGG_READY | last1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  uint32_t run0 = SENT, run1 = SENT;
#pragma unroll
  for (uint32_t i = 0; i < (uint32_t)CH; ++i)
    {
    M[i * (uint32_t)TAB + k0] = run0;
    if (two)
      M[i * (uint32_t)TAB + k1] = run1;
    run0 = t0[i] != SENT ? t0[i] : run0;
    run1 = t1[i] != SENT ? t1[i] : run1;
    }
  }
  // the guard's rows: the last workgroups of the grid
  if (tid < 256u)
    for (uint32_t q = gridDim.x - 1u - blockIdx.x; q < guard_rows; q += gridDim.x)
      guard_compare(arity, S, rawbytes, nrec, recs, slots, slot_stride, segcap, gslots, grecs, gmeta, q / (uint32_t)arity, q % (uint32_t)arity, tid);
  // (a wave's rows of the fix-up: r, r + 16.  The records of the first are asked for now - they do not depend on the look-back -, those of
  // the second when the first is coded: a row is two memory round trips otherwise, and the noisy component's workgroup had 2 x 2 of them
  // behind the look-back: 31.7 -> 30.2 us)
  const uint32_t rA = tid >> 6, rB = rA + SF_THREADS / 64u;
  static_assert((uint32_t)CH <= 2u * (SF_THREADS / 64u), "two rows per wave");
  FixRecs recA;
  uint32_t HA = 0;
  if (rA < cnt)
    {
    const uint32_t rowi = (g0 + rA) * (uint32_t)arity + c;
    HA = nrec[rowi] & 0xffffu;
    fix_load(recA, recs + (size_t)rowi * RCAP * RECW, HA, 0u, tid & 63u);
    }
  // the chunks before mine (the second column: the first wave only)
  bool late = false;
  const uint32_t spin_max = (HOOK && (sabotage & 2u) != 0u) ? 512u : SPIN_MAX;
  const uint32_t carry0 = look_back(agg, nch, y, (uint32_t)arity, c, k0, last0_kept, late, spin_max);
  uint32_t carry1 = 0;
  if (two)
    carry1 = look_back(agg, nch, y, (uint32_t)arity, c, k1, last1_kept, late, spin_max);
  if (late)
    atomicOr(&nrec[c], FLAG_SCAN << 16);                   // (row 0 of the component: k_fpc32_offsets collects the flags per component)
  cin[k0] = carry0;
  if (two)
    cin[k1] = carry1;
  __syncthreads();
  {
  const uint32_t lane = tid & 63u;
  FixRecs recB;
  uint32_t HB = 0;
  if (rB < cnt)
    {
    HB = Hs[rB];
    fix_load(recB, recs + (size_t)((g0 + rB) * (uint32_t)arity + c) * RCAP * RECW, HB, 0u, lane);
    }
  if (rA < cnt)
    fixup_row_wave(recA, M + rA * (uint32_t)TAB, cin, HA, segbytes, rawbytes, recs, S, (uint32_t)arity, (g0 + rA) * (uint32_t)arity + c, lane);
  if (rB < cnt)
    fixup_row_wave(recB, M + rB * (uint32_t)TAB, cin, HB, segbytes, rawbytes, recs, S, (uint32_t)arity, (g0 + rB) * (uint32_t)arity + c, lane);
  }
  }

// the scan kernel's LDS is beyond the 64 KiB a kernel gets without asking: claimed once per device of the process
static bool scanfix_lds_claimed()
  {
  static std::atomic<int> state[16];             // 0 not asked, 1 claimed, 2 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16)
    return hipFuncSetAttribute((const void*)k_fpc32_scanfix<SWEEP_HOOK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SF_LDS_WORDS * 4u)) == hipSuccess;
  int st = state[dev].load();
  if (st == 0)
    {
    st = hipFuncSetAttribute((const void*)k_fpc32_scanfix<SWEEP_HOOK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SF_LDS_WORDS * 4u)) == hipSuccess ? 1 : 2;
    state[dev].store(st);
    }
  return st == 1;
  }

// ---- gather: segment slots -> contiguous payload ----------------------------------------------------------------------------
// grid (S, components); each workgroup moves one segment.  Without records: the destination is written as aligned 16-byte vectors,
// the source (a 256-byte aligned slot) is read as 4 + 1 dwords per vector and re-aligned with v_alignbyte.
// With records the slot is cut into sub-chunks of 2 KiB of SOURCE bytes (more for slots beyond 1 MiB).  A pass over the records
// counts the unused bytes of every sub-chunk - a prefix sum says where its output begins - and marks the ones a record touches.
// Then every WAVE takes sub-chunks by itself, no barrier: an untouched one is the copy with a shift again; a touched one passes
// through the wave's own LDS, a piece of 64 x 32 bytes at a time (64 x 16 until round 6: what a piece costs whatever it holds - the loads a
// piece ahead, the pass over the records, the scan, the flush of the ring - is ~210 of ~390 instructions; same box, alternating,
// benchmark mesh 150.5-153.1 -> 140.3-141.1 us): the records put their residual bytes and code bits into a patch area (ORed
// into the lanes' bytes: the sweep left zeros there) and mark the unused bytes of their fields, a wave scan over the valid bytes
// of every lane says where they go, and they land in a small output ring whose positions are congruent to the destination's
// modulo 16 and leave it as aligned 16-byte vectors.  Every output byte is written once, by the wave that owns it; the partial
// vectors at the ends of a sub-chunk's output go byte by byte.  Nothing is searched: a record knows its slot position.
// sub-chunks of 2 KiB of a slot (round 6; 1 KiB until then.  Same box, alternating: gather 166 -> 158 us on the benchmark mesh, 155 -> 153 us
// on the walk mesh; 4 KiB is slower again by as much - fewer sub-chunks mean fewer set-ups, larger ones more bytes through the record path
// for one record.  The boxes themselves differ by +-10 us on this kernel.)
#ifndef TRICO_GSUBSHIFT
#define TRICO_GSUBSHIFT 11
#endif
#ifndef TRICO_GLANEB
#define TRICO_GLANEB 32
#endif

constexpr uint32_t GLB = TRICO_GLANEB;             // source bytes of a piece a lane takes: 16, 32 or 64 (64 with sub-chunks of 4 KiB: 172 us against 133-139)
static_assert(GLB == 16u || GLB == 32u || GLB == 64u, "a lane's unused bytes are one 64-bit mask");
constexpr uint32_t GVN = GLB / 16u;
constexpr uint32_t GSUB = 64u * GLB;               // source bytes per piece
constexpr uint32_t MAXSUB = 512;                   // sub-chunks per slot (LDS: 8 workgroups per compute unit)
constexpr uint32_t WRING = GSUB + 64;               // a wave's output buffer: at most 15 + GSUB bytes are in it at a time (+ the slack of a 16-byte store)
struct GatherDst { uint8_t* p[3]; };

// 16 bytes in a row to LDS byte address ad.  An LDS dword store at an address that is not a multiple of four costs about eight
// aligned ones on gfx950, so: up to three single bytes until the next dword boundary, the dwords from there on (the vector shifted
// accordingly), and the up to three bytes behind the last whole dword.
__device__ __forceinline__ void lds_store16(uint32_t ad, const u32x4& vec)
  {
  // (the unwanted stores sent to a dump dword instead of around seven branches: no faster on the benchmark mesh, 8 us slower on the walk mesh)
  const uint32_t r = ad & 3u, lead = (4u - r) & 3u;              // bytes in front of the first aligned dword
  lds_vu8* ob = (lds_vu8*)(uintptr_t)ad;
  if (lead > 0u) ob[0] = (uint8_t)vec[0];
  if (lead > 1u) ob[1] = (uint8_t)(vec[0] >> 8);
  if (lead > 2u) ob[2] = (uint8_t)(vec[0] >> 16);
  lds_u32* ow = (lds_u32*)(uintptr_t)(ad + lead);
  ow[0] = __builtin_amdgcn_alignbyte(vec[1], vec[0], lead);
  ow[1] = __builtin_amdgcn_alignbyte(vec[2], vec[1], lead);
  ow[2] = __builtin_amdgcn_alignbyte(vec[3], vec[2], lead);
  if (r == 0u)
    ow[3] = vec[3];
  lds_vu8* oe = (lds_vu8*)(uintptr_t)(ad + 16u - r);
  if (r > 0u) oe[0] = (uint8_t)(vec[3] >> (8u * (4u - r)));
  if (r > 1u) oe[1] = (uint8_t)(vec[3] >> (8u * (5u - r)));
  if (r > 2u) oe[2] = (uint8_t)(vec[3] >> 24);
  }

// `len` bytes from sc (4-byte aligned) to dd, by one wave or one workgroup (`threads` of them, this one is `t0`)
__device__ __forceinline__ void copy_shifted(uint8_t* __restrict__ dd, const uint8_t* __restrict__ sc, uint32_t len, uint32_t t0, uint32_t threads)
  {
  const uint32_t head = (uint32_t)((16u - ((uintptr_t)dd & 15u)) & 15u);         // bytes until dd is 16-byte aligned
  const uint32_t h = head < len ? head : len;
  const uint32_t body = (len - h) >> 4;                                           // aligned destination vectors
  const uint32_t done = h + 16u * body;
  if (t0 < h)
    dd[t0] = sc[t0];
  const uint32_t* ss = (const uint32_t*)sc + (h >> 2);
  const uint32_t sh = h & 3u;
  u32x4* dv = (u32x4*)(dd + h);
  // destination vector t holds source bytes h + 16t .. h + 16t + 15
  for (uint32_t t = t0; t < body; t += threads)
    {
    const u32x4 lo = *(const u32x4*)(ss + 4u * t);                                // 4-byte aligned 16-byte load
    const uint32_t hi = ss[4u * t + 4u];
    u32x4 o;
    o[0] = __builtin_amdgcn_alignbyte(lo[1], lo[0], sh);
    o[1] = __builtin_amdgcn_alignbyte(lo[2], lo[1], sh);
    o[2] = __builtin_amdgcn_alignbyte(lo[3], lo[2], sh);
    o[3] = __builtin_amdgcn_alignbyte(hi, lo[3], sh);
    __builtin_nontemporal_store(o, dv + t);
    }
  if (t0 < len - done)
    dd[done + t0] = sc[done + t0];
  }

__global__ void __launch_bounds__(256) k_fpc32_gather(const uint8_t* __restrict__ slots, size_t slot_stride, uint32_t segcap, uint32_t S,
                                                      const uint32_t* __restrict__ segbytes, const uint32_t* __restrict__ rawbytes,
                                                      const uint32_t* __restrict__ segoff, GatherDst dst, const uint32_t* __restrict__ nrec,
                                                      const uint32_t* __restrict__ recs, int arity, int c0, const uint32_t* __restrict__ rectot,
                                                      const uint32_t* __restrict__ sizes)
  {
  const uint32_t g = blockIdx.x, tid = threadIdx.x;
  // The components in the order of their records (k_fpc32_offsets counts them), most first: a component full of them (a grid's z:
  // three quarters of the bytes, every sub-chunk through the slow path) takes its workgroups several times as long as the others,
  // and dispatched last its second round of workgroups was the kernel's tail (same box, benchmark mesh: 180-187 -> 156 us; by payload
  // bytes 157, simply reversed 161; the walk mesh, whose components are alike, does not care: 139-141).
  uint32_t c = blockIdx.y;
  if (gridDim.y > 1u)
    {
    uint32_t tot[3] = { 0u, 0u, 0u }, ord[3] = { 0u, 1u, 2u };    // (stable: equal counts keep their order)
    for (uint32_t k = 0; k < gridDim.y; ++k)
      tot[k] = rectot[(uint32_t)c0 + k];
    if (tot[ord[1]] > tot[ord[0]]) { const uint32_t t = ord[0]; ord[0] = ord[1]; ord[1] = t; }
    if (gridDim.y > 2u && tot[ord[2]] > tot[ord[1]]) { const uint32_t t = ord[1]; ord[1] = ord[2]; ord[2] = t; }
    if (tot[ord[1]] > tot[ord[0]]) { const uint32_t t = ord[0]; ord[0] = ord[1]; ord[1] = t; }
    c = ord[blockIdx.y];
    }
  const uint32_t cc = (uint32_t)c0 + c;                                           // component in the workspace's numbering
  const uint32_t len = segbytes[(size_t)cc * S + g];
  const uint8_t* s = slots + (size_t)cc * slot_stride + (size_t)g * segcap;       // 256-byte aligned, segcap has 280 bytes of slack
  uint8_t* d = dst.p[c];
  if (sizes)
    {
    // Chained destinations (the archive writer, device buffer): dst.p[0] is where the stream's FIRST size field goes; every component
    // is `u32 bytes, payload` right behind the component before it (trico.c:215-262), and where that is the sizes in device memory say
    // (k_fpc32_offsets) - the host need not have read them when this launch is queued.  Segment 0 writes the size field.
    d = dst.p[0] + 4u;
    for (uint32_t k = 0; k < cc; ++k)
      d += sizes[k] + 4u;
    if (g == 0u && tid < 4u)
      d[(int)tid - 4] = (uint8_t)(sizes[cc] >> (8u * tid));
    }
  d += segoff[(size_t)cc * S + g];
  const size_t rowi = (size_t)g * arity + cc;
  const uint32_t H = nrec[rowi] & 0xffffu;
  if (H == 0u)
    {
    copy_shifted(d, s, len, tid, 256u);
    return;
    }
  __shared__ uint32_t sub_out[MAXSUB + 1];                       // output offset of sub-chunk k (first: its unused bytes)
  __shared__ uint16_t rfirst[MAXSUB + 2];                        // first record whose field ends in sub-chunk >= k
  __shared__ uint32_t touched[MAXSUB / 32];                      // a record has a byte in sub-chunk k
  __shared__ uint32_t wsum[4], next_sub;
  __shared__ __attribute__((aligned(16))) uint32_t wlds[4][(GSUB + GSUB / 8 + WRING + 32 + 64) / 4];  // per wave: patches, unused marks, output ring, a dump byte per lane
  // (the patches inside the ring, which is idle while they are read: 147 against 133-139 us; fewer workgroups per unit through more LDS: no difference down to five)
  const uint32_t slen = rawbytes[(size_t)cc * S + g];
  const uint32_t* list = recs + rowi * RCAP * RECW;
  uint32_t subshift = TRICO_GSUBSHIFT;
  while (((slen - 1u) >> subshift) >= MAXSUB)
    ++subshift;
  const uint32_t subsize = 1u << subshift, nsub = ((slen - 1u) >> subshift) + 1u;
  for (uint32_t k = tid; k <= nsub; k += 256u)
    sub_out[k] = 0u;
  if (tid < MAXSUB / 32u)
    touched[tid] = 0u;
  if (tid == 0)
    next_sub = 4u;
  __syncthreads();
  uint32_t* rpos = &wlds[0][0];                                  // [H] the records' slot positions (the waves' buffers are idle until the setup is done)
  for (uint32_t j = tid; j < H; j += 256u)
    {
    const uint32_t pos = list[RECW * j] & REC_POS, hdr = list[RECW * j + 1u], ln = list[RECW * j + 7u] & 15u;
    rpos[j] = pos;
    for (uint32_t bb = ln; bb < 4u; ++bb)
      atomicAdd(&sub_out[(pos + bb) >> subshift], 1u);
    const uint32_t k0 = hdr >> subshift, k1 = (pos + 3u) >> subshift;           // (at most 34 bytes apart)
    atomicOr(&touched[k0 >> 5], 1u << (k0 & 31u));
    atomicOr(&touched[k1 >> 5], 1u << (k1 & 31u));
    }
  __syncthreads();
  for (uint32_t k = tid; k <= nsub; k += 256u)
    {
    // first record whose field ends at or behind the start of sub-chunk k (the records are in slot order); the search runs in LDS:
    // ten dependent loads from memory per workgroup were a third of this kernel's time
    const uint32_t lim = k << subshift;
    uint32_t lo = 0, hi = H;
    while (lo < hi)
      {
      const uint32_t mid = (lo + hi) >> 1;
      if (rpos[mid] + 3u < lim) lo = mid + 1u; else hi = mid;
      }
    rfirst[k] = (uint16_t)lo;
    }
  __syncthreads();
  {
  // exclusive prefix sum of the sub-chunks' valid bytes
  const uint32_t per = (nsub + 255u) / 256u;
  const uint32_t k0 = tid * per < nsub ? tid * per : nsub, k1 = k0 + per < nsub ? k0 + per : nsub;
  uint32_t sum = 0;
  for (uint32_t k = k0; k < k1; ++k)
    {
    const uint32_t sl = slen - (k << subshift);
    sum += (sl < subsize ? sl : subsize) - sub_out[k];
    }
  const uint32_t incl = wave_scan_incl(sum);
  if ((tid & 63u) == 63u)
    wsum[tid >> 6] = incl;
  __syncthreads();
  uint32_t run = incl - sum;
  for (uint32_t w = 0; w < (tid >> 6); ++w)
    run += wsum[w];
  for (uint32_t k = k0; k < k1; ++k)
    {
    const uint32_t sl = slen - (k << subshift);
    const uint32_t valid = (sl < subsize ? sl : subsize) - sub_out[k];
    sub_out[k] = run;
    run += valid;
    }
  if (tid == 255u)
    sub_out[nsub] = run;                                           // (= len)
  }
  __syncthreads();
  // The sub-chunks no record touches first, by the whole workgroup: ONE loop over their 16-byte vectors (a copy with a shift as for
  // a segment without records; a loop per stretch of such sub-chunks paid a memory round trip per stretch).
  {
  const uint32_t vshift = subshift - 4u, per = 1u << vshift;     // destination vectors a sub-chunk can have
  for (uint32_t idx = tid; idx < (nsub << vshift); idx += 256u)
    {
    const uint32_t kk = idx >> vshift, i = idx & (per - 1u);
    if ((touched[kk >> 5] >> (kk & 31u)) & 1u)
      continue;
    const uint32_t sb0 = kk << subshift;
    const uint32_t sl = slen - sb0 < subsize ? slen - sb0 : subsize;
    uint8_t* dd0 = d + sub_out[kk];
    const uint8_t* sc = s + sb0;
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)dd0 & 15u)) & 15u);
    const uint32_t h = head < sl ? head : sl;
    const uint32_t body = (sl - h) >> 4, done = h + 16u * body;
    if (i < body)
      {
      const uint32_t* ss = (const uint32_t*)sc + (h >> 2) + 4u * i;
      const uint32_t sh = h & 3u;
      const u32x4 lo = *(const u32x4*)ss;
      const uint32_t hi = ss[4];
      u32x4 o;
      o[0] = __builtin_amdgcn_alignbyte(lo[1], lo[0], sh);
      o[1] = __builtin_amdgcn_alignbyte(lo[2], lo[1], sh);
      o[2] = __builtin_amdgcn_alignbyte(lo[3], lo[2], sh);
      o[3] = __builtin_amdgcn_alignbyte(hi, lo[3], sh);
      __builtin_nontemporal_store(o, (u32x4*)(dd0 + h) + i);
      }
    if (i < h)
      dd0[i] = sc[i];
    if (i < sl - done)
      dd0[done + i] = sc[done + i];
    }
  }
  // the others: from here on every wave is on its own
  const uint32_t lane = tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  uint8_t* PB = (uint8_t*)wlds[wv];                              // [GSUB] patches of the piece
  uint32_t* U = wlds[wv] + GSUB / 4;                             // [GSUB / 32] its unused bytes, one bit each
  uint8_t* O = (uint8_t*)(wlds[wv] + (GSUB + GSUB / 8) / 4);     // [WRING + 32] output ring
  const uint32_t obase = (uint32_t)(uintptr_t)(lds_u8*)O;
  uint32_t k = wv;
  while (k < nsub)
    {
    const uint32_t sb0 = k << subshift;
    const uint32_t sl = slen - sb0 < subsize ? slen - sb0 : subsize;             // source bytes of this sub-chunk
    uint8_t* dd0 = d + sub_out[k];
    if ((touched[k >> 5] >> (k & 31u)) & 1u)
      {
      // buffer position p <-> destination byte rbase + p, rbase 16-byte aligned; after every piece the whole vectors leave and
      // the (fewer than 16) bytes behind them move to the front: positions never wrap
      uint32_t A = (uint32_t)((uintptr_t)dd0 & 15u);             // the first vector of the sub-chunk: its first A bytes are somebody else's
      uint8_t* rbase = dd0 - A;
      uint32_t wpos = A;
      uint32_t rj = rfirst[k];
      // The piece in hand and the loads of the next one: its 16 bytes per lane and the record each lane looks at first (a piece
      // rarely has more than 64).  A wave has nobody to cover its memory round trips here, so they are started a piece ahead.
      struct Rec { uint32_t pos, hdr, gi, x, lc; };
      auto load_rec = [&](uint32_t j) -> Rec
        {
        Rec r = { 0xffffffffu, 0u, 0u, 0u, 0u };
        if (j < H)
          {
          const u32x2 w = *(const u32x2*)(list + RECW * j);
          const uint32_t* t = list + RECW * j + 6u;
          r.pos = w[0] & REC_POS; r.gi = (w[0] >> REC_GI_SHIFT) & 7u; r.hdr = w[1]; r.x = t[0]; r.lc = t[1];
          }
        return r;
        };
      struct Vec { u32x4 v[GVN]; };
      auto load_vec = [&](uint32_t pb) -> Vec
        {
        Vec r;
#pragma unroll
        for (uint32_t h = 0; h < GVN; ++h)
          {
          const uint32_t sb = sb0 + pb + GLB * lane + 16u * h;
          r.v[h] = u32x4{ 0u, 0u, 0u, 0u };
          if (sb < sb0 + sl)
            r.v[h] = *(const u32x4*)(s + sb);
          }
        return r;
        };
      Vec vnext = load_vec(0u);
      Rec rnext = load_rec(rj + lane);
      const uint32_t pbase = (uint32_t)(uintptr_t)(lds_u8*)PB;
      for (uint32_t pb = 0; pb < sl; pb += GSUB)
        {
        const uint32_t cb = sb0 + pb, sb = cb + GLB * lane;
        const uint32_t pend = sl - pb < GSUB ? sb0 + sl : cb + GSUB;             // end of the piece in the slot
        Vec vec = vnext;
        const Rec r0 = rnext;
        const uint32_t rj0 = rj;
        // the records that are done with after this piece: their field ends inside it (they come first: slot order)
        {
        const uint32_t j = rj0 + lane;
        rj += (uint32_t)__popcll(__ballot(j < H && r0.pos + 3u < pend));
        }
        if (pb + GSUB < sl)
          {
          vnext = load_vec(pb + GSUB);
          rnext = load_rec(rj + lane);
          }
#pragma unroll
        for (uint32_t h = 0; h < GVN; ++h)
          *(u32x4*)(PB + GLB * lane + 16u * h) = u32x4{ 0u, 0u, 0u, 0u };
#pragma unroll
        for (uint32_t w = 0; w < (GSUB / 32u + 63u) / 64u; ++w)
          if (lane + 64u * w < GSUB / 32u)
            U[lane + 64u * w] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // the records with a byte in the piece: from the first whose field ends in it to the last whose header begins in it
        bool again = false;                                        // more than 64 of them: rnext was loaded before rj was final
        for (uint32_t j0 = rj0; j0 < H; j0 += 64u)
          {
          const uint32_t j = j0 + lane;
          const Rec r = j0 == rj0 ? r0 : load_rec(j);
          const uint32_t ln = r.lc & 15u, hdr = r.hdr, h24 = (r.lc >> 4) << (3u * r.gi);
          const bool in = j < H && hdr < pend;
          if (in)
            {
            const uint32_t q = r.pos - cb, hq = hdr - cb;          // (wrap to huge numbers before the piece)
            if (q <= GSUB - 4u)
              {
              // the whole field lies in the piece: its residual bytes, most significant first, with one store (nobody else has a
              // byte there), the unused ones behind them with one mask
              const uint32_t be = ln ? __builtin_bswap32(r.x << (8u * (4u - ln))) : 0u;
              asm volatile("ds_write_b32 %0, %1" :: "v"(pbase + q), "v"(be) : "memory");      // (not dword aligned: gfx950 executes it)
              const uint32_t um = (0xfu << ln) & 0xfu, sh = q & 31u;
              if (um)
                {
                atomicOr(&U[q >> 5], um << sh);
                if (sh > 28u)
                  atomicOr(&U[(q >> 5) + 1u], um >> (32u - sh));
                }
              }
            else
              for (uint32_t bb = 0; bb < 4u; ++bb)
                {
                const uint32_t qq = q + bb;
                if (qq < GSUB)
                  {
                  if (bb < ln)
                    PB[qq] = (uint8_t)(r.x >> (8u * (ln - 1u - bb)));
                  else
                    atomicOr(&U[qq >> 5], 1u << (qq & 31u));
                  }
                }
            if (h24)
              {
              if (hq <= GSUB - 3u)
                {
                // the three header bytes, big-endian, as one or two ORs on dwords (records of one group share its header)
                const uint32_t hw = __builtin_bswap32(h24 << 8), hs = 8u * (hq & 3u);
                atomicOr((uint32_t*)PB + (hq >> 2), hw << hs);
                if (hs > 8u)
                  atomicOr((uint32_t*)PB + (hq >> 2) + 1u, hw >> (32u - hs));
                }
              else
                for (uint32_t bb = 0; bb < 3u; ++bb)
                  {
                  const uint32_t qq = hq + bb;
                  const uint32_t by = (h24 >> (8u * (2u - bb))) & 255u;
                  if (qq < GSUB && by)
                    atomicOr((uint32_t*)PB + (qq >> 2), by << (8u * (qq & 3u)));
                  }
              }
            }
          if (j0 != rj0)
            rj += (uint32_t)__popcll(__ballot(j < H && r.pos + 3u < pend));     // (beyond the first 64: rnext was loaded too early)
          if (__ballot(in) != ~0ull)
            break;
          again = true;
          }
        if (again && pb + GSUB < sl)
          rnext = load_rec(rj + lane);                            // (not by looking at rnext: that would wait for the load just issued)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (uint32_t h = 0; h < GVN; ++h)
          {
          const u32x4 pt = *(const u32x4*)(PB + GLB * lane + 16u * h);
          vec.v[h][0] |= pt[0]; vec.v[h][1] |= pt[1]; vec.v[h][2] |= pt[2]; vec.v[h][3] |= pt[3];
          }
        // the unused bytes among the lane's GLB, one bit each
        constexpr uint64_t LMASK = GLB == 64u ? ~0ull : (1ull << (GLB & 63u)) - 1ull;
        uint64_t mu = GLB == 64u ? ((uint64_t)U[2u * lane + 1u] << 32) | U[2u * lane] : GLB == 32u ? (uint64_t)U[lane] : (uint64_t)((U[lane >> 1] >> (16u * (lane & 1u))) & 0xffffu);
        if (sb + GLB > pend)
          mu |= sb >= pend ? LMASK : (LMASK << (pend - sb)) & LMASK;             // beyond the sub-chunk's content
        const uint32_t cnt = GLB - (uint32_t)__popcll(mu);
        const uint32_t incl = wave_scan_incl(cnt);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t p = wpos + (incl - cnt);
        if (mu == 0ull)
          {
#pragma unroll
          for (uint32_t h = 0; h < GVN; ++h)
            lds_store16(obase + p + 16u * h, vec.v[h]);
          }
        else
          {
          // a lane with unused bytes among its own: byte by byte, an unused one to the lane's dump byte instead of a branch
          // around the store (sixteen branches were 150 instructions per piece, this is 80: gather 163 -> 153 us on the benchmark mesh,
          // 150 -> 144 us on the walk mesh, same box, alternating; every lane this way, without lds_store16: 159 us)
          uint32_t q = obase + p;
          const uint32_t dump = obase + WRING + 32u + lane;
#pragma unroll
          for (uint32_t bb = 0; bb < GLB; ++bb)
            {
            const uint32_t keep = ((uint32_t)(mu >> bb) & 1u) ^ 1u;
            *(lds_vu8*)(uintptr_t)(keep ? q : dump) = (uint8_t)(vec.v[bb >> 4][(bb >> 2) & 3u] >> (8u * (bb & 3u)));
            q += keep;
            }
          }
        wpos += total;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // whole vectors leave (at most GSUB / 16 + 1 are there), what is behind them moves to the front
        const uint32_t nvec = wpos >> 4;
        for (uint32_t t = lane; t < nvec; t += 64u)
          {
          const u32x4 o = *(const u32x4*)(O + 16u * t);
          if (t > 0u || A == 0u)
            __builtin_nontemporal_store(o, (u32x4*)(rbase + 16u * t));
          else
            for (uint32_t bb = A; bb < 16u; ++bb)
              rbase[bb] = (uint8_t)(o[bb >> 2] >> (8u * (bb & 3u)));
          }
        if (nvec)
          {
          const uint32_t rest = wpos & 15u;
          uint32_t tb = 0;
          if (lane < rest)
            tb = O[16u * nvec + lane];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane < rest)
            O[lane] = (uint8_t)tb;
          rbase += 16u * nvec;
          wpos = rest;
          A = 0u;
          }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      // the bytes of the last, partial vector
      if (lane < 16u && lane < wpos && lane >= A)
        rbase[lane] = O[lane];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    // the next sub-chunk nobody has taken
    uint32_t nk = 0;
    if (lane == 0)
      nk = atomicAdd(&next_sub, 1u);
    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)nk);
    }
  }

unsigned guard_workgroups(const Plan& p) { return p.S < GUARD_WGS ? p.S : GUARD_WGS; }


} // namespace

int fpc32_sweep_resident_workgroups(int arity)
  {
  // per device and arity: what the occupancy calculator says for this kernel with its LDS, times the compute units; one workgroup
  // per compute unit less than that, because a launch that needs the very last place was seen to run in two rounds
  static std::atomic<int> cache[16][4];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16)
    dev = 0;
  if (arity < 1) arity = 1;
  if (arity > 3) arity = 3;
  int have = cache[dev][arity].load();
  if (have == 0)
    {
    static const int forced = [] { const char* e = tune_env("TRICO_FPC32_WAVES"); return e ? atoi(e) : 0; }();      // tuning knob: waves per sweep
    int per_cu = 0, cus = 0;
    const size_t lds = (size_t)arity * LDSW * 4 + 16;
    if (forced > 0)
      have = forced / arity;
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fpc32_sweep<SWEEP_HOOK, true>, 64 * arity, lds) == hipSuccess && per_cu > 0 &&
             hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      {
      static const int spare = [] { const char* e = tune_env("TRICO_FPC32_SPARE"); return e ? atoi(e) : 1; }();
      have = (per_cu - spare) * cus;
      if (getenv("TRICO_HIP_DEBUG"))
        fprintf(stderr, "trico_hip: float encoder sweep: %d workgroups of %d waves per compute unit x %d compute units (spare %d)\n", per_cu, arity, cus, spare);
      }
    else
      {
      (void)hipGetLastError();
      have = 7680 / arity;
      }
    if (have < 1)
      have = 1;
    cache[dev][arity].store(have);
    }
  return have;
  }

int fpc32_sweep_class_size()
  {
  static std::atomic<int> cache[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16)
    dev = 0;
  int have = cache[dev].load();
  if (have == 0)
    {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      {
      (void)hipGetLastError();
      cus = 256;
      }
    have = cus;
    cache[dev].store(have);
    }
  return have;
  }

int fpc32_sweep_stagger_permille()
  {
  static const int beta = [] { const char* e = tune_env("TRICO_FPC32_STAGGER"); const int v = e ? atoi(e) : STAGGER_DEFAULT; return v < 0 ? 0 : v > 800 ? 800 : v; }();
  return beta;
  }

int fpc32_sweep_class_weights(uint32_t K, int arity, double* w)
  {
  // measurements (test-hooks library): TRICO_FPC32_STAGGER_W = K comma-separated relative lengths
  static const char* env = tune_env("TRICO_FPC32_STAGGER_W");
  (void)arity;
  if (!env)
    return 0;
  uint32_t got = 0;
  const char* q = env;
  while (*q && got < (uint32_t)STAGGER_MAX)
    {
    char* end = nullptr;
    const double v = strtod(q, &end);
    if (end == q)
      break;
    w[got++] = v > 0.05 ? v : 0.05;
    q = *end == ',' ? end + 1 : end;
    }
  return got == K ? 1 : 0;
  }

int launch_fpc32_sweep(const uint32_t* d_src, uint32_t n, int arity, const Plan& p, uint8_t* d_ws)
  {
  hipStream_t st = current_stream();
  uint32_t* outT = (uint32_t*)(d_ws + p.off_summ);
  uint64_t* agg = (uint64_t*)(d_ws + p.off_agg);
  uint32_t* segbytes = (uint32_t*)(d_ws + p.off_segbytes);
  uint32_t* rawbytes = (uint32_t*)(d_ws + p.off_rawbytes);
  uint32_t* nrec = (uint32_t*)(d_ws + p.off_nrec);
  uint32_t* recs = (uint32_t*)(d_ws + p.off_recs);
  uint8_t* slots = d_ws + p.off_slots;
  uint8_t* gslots = d_ws + p.off_gslots;
  uint32_t* grecs = (uint32_t*)(d_ws + p.off_grecs);
  GuardMeta* gmeta = (GuardMeta*)(d_ws + p.off_gmeta);
#ifdef TRICO_SWEEP_DIAG
  static uint64_t* diag = [] { void* q = nullptr; (void)hipMalloc(&q, 8 * (64 + 8192 * 3 * 8)); return (uint64_t*)q; }();
#else
  uint64_t* diag = (uint64_t*)(d_ws + p.off_diag);
#endif
  static const uint32_t rblocks = [] { const char* e = tune_env("TRICO_FPC32_REPLAY"); return e ? (uint32_t)atoi(e) : 0u; }();     // blocks of 8 steps replayed in front of a segment (SW_RSTEP; 0: measured, not worth it with this gather)
  static const bool use_asm = [] { const char* e = tune_env("TRICO_FPC32_ASM"); return !(e && e[0] == '0'); }();      // (0: the compiled step, for A/B runs)
  static std::atomic<uint32_t> encodes{ 0 };
  const uint32_t count = encodes.fetch_add(1u);
  const uint32_t seed = count * 0x85EBCA6Bu;                                // which segments the guard samples: another set every encode

  const unsigned threads = 64u * (unsigned)arity;
  const size_t lds = (size_t)arity * LDSW * 4 + 16;
  const unsigned G = guard_workgroups(p);
#ifdef TRICO_HIP_TEST_HOOKS
  static const uint32_t sabotage = [] { const char* e = getenv("TRICO_HIP_ENCODE_SABOTAGE"); return e ? (uint32_t)atoi(e) : 0u; }();
#else
  const uint32_t sabotage = 0u;
#endif
  constexpr bool HOOK = SWEEP_HOOK;
  if (use_asm)
    hipLaunchKernelGGL((k_fpc32_sweep<HOOK, true>), dim3(p.S + G), dim3(threads), lds, st, d_src, n, arity, p.L, p.sg, p.S, outT,
                       slots, p.slot_stride, p.segcap, segbytes, rawbytes, nrec, recs, sabotage & 1u, seed, gslots, grecs, gmeta, diag, rblocks, agg, (uint32_t)p.agg_words);
  else
    hipLaunchKernelGGL((k_fpc32_sweep<HOOK, false>), dim3(p.S + G), dim3(threads), lds, st, d_src, n, arity, p.L, p.sg, p.S, outT,
                       slots, p.slot_stride, p.segcap, segbytes, rawbytes, nrec, recs, sabotage & 1u, seed, gslots, grecs, gmeta, diag, rblocks, agg, (uint32_t)p.agg_words);
#ifdef TRICO_SWEEP_DIAG
  {
  // diagnostic build: clocks of the wave of segment S / 2 of every component, printed per launch; every wave's timeline to a file
  const size_t words = 64 + (size_t)(p.S + G) * arity * 8;
  static uint64_t* h = (uint64_t*)malloc(8 * (64 + 8192 * 3 * 8));
  if (hipMemcpyAsync(h, diag, words * 8, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
    {
    for (int c = 0; c < arity; ++c)
      fprintf(stderr, "sweep diag c%d: %llu shader clocks, %llu ticks of 100 MHz -> %.0f MHz; %llu steps in the loop = %.0f clocks per step; %llu records\n", c,
              (unsigned long long)h[4 * c], (unsigned long long)h[4 * c + 1], h[4 * c + 1] ? 100.0 * (double)h[4 * c] / (double)h[4 * c + 1] : 0.0,
              (unsigned long long)h[4 * c + 2], h[4 * c + 2] ? (double)h[4 * c] / (double)h[4 * c + 2] : 0.0, (unsigned long long)h[4 * c + 3]);
    if (const char* f = getenv("TRICO_SWEEP_DIAG_FILE"))
      if (FILE* fp = fopen(f, "wb"))
        {
        const uint64_t hdr[4] = { p.S, G, (uint64_t)arity, p.L };
        fwrite(hdr, 8, 4, fp);
        fwrite(h + 64, 8, words - 64, fp);
        fclose(fp);
        }
    }
  }
#endif
  if (!scanfix_lds_claimed())
    {
    set_error("float encoder: cannot claim 135 KiB of LDS for the scan");
    return 0;
    }
  hipLaunchKernelGGL(k_fpc32_scanfix<HOOK>, dim3(p.nch * (unsigned)arity), dim3(SF_THREADS), SF_LDS_WORDS * 4u, st, outT, p.S, arity, p.nch, agg,
                     segbytes, rawbytes, nrec, recs, slots, p.slot_stride, p.segcap, gslots, grecs, gmeta, G * (unsigned)arity, sabotage);
  return hip_ok(hipGetLastError(), "fpc32 encode kernels (sweep)") ? 1 : 0;
  }

int launch_fpc32_gather_rec(const Plan& p, int arity, int c0, int count, const uint8_t* d_ws, uint8_t* const d_dst[3], const uint32_t* d_sizes)
  {
  GatherDst dst = { { d_dst[0], count > 1 ? d_dst[1] : nullptr, count > 2 ? d_dst[2] : nullptr } };
  hipLaunchKernelGGL(k_fpc32_gather, dim3(p.S, count), dim3(256), 0, current_stream(), d_ws + p.off_slots, p.slot_stride, p.segcap, p.S,
                     (const uint32_t*)(d_ws + p.off_segbytes), (const uint32_t*)(d_ws + p.off_rawbytes), (const uint32_t*)(d_ws + p.off_segoff),
                     dst, (const uint32_t*)(d_ws + p.off_nrec), (const uint32_t*)(d_ws + p.off_recs), arity, c0, (const uint32_t*)(d_ws + p.off_diag + 256), d_sizes);
  return hip_ok(hipGetLastError(), "k_fpc32_gather") ? 1 : 0;
  }

} // namespace fpc32
} // namespace trico
