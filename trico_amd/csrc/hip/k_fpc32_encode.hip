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
//            of the segment.  ~25 VALU instructions per step.
//   scan     (k_fpc32_scan_*): prefix-max over segments per class = the table every segment starts
//            with, as value indices (0 = never written = the reference's zeroed table).
//   sweep C  (k_fpc32_code):   loads the incoming table (payloads gathered from the input by index),
//            then per step: the latest earlier value of my class is the previous lane inside a run
//            of equal classes; run starts look at the wave-private LDS table; several runs of one
//            class inside a step are detected with a lane-id table and fixed up with ballots.
//            Codes, residual lengths, wave prefix sums (mbcnt), 3-byte group headers (DPP or-reduce),
//            bytes staged in an LDS ring and flushed as aligned dwords into the segment's slot.
//   offsets  (k_fpc32_offsets): exclusive scan of the segment byte counts per component.
//   gather   (k_fpc32_gather): slot -> final position (this is the copy the reference does with
//            memcpy into the archive, trico.c:57-63; it can target the archive buffer directly).
//
// HBM traffic: 2 x raw input + 2 x payload bytes + ~5 % table traffic.  No MFMA: integer
// bit-twiddling bounded by HBM bandwidth; algorithmic bytes per value = 4 + its payload share.
#include "common.hpp"
#include <stdlib.h>

namespace trico {

namespace {

constexpr int TAB = 1040;      // 16 FCM entries followed by 1024 DFCM entries
constexpr int ROW = 1040;      // words per (segment, component) row in the global index tables
constexpr int CH = 32;         // segments per chunk in the cross-segment scan
constexpr int RING = 1024;     // bytes of per-wave output staging ring
constexpr int MASKW = 2 * TAB;  // words of the per-class lane-mask table (u64 per class)
constexpr int LDSW_A = TAB;                        // per-wave LDS words, sweep A
constexpr int LDSW_C = TAB + RING / 4 + MASKW;     // per-wave LDS words, sweep C (13,504 B: 4 x 3 waves per CU)
constexpr int PF = 8;          // steps (of 64 values) whose loads are kept in flight per wave

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
__device__ __forceinline__ void load_block(uint32_t (&r)[PF], const uint32_t* __restrict__ src, uint32_t i0, uint32_t i_end,
                                           int arity, int c, int lane)
  {
#pragma unroll
  for (int pu = 0; pu < PF; ++pu)
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

// ---- sweep A: last writer index (+1) per class of every segment ---------------------------------------
// "Last writer" is a maximum over value indices, so a segment may be swept by several waves at once:
// every segment is cut into ISPLIT sub-ranges (grid.y), each with its own LDS table, combined into the
// segment's row with global atomicMax (the rows are zeroed before the launch).
constexpr uint32_t ISPLIT = 2;

__global__ void __launch_bounds__(192) k_fpc32_index(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L,
                                                     uint32_t* __restrict__ summ)
  {
  extern __shared__ uint32_t lds[];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x;
  uint32_t* T = lds + c * LDSW_A;
  for (int i = lane; i < TAB; i += 64)
    T[i] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint32_t sub = L / ISPLIT;                       // L is a multiple of 64 * ISPLIT
  const uint32_t seg_end = (n - g * L < L) ? n : g * L + L;
  const uint32_t i_begin = g * L + blockIdx.y * sub;
  if (i_begin >= seg_end)
    return;
  const uint32_t i_end = (seg_end - i_begin < sub) ? seg_end : i_begin + sub;
  Carry cy = load_carry(src, i_begin, arity, c);
  uint32_t cur[PF], nxt[PF];
  load_block(cur, src, i_begin, i_end, arity, c, lane);
  for (uint32_t ib = i_begin; ib < i_end; ib += 64u * PF)
    {
    load_block(nxt, src, ib + 64u * PF, i_end, arity, c, lane);
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      {
      const uint32_t i0 = ib + 64u * pu;
      if (i0 >= i_end)
        break;
      const uint32_t i = i0 + lane;
      const bool act = i < i_end;
      const uint32_t v = cur[pu];
      uint32_t a, b, k1, k2;
      classes(v, cy, act, a, b, k1, k2);
      // only the last lane of a run of equal classes can be the class's last writer in this step
      const uint32_t kn1 = dpp_shl1(0xfffffffeu, k1), kn2 = dpp_shl1(0xfffffffeu, k2);
      if (act && k1 != kn1) atomicMax(&T[k1], i + 1u);
      if (act && k2 != kn2) atomicMax(&T[k2], i + 1u);
      next_carry(cy, v);
      }
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      cur[pu] = nxt[pu];
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
                                                      uint32_t* __restrict__ chmax)
  {
  const uint32_t col = blockIdx.x * 256u + threadIdx.x;      // (component, class)
  const uint32_t ncol = (uint32_t)arity * TAB;
  if (col >= ncol)
    return;
  const uint32_t c = col / TAB, k = col % TAB;
  const uint32_t g0 = blockIdx.y * CH, g1 = (g0 + CH < S) ? g0 + CH : S;
  uint32_t m = 0;
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
  for (uint32_t g = g0; g < g1; ++g)
    {
    const size_t r = ((size_t)g * arity + c) * ROW + k;
    inc[r] = carry;
    carry = max(carry, summ[r]);
    }
  }

// ---- sweep C -------------------------------------------------------------------------------------------

// Who wrote my class last, inside this step?  Lanes form runs of equal class.  Inside a run it is the
// previous lane.  A run START needs the nearest lower lane of its class, which is the END lane of an
// earlier run: run ends OR their lane bit into M[class] (u64 per class, zero between steps), every lane
// reads its class's mask back, ends clear it.  Constant cost for any number of classes; both predictors
// (FCM classes [0,16), DFCM classes [16,1040)) are resolved together so their LDS round trips overlap.
//   src  : lane holding the latest earlier value of my class inside this step, -1 if none
//   last : I am the last value of my class in this step (I own the table write)
struct Pred { bool st1, st2, last1, last2; int src1, src2; };

__device__ __forceinline__ Pred wave_pred2(uint32_t k1, uint32_t k2, bool act, uint64_t* __restrict__ M, uint64_t lt, int lane)
  {
  Pred r;
  const uint32_t kp1 = dpp_shr1(0xfffffffeu, k1), kn1 = dpp_shl1(0xfffffffeu, k1);
  const uint32_t kp2 = dpp_shr1(0xfffffffeu, k2), kn2 = dpp_shl1(0xfffffffeu, k2);
  r.st1 = act && k1 != kp1;
  r.st2 = act && k2 != kp2;
  const bool en1 = act && k1 != kn1, en2 = act && k2 != kn2;
  const unsigned long long bit = 1ull << lane;
  if (en1) atomicOr((unsigned long long*)&M[k1], bit);
  if (en2) atomicOr((unsigned long long*)&M[k2], bit);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint64_t m1 = M[act ? k1 : 0u], m2 = M[act ? k2 : 16u];       // unconditional (broadcast) reads
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (en1) M[k1] = 0ull;
  if (en2) M[k2] = 0ull;
  const uint64_t lo1 = m1 & lt, lo2 = m2 & lt;
  r.src1 = r.st1 ? (lo1 ? 63 - __builtin_clzll(lo1) : -1) : lane - 1;
  r.src2 = r.st2 ? (lo2 ? 63 - __builtin_clzll(lo2) : -1) : lane - 1;
  r.last1 = en1 && (m1 >> lane) == 1ull;
  r.last2 = en2 && (m2 >> lane) == 1ull;
  return r;
  }

// store the bytes of ring word `w` (byte offset off inside the slot, multiple of 4) that lie below hi
__device__ __forceinline__ void store_span(uint8_t* __restrict__ gbase, uint32_t off, uint32_t w, uint32_t hi)
  {
  if (off + 4u <= hi)
    *(uint32_t*)(gbase + off) = w;
  else
    for (uint32_t bb = 0; bb < 4u; ++bb)
      if (off + bb < hi)
        gbase[off + bb] = (uint8_t)(w >> (8u * bb));
  }

__global__ void __launch_bounds__(192) k_fpc32_code(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L, uint32_t S,
                                                    const uint32_t* __restrict__ inc, uint8_t* __restrict__ slots, size_t slot_stride,
                                                    uint32_t segcap, uint32_t* __restrict__ segbytes)
  {
  extern __shared__ uint32_t lds[];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x;
  uint32_t* T = lds + c * LDSW_C;                      // [TAB] payload table, directly followed by the ring
  uint32_t* ringw = T + TAB;
  uint8_t* ring = (uint8_t*)ringw;
  uint64_t* M = (uint64_t*)(T + TAB + RING / 4);       // [TAB] lane masks
  for (int k = lane; k < TAB; k += 64)
    M[k] = 0ull;
  // incoming table: payload of the last writer of every class before this segment (0 if none)
  const uint32_t* row = inc + ((size_t)g * arity + c) * ROW;
  for (int k = lane; k < TAB; k += 64)
    {
    const uint32_t idx = row[k];
    uint32_t pay = 0;
    if (idx)
      {
      const uint32_t vi = src[(size_t)(idx - 1u) * arity + c];
      const uint32_t vp = idx >= 2u ? src[(size_t)(idx - 2u) * arity + c] : 0u;
      pay = k < 16 ? vi : vi - vp;
      }
    T[k] = pay;
    }
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t i_begin = g * L;
  const uint32_t i_end = (n - i_begin < L) ? n : i_begin + L;
  const uint32_t n8 = (n + 7u) & ~7u;
  uint8_t* gbase = slots + (size_t)c * slot_stride + (size_t)g * segcap;
  uint32_t pos = 0, flushed = 0;
  if (g == 0)
    {
    if (lane == 0)
      {
      ring[0] = 0x25;                       // (4/2) << 4 | (10/2), fpsc.c:120
      ring[1] = (uint8_t)(n >> 24); ring[2] = (uint8_t)(n >> 16); ring[3] = (uint8_t)(n >> 8); ring[4] = (uint8_t)n;
      }
    pos = 5u;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  Carry cy = load_carry(src, i_begin, arity, c);
  uint32_t cur[PF], nxt[PF];
  load_block(cur, src, i_begin, i_end, arity, c, lane);
  for (uint32_t ib = i_begin; ib < i_end; ib += 64u * PF)
    {
    load_block(nxt, src, ib + 64u * PF, i_end, arity, c, lane);
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      {
      const uint32_t i0 = ib + 64u * pu;
      if (i0 >= i_end)
        break;
      const uint32_t i = i0 + lane;
      const bool act = i < i_end;
      const uint32_t v = cur[pu];
      uint32_t a, b, k1, k2;
      classes(v, cy, act, a, b, k1, k2);
      const uint32_t s = v - a;
      const Pred pr = wave_pred2(k1, k2, act, M, lt, lane);
      // Branch-free LDS traffic: lanes that have nothing to read/write are redirected instead of masked
      // (every exec-mask region costs 3-4 instructions on the CU's single scalar unit).  The ring bytes
      // [pos+512, pos+768) are never live (at most 536 bytes are in flight), they serve as the dump.
      const uint32_t dump = (pos + 512u) & (RING - 1);
      const uint32_t tv1 = T[act ? k1 : 0u], tv2 = T[act ? k2 : 0u];
      const uint32_t q1 = (uint32_t)__builtin_amdgcn_ds_bpermute(pr.src1 << 2, (int)v);
      const uint32_t q2 = (uint32_t)__builtin_amdgcn_ds_bpermute(pr.src2 << 2, (int)s);
      const uint32_t p1 = pr.st1 ? (pr.src1 >= 0 ? q1 : tv1) : a;            // inside a run: previous lane's value
      const uint32_t p2 = pr.st2 ? (pr.src2 >= 0 ? q2 : tv2) : a - b;        //               previous lane's stride
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint32_t dumpw = TAB + (((dump >> 2) + (uint32_t)lane) & (RING / 4 - 1));
      T[pr.last1 ? k1 : dumpw] = v;
      T[pr.last2 ? k2 : dumpw] = s;
      uint32_t len, x;
      uint32_t code = pick(v ^ p1, v ^ (a + p2), len, x);
      const bool slot = act || (i_end == n && i < n8);          // value or tail padding slot (fpsc.c:196-204)
      if (!act)
        {
        code = slot ? 1u : 0u;
        len = slot ? 1u : 0u;
        x = 0u;
        }
      // byte layout of the step: [hdr g0][residuals 0..7][hdr g1][residuals 8..15]...
      const uint64_t b0 = __ballot(len & 1u), b1 = __ballot(len & 2u), b2 = __ballot(len & 4u);
      const uint32_t pre = popc_below(b0) + 2u * popc_below(b1) + 4u * popc_below(b2);
      uint32_t bc = code << (3u * (lane & 7u));
      bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
      bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
      bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0x141, 0xf, 0xf, true);    // row_half_mirror
      const uint32_t grp = lane >> 3;
      const uint32_t rpos = pos + 3u * (grp + 1u) + pre;
      const uint32_t dumpb = dump + 4u * (uint32_t)lane;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (uint32_t kb = 0; kb < 4u; ++kb)
        ring[(len > kb ? rpos + kb : dumpb + kb) & (RING - 1)] = (uint8_t)(x >> (8u * ((len - 1u - kb) & 3u)));
      {
      const bool lead = slot && (lane & 7) == 0;
      const uint32_t hpos = lead ? pos + 3u * grp + pre : dumpb;
      ring[hpos & (RING - 1)] = (uint8_t)(bc >> 16);
      ring[(hpos + 1u) & (RING - 1)] = (uint8_t)(bc >> 8);
      ring[(hpos + 2u) & (RING - 1)] = (uint8_t)bc;
      }
      const uint32_t nslots = (uint32_t)__popcll(__ballot(slot));
      pos += 3u * (nslots >> 3) + (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      while (pos - flushed >= 256u)
        {
        const uint32_t off = flushed + 4u * lane;
        *(uint32_t*)(gbase + off) = ringw[(off & (RING - 1)) >> 2];
        flushed += 256u;
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      next_carry(cy, v);
      }
#pragma unroll
    for (int pu = 0; pu < PF; ++pu)
      cur[pu] = nxt[pu];
    }
  while (flushed < pos)
    {
    const uint32_t off = flushed + 4u * lane;
    if (off < pos)
      store_span(gbase, off, ringw[(off & (RING - 1)) >> 2], pos);
    flushed += 256u;
    }
  if (lane == 0)
    segbytes[(size_t)c * S + g] = pos;
  }

// ---- offsets: exclusive scan of segment sizes per component (one workgroup per component) -------------
__global__ void __launch_bounds__(1024) k_fpc32_offsets(const uint32_t* __restrict__ segbytes, uint32_t S, uint32_t* __restrict__ segoff,
                                                        uint32_t* __restrict__ sizes)
  {
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
// grid (S, arity); each workgroup moves one segment.  Destination dwords are written aligned; the source
// is read as aligned dwords and re-aligned with v_alignbyte.
__global__ void __launch_bounds__(256) k_fpc32_gather(const uint8_t* __restrict__ slots, size_t slot_stride, uint32_t segcap, uint32_t S,
                                                      const uint32_t* __restrict__ segbytes, const uint32_t* __restrict__ segoff,
                                                      uint8_t* __restrict__ out, size_t out_stride)
  {
  const uint32_t g = blockIdx.x, c = blockIdx.y;
  const uint32_t len = segbytes[(size_t)c * S + g];
  const uint8_t* s = slots + (size_t)c * slot_stride + (size_t)g * segcap;       // 4-byte aligned
  uint8_t* d = out + (size_t)c * out_stride + segoff[(size_t)c * S + g];
  const uint32_t head = (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u);              // bytes until d is aligned
  const uint32_t h = head < len ? head : len;
  if (threadIdx.x < h)
    d[threadIdx.x] = s[threadIdx.x];
  const uint32_t body = (len - h) >> 2;                                           // aligned destination dwords
  uint32_t* dd = (uint32_t*)(d + h);
  const uint32_t* ss = (const uint32_t*)s;
  // destination dword t holds source bytes h + 4t .. h + 4t + 3  (h < 4)
  for (uint32_t t = threadIdx.x; t < body; t += 256u)
    {
    const uint32_t lo = ss[t], hi = ss[t + 1u];
    dd[t] = h ? __builtin_amdgcn_alignbyte(hi, lo, h) : lo;
    }
  const uint32_t done = h + 4u * body;
  if (threadIdx.x < len - done)
    d[done + threadIdx.x] = s[done + threadIdx.x];
  }

// n == 0: undefined in the reference; defined as header + one full pad group
__global__ void k_fpc32_empty(uint8_t* out, size_t out_stride, uint32_t* sizes)
  {
  uint8_t* o = out + (size_t)blockIdx.x * out_stride;
  const uint8_t bts[16] = { 0x25, 0, 0, 0, 0, 0x24, 0x92, 0x49, 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int i = 0; i < 16; ++i) o[i] = bts[i];
  sizes[blockIdx.x] = 16;
  }

struct Plan { uint32_t L, S, segcap, nch; size_t rows, slot_stride, off_summ, off_inc, off_chmax, off_segbytes, off_segoff, off_slots, total; };

Plan make_plan(uint32_t n, int arity)
  {
  static int waves = 0;
  if (!waves)
    {
    const char* e = getenv("TRICO_FPC32_WAVES");      // tuning knob: waves per sweep
    waves = e ? atoi(e) : 3072;
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
  p.off_segbytes = o;  o += align_up(p.rows * 4, 256);
  p.off_segoff = o;    o += align_up(p.rows * 4, 256);
  p.off_slots = o;     o += p.slot_stride * arity;
  p.total = o + 256;
  return p;
  }

} // namespace

size_t fpc32_encode_workspace(uint32_t n, int arity)
  {
  return make_plan(n, arity).total;
  }

int launch_fpc32_encode(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                        uint8_t* d_ws, size_t ws_bytes)
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
  if (!hip_ok(hipMemsetAsync(summ, 0, p.rows * ROW * 4, st), "memset(summ)"))
    return 0;
  hipLaunchKernelGGL(k_fpc32_index, dim3(p.S, ISPLIT), dim3(threads), (size_t)arity * LDSW_A * 4, st, src, n, arity, p.L, summ);
  const unsigned colblocks = ((unsigned)arity * TAB + 255u) / 256u;
  hipLaunchKernelGGL(k_fpc32_scan_a, dim3(colblocks, p.nch), dim3(256), 0, st, summ, p.S, arity, chmax);
  hipLaunchKernelGGL(k_fpc32_scan_b, dim3(colblocks, p.nch), dim3(256), 0, st, summ, p.S, arity, chmax, inc);
  hipLaunchKernelGGL(k_fpc32_code, dim3(p.S), dim3(threads), (size_t)arity * LDSW_C * 4, st,
                     src, n, arity, p.L, p.S, inc, slots, p.slot_stride, p.segcap, segbytes);
  hipLaunchKernelGGL(k_fpc32_offsets, dim3(arity), dim3(1024), 0, st, segbytes, p.S, segoff, d_sizes);
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
  hipLaunchKernelGGL(k_fpc32_gather, dim3(p.S, 1), dim3(256), 0, current_stream(), slots + (size_t)c * p.slot_stride, (size_t)0,
                     p.segcap, p.S, segbytes + (size_t)c * p.S, segoff + (size_t)c * p.S, d_dst, (size_t)0);
  return hip_ok(hipGetLastError(), "k_fpc32_gather") ? 1 : 0;
  }

} // namespace trico
