// k_fpc32_encode.hip — throughput encoder for 32-bit floating-point streams (gfx950, wave64).
//
// Replaces, fused: trico_transpose_xyz/uv_aos_to_soa (transpose_aos_to_soa.c:8-16, 48-56) and
// trico_compress(..., 4, 10) (fpsc.c:86-210) for every component of a vec3 / vec2 / scalar stream.
//
// Why this can be parallel and still bit-exact: the FCM hash of value
// i is a pure function of v[i-1] (top 4 bits) and the DFCM hash a pure function of v[i-3..i-1], so
// each value belongs to a *class* known from the input alone, and the reference's table read for
// value i returns the payload (value / stride) of the latest earlier value of the same class, or 0.
//
// Structure: each component stream is cut into S contiguous segments of L values (L % 64 == 0).
// One wave owns one (segment, component) and sweeps it 64 values per step with its own 16+1024
// entry table in LDS, exactly like the reference's tables but written once per class per step:
//   * inside a step the "latest earlier value of my class" is found with ballots over the distinct
//     classes present (1-3 iterations on smooth data), the payload moves by ds_bpermute;
//   * across steps it comes from the wave-private LDS table.
// What a segment cannot know is the table content at its start.  Pass A therefore sweeps with an
// empty table, records the few values whose class had not occurred yet in the segment (at most one
// per class) and the segment's end-of-segment table; B1 turns the per-segment tables into incoming
// tables with a last-writer scan across segments; B2 corrects the byte counts of the recorded values;
// B3 prefix-sums segment sizes; pass C sweeps again with the exact incoming table and packs bytes.
//
// HBM traffic: 2 x raw input (pass A + C) + payload bytes + ~6 % table traffic.  No MFMA: this is
// integer bit-twiddling bounded by HBM bandwidth; algorithmic bytes per value = 4 + payload share.
#include "common.hpp"

namespace trico {

namespace {

constexpr int TAB = 1040;      // 16 FCM entries followed by 1024 DFCM entries
constexpr int SEENW = 33;      // presence bits for TAB classes
constexpr int LDSW_A = TAB + 48;   // per-wave LDS words in pass A (table + seen bits, padded)
constexpr int ROW = 1088;      // words per (segment, component) row in the global tables
constexpr int UCAP = 1040;     // a class can be unresolved at most once per segment
constexpr int CH = 32;         // segments per chunk in the cross-segment scan
constexpr int RING = 1024;     // bytes of per-wave output staging ring in pass C
constexpr int LDSW_C = TAB + RING / 4;

struct UEntry { uint32_t v, a, meta, pk; };   // meta: k1 | (k2 << 4) | need1 << 16 | need2 << 17

__device__ __forceinline__ uint32_t dpp_shr1(uint32_t carry, uint32_t v)
  {
  // lane l <- lane l-1, lane 0 <- carry   (DPP wave_shr:1)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
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

// For every active lane: src = nearest lower active lane with the same key (-1 if none) and
// last = no higher active lane has this key.  Inactive lanes carry key 0xffffffff.
__device__ __forceinline__ void wave_pred(uint32_t key, uint64_t actmask, uint64_t lt, int lane, int& src, bool& last)
  {
  src = -1;
  last = false;
  uint64_t todo = actmask;
  while (todo)
    {
    const int leader = __builtin_ctzll(todo);
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
    const uint64_t m = __ballot(key == k);
    if (key == k)
      {
      const uint64_t lower = m & lt;
      src = lower ? 63 - __builtin_clzll(lower) : -1;
      last = (m >> lane) == 1ull;
      }
    todo &= ~m;
    }
  }

struct Carry { uint32_t m1, m2, m3; };

__device__ __forceinline__ Carry load_carry(const uint32_t* __restrict__ src, uint32_t i_begin, int arity, int c)
  {
  Carry k;
  k.m1 = i_begin >= 1u ? src[(size_t)(i_begin - 1u) * arity + c] : 0u;
  k.m2 = i_begin >= 2u ? src[(size_t)(i_begin - 2u) * arity + c] : 0u;
  k.m3 = i_begin >= 3u ? src[(size_t)(i_begin - 3u) * arity + c] : 0u;
  return k;
  }

// classes and stride of the 64 values of a step
__device__ __forceinline__ void classes(uint32_t v, const Carry& cy, bool act, uint32_t& a, uint32_t& s, uint32_t& k1, uint32_t& k2)
  {
  a = dpp_shr1(cy.m1, v);                    // v[i-1]
  const uint32_t b = dpp_shr1(cy.m2, a);     // v[i-2]
  const uint32_t d = dpp_shr1(cy.m3, b);     // v[i-3]
  s = v - a;                                 // stride of value i
  const uint32_t s1 = a - b, s2 = b - d;     // strides of values i-1, i-2
  k1 = a >> 28;                                                    // fpsc.c:76-79 with e1 = 4
  k2 = 16u + ((((s2 >> 22) & 31u) << 5) ^ (s1 >> 22));            // fpsc.c:81-84 with e2 = 10
  if (!act)
    k1 = k2 = 0xffffffffu;
  }

// ---- pass A ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(192) k_fpc32_pass_a(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L, uint32_t S,
                                                      uint32_t* __restrict__ summ, uint32_t* __restrict__ segbytes,
                                                      uint32_t* __restrict__ ucount, UEntry* __restrict__ ulist)
  {
  extern __shared__ uint32_t lds[];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x;
  uint32_t* T = lds + c * LDSW_A;
  uint32_t* seen = T + TAB;
  for (int i = lane; i < LDSW_A; i += 64)
    T[i] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t i_begin = g * L;
  const uint32_t i_end = (n - i_begin < L) ? n : i_begin + L;
  Carry cy = load_carry(src, i_begin, arity, c);
  uint32_t bytes = 0, ucnt = 0;
  UEntry* ul = ulist + ((size_t)g * arity + c) * UCAP;
  for (uint32_t i0 = i_begin; i0 < i_end; i0 += 64u)
    {
    const uint32_t i = i0 + lane;
    const bool act = i < i_end;
    const uint32_t v = act ? src[(size_t)i * arity + c] : 0u;
    uint32_t a, s, k1, k2;
    classes(v, cy, act, a, s, k1, k2);
    const uint64_t actmask = __ballot(act);
    int src1, src2;
    bool last1, last2;
    wave_pred(k1, actmask, lt, lane, src1, last1);
    wave_pred(k2, actmask, lt, lane, src2, last2);
    uint32_t p1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src1 << 2, (int)v);
    uint32_t p2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src2 << 2, (int)s);
    bool need1 = false, need2 = false;
    if (act && src1 < 0)
      {
      need1 = ((seen[k1 >> 5] >> (k1 & 31u)) & 1u) == 0u;
      p1 = need1 ? 0u : T[k1];
      }
    if (act && src2 < 0)
      {
      need2 = ((seen[k2 >> 5] >> (k2 & 31u)) & 1u) == 0u;
      p2 = need2 ? 0u : T[k2];
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (last1) { T[k1] = v; atomicOr(&seen[k1 >> 5], 1u << (k1 & 31u)); }
    if (last2) { T[k2] = s; atomicOr(&seen[k2 >> 5], 1u << (k2 & 31u)); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t len, x;
    pick(v ^ p1, v ^ (a + p2), len, x);
    if (!act)
      len = (i < ((n + 7u) & ~7u) && i_end == n) ? 1u : 0u;     // tail padding slots (fpsc.c:196-204)
    const uint32_t cnt = (i_end - i0 < 64u) ? i_end - i0 : 64u;
    bytes += 3u * ((cnt + 7u) >> 3);
    bytes += (uint32_t)__popcll(__ballot(len & 1u)) + 2u * (uint32_t)__popcll(__ballot(len & 2u)) + 4u * (uint32_t)__popcll(__ballot(len & 4u));
    const bool u = need1 || need2;
    const uint64_t um = __ballot(u);
    if (um)
      {
      if (u)
        {
        UEntry e;
        e.v = v;
        e.a = a;
        e.meta = k1 | ((k2 - 16u) << 4) | (need1 ? 1u << 16 : 0u) | (need2 ? 1u << 17 : 0u);
        e.pk = need1 ? p2 : p1;
        ul[ucnt + popc_below(um)] = e;
        }
      ucnt += (uint32_t)__popcll(um);
      }
    cy.m1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    cy.m2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 62);
    cy.m3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 61);
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t* row = summ + ((size_t)g * arity + c) * ROW;
  for (int i = lane; i < TAB + SEENW; i += 64)
    row[i] = T[i];
  if (lane == 0)
    {
    segbytes[(size_t)c * S + g] = bytes;
    ucount[(size_t)g * arity + c] = ucnt;
    }
  }

// ---- B1: incoming table of segment g = payload of the last earlier segment that wrote the class ----
// B1a: per chunk of CH segments, last written payload per class.  B1b: rewrite each chunk with the carry.
__global__ void __launch_bounds__(256) k_fpc32_b1a(const uint32_t* __restrict__ summ, uint32_t S, int arity,
                                                   uint32_t* __restrict__ chval, uint32_t* __restrict__ chseen)
  {
  const uint32_t col = blockIdx.x * 256u + threadIdx.x;      // (component, class)
  const uint32_t ncol = (uint32_t)arity * TAB;
  if (col >= ncol)
    return;
  const uint32_t c = col / TAB, k = col % TAB;
  const uint32_t g0 = blockIdx.y * CH, g1 = (g0 + CH < S) ? g0 + CH : S;
  uint32_t val = 0, has = 0;
  for (uint32_t g = g0; g < g1; ++g)
    {
    const uint32_t* row = summ + ((size_t)g * arity + c) * ROW;
    if ((row[TAB + (k >> 5)] >> (k & 31u)) & 1u)
      {
      val = row[k];
      has = 1u;
      }
    }
  chval[(size_t)blockIdx.y * ncol + col] = val;
  chseen[(size_t)blockIdx.y * ncol + col] = has;
  }

__global__ void __launch_bounds__(256) k_fpc32_b1b(const uint32_t* __restrict__ summ, uint32_t S, int arity,
                                                   const uint32_t* __restrict__ chval, const uint32_t* __restrict__ chseen,
                                                   uint32_t* __restrict__ inc)
  {
  const uint32_t col = blockIdx.x * 256u + threadIdx.x;
  const uint32_t ncol = (uint32_t)arity * TAB;
  if (col >= ncol)
    return;
  const uint32_t c = col / TAB, k = col % TAB;
  uint32_t carry = 0;
  for (int j = (int)blockIdx.y - 1; j >= 0; --j)
    if (chseen[(size_t)j * ncol + col])
      {
      carry = chval[(size_t)j * ncol + col];
      break;
      }
  const uint32_t g0 = blockIdx.y * CH, g1 = (g0 + CH < S) ? g0 + CH : S;
  for (uint32_t g = g0; g < g1; ++g)
    {
    const size_t r = ((size_t)g * arity + c) * ROW;
    inc[r + k] = carry;
    if ((summ[r + TAB + (k >> 5)] >> (k & 31u)) & 1u)
      carry = summ[r + k];
    }
  }

// ---- B2: exact byte count of the values pass A could not resolve ------------------------------------
__global__ void __launch_bounds__(64) k_fpc32_b2(const uint32_t* __restrict__ inc, int arity, uint32_t S,
                                                 const uint32_t* __restrict__ ucount, const UEntry* __restrict__ ulist,
                                                 uint32_t* __restrict__ segbytes)
  {
  const uint32_t gc = blockIdx.x;                 // g * arity + c
  const uint32_t g = gc / arity, c = gc % arity;
  const uint32_t cnt = ucount[gc];
  if (cnt == 0)
    return;
  const uint32_t* row = inc + (size_t)gc * ROW;
  const UEntry* ul = ulist + (size_t)gc * UCAP;
  int delta = 0;
  for (uint32_t e = threadIdx.x; e < cnt; e += 64u)
    {
    const UEntry u = ul[e];
    const uint32_t k1 = u.meta & 15u, k2 = 16u + ((u.meta >> 4) & 1023u);
    const bool need1 = (u.meta >> 16) & 1u, need2 = (u.meta >> 17) & 1u;
    const uint32_t p1 = need1 ? row[k1] : u.pk;
    const uint32_t p2 = need2 ? row[k2] : u.pk;
    const uint32_t q1 = need1 ? 0u : u.pk, q2 = need2 ? 0u : u.pk;     // what pass A assumed
    uint32_t lt, la, x;
    pick(u.v ^ p1, u.v ^ (u.a + p2), lt, x);
    pick(u.v ^ q1, u.v ^ (u.a + q2), la, x);
    delta += (int)lt - (int)la;
    }
  for (int o = 32; o > 0; o >>= 1)
    delta += __shfl_xor(delta, o);
  if (threadIdx.x == 0 && delta != 0)
    segbytes[(size_t)c * S + g] = (uint32_t)((int)segbytes[(size_t)c * S + g] + delta);
  }

// ---- B3: exclusive scan of segment sizes per component (one workgroup per component) ----------------
__global__ void __launch_bounds__(1024) k_fpc32_b3(const uint32_t* __restrict__ segbytes, uint32_t S, uint32_t* __restrict__ segoff,
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
  uint32_t run = 5u + part[threadIdx.x] - sum;       // 5 = stream header (fpsc.c:120-126)
  for (uint32_t g = g0; g < g1; ++g)
    {
    segoff[(size_t)c * S + g] = run;
    run += segbytes[(size_t)c * S + g];
    }
  if (threadIdx.x == 1023u)
    sizes[c] = 5u + part[1023];
  }

// ---- pass C -------------------------------------------------------------------------------------------
// store the bytes of ring word `w` (relative byte offset off, multiple of 4) that lie in [lo, hi)
__device__ __forceinline__ void store_span(uint8_t* __restrict__ gbase, uint32_t off, uint32_t w, uint32_t lo, uint32_t hi)
  {
  if (off >= lo && off + 4u <= hi)
    *(uint32_t*)(gbase + off) = w;
  else
    {
    for (uint32_t b = 0; b < 4u; ++b)
      if (off + b >= lo && off + b < hi)
        gbase[off + b] = (uint8_t)(w >> (8u * b));
    }
  }

__global__ void __launch_bounds__(192) k_fpc32_pass_c(const uint32_t* __restrict__ src, uint32_t n, int arity, uint32_t L, uint32_t S,
                                                      const uint32_t* __restrict__ inc, const uint32_t* __restrict__ segoff,
                                                      uint8_t* __restrict__ out, size_t out_stride)
  {
  extern __shared__ uint32_t lds[];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x;
  uint32_t* T = lds + c * LDSW_C;
  uint32_t* ringw = T + TAB;
  uint8_t* ring = (uint8_t*)ringw;
  const uint32_t* row = inc + ((size_t)g * arity + c) * ROW;
  for (int i = lane; i < TAB; i += 64)
    T[i] = row[i];
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t i_begin = g * L;
  const uint32_t i_end = (n - i_begin < L) ? n : i_begin + L;
  const uint32_t n8 = (n + 7u) & ~7u;
  // output window: ring byte r <-> global byte (A & ~3) + r
  const uint32_t A = (g == 0) ? 0u : segoff[(size_t)c * S + g];
  uint8_t* gbase = out + (size_t)c * out_stride + (A & ~3u);
  const uint32_t own_lo = A & 3u;
  uint32_t pos = own_lo, flushed = 0;
  if (g == 0)
    {
    if (lane == 0)
      {
      ring[0] = 0x25;                       // (4/2) << 4 | (10/2)
      ring[1] = (uint8_t)(n >> 24); ring[2] = (uint8_t)(n >> 16); ring[3] = (uint8_t)(n >> 8); ring[4] = (uint8_t)n;
      }
    pos = 5u;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  Carry cy = load_carry(src, i_begin, arity, c);
  for (uint32_t i0 = i_begin; i0 < i_end; i0 += 64u)
    {
    const uint32_t i = i0 + lane;
    const bool act = i < i_end;
    const uint32_t v = act ? src[(size_t)i * arity + c] : 0u;
    uint32_t a, s, k1, k2;
    classes(v, cy, act, a, s, k1, k2);
    const uint64_t actmask = __ballot(act);
    int src1, src2;
    bool last1, last2;
    wave_pred(k1, actmask, lt, lane, src1, last1);
    wave_pred(k2, actmask, lt, lane, src2, last2);
    uint32_t p1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src1 << 2, (int)v);
    uint32_t p2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src2 << 2, (int)s);
    if (act && src1 < 0) p1 = T[k1];
    if (act && src2 < 0) p2 = T[k2];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (last1) T[k1] = v;
    if (last2) T[k2] = s;
    uint32_t len, x;
    uint32_t code = pick(v ^ p1, v ^ (a + p2), len, x);
    const bool slot = act || (i_end == n && i < n8);          // value or tail padding slot
    if (!act)
      {
      code = slot ? 1u : 0u;
      len = slot ? 1u : 0u;
      x = 0u;
      }
    // byte offsets inside the step: [hdr g0][res 0..7][hdr g1][res 8..15]...
    const uint32_t pre = popc_below(__ballot(len & 1u)) + 2u * popc_below(__ballot(len & 2u)) + 4u * popc_below(__ballot(len & 4u));
    uint32_t bc = code << (3u * (lane & 7u));
    bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    bc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bc, 0x141, 0xf, 0xf, true);    // row_half_mirror
    const uint32_t grp = lane >> 3;
    const uint32_t rpos = pos + 3u * (grp + 1u) + pre;
    if (len > 0u) ring[(rpos) & (RING - 1)] = (uint8_t)(x >> (8u * (len - 1u)));
    if (len > 1u) ring[(rpos + 1u) & (RING - 1)] = (uint8_t)(x >> (8u * (len - 2u)));
    if (len > 2u) ring[(rpos + 2u) & (RING - 1)] = (uint8_t)(x >> (8u * (len - 3u)));
    if (len > 3u) ring[(rpos + 3u) & (RING - 1)] = (uint8_t)x;
    if (slot && (lane & 7) == 0)
      {
      const uint32_t hpos = pos + 3u * grp + pre;
      ring[hpos & (RING - 1)] = (uint8_t)(bc >> 16);
      ring[(hpos + 1u) & (RING - 1)] = (uint8_t)(bc >> 8);
      ring[(hpos + 2u) & (RING - 1)] = (uint8_t)bc;
      }
    const uint32_t nslots = (uint32_t)__popcll(__ballot(slot));
    pos += 3u * (nslots >> 3);
    pos += (uint32_t)__popcll(__ballot(len & 1u)) + 2u * (uint32_t)__popcll(__ballot(len & 2u)) + 4u * (uint32_t)__popcll(__ballot(len & 4u));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    while (pos - flushed >= 256u)
      {
      const uint32_t off = flushed + 4u * lane;
      store_span(gbase, off, ringw[(off & (RING - 1)) >> 2], own_lo, 0xffffffffu);
      flushed += 256u;
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    cy.m1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    cy.m2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 62);
    cy.m3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 61);
    }
  while (flushed < pos)
    {
    const uint32_t off = flushed + 4u * lane;
    if (off < pos)
      store_span(gbase, off, ringw[(off & (RING - 1)) >> 2], own_lo, pos);
    flushed += 256u;
    }
  }

// n == 0: undefined in the reference; defined as header + one full pad group
__global__ void k_fpc32_empty(uint8_t* out, size_t out_stride, uint32_t* sizes)
  {
  uint8_t* o = out + (size_t)blockIdx.x * out_stride;
  const uint8_t b[16] = { 0x25, 0, 0, 0, 0, 0x24, 0x92, 0x49, 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int i = 0; i < 16; ++i) o[i] = b[i];
  sizes[blockIdx.x] = 16;
  }

} // namespace

size_t fpc32_encode_workspace(uint32_t n, int arity, uint32_t* L_out, uint32_t* S_out)
  {
  const uint32_t target = 3840u / (uint32_t)arity;
  uint64_t L = ((uint64_t)n + target - 1) / target;
  L = (L + 63) / 64 * 64;
  if (L < 1024) L = 1024;
  const uint32_t S = (uint32_t)(((uint64_t)n + L - 1) / L);
  *L_out = (uint32_t)L;
  *S_out = S ? S : 1;
  const size_t rows = (size_t)(*S_out) * arity;
  const size_t nch = ((size_t)(*S_out) + CH - 1) / CH;
  size_t bytes = 0;
  bytes += rows * ROW * 4 * 2;                    // summ, inc
  bytes += nch * arity * TAB * 4 * 2;             // chval, chseen
  bytes += rows * 4 * 3;                          // segbytes, segoff, ucount
  bytes += rows * UCAP * sizeof(UEntry);          // ulist
  return bytes + 4096;
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
  uint32_t L, S;
  const size_t need = fpc32_encode_workspace(n, arity, &L, &S);
  if (need > ws_bytes)
    {
    set_error("fpc32 encode: workspace too small");
    return 0;
    }
  const size_t rows = (size_t)S * arity;
  const uint32_t nch = (S + CH - 1) / CH;
  uint8_t* w = d_ws;
  uint32_t* summ = (uint32_t*)w;      w += rows * ROW * 4;
  uint32_t* inc = (uint32_t*)w;       w += rows * ROW * 4;
  uint32_t* chval = (uint32_t*)w;     w += (size_t)nch * arity * TAB * 4;
  uint32_t* chseen = (uint32_t*)w;    w += (size_t)nch * arity * TAB * 4;
  uint32_t* segbytes = (uint32_t*)w;  w += rows * 4;
  uint32_t* segoff = (uint32_t*)w;    w += rows * 4;
  uint32_t* ucount = (uint32_t*)w;    w += rows * 4;
  w = (uint8_t*)(((uintptr_t)w + 15) & ~(uintptr_t)15);
  UEntry* ulist = (UEntry*)w;
  const uint32_t* src = (const uint32_t*)d_src;
  const unsigned threads = 64u * (unsigned)arity;
  hipLaunchKernelGGL(k_fpc32_pass_a, dim3(S), dim3(threads), (size_t)arity * LDSW_A * 4, st,
                     src, n, arity, L, S, summ, segbytes, ucount, ulist);
  const unsigned colblocks = ((unsigned)arity * TAB + 255u) / 256u;
  hipLaunchKernelGGL(k_fpc32_b1a, dim3(colblocks, nch), dim3(256), 0, st, summ, S, arity, chval, chseen);
  hipLaunchKernelGGL(k_fpc32_b1b, dim3(colblocks, nch), dim3(256), 0, st, summ, S, arity, chval, chseen, inc);
  hipLaunchKernelGGL(k_fpc32_b2, dim3((unsigned)rows), dim3(64), 0, st, inc, arity, S, ucount, ulist, segbytes);
  hipLaunchKernelGGL(k_fpc32_b3, dim3(arity), dim3(1024), 0, st, segbytes, S, segoff, d_sizes);
  hipLaunchKernelGGL(k_fpc32_pass_c, dim3(S), dim3(threads), (size_t)arity * LDSW_C * 4, st,
                     src, n, arity, L, S, inc, segoff, d_out, out_stride);
  return hip_ok(hipGetLastError(), "fpc32 encode kernels") ? 1 : 0;
  }

} // namespace trico
