// k_fpc64_sort.hip — throughput encoder for 64-bit floating-point streams: the table walk is handed to 1024 owners per table.
//
// Replaces trico_compress_double_precision(..., 20, 20) (fpsc.c:576-800) + the xyz / uv transposes for streams
// large enough to pay for it.  The predictors' tables have 2^20 entries each, far too many classes for the
// per-segment LDS tables of the float encoder, but the exactness argument is the same: the FCM hash
// of value i is the top 20 bits of v[i-1], the DFCM hash a function of the strides of v[i-1] and v[i-2], so every
// value's two hashes are known from the input alone, and a table read returns the payload (value / stride) of the
// latest earlier value with the same hash, or 0.
//
// Round 5 (the path every stream takes unless its hashes are badly skewed):
//   * runs: where value i has the hash of value i-1, its table read returns what i-1 wrote - the payload of i-1, known
//     from the input.  Only the FIRST value of a run of equal hashes reads the table and only the LAST one's write is
//     ever read, so a run is at most two table operations.  Smooth components (a grid's x and y, a constant normal)
//     are almost nothing but runs.
//   * owners: ten bits of the hash (FCM: its two halves xor-ed, DFCM: the lower half) name one of 1024 owners per
//     component, the upper half an entry of the owner's table (8 KiB: LDS).  k64_part_hist / k64_part_scatter write
//     every owner's operations, in value order, into its list (a stable one-digit partition: count matrix, scan,
//     ballots for the order inside a step - one launch per table for all components, whole vertices staged through
//     LDS); an operation is 16 bytes: value index, table entry, read / write flags, and the payload it writes.
//   * walk (k64_walk): one wave per owner streams through its list 64 operations at a time and applies them to its LDS
//     table in list order: the lanes of a step that share an entry find each other with ballots, a read takes the
//     payload of the nearest lower writer among them or the table's word, the last writer writes the table.  A read's
//     result goes to the operation's place in a second list.
//   * home (k64_home): the lists hold a tile's operations as 1024 short runs per component; one workgroup per tile and
//     component collects them, builds the tile's window of pred[] in LDS and writes it out whole.  (Storing the
//     results from the walk itself, one 8-byte store per operation anywhere in pred, and fetching the payloads from
//     the source the same way, cost 35-80 ps per operation on this machine - as much as the sort it replaces.)
//   * sizes (k64_sizes), scan, emit (k64_emit): code selection -> bytes per group of two values -> byte offsets ->
//     tiles of 512 groups packed in LDS and written with aligned dword stores; all components per launch, vertices
//     staged through LDS; a value inside a run takes its prediction from its neighbour instead of pred[].
//   * both tables are counted first (one host wait for the two longest lists); then the FCM table's walk runs beside
//     the DFCM table's scatter on a second stream, its way home beside the DFCM table's walk (both_tables).
// A stream whose longest list would make the walk slower than sorting (one owner with most of the operations and no
// runs: e.g. values alternating between two hashes) takes the path of round 2 for that table instead, which is
// indifferent to skew:
//   keys    both hashes of every value, and an index array
//   sort    (k_sort.hip: stable LSD radix sort, two 10-bit passes): (hash, index) pairs, one sort per table
//   preds   (k64_pred):   sorted neighbour with the same hash -> its payload, scattered back to the value's index
#include <atomic>
#include "common.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace trico {

namespace {

typedef uint64_t u64;

constexpr int TILE_G = 512;                  // groups per emit tile
constexpr int TILE_T = 256;                  // threads per emit workgroup (2 groups each)
constexpr int TILE_BYTES = TILE_G * 17;      // a group is at most 1 + 8 + 8 bytes

constexpr int OWN_BITS = 10;
constexpr uint32_t OWNERS = 1u << OWN_BITS;               // lists per component and table
constexpr uint32_t ENTRIES = 1u << (20 - OWN_BITS);       // table entries of an owner
constexpr u64 OP_READ = 1ull << 42, OP_WRITE = 1ull << 43; // an operation: value index | entry << 32 | flags

__device__ __forceinline__ u64 val_at(const u64* __restrict__ src, int64_t i, int arity, int c)
  {
  return i >= 0 ? src[(size_t)i * arity + c] : 0ull;          // values before the stream count as 0 (fpsc.c:596-600)
  }

// hash of value i in table T (0: FCM, fpsc.c:565-568 with a 20-bit table; 1: DFCM, fpsc.c:570-573, two strides of history)
template <int T>
__device__ __forceinline__ uint32_t key_from(u64 v1, u64 v2, u64 v3)
  {
  if (T == 0)
    return (uint32_t)(v1 >> 44);
  const u64 s1 = v1 - v2, s2 = v2 - v3;
  return (uint32_t)(((((s2 >> 44) & 1023ull) << 10) ^ (s1 >> 44)) & 0xfffffull);
  }
template <int T>
__device__ __forceinline__ uint32_t key_of(const u64* __restrict__ src, int64_t i, int arity, int c)
  {
  const u64 v1 = val_at(src, i - 1, arity, c), v2 = val_at(src, i - 2, arity, c);
  return key_from<T>(v1, v2, T ? val_at(src, i - 3, arity, c) : 0ull);
  }

// ---- the skew-proof path: keys, sort, predecessor in sorted order ------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(256) k64_keys(const u64* __restrict__ src, uint32_t n, int arity, int c, uint32_t* __restrict__ k)
  {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n)
    k[i] = key_of<T>(src, (int64_t)i, arity, c);
  }

// sorted position p holds value index vs[p] with hash ks[p]; its table read is the payload of the previous
// sorted element if that has the same hash (stable sort: the latest earlier value), else the zeroed table
__global__ void __launch_bounds__(256) k64_pred(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ vs, const u64* __restrict__ src,
                                                uint32_t n, int arity, int c, int table, u64* __restrict__ pred)
  {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n)
    return;
  u64 pay = 0;
  if (p > 0 && ks[p - 1] == ks[p])
    {
    const int64_t j = vs[p - 1];
    const u64 vj = val_at(src, j, arity, c);
    pay = table == 0 ? vj : vj - val_at(src, j - 1, arity, c);
    }
  pred[vs[p]] = pay;
  }

// ---- owners: count, place, walk -----------------------------------------------------------------------------------------------
// What value i does to table T of its component: nothing inside a run of equal hashes, a read at the run's first value, a write at
// its last.  op_of returns the operation word (0: none), the owner and what the value writes (FCM: itself, DFCM: its stride).
struct Op { u64 w, pay; };                   // (16 bytes, loaded and stored as one)

// The kernels that look at every value read the source through LDS: a thread needs v[i-4 .. i] of its component, 24 bytes apart
// in a vec3 stream - fetched directly, every such load touches as many cache lines as the wave has lanes.
// stage: buf[(k + 4) * arity + c] = v[i0 + k][c] for k in [-4, count), zeros before the stream (fpsc.c:596-600); coalesced
constexpr int HIST = 4;                      // values of history before the first one
__device__ __forceinline__ void stage_values(const u64* __restrict__ src, uint32_t n, int arity, uint32_t i0, uint32_t count, u64* buf,
                                             uint32_t tid, uint32_t nthreads)
  {
  const uint32_t total = (count + HIST) * (uint32_t)arity;
  const int64_t base = ((int64_t)i0 - HIST) * arity, end = (int64_t)n * arity;
  for (uint32_t k = tid; k < total; k += nthreads)
    {
    const int64_t idx = base + k;
    buf[k] = (idx >= 0 && idx < end) ? src[idx] : 0ull;
    }
  }

// the same for one wave and a step of 64 values, in two halves, so that the words of the next step are on their way while this one is
// looked at: chunk_fetch issues the loads (clamped to the stream, nothing is waited for), chunk_stash puts them into LDS
struct Chunk { u64 r[4]; };                  // (64 + HIST) * 3 = 204 words at most
__device__ __forceinline__ void chunk_fetch(const u64* __restrict__ src, uint32_t n, int arity, uint32_t ib, int lane, Chunk& ch)
  {
  const int64_t base = ((int64_t)ib - HIST) * arity, end = (int64_t)n * arity;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    {
    int64_t idx = base + lane + 64 * k;
    idx = idx < 0 ? 0 : (idx < end ? idx : end - 1);
    ch.r[k] = src[idx];
    }
  }
__device__ __forceinline__ void chunk_stash(uint32_t n, int arity, uint32_t ib, int lane, const Chunk& ch, u64* buf)
  {
  const int64_t base = ((int64_t)ib - HIST) * arity, end = (int64_t)n * arity;
  const int total = (64 + HIST) * arity;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    {
    const int w = lane + 64 * k;
    const int64_t idx = base + w;
    if (w < total)
      buf[w] = (idx >= 0 && idx < end) ? ch.r[k] : 0ull;
    }
  }

// p points at the staged v[i] of the component, the values before it are `arity` words apart
template <int T>
__device__ __forceinline__ u64 op_of(const u64* p, uint32_t i, uint32_t n, int arity, uint32_t& owner, u64& pay)
  {
  const u64 v0 = p[0], v1 = p[-arity], v2 = p[-2 * arity];
  const u64 v3 = T ? p[-3 * arity] : 0ull, v4 = T ? p[-4 * arity] : 0ull;
  const uint32_t k = key_from<T>(v1, v2, v3), kprev = key_from<T>(v2, v3, v4), knext = key_from<T>(v0, v1, v2);
  const bool first = i == 0u || k != kprev, last = i + 1u == n || knext != k;
  // FCM: sign and upper exponent bits folded onto the mantissa bits (quantised data often varies in one half only); the DFCM hash
  // is a mix of two strides already - folded again, a constant second difference puts most of a stream into a few lists
  owner = (T ? k : k ^ (k >> OWN_BITS)) & (OWNERS - 1u);
  pay = T ? v0 - v1 : v0;
  return (first || last) ? ((u64)i | ((u64)(k >> OWN_BITS) << 32) | (first ? OP_READ : 0ull) | (last ? OP_WRITE : 0ull)) : 0ull;
  }

// one wave per tile of `tile` values: cnt[c][owner] of its operations -> column `t` of hist[(c * OWNERS + owner)][t]
template <int T>
__global__ void __launch_bounds__(256) k64_part_hist(const u64* __restrict__ src, uint32_t n, int arity, uint32_t tile, uint32_t ntiles,
                                                     uint32_t* __restrict__ hist)
  {
  __shared__ uint32_t cnt[4][3][OWNERS];
  __shared__ u64 vals[4][(64 + HIST) * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t t = blockIdx.x * 4u + (uint32_t)wave;
  for (int c = 0; c < arity; ++c)
    for (uint32_t o = lane; o < OWNERS; o += 64u)
      cnt[wave][c][o] = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    hist[(size_t)arity * OWNERS * ntiles] = 0u;                 // the scan's last cell: the number of operations
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (t >= ntiles)
    return;
  const uint32_t i0 = t * tile, i1 = (n - i0 < tile) ? n : i0 + tile;
  Chunk ch;
  chunk_fetch(src, n, arity, i0, lane, ch);
  for (uint32_t ib = i0; ib < i1; ib += 64u)
    {
    const uint32_t i = ib + (uint32_t)lane;
    chunk_stash(n, arity, ib, lane, ch, vals[wave]);
    chunk_fetch(src, n, arity, ib + 64u, lane, ch);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int c = 0; c < arity; ++c)
      {
      uint32_t owner = 0;
      u64 pay;
      const u64 op = i < i1 ? op_of<T>(vals[wave] + (lane + HIST) * arity + c, i, n, arity, owner, pay) : 0ull;
      if (op)
        atomicAdd(&cnt[wave][c][owner], 1u);
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int c = 0; c < arity; ++c)
    for (uint32_t o = lane; o < OWNERS; o += 64u)
      hist[((size_t)c * OWNERS + o) * ntiles + t] = cnt[wave][c][o];
  }

// the same wave walks its tile again and writes every operation to its place in its owner's list: the list's next free position,
// kept in LDS, + the number of lower lanes with an operation for the same owner (10 ballots); steps, tiles, owners are visited in
// order, so a list holds its operations in value order
template <int T>
__global__ void __launch_bounds__(256) k64_part_scatter(const u64* __restrict__ src, uint32_t n, int arity, uint32_t tile, uint32_t ntiles,
                                                        const uint32_t* __restrict__ offs, Op* __restrict__ ops)
  {
  __shared__ uint32_t pos[4][3][OWNERS];
  __shared__ u64 vals[4][(64 + HIST) * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t t = blockIdx.x * 4u + (uint32_t)wave;
  if (t >= ntiles)
    return;
  for (int c = 0; c < arity; ++c)
    for (uint32_t o = lane; o < OWNERS; o += 64u)
      pos[wave][c][o] = offs[((size_t)c * OWNERS + o) * ntiles + t];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint32_t i0 = t * tile, i1 = (n - i0 < tile) ? n : i0 + tile;
  const uint64_t below = (1ull << lane) - 1ull;
  Chunk ch;
  chunk_fetch(src, n, arity, i0, lane, ch);
  for (uint32_t ib = i0; ib < i1; ib += 64u)
    {
    const uint32_t i = ib + (uint32_t)lane;
    chunk_stash(n, arity, ib, lane, ch, vals[wave]);
    chunk_fetch(src, n, arity, ib + 64u, lane, ch);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int c = 0; c < arity; ++c)
      {
      uint32_t owner = 0;
      u64 pay = 0ull;
      const u64 op = i < i1 ? op_of<T>(vals[wave] + (lane + HIST) * arity + c, i, n, arity, owner, pay) : 0ull;
      uint64_t same = __ballot(op != 0ull);
      if (same == 0ull)
        continue;                                              // (a step inside runs: nothing to place)
#pragma unroll
      for (int b = 0; b < OWN_BITS; ++b)
        {
        const bool bit = (owner >> b) & 1u;
        const uint64_t m = __ballot(bit);
        same &= bit ? m : ~m;
        }
      const uint32_t base = pos[wave][c][owner];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (op)
        {
        ops[base + (uint32_t)__popcll(same & below)] = Op{ op, pay };
        if ((same >> lane) == 1ull)                            // highest lane of this owner
          pos[wave][c][owner] = base + (uint32_t)__popcll(same);
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
  }

// longest list and the number of operations -> res[0], res[1]
__global__ void __launch_bounds__(256) k64_list_max(const uint32_t* __restrict__ offs, uint32_t lists, uint32_t ntiles, uint32_t* __restrict__ res)
  {
  const uint32_t l = blockIdx.x * 256u + threadIdx.x;
  if (l < lists)
    atomicMax(&res[0], offs[(size_t)(l + 1u) * ntiles] - offs[(size_t)l * ntiles]);
  if (l == 0u)
    res[1] = offs[(size_t)lists * ntiles];
  }

// One wave per list, four steps of 64 operations per round; the operations of the next two rounds are in flight while a round is
// applied.  The loop body is three rounds, so that the registers of the rounds in flight change roles without being moved, and
// every fetch is unconditional (clamped to the list), so that nothing is waited for before it is used.
// A step is applied without a loop: the lanes with the same entry find each other with 10 ballots; a read returns the payload of
// the nearest lower lane that writes the same entry, or the table's word; the highest writing lane of an entry writes the table.
// The results go into a second list, in the same places.
template <int T>
__global__ void __launch_bounds__(64) k64_walk(const Op* __restrict__ ops, Op* __restrict__ done, const uint32_t* __restrict__ offs, uint32_t ntiles, Op* __restrict__ sink)
  {
  __shared__ u64 tab[ENTRIES];
  const uint32_t lane = threadIdx.x, l = blockIdx.x;
  const uint32_t b = offs[(size_t)l * ntiles], e = offs[(size_t)(l + 1u) * ntiles];
  if (b == e)
    return;
  for (uint32_t k = lane; k < ENTRIES; k += 64u)
    tab[k] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint64_t below = (1ull << lane) - 1ull;
  const uint32_t last = e - 1u;
  auto fetch = [&](uint32_t p, Op (&op)[4])
    {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      {
      const uint32_t q = p + 64u * k + lane;
      op[k] = ops[q < last ? q : last];
      }
    };
  auto apply = [&](uint32_t p, const Op (&ops4)[4])
    {
    // who meets whom (vector compares and ballots only) ...
    uint32_t ent[4];
    uint64_t prior[4];
    bool lastw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      {
      const u64 op = p + 64u * k + lane < e ? ops4[k].w : 0ull;
      ent[k] = (uint32_t)(op >> 32) & (ENTRIES - 1u);
      const bool rd = (op & OP_READ) != 0ull, wr = (op & OP_WRITE) != 0ull;
      uint64_t same = __ballot(rd || wr);
#pragma unroll
      for (int bit = 0; bit < 20 - OWN_BITS; ++bit)
        {
        const bool s1 = (ent[k] >> bit) & 1u;
        const uint64_t m = __ballot(s1);
        same &= s1 ? m : ~m;
        }
      const uint64_t writers = same & __ballot(wr);
      prior[k] = writers & below;
      lastw[k] = wr && (writers >> lane) == 1ull;              // no higher lane writes this entry
      }
    // ... the table, step by step (the LDS unit keeps a wave's accesses in order: nothing is waited for in between) ...
    u64 word[4];
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      {
      const u64 pay = ops4[k].pay;
      word[k] = tab[ent[k]];
      const int from = prior[k] ? 63 - __builtin_clzll(prior[k]) : (int)lane;
      lo[k] = (uint32_t)__shfl((int)(uint32_t)pay, from);
      hi[k] = (uint32_t)__shfl((int)(uint32_t)(pay >> 32), from);
      if (lastw[k])
        tab[ent[k]] = pay;
      }
    // ... the results, into a list of their own (stores into the list being read would have to wait for its fetches).  Every lane
    // stores, the ones past the list's end into a word of the list's own: a round is straight-line code and the compiler's count
    // of the memory operations in flight stays exact.
#pragma unroll
    for (int k = 0; k < 4; ++k)
      {
      const uint32_t q = p + 64u * k + lane;
      Op* dst = q < e ? done + q : sink + 8u * l;
      *dst = Op{ ops4[k].w, prior[k] ? ((u64)hi[k] << 32 | lo[k]) : word[k] };
      }
    };
  // (the stores of zeros make the queue of memory operations look at the loop's entry as it does at its end - two fetches with four
  // stores after each - so that the wait the compiler puts at the loop's head leaves the younger fetch in flight there too)
  Op oa[4], ob[4], oc[4];
  fetch(b, oa);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    sink[8u * l + k] = Op{ 0ull, 0ull };
  fetch(b + 256u, ob);
#pragma unroll
  for (int k = 4; k < 8; ++k)
    sink[8u * l + k] = Op{ 0ull, 0ull };
#pragma unroll 1
  for (uint32_t p = b;; p += 768u)
    {
    fetch(p + 512u, oc);
    apply(p, oa);
    if (p + 256u >= e)
      break;
    fetch(p + 768u, oa);
    apply(p + 256u, ob);
    if (p + 512u >= e)
      break;
    fetch(p + 1024u, ob);
    apply(p + 512u, oc);
    if (p + 768u >= e)
      break;
    }
  }

// One workgroup per tile and component: the tile's operations are a run in every list; a quarter wave per run puts the results of the
// reads into the tile's window of pred, built in LDS and written out whole (the words of values that read nothing are never looked at:
// code_of takes those predictions from the neighbours).  A window without any read is not written.
constexpr uint32_t HOME_TILE_MAX = 16384;                     // values per tile: 128 KiB of LDS
__global__ void __launch_bounds__(1024) k64_home(const Op* __restrict__ ops, const uint32_t* __restrict__ offs, uint32_t ntiles, uint32_t tile,
                                                 uint32_t n, u64* __restrict__ pred)
  {
  extern __shared__ __attribute__((aligned(16))) u64 win[];
  __shared__ uint32_t run[OWNERS][2];                          // where the tile's run begins and ends in every list
  const uint32_t t = blockIdx.x, c = blockIdx.y, sub = threadIdx.x >> 4, l16 = threadIdx.x & 15u;
  const uint32_t i0 = t * tile, cnt = n - i0 < tile ? n - i0 : tile;
    {
    const size_t cell = ((size_t)c * OWNERS + threadIdx.x) * ntiles + t;
    run[threadIdx.x][0] = offs[cell];
    run[threadIdx.x][1] = offs[cell + 1];
    }
  __syncthreads();
  int found = 0;
  // a quarter wave per run, four runs in flight
  for (uint32_t o = sub; o < OWNERS; o += 256u)
    {
    uint32_t b[4], e[4];
    Op op[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      {
      b[k] = run[o + 64u * k][0] + l16;
      e[k] = run[o + 64u * k][1];
      if (b[k] < e[k])
        op[k] = ops[b[k]];
      }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (uint32_t q = b[k]; q < e[k]; q += 16u)
        {
        const Op x = q == b[k] ? op[k] : ops[q];
        if (x.w & OP_READ)
          {
          win[(uint32_t)x.w - i0] = x.pay;
          found = 1;
          }
        }
    }
  if (!__syncthreads_or(found))
    return;
  u64* __restrict__ out = pred + (size_t)c * n + i0;
  for (uint32_t k = threadIdx.x; k < cnt; k += 1024u)
    out[k] = win[k];
  }

__device__ __forceinline__ uint32_t blen64(u64 x) { return x ? (uint32_t)(71 - __builtin_clzll(x)) >> 3 : 0u; }

// code, residual and residual length of value i (fpsc.c:640-700): FCM codes 0..8, DFCM codes 9..15 (1..7 bytes).
// p points at the staged v[i]; pred1 / pred2 hold the table reads of the values that begin a run of equal hashes; inside a run
// the read returns what the value before wrote: v[i-1], and the stride v[i-1] - v[i-2].
__device__ __forceinline__ uint32_t code_of(const u64* p, int arity, uint32_t i, const u64* __restrict__ pred1, const u64* __restrict__ pred2,
                                            u64& x, uint32_t& len)
  {
  const u64 v = p[0], a = p[-arity], b = p[-2 * arity], b3 = p[-3 * arity], b4 = p[-4 * arity];
  const bool first1 = i == 0u || key_from<0>(a, b, 0ull) != key_from<0>(b, b3, 0ull);
  const bool first2 = i == 0u || key_from<1>(a, b, b3) != key_from<1>(b, b3, b4);
  const u64 p1 = first1 ? pred1[i] : a, p2 = first2 ? pred2[i] : a - b;
  const u64 x1 = v ^ p1, x2 = v ^ (a + p2);
  const uint32_t n1 = blen64(x1);
  uint32_t n2 = blen64(x2);
  n2 = n2 ? n2 : 1u;
  if (n1 > 1u && n2 < n1)
    {
    x = x2;
    len = n2;
    return 8u + n2;
    }
  x = x1;
  len = n1;
  return n1;
  }

// 256 groups of two values per workgroup, all components; pred1 / pred2 / gsz / goff of component c at c * n / c * gstride
__global__ void __launch_bounds__(256) k64_sizes(const u64* __restrict__ src, uint32_t n, int arity, const u64* __restrict__ pred1,
                                                 const u64* __restrict__ pred2, uint32_t ngroups, uint32_t* __restrict__ gsz, size_t gstride)
  {
  __shared__ u64 vals[(512 + HIST) * 3];
  const uint32_t g0 = blockIdx.x * 256u, g = g0 + threadIdx.x;
  stage_values(src, n, arity, 2u * g0, 512u, vals, threadIdx.x, 256u);
  __syncthreads();
  if (g >= ngroups)
    return;
  for (int c = 0; c < arity; ++c)
    {
    const u64* p = vals + (2u * threadIdx.x + HIST) * arity + c;
    u64 x;
    uint32_t l0, l1 = 1u;                                      // a missing second value is padded with code 1, one 0x00 byte
    code_of(p, arity, 2u * g, pred1 + (size_t)c * n, pred2 + (size_t)c * n, x, l0);
    if (2u * g + 1u < n)
      code_of(p + arity, arity, 2u * g + 1u, pred1 + (size_t)c * n, pred2 + (size_t)c * n, x, l1);
    gsz[(size_t)c * gstride + g] = 1u + l0 + l1;
    }
  }

// TILE_G groups per workgroup, the components one after the other
__global__ void __launch_bounds__(TILE_T) k64_emit(const u64* __restrict__ src, uint32_t n, int arity, const u64* __restrict__ pred1,
                                                   const u64* __restrict__ pred2, uint32_t ngroups, const uint32_t* __restrict__ goff,
                                                   const uint32_t* __restrict__ gsz, size_t gstride, uint8_t* __restrict__ out, size_t out_stride,
                                                   uint32_t* __restrict__ size_out)
  {
  __shared__ __attribute__((aligned(16))) uint8_t lds[TILE_BYTES + 16];
  __shared__ u64 vals[(2 * TILE_G + HIST) * 3];
  const uint32_t g0 = blockIdx.x * TILE_G;
  const uint32_t g1 = g0 + TILE_G < ngroups ? g0 + TILE_G : ngroups;
  stage_values(src, n, arity, 2u * g0, 2u * TILE_G, vals, threadIdx.x, TILE_T);
  for (int c = 0; c < arity; ++c)
    {
    __syncthreads();                                           // (the values are staged; the bytes of the component before are out)
    const u64* p1 = pred1 + (size_t)c * n;
    const u64* p2 = pred2 + (size_t)c * n;
    const uint32_t* go = goff + (size_t)c * gstride;
    const uint32_t* gs = gsz + (size_t)c * gstride;
    uint8_t* oc = out + (size_t)c * out_stride;
    const uint32_t base = go[g0];
    const uint32_t end = go[g1 - 1] + gs[g1 - 1];
    for (uint32_t g = g0 + threadIdx.x; g < g1; g += TILE_T)
      {
      uint8_t* o = lds + (go[g] - base);
      const u64* p = vals + (2u * (g - g0) + HIST) * arity + c;
      u64 x0, x1 = 0;
      uint32_t l0, l1 = 1u, c1 = 1u;
      const uint32_t c0 = code_of(p, arity, 2u * g, p1, p2, x0, l0);
      if (2u * g + 1u < n)
        c1 = code_of(p + arity, arity, 2u * g + 1u, p1, p2, x1, l1);
      *o++ = (uint8_t)((c1 << 4) | c0);                         // fpsc.c:706
      for (uint32_t k = l0; k > 0; --k) *o++ = (uint8_t)(x0 >> (8u * (k - 1u)));
      for (uint32_t k = l1; k > 0; --k) *o++ = (uint8_t)(x1 >> (8u * (k - 1u)));
      }
    __syncthreads();
    // tile bytes [base, end) of the group area go to out + 5 + base: bytes up to the first aligned dword, then dwords
    uint8_t* d = oc + 5u + base;
    const uint32_t len = end - base;
    const uint32_t head = (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u);
    const uint32_t h = head < len ? head : len;
    if (threadIdx.x < h)
      d[threadIdx.x] = lds[threadIdx.x];
    const uint32_t body = (len - h) >> 2;
    for (uint32_t t = threadIdx.x; t < body; t += TILE_T)
      {
      const uint8_t* s = lds + h + 4u * t;
      ((uint32_t*)(d + h))[t] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
      }
    const uint32_t done = h + 4u * body;
    if (threadIdx.x < len - done)
      d[done + threadIdx.x] = lds[done + threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0)
      {
      oc[0] = 0xaa;                                              // (20/2) << 4 | (20/2), fpsc.c:610
      oc[1] = (uint8_t)(n >> 24); oc[2] = (uint8_t)(n >> 16); oc[3] = (uint8_t)(n >> 8); oc[4] = (uint8_t)n;
      }
    if (g1 == ngroups && threadIdx.x == 0)
      size_out[c] = 5u + end;
    }
  }

// values per wave of the partition kernels: ~4096 tiles, 1024 .. 16384 values each
uint32_t part_tile(uint32_t n)
  {
  const uint32_t t = ((n + 4095u) / 4096u + 63u) & ~63u;
  return t < 1024u ? 1024u : (t > HOME_TILE_MAX ? HOME_TILE_MAX : t);
  }

struct EncPlan { size_t pred1, pred2, gsz, goff, gstride, hist[2], res, sink, area[3], tmp, tmp_bytes, total; uint32_t tile, ntiles; };

EncPlan plan_for(uint32_t n, int arity)
  {
  EncPlan p;
  const size_t ng = ((size_t)n + 1) / 2;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes + 16, 256); return at; };
  p.tile = part_tile(n);
  p.ntiles = (n + p.tile - 1) / p.tile;
  p.pred1 = take(8 * (size_t)n * arity); p.pred2 = take(8 * (size_t)n * arity);
  p.gstride = align_up(ng + 4, 64);
  p.gsz = take(4 * p.gstride * arity); p.goff = take(4 * p.gstride * arity);
  const size_t cells = (size_t)arity * OWNERS * p.ntiles + 1;
  p.hist[0] = take(4 * cells);
  p.hist[1] = take(4 * cells);
  p.res = take(64);
  p.sink = take(2 * 8 * sizeof(Op) * (size_t)arity * OWNERS);
  // three areas of one table's operations each: FCM operations / FCM results / DFCM operations, DFCM results where the FCM operations
  // were; a table that is sorted keeps keys, sorted keys and sorted indices of one component in the first
  for (int i = 0; i < 3; ++i)
    p.area[i] = take(sizeof(Op) * (size_t)n * arity);
  const size_t sort_bytes = sort_workspace(n), scan_bytes = scan_workspace((uint32_t)(cells > ng ? cells : ng));
  p.tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  p.tmp = take(p.tmp_bytes);
  p.total = o;
  return p;
  }

// operations in one list from which the table takes the sorting path: the walk of such a list alone (64 operations per step,
// a step ~1 us with its fetches in flight) would take longer than the sorts of the whole stream
uint32_t walk_limit(uint32_t n, int arity)
  {
  if (const char* e = getenv("TRICO_FPC64_WALK_MAX"))              // tuning knob (0: always sort); changes time, never bytes
    return (uint32_t)strtoul(e, nullptr, 10);
  const uint64_t even = 8ull * (uint64_t)n / OWNERS;             // eight times an even share of a stream without runs
  return even > 262144ull ? (uint32_t)(even > 0xffffffffull ? 0xffffffffull : even) : 262144u;
  }

// operations per list of table T: count matrix, its scan, the longest list -> res[0] (and the number of operations -> res[1])
template <int T>
int table_count(const u64* src, uint32_t n, int arity, const EncPlan& p, uint32_t* hist, uint32_t* res, uint8_t* tmp, size_t tmp_bytes, hipStream_t st)
  {
  const uint32_t lists = (uint32_t)arity * OWNERS;
  const size_t cells = (size_t)lists * p.ntiles + 1;
  const unsigned blocks = (p.ntiles + 3u) / 4u;
  hipLaunchKernelGGL(k64_part_hist<T>, dim3(blocks), dim3(256), 0, st, src, n, arity, p.tile, p.ntiles, hist);
  if (!exclusive_scan_u32(hist, hist, (uint32_t)cells, tmp, tmp_bytes))
    return 0;
  hipLaunchKernelGGL(k64_list_max, dim3((lists + 255u) / 256u), dim3(256), 0, st, hist, lists, p.ntiles, res);
  return 1;
  }

// k64_home takes a tile's window of predictions into 128 KiB of dynamic LDS: the attribute belongs to the (function, device) pair, so it
// is claimed once per device of the process, not once per process
static bool home_lds_claimed()
  {
  static std::atomic<int> state[16];             // 0 not asked, 1 claimed, 2 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16)
    return hipFuncSetAttribute((const void*)k64_home, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(HOME_TILE_MAX * 8u)) == hipSuccess;
  int s = state[dev].load();
  if (s == 0)
    {
    s = hipFuncSetAttribute((const void*)k64_home, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(HOME_TILE_MAX * 8u)) == hipSuccess ? 1 : 2;
    state[dev].store(s);
    }
  return s == 1;
  }

// the lists of table T written, walked, the results taken home to pred.  Events (each may be null): recorded behind the scatter and
// behind the walk; waited for in front of the walk and in front of the way home.
struct Marks { hipEvent_t scattered, walked, before_walk, before_home; };
template <int T>
int table_walk(const u64* src, uint32_t n, int arity, const EncPlan& p, const uint32_t* hist, Op* ops, Op* done, Op* sink, u64* pred,
               hipStream_t st, const Marks& m)
  {
  if (!home_lds_claimed())
    {
    set_error("fpc64 throughput encoder: cannot claim 128 KiB of LDS");
    return 0;
    }
  const uint32_t lists = (uint32_t)arity * OWNERS;
  const unsigned blocks = (p.ntiles + 3u) / 4u;
  hipLaunchKernelGGL(k64_part_scatter<T>, dim3(blocks), dim3(256), 0, st, src, n, arity, p.tile, p.ntiles, hist, ops);
  if (m.scattered && !hip_ok(hipEventRecord(m.scattered, st), "hipEventRecord"))
    return 0;
  if (m.before_walk && !hip_ok(hipStreamWaitEvent(st, m.before_walk, 0), "hipStreamWaitEvent"))
    return 0;
  hipLaunchKernelGGL(k64_walk<T>, dim3(lists), dim3(64), 0, st, ops, done, hist, p.ntiles, sink);
  if (m.walked && !hip_ok(hipEventRecord(m.walked, st), "hipEventRecord"))
    return 0;
  if (m.before_home && !hip_ok(hipStreamWaitEvent(st, m.before_home, 0), "hipStreamWaitEvent"))
    return 0;
  hipLaunchKernelGGL(k64_home, dim3(p.ntiles, arity), dim3(1024), p.tile * 8u, st, done, hist, p.ntiles, p.tile, n, pred);
  return 1;
  }

// the path of round 2 for table T (current_stream()): sorted (hash, index) pairs, predecessor = table read
template <int T>
int table_sort(const u64* src, uint32_t n, int arity, const EncPlan& p, uint8_t* d_ws, u64* pred)
  {
  hipStream_t st = current_stream();
  const size_t q = align_up(4 * (size_t)n + 16, 256);
  uint32_t* k = (uint32_t*)(d_ws + p.area[0]); uint32_t* ks = (uint32_t*)(d_ws + p.area[0] + q); uint32_t* vs = (uint32_t*)(d_ws + p.area[0] + 2 * q);
  const unsigned vb = (n + 255u) / 256u;
  for (int c = 0; c < arity; ++c)
    {
    hipLaunchKernelGGL(k64_keys<T>, dim3(vb), dim3(256), 0, st, src, n, arity, c, k);
    // values = identity: the sorted value of position p is the index of the p-th value in (hash, index) order
    if (!radix_sort_pairs(k, nullptr, ks, vs, n, 20, d_ws + p.tmp, p.tmp_bytes))
      return 0;
    hipLaunchKernelGGL(k64_pred, dim3(vb), dim3(256), 0, st, ks, vs, src, n, arity, c, T, pred + (size_t)c * n);
    }
  return 1;
  }

// A second stream per host thread.  What a table costs is four kernels of different kinds: the count and the scatter are bound by
// memory and instruction issue, the walk is 3072 chains of rounds with the machine mostly waiting, the way home wants the LDS.  With
// both tables walked, the FCM table's walk runs beside the DFCM table's scatter, and its way home beside the DFCM table's walk.
struct Side
  {
  hipStream_t st = nullptr;
  hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
  int device = -1;
  bool ok = false, tried = false;
  bool ready()
    {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess)
      return false;
    if (tried)
      return ok && dev == device;
    tried = true;
    device = dev;
    ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 4; ++i)
      ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok)
      (void)hipGetLastError();
    return ok;
    }
  };

int both_tables(const u64* src, uint32_t n, int arity, const EncPlan& p, uint8_t* d_ws, u64* pred1, u64* pred2)
  {
  hipStream_t A = current_stream();
  uint32_t* hist0 = (uint32_t*)(d_ws + p.hist[0]); uint32_t* hist1 = (uint32_t*)(d_ws + p.hist[1]);
  uint32_t* res = (uint32_t*)(d_ws + p.res);
  Op* X = (Op*)(d_ws + p.area[0]); Op* Y = (Op*)(d_ws + p.area[1]); Op* Z = (Op*)(d_ws + p.area[2]);
  Op* sink0 = (Op*)(d_ws + p.sink); Op* sink1 = sink0 + 8u * arity * OWNERS;
  bool walk0 = false, walk1 = false;
  if ((uint64_t)n * (uint64_t)arity < 0xffffffffull && walk_limit(n, arity) != 0u)      // (list positions are 32 bits)
    {
    // both tables counted, then the host waits once for the two longest lists
    uint32_t h[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (!hip_ok(hipMemsetAsync(res, 0, 32, A), "fpc64 encoder: memset") ||
        !table_count<0>(src, n, arity, p, hist0, res, d_ws + p.tmp, p.tmp_bytes, A) ||
        !table_count<1>(src, n, arity, p, hist1, res + 4, d_ws + p.tmp, p.tmp_bytes, A) ||
        !hip_ok(hipMemcpyAsync(h, res, 32, hipMemcpyDeviceToHost, A), "fpc64 encoder: list lengths") ||
        !hip_ok(hipStreamSynchronize(A), "fpc64 encoder: list lengths"))
      return 0;
    walk0 = h[0] <= walk_limit(n, arity);
    walk1 = h[4] <= walk_limit(n, arity);
    if (getenv("TRICO_HIP_DEBUG"))
      fprintf(stderr, "trico_hip: double encoder, %u x %d values: FCM table %u operations, longest of %u lists %u (an even share: %u); DFCM table %u, longest %u (%u)\n",
              n, arity, h[1], (unsigned)arity * OWNERS, h[0], h[1] / ((unsigned)arity * OWNERS), h[5], h[4], h[5] / ((unsigned)arity * OWNERS));
    }
  static thread_local Side side;
  const Marks none = { nullptr, nullptr, nullptr, nullptr };
  if (!(walk0 && walk1 && side.ready()))
    {
    // one after the other on this stream (the DFCM table's results go where the FCM table's operations were)
    if (!(walk0 ? table_walk<0>(src, n, arity, p, hist0, X, Y, sink0, pred1, A, none) : table_sort<0>(src, n, arity, p, d_ws, pred1)))
      return 0;
    return walk1 ? table_walk<1>(src, n, arity, p, hist1, Z, X, sink1, pred2, A, none) : table_sort<1>(src, n, arity, p, d_ws, pred2);
    }
  hipStream_t B = side.st;
  hipEvent_t scattered0 = side.ev[0], walked0 = side.ev[1], scattered1 = side.ev[2], home1 = side.ev[3];
  // this stream: scatter 0, walk 0, (scatter 1 done) home 0; the side stream: (scatter 0 done) scatter 1, (walk 0 done) walk 1, home 1
  // (the launches are issued in the order the device should start them)
  int ok = 1;
  const uint32_t lists = (uint32_t)arity * OWNERS;
  const unsigned blocks = (p.ntiles + 3u) / 4u;
  if (!home_lds_claimed())
    {
    set_error("fpc64 throughput encoder: cannot claim 128 KiB of LDS");
    return 0;
    }
  hipLaunchKernelGGL(k64_part_scatter<0>, dim3(blocks), dim3(256), 0, A, src, n, arity, p.tile, p.ntiles, hist0, X);
  ok = ok && hip_ok(hipEventRecord(scattered0, A), "hipEventRecord") && hip_ok(hipStreamWaitEvent(B, scattered0, 0), "hipStreamWaitEvent");
  hipLaunchKernelGGL(k64_walk<0>, dim3(lists), dim3(64), 0, A, X, Y, hist0, p.ntiles, sink0);
  ok = ok && hip_ok(hipEventRecord(walked0, A), "hipEventRecord");
  hipLaunchKernelGGL(k64_part_scatter<1>, dim3(blocks), dim3(256), 0, B, src, n, arity, p.tile, p.ntiles, hist1, Z);
  ok = ok && hip_ok(hipEventRecord(scattered1, B), "hipEventRecord") && hip_ok(hipStreamWaitEvent(A, scattered1, 0), "hipStreamWaitEvent");
  hipLaunchKernelGGL(k64_home, dim3(p.ntiles, arity), dim3(1024), p.tile * 8u, A, Y, hist0, p.ntiles, p.tile, n, pred1);
  ok = ok && hip_ok(hipStreamWaitEvent(B, walked0, 0), "hipStreamWaitEvent");      // (its results go where the FCM operations were)
  hipLaunchKernelGGL(k64_walk<1>, dim3(lists), dim3(64), 0, B, Z, X, hist1, p.ntiles, sink1);
  hipLaunchKernelGGL(k64_home, dim3(p.ntiles, arity), dim3(1024), p.tile * 8u, B, X, hist1, p.ntiles, p.tile, n, pred2);
  ok = ok && hip_ok(hipEventRecord(home1, B), "hipEventRecord") && hip_ok(hipStreamWaitEvent(A, home1, 0), "hipStreamWaitEvent");
  if (!ok)
    (void)hipStreamSynchronize(B);
  return ok;
  }

} // namespace

uint32_t fpc64_sorted_threshold()
  {
  static uint32_t t = 0;
  if (!t)
    {
    const char* e = getenv("TRICO_FPC64_SORT_MIN");              // tuning knob: values per stream from which this encoder is used
    t = e ? (uint32_t)strtoul(e, nullptr, 10) : 8192u;            // (measured, vec3 streams: 0.27 ms against 0.29 at 8192 values, 0.28 against 1.07 at 32768)
    if (t < 2) t = 2;
    }
  return t;
  }

size_t fpc64_sorted_workspace(uint32_t n, int arity)
  {
  return plan_for(n, arity).total;
  }

int launch_fpc64_encode_sorted(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                               uint8_t* d_ws, size_t ws_bytes)
  {
  if (n < 2 || n > 0x7fffffffu || arity < 1 || arity > 3)
    {
    set_error("fpc64 throughput encoder: unsupported count");
    return 0;
    }
  const EncPlan p = plan_for(n, arity);
  if (p.total > ws_bytes)
    {
    set_error("fpc64 throughput encoder: workspace too small");
    return 0;
    }
  hipStream_t st = current_stream();
  const u64* src = (const u64*)d_src;
  const uint32_t ng = (n + 1u) / 2u;
  const unsigned gb = (ng + 255u) / 256u, tiles = (ng + TILE_G - 1) / TILE_G;
  u64* pred1 = (u64*)(d_ws + p.pred1); u64* pred2 = (u64*)(d_ws + p.pred2);
  uint32_t* gsz = (uint32_t*)(d_ws + p.gsz); uint32_t* goff = (uint32_t*)(d_ws + p.goff);
  if (!both_tables(src, n, arity, p, d_ws, pred1, pred2))
    return 0;
  hipLaunchKernelGGL(k64_sizes, dim3(gb), dim3(256), 0, st, src, n, arity, pred1, pred2, ng, gsz, p.gstride);
  for (int c = 0; c < arity; ++c)
    if (!exclusive_scan_u32(gsz + (size_t)c * p.gstride, goff + (size_t)c * p.gstride, ng, d_ws + p.tmp, p.tmp_bytes))
      return 0;
  hipLaunchKernelGGL(k64_emit, dim3(tiles), dim3(TILE_T), 0, st, src, n, arity, pred1, pred2, ng, goff, gsz, p.gstride,
                     d_out, out_stride, d_sizes);
  return hip_ok(hipGetLastError(), "fpc64 throughput encoder") ? 1 : 0;
  }

} // namespace trico
