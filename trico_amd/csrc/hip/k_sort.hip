// k_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs and an exclusive u32 scan, hand-written for gfx950.
//
// Used by the sort-based double encoder (k_fpc64_sort.hip: "latest earlier value with the same 20-bit hash" = predecessor in a
// stable sort by hash) and by the STL vertex weld (k_weld.hip).  No library underneath.
//
// One pass sorts by B key bits (B = 10 for 20-bit keys: two passes; B = 8 for 32-bit keys: four):
//   hist     one wave per tile of TILE keys counts its keys per digit in LDS (ds_add) and writes column `tile` of the
//            digit-major count matrix hist[digit][tile]
//   scan     exclusive scan of that matrix in memory order = first output position of every (digit, tile)
//   scatter  the same wave walks its tile again, 64 keys per step: the lanes with equal digit find each other with B
//            ballots (match-any), a key's position is the digit's running offset + the number of lower lanes with the same
//            digit, and the highest lane of each digit bumps the offset.  Steps, tiles and digits are visited in order, so
//            equal keys keep their input order.
// Traffic per pass: keys read twice, pairs read and written once.  The scan is the usual three-level
// block-scan / scan-of-block-sums / add.
#include "common.hpp"

namespace trico {

namespace {

constexpr uint32_t SORT_TILE = 16384;            // keys per wave (256 steps of 64)
constexpr uint32_t SCAN_BLOCK = 2048;            // elements per 256-thread scan block

template <int B>
__global__ void __launch_bounds__(256) k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t ntiles,
                                                    uint32_t* __restrict__ hist)
  {
  __shared__ uint32_t cnt[4][1 << B];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x * 4u + (uint32_t)wave;
  for (int i = lane; i < (1 << B); i += 64)
    cnt[wave][i] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (tile >= ntiles)
    return;
  const uint32_t i0 = tile * SORT_TILE;
  const uint32_t i1 = (n - i0 < SORT_TILE) ? n : i0 + SORT_TILE;
  for (uint32_t i = i0 + (uint32_t)lane; i < i1; i += 64u)
    atomicAdd(&cnt[wave][(keys[i] >> shift) & ((1u << B) - 1u)], 1u);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int d = lane; d < (1 << B); d += 64)
    hist[(size_t)d * ntiles + tile] = cnt[wave][d];
  }

template <int B>
__global__ void __launch_bounds__(256) k_radix_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, int shift,
                                                       uint32_t ntiles, const uint32_t* __restrict__ offs, uint32_t* __restrict__ keys_out,
                                                       uint32_t* __restrict__ vals_out)
  {
  __shared__ uint32_t pos[4][1 << B];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x * 4u + (uint32_t)wave;
  if (tile >= ntiles)
    return;
  for (int d = lane; d < (1 << B); d += 64)
    pos[wave][d] = offs[(size_t)d * ntiles + tile];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint32_t i0 = tile * SORT_TILE;
  const uint32_t i1 = (n - i0 < SORT_TILE) ? n : i0 + SORT_TILE;
  const uint64_t below = (1ull << lane) - 1ull;
  for (uint32_t ib = i0; ib < i1; ib += 64u)
    {
    const uint32_t i = ib + (uint32_t)lane;
    const bool act = i < i1;
    const uint32_t k = act ? keys[i] : 0u;
    const uint32_t v = act ? (vals ? vals[i] : i) : 0u;
    const uint32_t d = (k >> shift) & ((1u << B) - 1u);
    uint64_t same = __ballot(act);                     // lanes with my digit
#pragma unroll
    for (int b = 0; b < B; ++b)
      {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      same &= bit ? m : ~m;
      }
    const uint32_t base = pos[wave][d];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (act)
      {
      const uint32_t o = base + (uint32_t)__popcll(same & below);
      keys_out[o] = k;
      vals_out[o] = v;
      if ((same >> lane) == 1ull)                      // highest lane of this digit
        pos[wave][d] = base + (uint32_t)__popcll(same);
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }

// ---- exclusive scan ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
  {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    {
    const uint32_t t = (uint32_t)__shfl_up((int)v, o);
    if (lane >= o) v += t;
    }
  return v;
  }

// block of 256 threads scans SCAN_BLOCK elements (8 consecutive per thread); totals[block] = block sum
__global__ void __launch_bounds__(256) k_scan_blocks(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n,
                                                     uint32_t* __restrict__ totals)
  {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t e0 = blockIdx.x * SCAN_BLOCK + 8u * threadIdx.x;
  uint32_t x[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    {
    x[k] = e0 + k < n ? in[e0 + k] : 0u;
    s += x[k];
    }
  const uint32_t inc = wave_incl_scan(s, lane);
  if (lane == 63)
    wsum[wave] = inc;
  __syncthreads();
  uint32_t run = inc - s;
  for (int w = 0; w < wave; ++w)
    run += wsum[w];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    {
    if (e0 + k < n)
      out[e0 + k] = run;
    run += x[k];
    }
  if (threadIdx.x == 255)
    totals[blockIdx.x] = run;
  }

__global__ void __launch_bounds__(256) k_scan_add(uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ block_offs)
  {
  const uint32_t add = block_offs[blockIdx.x];
  const uint32_t e0 = blockIdx.x * SCAN_BLOCK + 8u * threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (e0 + k < n)
      out[e0 + k] += add;
  }

size_t scan_levels_bytes(uint32_t n)
  {
  size_t total = 0;
  while (n > 1)
    {
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    total += align_up(4 * (size_t)nb + 16, 256);
    if (nb == 1)
      break;
    n = nb;
    }
  return total + 256;
  }

} // namespace

size_t scan_workspace(uint32_t n) { return scan_levels_bytes(n); }

// out[i] = in[0] + ... + in[i-1] (wrapping); in == out is allowed
int exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, uint32_t n, uint8_t* d_ws, size_t ws_bytes)
  {
  if (n == 0)
    return 1;
  if (scan_levels_bytes(n) > ws_bytes)
    {
    set_error("exclusive scan: workspace too small");
    return 0;
    }
  hipStream_t st = current_stream();
  const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  uint32_t* totals = (uint32_t*)d_ws;
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(256), 0, st, d_in, d_out, n, totals);
  if (nb > 1)
    {
    const size_t used = align_up(4 * (size_t)nb + 16, 256);
    if (!exclusive_scan_u32(totals, totals, nb, d_ws + used, ws_bytes - used))
      return 0;
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, st, d_out, n, totals);
    }
  return hip_ok(hipGetLastError(), "exclusive scan") ? 1 : 0;
  }

static uint32_t sort_tiles(uint32_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }

size_t sort_workspace(uint32_t n)
  {
  const size_t cells = (size_t)sort_tiles(n) << 10;                       // count matrix of the widest digit
  return 2 * align_up(4 * (size_t)n + 16, 256) + align_up(4 * cells + 16, 256) + scan_workspace((uint32_t)cells) + 256;
  }

// Stable sort of n pairs by the low `key_bits` bits of the key (20 -> two 10-bit passes, otherwise 8-bit passes).
// d_vals_in == nullptr sorts the identity permutation (vals_out[p] = original index of the p-th smallest key).
int radix_sort_pairs(const uint32_t* d_keys_in, const uint32_t* d_vals_in, uint32_t* d_keys_out, uint32_t* d_vals_out, uint32_t n,
                     int key_bits, uint8_t* d_ws, size_t ws_bytes)
  {
  if (n == 0)
    return 1;
  if (sort_workspace(n) > ws_bytes)
    {
    set_error("radix sort: workspace too small");
    return 0;
    }
  hipStream_t st = current_stream();
  const int B = (key_bits % 10 == 0) ? 10 : 8;
  const int passes = (key_bits + B - 1) / B;
  const uint32_t ntiles = sort_tiles(n);
  const uint32_t cells = ntiles << B;
  size_t o = 0;
  uint32_t* tk = (uint32_t*)(d_ws + o); o += align_up(4 * (size_t)n + 16, 256);
  uint32_t* tv = (uint32_t*)(d_ws + o); o += align_up(4 * (size_t)n + 16, 256);
  uint32_t* hist = (uint32_t*)(d_ws + o); o += align_up(4 * ((size_t)ntiles << 10) + 16, 256);
  uint8_t* scan_ws = d_ws + o;
  const size_t scan_bytes = ws_bytes - o;
  // ping-pong so that the last pass lands in the caller's output arrays
  const uint32_t* ki = d_keys_in;
  const uint32_t* vi = d_vals_in;
  for (int p = 0; p < passes; ++p)
    {
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    uint32_t* ko = to_out ? d_keys_out : tk;
    uint32_t* vo = to_out ? d_vals_out : tv;
    const unsigned blocks = (ntiles + 3u) / 4u;
    if (B == 10)
      hipLaunchKernelGGL(k_radix_hist<10>, dim3(blocks), dim3(256), 0, st, ki, n, p * B, ntiles, hist);
    else
      hipLaunchKernelGGL(k_radix_hist<8>, dim3(blocks), dim3(256), 0, st, ki, n, p * B, ntiles, hist);
    if (!exclusive_scan_u32(hist, hist, cells, scan_ws, scan_bytes))
      return 0;
    if (B == 10)
      hipLaunchKernelGGL(k_radix_scatter<10>, dim3(blocks), dim3(256), 0, st, ki, vi, n, p * B, ntiles, hist, ko, vo);
    else
      hipLaunchKernelGGL(k_radix_scatter<8>, dim3(blocks), dim3(256), 0, st, ki, vi, n, p * B, ntiles, hist, ko, vo);
    ki = ko;
    vi = vo;
    }
  return hip_ok(hipGetLastError(), "radix sort") ? 1 : 0;
  }

} // namespace trico
