// k_fpc64_sort.hip — throughput encoder for 64-bit floating-point streams: the table lookups become sorts.
//
// Replaces trico_compress_double_precision(..., 20, 20) (fpsc.c:576-800) + the xyz / uv transposes for streams
// large enough to pay for it.  The predictors' tables have 2^20 entries each, far too many classes for the
// per-segment LDS tables of the float encoder, but the exactness argument is the same: the FCM hash
// of value i is the top 20 bits of v[i-1], the DFCM hash a function of the strides of v[i-1] and v[i-2], so every
// value's two hashes are known from the input alone, and a table read returns the payload (value / stride) of the
// latest earlier value with the same hash, or 0.  "Latest earlier element with the same key" is the predecessor in a
// stable sort by key:
//   keys    (k64_keys):   both hashes of every value, and an index array
//   sort    (k_sort.hip: stable LSD radix sort, two 10-bit passes): (hash, index) pairs, one sort per table
//   preds   (k64_pred):   sorted neighbour with the same hash -> its payload, scattered back to the value's index
//   sizes   (k64_sizes):  code selection (fpsc.c:640-700) -> bytes per group of two values (1 header byte + residuals)
//   scan    (k_sort.hip: exclusive sum): byte offset of every group
//   emit    (k64_emit):   tiles of 1024 groups are packed in LDS and written out with aligned dword stores
// Everything is data-parallel; HBM traffic is ~25 passes over 4-8 byte arrays per value (sort-dominated), which on
// this machine is two orders of magnitude cheaper than walking the 16 MiB tables value by value.
#include "common.hpp"
#include <stdlib.h>

namespace trico {

namespace {

typedef uint64_t u64;

constexpr int TILE_G = 1024;                 // groups per emit tile
constexpr int TILE_T = 256;                  // threads per emit workgroup (4 groups each)
constexpr int TILE_BYTES = TILE_G * 17;      // a group is at most 1 + 8 + 8 bytes

__device__ __forceinline__ u64 val_at(const u64* __restrict__ src, int64_t i, int arity, int c)
  {
  return i >= 0 ? src[(size_t)i * arity + c] : 0ull;          // values before the stream count as 0 (fpsc.c:596-600)
  }

__global__ void __launch_bounds__(256) k64_keys(const u64* __restrict__ src, uint32_t n, int arity, int c,
                                                uint32_t* __restrict__ k1, uint32_t* __restrict__ k2)
  {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  const u64 v1 = val_at(src, (int64_t)i - 1, arity, c), v2 = val_at(src, (int64_t)i - 2, arity, c), v3 = val_at(src, (int64_t)i - 3, arity, c);
  const u64 s1 = v1 - v2, s2 = v2 - v3;
  k1[i] = (uint32_t)(v1 >> 44);                                             // fpsc.c:565-568 with a 20-bit table
  k2[i] = (uint32_t)(((((s2 >> 44) & 1023ull) << 10) ^ (s1 >> 44)) & 0xfffffull);   // fpsc.c:570-573: two strides of history
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

__device__ __forceinline__ uint32_t blen64(u64 x) { return x ? (uint32_t)(71 - __builtin_clzll(x)) >> 3 : 0u; }

// code, residual and residual length of value i (fpsc.c:640-700): FCM codes 0..8, DFCM codes 9..15 (1..7 bytes)
__device__ __forceinline__ uint32_t code_of(const u64* __restrict__ src, uint32_t i, int arity, int c, const u64* __restrict__ pred1,
                                            const u64* __restrict__ pred2, u64& x, uint32_t& len)
  {
  const u64 v = src[(size_t)i * arity + c], a = val_at(src, (int64_t)i - 1, arity, c);
  const u64 x1 = v ^ pred1[i], x2 = v ^ (a + pred2[i]);
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

__global__ void __launch_bounds__(256) k64_sizes(const u64* __restrict__ src, uint32_t n, int arity, int c, const u64* __restrict__ pred1,
                                                 const u64* __restrict__ pred2, uint32_t ngroups, uint32_t* __restrict__ gsz)
  {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= ngroups)
    return;
  u64 x;
  uint32_t l0, l1 = 1u;                                        // a missing second value is padded with code 1, one 0x00 byte
  code_of(src, 2u * g, arity, c, pred1, pred2, x, l0);
  if (2u * g + 1u < n)
    code_of(src, 2u * g + 1u, arity, c, pred1, pred2, x, l1);
  gsz[g] = 1u + l0 + l1;
  }

__global__ void __launch_bounds__(TILE_T) k64_emit(const u64* __restrict__ src, uint32_t n, int arity, int c, const u64* __restrict__ pred1,
                                                   const u64* __restrict__ pred2, uint32_t ngroups, const uint32_t* __restrict__ goff,
                                                   const uint32_t* __restrict__ gsz, uint8_t* __restrict__ out, uint32_t* __restrict__ size_out)
  {
  __shared__ __attribute__((aligned(16))) uint8_t lds[TILE_BYTES + 16];
  const uint32_t g0 = blockIdx.x * TILE_G;
  const uint32_t g1 = g0 + TILE_G < ngroups ? g0 + TILE_G : ngroups;
  const uint32_t base = goff[g0];
  const uint32_t end = goff[g1 - 1] + gsz[g1 - 1];
  for (uint32_t g = g0 + threadIdx.x; g < g1; g += TILE_T)
    {
    uint8_t* o = lds + (goff[g] - base);
    u64 x0, x1 = 0;
    uint32_t l0, l1 = 1u, c1 = 1u;
    const uint32_t c0 = code_of(src, 2u * g, arity, c, pred1, pred2, x0, l0);
    if (2u * g + 1u < n)
      c1 = code_of(src, 2u * g + 1u, arity, c, pred1, pred2, x1, l1);
    *o++ = (uint8_t)((c1 << 4) | c0);                           // fpsc.c:706
    for (uint32_t k = l0; k > 0; --k) *o++ = (uint8_t)(x0 >> (8u * (k - 1u)));
    for (uint32_t k = l1; k > 0; --k) *o++ = (uint8_t)(x1 >> (8u * (k - 1u)));
    }
  __syncthreads();
  // tile bytes [base, end) of the group area go to out + 5 + base: bytes up to the first aligned dword, then dwords
  uint8_t* d = out + 5u + base;
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
    out[0] = 0xaa;                                               // (20/2) << 4 | (20/2), fpsc.c:610
    out[1] = (uint8_t)(n >> 24); out[2] = (uint8_t)(n >> 16); out[3] = (uint8_t)(n >> 8); out[4] = (uint8_t)n;
    }
  if (g1 == ngroups && threadIdx.x == 0)
    *size_out = 5u + end;
  }

struct SortPlan { size_t k1, k1s, k2, k2s, v1s, v2s, pred1, pred2, gsz, goff, tmp, tmp_bytes, total; };

SortPlan plan_for(uint32_t n)
  {
  SortPlan p;
  const size_t ng = ((size_t)n + 1) / 2;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes + 16, 256); return at; };
  p.k1 = take(4 * (size_t)n); p.k1s = take(4 * (size_t)n); p.k2 = take(4 * (size_t)n); p.k2s = take(4 * (size_t)n);
  p.v1s = take(4 * (size_t)n); p.v2s = take(4 * (size_t)n);
  p.pred1 = take(8 * (size_t)n); p.pred2 = take(8 * (size_t)n);
  p.gsz = take(4 * ng); p.goff = take(4 * ng);
  const size_t sort_bytes = sort_workspace(n), scan_bytes = scan_workspace((uint32_t)ng);
  p.tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  p.tmp = take(p.tmp_bytes);
  p.total = o;
  return p;
  }

} // namespace

uint32_t fpc64_sorted_threshold()
  {
  static uint32_t t = 0;
  if (!t)
    {
    const char* e = getenv("TRICO_FPC64_SORT_MIN");              // tuning knob: values per stream from which the sort path is used
    t = e ? (uint32_t)strtoul(e, nullptr, 10) : 65536u;
    if (t < 2) t = 2;
    }
  return t;
  }

size_t fpc64_sorted_workspace(uint32_t n)
  {
  return plan_for(n).total;
  }

int launch_fpc64_encode_sorted(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                               uint8_t* d_ws, size_t ws_bytes)
  {
  if (n < 2 || n > 0x7fffffffu)
    {
    set_error("fpc64 sort encoder: unsupported count");
    return 0;
    }
  const SortPlan p = plan_for(n);
  if (p.total > ws_bytes)
    {
    set_error("fpc64 sort encoder: workspace too small");
    return 0;
    }
  hipStream_t st = current_stream();
  const u64* src = (const u64*)d_src;
  const uint32_t ng = (n + 1u) / 2u;
  const unsigned vb = (n + 255u) / 256u, gb = (ng + 255u) / 256u, tiles = (ng + TILE_G - 1) / TILE_G;
  uint32_t* k1 = (uint32_t*)(d_ws + p.k1); uint32_t* k1s = (uint32_t*)(d_ws + p.k1s);
  uint32_t* k2 = (uint32_t*)(d_ws + p.k2); uint32_t* k2s = (uint32_t*)(d_ws + p.k2s);
  uint32_t* v1s = (uint32_t*)(d_ws + p.v1s); uint32_t* v2s = (uint32_t*)(d_ws + p.v2s);
  u64* pred1 = (u64*)(d_ws + p.pred1); u64* pred2 = (u64*)(d_ws + p.pred2);
  uint32_t* gsz = (uint32_t*)(d_ws + p.gsz); uint32_t* goff = (uint32_t*)(d_ws + p.goff);
  for (int c = 0; c < arity; ++c)
    {
    hipLaunchKernelGGL(k64_keys, dim3(vb), dim3(256), 0, st, src, n, arity, c, k1, k2);
    // values = identity: the sorted value of position p is the index of the p-th value in (hash, index) order
    if (!radix_sort_pairs(k1, nullptr, k1s, v1s, n, 20, d_ws + p.tmp, p.tmp_bytes) ||
        !radix_sort_pairs(k2, nullptr, k2s, v2s, n, 20, d_ws + p.tmp, p.tmp_bytes))
      return 0;
    hipLaunchKernelGGL(k64_pred, dim3(vb), dim3(256), 0, st, k1s, v1s, src, n, arity, c, 0, pred1);
    hipLaunchKernelGGL(k64_pred, dim3(vb), dim3(256), 0, st, k2s, v2s, src, n, arity, c, 1, pred2);
    hipLaunchKernelGGL(k64_sizes, dim3(gb), dim3(256), 0, st, src, n, arity, c, pred1, pred2, ng, gsz);
    if (!exclusive_scan_u32(gsz, goff, ng, d_ws + p.tmp, p.tmp_bytes))
      return 0;
    hipLaunchKernelGGL(k64_emit, dim3(tiles), dim3(TILE_T), 0, st, src, n, arity, c, pred1, pred2, ng, goff, gsz,
                       d_out + (size_t)c * out_stride, d_sizes + c);
    }
  return hip_ok(hipGetLastError(), "fpc64 sort encoder") ? 1 : 0;
  }

} // namespace trico
