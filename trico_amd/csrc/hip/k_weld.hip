// k_weld.hip — vertex welding for the STL reader on the GPU (trico_io/iostl.c:69-134).
//
// A binary STL stores three positions per triangle; the reader welds identical positions into one vertex.  The
// reference does it with a quicksort of (x, y, z, corner) records; the outcome is "unique positions in (x, y, z)
// lexicographic order, every corner re-indexed to the rank of its position", and it does not depend on the order of
// equal records as long as equal-comparing positions are bit-identical — i.e. unless -0.0 meets +0.0 (they compare
// equal but differ in bits: which one represents the vertex depends on the quicksort's swaps) or NaNs are present
// (they never compare equal or less).  Those inputs are detected here and left to the host implementation, which
// reproduces the reference's partition scheme; everything else is three stable radix sorts (z, then y, then x, on an
// order-preserving integer image of the floats; k_sort.hip), an adjacent compare, an exclusive scan and
// a scatter.
#include "common.hpp"

namespace trico {

namespace {

__device__ __forceinline__ uint32_t orderable(uint32_t f) { return f ^ ((f >> 31) ? 0xffffffffu : 0x80000000u); }

// component keys of every corner (already gathered through `perm` when given) + a flag for -0.0 / NaN
__global__ void __launch_bounds__(256) k_weld_keys(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ perm, uint32_t n, int comp,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ iota, uint32_t* __restrict__ flag)
  {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n)
    return;
  const uint32_t i = perm ? perm[p] : p;
  const uint32_t f = pos[3u * (size_t)i + comp];
  keys[p] = orderable(f);
  if (iota)
    iota[p] = p;
  if (flag && (f == 0x80000000u || (f & 0x7fffffffu) > 0x7f800000u))
    atomicOr(flag, 1u);
  }

__global__ void __launch_bounds__(256) k_weld_first(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ perm, uint32_t n,
                                                    uint32_t* __restrict__ first)
  {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n)
    return;
  uint32_t f = 1u;
  if (p > 0)
    {
    const uint32_t* a = pos + 3u * (size_t)perm[p];
    const uint32_t* b = pos + 3u * (size_t)perm[p - 1];
    f = (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) ? 1u : 0u;
    }
  first[p] = f;
  }

// rank[p] = exclusive sum of first[] = (vertex id of sorted position p) + first[p] - 1 ... see below
__global__ void __launch_bounds__(256) k_weld_scatter(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ first,
                                                      const uint32_t* __restrict__ excl, uint32_t n, uint32_t* __restrict__ out_pos,
                                                      uint32_t* __restrict__ out_tri, uint32_t* __restrict__ nv)
  {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n)
    return;
  const uint32_t vid = excl[p] + first[p] - 1u;                  // inclusive count of run starts up to p, minus one
  const uint32_t corner = perm[p];
  out_tri[corner] = vid;
  if (first[p])
    {
    const uint32_t* a = pos + 3u * (size_t)corner;
    out_pos[3u * (size_t)vid] = a[0];
    out_pos[3u * (size_t)vid + 1u] = a[1];
    out_pos[3u * (size_t)vid + 2u] = a[2];
    }
  if (p == n - 1u)
    *nv = vid + 1u;
  }

} // namespace

// d_pos: n = 3 * triangles corner positions (xyz floats as bits).  Returns 1 and fills d_out_pos (first *nv positions),
// d_out_tri (n indices) and h_result = { nv, special } ; special != 0 means -0.0 / NaN present: outputs are not valid.
int launch_weld(const uint32_t* d_pos, uint32_t n, uint32_t* d_out_pos, uint32_t* d_out_tri, uint8_t* d_ws, size_t ws_bytes, uint32_t* d_result)
  {
  hipStream_t st = current_stream();
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes + 16, 256); return at; };
  uint32_t* keys = (uint32_t*)(d_ws + take(4 * (size_t)n));
  uint32_t* keys2 = (uint32_t*)(d_ws + take(4 * (size_t)n));
  uint32_t* permA = (uint32_t*)(d_ws + take(4 * (size_t)n));
  uint32_t* permB = (uint32_t*)(d_ws + take(4 * (size_t)n));
  const size_t sort_bytes = sort_workspace(n), scan_bytes = scan_workspace(n);
  const size_t tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  uint8_t* tmp = d_ws + take(tmp_bytes);
  if (o > ws_bytes)
    {
    set_error("weld: workspace too small");
    return 0;
    }
  const unsigned blocks = (n + 255u) / 256u;
  if (!hip_ok(hipMemsetAsync(d_result, 0, 8, st), "memset(weld result)"))
    return 0;
  // least significant key first: z, then y, then x (each sort is stable)
  uint32_t* cur = nullptr;                                       // permutation so far (nullptr = identity)
  uint32_t* bufs[2] = { permA, permB };
  for (int pass = 0; pass < 3; ++pass)
    {
    const int comp = 2 - pass;
    uint32_t* vals_in = pass == 0 ? bufs[0] : cur;
    uint32_t* vals_out = (vals_in == bufs[0]) ? bufs[1] : bufs[0];
    hipLaunchKernelGGL(k_weld_keys, dim3(blocks), dim3(256), 0, st, d_pos, cur, n, comp, keys, pass == 0 ? bufs[0] : (uint32_t*)nullptr,
                       d_result + 1);
    if (!radix_sort_pairs(keys, vals_in, keys2, vals_out, n, 32, tmp, tmp_bytes))
      return 0;
    cur = vals_out;
    }
  hipLaunchKernelGGL(k_weld_first, dim3(blocks), dim3(256), 0, st, d_pos, cur, n, keys);
  if (!exclusive_scan_u32(keys, keys2, n, tmp, tmp_bytes))
    return 0;
  hipLaunchKernelGGL(k_weld_scatter, dim3(blocks), dim3(256), 0, st, d_pos, cur, keys, keys2, n, d_out_pos, d_out_tri, d_result);
  return hip_ok(hipGetLastError(), "weld kernels") ? 1 : 0;
  }

size_t weld_workspace(uint32_t n)
  {
  const size_t sort_bytes = sort_workspace(n), scan_bytes = scan_workspace(n);
  return 4 * align_up(4 * (size_t)n + 16, 256) + align_up((sort_bytes > scan_bytes ? sort_bytes : scan_bytes) + 16, 256) + 256;
  }

} // namespace trico
