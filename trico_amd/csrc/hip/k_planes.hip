// k_planes.hip — byte-plane split / merge for u16/u32/u64 element arrays.
// Replaces trico_transpose_uint{16,32,64}_aos_to_soa / _soa_to_aos (transpose_aos_to_soa.c:84-147):
// plane k holds byte k (little-endian order) of every element.  Pure HBM-bound byte shuffling:
// each lane moves 16 consecutive elements, reading WIDTH x 16-byte vectors of the AoS array and
// writing one 16-byte vector per plane, so every plane store is a fully coalesced 1 KiB per wave.
// Algorithmic bytes per element: WIDTH read + WIDTH written.
#include "common.hpp"

namespace trico {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// byte k of four consecutive WIDTH-byte elements held in `w` (WIDTH/4.. words), packed into a u32
template <int WIDTH> __device__ __forceinline__ uint32_t gather4(const uint32_t* w, int k);

template <> __device__ __forceinline__ uint32_t gather4<2>(const uint32_t* w, int k)
  {
  // 4 u16 elements in w[0..1]
  const uint32_t s = 8 * k;
  return ((w[0] >> s) & 0xffu) | (((w[0] >> (16 + s)) & 0xffu) << 8) | (((w[1] >> s) & 0xffu) << 16) | (((w[1] >> (16 + s)) & 0xffu) << 24);
  }
template <> __device__ __forceinline__ uint32_t gather4<4>(const uint32_t* w, int k)
  {
  const uint32_t s = 8 * k;
  return ((w[0] >> s) & 0xffu) | (((w[1] >> s) & 0xffu) << 8) | (((w[2] >> s) & 0xffu) << 16) | (((w[3] >> s) & 0xffu) << 24);
  }
template <> __device__ __forceinline__ uint32_t gather4<8>(const uint32_t* w, int k)
  {
  // 4 u64 elements in w[0..7]; byte k lives in word (k>>2) of each element
  const int h = k >> 2;
  const uint32_t s = 8 * (k & 3);
  return ((w[h] >> s) & 0xffu) | (((w[2 + h] >> s) & 0xffu) << 8) | (((w[4 + h] >> s) & 0xffu) << 16) | (((w[6 + h] >> s) & 0xffu) << 24);
  }

template <int WIDTH>
__global__ void __launch_bounds__(256) k_planes_split(const uint8_t* __restrict__ src, uint32_t count, uint8_t* __restrict__ planes,
                                                      size_t plane_stride, int vec_ok)
  {
  const size_t e0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (e0 >= count)
    return;
  if (vec_ok && e0 + 16 <= count)
    {
    uint32_t w[4 * WIDTH];   // 16 elements
    const u32x4* p = (const u32x4*)(src + e0 * WIDTH);
#pragma unroll
    for (int i = 0; i < WIDTH; ++i)
      {
      const u32x4 v = p[i];
      w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
      }
#pragma unroll
    for (int k = 0; k < WIDTH; ++k)
      {
      u32x4 o;
      o.x = gather4<WIDTH>(w, k);
      o.y = gather4<WIDTH>(w + WIDTH, k);
      o.z = gather4<WIDTH>(w + 2 * WIDTH, k);
      o.w = gather4<WIDTH>(w + 3 * WIDTH, k);
      *(u32x4*)(planes + (size_t)k * plane_stride + e0) = o;
      }
    }
  else
    {
    const size_t e1 = (e0 + 16 < count) ? e0 + 16 : count;
    for (size_t e = e0; e < e1; ++e)
      for (int k = 0; k < WIDTH; ++k)
        planes[(size_t)k * plane_stride + e] = src[e * WIDTH + k];
    }
  }

template <int WIDTH>
__global__ void __launch_bounds__(256) k_planes_merge(const uint8_t* __restrict__ planes, size_t plane_stride, uint32_t count,
                                                      uint8_t* __restrict__ dst, int vec_ok)
  {
  const size_t e0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (e0 >= count)
    return;
  if (vec_ok && e0 + 16 <= count)
    {
    uint32_t pw[WIDTH][4];   // plane k: 16 bytes = bytes k of elements e0..e0+15
#pragma unroll
    for (int k = 0; k < WIDTH; ++k)
      {
      const u32x4 v = *(const u32x4*)(planes + (size_t)k * plane_stride + e0);
      pw[k][0] = v.x; pw[k][1] = v.y; pw[k][2] = v.z; pw[k][3] = v.w;
      }
    u32x4* q = (u32x4*)(dst + e0 * WIDTH);
    uint32_t w[4 * WIDTH];
#pragma unroll
    for (int e = 0; e < 16; ++e)
      {
      // element e: byte k comes from plane k, word e>>2, byte e&3
#pragma unroll
      for (int h = 0; h < (WIDTH + 3) / 4; ++h)
        {
        uint32_t x = 0;
#pragma unroll
        for (int b = 0; b < 4 && 4 * h + b < WIDTH; ++b)
          x |= ((pw[4 * h + b][e >> 2] >> (8 * (e & 3))) & 0xffu) << (8 * b);
        if (WIDTH == 2)
          {
          if (e & 1) w[e >> 1] |= x << 16;
          else w[e >> 1] = x;
          }
        else
          w[e * (WIDTH / 4) + h] = x;
        }
      }
#pragma unroll
    for (int i = 0; i < WIDTH; ++i)
      {
      u32x4 o;
      o.x = w[4 * i]; o.y = w[4 * i + 1]; o.z = w[4 * i + 2]; o.w = w[4 * i + 3];
      q[i] = o;
      }
    }
  else
    {
    const size_t e1 = (e0 + 16 < count) ? e0 + 16 : count;
    for (size_t e = e0; e < e1; ++e)
      for (int k = 0; k < WIDTH; ++k)
        dst[e * WIDTH + k] = planes[(size_t)k * plane_stride + e];
    }
  }

int launch_planes_split(const void* d_src, uint32_t count, int width, uint8_t* d_planes, size_t plane_stride)
  {
  if (count == 0)
    return 1;
  const int vec_ok = (((uintptr_t)d_src | (uintptr_t)d_planes | plane_stride) & 15) == 0;
  const unsigned blocks = (unsigned)(((size_t)count + 16 * 256 - 1) / (16 * 256));
  const uint8_t* s = (const uint8_t*)d_src;
  switch (width)
    {
    case 2: hipLaunchKernelGGL(k_planes_split<2>, dim3(blocks), dim3(256), 0, current_stream(), s, count, d_planes, plane_stride, vec_ok); break;
    case 4: hipLaunchKernelGGL(k_planes_split<4>, dim3(blocks), dim3(256), 0, current_stream(), s, count, d_planes, plane_stride, vec_ok); break;
    case 8: hipLaunchKernelGGL(k_planes_split<8>, dim3(blocks), dim3(256), 0, current_stream(), s, count, d_planes, plane_stride, vec_ok); break;
    default: set_error("planes_split: unsupported width"); return 0;
    }
  return hip_ok(hipGetLastError(), "k_planes_split") ? 1 : 0;
  }

int launch_planes_merge(const uint8_t* d_planes, size_t plane_stride, uint32_t count, int width, void* d_dst)
  {
  if (count == 0)
    return 1;
  const int vec_ok = (((uintptr_t)d_dst | (uintptr_t)d_planes | plane_stride) & 15) == 0;
  const unsigned blocks = (unsigned)(((size_t)count + 16 * 256 - 1) / (16 * 256));
  uint8_t* d = (uint8_t*)d_dst;
  switch (width)
    {
    case 2: hipLaunchKernelGGL(k_planes_merge<2>, dim3(blocks), dim3(256), 0, current_stream(), d_planes, plane_stride, count, d, vec_ok); break;
    case 4: hipLaunchKernelGGL(k_planes_merge<4>, dim3(blocks), dim3(256), 0, current_stream(), d_planes, plane_stride, count, d, vec_ok); break;
    case 8: hipLaunchKernelGGL(k_planes_merge<8>, dim3(blocks), dim3(256), 0, current_stream(), d_planes, plane_stride, count, d, vec_ok); break;
    default: set_error("planes_merge: unsupported width"); return 0;
    }
  return hip_ok(hipGetLastError(), "k_planes_merge") ? 1 : 0;
  }

// ---- components of interleaved reals (stand-alone transposes of the reference's low-level API) --------------
// Bit copies of 4- or 8-byte elements; one thread per element, component stores coalesced, interleaved side strided.
template <typename T>
__global__ void __launch_bounds__(256) k_deinterleave(const T* __restrict__ aos, uint32_t n, int arity, uint8_t* __restrict__ soa, size_t comp_stride)
  {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  for (int c = 0; c < arity; ++c)
    ((T*)(soa + (size_t)c * comp_stride))[i] = aos[(size_t)i * arity + c];
  }

template <typename T>
__global__ void __launch_bounds__(256) k_interleave(const uint8_t* __restrict__ soa, size_t comp_stride, uint32_t n, int arity, T* __restrict__ aos)
  {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  for (int c = 0; c < arity; ++c)
    aos[(size_t)i * arity + c] = ((const T*)(soa + (size_t)c * comp_stride))[i];
  }

int launch_deinterleave(const void* d_aos, uint32_t n, int arity, int width, uint8_t* d_soa, size_t comp_stride)
  {
  const unsigned blocks = (n + 255u) / 256u;
  if (width == 4)
    hipLaunchKernelGGL(k_deinterleave<uint32_t>, dim3(blocks), dim3(256), 0, current_stream(), (const uint32_t*)d_aos, n, arity, d_soa, comp_stride);
  else
    hipLaunchKernelGGL(k_deinterleave<uint64_t>, dim3(blocks), dim3(256), 0, current_stream(), (const uint64_t*)d_aos, n, arity, d_soa, comp_stride);
  return hip_ok(hipGetLastError(), "k_deinterleave") ? 1 : 0;
  }

int launch_interleave(const uint8_t* d_soa, size_t comp_stride, uint32_t n, int arity, int width, void* d_aos)
  {
  const unsigned blocks = (n + 255u) / 256u;
  if (width == 4)
    hipLaunchKernelGGL(k_interleave<uint32_t>, dim3(blocks), dim3(256), 0, current_stream(), d_soa, comp_stride, n, arity, (uint32_t*)d_aos);
  else
    hipLaunchKernelGGL(k_interleave<uint64_t>, dim3(blocks), dim3(256), 0, current_stream(), d_soa, comp_stride, n, arity, (uint64_t*)d_aos);
  return hip_ok(hipGetLastError(), "k_interleave") ? 1 : 0;
  }


// ---- payload comparison (the decoders' self-check, shim.hip) ---------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k_bytes_compare(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t n,
                                                       const uint32_t* __restrict__ d_size, uint32_t* __restrict__ status, uint32_t flag)
  {
  if (*d_size != n)
    {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, flag);
    return;
    }
  bool diff = false;
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride)     // bytes: the two sides are aligned differently
    diff = diff || a[i] != b[i];
  if (diff)
    atomicOr(status, flag);
  }
}

int launch_bytes_compare(const uint8_t* d_a, const uint8_t* d_b, uint32_t n_expected, const uint32_t* d_size, uint32_t* d_status, uint32_t flag)
  {
  hipLaunchKernelGGL(k_bytes_compare, dim3(4096), dim3(256), 0, current_stream(), d_a, d_b, n_expected, d_size, d_status, flag);
  return hip_ok(hipGetLastError(), "k_bytes_compare") ? 1 : 0;
  }

} // namespace trico
