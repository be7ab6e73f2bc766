// Known-traffic check of the FETCH_SIZE counter on gfx950: how many bytes does it report for (a) wide streaming reads
// (16 bytes per lane, every byte once) and (b) the float encoder's pattern (a dword per lane at a stride of 12 bytes, the three
// components of an interleaved xyz array read by three different waves)?  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetchsize
// and compare the counter (KB per dispatch) with the bytes printed here.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_wide(const u32x4* __restrict__ p, size_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { u32x4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345) sink[0] = acc;
}
// one wave per (segment, component): lane l reads src[(i0 + 64 * step + l) * 3 + c], like k_fpc32_code's load_block
__global__ void k_strided(const uint32_t* __restrict__ src, uint32_t n, uint32_t L, uint32_t* sink) {
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t i0 = blockIdx.x * L, i1 = i0 + L < n ? i0 + L : n;
  uint32_t acc = 0;
  for (uint32_t i = i0 + lane; i < i1; i += 64) acc += src[(size_t)i * 3 + c];
  if (acc == 0x12345) sink[0] = acc;
}
int main() {
  const uint32_t n = 50000000;                       // vertices
  const size_t bytes = (size_t)n * 12;
  uint32_t *d, *sink; (void)hipMalloc(&d, bytes + 64); (void)hipMalloc(&sink, 64); (void)hipMemset(d, 1, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    k_wide<<<8192, 256>>>((const u32x4*)d, bytes / 16, sink);
    (void)hipDeviceSynchronize();
    const uint32_t L = 19584;                        // the encoder's segment length at 7680 waves
    k_strided<<<(n + L - 1) / L, 192>>>(d, n, L, sink);
    (void)hipDeviceSynchronize();
  }
  printf("each kernel reads %zu bytes = %zu KB exactly once\n", bytes, bytes / 1024);
  return 0;
}
