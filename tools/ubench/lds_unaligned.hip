#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t __attribute__((aligned(1))) u32u;
__global__ void k(uint32_t* out, int shift)
{
  __shared__ uint8_t buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = 0;
  __syncthreads();
  // each lane writes 0xA0+lane.. pattern dword at byte address 5*lane + shift (unaligned)
  uint32_t w = 0x04030201u + 0x10101010u * (threadIdx.x & 15);
  uint32_t addr = 5 * threadIdx.x + shift;
  asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(addr + (uint32_t)(uintptr_t)0), "v"(w) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((uint32_t*)buf)[i];
}
int main()
{
  uint32_t* d; hipMalloc(&d, 1024);
  for (int shift = 0; shift < 4; ++shift) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, shift);
    uint8_t h[1024]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int b = 0; b < 4; ++b) { uint8_t want = (uint8_t)(b + 1 + 0x10 * (l & 15)); if (h[5*l+shift+b] != want) ok = 0; }
    printf("shift %d unaligned ds_write_b32 %s: bytes at lane1: %02x %02x %02x %02x %02x\n", shift, ok ? "OK" : "WRONG", h[5+shift], h[6+shift], h[7+shift], h[8+shift], h[9+shift]);
  }
  return 0;
}
