// micro-benchmark: how fast does ONE wave run dependent scalar / vector / LDS chains on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k_salu(uint32_t* out, uint32_t n, uint64_t* t) {
  uint32_t x = __builtin_amdgcn_readfirstlane(out[0]);
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < n; ++i) { x = (x ^ (x >> 3)) + i; x = __builtin_amdgcn_readfirstlane(x); }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[1] = x; t[0] = t1 - t0; t[1] = r1 - r0; }
}
__global__ void k_valu(uint32_t* out, uint32_t n, uint64_t* t) {
  uint32_t x = out[threadIdx.x];
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < n; ++i) { x = (x ^ (x >> 3)) + i; }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = x; if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = r1 - r0; }
}
__global__ void k_mix(uint32_t* out, uint32_t n, uint64_t* t) {   // VALU -> readlane -> SALU -> VALU ...
  uint32_t v = out[threadIdx.x]; uint32_t s = 1;
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < n; ++i) { s = __builtin_amdgcn_readlane(v, s & 63) ^ i; v = (threadIdx.x == (s & 63)) ? s : v; }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = v + s; if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = r1 - r0; }
}
__global__ void k_lds(uint32_t* out, uint32_t n, uint64_t* t) {   // dependent LDS reads (uniform address)
  __shared__ uint32_t tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 7 + 1) & 1023;
  __syncthreads();
  uint32_t x = out[0] & 1023;
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < n; ++i) { x = tab[x]; }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = x; if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = r1 - r0; }
}
__global__ void k_gld(const uint32_t* __restrict__ buf, uint32_t* out, uint32_t n, uint64_t* t) {   // dependent global loads, L2-resident
  uint32_t x = out[0] & 0xffff;
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < n; ++i) { x = buf[x]; }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = x; if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = r1 - r0; }
}
int main() {
  uint32_t* out; uint64_t* t; uint32_t* buf;
  hipMalloc(&out, 1024); hipMemset(out, 0, 1024); hipMalloc(&t, 64); hipMalloc(&buf, 65536 * 4);
  uint32_t* h = (uint32_t*)malloc(65536 * 4); for (int i = 0; i < 65536; ++i) h[i] = (i * 4099 + 17) & 0xffff;
  hipMemcpy(buf, h, 65536 * 4, hipMemcpyHostToDevice);
  uint64_t ht[2]; const uint32_t n = 2000000;
  for (int rep = 0; rep < 2; ++rep) {
  k_salu<<<1, 64>>>(out, n, t); hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost); printf("salu chain (3 salu+readfirstlane): %.1f cyc/iter, clk %.0f MHz\n", (double)ht[0] / n, (double)ht[0] / ht[1] * 100.0);
  k_valu<<<1, 64>>>(out, n, t); hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost); printf("valu chain (3 dependent valu): %.1f cyc/iter, clk %.0f MHz\n", (double)ht[0] / n, (double)ht[0] / ht[1] * 100.0);
  k_mix<<<1, 64>>>(out, n, t); hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost); printf("readlane->salu->cmp/cndmask chain: %.1f cyc/iter, clk %.0f MHz\n", (double)ht[0] / n, (double)ht[0] / ht[1] * 100.0);
  k_lds<<<1, 64>>>(out, n, t); hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost); printf("dependent LDS read: %.1f cyc/iter, clk %.0f MHz\n", (double)ht[0] / n, (double)ht[0] / ht[1] * 100.0);
  k_gld<<<1, 64>>>(buf, out, n / 10, t); hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost); printf("dependent global load (256 KiB table): %.1f cyc/iter, clk %.0f MHz\n", (double)ht[0] / (n / 10), (double)ht[0] / ht[1] * 100.0);
  }
  return 0;
}
