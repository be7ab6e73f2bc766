// stress test: do scalar stores (dirty lines in the scalar data cache) survive other kernels being dispatched on the same GPU?
// plus: is a scalar load issued right after a scalar store to the same address ordered behind it?  are the low address bits ignored?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <stdlib.h>
__global__ void kcheck(uint32_t* tab, uint32_t* res, uint32_t n, int store_first) {
  uint32_t x = 12345u + blockIdx.x, acc = 0;
  tab += 1024 * blockIdx.x;
  for (uint32_t i = 0; i < 1024; i += 4) {   // the kernel zeroes its own table with scalar stores
    uint32_t off = __builtin_amdgcn_readfirstlane(i * 4);
    asm volatile("s_mov_b32 s40, 0\n s_mov_b32 s41, 0\n s_mov_b32 s42, 0\n s_mov_b32 s43, 0\n s_store_dwordx4 s[40:43], %0, %1" :: "s"(tab), "s"(off) : "s40", "s41", "s42", "s43", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  for (uint32_t i = 0; i < n; ++i) {
    x = x * 1664525u + 1013904223u;
    uint32_t aw = ((x >> 10) & 1023u) * 4u, ar = ((x >> 20) & 1023u) * 4u, val = x ^ acc, got;
    if ((x & 7u) == 0u) ar = aw;                       // same address often
    x = __builtin_amdgcn_readfirstlane(x); aw = __builtin_amdgcn_readfirstlane(aw); ar = __builtin_amdgcn_readfirstlane(ar); val = __builtin_amdgcn_readfirstlane(val);
    if (store_first)
      {
      uint32_t arj = ar | (x & 3u);                    // junk in the two low address bits of the load
      arj = __builtin_amdgcn_readfirstlane(arj);
      asm volatile("s_store_dword %3, %1, %4\n s_load_dword %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=&s"(got) : "s"(tab), "s"(arj), "s"(val), "s"(aw) : "memory");
      }
    else
      {
      asm volatile("s_load_dword %0, %1, %2\n s_store_dword %3, %1, %4\n s_waitcnt lgkmcnt(0)" : "=&s"(got) : "s"(tab), "s"(ar), "s"(val), "s"(aw) : "memory");
      if (ar == aw) got = val;
      }
    acc = acc * 31u + got;
  }
  asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)");
  if (threadIdx.x == 0) res[blockIdx.x] = acc;
}
__global__ void noise(uint32_t* p, uint32_t k) { p[blockIdx.x * 64 + threadIdx.x] += k; }
// a neighbour that uses the scalar cache: every wave walks a big buffer with scalar loads (argv[4] = 1)
__global__ void evictor(const uint32_t* big, uint32_t words, uint32_t* sink, uint32_t iters) {
  uint32_t x = 99991u * (blockIdx.x + 1u), acc = 0;
  for (uint32_t i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    uint32_t off = __builtin_amdgcn_readfirstlane(((x >> 8) % words) * 4u), got;
    asm volatile("s_load_dword %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(got) : "s"(big), "s"(off) : "memory");
    acc += got;
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}
static uint32_t host_ref(uint32_t n, uint32_t b) {
  std::vector<uint32_t> T(1024, 0); uint32_t x = 12345u + b, acc = 0;
  for (uint32_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; uint32_t aw = (x >> 10) & 1023u, ar = (x >> 20) & 1023u, val = x ^ acc; if ((x & 7u) == 0u) ar = aw; T[aw] = val; uint32_t got = T[ar]; acc = acc * 31u + got; }
  return acc;
}
static uint32_t host_ref_loadfirst(uint32_t n, uint32_t b) {
  std::vector<uint32_t> T(1024, 0); uint32_t x = 12345u + b, acc = 0;
  for (uint32_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; uint32_t aw = (x >> 10) & 1023u, ar = (x >> 20) & 1023u, val = x ^ acc; if ((x & 7u) == 0u) ar = aw; uint32_t got = T[ar]; T[aw] = val; if (ar == aw) got = val; acc = acc * 31u + got; }
  return acc;
}
int main(int argc, char** argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 24; const uint32_t N = argc > 2 ? (uint32_t)atoi(argv[2]) : 8000000u;
  const int LDS = argc > 3 ? atoi(argv[3]) : 0;     // dynamic LDS claimed per workgroup (88 KB: one chain per CU, as the decoder does)
  if (LDS) (void)hipFuncSetAttribute((const void*)kcheck, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const int EVICT = argc > 4 ? atoi(argv[4]) : 0;
  uint32_t *big, *sink; (void)hipMalloc(&big, 32u << 20); (void)hipMemset(big, 1, 32u << 20); (void)hipMalloc(&sink, 4 * 8192);
  uint32_t *tab, *res, *np; (void)hipMalloc(&tab, 4096 * NB); (void)hipMalloc(&res, 4 * NB); (void)hipMalloc(&np, 4 * 64 * 1024);
  (void)hipMemset(np, 0, 4 * 64 * 1024);
  hipStream_t sa, sb; (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  for (int mode = 0; mode < 4; ++mode) {
    const int store_first = mode & 1, with_noise = mode >> 1;
    (void)hipMemset(tab, 0xff, 4096 * NB);
    (void)hipDeviceSynchronize();
    kcheck<<<NB, 64, LDS, sa>>>(tab, res, N, store_first);
    int launches = 0;
    if (with_noise)
      while (hipStreamQuery(sa) == hipErrorNotReady) { if (EVICT) evictor<<<2048, 64, 0, sb>>>(big, 8u << 20, sink, 2000); else noise<<<1024, 64, 0, sb>>>(np, 1); (void)hipMemsetAsync(np, 0, 4096, sb); ++launches; if ((launches & 63) == 0) (void)hipStreamSynchronize(sb); }
    (void)hipDeviceSynchronize();
    std::vector<uint32_t> h(NB); (void)hipMemcpy(h.data(), res, 4 * NB, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; ++b) bad += h[b] != (store_first ? host_ref(N, b) : host_ref_loadfirst(N, b));
    printf("%s, %s (%d noise launches): %d of %d chains differ from the host\n", store_first ? "store then load (no forwarding, junk low address bits)" : "load then store (forwarded by hand)",
           with_noise ? "other kernels dispatched meanwhile" : "alone", launches, bad, NB);
  }
  return 0;
}
