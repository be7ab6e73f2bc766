// Latency of a dependent scalar load (s_load_dwordx2) as a function of the table it walks: what a chain decoder pays per table entry it has
// to wait for.  The double decoder's DFCM table is 8 MiB per stream and is hit at random (k_fpc64.hip); this program chases a random cycle
// through tables of 16 KiB .. 64 MiB with ONE wave (one workgroup), like one chain, and prints cycles per hop (s_memtime) and ns at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ void k_chase(const uint64_t* __restrict__ tab, uint32_t hops, uint64_t* __restrict__ out)
  {
  uint64_t idx = 0, t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (uint32_t i = 0; i < hops; ++i)
    {
    uint64_t nxt;
    asm volatile("s_load_dwordx2 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(nxt) : "s"(tab), "s"((uint32_t)(idx * 8u)) : "memory");
    idx = nxt;
    }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
  }

int main()
  {
  const size_t sizes[] = { 16u << 10, 256u << 10, 2u << 20, 8u << 20, 16u << 20, 64u << 20 };
  uint64_t* dout;
  if (hipMalloc(&dout, 16) != hipSuccess) return 1;
  for (size_t s : sizes)
    {
    const size_t n = s / 8;
    std::vector<uint64_t> perm(n), tab(n);
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    uint32_t x = 0x9e3779b9u;
    for (size_t i = n - 1; i > 0; --i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; std::swap(perm[i], perm[x % (i + 1)]); }
    for (size_t i = 0; i < n; ++i) tab[perm[i]] = perm[(i + 1) % n];     // one cycle through all entries, in random order
    uint64_t* dtab;
    if (hipMalloc(&dtab, s) != hipSuccess) return 1;
    (void)hipMemcpy(dtab, tab.data(), s, hipMemcpyHostToDevice);
    const uint32_t hops = 200000;
    for (int rep = 0; rep < 2; ++rep)
      hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, dtab, hops, dout);
    uint64_t h[2];
    (void)hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
    printf("table %8zu KiB: %7.1f cycles per dependent scalar load (s_memtime ticks / hop; second pass over the same cycle)\n", s >> 10, (double)h[0] / hops);
    (void)hipFree(dtab);
    }
  return 0;
  }
