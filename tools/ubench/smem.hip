// micro-benchmark: scalar memory (s_load / s_store through the scalar data cache) as a table for a serial chain, gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R32(x) R16(x) R16(x)
template <int MODE>
__global__ void k(uint32_t* tab, uint32_t* out, uint32_t n, uint64_t* t) {
  uint32_t sa = __builtin_amdgcn_readfirstlane(out[0]) & 1023u, sb = 0, sc = 77;
  sa *= 4;
  asm volatile("s_dcache_inv\n s_waitcnt lgkmcnt(0)");
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t i = 0; i < n; ++i) {
    if (MODE == 0) asm volatile(R32("s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+s"(sa) : "s"(tab));                       // dependent loads (table of byte offsets)
    if (MODE == 1) asm volatile(R32("s_load_dword %0, %2, %0\n s_waitcnt lgkmcnt(0)\n s_add_u32 %1, %1, %0\n s_xor_b32 %1, %1, %0\n") : "+s"(sa), "+s"(sb) : "s"(tab), "s"(sc) : "scc");
    if (MODE == 2) asm volatile(R32("s_store_dword %2, %1, %0\n s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+s"(sa) : "s"(tab), "s"(sc));   // store elsewhere?? same address then dependent load
    if (MODE == 3) asm volatile(R32("s_load_dword %0, %1, %0\n s_store_dword %2, %3, 0x0\n s_waitcnt lgkmcnt(0)\n") : "+s"(sa) : "s"(tab), "s"(sc), "s"(out + 256));   // load + store to another line
    if (MODE == 4) asm volatile(R32("s_store_dword %1, %2, 0x0\n s_add_u32 %0, %0, 1\n") : "+s"(sb) : "s"(sc), "s"(out + 256) : "scc");     // store issue cost
    if (MODE == 5) { uint32_t tmp; asm volatile(R32("s_load_dword %2, %3, %0\n s_add_u32 %1, %1, 1\n") "s_waitcnt lgkmcnt(0)\n" : "+s"(sa), "+s"(sb), "=&s"(tmp) : "s"(tab) : "scc"); sc += tmp; }  // load issue cost (not waited)
    if (MODE == 6) asm volatile(R32("s_mov_b32 m0, %1\n s_nop 0\n s_movrels_b32 %0, s40\n s_movreld_b32 s40, %0\n s_add_u32 %1, %0, 1\n s_and_b32 %1, %1, 15\n") : "+s"(sa), "+s"(sb) :: "scc", "m0", "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)");
  out[64 + threadIdx.x] = sa + sb + sc;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
// correctness: a pseudo-random sequence of scalar stores and loads against the same sequence on the host
__global__ void kcheck(uint32_t* tab, uint32_t* res, uint32_t n) {
  uint32_t x = 12345u, acc = 0;
  asm volatile("s_dcache_inv\n s_waitcnt lgkmcnt(0)");
  for (uint32_t i = 0; i < n; ++i) {
    x = x * 1664525u + 1013904223u;
    uint32_t aw = ((x >> 10) & 1023u) * 4u, ar = ((x >> 20) & 1023u) * 4u, val = x ^ acc, got;
    x = __builtin_amdgcn_readfirstlane(x); aw = __builtin_amdgcn_readfirstlane(aw); ar = __builtin_amdgcn_readfirstlane(ar); val = __builtin_amdgcn_readfirstlane(val);
    // load issued BEFORE the store (forwarded by hand when the addresses are equal), as the decoder chain will do
    asm volatile("s_load_dword %0, %1, %2\n s_store_dword %3, %1, %4\n s_waitcnt lgkmcnt(0)" : "=&s"(got) : "s"(tab), "s"(ar), "s"(val), "s"(aw) : "memory");
    if (ar == aw) got = val;
    acc = acc * 31u + got;
  }
  asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)");
  if (threadIdx.x == 0) res[0] = acc;
}
int main() {
  uint32_t *tab, *out, *res; uint64_t* t;
  (void)hipMalloc(&tab, 4096); (void)hipMalloc(&out, 4096); (void)hipMalloc(&res, 64); (void)hipMalloc(&t, 64);
  uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = ((i * 7 + 1) & 1023) * 4;
  (void)hipMemset(out, 0, 4096);
  uint64_t ht; const uint32_t n = 20000;
#define RUN(M, name, per) (void)hipMemcpy(tab, h, 4096, hipMemcpyHostToDevice); k<M><<<1, 64>>>(tab, out, n, t); (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); printf("%-64s %.2f cycles   (%s)\n", name, (double)ht / n / per, hipGetErrorString(hipDeviceSynchronize()));
  RUN(0, "dependent s_load_dword (scalar cache hit) + wait", 32)
  RUN(1, "dependent s_load + wait + s_add + s_xor (per group)", 32)
  RUN(2, "s_store then s_load of the same address + wait (per group)", 32)
  RUN(3, "s_load + s_store other line + wait (per group)", 32)
  RUN(4, "s_store + s_add (per pair)", 32)
  RUN(5, "s_load not waited + s_add (per pair)", 32)
  RUN(6, "s_mov m0, nop, s_movrels, s_movreld, s_add, s_and (per group)", 32)
  // correctness against the host
  const uint32_t N = 200000; uint32_t T[1024]; for (int i = 0; i < 1024; ++i) T[i] = 0;
  (void)hipMemset(tab, 0, 4096);
  kcheck<<<1, 64>>>(tab, res, N);
  uint32_t x = 12345u, acc = 0;
  for (uint32_t i = 0; i < N; ++i) { x = x * 1664525u + 1013904223u; uint32_t aw = (x >> 10) & 1023u, ar = (x >> 20) & 1023u, val = x ^ acc; uint32_t got = T[ar]; T[aw] = val; if (ar == aw) got = val; acc = acc * 31u + got; }
  uint32_t dres; (void)hipMemcpy(&dres, res, 4, hipMemcpyDeviceToHost);
  uint32_t dT[1024]; (void)hipMemcpy(dT, tab, 4096, hipMemcpyDeviceToHost);
  int same = 1; for (int i = 0; i < 1024; ++i) same &= (dT[i] == T[i]);
  printf("scalar store/load sequence: device %08x host %08x -> %s; table after s_dcache_wb %s\n", dres, acc, dres == acc ? "EQUAL" : "DIFFERENT", same ? "EQUAL" : "DIFFERENT");
  return 0;
}
