// micro-benchmark: issue rate of ONE wave on gfx950 for dependent / independent SALU and VALU streams
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(uint32_t* out, uint32_t n, uint64_t* t) {
  uint32_t a = out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3;
  uint32_t sa = __builtin_amdgcn_readfirstlane(a), sb = sa + 1, sc = sa + 2, sd = sa + 3;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t i = 0; i < n; ++i) {
    if (MODE == 0) { REP16(asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));) }                       // dependent VALU
    if (MODE == 1) { REP16(asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }  // 4 independent chains
    if (MODE == 2) { REP16(asm volatile("s_add_u32 %0, %0, %0" : "+s"(sa) :: "scc");) }                      // dependent SALU
    if (MODE == 3) { REP16(asm volatile("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3" : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd) :: "scc");) }
    if (MODE == 4) { REP16(asm volatile("v_readlane_b32 %0, %1, 3\n s_nop 0\n v_add_u32 %1, %0, %1" : "+s"(sa), "+v"(a));) }   // VALU->SGPR->VALU ping-pong
    if (MODE == 5) { REP16(asm volatile("s_add_u32 %0, %0, 1\n v_add_u32 %1, %0, %1" : "+s"(sa), "+v"(a) :: "scc");) }          // SALU feeding VALU
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[64 + threadIdx.x] = a + b + c + d + sa + sb + sc + sd;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  uint32_t* out; uint64_t* t; (void)hipMalloc(&out, 1024); (void)hipMemset(out, 0, 1024); (void)hipMalloc(&t, 64);
  uint64_t ht; const uint32_t n = 100000;
  const char* names[6] = {"16 dependent v_add", "16 x 4 independent v_add", "16 dependent s_add", "16 x 4 independent s_add", "16 x (readlane, nop, v_add) ping-pong", "16 x (s_add -> v_add)"};
  const int per[6] = {16, 64, 16, 64, 48, 32};
#define RUN(M) k<M><<<1, 64>>>(out, n, t); (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); printf("%-44s %.2f cycles/instruction\n", names[M], (double)ht / n / per[M]);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  return 0;
}
