// micro-benchmark 2: what sets the issue rate of ONE wave on gfx950?  (companions, priority, instruction mix, outstanding LDS)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
#define V4 "v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n"
template <int MODE>
__global__ void k(uint32_t* out, uint32_t n, uint64_t* t, int companions) {
  __shared__ uint32_t tab[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) tab[i] = ((i * 7 + 1) & 1023) * 4;
  __syncthreads();
  uint32_t a = out[threadIdx.x & 63] & 1023, b = a + 1, c = a + 2;
  uint32_t sa = __builtin_amdgcn_readfirstlane(a), sb = sa + 1;
  uint32_t addr = a * 4;
  if (threadIdx.x >= 64) {
    // companion waves: 1 = spin on VALU, 2 = spin on s_sleep, for roughly the duration of the measurement
    if (companions == 1) for (uint32_t i = 0; i < n * 40; ++i) { asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1" : "+v"(a), "+v"(b)); }
    if (companions == 2) for (uint32_t i = 0; i < n * 4; ++i) { asm volatile("s_sleep 10"); }
    out[64 + threadIdx.x] = a + b;
    return;
  }
  if (MODE == 20) asm volatile("s_setprio 3");
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t i = 0; i < n; ++i) {
    if (MODE == 0 || MODE == 20) { REP32(asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));) }
    if (MODE == 1) { REP32(asm volatile("ds_read_b32 %0, %0\n" V4 "s_waitcnt lgkmcnt(0)" : "+v"(addr), "+v"(b));) }
    if (MODE == 2) { REP32(asm volatile("ds_read_b32 %0, %0\n" V4 V4 "s_waitcnt lgkmcnt(0)" : "+v"(addr), "+v"(b));) }
    if (MODE == 3) { REP32(asm volatile("ds_read_b32 %0, %0\n" V4 V4 V4 V4 "s_waitcnt lgkmcnt(0)" : "+v"(addr), "+v"(b));) }
    if (MODE == 4) { REP32(asm volatile("ds_read_b32 %0, %0\n" V4 V4 V4 V4 V4 V4 V4 V4 "s_waitcnt lgkmcnt(0)" : "+v"(addr), "+v"(b));) }
    if (MODE == 5) { REP32(asm volatile("v_add_u32 %0, %0, %0\n s_nop 0" : "+v"(a));) }
    if (MODE == 6) { REP32(asm volatile("v_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1" : "+v"(a), "+s"(sa) :: "scc");) }
    if (MODE == 7) { REP32(asm volatile("v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n v_sub_u32 %0, %0, %1\n v_lshrrev_b32 %0, 1, %0" : "+v"(a) : "v"(b));) }
    if (MODE == 8) { REP32(asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2" : "+v"(a), "+v"(b), "+v"(c));) }
    if (MODE == 9) { REP32(asm volatile("v_add3_u32 %0, %0, %0, %0" : "+v"(a));) }     // VOP3 dependent
    if (MODE == 10) { REP32(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a) : "v"(b), "v"(c));) }
    if (MODE == 11) { REP32(asm volatile("v_mov_b32 %0, %0" : "+v"(a));) }
    if (MODE == 12) { REP32(asm volatile("s_nop 0");) }
    if (MODE == 13) { REP32(asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3" : "+v"(a), "+v"(b), "+s"(sa), "+s"(sb) :: "scc");) }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[64 + threadIdx.x] = a + b + c + sa + sb + addr;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  uint32_t* out; uint64_t* t; (void)hipMalloc(&out, 4096); (void)hipMemset(out, 0, 4096); (void)hipMalloc(&t, 64);
  uint64_t ht; const uint32_t n = 20000;
#define RUN(M, name, per, thr, comp) k<M><<<1, thr>>>(out, n, t, comp); (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); printf("%-64s %.2f cycles\n", name, (double)ht / n / per);
  RUN(0, "32 dependent v_add, alone (per instr)", 32, 64, 0)
  RUN(0, "same, 3 companion waves spinning on VALU", 32, 256, 1)
  RUN(0, "same, 3 companion waves sleeping", 32, 256, 2)
  RUN(0, "same, 7 companion waves spinning on VALU (2 per SIMD)", 32, 512, 1)
  RUN(20, "same alone, s_setprio 3", 32, 64, 0)
  RUN(1, "ds_read + 4 dep v_add + wait (per group)", 32, 64, 0)
  RUN(2, "ds_read + 8 dep v_add + wait (per group)", 32, 64, 0)
  RUN(3, "ds_read + 16 dep v_add + wait (per group)", 32, 64, 0)
  RUN(4, "ds_read + 32 dep v_add + wait (per group)", 32, 64, 0)
  RUN(5, "v_add + s_nop 0 (per pair)", 32, 64, 0)
  RUN(6, "dep v_add + dep s_add interleaved (per pair)", 32, 64, 0)
  RUN(7, "dep chain add/xor/sub/lshr (per instr)", 128, 64, 0)
  RUN(8, "3 independent v_add chains (per instr)", 96, 64, 0)
  RUN(9, "dependent v_add3 (VOP3)", 32, 64, 0)
  RUN(10, "dependent v_bitop3", 32, 64, 0)
  RUN(11, "dependent v_mov", 32, 64, 0)
  RUN(12, "s_nop 0", 32, 64, 0)
  RUN(13, "2 v_add chains + 2 s_add chains interleaved (per instr)", 128, 64, 0)
  return 0;
}
