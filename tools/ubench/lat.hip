// micro-benchmark: per-instruction cost of ONE wave on gfx950 for the instruction mixes a serial decoder chain can be built from
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
template <int MODE>
__global__ void k(uint32_t* out, uint32_t n, uint64_t* t) {
  __shared__ uint32_t tab[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) tab[i] = ((i * 7 + 1) & 1023) * 4;   // byte offsets, a permutation
  __syncthreads();
  uint32_t a = out[threadIdx.x] & 1023, b = a + 1, c = a + 2, d = a + 3;
  uint32_t sa = __builtin_amdgcn_readfirstlane(a), sb = sa + 1;
  uint32_t base = (uint32_t)(uintptr_t)tab;
  uint32_t addr = base + a * 4;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t i = 0; i < n; ++i) {
    if (MODE == 0) { REP32(asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));) }                        // dependent VALU
    if (MODE == 1) { REP32(asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1" : "+v"(a), "+v"(b));) }   // 2 independent chains
    if (MODE == 2) { REP32(asm volatile("s_add_u32 %0, %0, %0" : "+s"(sa) :: "scc");) }              // dependent SALU
    if (MODE == 3) { REP32(asm volatile("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1" : "+s"(sa), "+s"(sb) :: "scc");) }
    if (MODE == 4) { REP32(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr));) }   // dependent LDS read, nothing else
    if (MODE == 5) { REP32(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(addr) : "v"(b));) }   // + 2 dependent VALU
    if (MODE == 6) { REP32(asm volatile("ds_read_b32 %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n s_waitcnt lgkmcnt(0)" : "+v"(addr), "+v"(b));) }   // LDS read with 8 VALU underneath
    if (MODE == 7) { REP32(asm volatile("ds_write_b32 %0, %1\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr) : "v"(b));) }   // write + dependent read
    if (MODE == 8) { REP32(asm volatile("v_readlane_b32 %0, %1, 3\n v_xor_b32 %1, %0, %1" : "+s"(sa), "+v"(a));) }   // VALU -> SGPR -> VALU
    if (MODE == 9) { REP32(asm volatile("v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n v_mov_b32 %1, %0" : "+s"(sa), "+v"(a) :: "scc");) }   // VALU -> SALU -> VALU
    if (MODE == 10) { REP32(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(c));) }  // dependent bpermute
    if (MODE == 11) { REP32(asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:" : "+s"(sa) :: "scc");) }   // compare + not-taken branch + add
    if (MODE == 12) { REP32(asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n s_add_u32 %0, %0, 1" : "+s"(sa) :: "scc");) }   // taken branch
    if (MODE == 13) { REP32(asm volatile("v_add_u32 %0, %0, %0\n s_nop 0\n s_nop 0\n s_nop 0" : "+v"(a));) }        // dependent VALU with 3 nops between
    if (MODE == 14) { REP32(asm volatile("v_cmp_ne_u32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, %0\n1:" : "+v"(a) : "v"(b) : "vcc");) }   // VALU compare -> branch
    if (MODE == 16) { REP32(asm volatile("ds_write_b32 %2, %1 offset:4160\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr) : "v"(b), "v"(base));) }   // write elsewhere + dependent read
    if (MODE == 17) { REP32(asm volatile("ds_read_b32 %0, %0\n ds_write_b32 %2, %1 offset:4160\n s_waitcnt lgkmcnt(1)" : "+v"(addr) : "v"(b), "v"(base));) }   // dependent read, then a write that is not waited for
    if (MODE == 18) { REP32(asm volatile("ds_write_b32 %0, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr), "+v"(b));) }   // write, 8 VALU, dependent read
    if (MODE == 19) { REP32(asm volatile("ds_read_b32 %0, %0\n ds_write_b32 %0, %1 offset:8\n s_waitcnt lgkmcnt(1)" : "+v"(addr) : "v"(b));) }   // read then write through the (old) same address register
    if (MODE == 20) { REP32(asm volatile("s_load_dwordx16 s[36:51], %1, 0x0\n s_waitcnt lgkmcnt(0)\n s_add_u32 %0, %0, s36" : "+s"(sa) : "s"(out) : "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","scc");) }   // scalar load latency
    if (MODE == 21) { REP32(asm volatile("s_load_dwordx16 s[36:51], %1, 0x0\n v_add_u32 %2, %2, %2\n v_add_u32 %2, %2, %2\n v_add_u32 %2, %2, %2\n v_add_u32 %2, %2, %2\n v_add_u32 %2, %2, %2\n v_add_u32 %2, %2, %2\n v_add_u32 %2, %2, %2\n v_add_u32 %2, %2, %2" : "+s"(sa) : "s"(out), "v"(b) : "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","scc");) }   // scalar load issued under 8 VALU, never waited (issue cost)
    if (MODE == 22) { REP32(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:4\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr) : "v"(b));) }   // two writes + dependent read
    if (MODE == 23) { REP32(asm volatile("ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_lshrrev_b32 %0, 26, %0" : "+v"(a) : "v"(c));) }  // bpermute + 1 VALU
    if (MODE == 15) { REP32(asm volatile("ds_read_b32 %0, %0\n ds_read_b32 %1, %2\n s_waitcnt lgkmcnt(0)" : "+v"(addr), "=v"(c) : "v"(base));) }   // 2 LDS reads in flight, one dependent
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[64 + threadIdx.x] = a + b + c + d + sa + sb + addr;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  uint32_t* out; uint64_t* t; (void)hipMalloc(&out, 1024); (void)hipMemset(out, 0, 1024); (void)hipMalloc(&t, 64);
  uint64_t ht; const uint32_t n = 20000;
  const char* names[24] = {"dependent v_add", "2 independent v_add chains (per instr)", "dependent s_add", "2 independent s_add chains (per instr)",
    "dependent ds_read_b32 (per read)", "ds_read + 2 dependent v_xor (per group)", "ds_read + 8 v_add underneath (per group)", "ds_write + dependent ds_read (per group)",
    "v_readlane -> v_xor (per pair)", "readfirstlane -> s_add -> v_mov (per triple)", "dependent ds_bpermute", "s_cmp + branch not taken + s_add (per group)",
    "s_cmp + branch taken + s_add (per group)", "v_add + 3 s_nop (per group)", "v_cmp + s_cbranch_vccz + v_add (per group)", "2 ds_read in flight (per group)",
    "ds_write elsewhere + dependent ds_read", "dependent ds_read, then ds_write not waited for", "ds_write, 8 v_add, dependent ds_read", "ds_read then ds_write (same reg) not waited",
    "s_load_dwordx16 + wait + s_add (latency)", "s_load_dwordx16 under 8 v_add, no wait (per group)", "2 ds_write + dependent ds_read", "ds_bpermute + v_lshrrev (per group)"};
  const int per[24] = {32, 64, 32, 64, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32};
#define RUN(M) k<M><<<1, 64>>>(out, n, t); (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); printf("%-52s %.2f cycles\n", names[M], (double)ht / n / per[M]);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23)
  return 0;
}
