// micro-benchmark 3: cost of the instruction kinds of the float decoder chain for ONE wave on gfx950.  Every group of 32 repetitions is
// ONE asm statement (hipcc pads each asm statement with an s_nop, which inflated lat.hip's single-instruction numbers by ~4.5 cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R32(x) R16(x) R16(x)
template <int MODE>
__global__ void k(uint32_t* out, uint32_t n, uint64_t* t) {
  __shared__ uint32_t tab[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = ((i * 7 + 1) & 1023) * 4;
  __syncthreads();
  uint32_t a = out[threadIdx.x & 63] & 1023, b = a + 1, c = a + 2, lane4 = 4096 + threadIdx.x * 4;
  uint32_t sa = __builtin_amdgcn_readfirstlane(a) + 5, sb = sa + 1;
  uint32_t addr = a * 4;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t i = 0; i < n; ++i) {
    if (MODE == 0) asm volatile(R32("v_add_u32 %0, %0, %0\n") : "+v"(a));
    if (MODE == 1) asm volatile(R32("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c\n") : "+v"(a) : "v"(b), "v"(c));
    if (MODE == 2) asm volatile(R32("v_and_b32 %0, 0xffc00000, %0\n") : "+v"(a));
    if (MODE == 3) asm volatile(R32("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(b) : "vcc");
    if (MODE == 4) asm volatile(R32("v_readlane_b32 %1, %0, 3\n v_xor_b32 %0, %1, %0\n") : "+v"(a), "+s"(sa));
    if (MODE == 5) asm volatile(R32("s_bfe_i32 %1, %2, 0x10003\n v_bitop3_b32 %0, %0, %3, %1 bitop3:0xe4\n") : "+v"(a), "+s"(sa) : "s"(sb), "v"(b) : "scc");
    if (MODE == 6) asm volatile(R32("ds_write_b32 %1, %0\n v_add_u32 %0, %0, %0\n") : "+v"(a) : "v"(addr));                         // uniform address, 64 lanes
    if (MODE == 7) asm volatile(R32("ds_write_b32 %1, %0\n v_add_u32 %0, %0, %0\n") : "+v"(a) : "v"(lane4));                        // per-lane addresses
    if (MODE == 8) asm volatile(R32("ds_read_b32 %2, %1\n v_add_u32 %0, %0, %0\n") "s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(addr), "=v"(c));   // reads not waited (issue cost)
    if (MODE == 9) asm volatile(R32("ds_bpermute_b32 %2, %1, %0\n v_add_u32 %0, %0, %0\n") "s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(addr), "=v"(c));
    if (MODE == 10) asm volatile(R32("v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n ds_write_b32 %1, %0\n") : "+v"(a) : "v"(addr));  // 1 write per 3 VALU
    if (MODE == 11) asm volatile(R32("v_cmp_gt_u32 vcc, %2, %0\n v_cndmask_b32 %1, %1, %0, vcc\n v_add_u32 %0, %0, %1\n") : "+v"(a), "+v"(b) : "s"(sa) : "vcc");
    if (MODE == 12) asm volatile(R32("v_lshrrev_b32 %0, 1, %0\n v_sub_u32 %0, %0, %1\n v_xor_b32 %0, %2, %0\n") : "+v"(a) : "v"(b), "s"(sa));
    if (MODE == 13) asm volatile(R32("s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %0\n") : "+v"(a));
    if (MODE == 14) asm volatile(R32("ds_write_b32 %1, %0 offset:64\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(addr));  // write one address, read another, wait
    if (MODE == 15) asm volatile(R32("ds_read_b32 %0, %1\n ds_write_b32 %1, %2 offset:64\n s_waitcnt lgkmcnt(1)\n") : "+v"(a) : "v"(addr), "v"(b)); // read, then write, wait for the read only
    if (MODE == 16) asm volatile(R32("ds_write_b128 %1, %0\n v_add_u32 %2, %2, %2\n") : "+v"(*(uint4*)tab) , "+v"(addr), "+v"(a));
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[64 + threadIdx.x] = a + b + c + sa + sb + addr;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  uint32_t* out; uint64_t* t; (void)hipMalloc(&out, 4096); (void)hipMemset(out, 0, 4096); (void)hipMalloc(&t, 64);
  uint64_t ht; const uint32_t n = 20000;
#define RUN(M, name, per) k<M><<<1, 64>>>(out, n, t); (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); printf("%-72s %.2f cycles\n", name, (double)ht / n / per);
  RUN(0, "dependent v_add_u32", 32)
  RUN(1, "dependent v_bitop3_b32 (VOP3)", 32)
  RUN(2, "dependent v_and_b32 with 32-bit literal", 32)
  RUN(3, "v_cmp_eq -> vcc -> v_cndmask, dependent (per pair)", 32)
  RUN(4, "v_readlane -> sgpr -> v_xor, dependent (per pair)", 32)
  RUN(5, "s_bfe_i32 -> sgpr -> v_bitop3 (per pair)", 32)
  RUN(6, "ds_write_b32 uniform address + v_add (per pair)", 32)
  RUN(7, "ds_write_b32 per-lane addresses + v_add (per pair)", 32)
  RUN(8, "ds_read_b32 uniform, not waited + v_add (per pair)", 32)
  RUN(9, "ds_bpermute, not waited + v_add (per pair)", 32)
  RUN(10, "3 v_add + ds_write uniform (per group of 4)", 32)
  RUN(11, "v_cmp_gt(sgpr) -> v_cndmask -> v_add (per triple)", 32)
  RUN(12, "v_lshrrev, v_sub, v_xor(sgpr) dependent (per triple)", 32)
  RUN(13, "s_waitcnt lgkmcnt(0) (nothing outstanding) + v_add (per pair)", 32)
  RUN(14, "ds_write A, ds_read B, wait (per group)", 32)
  RUN(15, "ds_read B, ds_write A, wait for the read only (per group)", 32)
  return 0;
}
