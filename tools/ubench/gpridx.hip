// micro-test: does VGPR index mode (s_set_gpr_idx_on ... DST / SRC0) apply to v_writelane_b32 / v_readlane_b32 on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__global__ void k(uint32_t* out, uint32_t idx, uint32_t lanesel, uint32_t val)
  {
  u32x16 t;
  for (int j = 0; j < 16; ++j) t[j] = 1000u * j + threadIdx.x;
  const uint32_t sidx = __builtin_amdgcn_readfirstlane(idx), sl = __builtin_amdgcn_readfirstlane(lanesel), sv = __builtin_amdgcn_readfirstlane(val);
  uint32_t rd = 0;
  asm volatile("s_mov_b32 m0, %3\n\t"
               "s_set_gpr_idx_on %2, gpr_idx(DST)\n\t"
               "v_writelane_b32 v20, %4, m0\n\t"
               "s_set_gpr_idx_off\n\t"
               "s_set_gpr_idx_on %2, gpr_idx(SRC0)\n\t"
               "v_readlane_b32 %1, v20, %3\n\t"
               "s_set_gpr_idx_off\n\t"
               : "+{v[20:35]}"(t), "=s"(rd) : "s"(sidx), "s"(sl), "s"(sv) : "m0");
  for (int j = 0; j < 16; ++j) out[j * 64 + threadIdx.x] = t[j];
  if (threadIdx.x == 0) out[1024] = rd;
  }
int main()
  {
  uint32_t* d; (void)hipMalloc(&d, 4200); (void)hipMemset(d, 0, 4200);
  k<<<1, 64>>>(d, 5, 17, 777777);
  uint32_t h[1025]; (void)hipMemcpy(h, d, 4100, hipMemcpyDeviceToHost);
  printf("reg0 lane17 = %u (untouched would be 17)\n", h[0 * 64 + 17]);
  printf("reg5 lane17 = %u (index mode applied: 777777; else 5017)\n", h[5 * 64 + 17]);
  printf("readlane with SRC0 index = %u (index mode applied: 777777; reg0: %u)\n", h[1024], h[17]);
  return 0;
  }
