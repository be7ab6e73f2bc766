// Does ds_wrxchg_rtn_b32 apply the ACTIVE lanes of one instruction that hit the same address in increasing lane order?
// If so, "read the table entry of my class, then write my payload there" of the float coder (fpsc.c:133-143: the reference does exactly
// that, value after value) is ONE LDS instruction for a whole step of 64 values: every lane gets what the nearest lower active lane of
// its key left (or what earlier steps left), and the entry ends up with the highest lane's payload.  32-bit entries, no tags.
// Checked against a ballot-built ground truth and a shadow table, with random keys (1 .. 1024 distinct), random exec masks, four waves
// per workgroup hammering the same LDS unit with their own tables, and byte stores of other waves in between.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) k_xchg(uint32_t rounds, uint32_t nkeys, uint32_t seed, unsigned long long* __restrict__ out)
  {
  __shared__ uint32_t T[4][1024];
  __shared__ uint32_t shadow[4][1024];
  __shared__ uint8_t noise[4][512];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  for (uint32_t i = lane; i < 1024; i += 64) { T[w][i] = 0u; shadow[w][i] = 0u; }
  __syncthreads();
  uint32_t x = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (threadIdx.x * 0x85EBCA6Bu) ^ 1u;
  unsigned long long wrong = 0, wrong_final = 0;
  const long long t0 = clock64();
  for (uint32_t r = 0; r < rounds; ++r)
    {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t k = (x >> 8) % nkeys;
    const bool active = ((x >> 3) & 7u) != 0u || (r & 15u) == 0u;        // 7/8 of the lanes, all of them every 16th round
    const uint32_t val = ((r + 1u) << 6) | lane;
    // ground truth: nearest lower active lane with my key, else the shadow table; the highest active lane of a key writes the shadow
    uint32_t expect = shadow[w][k];
    bool last = true;
    for (uint32_t j = 0; j < 64; ++j)
      {
      const uint32_t kj = __shfl(k, j, 64), vj = __shfl(val, j, 64);
      const bool aj = __shfl((int)active, j, 64) != 0;
      if (aj && kj == k) { if (j < lane) expect = vj; if (j > lane) last = false; }
      }
    uint32_t old = 0;
    if (active)
      old = __hip_atomic_exchange(&T[w][k], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    noise[w][(x >> 20) & 511u] = (uint8_t)x;                               // other LDS traffic of this wave between the atomics
    if (active && old != expect) ++wrong;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (active && last) shadow[w][k] = val;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (active && last && T[w][k] != val) ++wrong_final;
    }
  const long long t1 = clock64();
  atomicAdd(&out[0], wrong);
  atomicAdd(&out[1], wrong_final);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = (unsigned long long)(t1 - t0);
  }

int main()
  {
  unsigned long long* dout;
  if (hipMalloc(&dout, 32) != hipSuccess) return 1;
  const uint32_t rounds = 4000;
  const uint32_t nk[] = { 1, 2, 3, 8, 16, 40, 400, 1024 };
  for (uint32_t t = 0; t < 8; ++t)
    {
    (void)hipMemset(dout, 0, 32);
    hipLaunchKernelGGL(k_xchg, dim3(1024), dim3(256), 0, 0, rounds, nk[t], 0x1234567u + t, dout);
    unsigned long long h[4];
    if (hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("keys %4u: 4096 waves x %u ds_wrxchg_rtn_b32 (%.1f M instructions): wrong value returned %llu, wrong final entry %llu\n", nk[t], rounds,
           4096.0 * rounds / 1e6, h[0], h[1]);
    }
  return 0;
  }
