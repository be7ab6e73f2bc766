// In which order does the LDS unit apply the lanes of ONE ds_max_rtn_u64 instruction that hit the same address?
// If it is increasing lane order, `old` of lane i is what the nearest lower lane with the same key left there: the "latest earlier value of my class"
// lookup of the float encoder (k_fpc32_encode.hip, resolve(): ten ballots + bpermute + table read + table write) would be ONE LDS instruction.
// The order is not documented, so a kernel relying on it must verify it per step: in any other order some lane sees a tag >= its own.
// This program counts such violations over random key patterns (few keys .. all distinct) and times the instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

__global__ void k_order(const uint32_t* __restrict__ keys, uint32_t rounds, uint32_t nkeys, unsigned long long* __restrict__ out)
  {
  __shared__ unsigned long long T[1024];
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < 1024; i += 64) T[i] = 0ull;
  __syncthreads();
  unsigned long long violations = 0, wrong = 0;
  const long long t0 = clock64();
  for (uint32_t r = 0; r < rounds; ++r)
    {
    const uint32_t k = keys[(size_t)r * 64 + lane] % nkeys;
    const unsigned long long tag = ((unsigned long long)(r + 1) << 6) | lane;
    const unsigned long long val = (tag << 32) | (0x1000u * lane + r);
    const unsigned long long old = atomicMax(&T[k], val);
    const uint32_t otag = (uint32_t)(old >> 32);
    const bool same_step = (otag >> 6) == r + 1;
    if (same_step && (otag & 63u) >= lane)
      ++violations;
    // ground truth: nearest lower lane with my key (ballot based)
    uint32_t truth = 64;
    for (uint32_t j = 0; j < 64; ++j)
      {
      const uint32_t kj = __shfl(k, j, 64);
      if (j < lane && kj == k) truth = j;
      }
    if (same_step ? (otag & 63u) != truth : truth != 64u)
      ++wrong;
    }
  const long long t1 = clock64();
  atomicAdd(&out[0], violations);
  atomicAdd(&out[1], wrong);
  if (lane == 0 && blockIdx.x == 0) out[2] = (unsigned long long)(t1 - t0);
  }

int main()
  {
  const uint32_t rounds = 20000;
  uint32_t* hk = (uint32_t*)malloc(sizeof(uint32_t) * 64 * rounds);
  uint32_t x = 0x12345678u;
  for (size_t i = 0; i < (size_t)64 * rounds; ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; hk[i] = x >> 8; }
  uint32_t* dk; unsigned long long* dout;
  if (hipMalloc(&dk, sizeof(uint32_t) * 64 * rounds) != hipSuccess || hipMalloc(&dout, 32) != hipSuccess) return 1;
  (void)hipMemcpy(dk, hk, sizeof(uint32_t) * 64 * rounds, hipMemcpyHostToDevice);
  const uint32_t nk[] = { 1, 2, 8, 40, 400, 1024 };
  for (uint32_t t = 0; t < 6; ++t)
    {
    (void)hipMemset(dout, 0, 32);
    hipLaunchKernelGGL(k_order, dim3(64), dim3(64), 0, 0, dk, rounds, nk[t], dout);
    unsigned long long h[4];
    (void)hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost);
    printf("keys %4u: 64 waves x %u rounds: order violations %llu, wrong predecessor %llu, %.1f cycles per round (incl. the ballot check loop)\n", nk[t], rounds, h[0], h[1],
           (double)h[2] / rounds);
    }
  return 0;
  }
