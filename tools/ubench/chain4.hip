// micro-benchmark: cycles per value of the scalar decoder chain (k_fpc32_decode.hip, v4) and of ablated forms of it, gfx950, one wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define SEL_BEFORE(K, D, LAST, SP) \
  "s_bitcmp1_b32 %[" D "], (" #K ") & 31\n" \
  "s_cselect_b32 %[lm], %[" LAST "], 0\n" \
  "s_cselect_b32 %[cand], %[" SP "], %[t1]\n" \
  "s_bitcmp1_b32 %[g], (" #K ") & 31\n"
#define CORE(K, LAST, V, S, AO, AN) \
  "s_waitcnt lgkmcnt(0)\n" \
  "s_cselect_b32 %[q], %[t2], %[cand]\n" \
  "s_add_u32 %[q], %[q], %[lm]\n" \
  "s_xor_b32 %[" V "], %[x], %[q]\n" \
  "s_sub_u32 %[" S "], %[" V "], %[" LAST "]\n" \
  "s_and_b32 %[h], %[" S "], 0xffc00000\n" \
  "s_xor_b32 %[q], %[h], %[P]\n" \
  "s_lshr_b32 %[" AN "], %[q], 20\n" \
  "s_load_dword %[t2], %[T2b], %[" AN "]\n"
#define STORE(S, AO) "s_store_dword %[" S "], %[T2b], %[" AO "]\n"
#define T1OPS(V) \
  "s_movreld_b32 s84, %[" V "]\n" \
  "s_lshr_b32 m0, %[" V "], 28\n" \
  "s_lshl_b32 %[P], %[h], 5\n" \
  "s_movrels_b32 %[t1], s84\n"
#define T1NONE(V) "s_lshl_b32 %[P], %[h], 5\n s_mov_b32 %[t1], %[" V "]\n"
#define GOPS(D, AO, AN) "s_cmp_lg_u32 %[" AN "], %[" AO "]\n s_cselect_b32 %[g], %[" D "], 0\n"
#define LANES(K, V) "v_writelane_b32 %[outv], %[" V "], " #K "\n v_readlane_b32 %[x], %[vx], ((" #K ") + 1) & 63\n"
#define LANES_W(K, V) "v_writelane_b32 %[outv], %[" V "], " #K "\n"
#define NOTHING ""

#define STEP0(K, D, LAST, V, SP, S, AO, AN) SEL_BEFORE(K, D, LAST, SP) CORE(K, LAST, V, S, AO, AN) STORE(S, AO) T1OPS(V) GOPS(D, AO, AN) LANES(K, V)
#define STEP1(K, D, LAST, V, SP, S, AO, AN) SEL_BEFORE(K, D, LAST, SP) CORE(K, LAST, V, S, AO, AN) STORE(S, AO) T1OPS(V) GOPS(D, AO, AN)
#define STEP2(K, D, LAST, V, SP, S, AO, AN) SEL_BEFORE(K, D, LAST, SP) CORE(K, LAST, V, S, AO, AN) STORE(S, AO) T1NONE(V) GOPS(D, AO, AN) LANES(K, V)
#define STEP3(K, D, LAST, V, SP, S, AO, AN) SEL_BEFORE(K, D, LAST, SP) CORE(K, LAST, V, S, AO, AN) T1OPS(V) GOPS(D, AO, AN) LANES(K, V)
#define STEP4(K, D, LAST, V, SP, S, AO, AN) SEL_BEFORE(K, D, LAST, SP) CORE(K, LAST, V, S, AO, AN) STORE(S, AO) T1OPS(V) LANES(K, V)
#define STEP5(K, D, LAST, V, SP, S, AO, AN) SEL_BEFORE(K, D, LAST, SP) CORE(K, LAST, V, S, AO, AN) STORE(S, AO) T1NONE(V)
#define STEP6(K, D, LAST, V, SP, S, AO, AN) SEL_BEFORE(K, D, LAST, SP) CORE(K, LAST, V, S, AO, AN) STORE(S, AO) T1OPS(V) GOPS(D, AO, AN) LANES_W(K, V)
#define STEP7(K, D, LAST, V, SP, S, AO, AN) CORE(K, LAST, V, S, AO, AN) STORE(S, AO) "s_lshl_b32 %[P], %[h], 5\n"

#define PAIR(ST, K0, K1, D) ST(K0, D, "va", "vb", "sa", "sb", "a2a", "a2b") ST(K1, D, "vb", "va", "sb", "sa", "a2b", "a2a")
#define OCT(ST, B, D) PAIR(ST, B + 0, B + 1, D) PAIR(ST, B + 2, B + 3, D) PAIR(ST, B + 4, B + 5, D) PAIR(ST, B + 6, B + 7, D)
#define BATCH(ST) OCT(ST, 0, "dlo") OCT(ST, 8, "dlo") OCT(ST, 16, "dlo") OCT(ST, 24, "dlo") OCT(ST, 32, "dhi") OCT(ST, 40, "dhi") OCT(ST, 48, "dhi") OCT(ST, 56, "dhi")

#define KERNEL(NAME, ST) \
__global__ void NAME(uint32_t* T2b, uint32_t* out, uint32_t n, uint64_t* t) { \
  uint32_t va = 1, sa = 2, a2a = 0, P = 0, t2 = 0, outv = 0, vb, sb, a2b, lm = 0, cand = 0, q, h = 0, x = 3, t1 = 5, g = 0; \
  uint32_t dlo = __builtin_amdgcn_readfirstlane(out[0]) | 0x5a5a5a5au, dhi = ~dlo, vx = threadIdx.x * 2654435761u; \
  for (uint32_t off = 0; off < 4096u; off += 16u) \
    asm volatile("s_mov_b64 s[40:41], 0\n s_mov_b64 s[42:43], 0\n s_store_dwordx4 s[40:43], %0, %1" :: "s"(T2b), "s"(off) : "s40", "s41", "s42", "s43", "memory"); \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  uint64_t t0 = __builtin_amdgcn_s_memtime(); \
  for (uint32_t i = 0; i < n; ++i) { \
    asm volatile("s_mov_b32 m0, 0\n" BATCH(ST) "s_waitcnt lgkmcnt(0)\n" \
      : [va] "+s"(va), [vb] "=&s"(vb), [sa] "+s"(sa), [sb] "=&s"(sb), [a2a] "+s"(a2a), [a2b] "=&s"(a2b), [P] "+s"(P), [t2] "+s"(t2), [outv] "+v"(outv), \
        [lm] "+s"(lm), [cand] "+s"(cand), [q] "=&s"(q), [h] "+s"(h), [x] "+s"(x), [t1] "+s"(t1), [g] "+s"(g) \
      : [T2b] "s"(T2b), [dlo] "s"(dlo), [dhi] "s"(dhi), [vx] "v"(vx) \
      : "scc", "memory", "m0", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"); \
    vx += outv; \
  } \
  uint64_t t1s = __builtin_amdgcn_s_memtime(); \
  asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory"); \
  out[64 + threadIdx.x] = va + sa + a2a + P + t2 + outv; \
  if (threadIdx.x == 0) t[0] = t1s - t0; \
}
KERNEL(k0, STEP0) KERNEL(k1, STEP1) KERNEL(k2, STEP2) KERNEL(k3, STEP3) KERNEL(k4, STEP4) KERNEL(k5, STEP5) KERNEL(k6, STEP6) KERNEL(k7, STEP7)
int main() {
  uint32_t *tab, *out; uint64_t* t; (void)hipMalloc(&tab, 8192); (void)hipMalloc(&out, 4096); (void)hipMalloc(&t, 64); (void)hipMemset(out, 0, 4096);
  uint64_t ht; const uint32_t n = 2000;
#define RUN(K, name) K<<<1, 64>>>(tab, out, n, t); (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); printf("%-72s %.1f cycles per value (%s)\n", name, (double)ht / n / 64, hipGetErrorString(hipDeviceSynchronize()));
  RUN(k0, "full step (22 instructions)")
  RUN(k1, "without v_writelane / v_readlane")
  RUN(k2, "without the FCM table in SGPRs (s_movreld, s_lshr m0, s_movrels)")
  RUN(k3, "without the s_store")
  RUN(k4, "without the forwarding flag (s_cmp_lg, s_cselect)")
  RUN(k5, "without lanes, FCM table")
  RUN(k6, "without v_readlane only")
  RUN(k7, "core only: wait, cselect, add, xor, sub, and, xor, lshr, load, store, lshl")
  return 0;
}
