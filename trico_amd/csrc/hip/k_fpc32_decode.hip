// k_fpc32_decode.hip — decoder for 32-bit floating-point streams: one workgroup of two waves per component stream.
//
// Replaces trico_decompress (fpsc.c:212-417) + trico_transpose_*_soa_to_aos (transpose_aos_to_soa.c:18-26,
// 58-66): the decoded component is written straight into its slot of the interleaved output.
//
// The format leaves no parallelism inside a stream: value i is xor_i ^ prediction_i, and the table keys for
// prediction_{i+1} come from the decoded value i (fpsc.c:308-326).  What bounds the kernel is therefore the number of
// cycles ONE wave needs per value, and the kernel is organised around what one wave of gfx950 pays per instruction
// (measured: tools/ubench/lat3.hip, smem.hip, smem2.hip, chain4.hip):
//     any ALU instruction, scalar or vector, dependent or not        4 cycles
//     an LDS instruction                                             13-17 cycles of issue, ~48 of latency; a read of an
//                                                                    address whose write is in flight ~120
//     a scalar load / store                                          5-6 cycles of issue, ~37 of latency (scalar-cache hit)
//     a branch, taken or not                                         ~26 cycles
//     v_readlane inside a scalar instruction stream                  ~21 cycles
//   * wave 1 (parser) stages the payload through LDS, finds the 8 groups of the next 64 values with a short scalar walk
//     over the 3-byte headers, lets all 64 lanes locate, align and byte-swap their residual at once, and hands the
//     residuals, grouped in quads with the address of the code for their kinds, to wave 0 through a ring of four batches in
//     scalar memory.
//   * wave 0 (chain) runs the recurrence on the SCALAR unit, in straight-line bodies of four values chosen by the kinds the
//     parser found, with the DFCM table (1024 entries) in global memory behind the scalar data cache and the FCM table (16
//     entries) in SGPRs: see below.  It is one asm statement from the first value to the last: it polls the parser's counter
//     with a scalar load, stores the 64 values of a batch with four vector stores of 16 lanes straight into the interleaved
//     output, and publishes its own counter with a scalar store — no LDS instruction, no barrier and no v_readlane on its path.
// History: one wave, scalar chain with branches and both tables in registers: 165-225 cycles per value (69-93 ns); vector
// chain with the tables in LDS: 146; scalar chain, one barrier per batch: 38-43 ns; LDS counters instead of the barrier and
// the parser storing the values: 37-41 ns (~400 cycles of handshake, table save / restore and LDS traffic per batch); this
// design with one branch-free body for every value (round 3): 35-37 ns per value whatever the stream; bodies per pattern
// of kinds (round 4): 30.4 ns on the noisiest stream of the benchmark mesh.
// Streams with table exponents below the (4,10) the archive API writes, and stream tails (< 64 values), take the
// reference-order loop of one lane at the end of the file.
//
// Latency-bound by construction; algorithmic bytes per value: its payload share read + 4 written.
#include "common.hpp"

// The scalar chain names M0 and fixed SGPRs in its clobber lists; the compiler only ever sets M0 right before its own uses.
#pragma clang diagnostic ignored "-Winline-asm"

namespace trico {

namespace {

constexpr int WINW = 512;                  // staging window, dwords (2 KiB): a refill must fit into the slack of the ring (8 and 32 KiB windows cost the noisy stream 2-4 ns per value)
constexpr uint32_t BATCH_BYTES = 8 * 35;   // a batch of 8 groups needs at most this many payload bytes

// (the descriptor of one chain, Fpc32ChainJob, is declared in common.hpp: the batch engine of shim.hip fills tables of them)
struct ChainJobs3 { Fpc32ChainJob j[3]; };     // the single-stream launch passes its (at most three) chains by value

__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t rfl64(uint64_t x) { return ((uint64_t)rfl((uint32_t)(x >> 32)) << 32) | rfl((uint32_t)x); }

// sum of the residual lengths of the 3-bit codes packed in x (codes 0..4 -> 0..4 bytes, 5..7 -> 1..3 bytes)
__device__ __forceinline__ uint32_t lens_sum(uint32_t x)
  {
  const uint32_t b0 = x & 0x249249u, b1 = (x >> 1) & 0x249249u, b2 = (x >> 2) & 0x249249u;
  const uint32_t hi = b2 & (b1 | b0);                            // codes 5, 6, 7
  return (uint32_t)__popc(b0) + 2u * (uint32_t)__popc(b1) + 4u * ((uint32_t)__popc(b2) - (uint32_t)__popc(hi));
  }

// ---- the chain (wave 0): scalar unit, DFCM table behind the scalar data cache ------------------------------------------
// gfx950 still executes scalar STORES (s_store_dword), a
// scalar load issued after a scalar store to the same address returns the stored value, the two low address bits are
// ignored, and dirty lines survive other kernels being dispatched (smem2.hip: 24 chains x 8M operations against the host
// while 10^5 other kernels were launched).  So the tables go where the cheap instructions can reach them: a 4 KiB + 64 B
// scratch per stream in global memory that only this wave touches, zeroed here with scalar stores and written back
// (s_dcache_wb) before the kernel ends so that no dirty line outlives the buffer.
// Which of the two predictions a value takes is in the headers, so the parser knows it before the chain gets there: it
// hands the values over in quads, each with the address of straight-line code for exactly its pattern of kinds (F = FCM-coded,
// D = DFCM-coded) and the kind of the value behind it - 2 x 32 bodies of 512 bytes, generated by tools/gen_chain5.py
// (chain5_bodies.inc, registers and layout documented there).  Per value, all on the scalar unit (a2 = byte address of the
// current DFCM entry, P = (stride & 0xffc00000) << 5 of the previous value: the hash of the strides lives in the top ten
// bits, where the shift by five drops the old bits; M0 = top four bits of the previous value):
//     D: wait; q = forwarded ? previous stride : loaded T2 entry; v = x ^ (q + last)     s_waitcnt, s_cselect, s_add, s_xor
//     F: v = x ^ T1 entry                                                                 s_xor               (fpsc.c:308-311)
//     s = v - last; T2[a2] = s; a2' = ((s & 0xffc00000) ^ P) >> 20                        s_sub, s_store, s_and, s_xor, s_lshr   (fpsc.c:81-84, 323-326)
//     T1[M0] = v; M0 = v >> 28; P = (s & 0xffc00000) << 5                                 s_movreld, s_lshr, s_lshl              (fpsc.c:76-79, 312-314)
//     next is D: load T2[a2']; forwarded = (a2' == a2)      next is F: T1 entry = T1[M0]  s_load, s_cmp_lg  /  s_movrels
//     lane of the quad in output register j = v                                           v_mov (EXEC = that lane)
// 11 instructions for an F value before an F value (no wait: 44 cycles), 15 for a D value before a D value, whose critical
// path (wait ... load) is the table load's latency + 9 instructions, ~80 cycles; per quad a record load, the EXEC move and
// the jump (s_setpc_b64 to the address in the next record), ~35 cycles.  Round 3's chain ran the same 20 branch-free
// instructions for every value (85 cycles, 35.5 ns on the benchmark mesh's z: half its values D, kinds switching at random);
// this one 30.4 ns there.  Measured and dropped (profiles/r04_chain_variants.txt): bodies that do not know the kind behind the
// quad (30.7); the values of a quad leaving through one s_store_dwordx4 into a ring the parser wave empties (34.9: the waits
// cover the wide store, and the parser's loads share the scalar cache with the chain's).
// A scalar load issued after a scalar store to the same address is NOT reliably ordered behind it when the line misses
// (smem2.hip under cache pressure; a parity test caught it too), so when the address of the load equals the address of the
// store just before it the loaded word is ignored and the next value takes the stride from the register instead (SCC carries
// that from s_cmp_lg to the next value's s_cselect, also across the jump).  Every OLDER store must be complete before a
// load is issued, and only lgkmcnt(0) means anything for scalar memory: every value that loads, or needs a loaded entry,
// begins with s_waitcnt lgkmcnt(0); only an F value before an F value runs without one.
// v_readlane costs ~21 cycles in a scalar instruction stream (tools/ubench/chain4.hip), so the residuals do not come from
// a VGPR: the parser wave puts the records into global memory with scalar stores (same scalar cache, same CU) and every
// body loads the record of the next quad with s_load_dwordx8 before it starts on its own.
#include "chain5_bodies.inc"

// Scratch of one stream in global memory (FPC32_DECODE_TABLE_BYTES), touched by this workgroup only and only through the
// scalar cache: DFCM table (4 KiB), FCM table for the tail (64 B), at SCRATCH_X RING slots of 17 records of 32 bytes (16 quads
// and the record that ends the batch), then the two counters that couple the waves and the address of the chain's code, each in
// a cache line of its own.
constexpr uint32_t SCRATCH_DWORDS = 2048, SCRATCH_T1 = 1024, SCRATCH_X = 1088;
constexpr uint32_t RING = 4;                  // batches the parser may run ahead of the chain
constexpr uint32_t REC_DWORDS = 8, SLOT_DWORDS = 18 * REC_DWORDS;
constexpr uint32_t SCRATCH_PRODUCED = SCRATCH_X + RING * SLOT_DWORDS, SCRATCH_CONSUMED = SCRATCH_PRODUCED + 16, SCRATCH_CODE = SCRATCH_CONSUMED + 16;
constexpr uint32_t SCRATCH_USED = SCRATCH_CODE + 16;              // dwords zeroed at the start (a multiple of 16)
static_assert(SCRATCH_X * 4 == 0x1100 && SLOT_DWORDS * 4 == 0x240 && SCRATCH_PRODUCED * 4 == 0x1a00 && SCRATCH_CONSUMED * 4 == 0x1a40 &&
              SCRATCH_CODE * 4 == 0x1a80 && SCRATCH_USED <= SCRATCH_DWORDS, "offsets are spelled out in the chain");
constexpr uint32_t ABORT = 0xffffffffu;       // `produced` when the parser gives up: the chain stops

struct ChainEnd { uint32_t last, a2, timed_out; };   // what the tail loop needs: last value, byte address of the current DFCM entry

// The two waves wait for each other by polling; neither waits forever.  A poll of the chain costs ~100 cycles (s_sleep 1 + a
// scalar-cache hit), one of the parser ~300: both limits are seconds of no progress at all, which a healthy pair never sees
// (the other wave publishes every few microseconds).  What it guards against: counters that went stale because the workgroup
// was saved and restored on another compute unit (shim.hip, "the chain decoders and their self-check") - the kernel then
// ends with status FPC_STATUS_TIMEOUT instead of spinning for ever, and the host repeats the stream.
constexpr uint32_t SPIN_LIMIT_CHAIN = 1u << 25, SPIN_LIMIT_PARSER = 1u << 23;

// The whole chain of a stream: batches 0 .. nb-1 of 64 values, each 16 quads.  The parser writes a record per quad: its four
// residuals, the address of the body that decodes its pattern of kinds (tools/gen_chain5.py), the bit of the lane its values go to and where
// the next record is.  A body requests the next record first, decodes, and jumps to the address in it; the record behind the
// sixteenth quad leads to the end of the batch: four vector stores of 16 lanes straight into the interleaved output (lane q
// of register j holds value 4 q + j), `consumed` published, `produced` polled, the first record of the next batch loaded.
// One asm statement, so that the FCM table and the rest of the state stay in their registers from the first value to the last.
//   s46 the SCC between values, kept across the end of a batch    s76 scratch    s77 batch counter    s[80:81] output address of the batch's value 0
__device__ __forceinline__ ChainEnd chain5_run(const uint32_t* T2b, uint32_t nb, uint32_t* out0, uint32_t va0, uint32_t va1, uint32_t va2,
                                               uint32_t va3, uint32_t out_step, uint32_t lane4m1)
  {
  const uint64_t tb = (uint64_t)(uintptr_t)T2b, ob = (uint64_t)(uintptr_t)out0;
  const uint32_t tlo = (uint32_t)tb, thi = (uint32_t)(tb >> 32), olo = (uint32_t)ob, ohi = (uint32_t)(ob >> 32);
  uint32_t last, a2, o0, o1, o2, o3, vt, spin, tmo;
  asm volatile(
    "s_mov_b64 s[84:85], 0\n s_mov_b64 s[86:87], 0\n s_mov_b64 s[88:89], 0\n s_mov_b64 s[90:91], 0\n"
    "s_mov_b64 s[92:93], 0\n s_mov_b64 s[94:95], 0\n s_mov_b64 s[96:97], 0\n s_mov_b64 s[98:99], 0\n"
    "s_mov_b64 s[52:53], 0\n s_mov_b64 s[54:55], 0\n s_mov_b32 s56, 0\n s_mov_b32 m0, 0\n"
    "s_mov_b32 s42, 0\n s_mov_b32 s43, 0\n s_mov_b32 s44, 0\n s_mov_b32 s45, 0\n s_mov_b32 s46, 1\n"
    "s_mov_b32 s36, %[tlo]\n s_mov_b32 s37, %[thi]\n"
    "s_mov_b32 s77, 0\n s_mov_b32 s80, %[olo]\n s_mov_b32 s81, %[ohi]\n"
    "s_mov_b32 %[spin], 0\n s_mov_b32 %[tmo], 0\n"
    /* where the bodies are: the parser puts their addresses into the records */
    "s_getpc_b64 s[38:39]\n"
    ".Lc5_pc_%=:\n"
    "s_add_u32 s38, s38, .Lc5_body_%= - .Lc5_pc_%=\n"
    "s_addc_u32 s39, s39, 0\n"
    "s_store_dwordx2 s[38:39], s[36:37], 0x1a80\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_mov_b32 s40, 1\n"
    "s_store_dword s40, s[36:37], 0x1a88\n"
    "s_cmp_lt_u32 s77, %[nb]\n"
    "s_cbranch_scc0 .Lc5_done_%=\n"
    "s_branch .Lc5_poll_%=\n"
    CH5_BODIES
    ".Lc5_end_%=:\n"
    /* (the last quad of a batch does not know what follows it: the DFCM entry of the next value is requested here) */
    "s_waitcnt lgkmcnt(0)\n"
    "s_load_dword s43, s[36:37], s44\n"
    "s_cmp_lg_u32 s44, s45\n"
    "s_cselect_b32 s46, 1, 0\n"
    "s_mov_b64 exec, 0xffff\n"
    "global_store_dword %[va0], %[o0], s[80:81]\n"
    "global_store_dword %[va1], %[o1], s[80:81]\n"
    "global_store_dword %[va2], %[o2], s[80:81]\n"
    "global_store_dword %[va3], %[o3], s[80:81]\n"
    "s_add_u32 s80, s80, %[ostep]\n"
    "s_addc_u32 s81, s81, 0\n"
    "s_add_u32 s77, s77, 1\n"
    "s_store_dword s77, s[36:37], 0x1a40\n"
    "s_cmp_lt_u32 s77, %[nb]\n"
    "s_cbranch_scc0 .Lc5_done_%=\n"
    ".Lc5_poll_%=:\n"
    "s_load_dword s76, s[36:37], 0x1a00\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_cmp_gt_u32 s76, s77\n"
    "s_cbranch_scc1 .Lc5_go_%=\n"
    /* bounded wait: a parser that does not publish for SPIN_LIMIT polls (seconds) will not publish at all */
    "s_add_u32 %[spin], %[spin], 1\n"
    "s_cmp_lt_u32 %[spin], %[limit]\n"
    "s_cbranch_scc0 .Lc5_tmo_%=\n"
    "s_sleep 1\n"
    "s_branch .Lc5_poll_%=\n"
    ".Lc5_tmo_%=:\n"
    "s_mov_b32 %[tmo], 1\n"
    "s_mov_b32 s76, -1\n"
    "s_store_dword s76, s[36:37], 0x1a40\n"            /* consumed = ABORT: the parser stops too */
    "s_branch .Lc5_done_%=\n"
    ".Lc5_go_%=:\n"
    "s_mov_b32 %[spin], 0\n"
    "s_cmp_eq_u32 s76, -1\n"
    "s_cbranch_scc1 .Lc5_done_%=\n"
    "s_and_b32 s76, s77, 3\n"
    "s_mul_i32 s76, s76, 0x240\n"
    "s_add_u32 s76, s76, 0x1100\n"
    "s_load_dwordx8 s[60:67], s[36:37], s76\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_cmp_lg_u32 s46, 0\n"
    "s_setpc_b64 s[64:65]\n"
    ".Lc5_done_%=:\n"
    "s_mov_b64 exec, -1\n"
    "s_mov_b32 %[last], s53\n"
    "s_mov_b32 %[a2], s44\n"
    "s_store_dwordx4 s[84:87], s[36:37], 0x1000\n"
    "s_store_dwordx4 s[88:91], s[36:37], 0x1010\n"
    "s_store_dwordx4 s[92:95], s[36:37], 0x1020\n"
    "s_store_dwordx4 s[96:99], s[36:37], 0x1030\n"
    "s_waitcnt lgkmcnt(0)\n"
    : [last] "=&s"(last), [a2] "=&s"(a2), [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [vt] "=&v"(vt), [spin] "=&s"(spin),
      [tmo] "=&s"(tmo)
    : [tlo] "s"(tlo), [thi] "s"(thi), [olo] "s"(olo), [ohi] "s"(ohi), [va0] "v"(va0), [va1] "v"(va1), [va2] "v"(va2), [va3] "v"(va3),
      [ostep] "s"(out_step), [nb] "s"(nb), [lm] "v"(lane4m1), [limit] "s"(SPIN_LIMIT_CHAIN)
    : "scc", "memory", "m0", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s52", "s53", "s54", "s55",
      "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
      "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87",
      "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
  return ChainEnd{ last, a2, tmo };
  }

// parser wave: the 16 records of a batch go to the ring slot with scalar stores, through the scalar cache the chain reads; complete
// on return.  xr = the residuals (lane K = value K); lane q of tlo / thi = address of quad q's body; off = byte offset of the
// slot from the scratch base.
__device__ __forceinline__ void chain5_put_batch(uint32_t xr, uint32_t tlo, uint32_t thi, const uint32_t* slot, uint32_t off)
  {
#define CH5_PUT(Q, R0, R1, R2, R3, R4, R5, R6, R7) \
  "v_readlane_b32 s" #R0 ", %[xr], 4 * (" #Q ")\n v_readlane_b32 s" #R1 ", %[xr], 4 * (" #Q ") + 1\n" \
  "v_readlane_b32 s" #R2 ", %[xr], 4 * (" #Q ") + 2\n v_readlane_b32 s" #R3 ", %[xr], 4 * (" #Q ") + 3\n" \
  "v_readlane_b32 s" #R4 ", %[tlo], " #Q "\n v_readlane_b32 s" #R5 ", %[thi], " #Q "\n" \
  "s_mov_b32 s" #R6 ", 1 << (" #Q ")\n s_add_u32 s" #R7 ", %[off], 32 * ((" #Q ") + 1)\n" \
  "s_store_dwordx4 s[" #R0 ":" #R3 "], %[slot], 32 * (" #Q ")\n s_store_dwordx4 s[" #R4 ":" #R7 "], %[slot], 32 * (" #Q ") + 16\n"
#define CH5_PUT_E(Q) CH5_PUT(Q, 52, 53, 54, 55, 56, 57, 58, 59)
#define CH5_PUT_O(Q) CH5_PUT(Q, 60, 61, 62, 63, 64, 65, 66, 67)
  asm volatile(
    CH5_PUT_E(0) CH5_PUT_O(1) CH5_PUT_E(2) CH5_PUT_O(3) CH5_PUT_E(4) CH5_PUT_O(5) CH5_PUT_E(6) CH5_PUT_O(7)
    CH5_PUT_E(8) CH5_PUT_O(9) CH5_PUT_E(10) CH5_PUT_O(11) CH5_PUT_E(12) CH5_PUT_O(13) CH5_PUT_E(14) CH5_PUT_O(15)
    "s_waitcnt lgkmcnt(0)\n"
    :: [xr] "v"(xr), [tlo] "v"(tlo), [thi] "v"(thi), [slot] "s"(slot), [off] "s"(off)
    : "scc", "memory", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
  }

// the record that ends a batch: only the address in it counts
__device__ __forceinline__ void chain5_put_end(uint64_t target, const uint32_t* slot)
  {
  asm volatile("s_store_dwordx2 %[t], %[slot], 32 * 16 + 16\n s_waitcnt lgkmcnt(0)\n" :: [t] "s"(target), [slot] "s"(slot) : "memory");
  }

// the counters that couple the two waves live in the scalar cache both of them go through
__device__ __forceinline__ uint32_t counter_load(const uint32_t* base, uint32_t dword)
  {
  uint32_t r;
  asm volatile("s_load_dword %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(base), "s"(4u * dword) : "memory");
  return r;
  }
__device__ __forceinline__ void counter_store(const uint32_t* base, uint32_t dword, uint32_t v)
  {
  asm volatile("s_store_dword %0, %1, %2\n s_waitcnt lgkmcnt(0)" :: "s"(v), "s"(base), "s"(4u * dword) : "memory");
  }

// reference-order loop of one lane over LDS tables: stream tails (< 64 values) and table exponents below the API's (4,10)
__device__ void serial_values(const uint8_t* __restrict__ in, uint32_t len, uint32_t& pos, uint32_t i0, uint32_t n, uint32_t e1,
                              uint32_t e2, uint32_t& h1, uint32_t& h2, uint32_t& last, uint32_t* T1, uint32_t* T2,
                              uint32_t* __restrict__ dst, int arity, int comp, bool& bad)
  {
  const uint32_t m1 = (1u << e1) - 1u, m2 = (1u << e2) - 1u;
  for (uint32_t i = i0; i < n; i += 8u)
    {
    if (pos + 3u > len) { bad = true; return; }
    const uint32_t bc = ((uint32_t)in[pos] << 16) | ((uint32_t)in[pos + 1] << 8) | in[pos + 2];
    pos += 3u;
    const uint32_t m = (n - i < 8u) ? (n - i) : 8u;
    for (uint32_t k = 0; k < m; ++k)
      {
      const uint32_t code = (bc >> (3u * k)) & 7u;
      const uint32_t nb = code <= 4u ? code : code - 4u;
      if (pos + nb > len) { bad = true; return; }
      uint32_t x = 0;
      for (uint32_t b = 0; b < nb; ++b) x = (x << 8) | in[pos++];
      const uint32_t p = code > 4u ? last + T2[h2] : T1[h1];
      const uint32_t v = x ^ p;
      T1[h1] = v;
      h1 = ((h1 << e1) ^ (v >> (32u - e1))) & m1;
      const uint32_t s = v - last;
      T2[h2] = s;
      h2 = ((h2 << (e2 >> 1)) ^ (s >> (32u - e2))) & m2;
      last = v;
      dst[(size_t)(i + k) * arity + comp] = v;
      }
    }
  }

// LDS of one chain: the parser's staging window and the tables of the tail loop
struct ChainLds
  {
  uint32_t win[WINW + 4];
  uint32_t T2[1024], T1[16];          // tail loop only
  uint32_t bad, q;
  };

// One chain = two waves of a workgroup (role 0: chain, role 1: parser; `t2` = threadIdx within the pair, 0..127).  Every wave of
// the workgroup reaches every barrier, whatever its chain looks like: a workgroup may carry several chains (CPW), and chains
// beyond the end of the job table or with unusable headers only wait.
__device__ __forceinline__ void decode_pair(const Fpc32ChainJob& job, bool exists, uint32_t* __restrict__ T2gw, ChainLds& L, int role, int lane)
  {
  const uint32_t t2 = (uint32_t)role * 64u + (uint32_t)lane;
  for (uint32_t i = t2; i < 1024u; i += 128u)
    L.T2[i] = 0u;
  if (t2 < 16u)
    L.T1[t2] = 0u;
  if (t2 == 0u)
    {
    L.bad = 0u;
    L.q = 5u;
    }
  const uint8_t* in = job.pay;
  const uint32_t len = exists ? job.size : 0u;
  const uint32_t n = job.n;
  const int arity = (int)job.stride;
  uint32_t* dst = job.dst;
  uint32_t early = 0;                               // status bits known before any work
  uint32_t e1 = 4u, e2 = 10u;
  if (exists)
    {
    if (len < 5u)
      early = FPC_STATUS_SHORT;
    else
      {
      e1 = (uint32_t)(in[0] >> 4) << 1;
      e2 = (uint32_t)(in[0] & 15) << 1;
      const uint32_t cnt = ((uint32_t)in[1] << 24) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 8) | in[4];
      if (cnt != n || e1 == 0u || e2 == 0u || e1 > 4u || e2 > 10u)
        early = FPC_STATUS_HEADER;
      }
    }
  // (header bytes come through vector loads; what is derived from them is wave-uniform all the same)
  early = rfl(early);
  e1 = rfl(e1);
  e2 = rfl(e2);
  const bool work = exists && early == 0u;
  const bool standard = (e1 == 4u && e2 == 10u);
  const uint32_t nb = (work && standard) ? n / 64u : 0u;
  const uint32_t* T2g = T2gw;
  if (role == 0 && nb)
    {
    // the tables start at zero (fpsc.c:219-228) and so do the counters; scalar stores of whole lines, so that every line is in
    // the scalar cache whatever it held
    for (uint32_t off = 0; off < 4u * SCRATCH_USED; off += 16u)
      asm volatile("s_mov_b64 s[40:41], 0\n s_mov_b64 s[42:43], 0\n s_store_dwordx4 s[40:43], %0, %1" :: "s"(T2g), "s"(off) : "s40", "s41", "s42", "s43", "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  __syncthreads();
  ChainEnd ce = { 0u, 0u, 0u };
  // The two waves run decoupled through a ring of RING batches in the scratch: `produced` = batches parsed (slot t % RING holds
  // batch t's residuals and mask), `consumed` = batches decoded; both counters are scalar-memory words, so the chain wave
  // never touches LDS or a VGPR-to-SGPR path.  (First version of round 2: one barrier per batch, the chain lost 2-10 % waiting
  // at it; second: LDS counters and the parser storing the values, ~400 cycles of chain time per batch.)
  if (role == 1 && nb)
    {
    // ---- parser: window over the payload, in units of aligned dwords of the underlying buffer --------------------------
    const uint32_t al = (uint32_t)((uintptr_t)in & 3u);
    const uint32_t* abase = (const uint32_t*)(in - al);
    const uint32_t total_q = len + al;                   // payload end in aligned-byte coordinates
    const uint32_t ndw = (total_q + 3u) >> 2;
    uint32_t wd = 0;                                     // first dword of the window
    uint32_t q = 5u + al;                                // read cursor, aligned-byte coordinates
    uint32_t* win = L.win;
    auto refill = [&](uint32_t from_q)
      {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      wd = from_q >> 2;
      for (uint32_t i = (uint32_t)lane; i < (uint32_t)WINW + 4u; i += 64u)
        win[i] = (wd + i < ndw) ? abase[wd + i] : 0u;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };
    refill(q);
    uint32_t t = 0;
    uint32_t failed = 0;                                 // 1: malformed payload, 2: the chain stopped answering
    // where the chain's bodies are (it publishes that first thing), and the records that end the batches
    uint32_t spins = 0;
    while (counter_load(T2g, SCRATCH_CODE + 2u) == 0u)
      {
      if (++spins > SPIN_LIMIT_PARSER)
        {
        failed = 2u;
        break;
        }
      __builtin_amdgcn_s_sleep(4);
      }
    const uint64_t code_base = ((uint64_t)counter_load(T2g, SCRATCH_CODE + 1u) << 32) | counter_load(T2g, SCRATCH_CODE);
    for (uint32_t sl = 0; sl < RING && !failed; ++sl)
      chain5_put_end(code_base + (uint64_t)CH5_SLOT_END * (uint64_t)CH5_STRIDE, T2g + SCRATCH_X + SLOT_DWORDS * sl);
    spins = 0;
    uint32_t flagged_slots = 0;                          // two bits per ring slot: it holds the records of a flagged batch of that kind
    while (t < nb && !failed)
      {
      const uint32_t cons = counter_load(T2g, SCRATCH_CONSUMED);
      if (cons == ABORT)                                 // the chain gave up waiting (it reports the timeout itself)
        break;
      if (t >= cons + RING)
        {
        if (++spins > SPIN_LIMIT_PARSER)
          {
          failed = 2u;
          break;
          }
        __builtin_amdgcn_s_sleep(4);
        continue;
        }
      spins = 0;
      if (q + BATCH_BYTES + 8u > 4u * (wd + (uint32_t)WINW))
        refill(q);
      // ---- positions of the 8 groups: scalar walk over the headers -----------------------------------
      uint32_t lq = q - 4u * wd;
      uint32_t bcv = 0, myq = 0;
#pragma unroll
      for (uint32_t g = 0; g < 8u; ++g)
        {
        const uint32_t w = rfl(__builtin_amdgcn_alignbyte(win[(lq >> 2) + 1u], win[lq >> 2], lq & 3u));
        const uint32_t bc = __builtin_bswap32(w) >> 8;              // 3 header bytes, big-endian (fpsc.c:245-247)
        if (((uint32_t)lane >> 3) == g)
          {
          bcv = bc;
          myq = lq;
          }
        lq += 3u + lens_sum(bc);
        }
      const uint32_t qend = 4u * wd + lq;
      if (qend > total_q)
        {
        failed = 1u;
        break;
        }
      q = qend;
      // ---- all 64 lanes fetch their residual ------------------------------------------------------------
      const uint32_t j3 = 3u * ((uint32_t)lane & 7u);
      const uint32_t code = (bcv >> j3) & 7u;
      const uint32_t nbytes = code <= 4u ? code : code - 4u;
      const uint32_t rp = myq + 3u + lens_sum(bcv & ((1u << j3) - 1u));
      const uint32_t raw = __builtin_amdgcn_alignbyte(win[(rp >> 2) + 1u], win[rp >> 2], rp & 3u);
      const uint32_t xr = nbytes ? __builtin_bswap32(raw) >> (8u * (4u - nbytes)) : 0u;
      const uint64_t dfcm = __ballot(code > 4u);
      // batches of 64 exact hits (the chain extrapolates them, see run_check in tools/gen_chain5.py): 1 = all FCM-coded
      // without residual, 2 = all DFCM-coded with a zero residual byte
      const uint32_t kind = __ballot(code != 0u) == 0ull ? 1u : (__ballot(code != 5u || xr != 0u) == 0ull ? 2u : 0u);
      // the body of quad q (lanes 0..15): by the parity of q and the kinds of its four values; the first quad of a flagged
      // batch goes to the body that tries the extrapolation
      // (bit 4: the kind of the value behind the quad; the last quad of a batch takes the bodies that assume an F there)
      uint32_t slot_idx = (((uint32_t)lane & 1u) << 5) | ((uint32_t)(dfcm >> (4u * ((uint32_t)lane & 15u))) & 31u);
      if (lane == 0 && kind != 0u)
        slot_idx = kind == 1u ? (uint32_t)CH5_SLOT_RUN1 : (uint32_t)CH5_SLOT_RUN2;
      const uint64_t target = code_base + (uint64_t)slot_idx * (uint64_t)CH5_STRIDE;
      // (a slot that holds the records of an earlier flagged batch of the same kind is not written again - they are the same,
      // residuals zero: on smooth streams the parser, not the chain, is what the stream waits for)
      const uint32_t sl = t % RING;
      if (kind == 0u || ((flagged_slots >> (2u * sl)) & 3u) != kind)
        chain5_put_batch(xr, (uint32_t)target, (uint32_t)(target >> 32), T2g + SCRATCH_X + SLOT_DWORDS * sl, 4u * (SCRATCH_X + SLOT_DWORDS * sl));
      flagged_slots = (flagged_slots & ~(3u << (2u * sl))) | (kind << (2u * sl));
      ++t;
      counter_store(T2g, SCRATCH_PRODUCED, t);
      }
    if (failed)
      counter_store(T2g, SCRATCH_PRODUCED, ABORT);
    if (lane == 0)
      {
      L.q = q - al;
      if (failed)
        atomicOr(&L.bad, failed);
      }
    }
  else if (role == 0 && nb)
    {
    // the chain owns its SIMD's issue slots whenever it can issue: other kernels' waves (the sweeps of the LZ4 decoder, other
    // archives) may share the CU
    __builtin_amdgcn_s_setprio(3);
    const uint32_t vb = 16u * (uint32_t)lane * (uint32_t)arity, vs = 4u * (uint32_t)arity;       // byte offset of value 4 lane (+ j) in its batch
    ce = chain5_run(T2g, nb, dst, vb, vb + vs, vb + 2u * vs, vb + 3u * vs, 256u * (uint32_t)arity, 4u * (uint32_t)lane - 1u);
    __builtin_amdgcn_s_setprio(0);
    if (ce.timed_out && lane == 0)
      atomicOr(&L.bad, 2u);
    }
  __syncthreads();
  const uint32_t i0 = 64u * nb;
  if (nb)
    {
    // no dirty line of the scalar cache may outlive the scratch buffer; the tail loop below works on LDS copies of the tables
    if (role == 0)
      asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  __syncthreads();
  bool bad = L.bad != 0u;
  if (nb && i0 < n && !bad)
    {
    for (uint32_t i = t2; i < 1024u; i += 128u)
      L.T2[i] = __builtin_nontemporal_load(T2g + i);
    if (t2 < 16u)
      L.T1[t2] = __builtin_nontemporal_load(T2g + SCRATCH_T1 + t2);
    }
  __syncthreads();
  uint32_t st = early | ((L.bad & 1u) ? FPC_STATUS_MALFORMED : 0u) | ((L.bad & 2u) ? FPC_STATUS_TIMEOUT : 0u);
  if (work && !bad && i0 < n && t2 == 0u)
    {
    // tail of the stream (fewer than 64 values, fpsc.c:329-414), or a stream with smaller tables than the API's
    uint32_t pos = L.q, h1 = ce.last >> 28, h2 = ce.a2 >> 2, last = ce.last;
    bool tail_bad = false;
    serial_values(in, len, pos, i0, n, e1, e2, h1, h2, last, L.T1, L.T2, dst, arity, 0, tail_bad);
    if (tail_bad)
      st |= FPC_STATUS_MALFORMED;
    }
  if (exists && st != 0u && t2 == 0u)
    atomicOr(job.status, st);
  }

// chains per workgroup: waves 0 .. CPW-1 run the chains (SIMDs 0 .. CPW-1 when the dispatcher places a workgroup's waves round
// robin), waves CPW .. 2 CPW-1 the parsers
template <int CPW>
__device__ __forceinline__ void decode_group(const Fpc32ChainJob* __restrict__ jobs, uint32_t njobs, uint32_t first, uint32_t* __restrict__ scratch)
  {
  __shared__ ChainLds lds[CPW];
  const int lane = threadIdx.x & 63;
  const int wave = (int)rfl(threadIdx.x >> 6);
  const int role = wave >= CPW ? 1 : 0;
  const int slot = wave - role * CPW;
  const uint32_t id = first + (uint32_t)slot;
  const bool exists = id < njobs;
  // everything about the chain is wave-uniform, and the scalar code needs it in SGPRs
  const Fpc32ChainJob* jp = jobs + (exists ? id : 0u);
  Fpc32ChainJob job;
  job.pay = (const uint8_t*)rfl64((uint64_t)(uintptr_t)jp->pay);
  job.dst = (uint32_t*)rfl64((uint64_t)(uintptr_t)jp->dst);
  job.status = (uint32_t*)rfl64((uint64_t)(uintptr_t)jp->status);
  job.size = rfl(jp->size);
  job.n = rfl(jp->n);
  job.stride = rfl(jp->stride);
  job.pad = 0u;
  uint32_t* sc = (uint32_t*)rfl64((uint64_t)(uintptr_t)(scratch + (size_t)SCRATCH_DWORDS * id));
  decode_pair(job, exists, sc, lds[slot], role, lane);
  }

__global__ void __launch_bounds__(128) k_fpc32_decode(ChainJobs3 args, uint32_t njobs, uint32_t* __restrict__ scratch)
  {
  decode_group<1>(args.j, njobs, blockIdx.x, scratch);
  }

template <int CPW>
__global__ void __launch_bounds__(128 * CPW) k_fpc32_decode_batch(const Fpc32ChainJob* __restrict__ jobs, uint32_t njobs, uint32_t* __restrict__ scratch)
  {
  decode_group<CPW>(jobs, njobs, blockIdx.x * (uint32_t)CPW, scratch);
  }

// ---- the same decoder with NOTHING behind the scalar cache ---------------------------------------------------------------
// k_fpc32_decode_robust: one workgroup of two waves per chain like the kernel above, but both predictor tables, the ring of
// residuals and the two counters live in LDS, and the chain is ordinary vector code.  Everything the pair knows is then in
// registers and LDS, which travel with the workgroup when the hardware scheduler saves and restores it: this kernel cannot lose
// its tables the way the scalar-cache chain can (shim.hip, "the chain decoders and their self-check").  The price is four LDS
// instructions per value on the dependent path (13-17 cycles of issue each, a read behind a write ~120): measured ~2 x the
// scalar chain's time per value.  It is the third rung of the repeat ladder for float streams (before the reference-order
// kernel), and TRICO_HIP_DECODE_ROBUST=1 makes it the first choice for processes that expect to be preempted.
constexpr uint32_t RRING = 4;                                   // batches the parser may run ahead

struct RobustLds
  {
  uint32_t win[WINW + 4];
  uint32_t T2[1024], T1[16];
  uint32_t x[RRING][64];                                        // residuals of a batch, lane K = value K
  uint32_t mlo[RRING], mhi[RRING];                              // its mask of DFCM-coded values
  uint32_t produced, consumed, bad, q;
  };

__global__ void __launch_bounds__(128) k_fpc32_decode_robust(const Fpc32ChainJob* __restrict__ jobs, uint32_t njobs)
  {
  __shared__ RobustLds L;
  const int lane = threadIdx.x & 63;
  const int role = (int)rfl(threadIdx.x >> 6);
  const uint32_t id = blockIdx.x;
  if (id >= njobs)
    return;
  const Fpc32ChainJob* jp = jobs + id;
  const uint8_t* in = (const uint8_t*)rfl64((uint64_t)(uintptr_t)jp->pay);
  uint32_t* dst = (uint32_t*)rfl64((uint64_t)(uintptr_t)jp->dst);
  uint32_t* status = (uint32_t*)rfl64((uint64_t)(uintptr_t)jp->status);
  const uint32_t len = rfl(jp->size), n = rfl(jp->n);
  const int arity = (int)rfl(jp->stride);
  for (uint32_t i = threadIdx.x; i < 1024u; i += 128u)
    L.T2[i] = 0u;
  if (threadIdx.x < 16u)
    L.T1[threadIdx.x] = 0u;
  if (threadIdx.x == 0u)
    {
    L.produced = 0u; L.consumed = 0u; L.bad = 0u; L.q = 5u;
    }
  uint32_t early = 0, e1 = 4u, e2 = 10u;
  if (len < 5u)
    early = FPC_STATUS_SHORT;
  else
    {
    e1 = (uint32_t)(in[0] >> 4) << 1;
    e2 = (uint32_t)(in[0] & 15) << 1;
    const uint32_t cnt = ((uint32_t)in[1] << 24) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 8) | in[4];
    if (cnt != n || e1 == 0u || e2 == 0u || e1 > 4u || e2 > 10u)
      early = FPC_STATUS_HEADER;
    }
  early = rfl(early); e1 = rfl(e1); e2 = rfl(e2);
  const bool work = early == 0u;
  const uint32_t nb = (work && e1 == 4u && e2 == 10u) ? n / 64u : 0u;
  __syncthreads();
  volatile uint32_t* produced = &L.produced;
  volatile uint32_t* consumed = &L.consumed;
  uint32_t last = 0, h1 = 0, h2 = 0;
  if (role == 1 && nb)
    {
    // ---- parser: the same walk as in decode_pair, residuals to the LDS ring ------------------------------------------------
    const uint32_t al = (uint32_t)((uintptr_t)in & 3u);
    const uint32_t* abase = (const uint32_t*)(in - al);
    const uint32_t total_q = len + al;
    const uint32_t ndw = (total_q + 3u) >> 2;
    uint32_t wd = 0, q = 5u + al;
    uint32_t* win = L.win;
    auto refill = [&](uint32_t from_q)
      {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      wd = from_q >> 2;
      for (uint32_t i = (uint32_t)lane; i < (uint32_t)WINW + 4u; i += 64u)
        win[i] = (wd + i < ndw) ? abase[wd + i] : 0u;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };
    refill(q);
    uint32_t t = 0, failed = 0, spins = 0;
    while (t < nb)
      {
      const uint32_t cons = *consumed;
      if (cons == ABORT)
        break;
      if (t >= cons + RRING)
        {
        if (++spins > SPIN_LIMIT_PARSER) { failed = 2u; break; }
        __builtin_amdgcn_s_sleep(4);
        continue;
        }
      spins = 0;
      if (q + BATCH_BYTES + 8u > 4u * (wd + (uint32_t)WINW))
        refill(q);
      uint32_t lq = q - 4u * wd, bcv = 0, myq = 0;
#pragma unroll
      for (uint32_t g = 0; g < 8u; ++g)
        {
        const uint32_t w = rfl(__builtin_amdgcn_alignbyte(win[(lq >> 2) + 1u], win[lq >> 2], lq & 3u));
        const uint32_t bc = __builtin_bswap32(w) >> 8;
        if (((uint32_t)lane >> 3) == g) { bcv = bc; myq = lq; }
        lq += 3u + lens_sum(bc);
        }
      const uint32_t qend = 4u * wd + lq;
      if (qend > total_q) { failed = 1u; break; }
      q = qend;
      const uint32_t j3 = 3u * ((uint32_t)lane & 7u);
      const uint32_t code = (bcv >> j3) & 7u;
      const uint32_t nbytes = code <= 4u ? code : code - 4u;
      const uint32_t rp = myq + 3u + lens_sum(bcv & ((1u << j3) - 1u));
      const uint32_t raw = __builtin_amdgcn_alignbyte(win[(rp >> 2) + 1u], win[rp >> 2], rp & 3u);
      const uint32_t xr = nbytes ? __builtin_bswap32(raw) >> (8u * (4u - nbytes)) : 0u;
      const uint64_t dfcm = __ballot(code > 4u);
      const uint32_t sl = t % RRING;
      L.x[sl][lane] = xr;
      if (lane == 0) { L.mlo[sl] = (uint32_t)dfcm; L.mhi[sl] = (uint32_t)(dfcm >> 32); }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      ++t;
      if (lane == 0) *produced = t;
      }
    if (failed && lane == 0)
      {
      *produced = ABORT;
      atomicOr(&L.bad, failed);
      }
    if (lane == 0)
      L.q = q - al;
    }
  else if (role == 0 && nb)
    {
    // ---- chain: every lane runs the recurrence on the same values (LDS broadcast reads; lane 0 writes the tables) ----------
    // (volatile: lane 0 writes the table, every lane reads it; the accesses must stay where they are written)
    volatile uint32_t* T2v = L.T2;
    uint32_t t1reg = 0;                                  // the FCM table: lane j < 16 holds entry j (read with v_readlane)
    uint32_t spins = 0;
    bool stop = false;
    for (uint32_t b = 0; b < nb && !stop; ++b)
      {
      for (;;)
        {
        const uint32_t p = *produced;
        if (p == ABORT) { stop = true; break; }
        if (p > b) break;
        if (++spins > SPIN_LIMIT_CHAIN)
          {
          if (lane == 0) { *consumed = ABORT; atomicOr(&L.bad, 2u); }
          stop = true;
          break;
          }
        __builtin_amdgcn_s_sleep(1);
        }
      if (stop)
        break;
      spins = 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      const uint32_t sl = b % RRING;
      const uint32_t xk = L.x[sl][lane];
      const uint64_t mask = ((uint64_t)L.mhi[sl] << 32) | L.mlo[sl];
      uint32_t outv = 0;
#pragma unroll 8
      for (int k = 0; k < 64; ++k)
        {
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xk, k);
        const uint32_t t2 = T2v[h2];
        const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)t1reg, (int)rfl(h1));
        const uint32_t pred = ((mask >> k) & 1ull) ? last + t2 : t1;          // fpsc.c:308-311
        const uint32_t v = x ^ pred;
        const uint32_t s = v - last;
        t1reg = (uint32_t)lane == h1 ? v : t1reg;                              // fpsc.c:312-314
        if (lane == 0)
          T2v[h2] = s;                                                         // fpsc.c:323-326
        h1 = v >> 28;
        h2 = ((h2 << 5) ^ (s >> 22)) & 1023u;
        last = v;
        outv = lane == k ? v : outv;
        }
      dst[((size_t)64u * b + (uint32_t)lane) * (uint32_t)arity] = outv;
      if (lane == 0) *consumed = b + 1u;
      }
    if (lane < 16)
      L.T1[lane] = t1reg;                                // for the tail loop
    }
  __syncthreads();
  // hand the chain's state to thread 0 for the tail (fewer than 64 values, or table shapes below the API's)
  __shared__ uint32_t hand[3];
  if (role == 0 && lane == 0) { hand[0] = last; hand[1] = h1; hand[2] = h2; }
  __syncthreads();
  const bool bad = L.bad != 0u;
  uint32_t st = early | ((L.bad & 1u) ? FPC_STATUS_MALFORMED : 0u) | ((L.bad & 2u) ? FPC_STATUS_TIMEOUT : 0u);
  const uint32_t i0 = 64u * nb;
  if (work && !bad && i0 < n && threadIdx.x == 0)
    {
    uint32_t pos = L.q, t_h1 = hand[1], t_h2 = hand[2], t_last = hand[0];
    bool tail_bad = false;
    serial_values(in, len, pos, i0, n, e1, e2, t_h1, t_h2, t_last, L.T1, L.T2, dst, arity, 0, tail_bad);
    if (tail_bad)
      st |= FPC_STATUS_MALFORMED;
    }
  if (st != 0u && threadIdx.x == 0)
    atomicOr(status, st);
  }

// One chain per CU: the chain wave needs its SIMD's issue slots and its CU's scalar cache; the workgroup asks for more than
// half of the CU's LDS (it uses a fraction of it) so that no second workgroup of this kernel can be placed beside it.
constexpr size_t CLAIM = 88u << 10;

template <typename K>
bool claim_lds(K kernel)
  {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLAIM) == hipSuccess;
  }

} // namespace

int launch_fpc32_decode(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, uint32_t n, void* d_dst,
                        uint32_t* d_status, uint32_t* d_scratch)
  {
  ChainJobs3 a;
  for (int c = 0; c < 3; ++c)
    a.j[c] = Fpc32ChainJob{ c < arity ? d_payloads[c] : nullptr, (uint32_t*)d_dst + c, d_status, c < arity ? sizes[c] : 0u, n, (uint32_t)arity, 0u };
  static const bool claimed = claim_lds(k_fpc32_decode);
  hipLaunchKernelGGL(k_fpc32_decode, dim3(arity), dim3(128), claimed ? CLAIM : 0, current_stream(), a, (uint32_t)arity, d_scratch);
  return hip_ok(hipGetLastError(), "k_fpc32_decode") ? 1 : 0;
  }

// the LDS-only decoder for every chain of the table (no scratch: nothing lives outside registers and LDS)
int launch_fpc32_decode_robust(const Fpc32ChainJob* d_jobs, uint32_t njobs)
  {
  if (njobs == 0)
    return 1;
  hipLaunchKernelGGL(k_fpc32_decode_robust, dim3(njobs), dim3(128), 0, current_stream(), d_jobs, njobs);
  return hip_ok(hipGetLastError(), "k_fpc32_decode_robust") ? 1 : 0;
  }

// All chains of a batch in ONE launch (`d_jobs`: device table of njobs descriptors, `d_scratch`: FPC32_DECODE_TABLE_BYTES per chain):
// throughput no longer depends on how many hardware queues the process has.  chains_per_group 1, 2 or 4.
int launch_fpc32_decode_batch(const Fpc32ChainJob* d_jobs, uint32_t njobs, uint32_t* d_scratch, int chains_per_group)
  {
  if (njobs == 0)
    return 1;
  hipStream_t st = current_stream();
  if (chains_per_group >= 4)
    {
    static const bool claimed = claim_lds(k_fpc32_decode_batch<4>);
    hipLaunchKernelGGL(k_fpc32_decode_batch<4>, dim3((njobs + 3u) / 4u), dim3(512), claimed ? CLAIM : 0, st, d_jobs, njobs, d_scratch);
    }
  else if (chains_per_group >= 2)
    {
    static const bool claimed = claim_lds(k_fpc32_decode_batch<2>);
    hipLaunchKernelGGL(k_fpc32_decode_batch<2>, dim3((njobs + 1u) / 2u), dim3(256), claimed ? CLAIM : 0, st, d_jobs, njobs, d_scratch);
    }
  else
    {
    static const bool claimed = claim_lds(k_fpc32_decode_batch<1>);
    hipLaunchKernelGGL(k_fpc32_decode_batch<1>, dim3(njobs), dim3(128), claimed ? CLAIM : 0, st, d_jobs, njobs, d_scratch);
    }
  return hip_ok(hipGetLastError(), "k_fpc32_decode_batch") ? 1 : 0;
  }

} // namespace trico
