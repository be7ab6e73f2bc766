// fpc32_common.hpp — what the translation units of the 32-bit float encoder share (gfx950, wave64): table geometry, the workspace
// plan, the record of a deferred value, a few wave primitives.  Internal; nothing here crosses the C-ABI.
#pragma once

#include "common.hpp"
#include <stdlib.h>

namespace trico {
namespace fpc32 {

constexpr int TAB = 1040;      // 16 FCM entries followed by 1024 DFCM entries (fpsc.c:96-97 with the API's exponents 4 and 10)
constexpr int ROW = 1040;      // words per (segment, component) row in the global tables
constexpr int CH = 32;         // segments per chunk in the cross-segment scans
constexpr uint32_t RCAP = 1040;   // deferred values per (segment, component): a class is met for the first time at most once
constexpr uint32_t RECW = 8;      // words per record (two 16-byte stores)
constexpr uint32_t GUARD_WGS = 16;          // workgroups of the sweep that code the beginning of a sampled segment again (write-side guard)
constexpr uint32_t GUARD_SLOT = 64 * 280 + 512;   // bytes such a workgroup can produce per component (k_fpc32_sweep.hip: GUARD_CAP)

// A table entry nobody has written in this segment.  A payload that equals it cannot be told from "never written", so a wave that
// WRITES this value raises the collision flag and the host codes the stream again with the two-sweep coder (k_fpc32_encode.hip),
// which has no sentinel: 2^-32 per DFCM write on random bits, a quiet NaN nobody stores for the FCM table.
constexpr uint32_t SENT = 0x7fc0dead;

// what a wave of the one-sweep coder can raise; the flags travel in the upper half of its record count (Plan::off_nrec) and reach
// the host with the sizes (k_fpc32_offsets)
constexpr uint32_t FLAG_ORDER = FPC32_FLAG_ORDER, FLAG_SENTINEL = FPC32_FLAG_SENTINEL, FLAG_SCAN = FPC32_FLAG_SCAN;

// Record of a deferred value (k_fpc32_sweep -> k_fpc32_fixup -> k_fpc32_gather), RECW words:
//   w0 slot offset of the value's four reserved bytes | gi << 24 | ft1 << 28 | ft2 << 29
//   w1 slot offset of the three header bytes of the value's group
//   w2 the value   w3 its predecessor   w4 the prediction that IS known (one open class), else 0
//   w5 4 * DFCM class of the value      w6 residual, w7 length | code << 4  (written by the fix-up)
// gi = index in the group; ft1 / ft2 = the FCM / DFCM prediction is the one the segment does not know (the FCM class is the top
// four bits of w3).  Slot offsets stay below 2^24: make_plan caps the segment length (L_MAX) whatever the device's resident target is.
constexpr uint32_t REC_POS = 0x00ffffffu, REC_GI_SHIFT = 24, REC_FT1 = 1u << 28, REC_FT2 = 1u << 29;
constexpr uint32_t REC_CMPW = 6;  // words the sweep writes (what the write-side guard compares)

// Segment lengths by order of dispatch.  The workgroups of the one-sweep coder are all resident at once, a compute unit holds K of
// them, and the k-th one it was given (blockIdx / number of compute units: the dispatcher hands every unit its k-th workgroup before
// any gets its k+1-th) is the k-th oldest on its SIMDs: the hardware issues the oldest ready wave first, so with equal segments the
// youngest workgroups finished 25 % (benchmark mesh) to 58 % (three noisy components) later than the oldest and the last fifth of the
// kernel ran on a half-empty device (profiles/r06_sweep_stagger.txt).  Hence classes of C consecutive segments whose length falls
// with the class: everybody ends at about the same time.  Class k: `first[k]` segments of len[k] values, the others 512 fewer,
// beginning at value index base[k]; every length is a multiple of 512.  K == 0: all segments have Plan::L values.
constexpr int STAGGER_MAX = 32;
struct Stagger
  {
  uint32_t C, K;
  uint32_t base[STAGGER_MAX], len[STAGGER_MAX], first[STAGGER_MAX];
  };

struct Plan
  {
  uint32_t L, S, segcap, nch;
  Stagger sg;
  size_t rows, slot_stride;
  size_t off_summ, off_inc, off_chmax, off_agg, agg_words, off_nrec, off_recs, off_segbytes, off_rawbytes, off_segoff, off_gslots, off_grecs, off_gmeta, off_diag, off_slots, total;
  };

// Workgroups of k_fpc32_sweep (one wave per component) the current device holds at once (k_fpc32_sweep.hip).  The sweep wants its whole
// grid resident in ONE round: a workgroup that has to wait for a free place starts when the first ones are done and ends a segment's
// time later (measured with TRICO_SWEEP_DIAG in round 5: 10 % of the workgroups in a second round were 40 % of the kernel's time).
int fpc32_sweep_resident_workgroups(int arity);
int fpc32_sweep_class_size();                 // segments per class of the staggered geometry: the compute units of the current device
int fpc32_sweep_stagger_permille();           // how much longer than the mean the first class's segments are (the last: shorter), in 1/1000
int fpc32_sweep_class_weights(uint32_t K, int arity, double* w);     // 1: w[0..K) are the relative lengths of the classes (measured tables); 0: the linear rule

inline Plan make_plan(uint32_t n, int arity)
  {
  Plan p;
  // (the guard's workgroups take part in the sweep's launch: leave them room, so that the whole grid is resident at once)
  uint32_t target = (uint32_t)fpc32_sweep_resident_workgroups(arity);
  if (target > 8u * GUARD_WGS)
    target -= GUARD_WGS;
  uint64_t L = ((uint64_t)n + target - 1) / target;
  L = (L + 511) / 512 * 512;                    // whole blocks of eight steps: the hand-written loop of the sweep takes those, the compiled step the rest
  if (L < 1024) L = 1024;
  // a record keeps slot offsets in 24 bits (REC_POS): on a device with few compute units (a partition) the resident target would
  // make the segments of a large stream longer than that - then there are more segments than resident workgroups instead
  constexpr uint64_t L_MAX = 3800064;           // 7422 blocks of 512 values: 5 + 4.375 L + 296 rounded up to 256 stays below 2^24
  static_assert(5 + 4 * L_MAX + 3 * (L_MAX / 8) + 16 + 280 + 256 <= (uint64_t)REC_POS + 1, "slot offsets of a record");
  if (L > L_MAX) L = L_MAX;
  p.L = (uint32_t)L;
  p.S = (uint32_t)(((uint64_t)n + L - 1) / L);
  if (p.S == 0) p.S = 1;
  // the staggered geometry of the one-sweep coder: the same S segments, the same blocks of 512 values in total
  p.sg.C = p.sg.K = 0;
  uint64_t Lmax = L;
  {
  const uint32_t C = (uint32_t)fpc32_sweep_class_size();
  const uint32_t beta = (uint32_t)fpc32_sweep_stagger_permille();
  const uint32_t K = C ? (p.S + C - 1) / C : 0u;
  const uint64_t B = ((uint64_t)n + 511) / 512;                       // blocks of the stream
  if (beta > 0 && K >= 2 && K <= (uint32_t)STAGGER_MAX && B >= 4ull * p.S)
    {
    // weight of class k: 1 + beta * (1 - 2 (k + 1/2) / K), normalised over the segments; blocks of the class = its share, rounded so
    // that the classes add up to B
    double wsum = 0;
    double w[STAGGER_MAX];
    uint32_t cnt[STAGGER_MAX];
    const int table = fpc32_sweep_class_weights(K, arity, w);
    for (uint32_t k = 0; k < K; ++k)
      {
      cnt[k] = k + 1 < K ? C : p.S - k * C;
      if (!table)
        w[k] = 1.0 + (double)beta / 1000.0 * (1.0 - (2.0 * k + 1.0) / (double)K);
      wsum += w[k] * cnt[k];
      }
    uint64_t given = 0, at = 0;
    double acc = 0;
    bool ok = true;
    for (uint32_t k = 0; k < K; ++k)
      {
      acc += w[k] * cnt[k] / wsum * (double)B;
      uint64_t upto = k + 1 < K ? (uint64_t)(acc + 0.5) : B;
      if (upto > B) upto = B;
      const uint64_t Tk = upto - given;                                 // blocks of class k
      given = upto;
      const uint64_t lk = (Tk + cnt[k] - 1) / cnt[k];                   // blocks of its longer segments
      if (lk < 3 || lk * 512 > L_MAX) { ok = false; break; }
      p.sg.base[k] = (uint32_t)(at * 512);
      p.sg.len[k] = (uint32_t)(lk * 512);
      p.sg.first[k] = (uint32_t)(cnt[k] - (lk * cnt[k] - Tk));          // the others: one block fewer
      at += Tk;
      if (lk * 512 > Lmax) Lmax = lk * 512;
      }
    if (ok)
      {
      p.sg.C = C;
      p.sg.K = K;
      }
    else
      Lmax = L;
    }
  }
  p.segcap = (uint32_t)align_up(5 + 4 * (size_t)Lmax + 3 * ((size_t)Lmax / 8) + 16 + 280, 256);
  p.nch = (p.S + CH - 1) / CH;
  p.rows = (size_t)p.S * arity;
  p.slot_stride = (size_t)p.S * p.segcap;
  size_t o = 0;
  p.off_summ = o;      o += align_up(p.rows * ROW * 4, 256);
  p.off_inc = o;       o += align_up(p.rows * ROW * 4, 256);
  p.off_chmax = o;     o += align_up((size_t)p.nch * arity * TAB * 4, 256);
  // the one-sweep coder's scan (k_fpc32_scanfix): per (chunk, component, class) one 64-bit word "ready | latest entry"; the sweep zeroes them
  p.agg_words = (size_t)(p.nch + (p.nch + 7) / 8) * arity * TAB;        // (the chunks' words, then one per group of eight chunks)
  p.off_agg = o;       o += align_up(p.agg_words * 8, 256);
  p.off_nrec = o;      o += align_up(p.rows * 4, 256);
  p.off_recs = o;      o += align_up(p.rows * RCAP * RECW * 4, 256);
  p.off_segbytes = o;  o += align_up(p.rows * 4, 256);
  p.off_rawbytes = o;  o += align_up(p.rows * 4, 256);
  p.off_segoff = o;    o += align_up(p.rows * 4, 256);
  p.off_gslots = o;    o += align_up((size_t)GUARD_WGS * arity * GUARD_SLOT, 256);
  p.off_grecs = o;     o += align_up((size_t)GUARD_WGS * arity * RCAP * RECW * 4, 256);
  p.off_gmeta = o;     o += align_up((size_t)GUARD_WGS * 3 * 16, 256);
  p.off_diag = o;      o += 512;                 // diagnostic builds (-DTRICO_SWEEP_DIAG): clocks of one wave per component; from byte 256: records per component (offsets -> gather)
  p.off_slots = o;     o += p.slot_stride * arity;
  p.total = o + 256;
  return p;
  }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// first value and number of values of segment g (the last segment of a stream ends with the stream: the caller clips)
__host__ __device__ __forceinline__ void segment_range(const Stagger& sg, uint32_t L, uint32_t g, uint32_t& begin, uint32_t& len)
  {
  if (sg.K == 0u)
    {
    begin = g * L;
    len = L;
    return;
    }
  const uint32_t k = g / sg.C, j = g - k * sg.C;
  const uint32_t f = sg.first[k], lk = sg.len[k];
  begin = sg.base[k] + j * lk - (j > f ? (j - f) * 512u : 0u);
  len = j < f ? lk : lk - 512u;
  }

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ uint32_t dpp_shr1(uint32_t carry, uint32_t v)
  {
  // lane l <- lane l-1, lane 0 <- carry   (DPP wave_shr:1)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
  }

__device__ __forceinline__ uint32_t dpp_shl1(uint32_t carry, uint32_t v)
  {
  // lane l <- lane l+1, lane 63 <- carry   (DPP wave_shl:1)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x130, 0xf, 0xf, false);
  }

// lane l <- lane l-1 of `cur`, lane 0 <- lane 63 of `prev` (the register the previous step kept): a shift across steps without
// a trip through the scalar registers (wave_ror:1 of prev, then wave_shr:1 of cur on top; lane 0 has no source and keeps it)
__device__ __forceinline__ uint32_t shr1_across(uint32_t prev, uint32_t cur)
  {
  const int r = __builtin_amdgcn_mov_dpp((int)prev, 0x13C, 0xf, 0xf, false);        // (every lane has a source: no old value)
  return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)cur, 0x138, 0xf, 0xf, false);
  }

// inclusive prefix sum over the wave (DPP: four steps inside the rows of 16 lanes, two row broadcasts)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
  {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);     // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);     // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);     // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);     // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);     // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);     // row_bcast:31 into rows 2 and 3
  return x;
  }

__device__ __forceinline__ uint32_t popc_below(uint64_t mask)
  {
  // number of set bits of `mask` strictly below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  }
#endif

// k_fpc32_sweep.hip: the one-sweep coder.  launch_fpc32_sweep queues sweep, cross-segment scan and fix-up (sizes of the segments are
// final afterwards: Plan::off_segbytes); launch_fpc32_gather_rec moves the slots of components [c0, c0 + count) to dst[0..count).
int launch_fpc32_sweep(const uint32_t* d_src, uint32_t n, int arity, const Plan& p, uint8_t* d_ws);
int launch_fpc32_gather_rec(const Plan& p, int arity, int c0, int count, const uint8_t* d_ws, uint8_t* const d_dst[3], const uint32_t* d_sizes = nullptr);
// (d_sizes given, c0 = 0, count = arity: d_dst[0] is where the stream's first size field goes, the components follow each other as
// `u32 bytes, payload`, placed by the sizes in device memory)

} // namespace fpc32
} // namespace trico
