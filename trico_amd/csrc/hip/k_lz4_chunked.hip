// k_lz4_chunked.hip — chunk-speculative, byte-exact LZ4 1.9.2 block compressor for large byte planes.
//
// Problem: LZ4_compress_generic (lz4.c:793-1181) is one dependent chain per plane: the 4096-entry hash
// table depends on the whole parse history, and Trico compresses each plane as ONE block (trico.c:343-368).
// A lone wave needs microseconds per sequence, and a 300 MB plane can hold millions of sequences.
//
// Observation that makes an exact parallel parse possible: the parser's behaviour from a point p on
// depends on the table only through entries within 65535 bytes of p (lz4.c:945-948: older candidates are
// rejected before they are even read), and every match end is a clean synchronisation point with
// anchor == ip.  So:
//   parse   (k_lz4_parse):  chunk k > 0 starts WARM bytes before its start with an empty table and
//            parses speculatively.  At the first match end at or after the chunk start it snapshots
//            (ip, table) and from there records its sequences as descriptors until the first match end
//            at or after the chunk end, where it stores its end state.  Chunk 0 is the true parse.
//   extend  (k_lz4_extend, round 5): a recorded match still running after a chunk's worth of bytes is its chunk's last sequence; the
//            parse leaves it open and the whole device counts it (a plane of zeros is one match of 300 MB: 36 ms for one wave).
//   stitch  (k_lz4_stitch): walks the chain: the end state of the last accepted chunk lies in some chunk
//            j; chunk j's speculation is accepted iff its snapshot is at the same ip and its table is
//            equal entry by entry, except where both entries are already out of range at ip.  From an
//            equivalent state the deterministic parser produces identical sequences, so accepted output
//            is exactly the reference's.  Otherwise chunk j is re-parsed from the true state (exact, just
//            slower).  Worst case = the serial parse; typical case = all chunks accepted.
//   sizes / offsets / emit: descriptors -> encoded byte counts -> exclusive scan -> block bytes
//            (token, length extension bytes, literals, offset) written in parallel.
// The descriptor form also takes literal copying off the parsing waves.
#include "common.hpp"
#include <stdlib.h>
#include <stdio.h>

namespace trico {

namespace {

constexpr uint32_t MAXD = 65535u;          // LZ4_DISTANCE_MAX (lz4.h:535)
constexpr int EMIT_T = 256;
constexpr uint32_t EMIT_STAGE = 8192;        // bytes of one emit round that are assembled in LDS

struct __attribute__((packed, aligned(1))) u32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) u64u { uint64_t v; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) u128u { u32x4 v; };
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return ((const u32u*)p)->v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { return ((const u64u*)p)->v; }
__device__ __forceinline__ u32x4 ld128(const uint8_t* p) { return ((const u128u*)p)->v; }
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t hash5(const uint8_t* p) { return (uint32_t)(((ld64(p) << 24) * 889523592379ull) >> 52); }   // lz4.c:643-648

__device__ __forceinline__ bool fast_extra_seen(uint64_t dx, int first) { return __builtin_amdgcn_readlane((int)(dx != 0ull), first) != 0; }
__device__ __forceinline__ uint32_t hash_w(uint64_t w) { return (uint32_t)(((w << 24) * 889523592379ull) >> 52); }       // hash5 of the 8 bytes w

// ---- source bytes through the SCALAR data cache ---------------------------------------------------------------------------------
// A plane of short sequences is a chain of ~40 M steps per plane, each of which looks at the bytes at ip and at ONE candidate: the
// positions are wave-uniform, the data is read-only, and a vector load of it is a round trip of ~0.5 us through the vector memory
// path that the step has to wait for twice.  A scalar load of the same bytes takes ~40 cycles when the line is in the compute unit's
// scalar cache - the bytes at ip are read front to back (every line is fetched once), the candidates of such a plane lie a few
// hundred bytes back - and ~200 from the L2.  The plane is addressed through the constant address space for that (it is not written
// while the kernel runs; the cache is invalidated when a kernel starts): see the chain of zero-literal sequences in lz4_parse.
struct Desc { uint32_t lit, ml, off; };    // literal run, match length incl. MINMATCH (0 = final run), offset

enum { END_NONE = 0, END_MATCH = 1, END_FINAL = 2, END_OPEN = 3 };   // END_OPEN: the last descriptor's match is still being counted (k_lz4_extend)
constexpr uint32_t OPEN_ML = 0xffffffffu;
struct Meta
  {
  uint32_t snap_valid, snap_ip;            // first match end at or after the chunk start (speculative chunks)
  uint32_t end_kind, end_ip;               // END_MATCH: state after a match ended at end_ip; END_FINAL: block finished; END_OPEN: end_ip = where the open match starts
  uint32_t ndesc;
  uint32_t accepted;                        // set by the stitch pass
  uint32_t first_in;                        // input position where this chunk's first descriptor starts
  uint32_t reparsed;
  };

// first differing byte among Q x 1 KiB of a[] and b[] starting at lane offset o0 (0xffffffff: none for this lane); all 2Q loads
// are issued before the first comparison
template <int Q>
__device__ __forceinline__ uint32_t batch_count(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t o0)
  {
  u32x4 xs[Q], ys[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q)
    {
    xs[q] = ld128(a + o0 + 1024u * (uint32_t)q);
    ys[q] = ld128(b + o0 + 1024u * (uint32_t)q);
    }
  uint32_t first = 0xffffffffu;
#pragma unroll
  for (int q = Q - 1; q >= 0; --q)
    {
    const u32x4 x = xs[q], y = ys[q];
    const uint64_t d0 = ((uint64_t)(x.y ^ y.y) << 32) | (x.x ^ y.x), d1 = ((uint64_t)(x.w ^ y.w) << 32) | (x.z ^ y.z);
    if (d0 | d1)
      first = o0 + 1024u * (uint32_t)q + (d0 ? (uint32_t)__builtin_ctzll(d0) >> 3 : 8u + ((uint32_t)__builtin_ctzll(d1) >> 3));
    }
  return first;
  }

// number of equal bytes of a[] and b[], at most `limit`; 4 KiB per iteration
// NW > 1: NW waves of a workgroup run the SAME parse redundantly (every wave its own table) and share the work of counting a
// long match: wave w takes slice w of every round and the waves exchange their results through `xch` (every wave calls
// this function with the same arguments, so the barriers match).
// QL: KiB of a long round per wave (16: 128 registers of loads in flight; the parse pass takes 8 to fit two waves on a SIMD)
template <int NW, int QL = 16>
__device__ __forceinline__ uint32_t wave_count(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t limit, int lane, int wave,
                                               uint32_t* xch)
  {
  uint32_t done = 0;
  // short probe first: most matches are short
  {
  const uint32_t o = 8u * (uint32_t)lane;
  uint64_t x = 0;
  uint32_t valid = 0;
  if (o < limit)
    {
    valid = limit - o < 8u ? limit - o : 8u;
    if (valid == 8u)
      x = ld64(a + o) ^ ld64(b + o);
    else
      for (uint32_t k = 0; k < valid; ++k)
        x |= (uint64_t)(a[o + k] ^ b[o + k]) << (8u * k);
    }
  const uint32_t eq = x ? (uint32_t)__builtin_ctzll(x) >> 3 : valid;
  const uint64_t stop = __ballot(eq < 8u);
  if (stop)
    {
    const int first = __builtin_ctzll(stop);
    return 8u * (uint32_t)first + (uint32_t)__builtin_amdgcn_readlane((int)eq, first);
    }
  done = 512u;
  }
  // long matches: 4 KiB first, then QL KiB per iteration and wave (the loop is bound by the latency of one round of loads,
  // and periodic byte planes consist of matches of tens of KiB)
  for (int Q = 4; done < limit; Q = QL)
    {
    uint32_t first = 0xffffffffu;
    if (done + 1024u * QL * NW > limit)
      Q = 4;
    const uint32_t base = done + 1024u * (uint32_t)(Q * wave);          // my wave's slice of this round
    if (base + 1024u * (uint32_t)Q <= limit)
      first = Q == QL ? batch_count<QL>(a, b, base + 16u * (uint32_t)lane) : batch_count<4>(a, b, base + 16u * (uint32_t)lane);
    else
#pragma unroll 4
    for (int q = 0; q < Q; ++q)
      {
      const uint32_t o = base + 1024u * (uint32_t)q + 16u * (uint32_t)lane;
      if (o + 16u <= limit)
        {
        const u32x4 x = ld128(a + o), y = ld128(b + o);
        const uint64_t d0 = ((uint64_t)(x.y ^ y.y) << 32) | (x.x ^ y.x), d1 = ((uint64_t)(x.w ^ y.w) << 32) | (x.z ^ y.z);
        if (d0 | d1)
          first = min(first, o + (d0 ? (uint32_t)__builtin_ctzll(d0) >> 3 : 8u + ((uint32_t)__builtin_ctzll(d1) >> 3)));
        }
      else if (o < limit)
        {
        uint32_t k = o;
        while (k < limit && a[k] == b[k]) ++k;
        first = min(first, k);                 // k == limit also ends the count
        }
      else
        first = min(first, limit);
      }
    // wave minimum of `first`
#pragma unroll
    for (int s = 32; s > 0; s >>= 1)
      first = min(first, (uint32_t)__shfl_xor((int)first, s));
    if (NW > 1)
      {
      if (lane == 0) xch[wave] = first;
      __syncthreads();
      first = xch[lane < NW ? lane : 0];
#pragma unroll
      for (int s = 32; s > 0; s >>= 1)
        first = min(first, (uint32_t)__shfl_xor((int)first, s));
      __syncthreads();
      }
    if (first != 0xffffffffu)
      return first < limit ? first : limit;
    done += 1024u * (uint32_t)(Q * NW);
    }
  return limit;
  }

// (for the chain of short sequences in lz4_parse: q0, q1, q2 are the 24 bytes from ip - 2 on)
constexpr uint32_t KS = 6;                   // attempts of a search made one by one on the scalar unit
template <int OFF> __device__ __forceinline__ uint64_t bytes_at(uint64_t q0, uint64_t q1, uint64_t q2)
  {
  // the 8 bytes at offset OFF (0 .. 16) of the 24
  if constexpr (OFF == 0) return q0;
  else if constexpr (OFF < 8) return (q0 >> (8 * OFF)) | (q1 << (64 - 8 * OFF));
  else if constexpr (OFF == 8) return q1;
  else if constexpr (OFF < 16) return (q1 >> (8 * (OFF - 8))) | (q2 << (64 - 8 * (OFF - 8)));
  else return q2;
  }
// the 8 bytes behind the 8 at offset t + 3 (attempt t < KS): offset t + 11 .. t + 18 of the 24
__device__ __forceinline__ uint64_t next_bytes(uint32_t t, uint64_t q1, uint64_t q2)
  {
  // offset t + 11: t = 0 .. 4 -> inside q1 / q2, t = 5 -> q2
  return t < 5u ? (q1 >> (8u * (t + 3u))) | (q2 << (64u - 8u * (t + 3u))) : q2;
  }
// The parser.  One wave; `tab` (LDS, 4096 x u32) holds the table for the start state.
//   start_match_end: true  -> state "a match just ended at ip0" (anchor == ip0), table = tab
//                    false -> state "search loop starts at ip0" (anchor == ip0), table = tab
//   fresh: chunk 0 start (lz4.c:865-867: insert position 0, ip = 1)
//   emit_from_start: record descriptors from the first sequence on (chunk 0, re-parse); otherwise wait for
//                    the first match end >= c_lo, snapshot there, then record
//   big (>= the chunk's size, 0: none): a recorded match that is still running after `big` bytes is not counted to its end by this
//                    wave - it ends beyond c_hi, so it is the parse's last sequence whatever its length: the descriptor gets OPEN_ML,
//                    the table is saved as for any match end and the whole device counts the rest (k_lz4_extend, END_OPEN)
// Stops at the first match end >= c_hi (END_MATCH) or at the end of the block (END_FINAL).
template <int NW, int QL = 16>
__device__ void lz4_parse(const uint8_t* __restrict__ src, uint32_t n, uint32_t* tab, uint8_t* dup, uint32_t ip0, bool start_match_end, bool fresh,
                          bool emit_from_start, uint32_t c_lo, uint32_t c_hi, Desc* __restrict__ desc, uint32_t dcap,
                          Meta* __restrict__ meta, uint32_t* __restrict__ snapT, uint32_t* __restrict__ endT, int lane, int wave = 0,
                          uint32_t* xch = nullptr, uint32_t big = 0u, int ks0 = 4)
  {
  const bool writer = lane == 0 && wave == 0;            // descriptors and chunk meta: one writer (NW > 1: all waves hold the same values)
  const uint32_t mfl1 = n - 11u, mlim = n - 5u;                                    // lz4.c:825-826 (n >= 13 here)
  bool emit = emit_from_start;
  uint32_t nd = 0;
  uint32_t ip = ip0, anchor = ip0;
  uint32_t end_kind = END_NONE, end_ip = 0;
  uint32_t first_in = ip0;
  bool at_match_end = start_match_end;
  if (fresh)
    {
    if (lane == 0) tab[uni(hash5(src))] = 0u;
    ip = 1;
    anchor = 0;
    first_in = 0;
    at_match_end = false;
    }
  bool overflow = false;
  int ks_credit = ks0;                       // > 0: searches begin with scalar attempts (see the chain of short sequences below)
  for (;;)
    {
    // (the parse state is wave-uniform, but the compiler cannot see it through the cross-lane reads that produce it: told so here, it
    // keeps the state in scalar registers and the windows below become scalar loads)
    ip = uni(ip);
    anchor = uni(anchor);
    nd = uni(nd);
    uint32_t cand = 0;
    uint32_t t0_chain = 0;                 // attempts of the coming search the chain below has already made
    bool have_match = false;
    bool fast = false;                     // the search round has already seen where the match ends (see the search loop)
    uint32_t fast_extra = 0;               // ... namely this many bytes behind its first four
    bool fast_seen = false;                // ... and the difference that ends it was among the bytes at hand
    uint8_t tok_lit0 = 0;
    (void)tok_lit0;
#ifndef TRICO_LZ4_SM
#define TRICO_LZ4_SM 1
#endif
    // ---- a chain of sequences without literals, on the scalar unit ------------------------------------------------------------------
    // On a plane of short sequences (a mesh's second index plane: 43 M sequences of 7 bytes) nearly every match is followed at once
    // by the next one (lz4.c:1088-1138: insert ip - 2, test ip, literal length 0).  What such a step cost was not memory but
    // instructions: ~300 of them through the general code below, issued by a wave that has its SIMD almost to itself at 5-9 cycles
    // each.  Here the step is a loop of its own with nothing in it but the step: the 24 bytes at ip - 2 and the 24 at the candidate
    // through the scalar cache (read-only data at wave-uniform positions; the bytes at ip are read front to back, the candidates of
    // such a plane lie a few hundred bytes back), both hashes on the scalar unit, the three table accesses by one lane.  It ends -
    // leaving the state exactly as the general code expects it - at the first step that is not of this kind: no match at ip, a match
    // whose end is not among the 22 bytes at hand, the chunk's or the block's end near, a snapshot due.
    if (TRICO_LZ4_SM != 0 && at_match_end)
      {
      const bool emit_u = uni(emit ? 1u : 0u) != 0u;                                // (uniform, like the rest of the state: see the top of the outer loop)
      const uint32_t stop = uni(emit_u ? c_hi : c_lo), ndcap = emit_u ? uni(dcap) : 0xffffffffu, nmax = n >= 80u ? uni(n) - 80u : 0u;       // (nmax = 0: no step qualifies, ip >= 8)
      for (;;)
        {
        ip = uni(ip);
        nd = uni(nd);
        if (!(ip >= 8u && ip < stop && ip <= nmax && nd < ndcap))
          break;
        typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
        typedef __attribute__((address_space(4))) const u32x8 const_u32x8;
        const uintptr_t a = (uintptr_t)(src + ip - 2u);
        const u32x8 d = *(const_u32x8*)(a & ~(uintptr_t)3);
        const uint32_t sh = 8u * (uint32_t)(a & 3u);
        // dword i of the bytes from ip - 2 on: the low half of (d[i + 1] : d[i]) >> sh
        uint32_t w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
          w[i] = (uint32_t)((((uint64_t)d[i + 1] << 32) | d[i]) >> sh);
        const uint64_t q0 = ((uint64_t)w[1] << 32) | w[0], q1 = ((uint64_t)w[3] << 32) | w[2], q2 = ((uint64_t)w[5] << 32) | w[4];
        const uint64_t i0 = (q0 >> 16) | (q1 << 48), i1 = (q1 >> 16) | (q2 << 48), i2 = q2 >> 16;      // bytes ip .. ip + 21
        const uint32_t h2 = hash_w(q0), h = hash_w(i0);
        uint32_t cnd = 0;
        if (lane == 0)
          {
          tab[h2] = ip - 2u;                                                         // lz4.c:1088
          cnd = tab[h];                                                              // (the LDS unit keeps one wave's accesses in order)
          tab[h] = ip;
          }
        cnd = uni(cnd);
        bool hit = false;
        uint32_t extra = 0;
        bool seen = false;
        if (cnd + MAXD >= ip)
          {
          const uintptr_t ca = (uintptr_t)(src + cnd);
          const u32x8 e = *(const_u32x8*)(ca & ~(uintptr_t)3);
          const uint32_t csh = 8u * (uint32_t)(ca & 3u);
          uint32_t v[6];
#pragma unroll
          for (int i = 0; i < 6; ++i)
            v[i] = (uint32_t)((((uint64_t)e[i + 1] << 32) | e[i]) >> csh);
          const uint64_t x0 = (((uint64_t)v[1] << 32) | v[0]) ^ i0, x1 = (((uint64_t)v[3] << 32) | v[2]) ^ i1,
                         x2 = ((((uint64_t)v[5] << 32) | v[4]) ^ i2) & 0xffffffffffffull;
          hit = (uint32_t)x0 == 0u;
          seen = true;
          if (x0 >> 32)
            extra = (uint32_t)__builtin_ctzll(x0 >> 32) >> 3;
          else if (x1)
            extra = 4u + ((uint32_t)__builtin_ctzll(x1) >> 3);
          else if (x2)
            extra = 12u + ((uint32_t)__builtin_ctzll(x2) >> 3);
          else
            seen = false;
          }
        if (!hit)
          {
          // No match at ip: the search (lz4.c:898-956) from ip + 1, its first KS attempts here - on such a plane a sequence is a few
          // literals and a match, and a round of 64 attempts of the general code costs two vector round trips and ~400 instructions
          // whether it needs one attempt or sixty.  Attempt t looks at ip + 1 + t (the first 64 attempts advance by one), reads and
          // writes the table entry of its hash, tests the candidate; the first hit ends the search - nothing to take back.
          // Where searches are long (literal runs: the low plane of a regular mesh's indices) the scalar attempts are wasted on top
          // of the rounds that follow: a search that did not end among them costs three credits, one that did earns one, and
          // without credit the search goes straight to the general code (and earns one back, so that it is tried again later).
          ks_credit = (int)uni((uint32_t)ks_credit);
          if (ks_credit <= 0)
            {
            ++ks_credit;
            ++ip;
            at_match_end = false;
            break;
            }
          const uint32_t start = ip + 1u;
          const uint64_t wt[KS] = { bytes_at<3>(q0, q1, q2), bytes_at<4>(q0, q1, q2), bytes_at<5>(q0, q1, q2), bytes_at<6>(q0, q1, q2),
                                    bytes_at<7>(q0, q1, q2), bytes_at<8>(q0, q1, q2) };                // the 8 bytes at start + t
          const uint64_t pt[KS] = { q0 << 40, q0 << 32, q0 << 24, q0 << 16, q0 << 8, q0 };            // the t + 3 bytes before them, as the top of a qword
          bool found = false;
          uint32_t fpos = 0, fback = 0, fextra = 0;
          bool fseen = false;
#pragma unroll
          for (uint32_t t = 0; t < KS; ++t)
            {
            if (found)
              break;
            const uint32_t pos = start + t;
            const uint32_t ht = hash_w(wt[t]);
            uint32_t c2 = 0;
            if (lane == 0)
              {
              c2 = tab[ht];
              tab[ht] = pos;
              }
            c2 = uni(c2);
            if (c2 + MAXD >= pos && c2 >= 8u)
              {
              // the bytes c2 - 8 .. c2 + 23 of the candidate: one s_load_dwordx16 from the dword below c2 - 8
              typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
              typedef __attribute__((address_space(4))) const u32x16 const_u32x16;
              const uintptr_t ca = (uintptr_t)(src + c2 - 8u);
              const u32x16 e = *(const_u32x16*)(ca & ~(uintptr_t)3);
              const uint32_t csh = 8u * (uint32_t)(ca & 3u);
              uint32_t v[8];
#pragma unroll
              for (int i = 0; i < 8; ++i)
                v[i] = (uint32_t)((((uint64_t)e[i + 1] << 32) | e[i]) >> csh);
              const uint64_t cpre = ((uint64_t)v[1] << 32) | v[0], c0 = ((uint64_t)v[3] << 32) | v[2], c1 = ((uint64_t)v[5] << 32) | v[4];
              const uint64_t x0 = c0 ^ wt[t];
              if ((uint32_t)x0 == 0u)
                {
                found = true;
                fpos = pos;
                cnd = c2;
                // catch-up (lz4.c:960-961): equal bytes right before the two positions, at most down to the anchor (= ip)
                const uint64_t dp = cpre ^ pt[t];
                const uint32_t eqb = dp ? (uint32_t)__builtin_clzll(dp) >> 3 : 8u;
                fback = eqb < t + 1u ? eqb : t + 1u;
                // the match's end, if it is among the bytes at hand: the 8 bytes at pos, and the 8 behind them where the window has them
                fseen = true;
                if (x0 >> 32)
                  fextra = (uint32_t)__builtin_ctzll(x0 >> 32) >> 3;
                else
                  {
                  const uint64_t x1 = c1 ^ next_bytes(t, q1, q2);
                  if (x1)
                    fextra = 4u + ((uint32_t)__builtin_ctzll(x1) >> 3);
                  else
                    fseen = false;
                  }
                }
              }
            else if (c2 + MAXD >= pos)
              {
              // (a candidate in the first eight bytes of the block: the general code makes this attempt, and the ones behind it)
              if (lane == 0)
                tab[ht] = c2;                                                        // taken back: it is made again
              t0_chain = t;
              fpos = 0xffffffffu;
              break;
              }
            }
          if (fpos == 0xffffffffu || !found)
            {
            ip = start;                                                              // the search goes on in the general code: attempt t0_chain
            at_match_end = false;
            if (fpos != 0xffffffffu)
              t0_chain = KS;
            ks_credit = ks_credit > -13 ? ks_credit - 3 : -16;
            break;
            }
          ks_credit = ks_credit < 8 ? ks_credit + 1 : 8;
          const uint32_t mip = fpos - fback, mcand = cnd - fback;
          if (!fseen)
            {
            ip = mip;                                                                // a longer match: counted by the general code
            cand = mcand;
            have_match = true;
            at_match_end = false;
            break;
            }
          const uint32_t ml = 4u + fback + fextra;
          if (emit_u)
            {
            if (writer) { desc[nd].lit = mip - ip; desc[nd].ml = ml; desc[nd].off = mip - mcand; }
            ++nd;
            }
          ip = uni(mip + ml);
          anchor = ip;
          continue;
          }
        if (!seen)
          {
          cand = cnd;                                                                // a longer match: counted by the general code
          have_match = true;
          at_match_end = false;
          break;
          }
        if (emit_u)
          {
          if (writer) { desc[nd].lit = 0u; desc[nd].ml = extra + 4u; desc[nd].off = ip - cnd; }
          ++nd;
          }
        ip += uni(extra) + 4u;
        anchor = ip;
        }
      }
    if (at_match_end)
      {
      // ---- state: a match ended at ip, anchor == ip ----
      if (!emit && ip >= c_lo)
        {
        // snapshot (speculative chunk): from here on sequences are recorded
        for (int i = lane; i < 4096; i += 64)
          snapT[i] = tab[i];
        if (lane == 0) { meta->snap_valid = 1u; meta->snap_ip = ip; }
        emit = true;
        first_in = ip;
        }
      if (emit && ip >= c_hi)
        {
        end_kind = END_MATCH;
        end_ip = ip;
        break;
        }
      if (!emit && ip >= c_hi)
        break;                                                                     // useless speculation: no match end inside the chunk
      if (ip >= mfl1)                                                              // lz4.c:1085 "Test end of chunk"
        {
        // final literals follow
        if (emit)
          {
          if (nd < dcap) { if (writer) { desc[nd].lit = n - anchor; desc[nd].ml = 0; desc[nd].off = 0; } ++nd; }
          else overflow = true;
          end_kind = END_FINAL;
          end_ip = n;
          }
        break;
        }
      if (lane == 0) tab[uni(hash5(src + ip - 2))] = ip - 2u;                      // lz4.c:1088
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint32_t h = uni(hash5(src + ip));
      cand = uni(tab[h]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane == 0) tab[h] = ip;
      // (8 bytes at both places instead of 4: ip < n - 11 and cand < ip, so both reads stay inside the block; a match that ends
      // within them - the usual case on a plane of short sequences - then needs no counting round trip, like the hits of the search)
      const uint64_t wcand = ld64(src + cand), wip = ld64(src + ip);
      if (cand + MAXD >= ip && uni((uint32_t)wcand) == uni((uint32_t)wip))
        {
        have_match = true;                                                         // lz4.c:1101-1138: literal length 0, no catch-up
        const uint32_t dx = uni((uint32_t)((wcand ^ wip) >> 32));
        if (dx)
          {
          fast = true;
          fast_seen = true;
          fast_extra = (uint32_t)__builtin_ctz(dx) >> 3;                           // equal bytes behind the first four: 0 .. 3
          }
        }
      else
        ++ip;
      at_match_end = false;
      }
    if (!have_match)
      {
      // ---- search loop (lz4.c:898-956), 64 attempts at a time ----
      // Attempt t of a search looks at start + off(t), off(0) = 0, off(t) = 1 + sum_{m < 63 + t} (m >> 6): the positions do not
      // depend on what is found (lz4.c:907-913: step = searchMatchNb++ >> LZ4_skipTrigger).  One attempt costs the reference a
      // hash of 8 source bytes, a table read and write and, if the candidate is in range, a 4-byte comparison; one wave pays
      // a global round trip of ~1 us for each of them, which is what bounded both the parse of planes that compress badly
      // and the re-parses of the stitch pass.  Here lane l evaluates attempt t0 + l: all hashes and candidate comparisons
      // are one round trip each.  Sequential semantics (attempt l sees the table writes of attempts < l) hold if the 64
      // hashes are distinct, which a byte-sized LDS scoreboard checks; otherwise one attempt is taken the serial way.
      // The first lane that ends the search (match, or the next position beyond mflimitPlusOne) decides; the table
      // writes of the attempts before it (and its own, for a match) are committed.
      const uint32_t start = ip;
      uint32_t t0 = t0_chain;
      bool final = false;
      // Short sequences (a mesh's second index plane: 7 bytes each) spent two more memory round trips per sequence on the
      // catch-up and on counting a match of 4-7 bytes.  The round that tests the candidates fetches 8 bytes at the candidate
      // instead of 4 and the 8 bytes before both positions, so the winner usually knows both answers already.
      bool fast_pre = false;
      uint32_t fast_eqb = 0;
      for (;;)
        {
        const uint32_t ta = t0 + (uint32_t)lane, tb = ta + 1u;
        const uint32_t xa = 63u + ta, xb = 63u + tb;
        const uint32_t offa = ta ? 1u + 32u * (xa >> 6) * ((xa >> 6) - 1u) + (xa >> 6) * (xa & 63u) : 0u;
        const uint32_t offb = 1u + 32u * (xb >> 6) * ((xb >> 6) - 1u) + (xb >> 6) * (xb & 63u);
        const uint32_t pos = start + offa, nxt = start + offb;
        const bool active = pos <= mfl1;                                           // else an earlier lane ends the search
        const bool fin = active && nxt > mfl1;
        uint64_t w = 0;
        if (active)
          w = ld64(src + pos);
        const uint32_t h = (uint32_t)(((w << 24) * 889523592379ull) >> 52);        // hash5 (lz4.c:643-648)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t cnd = 0;
        uint64_t later_same = 0;                                                   // higher lanes of this round with my hash
        if (!dup)
          {
          // Attempt l reads the table entry of its hash and writes its position there, and sees what the attempts before it wrote
          // (lz4.c:917-922): that is ONE exchange, because the LDS unit applies the active lanes of a ds_wrxchg_rtn_b32 in lane
          // order (tested on the device before this path is taken, lds_lane_order_ok()).  No scoreboard, no ballots per hash bit;
          // what the attempts behind the deciding one wrote is taken back below.
          if (active)
            cnd = __hip_atomic_exchange(&tab[h], pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
        else
          {
        if (active)
          dup[h] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const bool clash = active && dup[h] != (uint8_t)lane;
        cnd = active ? tab[h] : 0u;
        if (__ballot(clash))
          {
          // Attempts of this round share a hash.  In the reference attempt l sees what the attempts before it wrote: its
          // candidate is the position of the nearest lower lane with its hash (else the table's), and of the attempts that do
          // write, the highest of a hash is what stays in the table.  The lanes of a hash find each other with one ballot per
          // hash bit.  (Round 2a took ONE attempt the serial way here and batched again: on a plane of short sequences over
          // a small alphabet that was most rounds.)
          uint64_t same = __ballot(active);
#pragma unroll
          for (int b = 0; b < 12; ++b)
            {
            const bool bit = (h >> b) & 1u;
            const uint64_t m = __ballot(bit);
            same &= bit ? m : ~m;
            }
          if (!active)
            same = 0;
          const uint64_t lower = same & ((1ull << lane) - 1ull);
          const uint32_t from = lower ? 63u - (uint32_t)__builtin_clzll(lower) : (uint32_t)lane;
          const uint32_t ppos = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)pos);
          if (lower)
            cnd = ppos;
          later_same = same & ~((2ull << lane) - 1ull);
          }
          }
        const uint32_t found = cnd;                                                // what the table held when my turn came
        bool hit = false, pre = false;
        uint64_t cw = 0, pw = 0, pcw = 0;
        if (active && !fin && cnd + MAXD >= pos)
          {
          cw = ld64(src + cnd);                                                    // cnd < pos <= n - 11: inside the block
          hit = (uint32_t)cw == (uint32_t)w;
          if (pos >= 8u && cnd >= 8u)
            {
            pw = ld64(src + pos - 8u);
            pcw = ld64(src + cnd - 8u);
            pre = true;
            }
          }
        const uint64_t stop = __ballot(fin || hit);
        const int first = stop ? __builtin_ctzll(stop) : 64;
        const bool first_fin = stop && ((__ballot(fin) >> first) & 1ull);
        // table writes: every attempt before the deciding one, and the deciding one too unless it is the final one
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!dup)
          {
          // exchange mode: all attempts have written.  The positions in an entry only grow (the parse moves forward), so the minimum
          // with what each attempt behind the deciding one found restores the entry as the last attempt that counts left it.
          if (active && stop && (lane > first || (lane == first && first_fin)))
            atomicMin(&tab[h], found);
          }
        else
          {
          const int lastw = !stop ? 63 : (first_fin ? first - 1 : first);          // highest lane that writes
          const uint64_t writers = lastw >= 63 ? ~0ull : (lastw < 0 ? 0ull : (2ull << lastw) - 1ull);
          if (active && (lane < first || (lane == first && !first_fin)) && (later_same & writers) == 0ull)
            tab[h] = pos;
          }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (stop)
          {
          ip = (uint32_t)__builtin_amdgcn_readlane((int)pos, first);
          cand = (uint32_t)__builtin_amdgcn_readlane((int)cnd, first);
          final = first_fin;
          if (!first_fin)
            {
            const uint64_t dx = (w ^ cw) >> 32, dp = pw ^ pcw;
            const uint32_t extra = dx ? (uint32_t)__builtin_ctzll(dx) >> 3 : 4u;     // equal bytes behind the first four
            const uint32_t eqb = dp ? (uint32_t)__builtin_clzll(dp) >> 3 : 8u;       // equal bytes right before the two positions
            fast = true;
            fast_seen = fast_extra_seen(dx, first);
            fast_extra = (uint32_t)__builtin_amdgcn_readlane((int)extra, first);
            fast_eqb = (uint32_t)__builtin_amdgcn_readlane((int)eqb, first);
            fast_pre = __builtin_amdgcn_readlane((int)pre, first) != 0;
            }
          break;
          }
        t0 += 64u;
        }
      if (final)
        {
        if (emit)
          {
          if (nd < dcap) { if (writer) { desc[nd].lit = n - anchor; desc[nd].ml = 0; desc[nd].off = 0; } ++nd; }
          else overflow = true;
          end_kind = END_FINAL;
          end_ip = n;
          }
        break;
        }
      // catch up (lz4.c:960-961)
      const uint32_t maxback = ip - anchor < cand ? ip - anchor : cand;
      uint32_t back = 0;
      if (fast && maxback != 0u)
        {
        if (!fast_pre || (fast_eqb == 8u && maxback > 8u))
          fast = false;                                                            // not decided by the 8 bytes at hand
        else
          back = fast_eqb < maxback ? fast_eqb : maxback;
        }
      while (!fast && back < maxback)
        {
        const uint32_t k = back + (uint32_t)lane + 1u;
        const bool eq = k <= maxback && src[ip - k] == src[cand - k];
        const uint64_t ne = ~__ballot(eq);
        if (ne)
          {
          back += (uint32_t)__builtin_ctzll(ne);
          break;
          }
        back += 64u;
        }
      if (back > maxback) back = maxback;
      ip -= back;
      cand -= back;
      fast_extra += back;                                                          // equal bytes behind the first four of the moved match
      fast = fast && fast_seen;                                                    // (the difference that ends the match was among the bytes at hand)
      }
    // ---- a match starts at ip against cand (lz4.c:1007-1077) ----
    if (!emit && ip + 4u >= c_hi)
      break;                                                                       // warm-up ran past the chunk: useless speculation
    const uint32_t room = mlim > ip + 4u ? mlim - (ip + 4u) : 0u;
    uint32_t limit = room;
    if (!emit && ip + 4u < c_hi)
      {
      // warm-up: a match running past the chunk end makes this speculation useless; don't count further
      const uint32_t cap = c_hi - (ip + 4u);
      if (cap < limit) limit = cap;
      }
    const bool may_open = emit && big != 0u && big < limit;
    if (may_open)
      limit = big;
    const uint32_t m = (fast && fast_extra < limit) ? fast_extra : wave_count<NW, QL>(src + ip + 4u, src + cand + 4u, limit, lane, wave, xch);
    if (!emit && limit < room && m >= limit)
      break;                                                                       // ran past c_hi during warm-up
    if (may_open && m >= limit)
      {
      // still equal after `big` bytes
      if (nd < dcap) { if (writer) { desc[nd].lit = ip - anchor; desc[nd].ml = OPEN_ML; desc[nd].off = ip - cand; } ++nd; }
      else { overflow = true; break; }
      end_kind = END_OPEN;
      end_ip = ip;
      break;
      }
    if (emit)
      {
      if (nd < dcap) { if (writer) { desc[nd].lit = ip - anchor; desc[nd].ml = m + 4u; desc[nd].off = ip - cand; } ++nd; }
      else { overflow = true; break; }
      }
    ip += m + 4u;
    anchor = ip;
    at_match_end = true;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (end_kind == END_MATCH || end_kind == END_OPEN)
    for (int i = lane; i < 4096; i += 64)
      endT[i] = tab[i];
  if (lane == 0)
    {
    meta->end_kind = overflow ? END_NONE : end_kind;
    meta->end_ip = end_ip;
    meta->ndesc = nd;
    meta->first_in = first_in;
    }
  }

struct Geom { uint32_t n, chunk, warm, K, dcap; size_t plane_stride; uint32_t alt_rounds, alt_dcap, xchg; uint8_t order[8]; };   // xchg: search rounds through one LDS exchange (lz4_parse); order: the planes as the parse pass takes them, slowest first
constexpr uint32_t ALT_R = 4;             // alternative parses kept per chunk (k_lz4_alt)
// equal bytes after which a recorded match of the parse pass is left to k_lz4_extend: at least a chunk (so that the match is the
// chunk's last sequence), at least 1 MiB (what a wave counts in ~0.1 ms)
__host__ __device__ inline uint32_t open_after(const Geom& g) { return g.chunk > (1u << 20) ? g.chunk : (1u << 20); }

// (count rounds of 8 KiB, not 16: with the 280 registers of the wider round a compute unit holds four chunks instead of ten, and what a
// plane of short sequences costs is how many of its chunks run at once; matches of a MiB and more are k_lz4_extend's anyway)
// Scalar attempts at the start of a search (lz4_parse: ks0) only in the short-sequence geometry: in the long-match geometry (the one
// with alternative parses) a search is a literal run, the attempts are wasted and the code around them costs the match chains time
// (measured on the benchmark mesh's planes: parse pass 4.97 ms without, 5.37 with credit gating alone, 5.76 ungated).
__device__ __forceinline__ int scalar_search(const Geom& g) { return g.alt_rounds != 0u ? -(1 << 30) : 4; }

// Where the speculative parse of chunk k begins: `warm` bytes in front of the chunk, but never at position 0 - an empty table says
// "candidate = position 0" for every hash, and position 0 tested against itself is a match the reference never makes (it enters its
// loop at position 1 with position 0 in the table, lz4.c:866-867: exactly the state a speculative start at 1 has).
__device__ __forceinline__ uint32_t spec_start(uint32_t k, uint32_t c_lo, uint32_t warm)
  {
  if (k == 0u)
    return 0u;
  return c_lo > warm ? c_lo - warm : 1u;
  }

__global__ void __launch_bounds__(64) k_lz4_parse(const uint8_t* __restrict__ planes, Geom g, Desc* __restrict__ descs, Meta* __restrict__ metas,
                                                  uint32_t* __restrict__ snapTs, uint32_t* __restrict__ endTs)
  {
  // (dynamic: 16 KiB of table, and 4 KiB of scoreboard for the batched search loop where the exchange is not used - ten chunks per
  // compute unit instead of eight)
  extern __shared__ __attribute__((aligned(16))) uint32_t parse_lds[];
  uint32_t* tab = parse_lds;
  uint8_t* dup = (uint8_t*)(parse_lds + 4096);
  const int lane = threadIdx.x;
  // the chunks of the plane with the shortest sequences are dispatched first: they take longest, and what decides the pass is when
  // the last of them starts
  const uint32_t k = blockIdx.x, p = g.order[blockIdx.y];
  for (int i = lane; i < 4096; i += 64)
    tab[i] = 0u;
  __syncthreads();
  const size_t ck = (size_t)p * g.K + k;
  Meta* meta = metas + ck;
  if (lane == 0)
    {
    meta->snap_valid = 0; meta->snap_ip = 0; meta->end_kind = END_NONE; meta->end_ip = 0; meta->ndesc = 0; meta->accepted = 0;
    meta->first_in = 0; meta->reparsed = 0;
    }
  const uint8_t* src = planes + (size_t)p * g.plane_stride;
  const uint32_t c_lo = k * g.chunk;
  const uint32_t c_hi = (k + 1u == g.K) ? 0xffffffffu : c_lo + g.chunk;
  lz4_parse<1, 8>(src, g.n, tab, g.xchg ? nullptr : dup, spec_start(k, c_lo, g.warm), false, k == 0, k == 0, c_lo, c_hi, descs + ck * g.dcap, g.dcap, meta,
                  snapTs + ck * 4096, endTs + ck * 4096, lane, 0, nullptr, open_after(g), scalar_search(g));
  }

// ---- matches longer than a chunk: counted by the whole device ------------------------------------------------------------------
// One wave counts equal bytes at ~8 GB/s (16 KiB per round trip); the upper byte planes of mesh indices are a handful of runs of
// tens to hundreds of MB (a plane of zeros is ONE match of 300 MB: 36 ms for the wave of chunk 0, three quarters of the parse pass
// on u64 indices).  A parse that meets such a match stops with END_OPEN (lz4_parse); here the workgroups of a plane share its open
// matches: each match gets an equal share of them, a share takes the 64 KiB slices behind the counted part in turns and stops at
// the first slice that starts behind the best end known (atomicMin on the descriptor's length).  k_lz4_extend_done turns the result
// into the END_MATCH state the parse would have left.
constexpr uint32_t EXT_SLICE = 65536, EXT_G = 256;

__global__ void __launch_bounds__(256) k_lz4_extend(const uint8_t* __restrict__ planes, Geom g, Desc* __restrict__ descs, const Meta* __restrict__ metas)
  {
  __shared__ uint32_t open_list[1024], nopen, wmin[4], known;
  const uint32_t p = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint8_t* src = planes + (size_t)p * g.plane_stride;
  const Meta* pm = metas + (size_t)p * g.K;
  const uint32_t big = open_after(g), mlim = g.n - 5u;
  for (uint32_t k0 = 0; k0 < g.K; k0 += 1024u)                   // (a plane has a few hundred chunks: one batch)
    {
    // the open chunks of this batch in chunk order - the same list in every workgroup, which is what shares the work out
    {
    uint32_t mine[4], cnt = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q)
      {
      const uint32_t k = k0 + 4u * tid + q;
      mine[q] = (k < g.K && pm[k].end_kind == END_OPEN) ? 1u : 0u;
      cnt += mine[q];
      }
    uint32_t incl = cnt;                                        // inclusive scan over the wave, then over the four waves
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1)
      {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, s2);
      if ((int)lane >= s2) incl += up;
      }
    if (lane == 63u) wmin[wave] = incl;
    __syncthreads();
    uint32_t at = incl - cnt;
    for (uint32_t w2 = 0; w2 < wave; ++w2)
      at += wmin[w2];
    if (tid == 255u) nopen = at + cnt;
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q)
      if (mine[q])
        open_list[at++] = k0 + 4u * tid + q;
    __syncthreads();
    }
    const uint32_t C = nopen;
    if (C != 0u)
      {
      const uint32_t share = C <= gridDim.x ? gridDim.x / C : 1u;                  // workgroups per open match
      for (uint32_t i = blockIdx.x / share; i < C; i += (gridDim.x + share - 1u) / share)
        {
        if (C <= gridDim.x && blockIdx.x >= C * share)
          break;                                                                    // (the remainder of the division has nothing to do)
        const uint32_t k = open_list[i], rank = blockIdx.x % share;
        const size_t ck = (size_t)p * g.K + k;
        Desc* d = descs + ck * g.dcap + (pm[k].ndesc - 1u);
        const uint32_t ip = pm[k].end_ip, off = d->off;
        const uint8_t* a = src + ip + 4u;
        const uint8_t* b = a - off;
        const uint32_t room = mlim - (ip + 4u);                                     // equal bytes the match may have behind its first four
        for (uint32_t sl = rank;; sl += share)
          {
          const uint64_t o0 = (uint64_t)big + (uint64_t)sl * EXT_SLICE;
          if (o0 >= room)
            break;
          if (tid == 0)
            known = __hip_atomic_load(&d->ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __syncthreads();
          const uint32_t best = known;                                              // (one value for the whole workgroup: it decides a barrier)
          __syncthreads();
          if (best != OPEN_ML && (uint64_t)best <= o0 + 4u)
            break;                                                                  // the match ends before this slice
          const uint32_t lo = (uint32_t)o0, hi = room - lo < EXT_SLICE ? room : lo + EXT_SLICE;
          uint32_t first = 0xffffffffu;
          for (uint32_t o = lo + 16u * tid; o < hi; o += 16u * 256u)
            {
            if (o + 16u <= hi)
              {
              const u32x4 x = ld128(a + o), y = ld128(b + o);
              const uint64_t d0 = ((uint64_t)(x.y ^ y.y) << 32) | (x.x ^ y.x), d1 = ((uint64_t)(x.w ^ y.w) << 32) | (x.z ^ y.z);
              if (d0 | d1)
                first = min(first, o + (d0 ? (uint32_t)__builtin_ctzll(d0) >> 3 : 8u + ((uint32_t)__builtin_ctzll(d1) >> 3)));
              }
            else
              {
              uint32_t q = o;
              while (q < hi && a[q] == b[q]) ++q;
              if (q < hi)
                first = min(first, q);
              }
            }
#pragma unroll
          for (int s2 = 32; s2 > 0; s2 >>= 1)
            first = min(first, (uint32_t)__shfl_xor((int)first, s2));
          if (lane == 0) wmin[wave] = first;
          __syncthreads();
          first = min(min(wmin[0], wmin[1]), min(wmin[2], wmin[3]));
          __syncthreads();
          if (first != 0xffffffffu)
            {
            if (tid == 0)
              atomicMin(&d->ml, first + 4u);
            break;
            }
          }
        }
      }
    __syncthreads();
    }
  }

// one thread per chunk: an open match that nobody found the end of runs to the limit; the chunk ends where the match does
__global__ void __launch_bounds__(256) k_lz4_extend_done(Geom g, Desc* __restrict__ descs, Meta* __restrict__ metas)
  {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x, p = blockIdx.y;
  if (k >= g.K)
    return;
  Meta* m = metas + (size_t)p * g.K + k;
  if (m->end_kind != END_OPEN)
    return;
  Desc* d = descs + ((size_t)p * g.K + k) * g.dcap + (m->ndesc - 1u);
  const uint32_t ip = m->end_ip, room = (g.n - 5u) - (ip + 4u);
  uint32_t ml = d->ml;
  if (ml == OPEN_ML || ml > room + 4u)
    ml = room + 4u;
  d->ml = ml;
  m->end_ip = ip + ml;
  m->end_kind = END_MATCH;
  }

// ---- which geometry?  -----------------------------------------------------------------------------------------------------
// A plane of long matches (the grid's index planes: a few dozen sequences per chunk) is bound by the serial stitch pass, whose
// re-parses grow with the number of chunks and shrink with the warm-up; a plane of short sequences (a real mesh's second
// index plane: one sequence per 7 bytes) is bound by the ~1.5 us a wave needs per sequence, i.e. by how many chunk waves the
// GPU holds at once, and its table state converges within a few KiB.  Measured on the MI355X (4 x 300 MB planes):
//     chunk / warm-up        grid      walk                      (round 3, four chunks per compute unit)
//     1 MiB / 384 KiB        70 ms    294 ms
//     384 KiB / 96 KiB      136 ms    164 ms
// Round 5, ten chunks per compute unit (count rounds of 8 KiB: 126 registers instead of 280) and the densest plane's chunks
// dispatched first: walk 384 KiB / 96 KiB 100 ms, 256 KiB / 70,000 B 74 ms, 192 KiB / 70,000 B 63 ms, 128 KiB / 70,000 B 79 ms
// (chunk sizes that are not multiples of 64 KiB: ~20 % slower than their neighbours); before: 88 ms at 384 KiB / 96 KiB.
// So the launcher looks first: PROBE_S windows of PROBE_W bytes per plane are parsed as stand-alone blocks, and the densest
// plane's bytes per sequence decide.  The choice never changes the output, only the time.
constexpr uint32_t PROBE_S = 8, PROBE_W = 4096, PROBE_DCAP = PROBE_W / 4 + 16;

__global__ void __launch_bounds__(64) k_lz4_probe(const uint8_t* __restrict__ planes, uint32_t n, size_t plane_stride, Desc* __restrict__ descs,
                                                  Meta* __restrict__ metas, uint32_t xchg)
  {
  __shared__ uint32_t tab[4096];
  __shared__ uint8_t dup[4096];
  const int lane = threadIdx.x;
  const uint32_t s = blockIdx.x, p = blockIdx.y;
  for (int i = lane; i < 4096; i += 64)
    tab[i] = 0u;
  __syncthreads();
  const size_t pw = (size_t)p * PROBE_S + s;
  Meta* meta = metas + pw;
  if (lane == 0)
    {
    meta->snap_valid = 0; meta->snap_ip = 0; meta->end_kind = END_NONE; meta->end_ip = 0; meta->ndesc = 0; meta->accepted = 0;
    meta->first_in = 0; meta->reparsed = 0;
    }
  const uint32_t section = n / PROBE_S;                                     // n >= the chunked threshold (MiB)
  const uint32_t w0 = s * section + (section - PROBE_W) / 2u;
  lz4_parse<1>(planes + (size_t)p * plane_stride + w0, PROBE_W, tab, xchg ? nullptr : dup, 0u, false, true, true, 0u, 0xffffffffu, descs + pw * PROBE_DCAP,
            PROBE_DCAP, meta, nullptr, nullptr, lane);
  }

// The stitch pass below is one serial walk per plane.  Two things it would do chunk by chunk are done here for all chunks at
// once, in ALT_R rounds of one launch each:
//   * round 1 compares every chunk's snapshot with the SPECULATIVE end state of the chunk before it (entry by entry, or both
//     entries beyond the 64 KiB window: lz4.c:945-948) and sets bit 0 of agree[chunk] when they match: the walk accepts such a
//     chunk without touching the tables if the chunk before it was accepted as parsed;
//   * a chunk that does not match is parsed again right here, from that end state, into alternative slot 1 of the chunk (a
//     small descriptor buffer: planes that need re-parses are planes of long matches).  If the chunk before it was accepted
//     as parsed, that state is the true state and the walk adopts the alternative instead of re-parsing.
//   Round r > 1 does the same against the end state that round r - 1 produced for the chunk before (alternative slot r - 1):
//   agree bit r - 1, alternative slot r.  The grid's periodic index plane rejects chunks in pairs (the second one still does
//   not fit the re-parse of the first): 143 serial re-parses of 0.29 ms became two parallel rounds.
//   An alternative is only ever used when the walk arrives at the chunk with exactly the state it was parsed from, so junk
//   rounds (parsed from a state that turns out not to be the true one) cost time, never correctness.
__global__ void __launch_bounds__(64) k_lz4_alt(const uint8_t* __restrict__ planes, Geom g, uint32_t r, const Meta* __restrict__ metas,
                                                const uint32_t* __restrict__ snapTs, const uint32_t* __restrict__ endTs,
                                                Desc* __restrict__ altDescs, Meta* __restrict__ altMetas, uint32_t* __restrict__ altEndTs,
                                                uint32_t* __restrict__ agree)
  {
  __shared__ uint32_t tab[4096];
  __shared__ uint8_t dup[4096];
  const int lane = threadIdx.x;
  const uint32_t j = blockIdx.x, p = blockIdx.y;
  if (j == 0)
    return;
  const size_t ci = (size_t)p * g.K + j - 1, cj = ci + 1;
  Meta a;
  const uint32_t* T;
  if (r == 1)
    {
    a = metas[ci];
    T = endTs + ci * 4096;
    }
  else
    {
    a = altMetas[ci * ALT_R + (r - 2)];
    if (!a.snap_valid)
      return;
    T = altEndTs + (ci * ALT_R + (r - 2)) * 4096;
    }
  if (a.end_kind != END_MATCH)
    return;
  const uint32_t ip = a.end_ip;
  uint32_t jj = ip / g.chunk;
  if (jj >= g.K) jj = g.K - 1u;
  if (jj != j)
    return;                                          // the walk goes elsewhere from there: left to it
  const Meta b = metas[cj];
  bool ok = b.snap_valid != 0u && b.snap_ip == ip && b.end_kind != END_NONE;
  if (ok)
    {
    const uint32_t* snT = snapTs + cj * 4096;
    bool same = true;
    for (int i = lane; i < 4096; i += 64)
      {
      const uint32_t x = T[i], y = snT[i];
      same = same && (x == y || (x + MAXD < ip && y + MAXD < ip));
      }
    ok = __ballot(!same) == 0ull;
    }
  if (ok)
    {
    if (lane == 0) atomicOr(&agree[cj], 1u << (r - 1));
    return;
    }
  if (r > g.alt_rounds)
    return;
  // alternative parse of chunk j from that state, into slot r
  for (int i = lane; i < 4096; i += 64)
    tab[i] = T[i];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const size_t slot = cj * ALT_R + (r - 1);
  Meta* am = altMetas + slot;
  const uint32_t c_hi = (j + 1u == g.K) ? 0xffffffffu : (j + 1u) * g.chunk;
  lz4_parse<1>(planes + (size_t)p * g.plane_stride, g.n, tab, g.xchg ? nullptr : dup, ip, true, false, true, j * g.chunk, c_hi, altDescs + slot * g.alt_dcap,
            g.alt_dcap, am, nullptr, altEndTs + slot * 4096, lane, 0, nullptr, 0u, scalar_search(g));
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane == 0 && am->end_kind != END_NONE)
    am->snap_valid = 1u;                             // the slot holds a finished parse (END_NONE: descriptor buffer too small)
  }

// One workgroup per plane walks the chain: accept speculative chunks whose snapshot is equivalent to the true state, re-parse the
// others.  Re-parsing is serial by nature (the next chunk needs this one's end state) and on periodic planes it consists of a
// few dozen matches of tens of KiB per chunk, i.e. of counting equal bytes: STITCH_W waves run the same walk and the same
// re-parse redundantly, each with a private copy of the table in LDS, and split every long count among themselves
// (wave_count<NW>).  All decisions are functions of the same inputs, so the waves stay in step; the two barriers per
// chain step keep a faster wave from rewriting a chunk's meta words before a slower one has read them.
constexpr int STITCH_W = 1;

__global__ void __launch_bounds__(64 * STITCH_W) k_lz4_stitch(const uint8_t* __restrict__ planes, Geom g, Desc* __restrict__ descs,
                                                              Meta* __restrict__ metas, uint32_t* __restrict__ snapTs,
                                                              uint32_t* __restrict__ endTs, const uint32_t* __restrict__ agree,
                                                              const Meta* __restrict__ altMetas, const uint32_t* __restrict__ altEndTs,
                                                              uint32_t* __restrict__ status)
  {
  __shared__ uint32_t tabs[STITCH_W][4096];
  __shared__ uint8_t dups[STITCH_W][4096];
  __shared__ uint32_t xch[64];
  const int lane = threadIdx.x & 63;
  const int wave = (int)uni(threadIdx.x >> 6);
  uint32_t* tab = tabs[wave];
  const uint32_t p = blockIdx.x;
  const uint8_t* src = planes + (size_t)p * g.plane_stride;
  Meta* pm = metas + (size_t)p * g.K;
  uint32_t cur = 0;                               // last accepted chunk
  // where its true parse is: 0 = the parse pass's own (accepted as parsed), r = alternative slot r (k_lz4_alt), VER_SERIAL =
  // re-parsed here into the chunk's own buffers
  constexpr uint32_t VER_SERIAL = 99u;
  uint32_t ver = 0;
  if (threadIdx.x == 0) pm[0].accepted = 1u;
  for (uint32_t guard = 0; guard < g.K + 2u; ++guard)
    {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __syncthreads();
    if (ver == 0u)
      {
      // Runs of chunks whose speculation holds, 64 at a time (round 6: one chunk per step was 1,563 steps of three dependent memory
      // round trips per plane of short sequences: 4.6 ms).  Lane l looks at chunk cur + 1 + l: it is the chain's next chunk with its
      // parse accepted as it stands iff its predecessor is (lane l - 1; lane 0: cur, which is), the predecessor's parse ended with a
      // match end inside this chunk, this chunk's snapshot was taken exactly there, and k_lz4_alt found its table equal to the
      // predecessor's speculative end state (bit 0 of agree) - the very tests of the single step below for ver = 0.
      const uint32_t lane64 = threadIdx.x & 63u;
      const uint32_t jl = cur + 1u + lane64;
      bool good = false;
      if (jl < g.K)
        {
        const Meta prev = pm[jl - 1u], me = pm[jl];
        good = prev.end_kind == END_MATCH && prev.end_ip / g.chunk == jl && me.snap_valid != 0u && me.snap_ip == prev.end_ip &&
               me.end_kind != END_NONE && (agree[(size_t)p * g.K + jl] & 1u) != 0u;
        }
      const uint64_t gm = __ballot(good);
      const uint32_t run = gm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~gm);
      if (run != 0u)
        {
        if (threadIdx.x < run)
          pm[jl].accepted = 1u;
        if (threadIdx.x == 0)
          atomicAdd(status + 1, run);
        cur += run;
        continue;
        }
      }
    const uint32_t kind = uni(pm[cur].end_kind), ip = uni(pm[cur].end_ip);
    if (kind == END_FINAL)
      return;
    if (kind != END_MATCH)
      {
      if (threadIdx.x == 0) atomicOr(status, 16u);      // descriptor overflow / parser did not finish: must not happen
      return;
      }
    uint32_t j = ip / g.chunk;
    if (j >= g.K) j = g.K - 1u;
    if (j <= cur)
      {
      if (threadIdx.x == 0) atomicOr(status, 32u);      // no forward progress: must not happen
      return;
      }
    const size_t ccur = (size_t)p * g.K + cur, cj = (size_t)p * g.K + j;
    const uint32_t* curT = (ver >= 1u && ver <= ALT_R) ? altEndTs + (ccur * ALT_R + (ver - 1u)) * 4096 : endTs + ccur * 4096;
    const uint32_t* snT = snapTs + cj * 4096;
    bool ok = uni(pm[j].snap_valid) != 0u && uni(pm[j].snap_ip) == ip && uni(pm[j].end_kind) != END_NONE;
    const bool prepared = j == cur + 1u && ver != VER_SERIAL;          // k_lz4_alt has looked at this pair of states
    uint32_t adopt = 0;                                                  // alternative slot to adopt (1 .. ALT_R)
    if (prepared && ok && ((uni(agree[cj]) >> ver) & 1u))
      ;                                                                  // compared by k_lz4_alt against exactly this state
    else if (prepared && ver < ALT_R && uni(altMetas[cj * ALT_R + ver].snap_valid) != 0u)
      {
      adopt = ver + 1u;                                                  // parsed by k_lz4_alt from exactly this state
      ok = false;
      }
    else if (ok)
      {
      bool same = true;
      for (int i = lane; i < 4096; i += 64)
        {
        const uint32_t a = curT[i], b = snT[i];
        same = same && (a == b || (a + MAXD < ip && b + MAXD < ip));
        }
      ok = __ballot(!same) == 0ull;
      }
    __syncthreads();                               // every wave has read chunk j's words
    if (adopt)
      {
      if (threadIdx.x == 0)
        {
        const Meta am = altMetas[cj * ALT_R + (adopt - 1u)];
        pm[j].end_kind = am.end_kind;
        pm[j].end_ip = am.end_ip;
        pm[j].ndesc = am.ndesc;
        pm[j].first_in = am.first_in;
        pm[j].reparsed = 1u + adopt;               // k_lz4_sizes / k_lz4_emit take the descriptors of that slot
        }
      }
    else if (!ok)
      {
      // exact re-parse of chunk j from the true state
      for (int i = lane; i < 4096; i += 64)
        tab[i] = curT[i];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint32_t c_hi = (j + 1u == g.K) ? 0xffffffffu : (j + 1u) * g.chunk;
      lz4_parse<STITCH_W>(src, g.n, tab, g.xchg ? nullptr : dups[wave], ip, true, false, true, j * g.chunk, c_hi, descs + cj * g.dcap, g.dcap, pm + j,
                          snapTs + cj * 4096, endTs + cj * 4096, lane, wave, xch, 0u, scalar_search(g));
      if (threadIdx.x == 0) pm[j].reparsed = 1u;
      }
    if (threadIdx.x == 0)
      {
      pm[j].accepted = 1u;
      atomicAdd(status + 1, 1u);                 // statistics: accepted chunks, re-parsed chunks
      if (!ok) atomicAdd(status + 2, 1u);
      }
    cur = j;
    ver = adopt ? adopt : (ok ? 0u : VER_SERIAL);
    }
  if (threadIdx.x == 0) atomicOr(status, 64u);
  }

__device__ __forceinline__ uint32_t ext_bytes(uint32_t len) { return len >= 15u ? (len - 15u) / 255u + 1u : 0u; }
__device__ __forceinline__ uint32_t enc_size(const Desc& d)
  {
  uint32_t s = 1u + ext_bytes(d.lit) + d.lit;
  if (d.ml)
    s += 2u + ext_bytes(d.ml - 4u);
  return s;
  }

// encoded bytes of every accepted chunk
// descriptors of an accepted chunk: its own buffer, or the alternative slot the stitch pass adopted (Meta::reparsed >= 2)
__device__ __forceinline__ const Desc* chunk_descs(const Geom& g, const Desc* descs, const Desc* altDescs, const Meta& m, size_t ck)
  {
  return m.reparsed >= 2u ? altDescs + (ck * ALT_R + (m.reparsed - 2u)) * g.alt_dcap : descs + ck * g.dcap;
  }

__global__ void __launch_bounds__(256) k_lz4_sizes(Geom g, const Desc* __restrict__ descs, const Desc* __restrict__ altDescs,
                                                   const Meta* __restrict__ metas, uint32_t* __restrict__ chunk_bytes)
  {
  __shared__ uint32_t red[256];
  const uint32_t k = blockIdx.x, p = blockIdx.y;
  const size_t ck = (size_t)p * g.K + k;
  uint32_t sum = 0;
  if (metas[ck].accepted)
    {
    const Desc* d = chunk_descs(g, descs, altDescs, metas[ck], ck);
    const uint32_t nd = metas[ck].ndesc;
    for (uint32_t i = threadIdx.x; i < nd; i += 256u)
      sum += enc_size(d[i]);
    }
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1)
    {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
    }
  if (threadIdx.x == 0)
    chunk_bytes[ck] = red[0];
  }

// exclusive scan of chunk sizes per plane (accepted chunks are in increasing order)
__global__ void __launch_bounds__(1024) k_lz4_offsets(Geom g, const uint32_t* __restrict__ chunk_bytes, uint32_t* __restrict__ chunk_off,
                                                      uint32_t* __restrict__ sizes)
  {
  __shared__ uint32_t part[1024];
  const uint32_t p = blockIdx.x;
  const uint32_t per = (g.K + 1023u) / 1024u;
  const uint32_t k0 = threadIdx.x * per, k1 = (k0 + per < g.K) ? k0 + per : g.K;
  uint32_t sum = 0;
  for (uint32_t k = k0; k < k1; ++k)
    sum += chunk_bytes[(size_t)p * g.K + k];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t o = 1; o < 1024u; o <<= 1)
    {
    const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
    }
  uint32_t run = part[threadIdx.x] - sum;
  for (uint32_t k = k0; k < k1; ++k)
    {
    chunk_off[(size_t)p * g.K + k] = run;
    run += chunk_bytes[(size_t)p * g.K + k];
    }
  if (threadIdx.x == 1023u)
    sizes[p] = part[1023];
  }

// block-wide copy / fill helpers for the emit pass
__device__ __forceinline__ void block_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, int tid)
  {
  const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
  const uint32_t h = head < n ? head : n;
  if ((uint32_t)tid < h)
    dst[tid] = src[tid];
  uint32_t i = h + 16u * (uint32_t)tid;
  for (; i + 16u <= n; i += EMIT_T * 16u)
    *(u32x4*)(dst + i) = ld128(src + i);
  if (i < n && n - i < 16u)
    for (uint32_t k = i; k < n; ++k)
      dst[k] = src[k];
  }

// writes the block bytes of every accepted chunk
// A literal run of BIG_RUN bytes or more (an incompressible plane is ONE run of 300 MB) is not copied by the workgroup that
// meets it: it goes to a list, and k_lz4_bigcopy moves all listed runs in 256 KiB pieces with the whole GPU.
constexpr uint32_t BIG_RUN = 1u << 20, BIG_PIECE = 256u << 10, BIG_CAP = 4096;
struct BigJob { uint32_t plane, src, dst, len; };
struct BigList { uint32_t count, pad[3]; BigJob job[BIG_CAP]; };

__global__ void __launch_bounds__(EMIT_T) k_lz4_emit(const uint8_t* __restrict__ planes, Geom g, const Desc* __restrict__ descs,
                                                     const Desc* __restrict__ altDescs, const Meta* __restrict__ metas,
                                                     const uint32_t* __restrict__ chunk_off, uint8_t* __restrict__ out_base, size_t out_stride,
                                                     BigList* __restrict__ big)
  {
  __shared__ uint32_t sc_out[EMIT_T], sc_in[EMIT_T];
  __shared__ uint32_t wsum_out[EMIT_T / 64], wsum_in[EMIT_T / 64];
  __shared__ __attribute__((aligned(16))) uint8_t stage[EMIT_STAGE + 8];
  __shared__ uint32_t jobs[EMIT_T];
  __shared__ uint32_t njobs, big_slot;
  const uint32_t k = blockIdx.x, p = blockIdx.y;
  const size_t ck = (size_t)p * g.K + k;
  if (!metas[ck].accepted)
    return;
  const int tid = threadIdx.x;
  const uint8_t* src = planes + (size_t)p * g.plane_stride;
  uint8_t* out = out_base + (size_t)p * out_stride;
  const Desc* dl = chunk_descs(g, descs, altDescs, metas[ck], ck);
  const uint32_t nd = metas[ck].ndesc;
  uint32_t carry_out = chunk_off[ck], carry_in = metas[ck].first_in;
  for (uint32_t base = 0; base < nd; base += EMIT_T)
    {
    const uint32_t i = base + (uint32_t)tid;
    Desc d; d.lit = 0; d.ml = 0; d.off = 0;
    uint32_t es = 0, is = 0;
    if (i < nd)
      {
      d = dl[i];
      es = enc_size(d);
      is = d.lit + d.ml;
      }
    // inclusive scans of the encoded and the source sizes: inside the wave with DPP shifts, across the four waves through LDS
    {
    uint32_t a = es, b = is;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
      {
      const uint32_t ua = (uint32_t)__shfl_up((int)a, o), ub = (uint32_t)__shfl_up((int)b, o);
      if ((tid & 63) >= o) { a += ua; b += ub; }
      }
    if ((tid & 63) == 63) { wsum_out[tid >> 6] = a; wsum_in[tid >> 6] = b; }
    if (tid == 0) njobs = 0;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); ++w) { a += wsum_out[w]; b += wsum_in[w]; }
    sc_out[tid] = a;
    sc_in[tid] = b;
    __syncthreads();
    }
    const uint32_t opos = carry_out + sc_out[tid] - es, ipos = carry_in + sc_in[tid] - is;
    const uint32_t mcode = d.ml ? d.ml - 4u : 0u;
    const uint32_t le = ext_bytes(d.lit);
    const bool long_job = i < nd && (d.lit > 96u || le > 96u || ext_bytes(mcode) > 96u);
    if (long_job)
      jobs[atomicAdd(&njobs, 1u)] = (uint32_t)tid;
    __syncthreads();
    // Short sequences (a plane of 7-byte sequences has 50,000 per chunk) are assembled in LDS and leave as aligned dwords:
    // byte stores straight to memory made this pass 33 ms on the walk mesh.
    const uint32_t total_out = sc_out[EMIT_T - 1];
    const bool staged = njobs == 0u && total_out <= EMIT_STAGE;
    if (i < nd && !long_job)
      {
      uint8_t* o = staged ? stage + (opos - carry_out) : out + opos;
      const uint32_t tl = d.lit >= 15u ? 15u : d.lit, tm = d.ml ? (mcode >= 15u ? 15u : mcode) : 0u;
      *o++ = (uint8_t)((tl << 4) | tm);
      if (le)
        {
        for (uint32_t q = 0; q + 1u < le; ++q) *o++ = 255;
        *o++ = (uint8_t)((d.lit - 15u) % 255u);
        }
      for (uint32_t q = 0; q < d.lit; ++q) o[q] = src[ipos + q];
      o += d.lit;
      if (d.ml)
        {
        *o++ = (uint8_t)d.off;
        *o++ = (uint8_t)(d.off >> 8);
        const uint32_t me = ext_bytes(mcode);
        if (me)
          {
          for (uint32_t q = 0; q + 1u < me; ++q) *o++ = 255;
          *o++ = (uint8_t)((mcode - 15u) % 255u);
          }
        }
      }
    else if (long_job)
      out[opos] = (uint8_t)(((d.lit >= 15u ? 15u : d.lit) << 4) | (d.ml ? (mcode >= 15u ? 15u : mcode) : 0u));
    __syncthreads();
    if (staged)
      {
      uint8_t* dst = out + carry_out;
      const uint32_t head0 = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
      const uint32_t head = head0 < total_out ? head0 : total_out;
      if ((uint32_t)tid < head)
        dst[tid] = stage[tid];
      const uint32_t body = (total_out - head) >> 2;                      // aligned dwords of the destination
      const uint32_t* sw = (const uint32_t*)stage;
      for (uint32_t j = (uint32_t)tid; j < body; j += EMIT_T)
        {
        const uint32_t q = head + 4u * j;                                  // stage byte offset of this dword
        *(uint32_t*)(dst + q) = __builtin_amdgcn_alignbyte(sw[(q >> 2) + 1u], sw[q >> 2], q & 3u);
        }
      const uint32_t done = head + 4u * body;
      if ((uint32_t)tid < total_out - done)
        dst[done + tid] = stage[done + tid];
      }
    // long runs: the whole block works on one descriptor at a time
    const uint32_t nj = njobs;
    for (uint32_t jn = 0; jn < nj; ++jn)
      {
      const uint32_t t = jobs[jn];
      const Desc dj = dl[base + t];
      const uint32_t es_j = enc_size(dj);
      uint8_t* o = out + (carry_out + sc_out[t] - es_j) + 1u;
      const uint32_t ip_j = carry_in + sc_in[t] - (dj.lit + dj.ml);
      const uint32_t le = ext_bytes(dj.lit);
      if (le)
        {
        for (uint32_t q = (uint32_t)tid; q + 1u < le; q += EMIT_T) o[q] = 255;
        if (tid == 0) o[le - 1u] = (uint8_t)((dj.lit - 15u) % 255u);
        o += le;
        }
      bool listed = false;
      if (dj.lit >= BIG_RUN)
        {
        if (tid == 0)
          {
          const uint32_t slot = atomicAdd(&big->count, 1u);
          if (slot < BIG_CAP)
            big->job[slot] = BigJob{ p, ip_j, (uint32_t)(o - out), dj.lit };
          big_slot = slot;
          }
        __syncthreads();
        listed = big_slot < BIG_CAP;
        __syncthreads();
        }
      if (!listed)
        block_copy(o, src + ip_j, dj.lit, tid);
      o += dj.lit;
      if (dj.ml)
        {
        const uint32_t mcode = dj.ml - 4u;
        if (tid == 0) { o[0] = (uint8_t)dj.off; o[1] = (uint8_t)(dj.off >> 8); }
        o += 2;
        const uint32_t me = ext_bytes(mcode);
        if (me)
          {
          for (uint32_t q = (uint32_t)tid; q + 1u < me; q += EMIT_T) o[q] = 255;
          if (tid == 0) o[me - 1u] = (uint8_t)((mcode - 15u) % 255u);
          }
        }
      }
    carry_out += sc_out[EMIT_T - 1];
    carry_in += sc_in[EMIT_T - 1];
    __syncthreads();
    }
  }

__global__ void __launch_bounds__(EMIT_T) k_lz4_bigcopy(const uint8_t* __restrict__ planes, size_t plane_stride, uint8_t* __restrict__ out_base,
                                                        size_t out_stride, const BigList* __restrict__ big)
  {
  const uint32_t nj = big->count < BIG_CAP ? big->count : BIG_CAP;
  uint32_t first = 0;                                            // global index of job j's first piece
  for (uint32_t j = 0; j < nj; ++j)
    {
    const BigJob b = big->job[j];
    const uint32_t pieces = (b.len + BIG_PIECE - 1u) / BIG_PIECE;
    // pieces first .. first + pieces - 1 belong to this job; this workgroup takes those congruent to its index
    uint32_t q = (blockIdx.x + gridDim.x - first % gridDim.x) % gridDim.x;
    for (; q < pieces; q += gridDim.x)
      {
      const uint32_t o0 = q * BIG_PIECE;
      const uint32_t len = b.len - o0 < BIG_PIECE ? b.len - o0 : BIG_PIECE;
      block_copy(out_base + (size_t)b.plane * out_stride + b.dst + o0, planes + (size_t)b.plane * plane_stride + b.src + o0, len, threadIdx.x);
      }
    first += pieces;
    }
  }

// the search rounds of lz4_parse through ds_wrxchg_rtn_b32: only where the device has shown that it applies the lanes of one such
// instruction in lane order (the float encoder's test, k_fpc32_encode.hip); TRICO_LZ4_XCHG=0 keeps the scoreboard + ballot rounds
static bool lz4_use_xchg()
  {
  static const bool off = [] { const char* e = tune_env("TRICO_LZ4_XCHG"); return e && e[0] == '0'; }();
  return !off && lds_lane_order_ok();
  }

struct Plan { Geom g; size_t off_desc, off_meta, off_snap, off_end, off_cbytes, off_coff, off_altdesc, off_altmeta, off_altend, total; };

// mode 0: long matches (512 KiB chunks, 384 KiB warm-up), mode 1: short sequences (64-192 KiB / 70,000 B); TRICO_LZ4_CHUNK / TRICO_LZ4_WARM
// fix one geometry for both (tuning knobs)
static bool geometry_forced() { return getenv("TRICO_LZ4_CHUNK") || getenv("TRICO_LZ4_WARM"); }

Plan make_plan(uint32_t n, int nplanes, size_t plane_stride, int mode)
  {
  struct Env { uint32_t chunk, warm; bool forced; };
  static const Env env = []
    {
    Env v;
    v.forced = geometry_forced();
    const char* e = getenv("TRICO_LZ4_CHUNK");
    v.chunk = e ? (uint32_t)atoi(e) : (512u << 10);        // (round 6: 1 MiB until then - with alternative parses and open matches the parse pass is what is left, and it is one chunk's time)
    if (v.chunk < (1u << 17)) v.chunk = 1u << 17;
    const char* w = getenv("TRICO_LZ4_WARM");
    v.warm = w ? (uint32_t)atoi(w) : (384u << 10);
    if (v.warm < 70000u) v.warm = 70000u;
    if (v.warm > v.chunk) v.warm = v.chunk;
    return v;
    }();
  const uint32_t env_chunk = env.chunk, env_warm = env.warm;
  const bool forced = env.forced;
  // mode > 0: that many planes of short sequences.  Their chunks should all run at once (~2048 of them, see k_lz4_parse) and a chunk
  // costs its own bytes + the warm-up at ~0.17 us per byte: 64 KiB steps (other sizes are slower, see the table at k_lz4_probe)
  // between 64 KiB - middle-sized meshes: a 5 MB plane took the 44 ms of ONE 192 KiB chunk - and the 192 KiB measured best at 300 MB.
  uint32_t short_chunk = 192u << 10;
  if (mode > 0)
    {
    const uint64_t per_plane = 2048u / (uint32_t)mode;
    const uint64_t want = ((uint64_t)n + per_plane - 1) / per_plane;
    short_chunk = (uint32_t)(((want + 65535u) >> 16) << 16);
    if (short_chunk < (64u << 10)) short_chunk = 64u << 10;
    if (short_chunk > (192u << 10)) short_chunk = 192u << 10;
    }
  const uint32_t chunk = forced || mode == 0 ? env_chunk : short_chunk;
  const uint32_t warm = forced || mode == 0 ? env_warm : (chunk < 70000u ? chunk : 70000u);      // never more than a chunk: k_lz4_parse starts chunk k at k * chunk - warm
  Plan p;
  p.g.n = n;
  p.g.chunk = chunk;
  p.g.warm = warm;
  p.g.K = (uint32_t)(((uint64_t)n + chunk - 1) / chunk);
  p.g.dcap = chunk / 4u + 16u;
  p.g.plane_stride = plane_stride;
  const size_t cells = (size_t)p.g.K * nplanes;
  size_t o = 0;
  p.off_desc = o;   o += align_up(cells * p.g.dcap * sizeof(Desc), 256);
  p.off_meta = o;   o += align_up(cells * sizeof(Meta), 256);
  p.off_snap = o;   o += cells * 4096 * 4;
  p.off_end = o;    o += cells * 4096 * 4;
  p.off_cbytes = o; o += align_up(cells * 4, 256);
  p.off_coff = o;   o += align_up(cells * 4, 256);
  // alternative parses (k_lz4_alt): only for the long-match geometry; 4096 descriptors per slot (such chunks have dozens)
  p.g.alt_rounds = mode == 0 ? ALT_R : 0u;
  p.g.xchg = lz4_use_xchg() ? 1u : 0u;
  p.g.alt_dcap = 4096;
  p.off_altmeta = o; o += align_up(cells * ALT_R * sizeof(Meta), 256);
  p.off_altdesc = o; o += p.g.alt_rounds ? align_up(cells * ALT_R * p.g.alt_dcap * sizeof(Desc), 256) : 0;
  p.off_altend = o;  o += p.g.alt_rounds ? cells * ALT_R * 4096 * 4 : 0;
  p.total = o + 256;
  return p;
  }

// the probe's descriptors and chunk records live behind the larger of the two plans
size_t probe_bytes(int nplanes)
  {
  const size_t probe = align_up((size_t)nplanes * PROBE_S * (PROBE_DCAP * sizeof(Desc) + sizeof(Meta)), 256) + 256;
  return probe > sizeof(BigList) + 256 ? probe : sizeof(BigList) + 256;                 // k_lz4_emit's list of big runs reuses the area
  }
size_t plans_bytes(uint32_t n, int nplanes, size_t plane_stride)
  {
  // (the long-match geometry, and the short-sequence one with the smallest and the largest chunks it may choose)
  const size_t a = make_plan(n, nplanes, plane_stride, 0).total, b = make_plan(n, nplanes, plane_stride, 1).total, c = make_plan(n, nplanes, plane_stride, nplanes).total;
  const size_t m = a > b ? a : b;
  return align_up(m > c ? m : c, 256);
  }

} // namespace

// planes at or above this size take the chunked path (below, one workgroup per plane is faster)
uint32_t lz4_chunked_threshold()
  {
  static uint32_t t = 0;
  if (!t)
    {
    const char* e = getenv("TRICO_LZ4_CHUNKED_MIN");
    t = e ? (uint32_t)atoi(e) : (256u << 10);             // (measured: at 300 KB the chunked path is level with the one-workgroup compressor, at 2.4 MB 2.5 x (long matches) and 7 x (short sequences) faster)
    if (t < 65547u) t = 65547u;
    }
  return t;
  }

size_t lz4_chunked_workspace(uint32_t n, int nplanes, size_t plane_stride)
  {
  return plans_bytes(n, nplanes, plane_stride) + probe_bytes(nplanes);
  }

int launch_lz4_encode_chunked(const uint8_t* d_planes, size_t plane_stride, uint32_t n, int nplanes, uint8_t* d_out, size_t out_stride,
                              uint32_t* d_sizes, uint8_t* d_ws, size_t ws_bytes, uint32_t* d_status)
  {
  if (lz4_chunked_workspace(n, nplanes, plane_stride) > ws_bytes || nplanes > 8)
    {
    set_error("lz4 chunked encode: workspace too small");
    return 0;
    }
  hipStream_t st = current_stream();
  int mode = 0;
  uint8_t order[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
  {
  // look before parsing: bytes per sequence of every plane (see k_lz4_probe)
  uint8_t* pa = d_ws + plans_bytes(n, nplanes, plane_stride);
  Meta* pm = (Meta*)pa;
  Desc* pd = (Desc*)(pa + align_up((size_t)nplanes * PROBE_S * sizeof(Meta), 256));
  hipLaunchKernelGGL(k_lz4_probe, dim3(PROBE_S, nplanes), dim3(64), 0, st, d_planes, n, plane_stride, pd, pm, lz4_use_xchg() ? 1u : 0u);
  Meta h[8 * PROBE_S];
  if (!hip_ok(hipMemcpyAsync(h, pm, (size_t)nplanes * PROBE_S * sizeof(Meta), hipMemcpyDeviceToHost, st), "lz4 probe readback") ||
      !hip_ok(hipStreamSynchronize(st), "lz4 probe"))
    return 0;
  uint32_t nds[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int pl = 0; pl < nplanes; ++pl)
    {
    for (uint32_t k = 0; k < PROBE_S; ++k)
      nds[pl] += h[pl * PROBE_S + k].ndesc;
    if ((uint64_t)nds[pl] * 48u > (uint64_t)PROBE_S * PROBE_W)                 // fewer than 48 bytes per sequence
      ++mode;
    }
  // planes by sequences per byte, densest first (insertion sort of at most eight)
  for (int i = 1; i < nplanes; ++i)
    for (int j = i; j > 0 && nds[order[j]] > nds[order[j - 1]]; --j)
      {
      const uint8_t t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
      }
  }
  Plan p = make_plan(n, nplanes, plane_stride, mode);
  for (int i = 0; i < 8; ++i)
    p.g.order[i] = order[i];
  Desc* descs = (Desc*)(d_ws + p.off_desc);
  Meta* metas = (Meta*)(d_ws + p.off_meta);
  uint32_t* snapTs = (uint32_t*)(d_ws + p.off_snap);
  uint32_t* endTs = (uint32_t*)(d_ws + p.off_end);
  uint32_t* cbytes = (uint32_t*)(d_ws + p.off_cbytes);
  uint32_t* coff = (uint32_t*)(d_ws + p.off_coff);
  hipLaunchKernelGGL(k_lz4_parse, dim3(p.g.K, nplanes), dim3(64), p.g.xchg ? 16384u : 20480u, st, d_planes, p.g, descs, metas, snapTs, endTs);
  hipLaunchKernelGGL(k_lz4_extend, dim3(EXT_G, nplanes), dim3(256), 0, st, d_planes, p.g, descs, metas);
  hipLaunchKernelGGL(k_lz4_extend_done, dim3((p.g.K + 255u) / 256u, nplanes), dim3(256), 0, st, p.g, descs, metas);
  Desc* altDescs = (Desc*)(d_ws + p.off_altdesc);
  Meta* altMetas = (Meta*)(d_ws + p.off_altmeta);
  uint32_t* altEndTs = (uint32_t*)(d_ws + p.off_altend);
  uint32_t* agree = coff;                                            // free until k_lz4_offsets
  {
  const size_t cells = (size_t)p.g.K * nplanes;
  if (!hip_ok(hipMemsetAsync(agree, 0, cells * 4, st), "memset(agree)") ||
      !hip_ok(hipMemsetAsync(altMetas, 0, cells * ALT_R * sizeof(Meta), st), "memset(alt metas)"))
    return 0;
  }
  for (uint32_t r = 1; r <= (p.g.alt_rounds ? p.g.alt_rounds + 1u : 1u) && r <= ALT_R; ++r)
    hipLaunchKernelGGL(k_lz4_alt, dim3(p.g.K, nplanes), dim3(64), 0, st, d_planes, p.g, r, metas, snapTs, endTs, altDescs, altMetas, altEndTs, agree);
  hipLaunchKernelGGL(k_lz4_stitch, dim3(nplanes), dim3(64 * STITCH_W), 0, st, d_planes, p.g, descs, metas, snapTs, endTs, agree, altMetas,
                     altEndTs, d_status);
  if (getenv("TRICO_LZ4_DEBUG"))
    {
    const size_t cells = (size_t)p.g.K * nplanes;
    Meta* h = (Meta*)malloc(cells * sizeof(Meta));
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, metas, cells * sizeof(Meta), hipMemcpyDeviceToHost);
    for (int pl = 0; pl < nplanes; ++pl)
      {
      fprintf(stderr, "plane %d:", pl);
      uint32_t acc = 0, rep = 0;
      for (uint32_t k = 0; k < p.g.K; ++k)
        {
        const Meta& m = h[(size_t)pl * p.g.K + k];
        acc += m.accepted; rep += m.reparsed;
        if (m.reparsed && rep <= 40) fprintf(stderr, " %u(nd %u)", k, m.ndesc);
        }
      fprintf(stderr, "\n  accepted %u reparsed %u of %u\n", acc, rep, p.g.K);
      }
    free(h);
    }
  hipLaunchKernelGGL(k_lz4_sizes, dim3(p.g.K, nplanes), dim3(256), 0, st, p.g, descs, altDescs, metas, cbytes);
  hipLaunchKernelGGL(k_lz4_offsets, dim3(nplanes), dim3(1024), 0, st, p.g, cbytes, coff, d_sizes);
  // the list of big literal runs takes the place of the probe's records
  BigList* big = (BigList*)(d_ws + plans_bytes(n, nplanes, plane_stride));
  if (!hip_ok(hipMemsetAsync(big, 0, 16, st), "memset(big runs)"))
    return 0;
  hipLaunchKernelGGL(k_lz4_emit, dim3(p.g.K, nplanes), dim3(EMIT_T), 0, st, d_planes, p.g, descs, altDescs, metas, coff, d_out, out_stride, big);
  hipLaunchKernelGGL(k_lz4_bigcopy, dim3(2048), dim3(EMIT_T), 0, st, d_planes, plane_stride, d_out, out_stride, big);
  return hip_ok(hipGetLastError(), "lz4 chunked encode kernels") ? 1 : 0;
  }

} // namespace trico
