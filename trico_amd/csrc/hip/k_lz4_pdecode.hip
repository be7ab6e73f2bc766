// k_lz4_pdecode.hip — data-parallel LZ4 block decompressor.
//
// Replaces LZ4_decompress_safe (lz4.c:2078-2083 -> LZ4_decompress_generic lz4.c:1657-2072) as called per byte plane by the
// readers (trico.c:1100-1129), with the acceptance rules of k_lz4_decode.hip (in-bounds input, literal and match runs inside
// the output, offsets that are neither zero nor reach before the start, exact output size) and a status word for anything else.
//
// A block is one chain of sequences (token, literal run, offset, match), every match possibly reading what an earlier
// sequence wrote: decoded in order, a plane of short sequences (43 M sequences of 7 bytes in a scanned-mesh-like index plane)
// is one dependent chain of seconds.  Both dependences are broken here:
//   * WHERE the sequences are.  The compressed bytes are cut into tiles.  One wave per tile starts parsing a little
//     before its tile at an arbitrary byte; a wrong start falls into step with the real chain after a few sequences
//     (every hop lands on a real token with probability ~1 / sequence length).  The tile records the first token it sees
//     inside the tile, where its walk leaves the tile, and the sequences / output bytes in between (k_pd_tiles).  A
//     single wave then follows the chain from position 0 through those records: a tile whose first token is exactly
//     where its predecessor's walk arrived was walked from a true token, so its record is exact; the (rare) others are
//     re-walked on the spot (k_pd_chain).  That also yields every tile's first output position.
//   * WHAT the bytes are.  Every output byte gets a 32-bit source: a position in the compressed input (literal) or an
//     earlier output position (match), written by one wave per tile (k_pd_fill; runs of 1 KiB and more go to a job
//     list worked off by whole workgroups, k_pd_jobs).  Pointer jumping (src[i] = src[src[i]]) resolves chains of matches in
//     log2(longest chain) rounds without any ordering between threads (k_pd_jump), and a gather produces the plane
//     (k_pd_gather).
// Nothing waits on the host: the jump rounds are launched up front and return at once when the previous round left no
// word open.  Traffic: ~4 B written + 12 B per round and output byte.
// Round 6: a FINAL word carries the byte itself (literals are read where the words are written), so the gather is one pass over
// the words without a look into the compressed input; a round makes up to three hops per open word (20 launches instead of 31).
// Round 3: (1) a match that overlaps itself (offset < length: LZ4's way of writing a periodic run, lz4.c:1840-1870) points every
// byte at the FIRST period, src = op - off + (k mod off), instead of at the byte `off` before it: the run is resolved in one
// round whatever its length (the two upper byte planes of a grid's indices are ONE run of 300 MB each: 28 rounds before).
// (2) All planes of a stream go through every phase together (grid.y = plane, one workspace per plane): the serial phases
// (k_pd_chain: one wave per plane) overlap instead of adding up, and a round of jumps is one launch.  (3) The last workgroup
// of a jump round closes it (k_pd_round_end is gone).
#include "common.hpp"
#include <stdlib.h>

namespace trico {

namespace {

constexpr uint32_t PD_TILE = 8192;            // compressed bytes per tile
constexpr uint32_t PD_LEAD = 1024;            // a tile's walk starts this many bytes before the tile
constexpr uint32_t PD_STAGE = PD_LEAD + PD_TILE + 512;   // bytes of a tile staged in LDS (dword aligned base)
constexpr uint32_t PD_LONG = 1024;            // runs from this length go to the job list ...
constexpr uint32_t PD_PIECE = 65536;          // ... in pieces of at most this many bytes
constexpr uint32_t PD_END = 0xffffffffu;      // "walk ended with the last sequence of the block"
constexpr uint32_t PD_NONE = 0xfffffffeu;     // "no token seen / walk failed"
constexpr uint32_t PD_FINAL = 0x80000000u;    // src word: final, its low byte is the output byte (else: an earlier output index)
#ifndef TRICO_PD_HOPS
#define TRICO_PD_HOPS 4
#define TRICO_PD_ROUNDS 16
#endif
constexpr int PD_HOPS = TRICO_PD_HOPS;        // pointer hops of an open word per round
constexpr int PD_ROUNDS = TRICO_PD_ROUNDS;    // chains are shorter than 2^31 < 4^16 (3 hops x 20 rounds until round 6: 12.4 ms for the benchmark
                                              // mesh's planes; 4 x 16: 11.5; 5 x 14: 11.55; 6 x 12: 11.7 - a round is a pass over 4.8 GB of words)

struct Tile { uint32_t first, exit, nseq, pad; unsigned long long obytes; };
struct Job { uint32_t op, len, a, b; };       // b == PD_NONE: literals from input position a; else match: offset a, its first period starts at output b
struct Ctl { uint32_t error, njobs, changed, done, total_seq, arrived, run_end, pad; };   // run_end: the block's first literal run ends here (tile 0's walk)

// One plane of a stream and its workspace
struct PdPlane
  {
  const uint8_t* in; uint8_t* out;
  Tile* tiles; Tile* tiles0; uint32_t* onpath; uint32_t* out_base; Ctl* ctl; Job* jobs; uint32_t* src;
  uint32_t clen, nt;
  };
struct PdPlanes { PdPlane p[8]; };

__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// bytes of the compressed block: from the tile's LDS image when they are there
struct Bytes
  {
  const uint8_t* in;
  const uint8_t* lds;          // lds - al is dword aligned (al = base's misalignment inside the global buffer)
  uint32_t base, staged, clen, al;
  __device__ __forceinline__ uint32_t at(uint32_t pos) const
    {
    const uint32_t r = pos - base;
    return r < staged ? lds[r] : (pos < clen ? in[pos] : 0u);
    }
  // bytes pos .. pos + 7, little endian (zeros beyond the block): one LDS round trip when the image holds them
  __device__ __forceinline__ uint64_t at8(uint32_t pos) const
    {
    const uint32_t r = pos - base;
    if (r + 12u <= staged && r < staged)
      {
      const uint32_t o = r + al;                                  // offset from the aligned LDS base
      const uint32_t* w = (const uint32_t*)(lds - al) + (o >> 2);
      const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
      const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, o & 3u), hi = __builtin_amdgcn_alignbyte(w2, w1, o & 3u);
      return ((uint64_t)hi << 32) | lo;
      }
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8u; ++k)
      v |= (uint64_t)at(pos + k) << (8u * k);
    return v;
    }
  };

struct Seq { uint32_t lit_pos, lit_len, off, mlen, next; int kind; };   // kind 0: literals + match, 1: last (literals only), 2: malformed

// length extension bytes starting at q (lz4.c:1702-1710): adds them to `len`, advances q past them.  All 64 lanes of the wave
// call this together with the same arguments: 64 bytes per round (a 30 KiB match has 118 extension bytes).  false: malformed.
__device__ __forceinline__ bool ext_length(const Bytes& b, uint32_t& q, uint32_t& len, int lane)
  {
  // A run of megabytes is a megabyte / 255 of extension bytes (the upper byte planes of a regular mesh's indices are ONE match of
  // 300 MB: 1.18 MB of 0xFF), and 64 bytes per round trip made that 2 ms for the one wave that meets it - in the chain walk, in
  // the tile walk and in the fill, each the tail of its kernel.  Far from the block's end the run is first skipped 8 KiB at a time:
  // eight 16-byte loads per lane, all issued before the first is looked at; the first block that is not all 0xFF goes to the
  // byte-wise rounds below.
  for (;;)
    {
    if (q + 8192u + 16u > b.clen || len > 0x7fffffffu - 255u * 8192u)
      break;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) U { u32x4 v; };
    uint32_t all = 0xffffffffu;
    u32x4 x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      x[k] = ((const U*)(b.in + q + 1024u * (uint32_t)k + 16u * (uint32_t)lane))->v;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      all &= x[k][0] & x[k][1] & x[k][2] & x[k][3];
    if (__ballot(all != 0xffffffffu) != 0ull)
      break;
    len += 255u * 8192u;
    q += 8192u;
    }
  for (;;)
    {
    const uint32_t pos = q + (uint32_t)lane;
    const uint32_t x = pos < b.clen ? b.at(pos) : 0x100u;          // beyond the block: stops the run, flagged below
    const uint64_t stop = __ballot(x != 255u);
    if (stop)
      {
      const int k = __builtin_ctzll(stop);
      const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)x, k);
      if (last > 255u)
        return false;                                              // the run reaches the end of the block
      const uint64_t add = 255ull * (uint32_t)k + last;
      if (add + len > 0x7fffffffull)
        return false;
      len += (uint32_t)add;
      q += (uint32_t)k + 1u;
      return true;
      }
    if (len > 0x7fffffffu - 255u * 64u)
      return false;
    len += 255u * 64u;
    q += 64u;
    }
  }

// one sequence starting at p.  Wave-uniform: every lane calls it with the same p.
__device__ __forceinline__ void parse_seq(const Bytes& b, uint32_t p, Seq& s, int lane)
  {
  s.kind = 2;
  s.off = 0; s.mlen = 0; s.lit_len = 0; s.lit_pos = p; s.next = p;
  if (p >= b.clen)
    return;
  uint64_t w = b.at8(p);
  const uint32_t tok = (uint32_t)w & 255u;
  uint32_t q = p + 1u;
  uint32_t ll = tok >> 4;
  if (ll == 15u && !ext_length(b, q, ll, lane))
    return;
  s.lit_pos = q;
  s.lit_len = ll;
  if (ll > b.clen - q)
    return;
  q += ll;
  if (q == b.clen)
    {
    s.kind = 1;                                  // last sequence: literals only (lz4.c:1757-1790)
    s.next = q;
    return;
    }
  if (b.clen - q < 2u)
    return;
  // offset (+ the first extension byte of the match length) usually sit in the 8 bytes already fetched
  uint32_t o2;
  if (q + 2u <= p + 8u)
    o2 = (uint32_t)(w >> (8u * (q - p))) & 0xffffu;
  else
    o2 = (uint32_t)b.at8(q) & 0xffffu;
  s.off = o2;
  q += 2u;
  uint32_t ml = tok & 15u;
  if (ml == 15u && !ext_length(b, q, ml, lane))
    return;
  s.mlen = ml + 4u;
  s.next = q;
  s.kind = 0;
  }

// ---- 64 candidate tokens at once --------------------------------------------------------------------------------------------------
// parse_seq follows the chain one sequence at a time: four dependent LDS round trips and ~300 instructions per sequence by a wave
// that has nothing else to do - a tile of a mesh's second index plane holds ~2,500 sequences, and one wave per tile walked them for
// 10 ms (k_pd_tiles) and again for 11 ms (k_pd_fill).  Here lane l decodes the sequence that WOULD start at byte p + l (token, up to
// one length-extension byte each for the literals and the match, offset), all 64 at once, and the true chain from p is then followed
// through the lanes with v_readlane: ~15 instructions per sequence.  A candidate with more than three extension bytes (runs of 784
// bytes and more) is `slow`: the chain stops there and parse_seq takes that one sequence.  Only used where the window and everything a
// candidate can reach lie inside the staged image and far from the end of the block (win_ok): the last sequence, truncated blocks
// and everything else that needs care stay with parse_seq.
struct Win { uint32_t next, lit, lit_pos, off, mlen, slow; };
constexpr uint32_t WIN_EXT = 3;                                            // length-extension bytes a candidate may have (runs below 784 bytes: never a job, PD_LONG)
constexpr uint32_t WIN_REACH = 64u + 1u + WIN_EXT + (15u + 255u * WIN_EXT) + 2u + WIN_EXT + 16u;   // bytes from p on that a window may look at (with slack)
static_assert(15u + 255u * WIN_EXT + 4u < PD_LONG && WIN_REACH + PD_LEAD < PD_STAGE, "a window's sequences are short runs inside the staged image");

// k mod f for k < 2^16 and 0 < f < 2^16 without an integer division (the self-overlapping matches of a periodic plane: every word of
// them points at the first period): quotient from the float reciprocal, one correction either way
__device__ __forceinline__ uint32_t mod_small(uint32_t k, uint32_t f, float rf)
  {
  const uint32_t q = (uint32_t)((float)k * rf);
  int r = (int)(k - q * f);
  r = r < 0 ? r + (int)f : r;
  return (uint32_t)r >= f ? (uint32_t)r - f : (uint32_t)r;
  }

__device__ __forceinline__ bool win_ok(const Bytes& b, uint32_t p)
  {
  return p >= b.base && p - b.base + WIN_REACH <= b.staged && p + WIN_REACH <= b.clen;
  }

__device__ __forceinline__ Win scan_window(const Bytes& b, uint32_t p, int lane)
  {
  Win w;
  const uint32_t P = p + (uint32_t)lane;
  const uint32_t tok = b.at(P);
  uint32_t lit = tok >> 4, ml = tok & 15u, q = P + 1u;
  uint32_t slow = 0u;
  if (lit == 15u)
    {
    uint32_t e = 255u;
    for (uint32_t x = 0; x < WIN_EXT && e == 255u; ++x)
      {
      e = b.at(q);
      lit += e;
      ++q;
      }
    slow = e == 255u ? 1u : 0u;
    }
  w.lit_pos = q;
  w.lit = lit;
  q += lit;
  w.off = b.at(q) | (b.at(q + 1u) << 8);
  q += 2u;
  if (ml == 15u)
    {
    uint32_t e = 255u;
    for (uint32_t x = 0; x < WIN_EXT && e == 255u; ++x)
      {
      e = b.at(q);
      ml += e;
      ++q;
      }
    slow |= e == 255u ? 1u : 0u;
    }
  w.mlen = ml + 4u;
  w.next = q;
  w.slow = slow;
  return w;
  }

__device__ __forceinline__ uint32_t lane_of(uint32_t x, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l); }

__device__ __forceinline__ void stage_tile(uint32_t* lds, const uint8_t* __restrict__ in, uint32_t clen, uint32_t from, int lane, Bytes& b)
  {
  // dword-aligned image of in[from .. from + PD_STAGE)
  const uint32_t al = (uint32_t)((uintptr_t)(in + from) & 3u);
  const uint8_t* a = in + from - al;
  const uint32_t avail = clen - from + al;                    // bytes from a to the end of the block
  const uint32_t want = PD_STAGE < avail ? PD_STAGE : avail;
  for (uint32_t i = (uint32_t)lane; 4u * i < want; i += 64u)
    {
    uint32_t w = 0;
    if (4u * i + 4u <= avail)
      w = ((const uint32_t*)a)[i];
    else
      for (uint32_t k = 0; 4u * i + k < avail; ++k)
        w |= (uint32_t)a[4u * i + k] << (8u * k);
    lds[i] = w;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  b.in = in;
  b.lds = (const uint8_t*)lds + al;
  b.base = from;
  b.staged = want - al;
  b.clen = clen;
  b.al = al;
  }

// walk from `start`: first token at or after lo, position where the walk reaches hi (or PD_END), sequences and output bytes from
// that first token on.  A malformed sequence ends the walk with exit = PD_NONE.
__device__ __forceinline__ void walk_tile(const Bytes& b, uint32_t start, uint32_t lo, uint32_t hi, Tile& t, int lane)
  {
  t.first = PD_NONE; t.exit = PD_NONE; t.nseq = 0; t.pad = 0; t.obytes = 0;
  uint32_t p = start;
  for (;;)
    {
    if (p >= hi)
      {
      t.exit = p;
      if (t.first == PD_NONE) t.first = p;       // nothing starts inside the tile: the chain passes through
      return;
      }
    if (win_ok(b, p))
      {
      const Win w = scan_window(b, p, lane);
      const uint32_t outl = w.lit + w.mlen;
      uint32_t cur = 0;
      bool slow = false;
      while (cur < 64u && p + cur < hi)
        {
        if (lane_of(w.slow, cur)) { slow = true; break; }
        if (p + cur >= lo && t.first == PD_NONE)
          t.first = p + cur;
        if (t.first != PD_NONE)
          {
          ++t.nseq;
          t.obytes += lane_of(outl, cur);
          }
        cur = uni(lane_of(w.next, cur) - p);
        }
      p += cur;
      if (!slow)
        continue;                                  // (the window is used up, or the walk has reached hi: the loop's head sees to that)
      }
    Seq s;
    parse_seq(b, p, s, lane);
    if (s.kind == 2)
      return;
    if (p >= lo && t.first == PD_NONE)
      t.first = p;
    if (t.first != PD_NONE)
      {
      ++t.nseq;
      t.obytes += (unsigned long long)s.lit_len + s.mlen;
      }
    if (s.kind == 1)
      {
      t.exit = PD_END;
      return;
      }
    p = s.next;
    }
  }

// the block's first sequence, before anything else: where its literal run ends (Ctl::run_end; 0: nothing worth knowing)
__global__ void __launch_bounds__(64) k_pd_head(PdPlanes P)
  {
  __shared__ uint32_t lds[PD_STAGE / 4 + 2];
  const int lane = threadIdx.x;
  const PdPlane& pl = P.p[blockIdx.x];
  if (pl.clen == 0u)
    return;
  Bytes b;
  stage_tile(lds, pl.in, pl.clen, 0u, lane, b);
  Seq s0;
  parse_seq(b, 0u, s0, lane);
  if (lane == 0)
    pl.ctl->run_end = (s0.kind != 2 && s0.lit_len >= 2u * PD_TILE) ? s0.lit_pos + s0.lit_len : 0u;
  }

__global__ void __launch_bounds__(64) k_pd_tiles(PdPlanes P, int second)
  {
  __shared__ uint32_t lds[PD_STAGE / 4 + 2];
  const int lane = threadIdx.x;
  const PdPlane& pl = P.p[blockIdx.y];
  const uint32_t t = blockIdx.x;
  if (t >= pl.nt)
    return;
  const uint8_t* in = pl.in;
  const uint32_t clen = pl.clen;
  const uint32_t lo = t * PD_TILE, hi = lo + PD_TILE;
  uint32_t start = lo > PD_LEAD ? lo - PD_LEAD : 0u;
  Tile* dst = second ? pl.tiles : pl.tiles0;
  if (t != 0u && hi <= pl.ctl->run_end)
    {
    // An incompressible plane is ONE literal run of its whole size: every byte of it read as a token is garbage, every tile's
    // speculative walk ends in the next tile, and both rounds walked all 37,000 tiles of a 300 MB plane (7 + 1 ms).  The chain
    // enters none of them: k_pd_head has said where the block's first literal run ends.
    if (lane == 0)
      {
      Tile none;
      none.first = PD_NONE; none.exit = PD_NONE; none.nseq = 0; none.pad = 0; none.obytes = 0;
      dst[t] = none;                                  // (k_pd_chain walks the tile the chain really enters on the spot)
      pl.onpath[t] = 0u;
      }
    return;
    }
  if (second)
    {
    // second round: where the predecessor's walk of the first round left its tile.  Even if that walk started at a wrong
    // byte, it has normally fallen into step with the real chain within its 8 KiB, so this is a real token.
    const uint32_t e = t ? pl.tiles0[t - 1].exit : PD_NONE;
    // (a tile whose first-round walk already began exactly where its predecessor arrived has nothing to gain either)
    if (t == 0 || e < lo || e >= hi || e == pl.tiles0[t].first)
      {
      if (lane == 0)
        {
        dst[t] = pl.tiles0[t];                        // nothing (known) starts here: keep the record of the first round
        pl.onpath[t] = 0u;
        }
      return;
      }
    start = e;
    }
  Bytes b;
  stage_tile(lds, in, clen, start, lane, b);
  if (!second && start != 0u)
    {
    // An arbitrary byte inside a run of 0xFF length-extension bytes never falls into step: read as a token it announces a
    // literal run as long as the whole run of 0xFF says (a plane of 30 KiB matches is 118 such bytes per sequence, and every
    // tile of it would be left to the serial chain walk).  The run ends with its remainder byte, and behind the remainder of
    // a MATCH length the next token starts (lz4.c:1702-1710, 1830-1845): start there.  A wrong guess costs nothing but the
    // speculation: records are only used where the chain really arrives (k_pd_chain).
    uint32_t q = start;
    if (b.at(q) == 255u)
      {
      uint32_t skipped = 0;
      for (;;)
        {
        const uint32_t pos = q + (uint32_t)lane;
        const uint64_t stop = __ballot(pos >= clen || b.at(pos) != 255u);
        if (stop)
          {
          q += (uint32_t)__builtin_ctzll(stop);
          break;
          }
        q += 64u;
        skipped += 64u;
        if (skipped >= PD_LEAD)
          break;
        }
      if (q + 1u < clen && q + 1u < hi)
        start = q + 1u;                               // behind the remainder byte
      }
    }
  Tile r;
  walk_tile(b, start, lo, hi, r, lane);           // every lane walks the same chain (uniform control flow, LDS broadcast reads)
  if (lane == 0)
    {
    dst[t] = r;
    pl.onpath[t] = 0u;
    }
  }

// One wave per plane follows the chain of tiles from position 0, 64 tiles per step: lane l looks at tile t + l, a tile is on the
// chain if its record starts exactly where the tile before it left off (and that is inside the tile), so the longest run of
// such lanes from lane 0 is accepted at once, with a wave scan for the output positions.  Where the run breaks - the tile's own
// walk had not fallen into step when it entered the tile, or the chain jumps over tiles (a literal run of megabytes) - the
// tile the chain really enters is walked from the true token, on the spot.  out_base: output bytes before the tile.
__global__ void __launch_bounds__(64) k_pd_chain(PdPlanes P, uint32_t n, uint32_t* __restrict__ status)
  {
  const PdPlane& pl = P.p[blockIdx.x];
  const uint8_t* in = pl.in;
  const uint32_t clen = pl.clen, ntiles = pl.nt;
  Tile* tiles = pl.tiles;
  uint32_t* onpath = pl.onpath;
  uint32_t* out_base = pl.out_base;
  Ctl* ctl = pl.ctl;
  __shared__ uint32_t lds[PD_STAGE / 4 + 2];
  const int lane = threadIdx.x;
  uint32_t entry = 0;
  unsigned long long outp = 0;
  uint32_t seqs = 0;
  bool bad = clen == 0u;
  bool walked = false;                            // the tile at `entry` has just been walked from `entry`
  for (uint32_t guard = 0; !bad && guard <= 2u * ntiles + 2u; ++guard)
    {
    const uint32_t t = entry / PD_TILE;
    if (t >= ntiles) { bad = true; break; }
    const bool have = t + (uint32_t)lane < ntiles;
    Tile r;
    r.first = PD_NONE; r.exit = PD_NONE; r.nseq = 0; r.pad = 0; r.obytes = 0;
    if (have)
      r = tiles[t + (uint32_t)lane];
    // where the chain would enter my tile: where the lane below me left its tile
    const uint32_t from = (uint32_t)__builtin_amdgcn_update_dpp((int)entry, (int)r.exit, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 <- entry
    const bool good = have && r.first == from && r.exit != PD_NONE && from / PD_TILE == t + (uint32_t)lane &&
                      (r.exit == PD_END || r.exit > from);
    const uint64_t gm = __ballot(good);
    uint32_t run = gm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~gm);
    const uint64_t em = __ballot(good && r.exit == PD_END);
    if (em && (uint32_t)__builtin_ctzll(em) < run)
      run = (uint32_t)__builtin_ctzll(em) + 1u;
    if (run == 0u)
      {
      if (walked) { bad = true; break; }          // walked from the true token and still unusable: malformed
      // the tile's own walk had not fallen into step when it entered the tile (or the chain enters it from far away)
      Bytes b;
      stage_tile(lds, in, clen, entry, lane, b);
      Tile w;
      walk_tile(b, entry, entry, (t + 1u) * PD_TILE, w, lane);
      if (uni(w.exit) == PD_NONE) { bad = true; break; }
      if (lane == 0)
        tiles[t] = w;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
      walked = true;
      continue;
      }
    walked = false;
    // inclusive scan of the output bytes and sequences of the accepted lanes
    unsigned long long ob = (uint32_t)lane < run ? r.obytes : 0ull;
    uint32_t ns = (uint32_t)lane < run ? r.nseq : 0u;
    for (int d = 1; d < 64; d <<= 1)
      {
      const unsigned long long o2 = __shfl_up(ob, d, 64);
      const uint32_t n2 = __shfl_up(ns, d, 64);
      if (lane >= d)
        {
        ob += o2;
        ns += n2;
        }
      }
    const unsigned long long total = __shfl(ob, (int)run - 1, 64);
    if ((uint32_t)lane < run)
      {
      onpath[t + (uint32_t)lane] = 1u;
      out_base[t + (uint32_t)lane] = (uint32_t)(outp + ob - r.obytes);
      }
    outp += total;
    seqs += (uint32_t)__shfl((int)ns, (int)run - 1, 64);
    if (outp > n) { bad = true; break; }
    const uint32_t ex = (uint32_t)__shfl((int)r.exit, (int)run - 1, 64);
    if (ex == PD_END)
      break;
    if (ex <= entry) { bad = true; break; }
    entry = ex;
    }
  if (bad || outp != n)
    {
    if (lane == 0)
      {
      atomicOr(status, 8u);
      ctl->error = 1u;
      }
    }
  if (lane == 0)
    ctl->total_seq = seqs;
  }

// Places in the job list, reserved PD_JCHUNK at a time: a tile of a periodic plane holds ~300 runs of a KiB and more, and an atomic
// round trip per run was most of the fill's time.  What a wave has reserved and not used is filled with empty jobs (length 0).
constexpr uint32_t PD_JCHUNK = 16;
struct JobSlots
  {
  uint32_t base = 0, left = 0;
  __device__ __forceinline__ uint32_t take(uint32_t np, Ctl* ctl, Job* jobs, uint32_t job_cap, int lane)
    {
    if (np > left)
      {
      release(jobs, job_cap, lane);
      const uint32_t want = np > PD_JCHUNK ? np : PD_JCHUNK;
      uint32_t j0 = 0;
      if (lane == 0)
        j0 = atomicAdd(&ctl->njobs, want);
      base = uni(j0);
      left = want;
      }
    const uint32_t at = base;
    base += np;
    left -= np;
    return at;
    }
  __device__ __forceinline__ void release(Job* jobs, uint32_t job_cap, int lane)
    {
    for (uint32_t k = (uint32_t)lane; k < left; k += 64u)
      if (base + k < job_cap)
        jobs[base + k] = Job{ 0u, 0u, 0u, PD_NONE };
    left = 0;
    }
  };

// one wave per tile on the chain: source words of its output bytes
__global__ void __launch_bounds__(64) k_pd_fill(PdPlanes P, uint32_t n, uint32_t job_cap, uint32_t* __restrict__ status)
  {
  __shared__ uint32_t lds[PD_STAGE / 4 + 2];
  const int lane = threadIdx.x;
  const PdPlane& pl = P.p[blockIdx.y];
  const uint32_t t = blockIdx.x;
  if (t >= pl.nt)
    return;
  const uint8_t* in = pl.in;
  const uint32_t clen = pl.clen;
  const Tile* tiles = pl.tiles;
  const uint32_t* onpath = pl.onpath;
  const uint32_t* out_base = pl.out_base;
  uint32_t* src = pl.src;
  Job* jobs = pl.jobs;
  Ctl* ctl = pl.ctl;
  if (ctl->error || !onpath[t])
    return;
  const Tile r = tiles[t];
  const uint32_t hi = (t + 1u) * PD_TILE;
  Bytes b;
  stage_tile(lds, in, clen, r.first, lane, b);
  uint32_t p = r.first;
  uint32_t op = out_base[t];
  bool bad = false;
  JobSlots slots;
  while (p < hi)
    {
    if (win_ok(b, p))
      {
      // the sequences of the chain that start in the next 64 bytes, all at once (scan_window): which lanes they are, where their
      // output begins (a scan over their lengths), then every lane writes the words of its own sequence - a sequence of such a
      // plane is a handful of bytes - and only matches of more than 32 bytes are written by the whole wave, one after the other
      const Win w = scan_window(b, p, lane);
      uint64_t chain = 0;
      uint32_t cur = 0;
      bool slow = false;
      while (cur < 64u && p + cur < hi)
        {
        if (lane_of(w.slow, cur)) { slow = true; break; }
        chain |= 1ull << cur;
        cur = uni(lane_of(w.next, cur) - p);
        }
      const bool mine = (chain >> lane) & 1ull;
      const uint32_t outl = mine ? w.lit + w.mlen : 0u;
      uint32_t incl = outl, mxl = (mine && w.lit <= 32u) ? w.lit : 0u, mxm = (mine && w.mlen <= 32u) ? w.mlen : 0u;
      for (int d = 1; d < 64; d <<= 1)
        {
        const uint32_t o2 = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d)
          incl += o2;
        }
      for (int d = 32; d > 0; d >>= 1)
        {
        mxl = max(mxl, (uint32_t)__shfl_xor((int)mxl, d, 64));
        mxm = max(mxm, (uint32_t)__shfl_xor((int)mxm, d, 64));
        }
      const uint32_t total = lane_of(incl, 63u);
      if (total > n - op) { bad = true; break; }
      const uint32_t lop = op + incl - outl, mop = lop + w.lit;            // where my literals / my match begin
      // (lz4.c:1800-1830: the offset must stay inside what has been written)
      if (__ballot(mine && (w.off == 0u || w.off > mop)) != 0ull) { bad = true; break; }
      {
      const bool shortl = mine && w.lit <= 32u;
      for (uint32_t k = 0; k < mxl; ++k)
        if (shortl && k < w.lit)
          src[lop + k] = PD_FINAL | b.at(w.lit_pos + k);
      uint64_t longl = __ballot(mine && w.lit > 32u);
      while (longl)
        {
        const uint32_t l = (uint32_t)__builtin_ctzll(longl);
        longl &= longl - 1ull;
        const uint32_t o = lane_of(lop, l), f = lane_of(w.lit_pos, l), m = lane_of(w.lit, l);
        for (uint32_t k = (uint32_t)lane; k < m; k += 64u)
          src[o + k] = PD_FINAL | b.at(f + k);
        }
      }
      {
      // short matches, lane by lane; one that overlaps itself points at its first period (k mod off, kept as a running remainder)
      const bool shortm = mine && w.mlen <= 32u;
      const uint32_t first = mop - w.off;
      uint32_t r = 0;
      for (uint32_t k = 0; k < mxm; ++k)
        {
        if (shortm && k < w.mlen)
          src[mop + k] = first + r;
        ++r;
        r = r == w.off ? 0u : r;
        }
      }
      uint64_t longm = __ballot(mine && w.mlen > 32u);
      while (longm)
        {
        const uint32_t l = (uint32_t)__builtin_ctzll(longm);
        longm &= longm - 1ull;
        const uint32_t o = lane_of(mop, l), f = lane_of(w.off, l), m = lane_of(w.mlen, l);
        if (m <= f)
          for (uint32_t k = (uint32_t)lane; k < m; k += 64u)
            src[o + k] = o + k - f;
        else
          {
          const float rf = 1.0f / (float)f;
          for (uint32_t k = (uint32_t)lane; k < m; k += 64u)
            src[o + k] = o - f + mod_small(k, f, rf);
          }
        }
      op += total;
      p += cur;
      if (!slow)
        continue;
      if (p >= hi)
        break;
      }
    Seq s;
    parse_seq(b, p, s, lane);
    if (s.kind == 2) { bad = true; break; }
    // literals
    if (s.lit_len > n - op) { bad = true; break; }
    if (s.lit_len >= PD_LONG)
      {
      // pieces of at most PD_PIECE bytes, one workgroup each
      const uint32_t np = (s.lit_len + PD_PIECE - 1u) / PD_PIECE;
      const uint32_t j0 = slots.take(np, ctl, jobs, job_cap, lane);
      for (uint32_t k = (uint32_t)lane; k < np; k += 64u)
        if (j0 + k < job_cap)
          jobs[j0 + k] = Job{ op + k * PD_PIECE, (k + 1u == np) ? s.lit_len - k * PD_PIECE : PD_PIECE, s.lit_pos + k * PD_PIECE, PD_NONE };
      }
    else
      for (uint32_t k = (uint32_t)lane; k < s.lit_len; k += 64u)
        src[op + k] = PD_FINAL | b.at(s.lit_pos + k);                             // a final word carries the byte itself (round 6)
    op += s.lit_len;
    if (s.kind == 1)
      break;
    // match (lz4.c:1800-1830: the offset must stay inside what has been written)
    if (s.off == 0u || s.off > op || s.mlen > n - op) { bad = true; break; }
    if (s.mlen >= PD_LONG)
      {
      const uint32_t np = (s.mlen + PD_PIECE - 1u) / PD_PIECE;
      const uint32_t j0 = slots.take(np, ctl, jobs, job_cap, lane);
      for (uint32_t k = (uint32_t)lane; k < np; k += 64u)
        if (j0 + k < job_cap)
          jobs[j0 + k] = Job{ op + k * PD_PIECE, (k + 1u == np) ? s.mlen - k * PD_PIECE : PD_PIECE, s.off, op - s.off };
      }
    else if (s.mlen <= s.off)
      for (uint32_t k = (uint32_t)lane; k < s.mlen; k += 64u)
        src[op + k] = op + k - s.off;
    else
      // the match overlaps itself: a periodic run.  Every byte points at the first period (which lies before the match).
      {
      const float rf = 1.0f / (float)s.off;                                       // (mlen < PD_LONG here: mod_small's range)
      for (uint32_t k = (uint32_t)lane; k < s.mlen; k += 64u)
        src[op + k] = op - s.off + mod_small(k, s.off, rf);
      }
    op += s.mlen;
    p = s.next;
    }
  slots.release(jobs, job_cap, lane);
  if (bad && lane == 0)
    {
    atomicOr(status, 8u);
    ctl->error = 1u;
    }
  }

// long runs: whole workgroups, grid-stride over the job list
__global__ void __launch_bounds__(256) k_pd_jobs(PdPlanes P, uint32_t job_cap, uint32_t* __restrict__ status)
  {
  const PdPlane& pl = P.p[blockIdx.y];
  Ctl* ctl = pl.ctl;
  uint32_t* src = pl.src;
  if (pl.clen == 0u || ctl->error)
    return;
  const uint32_t nj = ctl->njobs;
  if (nj > job_cap)
    {
    if (blockIdx.x == 0 && threadIdx.x == 0) { atomicOr(status, 8u); ctl->error = 1u; }   // cannot happen: the list holds n / PD_LONG + tiles entries
    return;
    }
  for (uint32_t j = blockIdx.x; j < nj; j += gridDim.x)
    {
    const Job q = pl.jobs[j];
    if (q.b == PD_NONE)
      for (uint32_t k = threadIdx.x; k < q.len; k += 256u)
        src[q.op + k] = PD_FINAL | (uint32_t)pl.in[q.a + k];                      // (fill checked that the run lies inside the block)
    else
      {
      // piece of a match that starts at output b + a: byte i of the match comes from b + (i mod a), the first period
      const uint32_t first = q.op - (q.b + q.a);          // index of the piece's first byte inside the match
      if (first + q.len <= q.a)
        for (uint32_t k = threadIdx.x; k < q.len; k += 256u)
          src[q.op + k] = q.b + first + k;
      else
        {
        uint32_t r = (first + threadIdx.x) % q.a;
        const uint32_t step = 256u % q.a;
        for (uint32_t k = threadIdx.x; k < q.len; k += 256u)
          {
          src[q.op + k] = q.b + r;
          r += step;
          r = r >= q.a ? r - q.a : r;
          }
        }
      }
    }
  }

// one round of pointer jumping; returns at once when the round before left no word open.  The last workgroup to finish closes
// the round: done = every word it wrote is final (a round that only confirms that nothing changes any more is not needed: a word
// is open exactly when PD_FINAL is not set in it).
__global__ void __launch_bounds__(256) k_pd_jump(PdPlanes P, uint32_t n)
  {
  const PdPlane& pl = P.p[blockIdx.y];
  Ctl* ctl = pl.ctl;
  uint32_t* src = pl.src;
  if (pl.clen == 0u || ctl->done || ctl->error)
    return;
  bool open = false;                               // a word this thread wrote is not final yet
  // four words per thread (the workspace is 256-byte aligned): 16-byte loads of the words, dword gathers only where a word
  // is not final yet.  Words are read and written while other threads jump them: any value seen is final or an earlier byte.
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t n4 = n >> 2;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u)
    {
    u32x4 w = __builtin_nontemporal_load((const u32x4*)src + i);
    bool mine = false;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (!(w[k] & PD_FINAL))
        {
        // up to PD_HOPS hops per round: a round is a pass over the whole workspace whatever is still open, so fewer, longer
        // rounds (chains shrink to a quarter per round)
        uint32_t x = __hip_atomic_load(&src[w[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int hop = 1; hop < PD_HOPS; ++hop)
          if (!(x & PD_FINAL))
            x = __hip_atomic_load(&src[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w[k] = x;
        mine = true;
        open = open || !(x & PD_FINAL);
        }
    if (mine)
      *((u32x4*)src + i) = w;
    }
  for (uint32_t i = 4u * n4 + blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
    {
    const uint32_t s = src[i];
    if (!(s & PD_FINAL))
      {
      // s < i: an earlier output byte; whatever it holds right now is final or a still earlier byte
      uint32_t t = __hip_atomic_load(&src[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int hop = 1; hop < PD_HOPS && !(t & PD_FINAL); ++hop)
        t = __hip_atomic_load(&src[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&src[i], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      open = open || !(t & PD_FINAL);
      }
    }
  __shared__ uint32_t any;
  if (threadIdx.x == 0)
    any = 0u;
  __syncthreads();
  if (__ballot(open) && (threadIdx.x & 63) == 0)
    any = 1u;
  __syncthreads();
  if (threadIdx.x == 0)
    {
    if (any)
      __hip_atomic_store(&ctl->changed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const uint32_t arrived = atomicAdd(&ctl->arrived, 1u);
    if (arrived == gridDim.x - 1u)
      {
      __threadfence();
      const uint32_t ch = __hip_atomic_load(&ctl->changed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ctl->done = ch ? 0u : 1u;
      ctl->changed = 0u;
      ctl->arrived = 0u;
      }
    }
  }

__global__ void __launch_bounds__(256) k_pd_gather(PdPlanes P, uint32_t n, uint32_t* __restrict__ status)
  {
  // a final word carries its byte: the plane is the low bytes of the words, sixteen words per thread and store
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const PdPlane& pl = P.p[blockIdx.y];
  const uint32_t* src = pl.src;
  uint8_t* out = pl.out;
  if (pl.clen == 0u || pl.ctl->error)
    return;
  uint32_t open = 0;
  const uint32_t n16 = n >> 4;
  const bool aligned = ((uintptr_t)out & 15u) == 0u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u)
    {
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      {
      const u32x4 w = __builtin_nontemporal_load((const u32x4*)src + 4u * i + (uint32_t)q);
      open |= ~(w[0] & w[1] & w[2] & w[3]);
      o[q] = (w[0] & 255u) | ((w[1] & 255u) << 8) | ((w[2] & 255u) << 16) | (w[3] << 24);
      }
    if (aligned)
      __builtin_nontemporal_store(o, (u32x4*)out + i);
    else
      for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 4; ++k)
          out[16u * i + 4u * (uint32_t)q + (uint32_t)k] = (uint8_t)(o[q] >> (8 * k));
    }
  for (uint32_t i = 16u * n16 + blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
    {
    const uint32_t w = src[i];
    open |= ~w;
    out[i] = (uint8_t)w;
    }
  if (__ballot((open & PD_FINAL) != 0u) && (threadIdx.x & 63) == 0)
    atomicOr(status, 8u);                           // a chain that never reached a literal: cannot happen after PD_ROUNDS rounds
  }

struct PdPlan { size_t tiles, tiles0, onpath, out_base, ctl, jobs, src, total; uint32_t job_cap, max_tiles; };

PdPlan pd_plan(uint32_t plane_bytes, uint32_t max_clen)
  {
  PdPlan p;
  p.max_tiles = (max_clen + PD_TILE - 1) / PD_TILE + 1;
  // every job covers >= PD_LONG output bytes or ends a run; a tile leaves fewer than PD_JCHUNK reserved places unused at its end, and so
  // does every run of more than PD_JCHUNK pieces
  p.job_cap = plane_bytes / PD_LONG + 2 * p.max_tiles + 16 + PD_JCHUNK * (p.max_tiles + plane_bytes / (PD_JCHUNK * PD_PIECE) + 1);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes + 16, 256); return at; };
  p.tiles = take(sizeof(Tile) * (size_t)p.max_tiles);
  p.tiles0 = take(sizeof(Tile) * (size_t)p.max_tiles);
  p.onpath = take(4 * (size_t)p.max_tiles);
  p.out_base = take(4 * (size_t)p.max_tiles);
  p.ctl = take(sizeof(Ctl));
  p.jobs = take(sizeof(Job) * (size_t)p.job_cap);
  p.src = take(4 * (size_t)plane_bytes);
  p.total = o;
  return p;
  }

} // namespace

uint32_t lz4_pdecode_threshold()
  {
  static uint32_t t = 0;
  if (!t)
    {
    const char* e = getenv("TRICO_LZ4_PDECODE_MIN");             // planes of at least this many bytes take the data-parallel decoder
    t = e ? (uint32_t)strtoul(e, nullptr, 10) : (1u << 20);
    if (t < 1) t = 1;
    }
  return t;
  }

size_t lz4_pdecode_workspace(uint32_t plane_bytes, const uint32_t* sizes, int nplanes)
  {
  uint32_t mx = 0;
  for (int c = 0; c < nplanes; ++c)
    mx = sizes[c] > mx ? sizes[c] : mx;
  return pd_plan(plane_bytes, mx).total * (size_t)nplanes;       // every plane has a workspace of its own: they are decoded together
  }

int launch_lz4_decode_parallel(const uint8_t* const d_payloads[8], const uint32_t sizes[8], int nplanes, uint8_t* d_planes, size_t plane_stride,
                               uint32_t plane_bytes, uint32_t* d_status, uint8_t* d_ws, size_t ws_bytes)
  {
  uint32_t mx = 0;
  for (int c = 0; c < nplanes; ++c)
    mx = sizes[c] > mx ? sizes[c] : mx;
  const PdPlan p = pd_plan(plane_bytes, mx);
  if (nplanes < 1 || nplanes > 8 || p.total * (size_t)nplanes > ws_bytes)
    {
    set_error("lz4 parallel decode: workspace too small");
    return 0;
    }
  hipStream_t st = current_stream();
  PdPlanes P;
  uint32_t max_nt = 1;
  for (int c = 0; c < 8; ++c)
    {
    PdPlane& q = P.p[c];
    uint8_t* w = d_ws + (size_t)(c < nplanes ? c : 0) * p.total;
    q.in = c < nplanes ? d_payloads[c] : nullptr;
    q.clen = c < nplanes ? sizes[c] : 0u;
    q.nt = (q.clen + PD_TILE - 1) / PD_TILE;
    q.out = d_planes + (size_t)(c < nplanes ? c : 0) * plane_stride;
    q.tiles = (Tile*)(w + p.tiles);
    q.tiles0 = (Tile*)(w + p.tiles0);
    q.onpath = (uint32_t*)(w + p.onpath);
    q.out_base = (uint32_t*)(w + p.out_base);
    q.ctl = (Ctl*)(w + p.ctl);
    q.jobs = (Job*)(w + p.jobs);
    q.src = (uint32_t*)(w + p.src);
    max_nt = q.nt > max_nt ? q.nt : max_nt;
    if (c < nplanes && !hip_ok(hipMemsetAsync(q.ctl, 0, sizeof(Ctl), st), "memset(lz4 decode control)"))
      return 0;
    }
  const unsigned sweep = 2048;                                     // blocks of the element-wise sweeps (grid-stride)
  const unsigned np = (unsigned)nplanes;
  // (a plane with no block at all - an empty plane is the one-byte block 0x00, lz4.c:1146-1172 with n = 0 - is reported by
  // the chain kernel; the other phases skip it)
  hipLaunchKernelGGL(k_pd_head, dim3(np), dim3(64), 0, st, P);
  hipLaunchKernelGGL(k_pd_tiles, dim3(max_nt, np), dim3(64), 0, st, P, 0);
  hipLaunchKernelGGL(k_pd_tiles, dim3(max_nt, np), dim3(64), 0, st, P, 1);
  hipLaunchKernelGGL(k_pd_chain, dim3(np), dim3(64), 0, st, P, plane_bytes, d_status);
  hipLaunchKernelGGL(k_pd_fill, dim3(max_nt, np), dim3(64), 0, st, P, plane_bytes, p.job_cap, d_status);
  hipLaunchKernelGGL(k_pd_jobs, dim3(1024, np), dim3(256), 0, st, P, p.job_cap, d_status);
  for (int r = 0; r < PD_ROUNDS; ++r)
    hipLaunchKernelGGL(k_pd_jump, dim3(sweep, np), dim3(256), 0, st, P, plane_bytes);
  hipLaunchKernelGGL(k_pd_gather, dim3(sweep, np), dim3(256), 0, st, P, plane_bytes, d_status);
  return hip_ok(hipGetLastError(), "lz4 parallel decode kernels") ? 1 : 0;
  }

} // namespace trico
