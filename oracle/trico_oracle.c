/*
 * trico_oracle.c — CPU restatement of Trico's encode/decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there
 * only as the checker.  The product path (trico_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  This restatement is checked byte-for-byte against the reference
 * compiled from /root/reference (oracle/_ref/libtrico_ref.so, see oracle/Makefile) by
 * tests/test_oracle_vs_reference.py, and against the committed fixtures in tests/golden/
 * that were generated from that same compiled reference (oracle/gen_golden.py).
 *
 * Written from the algorithm description, one generic-width coder
 * instead of the reference's two unrolled ones.  Reference locations restated:
 *   fp coder   : trico/floating_point_stream_compression.c:86-210 (f32 enc), 212-417 (f32 dec),
 *                576-800 (f64 enc), 803-1164 (f64 dec), helpers 12-84 and 421-573
 *   planes     : trico/transpose_aos_to_soa.c:8-147
 *   LZ4 block  : lz4/lz4.c:793-1181 (LZ4_compress_generic as reached from LZ4_compress_default,
 *                lz4.c:1271 -> 1252 -> 1184), lz4.c:1657-2072 (LZ4_decompress_generic, safe mode)
 *   container  : trico/trico.c:12-124 (archive buffer + header), 215-858 (stream writers)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * FCM/DFCM XOR-residual coder, generic over the value width W (32 or 64 bits).
 * Values travel as uint64_t; arithmetic is reduced modulo 2^W with `wmask`.
 * ------------------------------------------------------------------------------------------ */

static unsigned byte_len(uint64_t x)
  {
  unsigned n = 0;
  while (x) { ++n; x >>= 8; }
  return n;
  }

static uint64_t load_val(const void* p, uint32_t i, int W)
  {
  if (W == 32) { uint32_t v; memcpy(&v, (const uint8_t*)p + 4 * (size_t)i, 4); return v; }
  uint64_t v; memcpy(&v, (const uint8_t*)p + 8 * (size_t)i, 8); return v;
  }

static void store_val(void* p, uint32_t i, int W, uint64_t v)
  {
  if (W == 32) { uint32_t w = (uint32_t)v; memcpy((uint8_t*)p + 4 * (size_t)i, &w, 4); }
  else memcpy((uint8_t*)p + 8 * (size_t)i, &v, 8);
  }

static unsigned clamp_exp(unsigned e)
  {
  e &= ~1u;               /* fpsc.c:88-93 / 578-583: force even, cap at 30 */
  if (e > 30) e = 30;
  return e;
  }

/* Worst-case payload size; unlike fpsc.c:95/585 it includes the 5 header bytes. */
ORC_API uint64_t orc_fpc_bound(uint32_t n, int W)
  {
  const uint64_t g = (W == 32) ? 8 : 2, hdr = (W == 32) ? 3 : 1;
  return 5 + (uint64_t)(W / 8) * n + hdr * ((n + g - 1) / g + 1) + g;
  }

static uint8_t* put_be(uint8_t* o, uint64_t x, unsigned nb)
  {
  while (nb) { --nb; *o++ = (uint8_t)(x >> (8 * nb)); }
  return o;
  }

/* Emit one group (fpsc.c:12-74 for W=32: 3-byte big-endian header of 3-bit codes;
 * fpsc.c:421-561 for W=64: 1 byte of two 4-bit codes) followed by the residual bytes. */
static uint8_t* emit_group(uint8_t* o, int W, const unsigned* code, const uint64_t* x1, const uint64_t* x2)
  {
  const unsigned g = (W == 32) ? 8 : 2, wb = (unsigned)W / 8;
  if (W == 32)
    {
    uint32_t bc = 0;
    for (unsigned k = 0; k < 8; ++k) bc |= (uint32_t)code[k] << (3 * k);
    o = put_be(o, bc, 3);
    }
  else
    *o++ = (uint8_t)((code[1] << 4) | code[0]);
  for (unsigned k = 0; k < g; ++k)
    {
    if (code[k] <= wb) o = put_be(o, x1[k], code[k]);
    else o = put_be(o, x2[k], code[k] - wb);
    }
  return o;
  }

/* Encoder: `in` holds n raw W-bit patterns.  Returns the number of payload bytes written to
 * `out` (caller provides orc_fpc_bound bytes).  Exponents as passed by the caller (trico.c
 * always passes 4,10 for float and 20,20 for double). */
ORC_API uint32_t orc_fpc_encode(const void* in, uint32_t n, int W, unsigned e1, unsigned e2, uint8_t* out)
  {
  e1 = clamp_exp(e1); e2 = clamp_exp(e2);
  const uint64_t wmask = (W == 32) ? 0xffffffffull : ~0ull;
  const unsigned g = (W == 32) ? 8 : 2, wb = (unsigned)W / 8;
  const uint64_t m1 = ((uint64_t)1 << e1) - 1, m2 = ((uint64_t)1 << e2) - 1;
  uint64_t* T1 = (uint64_t*)calloc((size_t)1 << e1, 8);
  uint64_t* T2 = (uint64_t*)calloc((size_t)1 << e2, 8);
  uint64_t h1 = 0, h2 = 0, p1 = 0, p2 = 0, last = 0;
  uint64_t x1[8], x2[8];
  unsigned code[8];
  uint8_t* o = out;
  *o++ = (uint8_t)(((e1 >> 1) << 4) | (e2 >> 1));   /* fpsc.c:120-126 */
  o = put_be(o, n, 4);
  unsigned j = 0;
  for (uint32_t i = 0; i < n; ++i)
    {
    j = i % g;
    const uint64_t v = load_val(in, i, W);
    x1[j] = v ^ p1;
    T1[h1] = v;
    h1 = (e1 ? (((h1 << e1) ^ (v >> (W - e1))) & m1) : 0);
    p1 = T1[h1];
    const uint64_t s = (v - last) & wmask;
    x2[j] = v ^ ((last + p2) & wmask);
    last = v;
    T2[h2] = s;
    h2 = (e2 ? (((h2 << (e2 / 2)) ^ (s >> (W - e2))) & m2) : 0);
    p2 = T2[h2];
    const unsigned n1 = byte_len(x1[j]);
    unsigned n2 = byte_len(x2[j]);
    if (n2 == 0) n2 = 1;
    code[j] = (n1 <= 1) ? n1 : (n2 < n1 ? wb + n2 : n1);
    if (j == g - 1) o = emit_group(o, W, code, x1, x2);
    }
  if (n == 0 || j != g - 1)
    {
    /* tail padding (fpsc.c:196-204 / 789-794): missing slots are code 1 with a zero byte.
     * n == 0 is undefined in the reference (reads uninitialised slot 0); defined here as a
     * full pad group. */
    for (unsigned l = (n == 0) ? 0 : j + 1; l < g; ++l) { code[l] = 1; x1[l] = 0; x2[l] = 0; }
    o = emit_group(o, W, code, x1, x2);
    }
  free(T1); free(T2);
  return (uint32_t)(o - out);
  }

/* Number of values announced by a payload header, or -1 if the payload is too short / W mismatch
 * cannot be told (the width is implied by the stream type, not stored). */
ORC_API int64_t orc_fpc_peek_count(const uint8_t* in, uint64_t in_len)
  {
  if (in_len < 5) return -1;
  return ((int64_t)in[1] << 24) | ((int64_t)in[2] << 16) | ((int64_t)in[3] << 8) | in[4];
  }

/* Decoder (fpsc.c:212-417 / 803-1164), bounds-checked.  Writes exactly the announced count of
 * values (must be <= out_cap).  Returns 1 on success, 0 on malformed input. */
ORC_API int orc_fpc_decode(const uint8_t* in, uint64_t in_len, int W, void* out, uint32_t out_cap, uint32_t* n_out)
  {
  if (in_len < 5) return 0;
  const unsigned e1 = (unsigned)(in[0] >> 4) << 1, e2 = (unsigned)(in[0] & 15) << 1;
  if (e1 == 0 || e2 == 0 || e1 > 26 || e2 > 26) return 0;   /* exponent 0 is UB in the reference */
  const uint32_t n = ((uint32_t)in[1] << 24) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 8) | in[4];
  if (n > out_cap) return 0;
  const uint64_t wmask = (W == 32) ? 0xffffffffull : ~0ull;
  const unsigned g = (W == 32) ? 8 : 2, wb = (unsigned)W / 8;
  const uint64_t m1 = ((uint64_t)1 << e1) - 1, m2 = ((uint64_t)1 << e2) - 1;
  uint64_t* T1 = (uint64_t*)calloc((size_t)1 << e1, 8);
  uint64_t* T2 = (uint64_t*)calloc((size_t)1 << e2, 8);
  uint64_t h1 = 0, h2 = 0, p1 = 0, p2 = 0, last = 0;
  uint64_t pos = 5;
  int ok = 1;
  for (uint32_t i = 0; i < n && ok; i += g)
    {
    unsigned code[8];
    if (W == 32)
      {
      if (pos + 3 > in_len) { ok = 0; break; }
      const uint32_t bc = ((uint32_t)in[pos] << 16) | ((uint32_t)in[pos + 1] << 8) | in[pos + 2];
      pos += 3;
      for (unsigned k = 0; k < 8; ++k) code[k] = (bc >> (3 * k)) & 7;
      }
    else
      {
      if (pos + 1 > in_len) { ok = 0; break; }
      const unsigned bc = in[pos++];
      code[0] = bc & 15; code[1] = bc >> 4;
      }
    const unsigned cnt = (n - i < g) ? (n - i) : g;
    for (unsigned k = 0; k < cnt; ++k)
      {
      const unsigned nb = code[k] <= wb ? code[k] : code[k] - wb;
      if (pos + nb > in_len) { ok = 0; break; }
      uint64_t x = 0;
      for (unsigned b = 0; b < nb; ++b) x = (x << 8) | in[pos++];
      if (code[k] > wb) p1 = p2;                         /* fpsc.c:310-311 */
      const uint64_t v = x ^ p1;
      T1[h1] = v;
      h1 = ((h1 << e1) ^ (v >> (W - e1))) & m1;
      p1 = T1[h1];
      const uint64_t s = (v - last) & wmask;
      T2[h2] = s;
      h2 = ((h2 << (e2 / 2)) ^ (s >> (W - e2))) & m2;
      p2 = (v + T2[h2]) & wmask;                         /* decoder keeps last+stride (fpsc.c:323) */
      last = v;
      store_val(out, i + k, W, v);
      }
    }
  free(T1); free(T2);
  if (ok && n_out) *n_out = n;
  return ok;
  }

/* ------------------------------------------------------------------------------------------
 * AoS <-> SoA (transpose_aos_to_soa.c:8-82) and byte planes (84-147).
 * ------------------------------------------------------------------------------------------ */

/* de-interleave `arity` components of `elem` bytes each: out[c*n + i] = in[i*arity + c] */
ORC_API void orc_deinterleave(const void* in, uint32_t n, unsigned arity, unsigned elem, void* out)
  {
  const uint8_t* s = (const uint8_t*)in;
  uint8_t* d = (uint8_t*)out;
  for (unsigned c = 0; c < arity; ++c)
    for (uint32_t i = 0; i < n; ++i)
      memcpy(d + ((size_t)c * n + i) * elem, s + ((size_t)i * arity + c) * elem, elem);
  }

ORC_API void orc_interleave(const void* in, uint32_t n, unsigned arity, unsigned elem, void* out)
  {
  const uint8_t* s = (const uint8_t*)in;
  uint8_t* d = (uint8_t*)out;
  for (unsigned c = 0; c < arity; ++c)
    for (uint32_t i = 0; i < n; ++i)
      memcpy(d + ((size_t)i * arity + c) * elem, s + ((size_t)c * n + i) * elem, elem);
  }

/* plane k (k = 0..width-1) holds byte k (little-endian order) of every element */
ORC_API void orc_split_planes(const void* in, uint32_t n, unsigned width, uint8_t* planes)
  {
  const uint8_t* s = (const uint8_t*)in;
  for (unsigned k = 0; k < width; ++k)
    for (uint32_t i = 0; i < n; ++i)
      planes[(size_t)k * n + i] = s[(size_t)i * width + k];
  }

ORC_API void orc_merge_planes(const uint8_t* planes, uint32_t n, unsigned width, void* out)
  {
  uint8_t* d = (uint8_t*)out;
  for (unsigned k = 0; k < width; ++k)
    for (uint32_t i = 0; i < n; ++i)
      d[(size_t)i * width + k] = planes[(size_t)k * n + i];
  }

/* ------------------------------------------------------------------------------------------
 * LZ4 block compressor exactly as LZ4 1.9.2's LZ4_compress_default behaves when the destination
 * is >= LZ4_compressBound (trico.c:343-346): greedy parse, acceleration 1, fresh zeroed table,
 * 13-bit u16 table + hash4 below 65547 bytes, 12-bit u32 table + hash5 otherwise.
 * ------------------------------------------------------------------------------------------ */

ORC_API uint32_t orc_lz4_bound(uint32_t n) { return n + n / 255 + 16; }   /* lz4.h:171 */

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

typedef struct { int small; uint32_t t32[4096]; uint16_t t16[8192]; } lz4_tab;

static uint32_t lz4_hash(const lz4_tab* t, const uint8_t* p)
  {
  if (t->small) return (rd32(p) * 2654435761u) >> 19;                 /* lz4.c:637-638 */
  return (uint32_t)(((rd64(p) << 24) * 889523592379ull) >> 52);       /* lz4.c:643-648 */
  }
static uint32_t tab_get(const lz4_tab* t, uint32_t h) { return t->small ? t->t16[h] : t->t32[h]; }
static void tab_set(lz4_tab* t, uint32_t h, uint32_t pos) { if (t->small) t->t16[h] = (uint16_t)pos; else t->t32[h] = pos; }

static uint8_t* put_len(uint8_t* op, uint32_t len)   /* length extension bytes after a 15 nibble */
  {
  for (; len >= 255; len -= 255) *op++ = 255;
  *op++ = (uint8_t)len;
  return op;
  }

ORC_API int orc_lz4_compress(const uint8_t* src, uint32_t n, uint8_t* dst)
  {
  if (n > 0x7E000000u) return 0;                                       /* lz4.c:844 */
  lz4_tab* t = (lz4_tab*)calloc(1, sizeof(lz4_tab));
  t->small = n < 65547u;                                               /* lz4.c:570,1190 */
  uint8_t* op = dst;
  uint32_t anchor = 0, ip = 0;
  if (n >= 13)                                                         /* lz4.c:863 */
    {
    const uint32_t mfl1 = n - 11;      /* mflimitPlusOne (lz4.c:825) */
    const uint32_t mlim = n - 5;       /* matchlimit (lz4.c:826) */
    tab_set(t, lz4_hash(t, src), 0);
    ip = 1;
    uint32_t fh = lz4_hash(t, src + 1);
    for (;;)
      {
      uint32_t cand, fwd = ip, step = 1, nb = 64;
      uint8_t* token;
      int done = 0;
      for (;;)                                                         /* lz4.c:898-956 */
        {
        const uint32_t h = fh, cur = fwd;
        cand = tab_get(t, h);
        ip = fwd;
        fwd += step;
        step = nb++ >> 6;
        if (fwd > mfl1) { done = 1; break; }
        fh = lz4_hash(t, src + fwd);
        tab_set(t, h, cur);
        if (!t->small && cand + 65535u < cur) continue;
        if (rd32(src + cand) == rd32(src + ip)) break;
        }
      if (done) break;
      while (ip > anchor && cand > 0 && src[ip - 1] == src[cand - 1]) { --ip; --cand; }   /* lz4.c:960-961 */
      {
      const uint32_t lit = ip - anchor;
      token = op++;
      if (lit >= 15) { *token = 0xf0; op = put_len(op, lit - 15); }
      else *token = (uint8_t)(lit << 4);
      memcpy(op, src + anchor, lit);
      op += lit;
      }
      for (;;)
        {
        /* next_match: offset, match length (lz4.c:1007-1077) */
        *op++ = (uint8_t)(ip - cand);
        *op++ = (uint8_t)((ip - cand) >> 8);
        uint32_t m = 0;
        while (ip + 4 + m < mlim && src[ip + 4 + m] == src[cand + 4 + m]) ++m;
        ip += m + 4;
        if (m >= 15) { *token += 15; op = put_len(op, m - 15); }
        else *token += (uint8_t)m;
        anchor = ip;
        if (ip >= mfl1) { done = 1; break; }
        tab_set(t, lz4_hash(t, src + ip - 2), ip - 2);                /* lz4.c:1088 */
        const uint32_t h = lz4_hash(t, src + ip);
        cand = tab_get(t, h);
        tab_set(t, h, ip);
        if ((t->small || cand + 65535u >= ip) && rd32(src + cand) == rd32(src + ip))
          { token = op++; *token = 0; continue; }                      /* lz4.c:1101-1138 */
        break;
        }
      if (done) break;
      fh = lz4_hash(t, src + (++ip));
      }
    }
  {
  const uint32_t run = n - anchor;                                     /* lz4.c:1146-1172 */
  if (run >= 15) { *op++ = 0xf0; op = put_len(op, run - 15); }
  else *op++ = (uint8_t)(run << 4);
  memcpy(op, src + anchor, run);
  op += run;
  }
  free(t);
  return (int)(op - dst);
  }

/* LZ4 block decoder, safe: returns bytes written, or -1 on malformed input / overflow. */
ORC_API int orc_lz4_decompress(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap)
  {
  uint32_t ip = 0, op = 0;
  if (n == 0) return -1;
  for (;;)
    {
    if (ip >= n) return -1;
    const unsigned tok = src[ip++];
    uint32_t lit = tok >> 4;
    if (lit == 15)
      {
      unsigned b;
      do { if (ip >= n) return -1; b = src[ip++]; lit += b; } while (b == 255);
      }
    if (lit > n - ip || lit > cap - op) return -1;
    memcpy(dst + op, src + ip, lit);
    ip += lit; op += lit;
    if (ip == n) break;                     /* last sequence: literals only */
    if (n - ip < 2) return -1;
    const uint32_t off = src[ip] | ((uint32_t)src[ip + 1] << 8);
    ip += 2;
    if (off == 0 || off > op) return -1;
    uint32_t ml = tok & 15;
    if (ml == 15)
      {
      unsigned b;
      do { if (ip >= n) return -1; b = src[ip++]; ml += b; } while (b == 255);
      }
    ml += 4;
    if (ml > cap - op) return -1;
    for (uint32_t k = 0; k < ml; ++k) dst[op + k] = dst[op + k - off];
    op += ml;
    }
  return (int)op;
  }

/* ------------------------------------------------------------------------------------------
 * Container (trico.c:90-98 header; 215-858 writers).  One generic appender per payload family;
 * the caller (tests) chooses type tag / count field / arity exactly as the reference writers do.
 * ------------------------------------------------------------------------------------------ */

typedef struct { uint8_t* buf; uint64_t size, cap; } orc_arch;

static void arch_put(orc_arch* a, const void* p, uint64_t n)
  {
  if (a->size + n > a->cap)
    {
    a->cap = (a->size + n) * 2 + 64;
    a->buf = (uint8_t*)realloc(a->buf, a->cap);
    }
  memcpy(a->buf + a->size, p, n);
  a->size += n;
  }

ORC_API void* orc_arch_new(void)
  {
  orc_arch* a = (orc_arch*)calloc(1, sizeof(orc_arch));
  const uint32_t magic = 0x6f637254u, version = 0;
  arch_put(a, &magic, 4);
  arch_put(a, &version, 4);
  return a;
  }
ORC_API void orc_arch_free(void* h) { orc_arch* a = (orc_arch*)h; free(a->buf); free(a); }
ORC_API const uint8_t* orc_arch_data(void* h) { return ((orc_arch*)h)->buf; }
ORC_API uint64_t orc_arch_size(void* h) { return ((orc_arch*)h)->size; }

/* floating-point stream: `arity` interleaved components of n elements, W = 32|64 */
ORC_API void orc_arch_write_fp(void* h, unsigned type, uint32_t count_field, const void* data, uint32_t n, unsigned arity, int W)
  {
  orc_arch* a = (orc_arch*)h;
  const uint8_t tag = (uint8_t)type;
  arch_put(a, &tag, 1);
  arch_put(a, &count_field, 4);
  const unsigned elem = (unsigned)W / 8;
  uint8_t* soa = (uint8_t*)malloc((size_t)n * arity * elem + 1);
  orc_deinterleave(data, n, arity, elem, soa);
  uint8_t* tmp = (uint8_t*)malloc(orc_fpc_bound(n, W));
  for (unsigned c = 0; c < arity; ++c)
    {
    const uint32_t nb = orc_fpc_encode(soa + (size_t)c * n * elem, n, W, W == 32 ? 4 : 20, W == 32 ? 10 : 20, tmp);
    arch_put(a, &nb, 4);
    arch_put(a, tmp, nb);
    }
  free(tmp); free(soa);
  }

/* integer stream: n elements of `width` bytes, one LZ4 block per byte plane */
ORC_API void orc_arch_write_int(void* h, unsigned type, uint32_t count_field, const void* data, uint32_t n, unsigned width)
  {
  orc_arch* a = (orc_arch*)h;
  const uint8_t tag = (uint8_t)type;
  arch_put(a, &tag, 1);
  arch_put(a, &count_field, 4);
  uint8_t* planes = (uint8_t*)malloc((size_t)n * width + 1);
  orc_split_planes(data, n, width, planes);
  uint8_t* tmp = (uint8_t*)malloc(orc_lz4_bound(n));
  for (unsigned k = 0; k < width; ++k)
    {
    const uint32_t nb = (uint32_t)orc_lz4_compress(planes + (size_t)k * n, n, tmp);
    arch_put(a, &nb, 4);
    arch_put(a, tmp, nb);
    }
  free(tmp); free(planes);
  }
