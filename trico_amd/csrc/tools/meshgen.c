/*
 * meshgen.c — synthetic mesh generators for tests and bench.
 * Integer-derived values only, so the arrays are bit-reproducible on any IEEE host.
 * Not part of the hot path; has no reference counterpart (the reference ships one STL fixture).
 */
#include <stdint.h>

#define GEN_API __attribute__((visibility("default")))

static inline uint32_t xs32(uint32_t* s)
  {
  uint32_t x = *s;
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;
  *s = x;
  return x;
  }

static void grid_cells_u32(uint32_t* t, uint32_t W, uint32_t H)
  {
  for (uint32_t j = 0; j < H; ++j)
    for (uint32_t i = 0; i < W; ++i)
      {
      const uint32_t a = j * W + i, b = j * W + (i + 1) % W;
      const uint32_t c = ((j + 1) % H) * W + i, d = ((j + 1) % H) * W + (i + 1) % W;
      t[0] = a; t[1] = b; t[2] = c; t[3] = b; t[4] = d; t[5] = c;
      t += 6;
      }
  }

/* grid(W,H): nv = W*H float xyz vertices, nt = 2*W*H u32 triangles */
GEN_API void trico_gen_grid(uint32_t W, uint32_t H, uint32_t seed, float* v, uint32_t* t)
  {
  uint32_t s = seed;
  for (uint32_t j = 0; j < H; ++j)
    for (uint32_t i = 0; i < W; ++i)
      {
      const uint32_t r = xs32(&s);
      v[0] = (float)i * 0.25f;
      v[1] = (float)j * 0.25f;
      v[2] = (float)((i * i + 3u * j * j) % 4096u) * (1.f / 64.f) + (float)(r & 0xffu) * (1.f / 1024.f);
      v += 3;
      }
  if (t) grid_cells_u32(t, W, H);
  }

/* multi(W,H): double xyz, double normals, float uv, u64 triangles (any output may be NULL) */
GEN_API void trico_gen_multi(uint32_t W, uint32_t H, uint32_t seed, double* v, double* nrm, float* uv, uint64_t* t)
  {
  uint32_t s = seed;
  for (uint32_t j = 0; j < H; ++j)
    for (uint32_t i = 0; i < W; ++i)
      {
      const uint32_t r = xs32(&s);
      if (v)
        {
        v[0] = (double)i * 0.25;
        v[1] = (double)j * 0.25;
        v[2] = (double)((i * i + 3u * j * j) % 4096u) / 64.0 + (double)(r & 0xffffu) / 4194304.0;
        v += 3;
        }
      if (nrm)
        {
        nrm[0] = (double)((int32_t)((r >> 8) & 0x3ffu) - 512) / 512.0;
        nrm[1] = (double)((int32_t)((r >> 18) & 0x3ffu) - 512) / 512.0;
        nrm[2] = 1.0;
        nrm += 3;
        }
      if (uv)
        {
        uv[0] = (float)i / 16384.f;
        uv[1] = (float)j / 16384.f;
        uv += 2;
        }
      }
  if (t)
    for (uint32_t j = 0; j < H; ++j)
      for (uint32_t i = 0; i < W; ++i)
        {
        const uint64_t a = (uint64_t)j * W + i, b = (uint64_t)j * W + (i + 1) % W;
        const uint64_t c = (uint64_t)((j + 1) % H) * W + i, d = (uint64_t)((j + 1) % H) * W + (i + 1) % W;
        t[0] = a; t[1] = b; t[2] = c; t[3] = b; t[4] = d; t[5] = c;
        t += 6;
        }
  }

/* walk(W,H): random-walk float vertices (bunny-like ratios), pseudo-random local triangles */
GEN_API void trico_gen_walk(uint32_t W, uint32_t H, uint32_t seed, float* v, uint32_t* t)
  {
  const uint32_t nv = W * H, nt = 2u * nv;
  uint32_t s = seed;
  int32_t q[3] = { 0, 0, 0 };
  for (uint32_t k = 0; k < nv; ++k)
    {
    const uint32_t r = xs32(&s);
    for (int c = 0; c < 3; ++c)
      {
      q[c] += (int32_t)((r >> (8 * c)) & 0xffu) - 128;
      if (q[c] > 8000000) q[c] -= 16000000;
      else if (q[c] < -8000000) q[c] += 16000000;
      v[c] = (float)q[c] * (1.f / 1024.f);
      }
    v += 3;
    }
  if (t)
    for (uint32_t k = 0; k < nt; ++k)
      {
      const uint32_t r = xs32(&s);
      const uint32_t a = (k / 2u) % nv;
      t[0] = a;
      t[1] = (a + 1u + (r & 0x3fu)) % nv;
      t[2] = (a + 1u + ((r >> 6) & 0x3ffu)) % nv;
      t += 3;
      }
  }
