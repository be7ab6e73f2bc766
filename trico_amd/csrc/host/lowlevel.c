/* lowlevel.c — the reference's stand-alone coder and transpose entry points (floating_point_stream_compression.h,
 * transpose_aos_to_soa.h) on top of the device shim.  Thin host glue: argument marshalling, the host-side output
 * buffers the reference's conventions ask for, and the walk that finds where a payload ends. */
#include "trico/floating_point_stream_compression.h"
#include "trico/transpose_aos_to_soa.h"
#include "trico/trico_hip.h"

#include <stdlib.h>
#include <string.h>

/* ---- coder ------------------------------------------------------------------------------------------------ */

static void compress_any(uint32_t* payload_bytes, uint8_t** payload, const void* input, uint32_t count, int width,
                         uint64_t e1, uint64_t e2)
  {
  *payload = NULL;
  *payload_bytes = 0;
  /* any exponent is legal: the reference rounds odd ones down and caps at 30 (fpsc.c:88-93, 578-583); the shim does the same */
  if (e1 > 30) e1 = 30;
  if (e2 > 30) e2 = 30;
  trico_hip_ctx* ctx = trico_hip_ctx_create();
  if (!ctx)
    return;
  uint32_t sizes[3] = { 0, 0, 0 };
  if (trico_hip_fpc_encode_ex(ctx, input, count, 1, width, (uint32_t)e1, (uint32_t)e2, sizes))
    {
    uint8_t* buf = (uint8_t*)malloc(sizes[0] ? sizes[0] : 1);
    if (buf && trico_hip_fetch_payload(ctx, 0, buf))
      {
      *payload = buf;
      *payload_bytes = sizes[0];
      }
    else
      free(buf);
    }
  trico_hip_ctx_destroy(ctx);
  }

void trico_compress(uint32_t* payload_bytes, uint8_t** payload, const float* input, const uint32_t count,
                    uint32_t table1_exponent, uint32_t table2_exponent)
  {
  compress_any(payload_bytes, payload, input, count, 4, table1_exponent, table2_exponent);
  }

void trico_compress_double_precision(uint32_t* payload_bytes, uint8_t** payload, const double* input, const uint32_t count,
                                     uint64_t table1_exponent, uint64_t table2_exponent)
  {
  compress_any(payload_bytes, payload, input, count, 8, table1_exponent, table2_exponent);
  }

/* The payload does not say how long it is (fpsc.c:212-245 just reads on).  Layout: 1 byte table exponents, 4 bytes
 * count (big endian), then groups: floats — 3 header bytes holding eight 3-bit codes, then the residual bytes of
 * the eight values (codes 0..4: that many bytes, 5..7: code - 4); doubles — 1 header byte holding two 4-bit codes
 * (0..8: that many bytes, 9..15: code - 8), then the residuals of the two values.  A float stream always ends with
 * a complete group (the tail is padded with code 1, fpsc.c:196-204), a double stream with a complete pair. */
static uint64_t payload_length(const uint8_t* p, int width, uint32_t* count)
  {
  const uint32_t n = ((uint32_t)p[1] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 8) | p[4];
  *count = n;
  uint64_t q = 5;
  if (width == 4)
    {
    const uint64_t groups = n ? ((uint64_t)n + 7) / 8 : 1;              /* n == 0: one pad group (this library's writer) */
    for (uint64_t g = 0; g < groups; ++g)
      {
      const uint32_t bc = ((uint32_t)p[q] << 16) | ((uint32_t)p[q + 1] << 8) | p[q + 2];
      q += 3;
      for (int k = 0; k < 8; ++k)
        {
        const uint32_t c = (bc >> (3 * k)) & 7u;
        q += c <= 4 ? c : c - 4;
        }
      }
    }
  else
    {
    const uint64_t groups = n ? ((uint64_t)n + 1) / 2 : 1;
    for (uint64_t g = 0; g < groups; ++g)
      {
      const uint32_t h = p[q++];
      const uint32_t c0 = h & 15u, c1 = h >> 4;
      q += (c0 <= 8 ? c0 : c0 - 8) + (c1 <= 8 ? c1 : c1 - 8);
      }
    }
  return q;
  }

static void decompress_any(uint32_t* count, void** values, const uint8_t* payload, int width)
  {
  *values = NULL;
  *count = 0;
  if (!payload || trico_hip_pointer_is_device(payload))
    return;                                   /* the end of the stream has to be found on the host */
  uint32_t n = 0;
  const uint64_t len = payload_length(payload, width, &n);
  if (len > 0xffffffffull)
    return;
  trico_hip_ctx* ctx = trico_hip_ctx_create();
  if (!ctx)
    return;
  void* out = malloc((size_t)n * (size_t)width + 1);
  const uint8_t* pay[3] = { payload, NULL, NULL };
  const uint32_t sizes[3] = { (uint32_t)len, 0, 0 };
  if (out && trico_hip_fpc_decode(ctx, pay, sizes, 1, width, n, out))
    {
    *values = out;
    *count = n;
    }
  else
    free(out);
  trico_hip_ctx_destroy(ctx);
  }

void trico_decompress(uint32_t* count, float** values, const uint8_t* payload)
  {
  decompress_any(count, (void**)values, payload, 4);
  }

void trico_decompress_double_precision(uint32_t* count, double** values, const uint8_t* payload)
  {
  decompress_any(count, (void**)values, payload, 8);
  }

/* ---- transposes -------------------------------------------------------------------------------------------- */

static void split(const void* aos, uint32_t n, int arity, int width, void* const* comps)
  {
  trico_hip_ctx* ctx = trico_hip_ctx_create();
  if (!ctx)
    return;
  (void)trico_hip_split_components(ctx, aos, n, arity, width, comps);
  trico_hip_ctx_destroy(ctx);
  }

static void merge(void* aos, uint32_t n, int arity, int width, const void* const* comps)
  {
  trico_hip_ctx* ctx = trico_hip_ctx_create();
  if (!ctx)
    return;
  (void)trico_hip_merge_components(ctx, comps, n, arity, width, aos);
  trico_hip_ctx_destroy(ctx);
  }

void trico_transpose_xyz_aos_to_soa(float** x, float** y, float** z, const float* xyz, uint32_t n)
  { void* c[3] = { *x, *y, *z }; split(xyz, n, 3, 4, c); }
void trico_transpose_xyz_soa_to_aos(float** xyz, const float* x, const float* y, const float* z, uint32_t n)
  { const void* c[3] = { x, y, z }; merge(*xyz, n, 3, 4, c); }
void trico_transpose_xyz_aos_to_soa_double_precision(double** x, double** y, double** z, const double* xyz, uint32_t n)
  { void* c[3] = { *x, *y, *z }; split(xyz, n, 3, 8, c); }
void trico_transpose_xyz_soa_to_aos_double_precision(double** xyz, const double* x, const double* y, const double* z, uint32_t n)
  { const void* c[3] = { x, y, z }; merge(*xyz, n, 3, 8, c); }
void trico_transpose_uv_aos_to_soa(float** u, float** v, const float* uv, uint32_t n)
  { void* c[2] = { *u, *v }; split(uv, n, 2, 4, c); }
void trico_transpose_uv_soa_to_aos(float** uv, const float* u, const float* v, uint32_t n)
  { const void* c[2] = { u, v }; merge(*uv, n, 2, 4, c); }
void trico_transpose_uv_aos_to_soa_double_precision(double** u, double** v, const double* uv, uint32_t n)
  { void* c[2] = { *u, *v }; split(uv, n, 2, 8, c); }
void trico_transpose_uv_soa_to_aos_double_precision(double** uv, const double* u, const double* v, uint32_t n)
  { const void* c[2] = { u, v }; merge(*uv, n, 2, 8, c); }

void trico_transpose_uint16_aos_to_soa(uint8_t** p0, uint8_t** p1, const uint16_t* values, uint32_t n)
  { void* c[2] = { *p0, *p1 }; split(values, n, 2, 1, c); }
void trico_transpose_uint16_soa_to_aos(uint16_t** values, const uint8_t* p0, const uint8_t* p1, uint32_t n)
  { const void* c[2] = { p0, p1 }; merge(*values, n, 2, 1, c); }
void trico_transpose_uint32_aos_to_soa(uint8_t** p0, uint8_t** p1, uint8_t** p2, uint8_t** p3, const uint32_t* values, uint32_t n)
  { void* c[4] = { *p0, *p1, *p2, *p3 }; split(values, n, 4, 1, c); }
void trico_transpose_uint32_soa_to_aos(uint32_t** values, const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3,
                                       uint32_t n)
  { const void* c[4] = { p0, p1, p2, p3 }; merge(*values, n, 4, 1, c); }
void trico_transpose_uint64_aos_to_soa(uint8_t** p0, uint8_t** p1, uint8_t** p2, uint8_t** p3, uint8_t** p4, uint8_t** p5,
                                       uint8_t** p6, uint8_t** p7, const uint64_t* values, uint32_t n)
  { void* c[8] = { *p0, *p1, *p2, *p3, *p4, *p5, *p6, *p7 }; split(values, n, 8, 1, c); }
void trico_transpose_uint64_soa_to_aos(uint64_t** values, const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3,
                                       const uint8_t* p4, const uint8_t* p5, const uint8_t* p6, const uint8_t* p7, uint32_t n)
  { const void* c[8] = { p0, p1, p2, p3, p4, p5, p6, p7 }; merge(*values, n, 8, 1, c); }
