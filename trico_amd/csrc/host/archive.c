/*
 * archive.c — host side of the drop-in boundary: the .trc container (framing, archive handle,
 * peeks, skip) behind the reference's 54-function C API, in plain C11.
 *
 * Reference behaviour restated (not copied) from trico/trico.c:
 *   archive handle + append/read primitives  trico.c:12-88
 *   header "Trco" + version                  trico.c:90-124
 *   open/close/accessors                     trico.c:126-213
 *   stream writers                           trico.c:215-858
 *   count peeks                              trico.c:860-941
 *   stream readers + skip                    trico.c:943-1698
 * The reference's ~40 hand-unrolled writers/readers collapse here into four generic routines
 * (fp / integer x write / read); all compression work is delegated to the HIP shim
 * (include/trico/trico_hip.h).  There is no CPU compression path in this library.
 *
 * Deliberate divergences from the reference (SURVEY.md §8 "quirks"):
 *   - buffer growth is geometric instead of exact-fit (same bytes, fewer reallocs);
 *   - a failing writer writes nothing, a failing reader consumes nothing;
 *   - compressed payloads are decoded in place from the borrowed archive bytes (no malloc+memcpy);
 *   - trico_read_attributes_uint8 writes into *attrib (the reference's (char*)attrib is a bug);
 *   - double uv readers accept tags 6/8 per the enum while the writers emit 5/7 like the reference;
 *   - the next-stream tag is read into a byte, integers are little-endian by construction.
 */
#define _DEFAULT_SOURCE                /* madvise */
#include "trico/trico.h"
#include "trico/trico_hip.h"

#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

struct trico_archive
  {
  /* write side */
  uint8_t* buffer;
  uint64_t buffer_size;
  uint64_t used;
  int buffer_on_device;
  /* read side (borrowed) */
  const uint8_t* data;
  uint64_t data_size;
  uint64_t pos;
  int data_on_device;

  uint32_t version;
  uint8_t next_stream_type;
  int writable;
  trico_hip_ctx* ctx;   /* created on first use */
  uint32_t other_writer_streams;   /* streams delivered although the reference's encoder would not have written their payload (trico_hip_set_strict) */
  /* read-ahead: streams that were decoded ahead as one batch, keyed by the cursor position of their count field */
  struct ra_entry* ra;
  int ra_n;
  int ra_started;
  /* device-resident archive: the framing bytes (type, count, component sizes) of the next streams, fetched with one
   * device-side walk and one copy instead of one copy per field (trico.c:100-124 reads them from host memory) */
  struct trico_hip_frame_bytes* fr;
  int fr_n;
  uint8_t head8[8];
  int head8_valid;
  };

struct ra_entry
  {
  uint64_t pos;
  void* parked;         /* device buffer holding the decoded stream; NULL: it was decoded straight into the caller's array */
  uint64_t bytes;
  int ok, live;
  };

#define TRICO_MAGIC 0x6f637254u   /* "Trco", trico.c:94 */

static trico_hip_ctx* arch_ctx(struct trico_archive* a)
  {
  if (!a->ctx)
    a->ctx = trico_hip_ctx_create();
  return a->ctx;
  }

/* ---- write primitives (trico.c:26-63) ---------------------------------------------------- */

/* A host archive of hundreds of MB is written once, front to back, into memory that has never been touched: with 4 KiB pages the
 * page faults of a 450 MB buffer cost ~10 ms of a 57 ms encode.  Asking for transparent huge pages (where the kernel offers them on
 * request; elsewhere the call does nothing) leaves a few hundred faults. */
static void advise_huge(uint8_t* p, uint64_t size)
  {
#ifdef MADV_HUGEPAGE
  const uintptr_t two = (uintptr_t)2 << 20;
  if (!p || size < 16u * two)
    return;
  const uintptr_t b = ((uintptr_t)p + two - 1) & ~(two - 1), e = ((uintptr_t)p + size) & ~(two - 1);
  if (e > b)
    (void)madvise((void*)b, (size_t)(e - b), MADV_HUGEPAGE);
#else
  (void)p; (void)size;
#endif
  }

static int reserve(struct trico_archive* a, uint64_t extra)
  {
  if (!a->writable)
    return 0;
  if (a->buffer_size - a->used >= extra)
    return 1;
  uint64_t want = a->used + extra;
  uint64_t cap = a->buffer_size + a->buffer_size / 2;
  if (cap < want)
    cap = want;
  if (a->buffer_on_device)
    {
    uint8_t* nb = (uint8_t*)trico_hip_device_alloc(cap);
    if (!nb)
      return 0;
    if (a->used && !trico_hip_copy(nb, a->buffer, a->used))
      {
      trico_hip_device_free(nb);
      return 0;
      }
    trico_hip_device_free(a->buffer);
    a->buffer = nb;
    }
  else
    {
    uint8_t* nb = (uint8_t*)realloc(a->buffer, cap ? cap : 1);
    if (!nb)
      return 0;
    a->buffer = nb;
    advise_huge(nb, cap);
    }
  a->buffer_size = cap;
  return 1;
  }

/* append small host-resident fields; space must have been reserved */
static int put_host(struct trico_archive* a, const void* p, uint64_t n)
  {
  if (a->buffer_on_device)
    {
    if (!trico_hip_copy(a->buffer + a->used, p, n))
      return 0;
    }
  else
    memcpy(a->buffer + a->used, p, n);
  a->used += n;
  return 1;
  }

static void store_le32(uint8_t* p, uint32_t v)
  {
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
  }

static uint32_t load_le32(const uint8_t* p)
  {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
  }

/* ---- read primitives (trico.c:65-124) ---------------------------------------------------- */

/* components per stream type (trico.c:215-858), 0 = unknown; what the device-side frame walk needs to know */
static const uint8_t NCOMP_OF_TYPE[21] = { 0, 3, 3, 4, 8, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 1, 1, 1, 2, 4, 8 };
#define FRAME_CACHE 32

/* serves [pos, pos + n) from the cached framing bytes of a device-resident archive; 0 = not cached */
static int framing_lookup(const struct trico_archive* a, uint64_t pos, void* dst, uint64_t n)
  {
  if (a->head8_valid && pos + n <= 8)
    {
    memcpy(dst, a->head8 + pos, n);
    return 1;
    }
  for (int i = 0; i < a->fr_n; ++i)
    {
    const struct trico_hip_frame_bytes* f = &a->fr[i];
    if (pos >= f->tpos && pos + n <= f->tpos + f->nbytes_head)
      {
      memcpy(dst, f->bytes + (pos - f->tpos), n);
      return 1;
      }
    for (uint32_t c = 0; c < f->ncomp; ++c)
      if (pos >= f->size_pos[c] && f->size_valid[c] && pos + n <= f->size_pos[c] + 4)
        {
        memcpy(dst, f->bytes + 5 + 4 * c + (pos - f->size_pos[c]), n);
        return 1;
        }
    }
  return 0;
  }

/* one device-side walk over the next FRAME_CACHE streams starting at the type byte at tpos (or at the file header) */
static void framing_fill(struct trico_archive* a, uint64_t tpos)
  {
  if (!a->fr)
    a->fr = (struct trico_hip_frame_bytes*)malloc(sizeof(struct trico_hip_frame_bytes) * FRAME_CACHE);
  if (!a->fr)
    return;
  uint8_t head[8];
  const int n = trico_hip_walk_frames(a->data, a->data_size, tpos, NCOMP_OF_TYPE, a->fr, FRAME_CACHE, head);
  a->fr_n = n > 0 ? n : 0;
  if (n >= 0 && a->data_size >= 8)
    {
    memcpy(a->head8, head, 8);
    a->head8_valid = 1;
    }
  }

static int peek_bytes(struct trico_archive* a, void* dst, uint64_t n)
  {
  if (a->writable)
    return 0;
  if (a->pos + n > a->data_size)
    return 0;
  if (a->data_on_device)
    {
    if (framing_lookup(a, a->pos, dst, n))
      return 1;
    if (n == 1)
      {
      /* a type byte the cache does not hold: the walk continues from here */
      framing_fill(a, a->pos);
      if (framing_lookup(a, a->pos, dst, n))
        return 1;
      }
    return trico_hip_copy(dst, a->data + a->pos, n);      /* always right, one copy per field */
    }
  memcpy(dst, a->data + a->pos, n);
  return 1;
  }

static int read_u32(struct trico_archive* a, uint32_t* v)
  {
  uint8_t b[4];
  if (!peek_bytes(a, b, 4))
    return 0;
  a->pos += 4;
  *v = load_le32(b);
  return 1;
  }

static void read_next_stream_type(struct trico_archive* a)
  {
  uint8_t t = 0;
  if (a->pos < a->data_size && peek_bytes(a, &t, 1))
    {
    a->pos += 1;
    a->next_stream_type = t;
    }
  else
    a->next_stream_type = (uint8_t)trico_empty;
  }

/* ---- open / close / accessors ------------------------------------------------------------ */

static struct trico_archive* new_archive(void)
  {
  return (struct trico_archive*)calloc(1, sizeof(struct trico_archive));
  }

static void* open_for_writing(uint64_t initial_buffer_size, int on_device)
  {
  struct trico_archive* a = new_archive();
  if (!a)
    return NULL;
  a->writable = 1;
  a->buffer_on_device = on_device;
  if (on_device)
    a->buffer = (uint8_t*)trico_hip_device_alloc(initial_buffer_size ? initial_buffer_size : 8);
  else
    {
    a->buffer = (uint8_t*)malloc(initial_buffer_size ? initial_buffer_size : 1);
    advise_huge(a->buffer, initial_buffer_size);
    }
  if (!a->buffer)
    {
    free(a);
    return NULL;
    }
  a->buffer_size = initial_buffer_size ? initial_buffer_size : (on_device ? 8 : 1);
  uint8_t hdr[8];
  store_le32(hdr, TRICO_MAGIC);
  store_le32(hdr + 4, a->version);
  if (!reserve(a, 8) || !put_host(a, hdr, 8))
    {
    trico_close_archive(a);
    return NULL;
    }
  return a;
  }

void* trico_open_archive_for_writing(uint64_t initial_buffer_size)
  {
  return open_for_writing(initial_buffer_size, 0);
  }

void* trico_hip_open_archive_for_writing_device(uint64_t initial_buffer_size)
  {
  if (!trico_hip_available())
    return NULL;
  return open_for_writing(initial_buffer_size, 1);
  }

void* trico_open_archive_for_reading(const uint8_t* data, uint64_t data_size)
  {
  struct trico_archive* a = new_archive();
  if (!a)
    return NULL;
  a->data = data;
  a->data_size = data_size;
  a->data_on_device = (data != NULL && data_size != 0) ? trico_hip_pointer_is_device(data) : 0;
  if (a->data_on_device && data_size >= 8)
    framing_fill(a, 8);                       /* file header + framing of the first streams: one kernel, one copy */
  uint8_t hdr[8];
  if (data == NULL || !peek_bytes(a, hdr, 8) || load_le32(hdr) != TRICO_MAGIC)
    {
    free(a->fr);
    free(a);
    return NULL;
    }
  a->pos = 8;
  a->version = load_le32(hdr + 4);
  read_next_stream_type(a);
  return a;
  }

void trico_close_archive(void* archive)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a)
    return;
  if (a->buffer)
    {
    if (a->buffer_on_device)
      trico_hip_device_free(a->buffer);
    else
      free(a->buffer);
    }
  for (int i = 0; i < a->ra_n; ++i)
    if (a->ra[i].live && a->ra[i].parked)
      trico_hip_device_free(a->ra[i].parked);               /* a stream that was never collected */
  free(a->ra);
  free(a->fr);
  if (a->ctx)
    trico_hip_ctx_destroy(a->ctx);
  free(a);
  }

uint8_t* trico_get_buffer_pointer(void* archive)
  {
  return ((struct trico_archive*)archive)->buffer;
  }

uint64_t trico_get_size(void* archive)
  {
  return ((struct trico_archive*)archive)->used;
  }

uint32_t trico_get_version(void* archive)
  {
  return ((struct trico_archive*)archive)->version;
  }

enum trico_stream_type trico_get_next_stream_type(void* archive)
  {
  return (enum trico_stream_type)((struct trico_archive*)archive)->next_stream_type;
  }

/* ---- generic stream writers ---------------------------------------------------------------
 * stream := u8 type, u32 count, then per component: u32 nbytes, payload  (trico.c:215-262 etc.) */

static int append_stream(struct trico_archive* a, enum trico_stream_type st, uint32_t count_field,
                         int ncomp, const uint32_t* sizes)
  {
  uint64_t total = 5;
  for (int c = 0; c < ncomp; ++c)
    total += 4 + (uint64_t)sizes[c];
  if (!reserve(a, total))
    return 0;
  const uint64_t rollback = a->used;
  uint8_t head[5];
  head[0] = (uint8_t)st;
  store_le32(head + 1, count_field);
  if (!put_host(a, head, 5))
    {
    a->used = rollback;
    return 0;
    }
  /* size fields first, then all payloads at once (one gather launch for float streams in a device buffer) */
  void* dsts[8];
  for (int c = 0; c < ncomp; ++c)
    {
    uint8_t nb[4];
    store_le32(nb, sizes[c]);
    if (!put_host(a, nb, 4))
      {
      a->used = rollback;
      return 0;
      }
    dsts[c] = a->buffer + a->used;
    a->used += sizes[c];
    }
  if (!trico_hip_fetch_payloads(a->ctx, ncomp, dsts))
    {
    a->used = rollback;
    return 0;
    }
  return 1;
  }


/* a stream framed from units that were encoded elsewhere (other contexts, other GPUs): same bytes as append_stream */
int trico_hip_append_encoded_stream(void* archive, int stream_type, uint32_t count_field, int nunits,
                                    const void* const* payloads, const uint32_t* sizes)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a || !a->writable || nunits < 1 || nunits > 8 || stream_type < 1 || stream_type > 20 || !payloads || !sizes)
    return 0;
  uint64_t total = 5;
  for (int c = 0; c < nunits; ++c)
    {
    if (!payloads[c] && sizes[c])
      return 0;
    total += 4 + (uint64_t)sizes[c];
    }
  if (!reserve(a, total))
    return 0;
  const uint64_t rollback = a->used;
  uint8_t head[5];
  head[0] = (uint8_t)stream_type;
  store_le32(head + 1, count_field);
  int ok = put_host(a, head, 5);
  for (int c = 0; ok && c < nunits; ++c)
    {
    uint8_t nb[4];
    store_le32(nb, sizes[c]);
    ok = put_host(a, nb, 4);
    if (ok && sizes[c])
      {
      /* host archive + host payload needs no device; every other combination is a HIP copy */
      if (!a->buffer_on_device && !trico_hip_pointer_is_device(payloads[c]))
        memcpy(a->buffer + a->used, payloads[c], sizes[c]);
      else
        ok = trico_hip_copy(a->buffer + a->used, payloads[c], sizes[c]);
      a->used += sizes[c];
      }
    }
  if (!ok)
    a->used = rollback;
  return ok;
  }

static int write_fp_stream(void* archive, enum trico_stream_type st, uint32_t count_field,
                           const void* data, uint32_t n, int arity, int width)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a || !a->writable || !arch_ctx(a))
    return 0;
  uint32_t sizes[3] = { 0, 0, 0 };
  if (width == 4 && n != 0 && a->buffer_on_device)
    {
    /* A device-resident archive with room for the stream's worst case: the encoder frames the stream body in place (one queue of
     * launches, one wait).  Otherwise - and whenever the encoder says "not this way" - sizes first, then the payloads. */
    const uint64_t bound = 5 + 4ull * n + 3ull * (((uint64_t)n + 7) / 8 + 1) + 8;
    const uint64_t worst = 5 + (uint64_t)arity * (4 + bound);
    if (a->buffer_size - a->used >= worst)
      {
      const uint64_t rollback = a->used;
      uint8_t head[5];
      head[0] = (uint8_t)st;
      store_le32(head + 1, count_field);
      if (!put_host(a, head, 5))
        {
        a->used = rollback;
        return 0;
        }
      const int r = trico_hip_fpc_encode_place(a->ctx, data, n, arity, width, a->buffer + a->used, sizes);
      if (r == 1)
        {
        for (int c = 0; c < arity; ++c)
          a->used += 4 + (uint64_t)sizes[c];
        return 1;
        }
      a->used = rollback;
      if (r == 0)
        return 0;
      }
    }
  if (!trico_hip_fpc_encode(a->ctx, data, n, arity, width, sizes))
    return 0;
  return append_stream(a, st, count_field, arity, sizes);
  }

static int write_int_stream(void* archive, enum trico_stream_type st, uint32_t count_field,
                            const void* data, uint32_t n, int width)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a || !a->writable || !arch_ctx(a))
    return 0;
  uint32_t sizes[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  if (!trico_hip_int_encode(a->ctx, data, n, width, sizes))
    return 0;
  return append_stream(a, st, count_field, width, sizes);
  }

/* vec3 float / double (trico.c:215-277, 380-442) */
int trico_write_vertices(void* a, const float* v, uint32_t n)
  { return write_fp_stream(a, trico_vertex_float_stream, n, v, n, 3, 4); }
int trico_write_vertex_normals(void* a, const float* v, uint32_t n)
  { return write_fp_stream(a, trico_vertex_normal_float_stream, n, v, n, 3, 4); }
int trico_write_triangle_normals(void* a, const float* v, uint32_t n)
  { return write_fp_stream(a, trico_triangle_normal_float_stream, n, v, n, 3, 4); }
int trico_write_vertices_double(void* a, const double* v, uint32_t n)
  { return write_fp_stream(a, trico_vertex_double_stream, n, v, n, 3, 8); }
int trico_write_vertex_normals_double(void* a, const double* v, uint32_t n)
  { return write_fp_stream(a, trico_vertex_normal_double_stream, n, v, n, 3, 8); }
int trico_write_triangle_normals_double(void* a, const double* v, uint32_t n)
  { return write_fp_stream(a, trico_triangle_normal_double_stream, n, v, n, 3, 8); }

/* vec2 (trico.c:534-628).  per-triangle float stores 3*nr positions (trico.c:579); the double
 * variants carry the FLOAT tags 5/7 and the unscaled count, exactly like trico.c:620-628. */
int trico_write_uv_per_vertex(void* a, const float* uv, uint32_t n)
  { return write_fp_stream(a, trico_uv_per_vertex_float_stream, n, uv, n, 2, 4); }
int trico_write_uv_per_triangle(void* a, const float* uv, uint32_t n)
  { return write_fp_stream(a, trico_uv_per_triangle_float_stream, n * 3u, uv, n * 3u, 2, 4); }
int trico_write_uv_per_vertex_double(void* a, const double* uv, uint32_t n)
  { return write_fp_stream(a, trico_uv_per_vertex_float_stream, n, uv, n, 2, 8); }
int trico_write_uv_per_triangle_double(void* a, const double* uv, uint32_t n)
  { return write_fp_stream(a, trico_uv_per_triangle_float_stream, n, uv, n, 2, 8); }

/* scalar attributes (trico.c:279-321) */
int trico_write_attributes_float(void* a, const float* p, uint32_t n)
  { return write_fp_stream(a, trico_attribute_float_stream, n, p, n, 1, 4); }
int trico_write_attributes_double(void* a, const double* p, uint32_t n)
  { return write_fp_stream(a, trico_attribute_double_stream, n, p, n, 1, 8); }

/* index / integer streams (trico.c:323-378, 444-532, 630-858): planes hold 3*count bytes for
 * triangles, count bytes for colors/attributes. */
int trico_write_triangles(void* a, const uint32_t* t, uint32_t n)
  { return write_int_stream(a, trico_triangle_uint32_stream, n, t, n * 3u, 4); }
int trico_write_triangles_long(void* a, const uint64_t* t, uint32_t n)
  { return write_int_stream(a, trico_triangle_uint64_stream, n, t, n * 3u, 8); }
int trico_write_vertex_colors(void* a, const uint32_t* c, uint32_t n)
  { return write_int_stream(a, trico_vertex_color_stream, n, c, n, 4); }
int trico_write_triangle_colors(void* a, const uint32_t* c, uint32_t n)
  { return write_int_stream(a, trico_triangle_color_stream, n, c, n, 4); }
int trico_write_attributes_uint8(void* a, const uint8_t* p, uint32_t n)
  { return write_int_stream(a, trico_attribute_uint8_stream, n, p, n, 1); }
int trico_write_attributes_uint16(void* a, const uint16_t* p, uint32_t n)
  { return write_int_stream(a, trico_attribute_uint16_stream, n, p, n, 2); }
int trico_write_attributes_uint32(void* a, const uint32_t* p, uint32_t n)
  { return write_int_stream(a, trico_attribute_uint32_stream, n, p, n, 4); }
int trico_write_attributes_uint64(void* a, const uint64_t* p, uint32_t n)
  { return write_int_stream(a, trico_attribute_uint64_stream, n, p, n, 8); }

/* ---- count peeks (trico.c:860-941) --------------------------------------------------------- */

static uint32_t peek_count(void* archive, uint32_t type_mask)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a || a->writable || a->next_stream_type > 20 || !((type_mask >> a->next_stream_type) & 1u))
    return 0;
  uint8_t b[4];
  if (!peek_bytes(a, b, 4))
    return 0;
  return load_le32(b);
  }

#define TM(t) (1u << (t))

uint32_t trico_get_number_of_vertices(void* a)
  { return peek_count(a, TM(trico_vertex_float_stream) | TM(trico_vertex_double_stream)); }
uint32_t trico_get_number_of_triangles(void* a)
  { return peek_count(a, TM(trico_triangle_uint32_stream) | TM(trico_triangle_uint64_stream)); }
uint32_t trico_get_number_of_uvs(void* a)
  { return peek_count(a, TM(trico_uv_per_vertex_float_stream) | TM(trico_uv_per_vertex_double_stream) |
                         TM(trico_uv_per_triangle_float_stream) | TM(trico_uv_per_triangle_double_stream)); }
uint32_t trico_get_number_of_normals(void* a)
  { return peek_count(a, TM(trico_vertex_normal_float_stream) | TM(trico_vertex_normal_double_stream) |
                         TM(trico_triangle_normal_float_stream) | TM(trico_triangle_normal_double_stream)); }
uint32_t trico_get_number_of_colors(void* a)
  { return peek_count(a, TM(trico_vertex_color_stream) | TM(trico_triangle_color_stream)); }
uint32_t trico_get_number_of_attributes(void* a)
  { return peek_count(a, TM(trico_attribute_float_stream) | TM(trico_attribute_double_stream) |
                         TM(trico_attribute_uint8_stream) | TM(trico_attribute_uint16_stream) |
                         TM(trico_attribute_uint32_stream) | TM(trico_attribute_uint64_stream)); }

/* ---- generic stream readers (trico.c:943-1668) -------------------------------------------- */

/* Parses count + ncomp (nbytes, payload) frames at the cursor without consuming them. */
static int parse_frames(struct trico_archive* a, int ncomp, uint32_t* count, const uint8_t** payloads,
                        uint32_t* sizes, uint64_t* end_pos)
  {
  const uint64_t start = a->pos;
  int ok = read_u32(a, count);
  for (int c = 0; ok && c < ncomp; ++c)
    {
    ok = read_u32(a, &sizes[c]);
    if (ok && a->pos + sizes[c] > a->data_size)
      ok = 0;
    if (ok)
      {
      payloads[c] = a->data + a->pos;
      a->pos += sizes[c];
      }
    }
  *end_pos = a->pos;
  a->pos = start;
  return ok;
  }

/* ---- read-ahead ------------------------------------------------------------------------------
 * The reference decodes stream after stream (trico.c:943-1668), and the format leaves one serial chain
 * per component stream.  On the GPU a chain occupies one wave, so the first read that actually wants
 * data walks the remaining frames and decodes every stream of the archive as ONE batch
 * (trico_hip_decode_jobs: all chains in one kernel launch, the integer streams beside them).  The stream
 * that is being read goes straight into the caller's array, the others are parked on the device and the
 * later trico_read_* calls copy them out.  Results and error behaviour are those of the one-by-one path:
 * a malformed stream fails in its own read call.
 * TRICO_HIP_READAHEAD_MB bounds the decoded bytes held at once (default 65536, 0 switches it off). */

struct stream_layout { int known, is_int, ncomp, arity, width; uint32_t per_count; };

static struct stream_layout layout_of(uint8_t type)
  {
  struct stream_layout fp3f = { 1, 0, 3, 3, 4, 1 }, fp3d = { 1, 0, 3, 3, 8, 1 }, fp2f = { 1, 0, 2, 2, 4, 1 }, fp2d = { 1, 0, 2, 2, 8, 1 };
  struct stream_layout fp1f = { 1, 0, 1, 1, 4, 1 }, fp1d = { 1, 0, 1, 1, 8, 1 }, none = { 0, 0, 0, 0, 0, 0 };
  struct stream_layout tri32 = { 1, 1, 4, 1, 4, 3 }, tri64 = { 1, 1, 8, 1, 8, 3 };
  struct stream_layout i8 = { 1, 1, 1, 1, 1, 1 }, i16 = { 1, 1, 2, 1, 2, 1 }, i32 = { 1, 1, 4, 1, 4, 1 }, i64 = { 1, 1, 8, 1, 8, 1 };
  switch ((enum trico_stream_type)type)
    {
    case trico_vertex_float_stream: case trico_vertex_normal_float_stream: case trico_triangle_normal_float_stream: return fp3f;
    case trico_vertex_double_stream: case trico_vertex_normal_double_stream: case trico_triangle_normal_double_stream: return fp3d;
    case trico_uv_per_vertex_float_stream: case trico_uv_per_triangle_float_stream: return fp2f;
    case trico_uv_per_vertex_double_stream: case trico_uv_per_triangle_double_stream: return fp2d;
    case trico_attribute_float_stream: return fp1f;
    case trico_attribute_double_stream: return fp1d;
    case trico_triangle_uint32_stream: return tri32;
    case trico_triangle_uint64_stream: return tri64;
    case trico_vertex_color_stream: case trico_triangle_color_stream: case trico_attribute_uint32_stream: return i32;
    case trico_attribute_uint8_stream: return i8;
    case trico_attribute_uint16_stream: return i16;
    case trico_attribute_uint64_stream: return i64;
    default: return none;
    }
  }

static int parse_frames(struct trico_archive* a, int ncomp, uint32_t* count, const uint8_t** payloads,
                        uint32_t* sizes, uint64_t* end_pos);

/* the stream at the cursor as a decode job (dst not set); 0 if the type is unknown or the framing broken.  *end = cursor
 * position behind the stream. */
static int job_at_cursor(struct trico_archive* a, trico_hip_decode_job* job, uint32_t* count_field, uint64_t* end)
  {
  const struct stream_layout L = layout_of(a->next_stream_type);
  uint32_t count = 0;
  memset(job, 0, sizeof(*job));
  if (!L.known || !parse_frames(a, L.ncomp, &count, job->payloads, job->sizes, end))
    return 0;
  const uint64_t n = (uint64_t)count * L.per_count;
  if (n > 0xffffffffull)
    return 0;
  job->is_int = L.is_int;
  job->arity = L.arity;
  job->width = L.width;
  job->n = (uint32_t)n;
  *count_field = count;
  return 1;
  }

static uint64_t job_bytes(const trico_hip_decode_job* job)
  {
  return (uint64_t)job->n * (uint64_t)job->width * (uint64_t)(job->is_int ? 1 : job->arity);
  }

/* device bytes the streams decoded ahead may occupy: TRICO_HIP_READAHEAD_MB (default 64 GiB), but never more than half
 * of what the device has free right now. */
static uint64_t readahead_budget(void)
  {
  const char* e = getenv("TRICO_HIP_READAHEAD_MB");
  const uint64_t mb = e ? strtoull(e, NULL, 10) : 65536ull;
  uint64_t budget = mb << 20;
  const uint64_t half_free = trico_hip_device_free_bytes() / 2;
  if (budget > half_free)
    budget = half_free;
  return budget;
  }

/* device memory a stream decoded ahead is charged: its parked values plus its share of the engine's working set (staged payload,
 * byte planes, the 4 bytes per plane byte of the LZ4 decoder, the self-check's encoder workspace; these are shared by the
 * streams of a batch, so the charge is generous) */
static uint64_t readahead_cost(uint64_t decoded_bytes)
  {
  return 3 * decoded_bytes + (64ull << 20);
  }

static void release_entry(struct ra_entry* e)
  {
  if (e->live && e->parked)
    trico_hip_device_free(e->parked);
  e->parked = NULL;
  e->live = 0;
  }

/* drops the decode that was made ahead for the stream at the cursor (the caller skips it) */
static void drop_readahead(struct trico_archive* a)
  {
  for (int i = 0; i < a->ra_n; ++i)
    if (a->ra[i].live && a->ra[i].pos == a->pos)
      release_entry(&a->ra[i]);
  }

/* the one-by-one decode of a stream failed while other streams are still parked on the device: give their memory back
 * (their reads will decode them again, one by one) so that the caller can retry */
static int drop_all_readahead(struct trico_archive* a)
  {
  int dropped = 0;
  for (int i = 0; i < a->ra_n; ++i)
    if (a->ra[i].live)
      {
      release_entry(&a->ra[i]);
      ++dropped;
      }
  return dropped;
  }

/* `out` = destination of the stream at the cursor (the read call that triggered the read-ahead) */
static void start_readahead(struct trico_archive* a, void* out)
  {
  a->ra_started = 1;
  const uint64_t budget = readahead_budget();
  if (budget == 0 || !trico_hip_available())
    return;
  const uint64_t save_pos = a->pos;
  const uint8_t save_type = a->next_stream_type;
  /* pass 1: count the streams left (a lone stream gains nothing from the detour) */
  int streams = 0;
  while (a->next_stream_type != (uint8_t)trico_empty)
    {
    trico_hip_decode_job job;
    uint32_t count = 0;
    uint64_t end = 0;
    if (!job_at_cursor(a, &job, &count, &end))
      break;
    ++streams;
    a->pos = end;
    read_next_stream_type(a);
    }
  a->pos = save_pos;
  a->next_stream_type = save_type;
  if (streams < 2)
    return;
  trico_hip_decode_job* jobs = (trico_hip_decode_job*)calloc((size_t)streams, sizeof(trico_hip_decode_job));
  struct ra_entry* ra = (struct ra_entry*)calloc((size_t)streams, sizeof(struct ra_entry));
  if (!jobs || !ra)
    {
    free(jobs);
    free(ra);
    return;
    }
  /* pass 2: one job per stream, as long as the budget lasts */
  uint64_t held = 0;
  int n = 0;
  while (a->next_stream_type != (uint8_t)trico_empty && n < streams)
    {
    uint32_t count = 0;
    uint64_t end = 0;
    if (!job_at_cursor(a, &jobs[n], &count, &end))
      break;
    const uint64_t bytes = job_bytes(&jobs[n]);
    if (jobs[n].n != 0)
      {
      void* dst = NULL;
      void* parked = NULL;
      if (a->pos == save_pos)
        dst = out;                                   /* the stream being read: straight into the caller's array */
      else if (held + readahead_cost(bytes) <= budget)
        {
        parked = trico_hip_device_alloc(bytes + 16);
        dst = parked;
        held += parked ? readahead_cost(bytes) : 0;
        }
      if (dst)
        {
        jobs[n].dst = dst;
        ra[n].pos = a->pos;
        ra[n].parked = parked;
        ra[n].bytes = bytes;
        ra[n].live = 1;
        ++n;
        }
      }
    a->pos = end;
    read_next_stream_type(a);
    }
  a->pos = save_pos;
  a->next_stream_type = save_type;
  if (n >= 2)
    {
    (void)trico_hip_decode_jobs(jobs, n);            /* per-job results below: a stream that failed fails in its own read call */
    for (int i = 0; i < n; ++i)
      {
      ra[i].ok = jobs[i].ok > 0;
      if (jobs[i].ok > 0 && jobs[i].other_writer)
        a->other_writer_streams += 1;
      if (jobs[i].ok < 0)
        release_entry(&ra[i]);                       /* not attempted (no memory for the batch, a launch that failed): the stream's own
                                                        read call decodes it alone, as if there had been no read-ahead */
      }
    a->ra = ra;
    a->ra_n = n;
    }
  else
    {
    for (int i = 0; i < n; ++i)
      release_entry(&ra[i]);
    free(ra);
    }
  free(jobs);
  }

/* collects the stream at the cursor if it was decoded ahead: 1 done, 0 failed, -1 not decoded ahead */
static int collect_readahead(struct trico_archive* a, void* out)
  {
  if (!a->ra_started)
    start_readahead(a, out);
  for (int i = 0; i < a->ra_n; ++i)
    if (a->ra[i].live && a->ra[i].pos == a->pos)
      {
      int ok = a->ra[i].ok;
      if (ok && a->ra[i].parked)
        ok = a->ra[i].bytes == 0 || trico_hip_copy(out, a->ra[i].parked, a->ra[i].bytes);
      release_entry(&a->ra[i]);
      return ok;
      }
  return -1;
  }

/* lib_alloc: the library mallocs *dst (trico_read_attributes_float/double, trico.c:1377,1408) */
static int read_fp_stream(void* archive, enum trico_stream_type st, void** dst, int arity, int width, int lib_alloc)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a || a->writable || a->next_stream_type != (uint8_t)st)
    return 0;
  uint32_t count = 0, sizes[3] = { 0, 0, 0 };
  const uint8_t* payloads[3] = { NULL, NULL, NULL };
  uint64_t end_pos = 0;
  if (!parse_frames(a, arity, &count, payloads, sizes, &end_pos))
    return 0;
  if (dst != NULL)
    {
    if (!arch_ctx(a))
      return 0;
    void* out = *dst;
    if (lib_alloc)
      {
      out = malloc((size_t)count * (size_t)width + 1);
      if (!out)
        return 0;
      }
    const int ahead = collect_readahead(a, out);
    int ok = ahead > 0;
    if (ahead < 0)
      {
      ok = trico_hip_fpc_decode(a->ctx, payloads, sizes, arity, width, count, out);
      if (!ok && drop_all_readahead(a))
        ok = trico_hip_fpc_decode(a->ctx, payloads, sizes, arity, width, count, out);
      }
    if (!ok)
      {
      if (lib_alloc)
        free(out);
      return 0;
      }
    if (lib_alloc)
      *dst = out;
    }
  else if (a->ra_started)
    drop_readahead(a);
  a->pos = end_pos;
  read_next_stream_type(a);
  return 1;
  }

static int read_int_stream(void* archive, enum trico_stream_type st, void** dst, int width, uint32_t per_count)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a || a->writable || a->next_stream_type != (uint8_t)st)
    return 0;
  uint32_t count = 0, sizes[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  const uint8_t* payloads[8] = { NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL };
  uint64_t end_pos = 0;
  if (!parse_frames(a, width, &count, payloads, sizes, &end_pos))
    return 0;
  if ((uint64_t)count * per_count > 0xffffffffull)
    return 0;
  if (dst != NULL)
    {
    if (!arch_ctx(a))
      return 0;
    const int ahead = collect_readahead(a, *dst);
    int ok = ahead > 0;
    if (ahead < 0)
      {
      ok = trico_hip_int_decode(a->ctx, payloads, sizes, width, count * per_count, *dst);
      if (!ok && drop_all_readahead(a))
        ok = trico_hip_int_decode(a->ctx, payloads, sizes, width, count * per_count, *dst);
      }
    if (!ok)
      return 0;
    }
  else if (a->ra_started)
    drop_readahead(a);
  a->pos = end_pos;
  read_next_stream_type(a);
  return 1;
  }

int trico_read_vertices(void* a, float** v)
  { return read_fp_stream(a, trico_vertex_float_stream, (void**)v, 3, 4, 0); }
int trico_read_vertex_normals(void* a, float** v)
  { return read_fp_stream(a, trico_vertex_normal_float_stream, (void**)v, 3, 4, 0); }
int trico_read_triangle_normals(void* a, float** v)
  { return read_fp_stream(a, trico_triangle_normal_float_stream, (void**)v, 3, 4, 0); }
int trico_read_vertices_double(void* a, double** v)
  { return read_fp_stream(a, trico_vertex_double_stream, (void**)v, 3, 8, 0); }
int trico_read_vertex_normals_double(void* a, double** v)
  { return read_fp_stream(a, trico_vertex_normal_double_stream, (void**)v, 3, 8, 0); }
int trico_read_triangle_normals_double(void* a, double** v)
  { return read_fp_stream(a, trico_triangle_normal_double_stream, (void**)v, 3, 8, 0); }
int trico_read_uv_per_vertex(void* a, float** uv)
  { return read_fp_stream(a, trico_uv_per_vertex_float_stream, (void**)uv, 2, 4, 0); }
int trico_read_uv_per_triangle(void* a, float** uv)
  { return read_fp_stream(a, trico_uv_per_triangle_float_stream, (void**)uv, 2, 4, 0); }
int trico_read_uv_per_vertex_double(void* a, double** uv)
  { return read_fp_stream(a, trico_uv_per_vertex_double_stream, (void**)uv, 2, 8, 0); }
int trico_read_uv_per_triangle_double(void* a, double** uv)
  { return read_fp_stream(a, trico_uv_per_triangle_double_stream, (void**)uv, 2, 8, 0); }
/* Not in trico.h: the reference's shared library happens to export the helper behind the two readers above (trico.c, the sibling of its
 * static trico_read_vec2_float at :1247 was left non-static), so a program linked against it may hold the name.  Same meaning: the
 * next stream must be the double uv stream type `st`. */
TRICO_API int trico_read_vec2_double(void* a, double** uv, enum trico_stream_type st);
int trico_read_vec2_double(void* a, double** uv, enum trico_stream_type st)
  {
  if (st != trico_uv_per_vertex_double_stream && st != trico_uv_per_triangle_double_stream)
    return 0;
  return read_fp_stream(a, st, (void**)uv, 2, 8, 0);
  }
int trico_read_attributes_float(void* a, float** p)
  { return read_fp_stream(a, trico_attribute_float_stream, (void**)p, 1, 4, 1); }
int trico_read_attributes_double(void* a, double** p)
  { return read_fp_stream(a, trico_attribute_double_stream, (void**)p, 1, 8, 1); }

int trico_read_triangles(void* a, uint32_t** t)
  { return read_int_stream(a, trico_triangle_uint32_stream, (void**)t, 4, 3); }
int trico_read_triangles_long(void* a, uint64_t** t)
  { return read_int_stream(a, trico_triangle_uint64_stream, (void**)t, 8, 3); }
int trico_read_vertex_colors(void* a, uint32_t** c)
  { return read_int_stream(a, trico_vertex_color_stream, (void**)c, 4, 1); }
int trico_read_triangle_colors(void* a, uint32_t** c)
  { return read_int_stream(a, trico_triangle_color_stream, (void**)c, 4, 1); }
int trico_read_attributes_uint8(void* a, uint8_t** p)
  { return read_int_stream(a, trico_attribute_uint8_stream, (void**)p, 1, 1); }
int trico_read_attributes_uint16(void* a, uint16_t** p)
  { return read_int_stream(a, trico_attribute_uint16_stream, (void**)p, 2, 1); }
int trico_read_attributes_uint32(void* a, uint32_t** p)
  { return read_int_stream(a, trico_attribute_uint32_stream, (void**)p, 4, 1); }
int trico_read_attributes_uint64(void* a, uint64_t** p)
  { return read_int_stream(a, trico_attribute_uint64_stream, (void**)p, 8, 1); }

/* ---- whole archives at once (include/trico/trico_hip.h) ------------------------------------------- */

int trico_hip_list_streams(void* archive, trico_hip_stream_info* out, int cap)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a || a->writable || (cap > 0 && !out))
    return -1;
  const uint64_t save_pos = a->pos;
  const uint8_t save_type = a->next_stream_type;
  int n = 0, broken = 0;
  while (a->next_stream_type != (uint8_t)trico_empty)
    {
    trico_hip_decode_job job;
    uint32_t count = 0;
    uint64_t end = 0;
    if (!job_at_cursor(a, &job, &count, &end))
      {
      broken = 1;
      break;
      }
    if (n < cap)
      {
      trico_hip_stream_info* o = &out[n];
      o->type = a->next_stream_type;
      o->is_int = job.is_int;
      o->arity = job.arity;
      o->width = job.width;
      o->count = count;
      o->n = job.n;
      o->decoded_bytes = job_bytes(&job);
      o->payload_bytes = 0;
      for (int c = 0; c < (job.is_int ? job.width : job.arity); ++c)
        o->payload_bytes += job.sizes[c];
      }
    ++n;
    a->pos = end;
    read_next_stream_type(a);
    }
  a->pos = save_pos;
  a->next_stream_type = save_type;
  return broken ? -1 : n;
  }

int trico_hip_read_archives(void* const* archives, int count, void* const* const* dsts, const int* nstreams)
  {
  if (!archives || count < 0 || (count > 0 && (!dsts || !nstreams)))
    return 0;
  int total = 0;
  for (int k = 0; k < count; ++k)
    {
    const struct trico_archive* a = (const struct trico_archive*)archives[k];
    if (!a || a->writable || nstreams[k] < 0 || (nstreams[k] > 0 && !dsts[k]))
      return 0;
    total += nstreams[k];
    }
  if (total == 0)
    return 1;
  trico_hip_decode_job* jobs = (trico_hip_decode_job*)calloc((size_t)total, sizeof(trico_hip_decode_job));
  uint64_t* ends = (uint64_t*)calloc((size_t)total, sizeof(uint64_t));
  int* parsed = (int*)calloc((size_t)count, sizeof(int));
  if (!jobs || !ends || !parsed)
    {
    free(jobs); free(ends); free(parsed);
    return 0;
    }
  /* the frames of every archive, from its cursor */
  int all = 1, at = 0;
  for (int k = 0; k < count; ++k)
    {
    struct trico_archive* a = (struct trico_archive*)archives[k];
    const uint64_t save_pos = a->pos;
    const uint8_t save_type = a->next_stream_type;
    for (int s_ = 0; s_ < nstreams[k]; ++s_)
      {
      uint32_t cf = 0;
      if (a->next_stream_type == (uint8_t)trico_empty || !job_at_cursor(a, &jobs[at + s_], &cf, &ends[at + s_]))
        {
        all = 0;
        break;
        }
      jobs[at + s_].dst = dsts[k][s_];              /* NULL: skipped */
      ++parsed[k];
      a->pos = ends[at + s_];
      read_next_stream_type(a);
      }
    a->pos = save_pos;
    a->next_stream_type = save_type;
    at += nstreams[k];
    }
  /* one batch over everything that parsed (jobs of unparsed streams have dst == NULL and are skipped) */
  if (!trico_hip_decode_jobs(jobs, total))
    all = 0;
  /* cursors: behind the last stream of the successful prefix, as the one-by-one reads would leave them */
  at = 0;
  for (int k = 0; k < count; ++k)
    {
    struct trico_archive* a = (struct trico_archive*)archives[k];
    for (int s_ = 0; s_ < parsed[k]; ++s_)
      {
      if (jobs[at + s_].ok <= 0)
        {
        all = 0;
        break;
        }
      if (jobs[at + s_].other_writer)
        a->other_writer_streams += 1;
      if (a->ra_started)
        drop_readahead(a);
      a->pos = ends[at + s_];
      read_next_stream_type(a);
      }
    at += nstreams[k];
    }
  free(jobs);
  free(ends);
  free(parsed);
  return all;
  }

uint32_t trico_hip_archive_other_writer_streams(void* archive)
  {
  const struct trico_archive* a = (const struct trico_archive*)archive;
  if (!a || a->writable)
    return 0;
  /* batch decodes report per job; streams decoded alone by this handle's own context are counted there */
  return a->other_writer_streams + (a->ctx ? trico_hip_ctx_other_writer_streams(a->ctx) : 0u);
  }

/* trico.c:1670-1698 */
int trico_skip_next_stream(void* archive)
  {
  struct trico_archive* a = (struct trico_archive*)archive;
  if (!a || a->writable)
    return 0;
  switch ((enum trico_stream_type)a->next_stream_type)
    {
    case trico_empty: return 1;
    case trico_vertex_float_stream: return trico_read_vertices(a, NULL);
    case trico_vertex_double_stream: return trico_read_vertices_double(a, NULL);
    case trico_triangle_uint32_stream: return trico_read_triangles(a, NULL);
    case trico_triangle_uint64_stream: return trico_read_triangles_long(a, NULL);
    case trico_uv_per_vertex_float_stream: return trico_read_uv_per_vertex(a, NULL);
    case trico_uv_per_vertex_double_stream: return trico_read_uv_per_vertex_double(a, NULL);
    case trico_uv_per_triangle_float_stream: return trico_read_uv_per_triangle(a, NULL);
    case trico_uv_per_triangle_double_stream: return trico_read_uv_per_triangle_double(a, NULL);
    case trico_vertex_normal_float_stream: return trico_read_vertex_normals(a, NULL);
    case trico_vertex_normal_double_stream: return trico_read_vertex_normals_double(a, NULL);
    case trico_triangle_normal_float_stream: return trico_read_triangle_normals(a, NULL);
    case trico_triangle_normal_double_stream: return trico_read_triangle_normals_double(a, NULL);
    case trico_vertex_color_stream: return trico_read_vertex_colors(a, NULL);
    case trico_triangle_color_stream: return trico_read_triangle_colors(a, NULL);
    case trico_attribute_float_stream: return trico_read_attributes_float(a, NULL);
    case trico_attribute_double_stream: return trico_read_attributes_double(a, NULL);
    case trico_attribute_uint8_stream: return trico_read_attributes_uint8(a, NULL);
    case trico_attribute_uint16_stream: return trico_read_attributes_uint16(a, NULL);
    case trico_attribute_uint32_stream: return trico_read_attributes_uint32(a, NULL);
    case trico_attribute_uint64_stream: return trico_read_attributes_uint64(a, NULL);
    }
  return 0;
  }
