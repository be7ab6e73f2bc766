/* iostl.c — binary STL reader / writer (see include/trico_io/iostl.h).
 *
 * File layout: 80-byte header, u32 triangle count, then 50 bytes per triangle: normal (3 floats), three
 * corners (9 floats), u16 attribute.  Files whose header starts with "solid" are taken for ASCII STL and
 * rejected, like iostl.c:159-163.
 *
 * Vertex welding.  The result is defined by the reference's procedure (iostl.c:36-134): corner records
 * (x, y, z, corner id) are ordered by a quicksort that partitions around the LAST record of a range with a
 * strict lexicographic "less" on (x, y, z), then runs of records whose positions compare equal become one
 * vertex, represented by the first record of the run.  Which record is first in a run of equal positions
 * depends on the exact sequence of swaps, and it matters when +0.0 and -0.0 meet, so the same partition
 * scheme is used here; only the bookkeeping differs (whole-file read, explicit stack instead of recursion
 * on the larger side, records as structs). */
#include "trico_io/iostl.h"
#include "trico/trico_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; uint32_t corner; } corner_rec;

static int pos_less(const corner_rec* a, const corner_rec* b)
  {
  if (a->x != b->x)
    return a->x < b->x;
  if (a->y != b->y)
    return a->y < b->y;
  return a->z < b->z;
  }

static int pos_equal(const corner_rec* a, const corner_rec* b)
  {
  return a->x == b->x && a->y == b->y && a->z == b->z;
  }

/* partition [lo, hi] around the record at hi; returns the pivot's final place (iostl.c:36-57) */
static int64_t partition_last(corner_rec* r, int64_t lo, int64_t hi)
  {
  const corner_rec pivot = r[hi];
  int64_t store = lo;
  for (int64_t j = lo; j < hi; ++j)
    if (pos_less(&r[j], &pivot))
      {
      const corner_rec t = r[store];
      r[store] = r[j];
      r[j] = t;
      ++store;
      }
  const corner_rec t = r[store];
  r[store] = r[hi];
  r[hi] = t;
  return store;
  }

/* Same partitions as the reference's recursion (left part first, then right part): the two parts of a range
 * are disjoint, so the order in which they are processed cannot change the outcome; the larger part is
 * deferred on an explicit stack to bound its depth. */
static int sort_corners(corner_rec* r, int64_t n)
  {
  if (n < 2)
    return 1;
  int64_t cap = 64, top = 0;
  int64_t* stack = (int64_t*)malloc((size_t)cap * 2 * sizeof(int64_t));
  if (!stack)
    return 0;
  stack[0] = 0; stack[1] = n - 1; top = 1;
  while (top > 0)
    {
    --top;
    int64_t lo = stack[2 * top], hi = stack[2 * top + 1];
    while (lo < hi)
      {
      const int64_t p = partition_last(r, lo, hi);
      const int64_t l0 = lo, l1 = p - 1, r0 = p + 1, r1 = hi;
      const int left_small = (l1 - l0) < (r1 - r0);
      const int64_t d0 = left_small ? r0 : l0, d1 = left_small ? r1 : l1;        /* deferred (larger) part */
      if (d0 < d1)
        {
        if (top == cap)
          {
          cap *= 2;
          int64_t* ns = (int64_t*)realloc(stack, (size_t)cap * 2 * sizeof(int64_t));
          if (!ns)
            {
            free(stack);
            return 0;
            }
          stack = ns;
          }
        stack[2 * top] = d0; stack[2 * top + 1] = d1; ++top;
        }
      if (left_small) { lo = l0; hi = l1; }
      else { lo = r0; hi = r1; }
      }
    }
  free(stack);
  return 1;
  }

/* corners: 3 * ntri positions (xyz); on return vertices holds the unique positions and triangles the ranks */
static int weld(uint32_t ntri, float* vertices, uint32_t* triangles, uint32_t* nr_of_vertices)
  {
  *nr_of_vertices = 0;
  if (ntri == 0)
    return 1;
  const int64_t n = 3 * (int64_t)ntri;
  corner_rec* rec = (corner_rec*)malloc((size_t)n * sizeof(corner_rec));
  if (!rec)
    return 0;
  for (int64_t c = 0; c < n; ++c)
    {
    const uint32_t v = triangles[c];
    rec[c].x = vertices[3 * (size_t)v];
    rec[c].y = vertices[3 * (size_t)v + 1];
    rec[c].z = vertices[3 * (size_t)v + 2];
    rec[c].corner = (uint32_t)c;
    }
  if (!sort_corners(rec, n))
    {
    free(rec);
    return 0;
    }
  uint32_t nv = 0;
  int64_t run = 0;                                   /* first record of the current run of equal positions */
  for (int64_t c = 0; c <= n; ++c)
    if (c == n || !pos_equal(&rec[run], &rec[c]))
      {
      vertices[3 * (size_t)nv] = rec[run].x;
      vertices[3 * (size_t)nv + 1] = rec[run].y;
      vertices[3 * (size_t)nv + 2] = rec[run].z;
      for (int64_t k = run; k < c; ++k)
        triangles[rec[k].corner] = nv;
      ++nv;
      run = c;
      }
  free(rec);
  *nr_of_vertices = nv;
  return 1;
  }

/* Large files: sort + unique on the MI355X (k_weld.hip).  Declines (0) when there is no device, the mesh is small, or
 * the positions contain -0.0 / NaN, where the outcome depends on the tie order of the reference's quicksort that weld()
 * above reproduces.  corners: 9 floats per triangle, rewritten in place like weld(). */
static int weld_on_device(uint32_t ntri, float* vertices, uint32_t* triangles, uint32_t* nr_of_vertices)
  {
  const char* e = getenv("TRICO_IO_WELD_GPU_MIN");
  const unsigned long min_tri = e ? strtoul(e, NULL, 10) : 50000ul;
  if (ntri < min_tri || !trico_hip_available())
    return 0;
  trico_hip_ctx* ctx = trico_hip_ctx_create();
  if (!ctx)
    return 0;
  float* uniq = (float*)malloc((size_t)ntri * 9 * sizeof(float) + 1);
  int done = 0;
  if (uniq && trico_hip_weld_vertices(ctx, vertices, ntri, uniq, triangles, nr_of_vertices) == 1)
    {
    memcpy(vertices, uniq, (size_t)*nr_of_vertices * 3 * sizeof(float));
    done = 1;
    }
  free(uniq);
  trico_hip_ctx_destroy(ctx);
  return done;
  }

static int read_impl(uint32_t* nr_of_vertices, float** vertices, uint32_t* nr_of_triangles, uint32_t** triangles,
                     float** normals, uint16_t** attributes, const char* filename)
  {
  *vertices = NULL;
  *triangles = NULL;
  *nr_of_vertices = 0;
  *nr_of_triangles = 0;
  if (normals) *normals = NULL;
  if (attributes) *attributes = NULL;
  FILE* f = fopen(filename, "rb");
  if (!f)
    return 0;
  unsigned char head[84];
  if (fread(head, 1, 84, f) != 84 || memcmp(head, "solid", 5) == 0)
    {
    fclose(f);
    return 0;
    }
  uint32_t ntri;
  memcpy(&ntri, head + 80, 4);
  const size_t body = (size_t)ntri * 50;
  /* a count the file cannot hold is an error before anything is allocated for it */
  long here = ftell(f), end = -1;
  if (here >= 0 && fseek(f, 0, SEEK_END) == 0)
    end = ftell(f);
  if (here < 0 || end < 0 || (size_t)(end - here) < body || fseek(f, here, SEEK_SET) != 0)
    {
    fclose(f);
    return 0;
    }
  unsigned char* raw = (unsigned char*)malloc(body ? body : 1);
  float* v = (float*)malloc((size_t)ntri * 9 * sizeof(float) + 1);
  uint32_t* t = (uint32_t*)malloc((size_t)ntri * 3 * sizeof(uint32_t) + 1);
  float* nr = normals ? (float*)malloc((size_t)ntri * 3 * sizeof(float) + 1) : NULL;
  uint16_t* at = attributes ? (uint16_t*)malloc((size_t)ntri * sizeof(uint16_t) + 1) : NULL;
  int ok = raw && v && t && (!normals || nr) && (!attributes || at);
  if (ok)
    ok = fread(raw, 1, body, f) == body;           /* a short file is an error (iostl.c:198-199) */
  fclose(f);
  if (ok)
    {
    for (size_t i = 0; i < ntri; ++i)
      {
      const unsigned char* rp = raw + 50 * i;
      if (nr) memcpy(nr + 3 * i, rp, 12);
      memcpy(v + 9 * i, rp + 12, 36);
      if (at) memcpy(at + i, rp + 48, 2);
      t[3 * i] = (uint32_t)(3 * i); t[3 * i + 1] = (uint32_t)(3 * i + 1); t[3 * i + 2] = (uint32_t)(3 * i + 2);
      }
    uint32_t nv = 3 * ntri;
    ok = weld_on_device(ntri, v, t, &nv) || weld(ntri, v, t, &nv);
    if (ok)
      {
      float* shrunk = (float*)realloc(v, (size_t)nv * 3 * sizeof(float) + 1);
      if (shrunk) v = shrunk;
      *nr_of_vertices = nv;
      }
    }
  free(raw);
  if (!ok)
    {
    free(v); free(t); free(nr); free(at);
    return 0;
    }
  *vertices = v;
  *triangles = t;
  *nr_of_triangles = ntri;
  if (normals) *normals = nr;
  if (attributes) *attributes = at;
  return 1;
  }

int trico_read_stl(uint32_t* nr_of_vertices, float** vertices, uint32_t* nr_of_triangles, uint32_t** triangles, const char* filename)
  {
  return read_impl(nr_of_vertices, vertices, nr_of_triangles, triangles, NULL, NULL, filename);
  }

int trico_read_stl_full(uint32_t* nr_of_vertices, float** vertices, uint32_t* nr_of_triangles, uint32_t** triangles,
                        float** normals, uint16_t** attributes, const char* filename)
  {
  return read_impl(nr_of_vertices, vertices, nr_of_triangles, triangles, normals, attributes, filename);
  }

/* iostl.c:262-320: fixed 80-byte header text, zero normal / attribute when the arrays are absent */
int trico_write_stl(const float* vertices, const uint32_t* triangles, const uint32_t nr_of_triangles,
                    const float* triangle_normals, const uint16_t* attributes, const char* filename)
  {
  FILE* f = fopen(filename, "wb");
  if (!f)
    return 0;
  char head[80];
  memset(head, 0, sizeof(head));
  memcpy(head, "STL Binary File Format written by Trico library for lossless mesh compression  ", 79);
  int ok = fwrite(head, 1, 80, f) == 80 && fwrite(&nr_of_triangles, 4, 1, f) == 1;
  enum { BATCH = 4096 };
  unsigned char* buf = (unsigned char*)malloc(50 * BATCH);
  ok = ok && buf;
  for (uint32_t t0 = 0; ok && t0 < nr_of_triangles; t0 += BATCH)
    {
    const uint32_t m = nr_of_triangles - t0 < BATCH ? nr_of_triangles - t0 : BATCH;
    for (uint32_t k = 0; k < m; ++k)
      {
      unsigned char* rp = buf + 50 * (size_t)k;
      const size_t t = (size_t)t0 + k;
      if (triangle_normals)
        memcpy(rp, triangle_normals + 3 * t, 12);
      else
        memset(rp, 0, 12);
      for (int c = 0; c < 3; ++c)
        memcpy(rp + 12 + 12 * c, vertices + 3 * (size_t)triangles[3 * t + c], 12);
      if (attributes)
        memcpy(rp + 48, attributes + t, 2);
      else
        memset(rp + 48, 0, 2);
      }
    ok = fwrite(buf, 50, m, f) == m;
    }
  free(buf);
  if (fclose(f) != 0)
    ok = 0;
  return ok ? 1 : 0;
  }
