/* ioply.c — PLY reader / writer (see include/trico_io/ioply.h).
 *
 * The reference reads PLY through the rply library and a set of per-property callbacks (ioply.c:9-66); what
 * matters for parity is which numbers end up in the output arrays, so this file parses the whole file in
 * memory with its own tokenizer and applies the same extraction rules:
 *   - every value is first taken as a double (rply.c:1418-1537: strtol / strtod for ascii with range
 *     checks per declared type, raw typed bytes for binary), then narrowed with a C cast to the output type;
 *   - vertex positions and normals are written with stride 3, colour channels with stride 4 into a uint32
 *     per vertex that starts as 0xffffffff (ioply.c:171-182);
 *   - of a face's index list only entries 0, 1, 2 are kept (ioply.c:35-40); of a texcoord list entries 0..5,
 *     and a list shorter than 6 is padded with zeros when its last entry has been seen (ioply.c:50-63);
 *   - property lookup order for colours: red/green/blue/alpha, then r/g/b/a, then diffuse_*, per channel;
 *     faces: vertex_indices, then vertex_index (ioply.c:146-195).
 * Errors (unknown type, short body, ascii token that is not a number of its type) return 0 with nothing
 * allocated (the reference leaks its arrays on a failed ply_read, ioply.c:241-242). */
#include "trico_io/ioply.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_NONE };
enum { M_ASCII, M_LE, M_BE };

static int type_of(const char* w)
  {
  static const char* const names[2][8] = {
    { "int8", "uint8", "int16", "uint16", "int32", "uint32", "float32", "float64" },
    { "char", "uchar", "short", "ushort", "int", "uint", "float", "double" } };
  for (int s = 0; s < 2; ++s)
    for (int t = 0; t < 8; ++t)
      if (strcmp(w, names[s][t]) == 0)
        return t;
  return T_NONE;
  }

static const int type_size[8] = { 1, 1, 2, 2, 4, 4, 4, 8 };

/* what a property feeds */
enum { ROLE_NONE, ROLE_POS, ROLE_NRM, ROLE_COL, ROLE_FACE, ROLE_UV };

typedef struct
  {
  char name[64];
  int is_list, type, len_type;
  int role, slot;                 /* slot: component (0..2) or colour channel (0..3) */
  } prop_t;

typedef struct
  {
  char name[64];
  long count;
  prop_t* props;
  int nprops;
  } elem_t;

typedef struct
  {
  const unsigned char* p;
  const unsigned char* end;
  int mode;
  } cursor_t;

static int is_space(unsigned char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }

/* next whitespace-delimited word of the header / an ascii body; 0 at the end of the data */
static int next_word(cursor_t* c, char* out, size_t cap)
  {
  while (c->p < c->end && is_space(*c->p))
    ++c->p;
  if (c->p >= c->end)
    return 0;
  size_t n = 0;
  while (c->p < c->end && !is_space(*c->p))
    {
    if (n + 1 >= cap)
      return 0;
    out[n++] = (char)*c->p++;
    }
  out[n] = 0;
  return 1;
  }

static void skip_line(cursor_t* c)
  {
  while (c->p < c->end && *c->p != '\n')
    ++c->p;
  }

static int read_value(cursor_t* c, int type, double* v)
  {
  if (c->mode == M_ASCII)
    {
    char w[256], *e;
    if (!next_word(c, w, sizeof(w)))
      return 0;
    if (type == T_F32 || type == T_F64)
      {
      *v = strtod(w, &e);
      const double lim = type == T_F32 ? FLT_MAX : DBL_MAX;
      return !(*e || *v < -lim || *v > lim);
      }
    static const double lo[6] = { -128.0, 0.0, -32768.0, 0.0, -2147483648.0, 0.0 };
    static const double hi[6] = { 127.0, 255.0, 32767.0, 65535.0, 2147483647.0, 4294967295.0 };
    *v = (double)strtol(w, &e, 10);
    return !(*e || *v > hi[type] || *v < lo[type]);
    }
  const int sz = type_size[type];
  if ((size_t)(c->end - c->p) < (size_t)sz)
    return 0;
  unsigned char b[8];
  for (int i = 0; i < sz; ++i)
    b[i] = c->mode == M_LE ? c->p[i] : c->p[sz - 1 - i];      /* to little endian (this host) */
  c->p += sz;
  switch (type)
    {
    case T_I8: *v = (double)(int8_t)b[0]; break;
    case T_U8: *v = (double)b[0]; break;
    case T_I16: { int16_t x; memcpy(&x, b, 2); *v = x; break; }
    case T_U16: { uint16_t x; memcpy(&x, b, 2); *v = x; break; }
    case T_I32: { int32_t x; memcpy(&x, b, 4); *v = x; break; }
    case T_U32: { uint32_t x; memcpy(&x, b, 4); *v = x; break; }
    case T_F32: { float x; memcpy(&x, b, 4); *v = x; break; }
    default: { double x; memcpy(&x, b, 8); *v = x; break; }
    }
  return 1;
  }

static void free_elems(elem_t* e, int n)
  {
  for (int i = 0; i < n; ++i)
    free(e[i].props);
  free(e);
  }

static prop_t* find_prop(elem_t* e, const char* name)
  {
  if (!e)
    return NULL;
  for (int i = 0; i < e->nprops; ++i)
    if (strcmp(e->props[i].name, name) == 0)
      return &e->props[i];
  return NULL;
  }

static elem_t* find_elem(elem_t* e, int n, const char* name)
  {
  for (int i = 0; i < n; ++i)
    if (strcmp(e[i].name, name) == 0)
      return &e[i];
  return NULL;
  }

/* binds a property to a role if it exists and is still free; returns the element's instance count (like
 * ply_set_read_cb) or 0 */
static long bind(elem_t* e, const char* name, int role, int slot)
  {
  prop_t* p = find_prop(e, name);
  if (!p)
    return 0;
  /* a scalar role (position, normal, colour) bound to a list property would never be stored: treat it as absent; the
   * face / texcoord roles are the list-typed ones */
  if (p->is_list != (role == ROLE_FACE || role == ROLE_UV))
    return 0;
  p->role = role;
  p->slot = slot;
  return e->count;
  }

int trico_read_ply(uint32_t* nr_of_vertices, float** vertices, float** vertex_normals, uint32_t** vertex_colors,
                   uint32_t* nr_of_triangles, uint32_t** triangles, float** texcoords, const char* filename)
  {
  *nr_of_vertices = 0; *vertices = NULL; *vertex_normals = NULL; *vertex_colors = NULL;
  *nr_of_triangles = 0; *triangles = NULL; *texcoords = NULL;
  FILE* f = fopen(filename, "rb");
  if (!f)
    return 0;
  if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return 0; }
  const long fsz = ftell(f);
  if (fsz < 4 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return 0; }
  unsigned char* data = (unsigned char*)malloc((size_t)fsz + 1);
  if (!data || fread(data, 1, (size_t)fsz, f) != (size_t)fsz)
    {
    free(data);
    fclose(f);
    return 0;
    }
  fclose(f);
  data[fsz] = 0;

  /* ---- header ---- */
  cursor_t c = { data, data + fsz, M_ASCII };
  elem_t* elems = NULL;
  int nelems = 0, ok = 1;
  const int crlf = fsz >= 5 && data[3] == '\r' && data[4] == '\n';
  char w[256];
  if (memcmp(data, "ply", 3) != 0 || !(data[3] == '\n' || crlf))
    ok = 0;
  c.p = data + 3;
  ok = ok && next_word(&c, w, sizeof(w)) && strcmp(w, "format") == 0 && next_word(&c, w, sizeof(w));
  if (ok)
    {
    if (strcmp(w, "ascii") == 0) c.mode = M_ASCII;
    else if (strcmp(w, "binary_little_endian") == 0) c.mode = M_LE;
    else if (strcmp(w, "binary_big_endian") == 0) c.mode = M_BE;
    else ok = 0;
    }
  ok = ok && next_word(&c, w, sizeof(w)) && strcmp(w, "1.0") == 0;
  const int body_mode = c.mode;
  c.mode = M_ASCII;
  while (ok)
    {
    if (!next_word(&c, w, sizeof(w))) { ok = 0; break; }
    if (strcmp(w, "end_header") == 0)
      break;
    if (strcmp(w, "comment") == 0 || strcmp(w, "obj_info") == 0)
      {
      skip_line(&c);
      continue;
      }
    if (strcmp(w, "element") == 0)
      {
      elem_t* ne = (elem_t*)realloc(elems, (size_t)(nelems + 1) * sizeof(elem_t));
      if (!ne) { ok = 0; break; }
      elems = ne;
      elem_t* e = &elems[nelems];
      memset(e, 0, sizeof(*e));
      char cnt[64], *endp;
      if (!next_word(&c, e->name, sizeof(e->name)) || !next_word(&c, cnt, sizeof(cnt))) { ok = 0; break; }
      e->count = strtol(cnt, &endp, 10);
      if (*endp || e->count < 0 || (unsigned long)e->count > 0xffffffffUL) { ok = 0; break; }   /* the API reports counts as uint32_t */
      ++nelems;
      continue;
      }
    if (strcmp(w, "property") == 0 && nelems > 0)
      {
      elem_t* e = &elems[nelems - 1];
      prop_t* np = (prop_t*)realloc(e->props, (size_t)(e->nprops + 1) * sizeof(prop_t));
      if (!np) { ok = 0; break; }
      e->props = np;
      prop_t* p = &e->props[e->nprops];
      memset(p, 0, sizeof(*p));
      if (!next_word(&c, w, sizeof(w))) { ok = 0; break; }
      if (strcmp(w, "list") == 0)
        {
        p->is_list = 1;
        if (!next_word(&c, w, sizeof(w)) || (p->len_type = type_of(w)) == T_NONE ||
            !next_word(&c, w, sizeof(w)) || (p->type = type_of(w)) == T_NONE) { ok = 0; break; }
        }
      else if ((p->type = type_of(w)) == T_NONE) { ok = 0; break; }
      if (!next_word(&c, p->name, sizeof(p->name))) { ok = 0; break; }
      ++e->nprops;
      continue;
      }
    ok = 0;
    }
  if (ok)
    {
    /* the body starts after the line end of "end_header" (one extra byte in \r\n files, rply.c:418-424) */
    if (crlf)
      {
      if (c.p + 2 <= c.end) c.p += 2; else ok = 0;
      }
    else if (c.p < c.end)
      c.p += 1;
    }
  /* element counts the rest of the file cannot hold are an error before anything is allocated for them */
  if (ok)
    {
    double need = 0.0;
    for (int ei = 0; ei < nelems; ++ei)
      {
      double per = 0.0;
      for (int pi = 0; pi < elems[ei].nprops; ++pi)
        {
        const prop_t* pr = &elems[ei].props[pi];
        per += body_mode == M_ASCII ? 2.0 : (double)type_size[pr->is_list ? pr->len_type : pr->type];
        }
      need += per * (double)elems[ei].count;
      }
    if (need > (double)(c.end - c.p) + 1.0)
      ok = 0;
    }
  if (!ok)
    {
    free_elems(elems, nelems);
    free(data);
    return 0;
    }
  c.mode = body_mode;

  /* ---- what to extract (ioply.c:92-233) ---- */
  elem_t* ev = find_elem(elems, nelems, "vertex");
  elem_t* ef = find_elem(elems, nelems, "face");
  const long nx = bind(ev, "x", ROLE_POS, 0), ny = bind(ev, "y", ROLE_POS, 1), nz = bind(ev, "z", ROLE_POS, 2);
  long nnx = 0, nny = 0, nnz = 0, ncol[4] = { 0, 0, 0, 0 }, ntri = 0, nuv = 0;
  ok = (nx == ny && nx == nz);
  uint32_t nv = ok ? (uint32_t)nx : 0;
  if (ok)
    {
    nnx = bind(ev, "nx", ROLE_NRM, 0); nny = bind(ev, "ny", ROLE_NRM, 1); nnz = bind(ev, "nz", ROLE_NRM, 2);
    ok = (nnx == nny && nnx == nnz && (!nnx || (uint32_t)nnx == nv));
    }
  if (ok)
    {
    static const char* const cn[3][4] = { { "red", "green", "blue", "alpha" }, { "r", "g", "b", "a" },
                                          { "diffuse_red", "diffuse_green", "diffuse_blue", "diffuse_alpha" } };
    for (int s = 0; s < 3; ++s)
      for (int ch = 0; ch < 4; ++ch)
        if (ncol[ch] == 0)
          ncol[ch] = bind(ev, cn[s][ch], ROLE_COL, ch);
    for (int ch = 0; ch < 4; ++ch)
      if (ncol[ch] && (uint32_t)ncol[ch] != nv)
        ok = 0;
    }
  if (ok)
    {
    ntri = bind(ef, "vertex_indices", ROLE_FACE, 0);
    if (ntri == 0)
      ntri = bind(ef, "vertex_index", ROLE_FACE, 0);
    nuv = bind(ef, "texcoord", ROLE_UV, 0);
    ok = !(nuv && (uint32_t)nuv != (uint32_t)ntri);
    }
  float* pos = NULL; float* nrm = NULL; uint32_t* col = NULL; uint32_t* tri = NULL; float* uv = NULL;
  const int has_col = ncol[0] > 0 || ncol[1] > 0 || ncol[2] > 0 || ncol[3] > 0;
  if (ok)
    {
    if (nv) pos = (float*)malloc((size_t)nv * 3 * sizeof(float));
    if (nnx > 0) nrm = (float*)malloc((size_t)nv * 3 * sizeof(float));
    if (has_col) col = (uint32_t*)malloc((size_t)nv * sizeof(uint32_t) + 1);
    if (ntri > 0) tri = (uint32_t*)malloc((size_t)ntri * 3 * sizeof(uint32_t));
    if (nuv > 0) uv = (float*)malloc((size_t)ntri * 6 * sizeof(float));
    ok = (!nv || pos) && (!(nnx > 0) || nrm) && (!has_col || col) && (!(ntri > 0) || tri) && (!(nuv > 0) || uv);
    if (ok && col)
      memset(col, 0xff, (size_t)nv * sizeof(uint32_t));
    }

  /* ---- body ---- */
  uint32_t* tri_w = tri;
  float* uv_w = uv;
  for (int ei = 0; ok && ei < nelems; ++ei)
    {
    elem_t* e = &elems[ei];
    for (long inst = 0; ok && inst < e->count; ++inst)
      for (int pi = 0; ok && pi < e->nprops; ++pi)
        {
        const prop_t* p = &e->props[pi];
        double val;
        if (!p->is_list)
          {
          if (!read_value(&c, p->type, &val)) { ok = 0; break; }
          switch (p->role)
            {
            case ROLE_POS: pos[3 * (size_t)inst + p->slot] = (float)val; break;
            case ROLE_NRM: nrm[3 * (size_t)inst + p->slot] = (float)val; break;
            case ROLE_COL: ((uint8_t*)col)[4 * (size_t)inst + p->slot] = (uint8_t)val; break;
            /* a scalar bound as face / texcoord behaves like a list of length 1 (value_index 0) */
            case ROLE_FACE: *tri_w++ = (uint32_t)val; break;
            case ROLE_UV: *uv_w++ = (float)val; for (int j = 1; j < 6; ++j) *uv_w++ = 0.f; break;
            default: break;
            }
          continue;
          }
        double dlen;
        if (!read_value(&c, p->len_type, &dlen)) { ok = 0; break; }
        const long len = (long)dlen;
        if (p->role == ROLE_UV && len == 0)
          for (int j = 0; j < 6; ++j)
            *uv_w++ = 0.f;                                       /* ioply.c:56-62 on the length callback */
        for (long k = 0; k < len; ++k)
          {
          if (!read_value(&c, p->type, &val)) { ok = 0; break; }
          if (p->role == ROLE_FACE)
            {
            if (k < 3)
              {
              if (tri_w >= tri + 3 * (size_t)ntri) { ok = 0; break; }
              *tri_w++ = (uint32_t)val;
              }
            }
          else if (p->role == ROLE_UV)
            {
            if (k < 6)
              {
              if (uv_w >= uv + 6 * (size_t)ntri) { ok = 0; break; }
              *uv_w++ = (float)val;
              }
            if (k == len - 1 && len != 6)
              for (long j = len; j < 6; ++j)
                {
                if (uv_w >= uv + 6 * (size_t)ntri) { ok = 0; break; }
                *uv_w++ = 0.f;
                }
            }
          }
        }
    }
  free_elems(elems, nelems);
  free(data);
  if (!ok)
    {
    free(pos); free(nrm); free(col); free(tri); free(uv);
    return 0;
    }
  *nr_of_vertices = nv;
  *vertices = pos;
  *vertex_normals = nrm;
  *vertex_colors = col;
  *nr_of_triangles = (uint32_t)ntri;
  *triangles = tri;
  *texcoords = uv;
  return 1;
  }

/* ioply.c:248-313: always binary, host byte order, float positions / normals, uchar rgba, int index lists */
int trico_write_ply(const uint32_t nr_of_vertices, const float* vertices, const float* vertex_normals,
                    const uint32_t* vertex_colors, const uint32_t nr_of_triangles, const uint32_t* triangles,
                    const float* texcoords, const char* filename)
  {
  if (!vertices || nr_of_vertices == 0)
    return 0;
  FILE* f = fopen(filename, "wb");
  if (!f)
    return 0;
  const uint16_t probe = 1;
  const int little = *(const unsigned char*)&probe == 1;
  fprintf(f, "ply\nformat %s 1.0\n", little ? "binary_little_endian" : "binary_big_endian");
  fprintf(f, "element vertex %d\nproperty float x\nproperty float y\nproperty float z\n", nr_of_vertices);
  if (vertex_normals)
    fprintf(f, "property float nx\nproperty float ny\nproperty float nz\n");
  if (vertex_colors)
    fprintf(f, "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\n");
  const int faces = nr_of_triangles && triangles;
  if (faces)
    {
    fprintf(f, "element face %d\nproperty list uchar int vertex_indices\n", nr_of_triangles);
    if (texcoords)
      fprintf(f, "property list uchar float texcoord\n");
    }
  fprintf(f, "end_header\n");
  /* rows are assembled in a buffer: one fwrite per few thousand rows instead of 2-4 per row */
  const size_t vrow = 12 + (vertex_normals ? 12 : 0) + (vertex_colors ? 4 : 0);
  const size_t frow = 13 + (texcoords ? 25 : 0);
  enum { BATCH = 4096 };
  unsigned char* buf = (unsigned char*)malloc(BATCH * (vrow > frow ? vrow : frow));
  int ok = buf != NULL;
  for (uint32_t i0 = 0; ok && i0 < nr_of_vertices; i0 += BATCH)
    {
    const uint32_t m = nr_of_vertices - i0 < BATCH ? nr_of_vertices - i0 : BATCH;
    unsigned char* q = buf;
    for (uint32_t k = 0; k < m; ++k)
      {
      const size_t i = (size_t)i0 + k;
      memcpy(q, vertices + 3 * i, 12); q += 12;
      if (vertex_normals) { memcpy(q, vertex_normals + 3 * i, 12); q += 12; }
      if (vertex_colors) { memcpy(q, vertex_colors + i, 4); q += 4; }
      }
    ok = fwrite(buf, vrow, m, f) == m;
    }
  /* the face rows are written for every triangle the caller announces (ioply.c:298-310 dereferences
   * `triangles` regardless); without an index array there is nothing meaningful to write */
  for (uint32_t i0 = 0; ok && faces && i0 < nr_of_triangles; i0 += BATCH)
    {
    const uint32_t m = nr_of_triangles - i0 < BATCH ? nr_of_triangles - i0 : BATCH;
    unsigned char* q = buf;
    for (uint32_t k = 0; k < m; ++k)
      {
      const size_t i = (size_t)i0 + k;
      *q++ = 3;
      memcpy(q, triangles + 3 * i, 12); q += 12;
      if (texcoords) { *q++ = 6; memcpy(q, texcoords + 6 * i, 24); q += 24; }
      }
    ok = fwrite(buf, frow, m, f) == m;
    }
  free(buf);
  if (fclose(f) != 0)
    ok = 0;
  return ok ? 1 : 0;
  }
