/* trico_decoder — command line front end: .trc -> binary STL / PLY  (§8(f), tools/trico_decoder/main.c).
 *
 * Same command line, messages, exit codes and output files as the reference tool, on libtrico.so and
 * libtrico_io.so.  Streams the tool knows (float vertices, uint32 triangles, float triangle / vertex normals,
 * vertex colours, per-triangle uv, uint16 attributes) are read, others skipped (main.c:255-411).  Output type:
 * the extension of -o if it is stl or ply, else PLY when colours, uv or vertex normals were found, else
 * STL; an STL without stored triangle normals gets flat normals computed in single precision
 * (main.c:436-468) — compiled without FMA contraction so the bytes match the reference build.
 *
 * One divergence: the reference hands its uv array (2 floats per stored uv) to the PLY writer, which reads 6
 * floats per face (main.c:473 / ioply.c:307-309), i.e. past the end of the allocation.  Here the array is
 * padded with zeros to 6 floats per face before it is written. */
#include "trico/trico.h"
#include "trico_io/iostl.h"
#include "trico_io/ioply.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

static int last_dot(const char* s)
  {
  for (int i = (int)strlen(s) - 1; i > 0; --i)
    if (s[i] == '.')
      return i;
  return -1;
  }

static int has_extension(const char* s, const char* ext3)
  {
  const int d = last_dot(s);
  return d >= 0 && strlen(s + d) == 4 && strcasecmp(s + d + 1, ext3) == 0;
  }

static void with_extension(char* out, size_t cap, const char* s, const char* ext3)
  {
  const int d = last_dot(s);
  const size_t stem = d >= 0 ? (size_t)d : strlen(s);
  snprintf(out, cap, "%.*s.%s", (int)stem, s, ext3);
  }

static void usage(void)
  {
  printf("Usage: trico_decoder -i <input> [options]\n\n");
  printf("Options:\n");
  printf("  -i <input>           input file name.\n");
  printf("  -o <output>          output file name of type stl or ply.\n");
  printf("\n");
  }

/* flat normals, operation by operation as main.c:441-466 */
static float* flat_normals(const float* v, const uint32_t* t, uint32_t nt)
  {
  float* n = (float*)malloc((size_t)nt * 3 * sizeof(float) + 1);
  if (!n)
    return NULL;
  for (uint32_t i = 0; i < nt; ++i)
    {
    const float* p0 = v + 3 * (size_t)t[3 * (size_t)i];
    const float* p1 = v + 3 * (size_t)t[3 * (size_t)i + 1];
    const float* p2 = v + 3 * (size_t)t[3 * (size_t)i + 2];
    const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
    const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
    const float nx = ay * bz - az * by;
    const float ny = az * bx - ax * bz;
    const float nz = ax * by - ay * bx;
    const float len = (float)sqrt((double)(nx * nx + ny * ny + nz * nz));
    n[3 * (size_t)i] = len ? nx / len : nx;
    n[3 * (size_t)i + 1] = len ? ny / len : ny;
    n[3 * (size_t)i + 2] = len ? nz / len : nz;
    }
  return n;
  }

int main(int argc, const char** argv)
  {
  if (argc < 3)
    {
    usage();
    return -1;
    }
  const char* input = NULL;
  char output[1024];
  int have_output = 0;
  for (int j = 1; j < argc; ++j)
    {
    const int is_i = !strcmp(argv[j], "-i"), is_o = !strcmp(argv[j], "-o");
    if (!is_i && !is_o)
      {
      printf("Unknown command %s\n", argv[j]);
      return -1;
      }
    if (j == argc - 1)
      {
      printf("I expect a filename after command %s\n", argv[j]);
      return -1;
      }
    ++j;
    if (is_i)
      input = argv[j];
    else
      {
      snprintf(output, sizeof(output), "%s", argv[j]);
      have_output = 1;
      }
    }
  if (!input)
    {
    printf("An input file name is required\n");
    return -1;
    }
  FILE* f = fopen(input, "rb");
  long size = -1;
  if (f && fseek(f, 0, SEEK_END) == 0)
    size = ftell(f);
  if (size < 0)
    {
    if (f) fclose(f);
    printf("There was an error reading file %s\n", input);
    return -1;
    }
  rewind(f);
  unsigned char* blob = (unsigned char*)malloc((size_t)size + 1);
  if (!blob || fread(blob, 1, (size_t)size, f) != (size_t)size)
    {
    printf("There was an error reading file %s\n", input);
    fclose(f);
    return -1;
    }
  fclose(f);
  void* arch = trico_open_archive_for_reading(blob, (uint64_t)size);
  if (!arch)
    {
    printf("The input file %s is not a trico archive.\n", input);
    return -1;
    }

  float* vertices = NULL; float* tri_normals = NULL; float* vtx_normals = NULL; float* uv = NULL;
  uint32_t* triangles = NULL; uint32_t* colors = NULL;
  uint16_t* attributes = NULL;
  uint32_t nv = 0, nt = 0, nuv = 0;
  const char* failed = NULL;
  for (enum trico_stream_type st = trico_get_next_stream_type(arch); st != trico_empty && !failed;
       st = trico_get_next_stream_type(arch))
    {
    switch (st)
      {
      case trico_vertex_float_stream:
        nv = trico_get_number_of_vertices(arch);
        vertices = (float*)malloc((size_t)nv * 12 + 1);
        if (!trico_read_vertices(arch, &vertices)) failed = "vertices";
        break;
      case trico_triangle_normal_float_stream:
        tri_normals = (float*)malloc((size_t)trico_get_number_of_normals(arch) * 12 + 1);
        if (!trico_read_triangle_normals(arch, &tri_normals)) failed = "triangle normals";
        break;
      case trico_vertex_normal_float_stream:
        vtx_normals = (float*)malloc((size_t)trico_get_number_of_normals(arch) * 12 + 1);
        if (!trico_read_vertex_normals(arch, &vtx_normals)) failed = "vertex normals";
        break;
      case trico_vertex_color_stream:
        colors = (uint32_t*)malloc((size_t)trico_get_number_of_colors(arch) * 4 + 1);
        if (!trico_read_vertex_colors(arch, &colors)) failed = "vertex colors";
        break;
      case trico_triangle_uint32_stream:
        nt = trico_get_number_of_triangles(arch);
        triangles = (uint32_t*)malloc((size_t)nt * 12 + 1);
        if (!trico_read_triangles(arch, &triangles)) failed = "triangles";
        break;
      case trico_attribute_uint16_stream:
        attributes = (uint16_t*)malloc((size_t)trico_get_number_of_attributes(arch) * 2 + 1);
        if (!trico_read_attributes_uint16(arch, &attributes)) failed = "attributes";
        break;
      case trico_uv_per_triangle_float_stream:
        nuv = trico_get_number_of_uvs(arch);
        uv = (float*)malloc((size_t)nuv * 8 + 1);
        if (!trico_read_uv_per_triangle(arch, &uv)) failed = "texture coordinates";
        break;
      default:
        trico_skip_next_stream(arch);
        break;
      }
    }
  trico_close_archive(arch);
  free(blob);
  if (failed)
    {
    printf("Something went wrong when reading the %s\n", failed);
    return -1;
    }

  int as_stl = have_output && has_extension(output, "stl");
  int as_ply = have_output && has_extension(output, "ply");
  if (!as_stl && !as_ply)
    {
    if (colors || uv || vtx_normals) as_ply = 1;
    else as_stl = 1;
    }
  if (!have_output)
    with_extension(output, sizeof(output), input, as_ply ? "ply" : "stl");

  /* the archive is untrusted input: triangle indices must address the vertex stream that came with them, and the per-vertex
   * streams must have the vertex count (the writers below index with both) */
  {
  int sane = 1;
  if (nt && (!vertices || !triangles || nv == 0))
    sane = 0;
  for (size_t k = 0; sane && triangles && k < (size_t)nt * 3; ++k)
    if (triangles[k] >= nv)
      sane = 0;
  if (!sane)
    {
    printf("Inconsistent archive: triangle indices without matching vertices\n");
    return -1;
    }
  }
  int ok;
  if (as_stl)
    {
    if (!tri_normals)
      tri_normals = flat_normals(vertices, triangles, nt);
    ok = trico_write_stl(vertices, triangles, nt, tri_normals, attributes, output);
    }
  else
    {
    float* uv6 = NULL;
    if (uv)
      {
      uv6 = (float*)calloc((size_t)nt * 6 + 1, sizeof(float));
      if (uv6)
        memcpy(uv6, uv, sizeof(float) * ((size_t)nuv * 2 < (size_t)nt * 6 ? (size_t)nuv * 2 : (size_t)nt * 6));
      }
    ok = trico_write_ply(nv, vertices, vtx_normals, colors, nt, triangles, uv6, output);
    free(uv6);
    }
  if (!ok)
    {
    printf("Could not write to %s\n", output);
    return -1;
    }
  free(vertices); free(tri_normals); free(vtx_normals); free(colors); free(uv); free(triangles); free(attributes);
  return 0;
  }
