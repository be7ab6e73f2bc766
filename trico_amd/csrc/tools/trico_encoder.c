/* trico_encoder — command line front end: binary STL / PLY -> .trc  (§8(f), tools/trico_encoder/main.c).
 *
 * Same command line, messages, exit codes and output bytes as the reference tool, built on libtrico.so (the
 * MI355X hot path) and libtrico_io.so.  Behaviour kept on purpose, because it decides which streams a
 * given command line produces (tools/trico_encoder/main.c:143-191, 296-302):
 *   - the handlers of the two attribute options are crossed: "-stladd normal|tex_coord|color" sets the PLY
 *     skip flags and "-plyskip normal|uint16" sets the STL include flags;
 *   - PLY texture coordinates (6 floats per face) are written with trico_write_uv_per_triangle(count =
 *     number of triangles), i.e. the first third of the array.
 * Stream order: vertices, triangles, then triangle normals / uint16 attributes (STL) or vertex normals,
 * vertex colours, texture coordinates (PLY). */
#include "trico/trico.h"
#include "trico_io/iostl.h"
#include "trico_io/ioply.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

/* position of the last '.' that is not at index 0, or -1 (the reference's scan stops before index 0) */
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
  printf("Usage: trico_encoder -i <input> [options]\n\n");
  printf("Options:\n");
  printf("  -i <input>           input file name of type binary stl or binary/ascii ply.\n");
  printf("  -o <output>          output file name.\n");
  printf("  -stladd <attribute>  add a given stl attribute (normal, uint16).\n");
  printf("  -plyskip <attribute> skip a given ply attribute (normal, tex_coord, color).\n");
  printf("\n");
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
  int stl_normals = 0, stl_uint16 = 0, skip_normals = 0, skip_texcoords = 0, skip_color = 0;
  for (int j = 1; j < argc; ++j)
    {
    const char* opt = argv[j];
    const int takes_value = !strcmp(opt, "-i") || !strcmp(opt, "-o") || !strcmp(opt, "-stladd") || !strcmp(opt, "-plyskip");
    if (!takes_value)
      {
      printf("Unknown command %s\n", opt);
      return -1;
      }
    if (j == argc - 1)
      {
      printf("I expect %s after command %s\n", (opt[1] == 'i' || opt[1] == 'o') ? "a filename" : "an attribute", opt);
      return -1;
      }
    const char* val = argv[++j];
    if (!strcmp(opt, "-i"))
      input = val;
    else if (!strcmp(opt, "-o"))
      {
      snprintf(output, sizeof(output), "%s", val);
      have_output = 1;
      }
    else if (!strcmp(opt, "-stladd"))                /* sic: sets the ply flags */
      {
      if (!strcmp(val, "normal")) skip_normals = 1;
      else if (!strcmp(val, "tex_coord")) skip_texcoords = 1;
      else if (!strcmp(val, "color")) skip_color = 1;
      else { printf("Unknown attribute %s\n", val); return -1; }
      }
    else                                             /* -plyskip, sic: sets the stl flags */
      {
      if (!strcmp(val, "normal")) stl_normals = 1;
      else if (!strcmp(val, "uint16")) stl_uint16 = 1;
      else { printf("Unknown attribute %s\n", val); return -1; }
      }
    }
  if (!input)
    {
    printf("An input file name is required\n");
    return -1;
    }
  if (!have_output)
    with_extension(output, sizeof(output), input, "trc");
  const int is_stl = has_extension(input, "stl"), is_ply = has_extension(input, "ply");
  if (!is_stl && !is_ply)
    {
    printf("I expect the input file to be of type stl or ply.\n");
    return -1;
    }

  uint32_t nv = 0, nt = 0;
  float* vertices = NULL; float* tri_normals = NULL; float* vtx_normals = NULL; float* texcoords = NULL;
  uint32_t* colors = NULL; uint32_t* triangles = NULL;
  uint16_t* attributes = NULL;
  if (is_stl)
    {
    const int ok = (stl_normals || stl_uint16) ? trico_read_stl_full(&nv, &vertices, &nt, &triangles, &tri_normals, &attributes, input)
                                               : trico_read_stl(&nv, &vertices, &nt, &triangles, input);
    if (ok != 1)
      {
      printf("Not a valid stl file: %s\n", input);
      return -1;
      }
    }
  else if (trico_read_ply(&nv, &vertices, &vtx_normals, &colors, &nt, &triangles, &texcoords, input) != 1)
    {
    printf("Not a valid ply file: %s\n", input);
    return -1;
    }

  void* arch = trico_open_archive_for_writing(1024 * 1024);
  const struct { int wanted; const char* what; } steps[7] = {
    { nv && vertices, "vertices" }, { nt && triangles, "triangles" },
    { is_stl && stl_normals && nt && tri_normals, "triangle normals" },
    { is_stl && stl_uint16 && nt && attributes, "uint16 attributes" },
    { is_ply && !skip_normals && nv && vtx_normals, "vertex normals" },
    { is_ply && !skip_color && nv && colors, "vertex colors" },
    { is_ply && !skip_texcoords && nt && texcoords, "texture coordinates" } };
  for (int s = 0; s < 7; ++s)
    {
    if (!steps[s].wanted)
      continue;
    int ok = 0;
    switch (s)
      {
      case 0: ok = trico_write_vertices(arch, vertices, nv); break;
      case 1: ok = trico_write_triangles(arch, triangles, nt); break;
      case 2: ok = trico_write_triangle_normals(arch, tri_normals, nt); break;
      case 3: ok = trico_write_attributes_uint16(arch, attributes, nt); break;
      case 4: ok = trico_write_vertex_normals(arch, vtx_normals, nv); break;
      case 5: ok = trico_write_vertex_colors(arch, colors, nv); break;
      default: ok = trico_write_uv_per_triangle(arch, texcoords, nt); break;
      }
    if (!ok)
      {
      printf("Something went wrong when writing the %s\n", steps[s].what);
      return -1;
      }
    }
  free(vertices); free(vtx_normals); free(colors); free(triangles); free(texcoords); free(tri_normals); free(attributes);

  FILE* f = fopen(output, "wb");
  if (!f)
    {
    printf("Cannot write to file %s\n", output);
    return -1;
    }
  fwrite(trico_get_buffer_pointer(arch), trico_get_size(arch), 1, f);
  fclose(f);
  trico_close_archive(arch);
  return 0;
  }
