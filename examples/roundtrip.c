/* roundtrip.c — smallest possible user of the drop-in API: a tetrahedron through a .trc archive and back.
 *
 *   gcc -Iinclude examples/roundtrip.c -Ltrico_amd/lib -ltrico -Wl,-rpath,$PWD/trico_amd/lib -o roundtrip && ./roundtrip
 *
 * The same source compiles unchanged against the reference's trico library (that is the point). */
#include <trico/trico.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(void)
  {
  const float xyz[12] = { 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f };
  const uint32_t tri[12] = { 0, 2, 1, 0, 1, 3, 0, 3, 2, 1, 2, 3 };

  void* w = trico_open_archive_for_writing(1024);
  if (!w || !trico_write_vertices(w, xyz, 4) || !trico_write_triangles(w, tri, 4))
    {
    fprintf(stderr, "encode failed (is there an MI355X in this machine?)\n");
    return 1;
    }
  const uint64_t size = trico_get_size(w);
  uint8_t* bytes = (uint8_t*)malloc(size);
  memcpy(bytes, trico_get_buffer_pointer(w), size);
  trico_close_archive(w);
  printf("archive: %llu bytes\n", (unsigned long long)size);

  void* r = trico_open_archive_for_reading(bytes, size);
  float* v = NULL;
  uint32_t* t = NULL;
  while (r && trico_get_next_stream_type(r) != trico_empty)
    {
    switch (trico_get_next_stream_type(r))
      {
      case trico_vertex_float_stream:
        v = (float*)malloc(sizeof(float) * 3 * trico_get_number_of_vertices(r));
        if (!trico_read_vertices(r, &v)) return 2;
        break;
      case trico_triangle_uint32_stream:
        t = (uint32_t*)malloc(sizeof(uint32_t) * 3 * trico_get_number_of_triangles(r));
        if (!trico_read_triangles(r, &t)) return 2;
        break;
      default:
        if (!trico_skip_next_stream(r)) return 2;
      }
    }
  trico_close_archive(r);
  const int same = v && t && !memcmp(v, xyz, sizeof(xyz)) && !memcmp(t, tri, sizeof(tri));
  printf("round trip %s\n", same ? "exact" : "DIFFERS");
  free(v); free(t); free(bytes);
  return same ? 0 : 3;
  }
