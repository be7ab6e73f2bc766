/* batch_decode.c — many archives, one launch: K small meshes are written, then ALL of their streams are decoded as one batch
 * with trico_hip_read_archives (include/trico/trico_hip.h): every float chain of every archive in one kernel launch.
 *
 *   gcc -Iinclude examples/batch_decode.c -Ltrico_amd/lib -ltrico -Wl,-rpath,$PWD/trico_amd/lib -o batch_decode && ./batch_decode
 *
 * (The archive API part is the reference's, trico/trico.h:36-94; trico_hip_* is this library's extension.) */
#include <trico/alloc.h>
#include <trico/trico.h>
#include <trico/trico_hip.h>

#include <stdio.h>
#include <string.h>

#define K 5
#define NV 1000

int main(void)
  {
  static float xyz[K][3 * NV];
  static uint32_t tri[K][3 * (NV - 2)];
  uint8_t* bytes[K];
  uint64_t size[K];
  for (int k = 0; k < K; ++k)
    {
    for (int i = 0; i < NV; ++i)
      {
      xyz[k][3 * i] = 0.25f * (float)i;
      xyz[k][3 * i + 1] = (float)(k + 1) * 0.5f;
      xyz[k][3 * i + 2] = (float)((i * i + 7 * k) % 97) * 0.015625f;
      }
    for (int i = 0; i < NV - 2; ++i)
      {
      tri[k][3 * i] = (uint32_t)i; tri[k][3 * i + 1] = (uint32_t)i + 1u; tri[k][3 * i + 2] = (uint32_t)i + 2u;
      }
    void* w = trico_open_archive_for_writing(1024);
    if (!w || !trico_write_vertices(w, xyz[k], NV) || !trico_write_triangles(w, tri[k], NV - 2))
      {
      fprintf(stderr, "encode failed: %s\n", trico_hip_last_error());
      return 1;
      }
    size[k] = trico_get_size(w);
    bytes[k] = (uint8_t*)trico_malloc(size[k]);
    memcpy(bytes[k], trico_get_buffer_pointer(w), size[k]);
    trico_close_archive(w);
    }
  /* open all, ask what is in them, allocate, decode everything at once */
  void* arch[K];
  void* rows[K][2];
  void* const* dsts[K];
  int nstreams[K];
  for (int k = 0; k < K; ++k)
    {
    arch[k] = trico_open_archive_for_reading(bytes[k], size[k]);
    trico_hip_stream_info info[2];
    if (!arch[k] || trico_hip_list_streams(arch[k], info, 2) != 2)
      return 2;
    for (int s = 0; s < 2; ++s)
      rows[k][s] = trico_malloc(info[s].decoded_bytes);
    dsts[k] = rows[k];
    nstreams[k] = 2;
    }
  if (!trico_hip_read_archives(arch, K, dsts, nstreams))
    {
    fprintf(stderr, "batch decode failed: %s\n", trico_hip_last_error());
    return 3;
    }
  int same = 1;
  for (int k = 0; k < K; ++k)
    {
    same = same && trico_get_next_stream_type(arch[k]) == trico_empty && !memcmp(rows[k][0], xyz[k], sizeof(xyz[k])) &&
           !memcmp(rows[k][1], tri[k], sizeof(tri[k]));
    trico_close_archive(arch[k]);
    trico_free(rows[k][0]); trico_free(rows[k][1]); trico_free(bytes[k]);
    }
  printf("%d archives, %d streams, one batch: round trip %s\n", K, 2 * K, same ? "exact" : "DIFFERS");
  return same ? 0 : 4;
  }
