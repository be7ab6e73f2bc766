/* Test-only: walks (possibly corrupted) archives through the container functions of archive.c with the device
 * shim replaced by stubs that report "no device", under ASan/UBSan.  Nothing here is part of the product. */
#include "trico/trico.h"
#include "trico/trico_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int trico_hip_available(void) { return 0; }
const char* trico_hip_last_error(void) { return "stub"; }
trico_hip_ctx* trico_hip_ctx_create(void) { return NULL; }
void trico_hip_ctx_destroy(trico_hip_ctx* c) { (void)c; }
int trico_hip_pointer_is_device(const void* p) { (void)p; return 0; }
void* trico_hip_device_alloc(size_t n) { (void)n; return NULL; }
void trico_hip_device_free(void* p) { (void)p; }
int trico_hip_copy(void* d, const void* s, size_t n) { memcpy(d, s, n); return 1; }
uint64_t trico_hip_device_free_bytes(void) { return 0; }
int trico_hip_fpc_encode(trico_hip_ctx* c, const void* s, uint32_t n, int a, int w, uint32_t z[3]) { (void)c; (void)s; (void)n; (void)a; (void)w; (void)z; return 0; }
int trico_hip_fpc_decode(trico_hip_ctx* c, const uint8_t* const p[3], const uint32_t z[3], int a, int w, uint32_t n, void* d) { (void)c; (void)p; (void)z; (void)a; (void)w; (void)n; (void)d; return 0; }
int trico_hip_int_encode(trico_hip_ctx* c, const void* s, uint32_t n, int w, uint32_t z[8]) { (void)c; (void)s; (void)n; (void)w; (void)z; return 0; }
int trico_hip_int_decode(trico_hip_ctx* c, const uint8_t* const p[8], const uint32_t z[8], int w, uint32_t n, void* d) { (void)c; (void)p; (void)z; (void)w; (void)n; (void)d; return 0; }
int trico_hip_fetch_payload(trico_hip_ctx* c, int i, void* d) { (void)c; (void)i; (void)d; return 0; }
int trico_hip_fetch_payloads(trico_hip_ctx* c, int n, void* const* d) { (void)c; (void)n; (void)d; return 0; }
int trico_hip_fpc_encode_place(trico_hip_ctx* c, const void* s, uint32_t n, int a, int w, void* d, uint32_t* z) { (void)c; (void)s; (void)n; (void)a; (void)w; (void)d; (void)z; return 0; }
int trico_hip_decode_jobs(trico_hip_decode_job* j, int n) { for (int i = 0; i < n; ++i) j[i].ok = j[i].dst == NULL; return 0; }
uint32_t trico_hip_ctx_other_writer_streams(const trico_hip_ctx* c) { (void)c; return 0; }
int trico_hip_walk_frames(const uint8_t* d, uint64_t z, uint64_t p, const uint8_t t[21], trico_hip_frame_bytes* o, int c, uint8_t h[8]) { (void)d; (void)z; (void)p; (void)t; (void)o; (void)c; (void)h; return -1; }

int main(int argc, char** argv)
  {
  for (int i = 1; i < argc; ++i)
    {
    FILE* f = fopen(argv[i], "rb");
    if (!f) continue;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    rewind(f);
    uint8_t* blob = (uint8_t*)malloc((size_t)n + 1);          /* exact size: reads past the end are caught */
    if (fread(blob, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(blob); continue; }
    fclose(f);
    void* a = trico_open_archive_for_reading(blob, (uint64_t)n);
    int streams = 0;
    if (a)
      {
      float* dummy = (float*)malloc(16);
      /* the whole-archive entry points walk the same frames: listing, then a batch of skips (NULL destinations) on a second handle */
      trico_hip_stream_info infos[8];
      const int listed = trico_hip_list_streams(a, infos, 8);
      if (listed > 0)
        {
        void* b = trico_open_archive_for_reading(blob, (uint64_t)n);
        void* dsts[8] = { NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL };
        void* const* rows[1] = { dsts };
        const int cnt = listed < 8 ? listed : 8;
        (void)trico_hip_read_archives(&b, 1, rows, &cnt);
        trico_close_archive(b);
        }
      for (int guard = 0; guard < 1000; ++guard)
        {
        const enum trico_stream_type st = trico_get_next_stream_type(a);
        if (st == trico_empty)
          break;
        (void)trico_get_number_of_vertices(a); (void)trico_get_number_of_triangles(a); (void)trico_get_number_of_uvs(a);
        (void)trico_get_number_of_normals(a); (void)trico_get_number_of_colors(a); (void)trico_get_number_of_attributes(a);
        (void)trico_read_vertices(a, &dummy);                   /* must fail cleanly without a device */
        if (!trico_skip_next_stream(a))
          break;
        ++streams;
        }
      free(dummy);
      trico_close_archive(a);
      }
    printf("%s opened=%d streams=%d\n", argv[i], a != NULL, streams);
    free(blob);
    }
  return 0;
  }
