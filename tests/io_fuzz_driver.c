#include "trico_io/iostl.h"
#include "trico_io/ioply.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
/* test-only stand-ins for the device entry points the STL reader may call: "no device" */
struct trico_hip_ctx;
int trico_hip_available(void) { return 0; }
struct trico_hip_ctx* trico_hip_ctx_create(void) { return NULL; }
void trico_hip_ctx_destroy(struct trico_hip_ctx* c) { (void)c; }
int trico_hip_weld_vertices(struct trico_hip_ctx* c, const float* a, uint32_t n, float* v, uint32_t* t, uint32_t* nv) { (void)c; (void)a; (void)n; (void)v; (void)t; (void)nv; return 0; }
int main(int argc, char** argv)
  {
  for (int i = 2; i < argc; ++i)   /* argv[1]: scratch output file */
    {
    uint32_t nv, nt; float *v, *n, *uv; uint32_t *c, *t; uint16_t* a;
    const char* f = argv[i];
    const size_t L = strlen(f);
    int rc;
    if (L > 4 && !strcmp(f + L - 4, ".stl"))
      {
      rc = trico_read_stl_full(&nv, &v, &nt, &t, &n, &a, f);
      if (rc) { trico_write_stl(v, t, nt, n, a, argv[1]); free(v); free(t); free(n); free(a); }
      rc = trico_read_stl(&nv, &v, &nt, &t, f);
      if (rc) { free(v); free(t); }
      }
    else
      {
      rc = trico_read_ply(&nv, &v, &n, &c, &nt, &t, &uv, f);
      if (rc) { trico_write_ply(nv, v, n, c, nt, t, uv, argv[1]); free(v); free(n); free(c); free(t); free(uv); }
      }
    printf("%s rc=%d\n", f, rc);
    }
  return 0;
  }
