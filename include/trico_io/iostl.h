/* iostl.h — binary STL reader / writer around the trico hot path (§8(f) "callers either side").
 *
 * Same three entry points, argument meaning and return convention (1 ok, 0 error) as the reference's
 * trico_io/iostl.h:18-22; arrays handed out are malloc'ed and freed by the caller with free().
 * The reader welds identical corner positions into one vertex exactly like iostl.c:69-134: the unique
 * vertices come out in (x, y, z) lexicographic order and every triangle corner is re-indexed to its
 * vertex's rank, including the reference's choice of representative among positions that compare
 * equal but differ in bits (+0.0 / -0.0). */
#ifndef TRICO_IO_IOSTL_H
#define TRICO_IO_IOSTL_H

#include "trico_io_api.h"
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

TRICO_IO_API int trico_read_stl(uint32_t* nr_of_vertices, float** vertices, uint32_t* nr_of_triangles, uint32_t** triangles,
                                const char* filename);

TRICO_IO_API int trico_read_stl_full(uint32_t* nr_of_vertices, float** vertices, uint32_t* nr_of_triangles, uint32_t** triangles,
                                     float** normals, uint16_t** attributes, const char* filename);

TRICO_IO_API int trico_write_stl(const float* vertices, const uint32_t* triangles, const uint32_t nr_of_triangles,
                                 const float* triangle_normals, const uint16_t* attributes, const char* filename);

#if defined(__cplusplus)
}
#endif

#endif
