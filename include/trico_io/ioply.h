/* ioply.h — PLY reader / writer around the trico hot path (§8(f)).
 *
 * Same two entry points, argument meaning and return convention as the reference's trico_io/ioply.h:12-14.
 * The reference parses with the third-party rply library (rply/rply.c, v1.1.4); this implementation has its
 * own parser for the same files: ascii, binary_little_endian and binary_big_endian bodies, scalar and list
 * properties of the eight PLY types under both naming schemes.  What is extracted (ioply.c:68-246):
 *   vertex x,y,z -> float xyz; nx,ny,nz -> float normals; red/green/blue/alpha (or r/g/b/a, or diffuse_*) ->
 *   one uint32 per vertex (bytes r,g,b,a; channels that are absent stay 0xff); face vertex_indices (or
 *   vertex_index) -> the first three indices of every face; face texcoord -> six floats per face, zero-padded. */
#ifndef TRICO_IO_IOPLY_H
#define TRICO_IO_IOPLY_H

#include "trico_io_api.h"
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

TRICO_IO_API int trico_read_ply(uint32_t* nr_of_vertices, float** vertices, float** vertex_normals, uint32_t** vertex_colors,
                                uint32_t* nr_of_triangles, uint32_t** triangles, float** texcoords, const char* filename);

TRICO_IO_API int trico_write_ply(const uint32_t nr_of_vertices, const float* vertices, const float* vertex_normals,
                                 const uint32_t* vertex_colors, const uint32_t nr_of_triangles, const uint32_t* triangles,
                                 const float* texcoords, const char* filename);

#if defined(__cplusplus)
}
#endif

#endif
