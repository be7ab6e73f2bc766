/*
 * trico.h — public C API of the MI355X-native Trico hot path.
 *
 * Drop-in boundary: same names, signatures, enum values, ownership and error convention as the
 * reference's trico/trico.h:11-94 (54 functions).  Written fresh; behaviour per SURVEY.md §8(b):
 *   - int results: 1 = ok, 0 = failure; open_* return NULL on failure; count peeks return 0 on
 *     type mismatch.  Nothing aborts, nothing logs.
 *   - a write archive owns its buffer (trico_get_buffer_pointer is invalidated by the next write);
 *     a read archive borrows the caller's bytes.
 *   - readers take T**: *ptr must be caller-allocated (count from the trico_get_number_of_* peek),
 *     except trico_read_attributes_float/double where the library mallocs *attrib (caller frees).
 *     Passing NULL decodes and discards the stream.
 *
 * MI355X extension (not in the reference): every data pointer handed to a writer/reader, and the
 * `data` pointer of trico_open_archive_for_reading, may be a HIP device pointer; it is detected
 * with hipPointerGetAttributes and then no PCIe staging happens.  See trico_hip.h for the
 * device-resident archive constructor.
 *
 * All compute runs on the GPU (hand-written HIP for gfx950).  There is no CPU fallback: if no HIP
 * device is usable every writer/reader returns 0 and trico_hip_last_error() says why.
 */
#ifndef TRICO_TRICO_H
#define TRICO_TRICO_H

#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#include "trico_api.h"

/* reference: trico/trico.h:11-34 */
enum trico_stream_type
  {
  trico_empty = 0,
  trico_vertex_float_stream = 1,
  trico_vertex_double_stream = 2,
  trico_triangle_uint32_stream = 3,
  trico_triangle_uint64_stream = 4,
  trico_uv_per_vertex_float_stream = 5,
  trico_uv_per_vertex_double_stream = 6,
  trico_uv_per_triangle_float_stream = 7,
  trico_uv_per_triangle_double_stream = 8,
  trico_vertex_normal_float_stream = 9,
  trico_vertex_normal_double_stream = 10,
  trico_triangle_normal_float_stream = 11,
  trico_triangle_normal_double_stream = 12,
  trico_vertex_color_stream = 13,
  trico_triangle_color_stream = 14,
  trico_attribute_float_stream = 15,
  trico_attribute_double_stream = 16,
  trico_attribute_uint8_stream = 17,
  trico_attribute_uint16_stream = 18,
  trico_attribute_uint32_stream = 19,
  trico_attribute_uint64_stream = 20
  };

/* ---- archives (reference: trico/trico.h:36-38, trico.c:126-189) ----------------------------------
 * A write archive owns a growing buffer (initial capacity in bytes); a read archive borrows `bytes`. */
TRICO_API void* trico_open_archive_for_writing(uint64_t initial_capacity);
TRICO_API void* trico_open_archive_for_reading(const uint8_t* bytes, uint64_t byte_count);
TRICO_API void  trico_close_archive(void* arc);

/* ---- writers (reference: trico/trico.h:40-60, trico.c:215-858) ----------------------------------
 * Each call appends one stream.  `n` counts elements of the stream's natural unit: vertices (3 reals),
 * triangles (3 indices), uv positions (2 reals; the per-triangle float writer stores 3 * n of them,
 * trico.c:579), normals (3 reals), colours (1 uint32), attributes (1 scalar). */
TRICO_API int trico_write_vertices                (void* arc, const float*    xyz,     uint32_t n);
TRICO_API int trico_write_vertices_double         (void* arc, const double*   xyz,     uint32_t n);
TRICO_API int trico_write_triangles               (void* arc, const uint32_t* corners, uint32_t n);
TRICO_API int trico_write_triangles_long          (void* arc, const uint64_t* corners, uint32_t n);
TRICO_API int trico_write_uv_per_vertex           (void* arc, const float*    uv,      uint32_t n);
TRICO_API int trico_write_uv_per_vertex_double    (void* arc, const double*   uv,      uint32_t n);
TRICO_API int trico_write_uv_per_triangle         (void* arc, const float*    uv,      uint32_t n);
TRICO_API int trico_write_uv_per_triangle_double  (void* arc, const double*   uv,      uint32_t n);
TRICO_API int trico_write_vertex_normals          (void* arc, const float*    nxyz,    uint32_t n);
TRICO_API int trico_write_vertex_normals_double   (void* arc, const double*   nxyz,    uint32_t n);
TRICO_API int trico_write_triangle_normals        (void* arc, const float*    nxyz,    uint32_t n);
TRICO_API int trico_write_triangle_normals_double (void* arc, const double*   nxyz,    uint32_t n);
TRICO_API int trico_write_vertex_colors           (void* arc, const uint32_t* rgba,    uint32_t n);
TRICO_API int trico_write_triangle_colors         (void* arc, const uint32_t* rgba,    uint32_t n);
TRICO_API int trico_write_attributes_float        (void* arc, const float*    values,  uint32_t n);
TRICO_API int trico_write_attributes_double       (void* arc, const double*   values,  uint32_t n);
TRICO_API int trico_write_attributes_uint8        (void* arc, const uint8_t*  values,  uint32_t n);
TRICO_API int trico_write_attributes_uint16       (void* arc, const uint16_t* values,  uint32_t n);
TRICO_API int trico_write_attributes_uint32       (void* arc, const uint32_t* values,  uint32_t n);
TRICO_API int trico_write_attributes_uint64       (void* arc, const uint64_t* values,  uint32_t n);

/* ---- the encoded bytes (reference: trico/trico.h:62-63, trico.c:191-201) ------------------------- */
TRICO_API uint8_t* trico_get_buffer_pointer(void* arc);     /* invalidated by the next write */
TRICO_API uint64_t trico_get_size(void* arc);

/* ---- cursor (reference: trico/trico.h:65-73, trico.c:203-213, 860-941) ---------------------------
 * The count peeks return 0 unless the next stream is of the matching kind; they do not advance. */
TRICO_API uint32_t               trico_get_version(void* arc);
TRICO_API enum trico_stream_type trico_get_next_stream_type(void* arc);
TRICO_API uint32_t trico_get_number_of_vertices  (void* arc);
TRICO_API uint32_t trico_get_number_of_triangles (void* arc);
TRICO_API uint32_t trico_get_number_of_uvs       (void* arc);
TRICO_API uint32_t trico_get_number_of_normals   (void* arc);
TRICO_API uint32_t trico_get_number_of_colors    (void* arc);
TRICO_API uint32_t trico_get_number_of_attributes(void* arc);

/* ---- readers (reference: trico/trico.h:75-95, trico.c:943-1698) ----------------------------------
 * `*out` is caller-allocated for the count the peek announced (attributes_float / _double: allocated by
 * the library with malloc, caller frees); out == NULL skips the stream, as does trico_skip_next_stream. */
TRICO_API int trico_read_vertices                (void* arc, float**    out);
TRICO_API int trico_read_vertices_double         (void* arc, double**   out);
TRICO_API int trico_read_triangles               (void* arc, uint32_t** out);
TRICO_API int trico_read_triangles_long          (void* arc, uint64_t** out);
TRICO_API int trico_read_uv_per_vertex           (void* arc, float**    out);
TRICO_API int trico_read_uv_per_vertex_double    (void* arc, double**   out);
TRICO_API int trico_read_uv_per_triangle         (void* arc, float**    out);
TRICO_API int trico_read_uv_per_triangle_double  (void* arc, double**   out);
TRICO_API int trico_read_vertex_normals          (void* arc, float**    out);
TRICO_API int trico_read_vertex_normals_double   (void* arc, double**   out);
TRICO_API int trico_read_triangle_normals        (void* arc, float**    out);
TRICO_API int trico_read_triangle_normals_double (void* arc, double**   out);
TRICO_API int trico_read_vertex_colors           (void* arc, uint32_t** out);
TRICO_API int trico_read_triangle_colors         (void* arc, uint32_t** out);
TRICO_API int trico_read_attributes_float        (void* arc, float**    out);
TRICO_API int trico_read_attributes_double       (void* arc, double**   out);
TRICO_API int trico_read_attributes_uint8        (void* arc, uint8_t**  out);
TRICO_API int trico_read_attributes_uint16       (void* arc, uint16_t** out);
TRICO_API int trico_read_attributes_uint32       (void* arc, uint32_t** out);
TRICO_API int trico_read_attributes_uint64       (void* arc, uint64_t** out);
TRICO_API int trico_skip_next_stream             (void* arc);

#if defined(__cplusplus)
}
#endif

#endif /* TRICO_TRICO_H */
