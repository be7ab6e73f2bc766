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

#ifndef TRICO_API
#define TRICO_API __attribute__((visibility("default")))
#endif

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

/* reference: trico/trico.h:36-38, trico.c:126-189 */
TRICO_API void* trico_open_archive_for_writing(uint64_t initial_buffer_size);
TRICO_API void* trico_open_archive_for_reading(const uint8_t* data, uint64_t data_size);
TRICO_API void trico_close_archive(void* archive);

/* reference: trico/trico.h:40-60, trico.c:215-858 */
TRICO_API int trico_write_vertices(void* archive, const float* vertices, uint32_t nr_of_vertices);
TRICO_API int trico_write_vertices_double(void* archive, const double* vertices, uint32_t nr_of_vertices);
TRICO_API int trico_write_triangles(void* archive, const uint32_t* tria_indices, uint32_t nr_of_triangles);
TRICO_API int trico_write_triangles_long(void* archive, const uint64_t* tria_indices, uint32_t nr_of_triangles);
TRICO_API int trico_write_uv_per_vertex(void* archive, const float* uv, uint32_t nr_of_uv_positions);
TRICO_API int trico_write_uv_per_vertex_double(void* archive, const double* uv, uint32_t nr_of_uv_positions);
TRICO_API int trico_write_uv_per_triangle(void* archive, const float* uv, uint32_t nr_of_uv_positions);
TRICO_API int trico_write_uv_per_triangle_double(void* archive, const double* uv, uint32_t nr_of_uv_positions);
TRICO_API int trico_write_vertex_normals(void* archive, const float* normals, uint32_t nr_of_normals);
TRICO_API int trico_write_vertex_normals_double(void* archive, const double* normals, uint32_t nr_of_normals);
TRICO_API int trico_write_triangle_normals(void* archive, const float* normals, uint32_t nr_of_normals);
TRICO_API int trico_write_triangle_normals_double(void* archive, const double* normals, uint32_t nr_of_normals);
TRICO_API int trico_write_vertex_colors(void* archive, const uint32_t* color, uint32_t nr_of_colors);
TRICO_API int trico_write_triangle_colors(void* archive, const uint32_t* color, uint32_t nr_of_colors);
TRICO_API int trico_write_attributes_float(void* archive, const float* attrib, uint32_t nr_of_attribs);
TRICO_API int trico_write_attributes_double(void* archive, const double* attrib, uint32_t nr_of_attribs);
TRICO_API int trico_write_attributes_uint8(void* archive, const uint8_t* attrib, uint32_t nr_of_attribs);
TRICO_API int trico_write_attributes_uint16(void* archive, const uint16_t* attrib, uint32_t nr_of_attribs);
TRICO_API int trico_write_attributes_uint32(void* archive, const uint32_t* attrib, uint32_t nr_of_attribs);
TRICO_API int trico_write_attributes_uint64(void* archive, const uint64_t* attrib, uint32_t nr_of_attribs);

/* reference: trico/trico.h:62-63, trico.c:191-201 */
TRICO_API uint8_t* trico_get_buffer_pointer(void* archive);
TRICO_API uint64_t trico_get_size(void* archive);

/* reference: trico/trico.h:65-66, trico.c:203-213 */
TRICO_API uint32_t trico_get_version(void* archive);
TRICO_API enum trico_stream_type trico_get_next_stream_type(void* archive);

/* reference: trico/trico.h:68-73, trico.c:860-941 (peek, no advance) */
TRICO_API uint32_t trico_get_number_of_vertices(void* archive);
TRICO_API uint32_t trico_get_number_of_triangles(void* archive);
TRICO_API uint32_t trico_get_number_of_uvs(void* archive);
TRICO_API uint32_t trico_get_number_of_normals(void* archive);
TRICO_API uint32_t trico_get_number_of_colors(void* archive);
TRICO_API uint32_t trico_get_number_of_attributes(void* archive);

/* reference: trico/trico.h:75-95, trico.c:943-1698 */
TRICO_API int trico_read_vertices(void* archive, float** vertices);
TRICO_API int trico_read_vertices_double(void* archive, double** vertices);
TRICO_API int trico_read_triangles(void* archive, uint32_t** triangles);
TRICO_API int trico_read_triangles_long(void* archive, uint64_t** triangles);
TRICO_API int trico_read_uv_per_vertex(void* archive, float** uv);
TRICO_API int trico_read_uv_per_vertex_double(void* archive, double** uv);
TRICO_API int trico_read_uv_per_triangle(void* archive, float** uv);
TRICO_API int trico_read_uv_per_triangle_double(void* archive, double** uv);
TRICO_API int trico_read_vertex_normals(void* archive, float** normals);
TRICO_API int trico_read_vertex_normals_double(void* archive, double** normals);
TRICO_API int trico_read_triangle_normals(void* archive, float** normals);
TRICO_API int trico_read_triangle_normals_double(void* archive, double** normals);
TRICO_API int trico_read_vertex_colors(void* archive, uint32_t** color);
TRICO_API int trico_read_triangle_colors(void* archive, uint32_t** color);
TRICO_API int trico_read_attributes_float(void* archive, float** attrib);
TRICO_API int trico_read_attributes_double(void* archive, double** attrib);
TRICO_API int trico_read_attributes_uint8(void* archive, uint8_t** attrib);
TRICO_API int trico_read_attributes_uint16(void* archive, uint16_t** attrib);
TRICO_API int trico_read_attributes_uint32(void* archive, uint32_t** attrib);
TRICO_API int trico_read_attributes_uint64(void* archive, uint64_t** attrib);
TRICO_API int trico_skip_next_stream(void* archive);

#if defined(__cplusplus)
}
#endif

#endif /* TRICO_TRICO_H */
