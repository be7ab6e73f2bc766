/*
 * alloc.h — the allocator hook of the public API.
 *
 * Counterpart of the reference's trico/alloc.h:12-30: four `static inline` wrappers over the C library, which is the
 * reference's only allocator customisation point (compile-time: a consumer edits or shadows this header).  Callers need
 * it for one thing: memory the library hands out is released with trico_free -
 *   - the arrays of trico_read_stl / trico_read_ply (README.md:70-73 of the reference),
 *   - the array trico_read_attributes_float / _double allocates (trico.c:1377, 1408),
 *   - the outputs of the low-level coders and transposes (floating_point_stream_compression.h, transpose_aos_to_soa.h).
 * libtrico.so / libtrico_io.so allocate all of those with malloc, so these wrappers pair with them.
 * (Device memory never goes through this header: see trico_hip_device_alloc / _free in trico_hip.h.)
 */
#ifndef TRICO_ALLOC_H
#define TRICO_ALLOC_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#if defined(__cplusplus)
extern "C" {
#endif

static inline void* trico_malloc(size_t bytes) { return malloc(bytes); }
static inline void* trico_calloc(size_t count, size_t bytes_each) { return calloc(count, bytes_each); }
static inline void* trico_realloc(void* p, size_t new_bytes) { return realloc(p, new_bytes); }
static inline void trico_free(void* p) { free(p); }

#if defined(__cplusplus)
}
#endif

#endif /* TRICO_ALLOC_H */
