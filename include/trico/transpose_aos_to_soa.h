/*
 * transpose_aos_to_soa.h — the layout changes of the path as a stand-alone API.
 *
 * Same fourteen entry points as the reference's trico/transpose_aos_to_soa.h:12-38.  The archive writers / readers
 * of this library fuse these steps into their kernels; the functions exist for callers of the reference's low-level
 * API.  Every output (`*x`, `*xyz`, `*p0` ...) is caller-allocated, passed through one level of indirection exactly
 * like the reference (transpose_aos_to_soa.c:8-16: `(*x)[i] = ...`); inputs and outputs may be host or HIP device
 * pointers.  The functions return void: without a usable device nothing is written and trico_hip_last_error() tells.
 */
#ifndef TRICO_TRANSPOSE_AOS_TO_SOA_H
#define TRICO_TRANSPOSE_AOS_TO_SOA_H

#include "trico.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* xyz / uv components of n interleaved positions */
TRICO_API void trico_transpose_xyz_aos_to_soa(float** x, float** y, float** z, const float* xyz, uint32_t n);
TRICO_API void trico_transpose_xyz_soa_to_aos(float** xyz, const float* x, const float* y, const float* z, uint32_t n);
TRICO_API void trico_transpose_xyz_aos_to_soa_double_precision(double** x, double** y, double** z, const double* xyz, uint32_t n);
TRICO_API void trico_transpose_xyz_soa_to_aos_double_precision(double** xyz, const double* x, const double* y, const double* z, uint32_t n);
TRICO_API void trico_transpose_uv_aos_to_soa(float** u, float** v, const float* uv, uint32_t n);
TRICO_API void trico_transpose_uv_soa_to_aos(float** uv, const float* u, const float* v, uint32_t n);
TRICO_API void trico_transpose_uv_aos_to_soa_double_precision(double** u, double** v, const double* uv, uint32_t n);
TRICO_API void trico_transpose_uv_soa_to_aos_double_precision(double** uv, const double* u, const double* v, uint32_t n);

/* byte planes of n little-endian integers: plane k holds byte k of every integer */
TRICO_API void trico_transpose_uint16_aos_to_soa(uint8_t** p0, uint8_t** p1, const uint16_t* values, uint32_t n);
TRICO_API void trico_transpose_uint16_soa_to_aos(uint16_t** values, const uint8_t* p0, const uint8_t* p1, uint32_t n);
TRICO_API void trico_transpose_uint32_aos_to_soa(uint8_t** p0, uint8_t** p1, uint8_t** p2, uint8_t** p3, const uint32_t* values, uint32_t n);
TRICO_API void trico_transpose_uint32_soa_to_aos(uint32_t** values, const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3,
                                                 uint32_t n);
TRICO_API void trico_transpose_uint64_aos_to_soa(uint8_t** p0, uint8_t** p1, uint8_t** p2, uint8_t** p3, uint8_t** p4, uint8_t** p5,
                                                 uint8_t** p6, uint8_t** p7, const uint64_t* values, uint32_t n);
TRICO_API void trico_transpose_uint64_soa_to_aos(uint64_t** values, const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3,
                                                 const uint8_t* p4, const uint8_t* p5, const uint8_t* p6, const uint8_t* p7, uint32_t n);

#if defined(__cplusplus)
}
#endif

#endif
