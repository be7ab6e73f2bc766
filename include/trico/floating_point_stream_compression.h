/*
 * floating_point_stream_compression.h — the stream coder as a stand-alone API.
 *
 * Same four entry points as the reference's trico/floating_point_stream_compression.h:11-17 (the archive
 * writers / readers call them per component, trico.c:231, 396, 1002, 1190).  All compute runs on the MI355X;
 * `input` may be a host or a HIP device pointer, results are returned in host memory allocated with malloc
 * (caller frees), as in the reference.
 *
 * Differences, forced by the missing error channel (the functions return void):
 *   - on failure (no device, malformed stream, tables that do not fit the device) *out is NULL and the count 0;
 *   - table size exponents: any value, normalised like the reference does (odd values rounded down, at most 30;
 *     fpsc.c:88-93, 578-583).  The archive API's (4,10) / (20,20) take the throughput kernels, every other shape the
 *     reference-order kernel with tables of 2^e1 + 2^e2 entries in device memory;
 *   - the decoders take a HOST pointer: the format does not carry its own length, so the group headers are
 *     walked on the host to find the end of the stream before it is handed to the device (the reference simply
 *     trusts its input, fpsc.c:212-417).
 */
#ifndef TRICO_FLOATING_POINT_STREAM_COMPRESSION_H
#define TRICO_FLOATING_POINT_STREAM_COMPRESSION_H

#include "trico.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* float stream -> payload; the archive writers pass exponents 4, 10 */
TRICO_API void trico_compress(uint32_t* payload_bytes, uint8_t** payload, const float* input, const uint32_t count,
                              uint32_t table1_exponent, uint32_t table2_exponent);
/* payload -> float stream */
TRICO_API void trico_decompress(uint32_t* count, float** values, const uint8_t* payload);
/* double stream -> payload; the archive writers pass exponents 20, 20 */
TRICO_API void trico_compress_double_precision(uint32_t* payload_bytes, uint8_t** payload, const double* input, const uint32_t count,
                                               uint64_t table1_exponent, uint64_t table2_exponent);
/* payload -> double stream */
TRICO_API void trico_decompress_double_precision(uint32_t* count, double** values, const uint8_t* payload);

#if defined(__cplusplus)
}
#endif

#endif
