/* lz4/lz4.h — the three block-API entry points of LZ4 1.9.2 that Trico's callers use next to the archive API
 * (/root/reference/lz4/lz4.h:127-171; called by trico/trico.c:339-514, 1100-1129 and by the reference's own test program,
 * trico.tests/int_compression.cpp:75-187), served by this library's HIP codec: LZ4_compress_default writes, byte for byte, the
 * block LZ4 1.9.2 writes at acceleration 1 (k_lz4.hip / k_lz4_chunked.hip restate lz4.c:793-1181), LZ4_decompress_safe accepts
 * exactly what the safe decoder accepts (k_lz4_decode.hip / k_lz4_pdecode.hip, lz4.c:1657-2072).
 *
 * NOT a copy of the reference's header: only what the path needs is declared.  The streaming, dictionary, HC and frame APIs of
 * LZ4 are not part of Trico's path and are not provided; link the real liblz4 for those (the names below would then clash:
 * define TRICO_NO_LZ4_API when building libtrico to leave them out).
 *
 * Pointers may be host or HIP device pointers.  Without a HIP device the functions fail (0 / negative): there is no CPU path. */
#ifndef TRICO_LZ4_H
#define TRICO_LZ4_H

#include <stddef.h>
#include "../trico/trico_api.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LZ4_VERSION_MAJOR    1      /* the format and the parse are those of 1.9.2 */
#define LZ4_VERSION_MINOR    9
#define LZ4_VERSION_RELEASE  2
#define LZ4_VERSION_NUMBER   (LZ4_VERSION_MAJOR * 100 * 100 + LZ4_VERSION_MINOR * 100 + LZ4_VERSION_RELEASE)

#define LZ4_MAX_INPUT_SIZE        0x7E000000   /* lz4.h:170 */
#define LZ4_COMPRESSBOUND(isize)  ((unsigned)(isize) > (unsigned)LZ4_MAX_INPUT_SIZE ? 0 : (isize) + ((isize) / 255) + 16)

/* worst-case size of the block LZ4_compress_default writes for `inputSize` bytes; 0 beyond LZ4_MAX_INPUT_SIZE (lz4.h:171-179) */
TRICO_API int LZ4_compressBound(int inputSize);

/* `srcSize` bytes at `src` as ONE block into `dst`: bytes written, or 0 if they do not fit `dstCapacity`, the input is larger than
 * LZ4_MAX_INPUT_SIZE, or no device is there (lz4.h:127-142; lz4.c:1271: acceleration 1, fresh state) */
TRICO_API int LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity);

/* the block of exactly `compressedSize` bytes at `src` into `dst`: bytes written (<= dstCapacity), or a negative number for a
 * malformed block or one that needs more room (lz4.h:144-160) */
TRICO_API int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity);

/* Callers of the reference declare a stream state on their stack and initialise it before the one-shot call
 * (trico.c:339-341).  The one-shot functions above keep no state; the type exists with the size and alignment LZ4 1.9.2 gives it
 * (lz4.h:612-616) and LZ4_initStream checks both, as the original does (lz4.c:1408-1420). */
#define LZ4_STREAMSIZE_U64 ((1 << (14 - 3)) + 4)
#define LZ4_STREAMSIZE     (LZ4_STREAMSIZE_U64 * sizeof(unsigned long long))
typedef union LZ4_stream_u { unsigned long long table[LZ4_STREAMSIZE_U64]; } LZ4_stream_t;
TRICO_API LZ4_stream_t* LZ4_initStream(void* buffer, size_t size);

#ifdef __cplusplus
}
#endif

#endif
