/* lz4_api.c — LZ4_compressBound / LZ4_compress_default / LZ4_decompress_safe / LZ4_initStream of LZ4 1.9.2 (the entry points
 * Trico's callers use beside the archive API: /root/reference/lz4/lz4.h:127-171, trico.c:339-514, 1100-1129,
 * trico.tests/int_compression.cpp:75-187) on this library's HIP codec.  Thin host glue over the shim: one byte plane of width 1 is
 * exactly one LZ4 block (trico_hip_int_encode / trico_hip_int_decode); nothing is computed on the host. */
#ifndef TRICO_NO_LZ4_API
#include "lz4/lz4.h"
#include "trico/trico_hip.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int LZ4_compressBound(int inputSize)
  {
  return LZ4_COMPRESSBOUND(inputSize);             /* (a negative size is larger than LZ4_MAX_INPUT_SIZE as unsigned: 0, as in lz4.c:651) */
  }

LZ4_stream_t* LZ4_initStream(void* buffer, size_t size)
  {
  /* lz4.c:1408-1420: NULL for a buffer that is too small or not aligned for the state */
  if (!buffer || size < sizeof(LZ4_stream_t) || ((uintptr_t)buffer & (sizeof(unsigned long long) - 1u)) != 0)
    return NULL;
  memset(buffer, 0, sizeof(LZ4_stream_t));
  return (LZ4_stream_t*)buffer;
  }

int LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity)
  {
  /* lz4.c:1271-1290 -> 1184-1262 -> 793-1181.  The output-limited variant of the reference gives up exactly when the finished
   * block would not fit: every check it makes on the way (lz4.c:975-980, 1057-1062, 1153-1160) leaves room for what the block
   * still has to hold at that point (the last five literals and their token), so "fits" is decided by the final size. */
  if (srcSize < 0 || (unsigned)srcSize > (unsigned)LZ4_MAX_INPUT_SIZE || dstCapacity <= 0 || !dst || (srcSize > 0 && !src))
    return 0;
  trico_hip_ctx* ctx = trico_hip_ctx_create();
  if (!ctx)
    return 0;
  uint32_t sizes[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  int written = 0;
  if (trico_hip_int_encode(ctx, src, (uint32_t)srcSize, 1, sizes) && sizes[0] > 0 && sizes[0] <= (uint32_t)dstCapacity &&
      trico_hip_fetch_payload(ctx, 0, dst))
    written = (int)sizes[0];
  trico_hip_ctx_destroy(ctx);
  return written;
  }

int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity)
  {
  if (!src || compressedSize <= 0 || dstCapacity < 0 || (dstCapacity > 0 && !dst) || (unsigned)dstCapacity > (unsigned)LZ4_MAX_INPUT_SIZE)
    return -1;
  trico_hip_ctx* ctx = trico_hip_ctx_create();
  if (!ctx)
    return -1;
  const uint8_t* pay[8] = { (const uint8_t*)src, 0, 0, 0, 0, 0, 0, 0 };
  const uint32_t sizes[8] = { (uint32_t)compressedSize, 0, 0, 0, 0, 0, 0, 0 };
  int result = -1;
  /* the caller of the archive format knows the exact size (trico.c:1100-1129): try that first; a block that decodes to fewer bytes is
   * measured on the device and decoded at its true size */
  if (dstCapacity == 0)
    {
    if (compressedSize == 1 && trico_hip_pointer_is_device(src) == 0 && src[0] == 0)
      result = 0;                                   /* the empty block: one zero token (lz4.c:1683) */
    else
      {
      uint32_t real = 0;
      if (trico_hip_lz4_decoded_size(ctx, src, (uint32_t)compressedSize, 0u, &real))
        result = 0;
      }
    }
  else if (trico_hip_int_decode(ctx, pay, sizes, 1, (uint32_t)dstCapacity, dst))
    result = dstCapacity;
  else
    {
    uint32_t real = 0;
    if (trico_hip_lz4_decoded_size(ctx, src, (uint32_t)compressedSize, (uint32_t)dstCapacity, &real) && real < (uint32_t)dstCapacity)
      {
      if (real == 0)
        result = 0;
      else if (trico_hip_int_decode(ctx, pay, sizes, 1, real, dst))
        result = (int)real;
      }
    }
  trico_hip_ctx_destroy(ctx);
  return result;
  }
#endif
