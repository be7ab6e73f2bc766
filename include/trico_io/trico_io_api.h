/* trico_io_api.h — visibility macro of libtrico_io.so (mirrors trico_io/trico_io_api.h:1-15 of the reference). */
#ifndef TRICO_IO_TRICO_IO_API
#define TRICO_IO_TRICO_IO_API

#if defined(_WIN32)
#  define TRICO_IO_API
#else
#  define TRICO_IO_API __attribute__((visibility("default")))
#endif

#endif
