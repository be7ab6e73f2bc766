/*
 * trico_api.h — the export macro of the public headers.
 *
 * Counterpart of the reference's trico/trico_api.h:4-14 (TRICO_API: dllexport / dllimport on Windows, empty elsewhere).
 * This library is an ELF shared object built with -fvisibility=hidden, so the macro marks the exported functions
 * with default visibility; a consumer that includes the headers sees plain declarations either way.  Defining
 * TRICO_API before including any header overrides it, exactly as with the reference's header.
 */
#ifndef TRICO_TRICO_API
#define TRICO_TRICO_API

#ifndef TRICO_API
#  if defined(_WIN32)
#    if defined(TRICO_DLL_EXPORT)
#      define TRICO_API __declspec(dllexport)
#    elif defined(TRICO_DLL_IMPORT)
#      define TRICO_API __declspec(dllimport)
#    else
#      define TRICO_API
#    endif
#  elif defined(__GNUC__)
#    define TRICO_API __attribute__((visibility("default")))
#  else
#    define TRICO_API
#  endif
#endif

#endif /* TRICO_TRICO_API */
