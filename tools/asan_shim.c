/* Test infrastructure for tools/asan_check.sh: clang's AddressSanitizer instrumentation (ROCm 7.2) references three runtime
 * helpers that the GCC 11 libasan preloaded as the runtime does not export; they are plain memory routines. */
#include <string.h>
void* __sanitizer_internal_memcpy(void* d, const void* s, size_t n) { return memcpy(d, s, n); }
void* __sanitizer_internal_memmove(void* d, const void* s, size_t n) { return memmove(d, s, n); }
void* __sanitizer_internal_memset(void* d, int c, size_t n) { return memset(d, c, n); }
