/* tools/asan_check.sh canary: a deliberate heap overflow in a clang-instrumented shared object, to prove that the preloaded
 * runtime really reports errors in this configuration (otherwise "0 reports" would mean nothing). */
#include <stdlib.h>
int q1_asan_canary(int n) {
    volatile char* p = (volatile char*)malloc(16);
    p[16 + (n & 1)] = 1;          /* one or two bytes past the end */
    int r = p[0];
    free((void*)p);
    return r;
}
