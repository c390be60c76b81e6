/* The reference's own radix sort, compiled from its header where it lies (REF_KSORT_H is passed by the test; nothing of
 * the reference is copied into this repository).  Only built when /root/reference is present. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include REF_KSORT_H
typedef struct { uint64_t x, y; } r128_t;
#define r_key_128x(a) ((a).x)
KRADIX_SORT_INIT(r128x, r128_t, r_key_128x, 8)
void ref_sort128(r128_t *a, int n) { radix_sort_r128x(a, a + n); }
