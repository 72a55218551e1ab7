/* oracle/zko_field.c — CPU ORACLE (test infrastructure).  Goldilocks p = 2^64 - 2^32 + 1.
 * Restates boojum::field::goldilocks [EXT]; canonical u64 representatives only
 * (reference relies on canonical compare: src/ram_permutation/mod.rs:517 `as_u64_reduced`). */
#include "zko.h"

uint64_t zko_gl_reduce(uint64_t a) { return a >= ZKO_P ? a - ZKO_P : a; }

uint64_t zko_gl_add(uint64_t a, uint64_t b) {
    /* a, b canonical */
    uint64_t s = a + b;
    if (s < a) s += ZKO_EPS; /* wrapped: 2^64 == eps (mod p) */
    return zko_gl_reduce(s);
}

uint64_t zko_gl_sub(uint64_t a, uint64_t b) {
    return a >= b ? a - b : a + (ZKO_P - b);
}

uint64_t zko_gl_mul(uint64_t a, uint64_t b) {
    unsigned __int128 w = (unsigned __int128)a * b;
    /* independent of the GPU's 96-bit folding: plain 128-bit remainder */
    return (uint64_t)(w % ZKO_P);
}

uint64_t zko_gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = zko_gl_mul(r, a);
        a = zko_gl_mul(a, a);
        e >>= 1;
    }
    return r;
}

uint64_t zko_gl_inv(uint64_t a) { return a == 0 ? 0 : zko_gl_pow(a, ZKO_P - 2); }

void zko_gl_fma_cols(uint64_t *dst, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                     uint64_t q, uint64_t l, size_t n) {
    for (size_t i = 0; i < n; ++i)
        dst[i] = zko_gl_add(zko_gl_mul(q, zko_gl_mul(a[i], b[i])), zko_gl_mul(l, c[i]));
}
