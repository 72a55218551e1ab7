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

/* the textbook definition: the 128-bit product's remainder (a libgcc __umodti3 call).  Kept as the cross-check of
 * zko_gl_mul (tests/test_oracle.py) — it is what the oracle used in round 1 and it made the CPU baseline slow. */
uint64_t zko_gl_mul_slow(uint64_t a, uint64_t b) {
    unsigned __int128 w = (unsigned __int128)a * b;
    return (uint64_t)(w % ZKO_P);
}

/* Standard Goldilocks reduction of a 128-bit product (the one production CPU provers use): with
 * x = lo + 2^64 (hi_lo + 2^32 hi_hi), 2^64 = 2^32 - 1 and 2^96 = -1 (mod p):  x = lo - hi_hi + hi_lo (2^32 - 1). */
uint64_t zko_gl_mul(uint64_t a, uint64_t b) {
    unsigned __int128 w = (unsigned __int128)a * b;
    uint64_t lo = (uint64_t)w, hi = (uint64_t)(w >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & ZKO_EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= ZKO_EPS;         /* borrow: subtract 2^64 = eps (mod p) */
    uint64_t t1 = hi_lo * ZKO_EPS;         /* < 2^64 */
    uint64_t r = t0 + t1;
    if (r < t1) r += ZKO_EPS;              /* carry */
    return r >= ZKO_P ? r - ZKO_P : r;
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
