/*
 * oracle/zko_ntt.c — CPU ORACLE (test infrastructure, NOT product code) for K11: Goldilocks NTT / coset LDE.
 *
 * boojum's transforms ([EXT], git dependency absent from /root/reference) are not available; the transform is defined by
 * the field alone (include/zkgl.h, zk_ntt) and this file restates it twice: zko_ntt_naive evaluates the polynomial at every
 * domain point by Horner (the definition; O(N^2), pins the fast version in tests/test_ntt.py) and zko_ntt is the textbook
 * radix-2 Gentleman-Sande / Cooley-Tukey pair the GPU results are compared with at larger sizes.  "parity unpinned" with
 * respect to boojum: its root of unity and output ordering conventions cannot be checked here.
 */
#include "zko.h"

static unsigned bitrev(unsigned x, unsigned bits) {
    unsigned r = 0;
    for (unsigned i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* omega_N = 7^((p-1)/N): 7 generates GF(p)^*, so this has order exactly N = 2^log_n */
uint64_t zko_two_adic_root(unsigned log_n) { return zko_gl_pow(7, (ZKO_P - 1) >> log_n); }

/* definition: out[bitrev(k)] = sum_i a[i] (g w^k)^i */
void zko_ntt_naive(const uint64_t *a, uint64_t *out, unsigned log_n, uint64_t shift) {
    const size_t n = (size_t)1 << log_n;
    const uint64_t w = zko_two_adic_root(log_n);
    uint64_t x = shift;
    for (size_t k = 0; k < n; ++k) {
        uint64_t acc = 0;
        for (size_t i = n; i-- > 0;) acc = zko_gl_add(zko_gl_mul(acc, x), a[i]);
        out[bitrev((unsigned)k, log_n)] = acc;
        x = zko_gl_mul(x, w);
    }
}

/* in place; forward: natural coefficients -> bit-reversed values on shift*<w>; inverse: the inverse map */
void zko_ntt(uint64_t *a, unsigned log_n, int inverse, uint64_t shift) {
    const size_t n = (size_t)1 << log_n;
    if (!inverse) {
        uint64_t s = 1;
        if (shift != 1)
            for (size_t i = 0; i < n; ++i) { a[i] = zko_gl_mul(a[i], s); s = zko_gl_mul(s, shift); }
        for (size_t len = n; len >= 2; len >>= 1) {          /* decimation in frequency: natural -> bit-reversed */
            const size_t half = len >> 1;
            const uint64_t wl = zko_gl_pow(zko_two_adic_root(log_n), n / len);
            for (size_t i = 0; i < n; i += len) {
                uint64_t w = 1;
                for (size_t j = 0; j < half; ++j) {
                    uint64_t u = a[i + j], v = a[i + j + half];
                    a[i + j] = zko_gl_add(u, v);
                    a[i + j + half] = zko_gl_mul(zko_gl_sub(u, v), w);
                    w = zko_gl_mul(w, wl);
                }
            }
        }
    } else {
        const uint64_t winv = zko_gl_inv(zko_two_adic_root(log_n));
        for (size_t len = 2; len <= n; len <<= 1) {          /* decimation in time: bit-reversed -> natural */
            const size_t half = len >> 1;
            const uint64_t wl = zko_gl_pow(winv, n / len);
            for (size_t i = 0; i < n; i += len) {
                uint64_t w = 1;
                for (size_t j = 0; j < half; ++j) {
                    uint64_t u = a[i + j], v = zko_gl_mul(a[i + j + half], w);
                    a[i + j] = zko_gl_add(u, v);
                    a[i + j + half] = zko_gl_sub(u, v);
                    w = zko_gl_mul(w, wl);
                }
            }
        }
        const uint64_t ginv = zko_gl_inv(shift);
        uint64_t s = zko_gl_inv((uint64_t)n % ZKO_P);
        for (size_t i = 0; i < n; ++i) { a[i] = zko_gl_mul(a[i], s); s = zko_gl_mul(s, ginv); }
    }
}

void zko_ntt_batch(uint64_t *a, unsigned log_n, size_t n_polys, size_t stride, int inverse, uint64_t shift) {
#pragma omp parallel for schedule(dynamic)
    for (size_t q = 0; q < n_polys; ++q) zko_ntt(a + q * stride, log_n, inverse, shift);
}

/* out[j * N .. (j+1) * N) = forward transform of coeffs on the coset shift * eta^bitrev(j) * <w_N>, eta = w_{N * 2^log_blowup} */
void zko_lde(const uint64_t *coeffs, uint64_t *out, unsigned log_n, unsigned log_blowup, uint64_t shift) {
    const size_t n = (size_t)1 << log_n, blow = (size_t)1 << log_blowup;
    const uint64_t eta = zko_two_adic_root(log_n + log_blowup);
    for (size_t j = 0; j < blow; ++j) {
        uint64_t *o = out + j * n;
        for (size_t i = 0; i < n; ++i) o[i] = coeffs[i];
        zko_ntt(o, log_n, 0, zko_gl_mul(shift, zko_gl_pow(eta, bitrev((unsigned)j, log_blowup))));
    }
}
