// gl_device.hpp — Goldilocks field (p = 2^64 - 2^32 + 1) for gfx950 device code.
//
// Replaces boojum::field::goldilocks::GoldilocksField as used by every circuit of the
// reference (`type F = GoldilocksField`, /root/reference/src/ram_permutation/mod.rs:414).
// CDNA4 has no 64x64->128 multiplier: a field multiply is a 32-bit partial-product tree
// (v_mad_u64_u32 / v_mul_hi_u32) followed by the 2^64 == 2^32-1, 2^96 == -1 folding.
// All values are kept canonical (< p) in memory so that bit-exact comparison is plain
// u64 equality; lazy (non-canonical) forms exist only inside registers where noted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ull;
constexpr uint64_t EPS = 0xFFFFFFFFull;

__host__ __device__ __forceinline__ uint64_t reduce(uint64_t a) { return a >= P ? a - P : a; }

__host__ __device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    // a wrap means the true sum is s + 2^64 == s + EPS (mod p) and that is already < p
    if (s < a) return s + EPS;
    return s >= P ? s - P : s;
}

__host__ __device__ __forceinline__ uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    return a >= b ? d : d - EPS;  // d + p (mod 2^64)
}

__host__ __device__ __forceinline__ uint64_t neg(uint64_t a) { return a ? P - a : 0; }

// 128 -> 64 reduction of hi*2^64 + lo; result canonical.
__host__ __device__ __forceinline__ uint64_t reduce128(uint64_t lo, uint64_t hi) {
    uint64_t hi_hi = hi >> 32, hi_lo = hi & EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;
    uint64_t t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1)
    uint64_t t2 = t0 + t1;
    if (t2 < t1) t2 += EPS;
    return t2 >= P ? t2 - P : t2;
}

__device__ __forceinline__ void mul_wide(uint64_t a, uint64_t b, uint64_t& lo, uint64_t& hi) {
    // four v_mad_u64_u32 (32x32 + 64 -> 64) instead of a 64-bit multiply plus a separate __umul64hi
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)a0 * b0;
    const uint64_t p01 = (uint64_t)a0 * b1 + (p00 >> 32);
    const uint64_t p10 = (uint64_t)a1 * b0 + (uint32_t)p01;
    hi = (uint64_t)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    lo = ((uint64_t)(uint32_t)p10 << 32) | (uint32_t)p00;
}

// lo + hi * 2^64 (hi < 2^31) -> canonical.  Used by the lazily accumulated MDS layers.
__device__ __forceinline__ uint64_t reduce96(uint64_t lo, uint32_t hi) {
    const uint64_t t = ((uint64_t)hi << 32) - hi;  // hi * (2^32 - 1) == hi * 2^64 (mod p)
    uint64_t r = lo + t;
    if (r < t) r += EPS;
    return r >= P ? r - P : r;
}

__device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) {
    uint64_t lo, hi;
    mul_wide(a, b, lo, hi);
    return reduce128(lo, hi);
}

__device__ __forceinline__ uint64_t sqr(uint64_t a) { return mul(a, a); }

// x * 2^k for k < 32 (Poseidon2 inner diagonal, encoding shifts)
__device__ __forceinline__ uint64_t mul_pow2(uint64_t a, unsigned k) {
    uint64_t lo = a << k;
    uint64_t hi = k ? (a >> (64 - k)) : 0;
    return reduce128(lo, hi);
}

// a*b + c
__device__ __forceinline__ uint64_t fma(uint64_t a, uint64_t b, uint64_t c) {
    uint64_t lo, hi;
    mul_wide(a, b, lo, hi);
    uint64_t l2 = lo + c;
    hi += (l2 < lo);  // a,b < 2^64 => hi <= 2^64-2, no overflow
    return reduce128(l2, hi);
}

__device__ __forceinline__ uint64_t pow7(uint64_t x) {
    uint64_t x2 = sqr(x), x3 = mul(x2, x), x4 = sqr(x2);
    return mul(x3, x4);
}

// x^(p-2); 0 -> 0.  Addition chain for p-2 = 2^64 - 2^32 - 1:
// exponent bits: 32 ones, one zero, 31 ones  => x^(2^32-1) shifted by 32, times x^(2^31-1)... computed
// with a sliding ladder of x^(2^k - 1).
__device__ __forceinline__ uint64_t inv(uint64_t x) {
    // e1 = x^(2^1-1) ... build x^(2^k-1) for k = 2,3,6,12,24,30,31,32
    uint64_t e1 = x;
    uint64_t e2 = mul(sqr(e1), e1);                       // 2^2-1
    uint64_t e3 = mul(sqr(e2), e1);                       // 2^3-1
    uint64_t t = e3;
#pragma unroll 1
    for (int i = 0; i < 3; ++i) t = sqr(t);
    uint64_t e6 = mul(t, e3);                             // 2^6-1
    t = e6;
#pragma unroll 1
    for (int i = 0; i < 6; ++i) t = sqr(t);
    uint64_t e12 = mul(t, e6);                            // 2^12-1
    t = e12;
#pragma unroll 1
    for (int i = 0; i < 12; ++i) t = sqr(t);
    uint64_t e24 = mul(t, e12);                           // 2^24-1
    t = e24;
#pragma unroll 1
    for (int i = 0; i < 6; ++i) t = sqr(t);
    uint64_t e30 = mul(t, e6);                            // 2^30-1
    uint64_t e31 = mul(sqr(e30), e1);                     // 2^31-1
    uint64_t e32 = mul(sqr(e31), e1);                     // 2^32-1
    // p-2 = (2^32-1)*2^32 - 1... write p-2 = 2^64 - 2^32 - 1 = (2^32 - 2)*2^32 + (2^32 - 1)
    // (2^32-2) = 2*(2^31-1)  => x^(p-2) = (x^(2^31-1))^(2^33) * x^(2^32-1)
    t = e31;
#pragma unroll 1
    for (int i = 0; i < 33; ++i) t = sqr(t);
    return mul(t, e32);
}

// ZK_OP_U8X4FMA (include/zkgl_ir.h): a*b + c + d over u32 operands given as bytes -> the eight bytes of the result and the two bytes
// of k = the carry of the low 32 bits of the byte-product sum.  Plain wrapping u64 arithmetic (the oracle restates it word for
// word): for operands that are not bytes the outputs are still defined, the gate rejects them.
__host__ __device__ __forceinline__ void u8x4_fma(const uint64_t in[16], uint64_t out[10]) {
    const uint64_t a = in[0] + (in[1] << 8) + (in[2] << 16) + (in[3] << 24), b = in[4] + (in[5] << 8) + (in[6] << 16) + (in[7] << 24);
    const uint64_t c = in[8] + (in[9] << 8) + (in[10] << 16) + (in[11] << 24), d = in[12] + (in[13] << 8) + (in[14] << 16) + (in[15] << 24);
    const uint64_t r = a * b + c + d;
    const uint64_t t = in[0] * b + ((in[1] * (b & 0xffffffull)) << 8) + ((in[2] * (b & 0xffffull)) << 16) + ((in[3] * (b & 0xffull)) << 24);
    const uint64_t k = (t + c + d) >> 32;
    for (int i = 0; i < 8; ++i) out[i] = (r >> (8 * i)) & 0xff;
    out[8] = k & 0xff;
    out[9] = (k >> 8) & 0xff;
}

// ZK_GATE_U8X4_FMA: the two relations over the 26 variables a0..3, b0..3, c0..3, d0..3, lo0..3, hi0..3, k0, k1, evaluated in the field
// (both are zero for a satisfied gate)
__device__ __forceinline__ uint64_t le4(const uint64_t* x) {   // x0 + 2^8 x1 + 2^16 x2 + 2^24 x3
    return add(add(x[0], mul_pow2(x[1], 8)), add(mul_pow2(x[2], 16), mul_pow2(x[3], 24)));
}
__device__ __forceinline__ void u8x4_relations(const uint64_t v[26], uint64_t& r0, uint64_t& r1) {
    const uint64_t* a = v; const uint64_t* b = v + 4;
    const uint64_t B1 = b[0], B2 = add(B1, mul_pow2(b[1], 8)), B3 = add(B2, mul_pow2(b[2], 16)), B4 = add(B3, mul_pow2(b[3], 24));
    const uint64_t H3 = b[3], H2 = add(b[2], mul_pow2(b[3], 8)), H1 = add(b[1], mul_pow2(H2, 8));
    const uint64_t T = add(add(mul(a[0], B4), mul_pow2(mul(a[1], B3), 8)), add(mul_pow2(mul(a[2], B2), 16), mul_pow2(mul(a[3], B1), 24)));
    const uint64_t K = add(v[24], mul_pow2(v[25], 8));
    r0 = sub(add(add(T, le4(v + 8)), le4(v + 12)), add(le4(v + 16), mul(K, 1ull << 32)));
    const uint64_t Y = add(add(mul(a[1], H3), mul(a[2], H2)), mul(a[3], H1));
    r1 = sub(add(Y, K), le4(v + 20));
}

}  // namespace gl
