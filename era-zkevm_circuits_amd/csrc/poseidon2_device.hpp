// poseidon2_device.hpp — Poseidon2 (Goldilocks, t = 12, rate 8, capacity 4) for gfx950.
//
// The round function every circuit of the reference is generic over
// (`R: CircuitRoundFunction<F, 8, 12, 4>`, /root/reference/src/utils.rs:15) and that the
// tests instantiate as boojum's `Poseidon2Goldilocks` (src/ram_permutation/mod.rs:411).
// Structure (boojum [EXT]):  M_E ; 4 full ; 22 partial ; 4 full, x^7
// S-box, M_E = circ(2*M4, M4, M4), M_I = J + diag(2^k).
//
// One lane owns one permutation; the 12-element state lives in 24 VGPRs, round constants come
// from constant memory through scalar loads (the round index is wave-uniform).  No MFMA: this
// is u64 modular arithmetic on the 32-bit integer pipes.
#pragma once
#include "gl_device.hpp"

namespace p2 {

// 360 Poseidon-Goldilocks round constants; filled by zk_init() from the host-side derivation.
// All device code is one translation unit (zkgl_device.hip), so the symbol is defined here.
__constant__ uint64_t RC[360];

// Inverses of the small field elements: INV_SMALL[k] = k^-1 mod p (0 for k = 0), filled by zk_init() beside the round constants.
// The is_zero gadget's witness is x^-1, a 72-multiplication dependent chain when computed as x^(p-2); the zero-checks of the FSM
// circuits compare buffer positions, counters and flags — x = d or p - d with d small — and there the inverse is one gather:
// inv(p - d) = p - inv(d).  (keccak256_round_function: 1 690 zero-checks per cycle, one on the critical path of nearly every
// dependency level of the strand kernel: 23.6 -> 18.6 ms with the inversions stubbed, profiles/r3_summary.md.)
constexpr uint32_t INV_SMALL_N = 4096;
__device__ uint64_t INV_SMALL[INV_SMALL_N];
// x^-1 (0 -> 0): the table when every lane of the wavefront holds a small |x|, the addition chain otherwise — same value either way
__device__ __forceinline__ uint64_t inv_wave(uint64_t x) {
    const uint64_t nx = gl::P - x;
    const bool pos = x < INV_SMALL_N, neg = nx < INV_SMALL_N;
    if (__builtin_amdgcn_ballot_w64(!(pos || neg)) == 0) {
        const uint64_t r = INV_SMALL[pos ? (uint32_t)x : (uint32_t)nx];
        return pos ? r : gl::P - r;   // neg: x != 0 and x != p, so r != 0
    }
    return gl::inv(x);
}

constexpr int INNER_SHIFT[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};

// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] through the 8-addition chain; M_E = circ(2*M4, M4, M4); M_I = J + diag(2^k).
// Lazily accumulated variants: sums are carried as 96-bit integers (lo, hi = number of 2^64 wraps) with 3-instruction
// additions and reduced once per output, instead of a canonicalising 8-instruction gl::add per term.
struct W { uint64_t lo; uint32_t hi; };
__device__ __forceinline__ W wadd(W a, W b) { W r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u); return r; }
__device__ __forceinline__ void m4w(W& x0, W& x1, W& x2, W& x3) {
    W t0 = wadd(x0, x1), t1 = wadd(x2, x3);
    W t2 = wadd(wadd(x1, x1), t1), t3 = wadd(wadd(x3, x3), t0);
    W t1_2 = wadd(t1, t1), t0_2 = wadd(t0, t0);
    W t4 = wadd(wadd(t1_2, t1_2), t3), t5 = wadd(wadd(t0_2, t0_2), t2);
    x0 = wadd(t3, t5); x1 = t5; x2 = wadd(t2, t4); x3 = t4;
}
__device__ __forceinline__ void mds_external(uint64_t s[12]) {
    W w[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { w[i].lo = s[i]; w[i].hi = 0; }
    m4w(w[0], w[1], w[2], w[3]);
    m4w(w[4], w[5], w[6], w[7]);
    m4w(w[8], w[9], w[10], w[11]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        W sum = wadd(wadd(w[i], w[4 + i]), w[8 + i]);
        W a = wadd(w[i], sum), b = wadd(w[4 + i], sum), c = wadd(w[8 + i], sum);
        s[i] = gl::reduce96(a.lo, a.hi); s[4 + i] = gl::reduce96(b.lo, b.hi); s[8 + i] = gl::reduce96(c.lo, c.hi);
    }
}
__device__ __forceinline__ void mds_inner(uint64_t s[12]) {
    W sum{s[0], 0};
#pragma unroll
    for (int i = 1; i < 12; ++i) { sum.lo += s[i]; sum.hi += (sum.lo < s[i] ? 1u : 0u); }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int k = INNER_SHIFT[i];
        W t{s[i] << k, k ? (uint32_t)(s[i] >> (64 - k)) : 0u};
        W r = wadd(sum, t);
        s[i] = gl::reduce96(r.lo, r.hi);
    }
}
__device__ __forceinline__ void full_round(uint64_t s[12], int r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl::pow7(gl::add(s[i], RC[12 * r + i]));
    mds_external(s);
}

__device__ __forceinline__ void partial_round(uint64_t s[12], int r) {
    s[0] = gl::pow7(gl::add(s[0], RC[12 * r]));
    mds_inner(s);
}

__device__ __forceinline__ void permute(uint64_t s[12]) {
    mds_external(s);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) full_round(s, r);
#pragma unroll 1
    for (int r = 4; r < 26; ++r) partial_round(s, r);
#pragma unroll 1
    for (int r = 26; r < 30; ++r) full_round(s, r);
}

}  // namespace p2
