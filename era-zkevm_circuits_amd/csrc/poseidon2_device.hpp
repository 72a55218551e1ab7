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

constexpr int INNER_SHIFT[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};

// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] through the 8-addition chain
__device__ __forceinline__ void m4(uint64_t& x0, uint64_t& x1, uint64_t& x2, uint64_t& x3) {
    uint64_t t0 = gl::add(x0, x1);
    uint64_t t1 = gl::add(x2, x3);
    uint64_t t2 = gl::add(gl::add(x1, x1), t1);
    uint64_t t3 = gl::add(gl::add(x3, x3), t0);
    uint64_t t1_2 = gl::add(t1, t1);
    uint64_t t0_2 = gl::add(t0, t0);
    uint64_t t4 = gl::add(gl::add(t1_2, t1_2), t3);
    uint64_t t5 = gl::add(gl::add(t0_2, t0_2), t2);
    x0 = gl::add(t3, t5);
    x1 = t5;
    x2 = gl::add(t2, t4);
    x3 = t4;
}

__device__ __forceinline__ void mds_external(uint64_t s[12]) {
    m4(s[0], s[1], s[2], s[3]);
    m4(s[4], s[5], s[6], s[7]);
    m4(s[8], s[9], s[10], s[11]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t sum = gl::add(gl::add(s[i], s[4 + i]), s[8 + i]);
        s[i] = gl::add(s[i], sum);
        s[4 + i] = gl::add(s[4 + i], sum);
        s[8 + i] = gl::add(s[8 + i], sum);
    }
}

__device__ __forceinline__ void mds_inner(uint64_t s[12]) {
    uint64_t sum = s[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) sum = gl::add(sum, s[i]);
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl::add(sum, gl::mul_pow2(s[i], INNER_SHIFT[i]));
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
