// keccak_macro.hpp — the STRUCTURE of Keccak-f[1600] over 8-bit lookup tables, written once and instantiated three ways:
//   * the host gadget (circuits/keccak_gadget.hpp): Lane = 8 byte variables; every primitive records its lookups / reduction gates;
//   * the device macro-op ZK_OP_KECCAK_F (kernels_engine2.hpp / kernels_engine.hpp): Lane = uint64; every primitive computes its
//     result in registers and STREAMS OUT the same intermediates in the same order (kernel K8: the witness of a whole permutation is
//     one op — 200 operand loads, then ~30 k stores — instead of ~30 k interpreted lookups / linear combinations over 650 dependency
//     levels);
//   * a counting backend (how many outputs the macro-op has).
// Because all three walk THIS function, the order of the macro-op's outputs is the gadget's allocation order by construction.
// Reference surface: keccak256_absorb_and_run_permutation, /root/reference/src/keccak256_round_function/mod.rs:796-838 (boojum's own
// decomposition is [EXT]; this one is the engine's, over Xor8 / AndN8 / ByteSplit<k>).
//
// Primitives a backend provides (outputs = values the trace holds, in this order):
//   xor_lane(a, b)   -> 8 outputs: byte k of a ^ b, k = 0..7                                   (8 Xor8 lookups)
//   andn_lane(a, b)  -> 8 outputs: byte k of ~a & b                                             (8 AndN8 lookups)
//   rotl(a, n)       -> n % 8 == 0: no output (a byte permutation); else with b = n % 8: 16 outputs (lo_k, hi_k) = byte k split at
//                       8 - b bits, k = 0..7 (ByteSplit<8-b> lookups), then 8 outputs byte k of rotl64(a, b) = lo_k 2^b + hi_(k-1)
//                       (reduction gates); returns rotl64(a, n)
//   xor_const(a, c)  -> one output per non-zero byte of c: byte k of a ^ c                     (Xor8 lookups against constants)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZKK_HD __host__ __device__ __forceinline__
#else
#define ZKK_HD inline
#endif
// Device code walks the structure ROLLED: the 25 lanes of the state are indexed dynamically and therefore live in scratch (200 B per
// lane of the wavefront, L1 / L2 resident), which keeps the macro-op at ~30 VGPRs inside the interpreter.  Fully unrolled with the
// state in registers it needs > 128 VGPRs and made the strand kernel spill 2.2 KB per lane (-Rpass-analysis=kernel-resource-usage).
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKK_LOOP _Pragma("unroll 1")
#else
#define ZKK_LOOP
#endif

namespace zkk {

// iota constants
#if defined(__HIP_DEVICE_COMPILE__)
__constant__
#endif
static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// rho offsets and pi lane order of the classic in-place chain: t = s[1]; for i: j = PI[i]; bc = s[j]; s[j] = rotl(t, RHO[i]); t = bc
ZKK_HD int rho(int i) {
    constexpr int R[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    return R[i];
}
ZKK_HD int pi(int i) {
    constexpr int P[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    return P[i];
}

// one body, two loop policies: LOOP = the pragma in front of every inner loop
#define ZKK_KECCAK_F_BODY(LOOP)                                                                                          \
    typedef typename B::Lane Lane;                                                                                        \
    ZKK_LOOP                                                                                                              \
    for (int rnd = 0; rnd < 24; ++rnd) {                                                                                  \
        Lane c[5], d[5];                                                                                                  \
        LOOP for (int x = 0; x < 5; ++x) {                                                                                \
            c[x] = be.xor_lane(s[x], s[x + 5]);                                                                           \
            c[x] = be.xor_lane(c[x], s[x + 10]);                                                                          \
            c[x] = be.xor_lane(c[x], s[x + 15]);                                                                          \
            c[x] = be.xor_lane(c[x], s[x + 20]);                                                                          \
        }                                                                                                                 \
        LOOP for (int x = 0; x < 5; ++x) d[x] = be.xor_lane(c[(x + 4) % 5], be.rotl(c[(x + 1) % 5], 1));                  \
        LOOP for (int i = 0; i < 25; ++i) s[i] = be.xor_lane(s[i], d[i % 5]);                                             \
        Lane t = s[1];                                                                                                    \
        LOOP for (int i = 0; i < 24; ++i) {                                                                               \
            const int j = pi(i);                                                                                          \
            const Lane bc = s[j];                                                                                         \
            s[j] = be.rotl(t, rho(i));                                                                                    \
            t = bc;                                                                                                       \
        }                                                                                                                 \
        LOOP for (int y = 0; y < 5; ++y) {                                                                                \
            Lane r[5];                                                                                                    \
            LOOP for (int x = 0; x < 5; ++x) r[x] = s[x + 5 * y];                                                         \
            LOOP for (int x = 0; x < 5; ++x) s[x + 5 * y] = be.xor_lane(r[x], be.andn_lane(r[(x + 1) % 5], r[(x + 2) % 5])); \
        }                                                                                                                 \
        s[0] = be.xor_const(s[0], rc[rnd]);                                                                               \
    }

// rolled on the device (state in scratch), plain loops on the host
template <class B>
ZKK_HD void keccak_f(B& be, typename B::Lane s[25], const uint64_t rc[24]) {
    ZKK_KECCAK_F_BODY(ZKK_LOOP)
}
// inner loops unrolled: every lane index is static, the state stays in registers (the round loop stays rolled on the device)
template <class B>
ZKK_HD void keccak_f_unrolled(B& be, typename B::Lane s[25], const uint64_t rc[24]) {
    ZKK_KECCAK_F_BODY(_Pragma("unroll"))
}

// counting backend: the number of outputs of the macro-op
struct CountBackend {
    typedef int Lane;
    uint32_t n = 0;
    Lane xor_lane(Lane, Lane) { n += 8; return 0; }
    Lane andn_lane(Lane, Lane) { n += 8; return 0; }
    Lane rotl(Lane, int r) { if (r % 8) n += 24; return 0; }
    Lane xor_const(Lane, uint64_t c) { for (int k = 0; k < 8; ++k) if ((c >> (8 * k)) & 0xff) ++n; return 0; }
};

// compute backend over uint64 lanes.  `Emit` receives the outputs in order: block8(v) = the eight bytes of v (k = 0..7), one(v) = one value
template <class Emit>
struct ComputeBackend {
    typedef uint64_t Lane;
    Emit& emit;
    ZKK_HD explicit ComputeBackend(Emit& e) : emit(e) {}
    ZKK_HD Lane xor_lane(Lane a, Lane b) { const Lane r = a ^ b; emit.block8(r); return r; }
    ZKK_HD Lane andn_lane(Lane a, Lane b) { const Lane r = ~a & b; emit.block8(r); return r; }
    ZKK_HD Lane rotl(Lane a, int n) {
        n %= 64;
        const int b = n % 8;
        const Lane full = n ? (a << n) | (a >> ((64 - n) & 63)) : a;
        if (b == 0) return full;
        // 16 outputs (lo_k, hi_k), k = 0..7: as two blocks of eight — bytes 0..3 then 4..7 of the lane, lo / hi interleaved
        const uint64_t lo_mask = 0x0101010101010101ull * ((1u << (8 - b)) - 1);
        const uint64_t lo = a & lo_mask, hi = (a >> (8 - b)) & (0x0101010101010101ull * ((1u << b) - 1));
        emit.block8(interleave(lo, hi));
        emit.block8(interleave(lo >> 32, hi >> 32));
        emit.block8((a << b) | (a >> (64 - b)));        // byte k = lo_k 2^b + hi_(k-1)
        return full;
    }
    // bytes 0..3 of lo and hi -> lo0, hi0, lo1, hi1, lo2, hi2, lo3, hi3
    ZKK_HD static uint64_t interleave(uint64_t lo, uint64_t hi) {
        uint64_t r = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) r |= (((lo >> (8 * k)) & 0xff) << (16 * k)) | (((hi >> (8 * k)) & 0xff) << (16 * k + 8));
        return r;
    }
    ZKK_HD Lane xor_const(Lane a, uint64_t c) {
        const Lane r = a ^ c;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((c >> (8 * k)) & 0xff) emit.one((r >> (8 * k)) & 0xff);
        return r;
    }
};

}  // namespace zkk
