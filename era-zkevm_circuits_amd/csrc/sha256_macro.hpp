// sha256_macro.hpp — the STRUCTURE of the SHA-256 compression function over 8-bit lookup tables, written once and walked by
//   * the host gadget (circuits/sha256_gadget.hpp): Word = 4 byte variables, every primitive records its lookups / reduction gates;
//   * the device macro-op ZK_OP_SHA256_ROUNDS (kernels_engine2.hpp): Word = uint32, every primitive computes in registers and STREAMS
//     OUT the same intermediates in the same order (kernel K8);
//   * a counting backend (the number of outputs).
// Reference surface: round_function_over_uint32, /root/reference/src/sha256_round_function/mod.rs:271-285 (boojum's own 4-bit
// decomposition is [EXT]; circuits/sha256_gadget4.hpp rebuilds it over the reference's table set, this file is the engine's 8-bit one).
//
// Backend primitives and their outputs (values the trace holds, in this order):
//   bytewise(T, a, b)     -> 4: byte k of a (op) b, op = xor / and / andn (~a & b)                           (4 lookups)
//   split4(a, at)         -> 8: (lo_k, hi_k) = byte k split at `at` bits, k = 0..3                           (4 ByteSplit<at> lookups)
//   rotr(a, n)            -> n % 8 == 0: none; else split4(a, n % 8) then 4: byte k of rotr32(a, n)         (reduction gates)
//   shr(a, n)             -> n % 8 == 0: none; else split4(a, n % 8) then one per k with k + n / 8 + 1 < 4: byte k of a >> n
//   add_mod32<N>(w, c)    -> the partial sums of the linear-combination chain over the 4 N byte terms (+ the constant term): one after the
//                            first four terms, one per further three; then 5: the four bytes and the carry of the sum; then 2: the low
//                            word, the whole sum; then 1: the range-check lookup of the carry (carry ^ carry = 0)
//   range_check_word(w)   -> 2: w0 ^ w1, w2 ^ w3                                                               (2 Xor8 lookups)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZKS_HD __host__ __device__ __forceinline__
#else
#define ZKS_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKS_LOOP _Pragma("unroll 1")
#else
#define ZKS_LOOP
#endif

namespace zks {

enum Table : int { T_XOR = 0, T_AND = 1, T_ANDN = 2 };

#if defined(__HIP_DEVICE_COMPILE__)
__constant__
#endif
static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// st[8] <- compress(st, block[16]); w: 64 words of working storage the caller provides (device: scratch)
template <class B>
ZKS_HD void compress(B& be, typename B::Word st[8], const typename B::Word block[16], typename B::Word w[64], const uint32_t k[64]) {
    typedef typename B::Word Word;
    ZKS_LOOP
    for (int i = 0; i < 16; ++i) w[i] = block[i];
    ZKS_LOOP
    for (int i = 16; i < 64; ++i) {
        // (every call in its own statement: the order of the outputs must not depend on the order a compiler evaluates arguments in)
        const Word w15 = w[i - 15], w2 = w[i - 2];
        const Word a7 = be.rotr(w15, 7);
        const Word a18 = be.rotr(w15, 18);
        const Word a3 = be.shr(w15, 3);
        const Word s0 = be.xor3(a7, a18, a3);
        const Word b17 = be.rotr(w2, 17);
        const Word b19 = be.rotr(w2, 19);
        const Word b10 = be.shr(w2, 10);
        const Word s1 = be.xor3(b17, b19, b10);
        const Word t[4] = {w[i - 16], s0, w[i - 7], s1};
        w[i] = be.template add_mod32<4>(t, 0);
    }
    be.range_check_word(w[62]);   // the only schedule words no sigma lookup consumes
    be.range_check_word(w[63]);
    Word a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    ZKS_LOOP
    for (int i = 0; i < 64; ++i) {
        const Word e6 = be.rotr(e, 6);
        const Word e11 = be.rotr(e, 11);
        const Word e25 = be.rotr(e, 25);
        const Word S1 = be.xor3(e6, e11, e25);
        const Word ef = be.bytewise(T_AND, e, f);
        const Word neg = be.bytewise(T_ANDN, e, g);
        const Word ch = be.bytewise(T_XOR, ef, neg);
        const Word a2 = be.rotr(a, 2);
        const Word a13 = be.rotr(a, 13);
        const Word a22 = be.rotr(a, 22);
        const Word S0 = be.xor3(a2, a13, a22);
        const Word ab = be.bytewise(T_AND, a, b);
        const Word axb = be.bytewise(T_XOR, a, b);
        const Word cx = be.bytewise(T_AND, c, axb);
        const Word maj = be.bytewise(T_XOR, ab, cx);
        const Word wi = w[i];
        const Word t1[5] = {d, h, S1, ch, wi};
        const Word new_e = be.template add_mod32<5>(t1, k[i]);
        const Word t2[6] = {h, S1, ch, wi, S0, maj};
        const Word new_a = be.template add_mod32<6>(t2, k[i]);
        h = g; g = f; f = e; e = new_e; d = c; c = b; b = a; a = new_a;
    }
    be.range_check_word(a);   // outputs of the last round feed additions only
    be.range_check_word(e);
    const Word out[8] = {a, b, c, d, e, f, g, h};
    ZKS_LOOP
    for (int i = 0; i < 8; ++i) {
        const Word t[2] = {st[i], out[i]};
        st[i] = be.template add_mod32<2>(t, 0);
        be.range_check_word(st[i]);
    }
}

struct CountBackend {
    typedef int Word;
    uint32_t n = 0;
    Word bytewise(int, Word, Word) { n += 4; return 0; }
    Word xor3(Word, Word, Word) { n += 8; return 0; }
    Word rotr(Word, int r) { if (r % 8) n += 12; return 0; }
    Word shr(Word, int r) {
        const int q = r / 8, b = r % 8;
        if (b) { n += 8; for (int k = 0; k < 4; ++k) if (k + q + 1 < 4) ++n; }
        return 0;
    }
    template <int N> Word add_mod32(const Word*, uint64_t c) {
        const int T = 4 * N + (c ? 1 : 0);
        n += 1 + (T > 4 ? (T - 4 + 2) / 3 : 0);   // the chain of the sum
        n += 5 + 2 + 1;
        return 0;
    }
    void range_check_word(Word) { n += 2; }
};

// compute backend over uint32 words.  Emit: block(vals, n) = n consecutive outputs (n <= 8), shared among the strands block by block
template <class Emit>
struct ComputeBackend {
    typedef uint32_t Word;
    Emit& emit;
    ZKS_HD explicit ComputeBackend(Emit& e) : emit(e) {}
    ZKS_HD void emit4(uint32_t v) {
        const uint64_t o[4] = {v & 0xffu, (v >> 8) & 0xffu, (v >> 16) & 0xffu, v >> 24};
        emit.block(o, 4);
    }
    ZKS_HD Word bytewise(int t, Word a, Word b) {
        const Word r = t == T_XOR ? a ^ b : t == T_AND ? a & b : ~a & b;
        emit4(r);
        return r;
    }
    ZKS_HD Word xor3(Word a, Word b, Word c) { const Word ab = bytewise(T_XOR, a, b); return bytewise(T_XOR, ab, c); }
    ZKS_HD void split4(Word a, int at) {
        uint64_t o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t byte = (a >> (8 * k)) & 0xff;
            o[2 * k] = byte & ((1u << at) - 1);
            o[2 * k + 1] = byte >> at;
        }
        emit.block(o, 8);
    }
    ZKS_HD Word rotr(Word a, int n) {
        const Word r = (a >> n) | (a << ((32 - n) & 31));
        if (n % 8) { split4(a, n % 8); emit4(r); }
        return r;
    }
    ZKS_HD Word shr(Word a, int n) {
        const int q = n / 8, b = n % 8;
        const Word r = a >> n;
        if (b) {
            split4(a, b);
            uint64_t o[4];
            int m = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k + q + 1 < 4) o[m++] = (r >> (8 * k)) & 0xff;
            if (m) emit.block(o, m);
        }
        return r;
    }
    template <int N>
    ZKS_HD Word add_mod32(const Word* w, uint64_t c) {
        // the chain of G::linear_combination over the terms (w[0] byte 0..3, w[1] byte 0..3, ..., then the constant): partial sums after
        // the first four terms and after every further three
        constexpr int TB = 4 * N;
        const int T = TB + (c ? 1 : 0);
        uint64_t acc = 0, o[8];
        int m = 0;
#pragma unroll
        for (int t = 0; t < TB + 1; ++t) {
            if (t < T) {
                acc += t < TB ? (uint64_t)((w[t / 4] >> (8 * (t % 4))) & 0xff) << (8 * (t % 4)) : c;
                const bool boundary = (t == 3) || (t > 3 && (t - 4) % 3 == 2) || (t == T - 1);
                if (boundary && !(t < 3)) { o[m++] = acc; if (m == 8) { emit.block(o, 8); m = 0; } }
            }
        }
        if (m) { emit.block(o, m); m = 0; }
        const uint32_t low = (uint32_t)acc;
        const uint64_t carry = acc >> 32;
        const uint64_t parts[5] = {low & 0xffu, (low >> 8) & 0xffu, (low >> 16) & 0xffu, low >> 24, carry};
        emit.block(parts, 5);
        const uint64_t tail[3] = {low, acc, 0};   // low word; low + 2^32 carry; the carry's range-check lookup (carry ^ carry)
        emit.block(tail, 3);
        return low;
    }
    ZKS_HD void range_check_word(Word w) {
        const uint64_t o[2] = {((w) ^ (w >> 8)) & 0xffu, ((w >> 16) ^ (w >> 24)) & 0xffu};
        emit.block(o, 2);
    }
};

}  // namespace zks
