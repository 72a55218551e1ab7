// sha256_macro4.hpp — the STRUCTURE of the SHA-256 compression over 4-bit chunks through the REFERENCE's table set (Maj4 / TriXor4 / Ch4 /
// Split4BitChunk<1,2>, /root/reference/src/code_unpacker_sha256/mod.rs:554-566; round function surface
// /root/reference/src/sha256_round_function/mod.rs:271-285), written once and walked by
//   * the host gadget (circuits/sha256_gadget4.hpp S4): W = eight nibble variables + the packed word, every primitive records its lookups / gates;
//   * the device macro-op ZK_OP_SHA256_ROUNDS with a = 1 (kernels_engine2.hpp: the kernels instantiated with X_SHA4): W = uint32, every primitive
//     computes in registers and STREAMS OUT the same intermediates in the same order;
//   * a counting backend (the number of outputs).
// (the engine's 8-bit decomposition: sha256_macro.hpp.  The oracle restates this walk in plain C: oracle/zko_engine.c sh4_compress.)
//
// Backend primitives and their outputs (the values the trace holds, in this order):
//   from_bytes(b)            -> 8: (low nibble, high nibble) of byte 0..3; then 1: the word b0 + 2^8 b1 + 2^16 b2 + 2^24 b3
//   rot(w, sp, r, shr)       -> the splits the rotation needs and the word does not have yet (sp remembers), s = r % 4:
//                                 s = 2: per nibble j = 0..7 the Split4BitChunk<2> row of x_j: (x & 3, x >> 2, swapped)                 3 each
//                                 s = 1: per nibble the Split4BitChunk<1> row: (x & 1, x >> 1, swapped)                                  3 each
//                                 s = 3: the s = 1 rows unless the word has them, then per nibble the
//                                        Split4BitChunk<2> row of x >> 1: ((x >> 1) & 3, x >> 3, swapped) and x & 7 = 2 ((x >> 1) & 3) + (x & 1)   3 + 3 + 1 each
//                               then, s != 0, one per output nibble i whose two halves come from different input nibbles:
//                                 hi_s(x_j) + 2^(4-s) lo_s(x_(j+1)), j = i + r / 4 (rotation: indices mod 8; shift: nothing past nibble 7)
//   tri(T, a, b, c)          -> 8: T(a_i, b_i, c_i), T = TriXor4 / Ch4 / Maj4
//   sum_*  (a linear combination, G::linear_combination's chain): one partial sum after the first four terms, one per further three,
//                               one for a remainder (a sum of fewer than four terms: one)
//   add_mod32 (after its sum) -> 9: the eight nibbles and the carry of the sum; 3: the chain of the packed low word; 1: 2^32 carry + low
//   to_bytes(w)              -> 4: byte k = 16 n_(2k+1) + n_(2k)
//   range_check_loose()      -> one TriXor4 value per three collected nibbles (nibbles no lookup consumes as a key, carries)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZKS4_HD __host__ __device__ __forceinline__
#else
#define ZKS4_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKS4_LOOP _Pragma("unroll 1")
#else
#define ZKS4_LOOP
#endif

namespace zks4 {

enum Table : int { T_TRI = 0, T_CH = 1, T_MAJ = 2 };

// st[8] <- compress(st, block[16]) over bytes; w / wsp: 64 words (and their split states) of working storage the caller provides
template <class B>
ZKS4_HD void compress(B& be, typename B::Bytes st[8], const typename B::Bytes block[16], typename B::W w[64], typename B::Splits wsp[64], const uint32_t k[64]) {
    typedef typename B::W W;
    typedef typename B::Nib Nib;
    typedef typename B::V V;
    typedef typename B::Splits Splits;
    ZKS4_LOOP
    for (int i = 0; i < 16; ++i) w[i] = be.from_bytes(block[i]);
    be.loose_nibs(be.nibs(w[0]));   // w[0] feeds round 0's addition only
    ZKS4_LOOP
    for (int i = 16; i < 64; ++i) {
        // (every call in its own statement: the order of the outputs must not depend on the order a compiler evaluates arguments in)
        Splits& a = wsp[i - 15];
        Splits& b = wsp[i - 2];
        const Nib a7 = be.rot(w[i - 15], a, 7, false);
        const Nib a18 = be.rot(w[i - 15], a, 18, false);
        const Nib a3 = be.rot(w[i - 15], a, 3, true);
        const Nib s0 = be.tri(T_TRI, a7, a18, a3);
        const Nib b17 = be.rot(w[i - 2], b, 17, false);
        const Nib b19 = be.rot(w[i - 2], b, 19, false);
        const Nib b10 = be.rot(w[i - 2], b, 10, true);
        const Nib s1 = be.tri(T_TRI, b17, b19, b10);
        be.sum_begin();
        be.sum_packed(w[i - 16]);
        be.sum_packed(w[i - 7]);
        be.sum_nibs(s0);
        be.sum_nibs(s1);
        V carry;
        w[i] = be.add_mod32(&carry);
        be.loose_v(carry);
    }
    be.loose_nibs(be.nibs(w[62]));   // the only schedule words no sigma lookup consumes
    be.loose_nibs(be.nibs(w[63]));
    W s[8];
    ZKS4_LOOP
    for (int i = 0; i < 8; ++i) s[i] = be.from_bytes(st[i]);
    W a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    ZKS4_LOOP
    for (int i = 0; i < 64; ++i) {
        Splits se = Splits(), sa = Splits();
        const Nib e6 = be.rot(e, se, 6, false);
        const Nib e11 = be.rot(e, se, 11, false);
        const Nib e25 = be.rot(e, se, 25, false);
        const Nib S1 = be.tri(T_TRI, e6, e11, e25);
        const Nib ch = be.tri(T_CH, be.nibs(e), be.nibs(f), be.nibs(g));
        const Nib a2 = be.rot(a, sa, 2, false);
        const Nib a13 = be.rot(a, sa, 13, false);
        const Nib a22 = be.rot(a, sa, 22, false);
        const Nib S0 = be.tri(T_TRI, a2, a13, a22);
        const Nib mj = be.tri(T_MAJ, be.nibs(a), be.nibs(b), be.nibs(c));
        be.sum_begin();
        be.sum_packed(h);
        be.sum_packed(w[i]);
        be.sum_const(k[i]);
        be.sum_nibs(S1);
        be.sum_nibs(ch);
        const V T1 = be.sum_end();                     // unreduced: < 5 * 2^32
        V c1, c2;
        be.sum_begin();
        be.sum_packed(d);
        be.sum_scalar(T1);
        const W new_e = be.add_mod32(&c1);
        be.sum_begin();
        be.sum_scalar(T1);
        be.sum_nibs(S0);
        be.sum_nibs(mj);
        const W new_a = be.add_mod32(&c2);
        be.loose_v(c1);
        be.loose_v(c2);
        h = g; g = f; f = e; e = new_e; d = c; c = b; b = a; a = new_a;
    }
    be.loose_nibs(be.nibs(a));   // outputs of the last round feed additions only
    be.loose_nibs(be.nibs(e));
    const W out[8] = {a, b, c, d, e, f, g, h};
    ZKS4_LOOP
    for (int i = 0; i < 8; ++i) {
        V carry;
        be.sum_begin();
        be.sum_packed(s[i]);
        be.sum_packed(out[i]);
        const W r = be.add_mod32(&carry);
        be.loose_v(carry);
        be.loose_nibs(be.nibs(r));
        st[i] = be.to_bytes(r);
    }
    be.range_check_loose();
}

// number of terms -> number of chain outputs (G::linear_combination: first gate folds four terms, every further one the running sum + three)
ZKS4_HD uint32_t chain_outputs(uint32_t n_terms) { return n_terms <= 4 ? 1u : 1u + (n_terms - 4 + 2) / 3; }

struct CountBackend {
    typedef int Bytes;
    typedef int W;
    typedef int Nib;
    typedef int V;
    struct Splits { uint8_t have = 0; };
    uint32_t n = 0, terms = 0, loose = 0;
    W from_bytes(Bytes) { n += 8 + 1; return 0; }
    Nib nibs(W) { return 0; }
    Nib rot(W, Splits& sp, int r, bool shift_only) {
        const int q = r / 4, s = r % 4;
        if (s && !(sp.have & (1 << s))) {
            if (s == 2) n += 8 * 3;
            else { if (!(sp.have & 2)) n += 8 * 3; sp.have |= 2; if (s == 3) n += 8 * 4; }
            sp.have |= (uint8_t)(1 << s);
        }
        for (int i = 0; i < 8; ++i) {
            const int j = i + q;
            if (s == 0 || (shift_only && j >= 8) || (shift_only && j + 1 >= 8)) continue;
            ++n;
        }
        return 0;
    }
    Nib tri(int, Nib, Nib, Nib) { n += 8; return 0; }
    void sum_begin() { terms = 0; }
    void sum_packed(W) { ++terms; }
    void sum_nibs(Nib) { terms += 8; }
    void sum_scalar(V) { ++terms; }
    void sum_const(uint64_t c) { if (c) ++terms; }
    V sum_end() { n += chain_outputs(terms); return 0; }
    W add_mod32(V* carry) { (void)sum_end(); n += 9 + chain_outputs(8) + 1; *carry = 0; return 0; }
    Bytes to_bytes(W) { n += 4; return 0; }
    void loose_nibs(Nib) { loose += 8; }
    void loose_v(V) { ++loose; }
    void range_check_loose() { n += (loose + 2) / 3; }
};

// compute backend over uint32 words.  Emit: one(v) = the next output (the strands of a tile share the stores run by run: kernels_engine2.hpp)
template <class Emit>
struct ComputeBackend {
    typedef uint32_t Bytes;   // four little-endian bytes
    typedef uint32_t W;       // the word: its nibbles ARE its bits, its packed value is itself
    typedef uint32_t Nib;
    typedef uint64_t V;
    struct Splits { uint8_t have; ZKS4_HD Splits() : have(0) {} };
    Emit& emit;
    uint64_t acc;
    uint32_t terms;
    uint32_t loose_buf[40];   // collected nibbles, eight per word (carries are < 16 for byte inputs; anything else is reported by the op's input test)
    uint32_t n_loose;
    ZKS4_HD explicit ComputeBackend(Emit& e) : emit(e), acc(0), terms(0), n_loose(0) {}
    ZKS4_HD static uint32_t nib(uint32_t w, int j) { return (w >> (4 * j)) & 15u; }
    ZKS4_HD W from_bytes(Bytes b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) emit.one(nib(b, j));
        emit.one(b);
        return b;
    }
    ZKS4_HD Nib nibs(W w) { return w; }
    ZKS4_HD void split_rows(uint32_t x8, int at) {   // the Split4BitChunk<at> row of each of eight chunks
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t x = nib(x8, j), lo = x & ((1u << at) - 1), hi = x >> at;
            emit.one(lo); emit.one(hi); emit.one((lo << (4 - at)) | hi);
        }
    }
    ZKS4_HD Nib rot(W w, Splits& sp, int r, bool shift_only) {
        const int q = r / 4, s = r % 4;
        if (s && !(sp.have & (1 << s))) {
            if (s == 2) split_rows(w, 2);
            else {
                if (!(sp.have & 2)) split_rows(w, 1);
                sp.have |= 2;
                if (s == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t x = nib(w, j), h1 = x >> 1, lo = h1 & 3u, hi = h1 >> 2;
                        emit.one(lo); emit.one(hi); emit.one((lo << 2) | hi);
                        emit.one(x & 7u);
                    }
                }
            }
            sp.have |= (uint8_t)(1 << s);
        }
        const uint32_t res = shift_only ? (w >> r) : ((w >> r) | (w << ((32 - r) & 31)));
        if (s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = i + q;
                if ((shift_only && j >= 8) || (shift_only && j + 1 >= 8)) continue;
                emit.one(nib(res, i));
            }
        }
        return res;
    }
    ZKS4_HD Nib tri(int t, Nib a, Nib b, Nib c) {
        const uint32_t r = t == T_TRI ? a ^ b ^ c : t == T_CH ? (a & b) ^ (~a & c) : (a & b) ^ (a & c) ^ (b & c);
#pragma unroll
        for (int i = 0; i < 8; ++i) emit.one(nib(r, i));
        return r;
    }
    ZKS4_HD void term(uint64_t v) {
        acc += v;
        ++terms;
        if (terms == 4 || (terms > 4 && (terms - 4) % 3 == 0)) emit.one(acc);
    }
    ZKS4_HD void sum_begin() { acc = 0; terms = 0; }
    ZKS4_HD void sum_packed(W w) { term(w); }
    ZKS4_HD void sum_nibs(Nib n) {
#pragma unroll
        for (int i = 0; i < 8; ++i) term((uint64_t)nib(n, i) << (4 * i));
    }
    ZKS4_HD void sum_scalar(V v) { term(v); }
    ZKS4_HD void sum_const(uint64_t c) { if (c) term(c); }
    ZKS4_HD V sum_end() {
        if (terms < 4 || (terms - 4) % 3 != 0) emit.one(acc);
        return acc;
    }
    ZKS4_HD W add_mod32(V* carry) {
        const uint64_t sum = sum_end();
        const uint32_t low = (uint32_t)sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) emit.one(nib(low, j));
        const uint64_t cy = sum >> 32;
        emit.one(cy);
        emit.one(low & 0xffffu);      // chain of the packed low word: after four nibbles,
        emit.one(low & 0xfffffffu);   // after seven,
        emit.one(low);                // after eight
        emit.one(sum);                // 2^32 carry + low (enforced equal to the sum)
        *carry = cy;
        return low;
    }
    ZKS4_HD Bytes to_bytes(W w) {
#pragma unroll
        for (int k = 0; k < 4; ++k) emit.one((w >> (8 * k)) & 0xffu);
        return w;
    }
    ZKS4_HD void push_loose(uint32_t v) {
        const uint32_t at = n_loose >> 3, sh = 4 * (n_loose & 7);
        loose_buf[at] = sh ? (loose_buf[at] | ((v & 15u) << sh)) : (v & 15u);
        ++n_loose;
    }
    ZKS4_HD void loose_nibs(Nib n) {
#pragma unroll
        for (int i = 0; i < 8; ++i) push_loose(nib(n, i));
    }
    ZKS4_HD void loose_v(V v) { push_loose((uint32_t)v); }
    ZKS4_HD uint32_t loose_at(uint32_t i) const { return i < n_loose ? (loose_buf[i >> 3] >> (4 * (i & 7))) & 15u : 0u; }
    ZKS4_HD void range_check_loose() {
        ZKS4_LOOP
        for (uint32_t i = 0; i < n_loose; i += 3) emit.one(loose_at(i) ^ loose_at(i + 1) ^ loose_at(i + 2));
    }
};

}  // namespace zks4
