// circuits/sha256_gadget.hpp — SHA-256 compression over byte variables through 8-bit lookup tables (kernel K8), shared by
// the block-chain circuit, the sha256 precompile FSM (sha256.cpp) and code_unpacker_sha256 (code_unpacker.cpp).
// See sha256.cpp for the decomposition notes.
#pragma once
#include <cstdlib>
#include "../gadgets.hpp"
#include "../sha256_macro.hpp"

namespace zkgl {
namespace sha256_gadget {

constexpr uint32_t T_ANDN8 = 32, T_SPLIT_BASE = 40;
inline const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
inline const uint32_t SHA_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

using Word = std::array<zk_var, 4>;  // little-endian bytes of a u32

// The gadget is the HOST backend of zks::compress (csrc/sha256_macro.hpp), the walk the device macro-op ZK_OP_SHA256_ROUNDS and the
// oracle make too.  With the macro-op (default) a compression records ONE witness op over 96 input bytes whose outputs are the
// pre-allocated variables the walk then constrains; ZKGL_NO_HASH_MACROS=1 records one ZK_OP_LOOKUP / LC4 / SPLIT per value (rounds 1-3).
struct S {
    typedef sha256_gadget::Word Word;
    G& g;
    uint32_t t_xor, t_and, t_andn, t_split[8];
    bool use_macro;
    zk_var macro_next = ZK_VAR_NONE;
    explicit S(G& g) : g(g) {
        t_xor = g.cs.table_id(TABLE_XOR8);
        t_and = g.cs.table_id(TABLE_AND8);
        t_andn = g.cs.table_id(T_ANDN8);
        for (int k = 1; k < 8; ++k) t_split[k] = g.cs.table_id(T_SPLIT_BASE + k);
        const char* e = getenv("ZKGL_NO_HASH_MACROS");
        use_macro = !(e && e[0] == '1');
    }
    // ---- recording primitives: plain (witness op + constraint) or macro mode (constraint over the next pre-allocated outputs)
    std::vector<zk_var> look(uint32_t table, const std::vector<zk_var>& keys, uint32_t n_vals) {
        if (macro_next == ZK_VAR_NONE) return g.lookup(table, keys, n_vals);
        std::vector<zk_var> vals(n_vals);
        for (uint32_t i = 0; i < n_vals; ++i) vals[i] = macro_next++;
        g.cs.lookup_given(table, keys.data(), (uint32_t)keys.size(), vals.data(), n_vals);
        return vals;
    }
    zk_var lc(const std::vector<std::pair<zk_var, uint64_t>>& terms) {   // G::linear_combination's chain, outputs given in macro mode
        if (macro_next == ZK_VAR_NONE) return g.linear_combination(terms);
        size_t pos = 0;
        zk_var acc = ZK_VAR_NONE;
        while (pos < terms.size() || acc == ZK_VAR_NONE) {
            zk_var t[4];
            uint64_t k[4];
            int n = 0;
            if (acc != ZK_VAR_NONE) { t[n] = acc; k[n] = 1; ++n; }
            while (n < 4 && pos < terms.size()) { t[n] = terms[pos].first; k[n] = terms[pos].second; ++n; ++pos; }
            while (n < 4) { t[n] = g.zero(); k[n] = 0; ++n; }
            zk_var r = macro_next++;
            zk_var vars[5] = {t[0], t[1], t[2], t[3], r};
            g.cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
            acc = r;
        }
        return acc;
    }
    // ---- the backend interface of zks::compress
    Word bytewise(int t, const Word& a, const Word& b) {
        const uint32_t table = t == zks::T_XOR ? t_xor : t == zks::T_AND ? t_and : t_andn;
        Word r;
        for (int k = 0; k < 4; ++k) r[k] = look(table, {a[k], b[k]}, 1)[0];
        return r;
    }
    Word xor3(const Word& a, const Word& b, const Word& c) { Word ab = bytewise(zks::T_XOR, a, b); return bytewise(zks::T_XOR, ab, c); }
    // split every byte at bit position `at`: byte = lo (at bits) + 2^at * hi (8-at bits)
    void split_all(const Word& a, int at, Word& lo, Word& hi) {
        for (int k = 0; k < 4; ++k) {
            auto v = look(t_split[at], {a[k]}, 2);
            lo[k] = v[0]; hi[k] = v[1];
        }
    }
    Word rotr(const Word& a, int n) {  // rotate right by n bits (1 <= n < 32)
        const int q = n / 8, b = n % 8;
        Word r;
        if (b == 0) {
            for (int k = 0; k < 4; ++k) r[k] = a[(k + q) % 4];
            return r;
        }
        Word lo, hi;  // byte = lo (b bits) + 2^b hi ; (x >> b): new byte j = hi[j] + lo[j+1] * 2^(8-b)
        split_all(a, b, lo, hi);
        for (int k = 0; k < 4; ++k) {
            int j = (k + q) % 4;
            r[k] = lc({{hi[j], 1}, {lo[(j + 1) % 4], 1ull << (8 - b)}});
        }
        return r;
    }
    Word shr(const Word& a, int n) {  // logical shift right by n bits
        const int q = n / 8, b = n % 8;
        Word r, lo, hi;
        if (b) split_all(a, b, lo, hi);
        for (int k = 0; k < 4; ++k) {
            int j = k + q;
            if (j >= 4) { r[k] = g.zero(); continue; }
            if (b == 0) { r[k] = a[j]; continue; }
            if (j + 1 < 4) r[k] = lc({{hi[j], 1}, {lo[j + 1], 1ull << (8 - b)}});
            else r[k] = hi[j];
        }
        return r;
    }
    // (sum of the given words + constant) mod 2^32, as range-checkable bytes
    template <int N>
    Word add_mod32(const Word* words, uint64_t constant) {
        std::vector<std::pair<zk_var, uint64_t>> terms;
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 4; ++k) terms.push_back({words[i][k], 1ull << (8 * k)});
        if (constant) terms.push_back({g.one(), constant});
        zk_var sum = lc(terms);  // < (n+1) * 2^32 << p
        zk_var parts[5];
        if (macro_next == ZK_VAR_NONE) {
            zk_var first = g.cs.alloc_vars(5);
            for (int i = 0; i < 5; ++i) parts[i] = first + i;
            g.cs.emit_op(ZK_OP_SPLIT, 5, 8, &sum, 1, parts, 5, nullptr, 0);  // 4 bytes + carry
        } else {
            for (int i = 0; i < 5; ++i) parts[i] = macro_next++;
        }
        zk_var low = lc({{parts[0], 1}, {parts[1], 1ull << 8}, {parts[2], 1ull << 16}, {parts[3], 1ull << 24}});
        g.enforce_equal(lc({{low, 1}, {parts[4], 1ull << 32}}), sum);
        (void)look(t_xor, {parts[4], parts[4]}, 1);  // carry < 2^8 (it is < 8); the 4 bytes are range-checked by their consumers
        return {parts[0], parts[1], parts[2], parts[3]};
    }
    void range_check_word(const Word& w) {
        (void)look(t_xor, {w[0], w[1]}, 1);
        (void)look(t_xor, {w[2], w[3]}, 1);
    }
    // compress + (loop scope) the seed hint ZK_OP_SHA256_COMPRESS over the same byte variables
    void compress_with_hint(std::array<Word, 8>& st, const std::array<Word, 16>& block_words) {
        std::vector<zk_var> ins;
        for (auto& w : st)
            for (auto b : w) ins.push_back(b);
        for (auto& w : block_words)
            for (auto b : w) ins.push_back(b);
        compress(st, block_words);
        if (g.cs.in_loop()) {
            std::vector<zk_var> outs;
            for (auto& w : st)
                for (auto b : w) outs.push_back(b);
            g.cs.seed_hint(ZK_OP_SHA256_COMPRESS, ins.data(), 96, outs.data(), 32);
        }
    }
    void compress(std::array<Word, 8>& st, const std::array<Word, 16>& block_words) {
        Word w[64];
        if (use_macro) {
            (void)g.zero(); (void)g.one();
            zks::CountBackend cb;
            int cst[8] = {0}, cblk[16] = {0}, cw[64];
            zks::compress(cb, cst, cblk, cw, SHA_K);
            std::vector<zk_var> ins;
            for (auto& x : st)
                for (auto b : x) ins.push_back(b);
            for (auto& x : block_words)
                for (auto b : x) ins.push_back(b);
            const zk_var first = g.cs.alloc_vars(cb.n);
            g.cs.emit_macro_op(ZK_OP_SHA256_ROUNDS, ins.data(), 96, first, cb.n);
            macro_next = first;
            zks::compress(*this, st.data(), block_words.data(), w, SHA_K);
            g.cs.end_macro_op();
            if (macro_next != first + cb.n) throw ZkError(ZK_ERR_INVALID, "internal: the SHA-256 gadget and its macro-op disagree on the output count");
            macro_next = ZK_VAR_NONE;
        } else {
            zks::compress(*this, st.data(), block_words.data(), w, SHA_K);
        }
    }
};


}  // namespace sha256_gadget
}  // namespace zkgl
