// circuits/sha256_gadget4.hpp — SHA-256 compression over 4-bit chunks through the REFERENCE's table set
// (/root/reference/src/code_unpacker_sha256/mod.rs:554-566: Maj4Table, TriXor4Table, Ch4Table, Split4BitChunkTable<1>,
// Split4BitChunkTable<2>, all of lookup width 4; LookupParameters width 4 x 8 repetitions, :491-495).  SURVEY §8 a16.
//
// boojum's own round function (gadgets::sha256) is absent from /root/reference ([EXT]); what the reference pins is the table set and
// the lookup geometry, so this file is a decomposition over exactly those five tables:
//   * a u32 is eight little-endian nibbles; Sigma / sigma rotations by r = 4 q + s move nibbles by q and bits by s:
//       s = 1: Split4BitChunk<1>(x) = (x & 1, x >> 1),  s = 2: Split4BitChunk<2>(x) = (x & 3, x >> 2),
//       s = 3: both tables in sequence (x >> 1 is a 3-bit chunk: Split<2> of it gives (x >> 1) & 3 and x >> 3);
//     a rotated nibble is high(x_j) + 2^(4-s) low(x_{j+1}) — one FMA gate;
//   * the three-way XORs are TriXor4 lookups, Ch and Maj one Ch4 / Maj4 lookup per nibble;
//   * additions mod 2^32 are one field sum re-split into eight nibbles + a carry (ZK_OP_SPLIT), recomposed by reduction gates; every
//     nibble is range-checked by the lookup that consumes it as a key, the few that feed additions only by a TriXor4 lookup.
// The byte interface (state bytes in / out, block bytes in) is the 8-bit gadget's, so the FSM circuits and their seed hints
// (ZK_OP_SHA256_COMPRESS over byte variables) do not change.
#pragma once
#include <memory>
#include "sha256_gadget.hpp"
#include "../sha256_macro4.hpp"

namespace zkgl {
namespace sha256_gadget4 {

using sha256_gadget::SHA_K;
using sha256_gadget::Word;   // four little-endian byte variables

// table markers of the reference's set (the Rust types of code_unpacker_sha256/mod.rs:554-566)
constexpr uint32_t TABLE_MAJ4 = 48, TABLE_CH4 = 50, TABLE_SPLIT4_1 = 51, TABLE_SPLIT4_2 = 52;   // TABLE_TRIXOR4 = 49: gadgets.hpp (the range checks use it too)

inline void add_reference_sha_tables(CS& cs) {
    std::vector<uint64_t> maj, tri, ch;
    for (uint64_t a = 0; a < 16; ++a)
        for (uint64_t b = 0; b < 16; ++b)
            for (uint64_t c = 0; c < 16; ++c) {
                for (auto* v : {&maj, &tri, &ch}) { v->push_back(a); v->push_back(b); v->push_back(c); }
                maj.push_back((a & b) ^ (a & c) ^ (b & c));
                tri.push_back(a ^ b ^ c);
                ch.push_back((a & b) ^ (~a & 0xf & c));
            }
    cs.add_table(TABLE_MAJ4, 3, 1, maj.data(), 4096);
    cs.add_table(TABLE_TRIXOR4, 3, 1, tri.data(), 4096);
    cs.add_table(TABLE_CH4, 3, 1, ch.data(), 4096);
    for (uint32_t at : {1u, 2u}) {   // chunk -> (low `at` bits, high 4 - at bits, the two halves swapped)
        std::vector<uint64_t> rows;
        for (uint64_t x = 0; x < 16; ++x) {
            const uint64_t lo = x & ((1u << at) - 1), hi = x >> at;
            rows.push_back(x); rows.push_back(lo); rows.push_back(hi); rows.push_back((lo << (4 - at)) | hi);
        }
        cs.add_table(at == 1 ? TABLE_SPLIT4_1 : TABLE_SPLIT4_2, 1, 3, rows.data(), 16);
    }
}

using Nib8 = std::array<zk_var, 8>;
struct W4 {                 // a u32 as nibbles; `packed` (the word as one field element) where an addition produced or needs it
    Nib8 n;
    zk_var packed = ZK_VAR_NONE;
};

// The gadget is the HOST backend of zks4::compress (csrc/sha256_macro4.hpp): the walk the device macro-op (ZK_OP_SHA256_ROUNDS, a = 1), the
// counting backend and the oracle make too.  Plain mode (ZKGL_SHA4_MACRO=0): every primitive records its witness op AND its constraint.  Macro mode
// (the default since round 6): a compression records ONE witness op over the 96 input bytes whose outputs are pre-allocated variables, and the
// walk places the same lookups / gates over them in the same order — the same circuit, cell for cell (tests/test_sha4_macro.py).
struct S4 {
    typedef Word Bytes;
    typedef W4 W;
    typedef Nib8 Nib;
    typedef zk_var V;
    // per-nibble split cache of one word: lo[s][j] = x_j & (2^s - 1), hi[s][j] = x_j >> s
    struct Splits { Nib8 lo[4], hi[4]; bool have[4] = {false, false, false, false}; };
    G& g;
    uint32_t t_maj, t_tri, t_ch, t_s1, t_s2;
    bool use_macro;
    zk_var macro_next = ZK_VAR_NONE;   // macro mode: the next pre-allocated output variable
    std::vector<std::pair<zk_var, uint64_t>> terms;
    std::vector<zk_var> loose;          // nibbles no lookup consumes as a key: range-checked explicitly, three per lookup
    explicit S4(G& g) : g(g) {
        t_maj = g.cs.table_id(TABLE_MAJ4); t_tri = g.cs.table_id(TABLE_TRIXOR4); t_ch = g.cs.table_id(TABLE_CH4);
        t_s1 = g.cs.table_id(TABLE_SPLIT4_1); t_s2 = g.cs.table_id(TABLE_SPLIT4_2);
        const char* e = getenv("ZKGL_SHA4_MACRO");
        use_macro = !(e && e[0] == '0');   // round 6: the macro-op is the default recording of the reference's table set (its kernels: k_witness_*_x<X_SHA4>)
    }
    // ---- recording primitives: plain (witness op + constraint) or macro mode (constraint over the next pre-allocated outputs)
    std::vector<zk_var> look(uint32_t table, const std::vector<zk_var>& keys, uint32_t n_vals) {
        if (macro_next == ZK_VAR_NONE) return g.lookup(table, keys, n_vals);
        std::vector<zk_var> vals(n_vals);
        for (uint32_t i = 0; i < n_vals; ++i) vals[i] = macro_next++;
        g.cs.lookup_given(table, keys.data(), (uint32_t)keys.size(), vals.data(), n_vals);
        return vals;
    }
    zk_var fma(uint64_t q, zk_var a, zk_var b, uint64_t l, zk_var c) {   // q a b + l c
        if (macro_next == ZK_VAR_NONE) return g.fma(q, a, b, l, c);
        const zk_var r = macro_next++;
        zk_var vars[4] = {a, b, c, r};
        uint64_t ks[2] = {q, l};
        g.cs.place_gate(ZK_GATE_FMA, vars, 4, ks, 2);
        return r;
    }
    zk_var lc(const std::vector<std::pair<zk_var, uint64_t>>& ts) {   // G::linear_combination's chain, outputs given in macro mode
        if (macro_next == ZK_VAR_NONE) return g.linear_combination(ts);
        size_t pos = 0;
        zk_var acc = ZK_VAR_NONE;
        while (pos < ts.size() || acc == ZK_VAR_NONE) {
            zk_var t[4];
            uint64_t k[4];
            int n = 0;
            if (acc != ZK_VAR_NONE) { t[n] = acc; k[n] = 1; ++n; }
            while (n < 4 && pos < ts.size()) { t[n] = ts[pos].first; k[n] = ts[pos].second; ++n; ++pos; }
            while (n < 4) { t[n] = g.zero(); k[n] = 0; ++n; }
            zk_var r = macro_next++;
            zk_var vars[5] = {t[0], t[1], t[2], t[3], r};
            g.cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
            acc = r;
        }
        return acc;
    }
    void split(zk_var x, uint32_t n_chunks, zk_var* parts) {   // ZK_OP_SPLIT into n_chunks 4-bit chunks (the last one keeps the residual)
        if (macro_next == ZK_VAR_NONE) {
            zk_var first = g.cs.alloc_vars(n_chunks);
            for (uint32_t i = 0; i < n_chunks; ++i) parts[i] = first + i;
            g.cs.emit_op(ZK_OP_SPLIT, n_chunks, 4, &x, 1, parts, n_chunks, nullptr, 0);
        } else {
            for (uint32_t i = 0; i < n_chunks; ++i) parts[i] = macro_next++;
        }
    }
    // ---- the backend interface of zks4::compress
    void need(const W4& w, Splits& sp, int s) {
        if (sp.have[s]) return;
        if (s == 2) {
            for (int j = 0; j < 8; ++j) { auto v = look(t_s2, {w.n[j]}, 3); sp.lo[2][j] = v[0]; sp.hi[2][j] = v[1]; }
        } else {   // s = 3 builds on s = 1 (round 5: only when the word does not have those rows yet — 8 lookups less per schedule word)
            if (!sp.have[1])
                for (int j = 0; j < 8; ++j) { auto v = look(t_s1, {w.n[j]}, 3); sp.lo[1][j] = v[0]; sp.hi[1][j] = v[1]; }
            sp.have[1] = true;
            if (s == 3)
                for (int j = 0; j < 8; ++j) {
                    auto v = look(t_s2, {sp.hi[1][j]}, 3);                       // (x >> 1) & 3, x >> 3
                    sp.lo[3][j] = fma(2, v[0], g.one(), 1, sp.lo[1][j]);          // x & 7
                    sp.hi[3][j] = v[1];
                }
        }
        sp.have[s] = true;
    }
    Nib8 nibs(const W4& w) { return w.n; }
    Nib8 rot(const W4& w, Splits& sp, int r, bool shift_only) {   // rotr (or shr) by r bits, 1 <= r < 32
        const int q = r / 4, s = r % 4;
        Nib8 o;
        if (s) need(w, sp, s);
        for (int i = 0; i < 8; ++i) {
            const int j = i + q;
            if (shift_only && j >= 8) { o[i] = g.zero(); continue; }
            const int jj = j % 8, jn = (j + 1) % 8;
            if (s == 0) { o[i] = w.n[jj]; continue; }
            if (shift_only && j + 1 >= 8) { o[i] = sp.hi[s][jj]; continue; }
            o[i] = fma(1ull << (4 - s), sp.lo[s][jn], g.one(), 1, sp.hi[s][jj]);
        }
        return o;
    }
    Nib8 tri(int t, const Nib8& a, const Nib8& b, const Nib8& c) {
        const uint32_t table = t == zks4::T_TRI ? t_tri : t == zks4::T_CH ? t_ch : t_maj;
        Nib8 o;
        for (int i = 0; i < 8; ++i) o[i] = look(table, {a[i], b[i], c[i]}, 1)[0];
        return o;
    }
    void sum_begin() { terms.clear(); }
    void sum_packed(const W4& w) { terms.push_back({w.packed, 1}); }
    void sum_nibs(const Nib8& n) { for (int i = 0; i < 8; ++i) terms.push_back({n[i], 1ull << (4 * i)}); }
    void sum_scalar(zk_var v) { terms.push_back({v, 1}); }
    void sum_const(uint64_t c) { if (c) terms.push_back({g.one(), c}); }
    zk_var sum_end() { return lc(terms); }
    // (the collected sum) mod 2^32 as nibbles + the packed word; the carry (< 16) is range-checked with the loose nibbles, the nibbles by
    // their consumers (or by the caller)
    W4 add_mod32(zk_var* carry_out) {
        zk_var sum = lc(terms);
        zk_var parts[9];
        split(sum, 9, parts);   // eight nibbles + the carry
        W4 r;
        std::vector<std::pair<zk_var, uint64_t>> low;
        for (int i = 0; i < 8; ++i) { r.n[i] = parts[i]; low.push_back({parts[i], 1ull << (4 * i)}); }
        r.packed = lc(low);
        g.enforce_equal(fma(1ull << 32, parts[8], g.one(), 1, r.packed), sum);
        *carry_out = parts[8];
        return r;
    }
    W4 from_bytes(const Word& b) {   // bytes (range-checked or not: the nibbles are checked by their consumers) -> nibbles
        W4 r;
        for (int k = 0; k < 4; ++k) {
            zk_var parts[2];
            split(b[k], 2, parts);
            zk_var vars[5] = {parts[0], parts[1], g.zero(), g.zero(), b[k]};
            uint64_t ks[4] = {1, 16, 0, 0};
            g.cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, ks, 4);
            r.n[2 * k] = parts[0]; r.n[2 * k + 1] = parts[1];
        }
        r.packed = lc({{b[0], 1}, {b[1], 1ull << 8}, {b[2], 1ull << 16}, {b[3], 1ull << 24}});
        return r;
    }
    Word to_bytes(const W4& w) {
        Word b;
        for (int k = 0; k < 4; ++k) b[k] = fma(16, w.n[2 * k + 1], g.one(), 1, w.n[2 * k]);
        return b;
    }
    void loose_nibs(const Nib8& n) { for (zk_var v : n) loose.push_back(v); }
    void loose_v(zk_var v) { loose.push_back(v); }
    void range_check_loose() {    // three chunks per TriXor4 lookup
        for (size_t i = 0; i < loose.size(); i += 3)
            (void)look(t_tri, {loose[i], i + 1 < loose.size() ? loose[i + 1] : g.zero(), i + 2 < loose.size() ? loose[i + 2] : g.zero()}, 1);
        loose.clear();
    }
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
        std::vector<W4> w(64);
        std::vector<Splits> wsp(64);
        (void)g.zero(); (void)g.one();   // the constants exist before anything of the compression is allocated (both modes: same variables)
        if (use_macro) {
            zks4::CountBackend cb;
            int cst[8] = {0}, cblk[16] = {0}, cw[64];
            zks4::CountBackend::Splits csp[64];
            zks4::compress(cb, cst, cblk, cw, csp, SHA_K);
            std::vector<zk_var> ins;
            for (auto& x : st)
                for (auto b : x) ins.push_back(b);
            for (auto& x : block_words)
                for (auto b : x) ins.push_back(b);
            const zk_var first = g.cs.alloc_vars(cb.n);
            g.cs.emit_macro_op(ZK_OP_SHA256_ROUNDS, ins.data(), 96, first, cb.n, 1);   // a = 1: the 4-bit-chunk decomposition
            macro_next = first;
            zks4::compress(*this, st.data(), block_words.data(), w.data(), wsp.data(), SHA_K);
            g.cs.end_macro_op();
            if (macro_next != first + cb.n) throw ZkError(ZK_ERR_INVALID, "internal: the 4-bit SHA-256 gadget and its macro-op disagree on the output count");
            macro_next = ZK_VAR_NONE;
        } else {
            zks4::compress(*this, st.data(), block_words.data(), w.data(), wsp.data(), SHA_K);
        }
    }
};

// the gadget a circuit gets: by the table set its CS was configured with
struct AnySha {
    std::unique_ptr<sha256_gadget::S> s8;
    std::unique_ptr<S4> s4;
    explicit AnySha(G& g) {
        if (g.cs.has_table(TABLE_TRIXOR4) && !g.cs.has_table(TABLE_XOR8)) s4 = std::make_unique<S4>(g);
        else s8 = std::make_unique<sha256_gadget::S>(g);
    }
    void compress_with_hint(std::array<Word, 8>& st, const std::array<Word, 16>& block_words) {
        if (s4) s4->compress_with_hint(st, block_words); else s8->compress_with_hint(st, block_words);
    }
};

}  // namespace sha256_gadget4
}  // namespace zkgl
