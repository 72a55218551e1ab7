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

struct S4 {
    G& g;
    uint32_t t_maj, t_tri, t_ch, t_s1, t_s2;
    explicit S4(G& g) : g(g) {
        t_maj = g.cs.table_id(TABLE_MAJ4); t_tri = g.cs.table_id(TABLE_TRIXOR4); t_ch = g.cs.table_id(TABLE_CH4);
        t_s1 = g.cs.table_id(TABLE_SPLIT4_1); t_s2 = g.cs.table_id(TABLE_SPLIT4_2);
    }
    // per-nibble split cache of one word: lo[s][j] = x_j & (2^s - 1), hi[s][j] = x_j >> s
    struct Splits { Nib8 lo[4], hi[4]; bool have[4] = {false, false, false, false}; };
    void need(const W4& w, Splits& sp, int s) {
        if (sp.have[s]) return;
        if (s == 2) {
            for (int j = 0; j < 8; ++j) { auto v = g.lookup(t_s2, {w.n[j]}, 3); sp.lo[2][j] = v[0]; sp.hi[2][j] = v[1]; }
        } else {   // s = 1 and s = 3 come together
            for (int j = 0; j < 8; ++j) { auto v = g.lookup(t_s1, {w.n[j]}, 3); sp.lo[1][j] = v[0]; sp.hi[1][j] = v[1]; }
            sp.have[1] = true;
            if (s == 3)
                for (int j = 0; j < 8; ++j) {
                    auto v = g.lookup(t_s2, {sp.hi[1][j]}, 3);                       // (x >> 1) & 3, x >> 3
                    sp.lo[3][j] = g.fma(2, v[0], g.one(), 1, sp.lo[1][j]);          // x & 7
                    sp.hi[3][j] = v[1];
                }
        }
        sp.have[s] = true;
    }
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
            o[i] = g.fma(1ull << (4 - s), sp.lo[s][jn], g.one(), 1, sp.hi[s][jj]);
        }
        return o;
    }
    Nib8 tri(uint32_t table, const Nib8& a, const Nib8& b, const Nib8& c) {
        Nib8 o;
        for (int i = 0; i < 8; ++i) o[i] = g.lookup(table, {a[i], b[i], c[i]}, 1)[0];
        return o;
    }
    void range_check_nibbles(const std::vector<zk_var>& v) {    // three chunks per TriXor4 lookup
        for (size_t i = 0; i < v.size(); i += 3)
            (void)g.lookup(t_tri, {v[i], i + 1 < v.size() ? v[i + 1] : g.zero(), i + 2 < v.size() ? v[i + 2] : g.zero()}, 1);
    }
    static void push_nibbles(std::vector<std::pair<zk_var, uint64_t>>& terms, const Nib8& n) {
        for (int i = 0; i < 8; ++i) terms.push_back({n[i], 1ull << (4 * i)});
    }
    // (sum of the terms + constant) mod 2^32 as nibbles + the packed word; the carry (< 16) is range-checked here, the nibbles by
    // their consumers (or by the caller)
    W4 add_mod32(std::vector<std::pair<zk_var, uint64_t>> terms, uint64_t constant, zk_var* carry_out) {
        if (constant) terms.push_back({g.one(), constant});
        zk_var sum = g.linear_combination(terms);
        zk_var parts[9];
        zk_var first = g.cs.alloc_vars(9);
        for (int i = 0; i < 9; ++i) parts[i] = first + i;
        g.cs.emit_op(ZK_OP_SPLIT, 9, 4, &sum, 1, parts, 9, nullptr, 0);   // eight nibbles + the carry
        W4 r;
        std::vector<std::pair<zk_var, uint64_t>> low;
        for (int i = 0; i < 8; ++i) { r.n[i] = parts[i]; low.push_back({parts[i], 1ull << (4 * i)}); }
        r.packed = g.linear_combination(low);
        g.enforce_equal(g.fma(1ull << 32, parts[8], g.one(), 1, r.packed), sum);
        *carry_out = parts[8];
        return r;
    }
    W4 from_bytes(const Word& b) {   // bytes (range-checked or not: the nibbles are checked by their consumers) -> nibbles
        W4 r;
        for (int k = 0; k < 4; ++k) {
            zk_var first = g.cs.alloc_vars(2);
            zk_var parts[2] = {first, first + 1};
            g.cs.emit_op(ZK_OP_SPLIT, 2, 4, &b[k], 1, parts, 2, nullptr, 0);
            zk_var vars[5] = {parts[0], parts[1], g.zero(), g.zero(), b[k]};
            uint64_t ks[4] = {1, 16, 0, 0};
            g.cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, ks, 4);
            r.n[2 * k] = parts[0]; r.n[2 * k + 1] = parts[1];
        }
        r.packed = g.linear_combination({{b[0], 1}, {b[1], 1ull << 8}, {b[2], 1ull << 16}, {b[3], 1ull << 24}});
        return r;
    }
    Word to_bytes(const W4& w) {
        Word b;
        for (int k = 0; k < 4; ++k) b[k] = g.fma(16, w.n[2 * k + 1], g.one(), 1, w.n[2 * k]);
        return b;
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
        std::vector<W4> w;
        std::vector<zk_var> loose;   // nibbles no lookup consumes as a key: range-checked explicitly, three per lookup
        for (int i = 0; i < 16; ++i) w.push_back(from_bytes(block_words[i]));
        for (zk_var v : w[0].n) loose.push_back(v);   // w[0] feeds round 0's addition only
        std::vector<Splits> wsp(64);
        for (int i = 16; i < 64; ++i) {
            Splits &a = wsp[i - 15], &b = wsp[i - 2];
            Nib8 s0 = tri(t_tri, rot(w[i - 15], a, 7, false), rot(w[i - 15], a, 18, false), rot(w[i - 15], a, 3, true));
            Nib8 s1 = tri(t_tri, rot(w[i - 2], b, 17, false), rot(w[i - 2], b, 19, false), rot(w[i - 2], b, 10, true));
            std::vector<std::pair<zk_var, uint64_t>> terms = {{w[i - 16].packed, 1}, {w[i - 7].packed, 1}};
            push_nibbles(terms, s0);
            push_nibbles(terms, s1);
            zk_var carry;
            w.push_back(add_mod32(terms, 0, &carry));
            loose.push_back(carry);
        }
        for (int i : {62, 63})   // the only schedule words no sigma lookup consumes
            for (zk_var v : w[i].n) loose.push_back(v);
        W4 s[8];
        for (int i = 0; i < 8; ++i) s[i] = from_bytes(st[i]);
        W4 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], gg = s[6], h = s[7];
        for (int i = 0; i < 64; ++i) {
            Splits se, sa;
            Nib8 S1 = tri(t_tri, rot(e, se, 6, false), rot(e, se, 11, false), rot(e, se, 25, false));
            Nib8 ch = tri(t_ch, e.n, f.n, gg.n);
            Nib8 S0 = tri(t_tri, rot(a, sa, 2, false), rot(a, sa, 13, false), rot(a, sa, 22, false));
            Nib8 mj = tri(t_maj, a.n, b.n, c.n);
            std::vector<std::pair<zk_var, uint64_t>> t1 = {{h.packed, 1}, {w[i].packed, 1}, {g.one(), SHA_K[i]}};
            push_nibbles(t1, S1);
            push_nibbles(t1, ch);
            zk_var T1 = g.linear_combination(t1);                       // unreduced: < 5 * 2^32
            std::vector<std::pair<zk_var, uint64_t>> t2 = {{T1, 1}};
            push_nibbles(t2, S0);
            push_nibbles(t2, mj);
            zk_var c1, c2;
            W4 new_e = add_mod32({{d.packed, 1}, {T1, 1}}, 0, &c1);
            W4 new_a = add_mod32(t2, 0, &c2);
            loose.push_back(c1); loose.push_back(c2);
            h = gg; gg = f; f = e; e = new_e; d = c; c = b; b = a; a = new_a;
        }
        for (zk_var v : a.n) loose.push_back(v);   // outputs of the last round feed additions only
        for (zk_var v : e.n) loose.push_back(v);
        const W4 out[8] = {a, b, c, d, e, f, gg, h};
        for (int i = 0; i < 8; ++i) {
            zk_var carry;
            W4 r = add_mod32({{s[i].packed, 1}, {out[i].packed, 1}}, 0, &carry);
            loose.push_back(carry);
            for (zk_var v : r.n) loose.push_back(v);
            st[i] = to_bytes(r);
        }
        range_check_nibbles(loose);
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
