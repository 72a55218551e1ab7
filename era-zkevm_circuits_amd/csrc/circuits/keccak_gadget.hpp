// circuits/keccak_gadget.hpp — Keccak-f[1600] over byte variables through 8-bit lookup tables (kernel K8), shared by
// the block-chain circuit, the precompile FSM (keccak.cpp) and eip_4844 (eip4844.cpp).  See keccak.cpp for the
// decomposition notes and the reference surface (/root/reference/src/keccak256_round_function/mod.rs:796-838).
#pragma once
#include <cstdlib>
#include "../gadgets.hpp"
#include "../keccak_macro.hpp"

namespace zkgl {

enum KeccakTables : uint32_t { TABLE_ANDN8 = 32, TABLE_SPLIT_BASE = 40 };  // TABLE_SPLIT_BASE + k: byte -> (low k bits, high 8-k bits)


inline const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
inline const int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]

using Lane = std::array<zk_var, 8>;  // little-endian bytes

// The gadget is the HOST backend of zkk::keccak_f (csrc/keccak_macro.hpp): the same walk that the device macro-op ZK_OP_KECCAK_F and
// the oracle make.  With the macro-op (default) a permutation records ONE witness op over 200 input bytes whose outputs are the
// pre-allocated variables the walk then constrains (lookup tuples through CS::lookup_given, rotated bytes through reduction gates);
// ZKGL_NO_HASH_MACROS=1 records the same constraints with one ZK_OP_LOOKUP / ZK_OP_LC4 per value instead (rounds 1-3; kept as the
// cross-check: same digests, same constraint count).
struct K {
    typedef zkgl::Lane Lane;
    G& g;
    uint32_t t_xor, t_andn, t_split[8];
    bool use_macro;
    zk_var macro_next = ZK_VAR_NONE;   // macro mode: the next pre-allocated output variable
    explicit K(G& g) : g(g) {
        t_xor = g.cs.table_id(TABLE_XOR8);
        t_andn = g.cs.table_id(TABLE_ANDN8);
        for (int k = 1; k < 8; ++k) t_split[k] = g.cs.table_id(TABLE_SPLIT_BASE + k);
        const char* e = getenv("ZKGL_NO_HASH_MACROS");
        use_macro = !(e && e[0] == '1');
    }
    std::vector<zk_var> look(uint32_t table, const std::vector<zk_var>& keys, uint32_t n_vals) {
        if (macro_next == ZK_VAR_NONE) return g.lookup(table, keys, n_vals);
        std::vector<zk_var> vals(n_vals);
        for (uint32_t i = 0; i < n_vals; ++i) vals[i] = macro_next++;
        g.cs.lookup_given(table, keys.data(), (uint32_t)keys.size(), vals.data(), n_vals);
        return vals;
    }
    zk_var xor8(zk_var a, zk_var b) { return look(t_xor, {a, b}, 1)[0]; }
    zk_var andn8(zk_var a, zk_var b) { return look(t_andn, {a, b}, 1)[0]; }  // (~a) & b
    // ---- the backend interface of zkk::keccak_f
    Lane xor_lane(const Lane& a, const Lane& b) {
        Lane r;
        for (int k = 0; k < 8; ++k) r[k] = xor8(a[k], b[k]);
        return r;
    }
    Lane andn_lane(const Lane& a, const Lane& b) {
        Lane r;
        for (int k = 0; k < 8; ++k) r[k] = andn8(a[k], b[k]);
        return r;
    }
    // 64-bit rotate left by n of a lane held as 8 LE bytes
    Lane rotl(const Lane& a, int n) {
        n %= 64;
        const int q = n / 8, b = n % 8;
        Lane r;
        if (b == 0) {
            for (int k = 0; k < 8; ++k) r[(k + q) % 8] = a[k];
            return r;
        }
        // byte = lo (8-b bits) + 2^(8-b) * hi (b bits);  rotated byte k' = lo[k] * 2^b + hi[k-1]
        std::array<zk_var, 8> lo, hi;
        for (int k = 0; k < 8; ++k) {
            auto v = look(t_split[8 - b], {a[k]}, 2);
            lo[k] = v[0]; hi[k] = v[1];
        }
        for (int k = 0; k < 8; ++k) {
            zk_var nb;
            if (macro_next == ZK_VAR_NONE) nb = g.linear_combination({{lo[k], 1ull << b}, {hi[(k + 7) % 8], 1}});
            else {
                nb = macro_next++;
                zk_var vars[5] = {lo[k], hi[(k + 7) % 8], g.zero(), g.zero(), nb};
                uint64_t ks[4] = {1ull << b, 1, 0, 0};
                g.cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, ks, 4);
            }
            r[(k + q) % 8] = nb;
        }
        return r;
    }
    Lane xor_const(const Lane& a, uint64_t c) {
        Lane r = a;
        for (int k = 0; k < 8; ++k) {
            const uint64_t byte = (c >> (8 * k)) & 0xff;
            if (byte) r[k] = xor8(a[k], g.constant(byte));
        }
        return r;
    }
    // state <- Keccak-f(state ^ block) recorded gate by gate, plus (loop scope) the seed hint that lets the seeding cone
    // compute the same 200 bytes with one native macro-op (include/zkgl_ir.h ZK_OP_KECCAK_ABSORB)
    void absorb_and_permute(std::array<Lane, 25>& s, const zk_var* block136) {
        std::vector<zk_var> ins;
        for (auto& lane : s)
            for (auto b : lane) ins.push_back(b);
        for (int j = 0; j < 136; ++j) {
            ins.push_back(block136[j]);
            s[j / 8][j % 8] = xor8(s[j / 8][j % 8], block136[j]);
        }
        permutation(s);
        if (g.cs.in_loop()) {
            std::vector<zk_var> outs;
            for (auto& lane : s)
                for (auto b : lane) outs.push_back(b);
            g.cs.seed_hint(ZK_OP_KECCAK_ABSORB, ins.data(), 336, outs.data(), 200);
        }
    }
    void permutation(std::array<Lane, 25>& s) {
        if (use_macro) {
            (void)g.zero();
            for (int r = 0; r < 24; ++r)      // the iota constants exist before the macro-op's outputs are allocated
                for (int k = 0; k < 8; ++k)
                    if ((KECCAK_RC[r] >> (8 * k)) & 0xff) (void)g.constant((KECCAK_RC[r] >> (8 * k)) & 0xff);
            zkk::CountBackend cb;
            int dummy[25] = {0};
            zkk::keccak_f(cb, dummy, KECCAK_RC);
            std::vector<zk_var> ins;
            for (auto& lane : s)
                for (auto b : lane) ins.push_back(b);
            const zk_var first = g.cs.alloc_vars(cb.n);
            g.cs.emit_macro_op(ZK_OP_KECCAK_F, ins.data(), 200, first, cb.n);
            macro_next = first;
            zkk::keccak_f(*this, s.data(), KECCAK_RC);
            g.cs.end_macro_op();
            if (macro_next != first + cb.n) throw ZkError(ZK_ERR_INVALID, "internal: the Keccak gadget and its macro-op disagree on the output count");
            macro_next = ZK_VAR_NONE;
        } else {
            zkk::keccak_f(*this, s.data(), KECCAK_RC);
        }
    }
};


}  // namespace zkgl
