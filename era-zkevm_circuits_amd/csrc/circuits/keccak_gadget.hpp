// circuits/keccak_gadget.hpp — Keccak-f[1600] over byte variables through 8-bit lookup tables (kernel K8), shared by
// the block-chain circuit, the precompile FSM (keccak.cpp) and eip_4844 (eip4844.cpp).  See keccak.cpp for the
// decomposition notes and the reference surface (/root/reference/src/keccak256_round_function/mod.rs:796-838).
#pragma once
#include "../gadgets.hpp"

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

struct K {
    G& g;
    uint32_t t_xor, t_andn, t_split[8];
    explicit K(G& g) : g(g) {
        t_xor = g.cs.table_id(TABLE_XOR8);
        t_andn = g.cs.table_id(TABLE_ANDN8);
        for (int k = 1; k < 8; ++k) t_split[k] = g.cs.table_id(TABLE_SPLIT_BASE + k);
    }
    zk_var xor8(zk_var a, zk_var b) { return g.lookup(t_xor, {a, b}, 1)[0]; }
    zk_var andn8(zk_var a, zk_var b) { return g.lookup(t_andn, {a, b}, 1)[0]; }  // (~a) & b
    Lane xor_lane(const Lane& a, const Lane& b) {
        Lane r;
        for (int k = 0; k < 8; ++k) r[k] = xor8(a[k], b[k]);
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
            auto v = g.lookup(t_split[8 - b], {a[k]}, 2);
            lo[k] = v[0]; hi[k] = v[1];
        }
        for (int k = 0; k < 8; ++k) {
            zk_var nb = g.linear_combination({{lo[k], 1ull << b}, {hi[(k + 7) % 8], 1}});
            r[(k + q) % 8] = nb;
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
        for (int rnd = 0; rnd < 24; ++rnd) {
            std::array<Lane, 5> c, d;
            for (int x = 0; x < 5; ++x) {
                c[x] = xor_lane(s[x], s[x + 5]);
                c[x] = xor_lane(c[x], s[x + 10]);
                c[x] = xor_lane(c[x], s[x + 15]);
                c[x] = xor_lane(c[x], s[x + 20]);
            }
            for (int x = 0; x < 5; ++x) d[x] = xor_lane(c[(x + 4) % 5], rotl(c[(x + 1) % 5], 1));
            for (int i = 0; i < 25; ++i) s[i] = xor_lane(s[i], d[i % 5]);
            std::array<Lane, 25> b;
            for (int x = 0; x < 5; ++x)
                for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(s[x + 5 * y], KECCAK_ROT[x + 5 * y]);
            for (int y = 0; y < 5; ++y)
                for (int x = 0; x < 5; ++x) {
                    Lane t;
                    for (int k = 0; k < 8; ++k) t[k] = andn8(b[(x + 1) % 5 + 5 * y][k], b[(x + 2) % 5 + 5 * y][k]);
                    s[x + 5 * y] = xor_lane(b[x + 5 * y], t);
                }
            for (int k = 0; k < 8; ++k) {
                uint64_t byte = (KECCAK_RC[rnd] >> (8 * k)) & 0xff;
                if (byte) s[0][k] = xor8(s[0][k], g.constant(byte));
            }
        }
    }
};


}  // namespace zkgl
