// circuits/keccak.cpp — Keccak-f[1600] over byte variables through 8-bit lookup tables (kernel K8 of
// SURVEY.md §2) and the Keccak-256 sponge over pre-padded 136-byte blocks.
//
// Reference surface: `keccak256_absorb_and_run_permutation`
// (/root/reference/src/keccak256_round_function/mod.rs:796-838: xor the 136-byte block into the state
// `[[[UInt8; 8]; 5]; 5]`, then boojum's `keccak_256_round_function` [EXT]) and the whole-message
// `keccak256(cs, &bytes)` gadget used by eip_4844 (src/eip_4844/mod.rs:156-163, 207, 229-237).
// The precompile FSM around it (request queue, unaligned memory reads, ByteBuffer:
// src/keccak256_round_function/mod.rs:155-670) is NOT built yet.
//
// boojum's own decomposition is absent; this one uses, per round, theta 160+40+200 xor lookups and 40
// bit-rotation splits, rho/pi <= 192 splits, chi 200 andn + 200 xor lookups, iota <= 8 xor lookups
// (~1030 lookups + ~230 ReductionGates per round, 24 rounds).  One loop iteration = one block, the
// 200-byte sponge state is the carried state, so a message of n blocks runs as n GPU lanes.
//
// INPUT STREAMS: outer none; loop 336 words = carried state[200] (byte (x,y,k) at 8*(x+5y)+k) | block[136].
#include "../gadgets.hpp"

namespace zkgl {

enum KeccakTables : uint32_t { TABLE_ANDN8 = 32, TABLE_SPLIT_BASE = 40 };  // TABLE_SPLIT_BASE + k: byte -> (low k bits, high 8-k bits)

namespace {

const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
const int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]

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

}  // namespace

void keccak_configure(CS& cs) {  // geometry as the reference keccak tests: 100/0/8/4 (src/keccak256_round_function/mod.rs:847-852)
    cs.allow_lookup(3, 8, true);
    for (uint32_t k : {ZK_GATE_CONST, ZK_GATE_FMA, ZK_GATE_REDUCTION4, ZK_GATE_BOOLEAN, ZK_GATE_UINTX_ADD, ZK_GATE_SELECT,
                       ZK_GATE_ZEROCHECK, ZK_GATE_DOT4, ZK_GATE_MATMUL12_EXT, ZK_GATE_MATMUL12_INT, ZK_GATE_NOP,
                       ZK_GATE_PUBLIC_INPUT})
        cs.allow_gate(k);
    add_xor8_table(cs);
    {
        std::vector<uint64_t> rows;
        rows.reserve(65536 * 3);
        for (uint64_t a = 0; a < 256; ++a)
            for (uint64_t b = 0; b < 256; ++b) { rows.push_back(a); rows.push_back(b); rows.push_back((~a & 0xff) & b); }
        cs.add_table(TABLE_ANDN8, 2, 1, rows.data(), 65536);
    }
    for (int k = 1; k < 8; ++k) {  // ByteSplitTable<k> (src/keccak256_round_function/mod.rs:953-966)
        std::vector<uint64_t> rows;
        for (uint64_t a = 0; a < 256; ++a) { rows.push_back(a); rows.push_back(a & ((1u << k) - 1)); rows.push_back(a >> k); }
        cs.add_table(TABLE_SPLIT_BASE + k, 1, 2, rows.data(), 256);
    }
}

// Keccak-256 over `n_blocks` pre-padded 136-byte blocks; public inputs = the 32 digest bytes.
void keccak256_blocks_entry_point(CS& cs, uint32_t n_blocks) {
    G g(cs);
    zk_var outer_zero = g.zero();
    cs.loop_begin(n_blocks);
    K k(g);
    std::array<Lane, 25> s;
    std::vector<zk_var> state_in, state_out;
    for (int i = 0; i < 25; ++i)
        for (int b = 0; b < 8; ++b) {
            zk_var v = g.next_input();  // carried sponge state: every byte is an output of a lookup in the previous block
            cs.link(ZK_LINK_FIRST, v, outer_zero);
            state_in.push_back(v);
            s[i][b] = v;
        }
    // absorb: state[0..136) ^= block (keccak256_absorb_and_run_permutation, mod.rs:803-817); the xor lookup
    // range-checks both the carried byte and the fresh input byte
    for (int j = 0; j < 136; ++j) {
        zk_var in_byte = g.next_input();
        s[j / 8][j % 8] = k.xor8(s[j / 8][j % 8], in_byte);
    }
    // capacity bytes of the carried state are range-checked through a pair lookup (they enter theta's xor anyway,
    // but only as the first key: make the check explicit)
    for (int j = 136; j < 200; j += 2) g.range_check_u8_pair(s[j / 8][j % 8], s[(j + 1) / 8][(j + 1) % 8]);
    k.permutation(s);
    for (int i = 0; i < 25; ++i)
        for (int b = 0; b < 8; ++b) state_out.push_back(s[i][b]);
    for (size_t i = 0; i < 200; ++i) cs.link(ZK_LINK_CARRY, state_in[i], state_out[i]);
    cs.loop_end();
    for (int j = 0; j < 32; ++j) {
        zk_var d = cs.loop_last(state_out[j]);
        cs.place_gate(ZK_GATE_PUBLIC_INPUT, &d, 1, nullptr, 0);
    }
}

}  // namespace zkgl
