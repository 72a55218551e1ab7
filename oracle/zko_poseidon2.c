/* oracle/zko_poseidon2.c — CPU ORACLE (test infrastructure).
 *
 * Poseidon2 over Goldilocks, width 12 / rate 8 / capacity 4 — the round function every
 * circuit of the reference is generic over (`CircuitRoundFunction<F, 8, 12, 4>`,
 * e.g. /root/reference/src/utils.rs:15, src/ram_permutation/mod.rs:34) and that its tests
 * instantiate as `Poseidon2Goldilocks` (src/ram_permutation/mod.rs:411).
 *
 * The implementation lives in boojum (absent, [EXT]); restated from its published structure:
 *   permute = M_E ; 4 x full ; 22 x partial ; 4 x full
 *   full(r)    : s[i] += RC[12r+i] ; s[i] = s[i]^7 ; M_E
 *   partial(r) : s[0] += RC[12r]   ; s[0] = s[0]^7 ; M_I
 *   M_E = circ(2*M4, M4, M4), M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
 *   M_I = J + diag(2^k), k = {4,14,11,8,0,5,2,9,13,6,3,12}
 * RC = the 360 Poseidon-Goldilocks round constants (12 x 30) boojum shares with its
 * Poseidon; they are NOT typed in here: they are re-derived by the generation procedure
 * (ChaCha8Rng::seed_from_u64(0), then rand 0.8 `gen_range(0..p)` per constant) and the
 * derivation is pinned by tests/golden/poseidon_rc_known.json.
 */
#include "zko.h"
#include <string.h>

/* ---- ChaCha8 block function (RFC 7539 layout, 8 rounds, 64-bit counter, stream 0) ---- */
static uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d)                                                                             \
    do {                                                                                           \
        s[a] += s[b]; s[d] = rotl32(s[d] ^ s[a], 16);                                              \
        s[c] += s[d]; s[b] = rotl32(s[b] ^ s[c], 12);                                              \
        s[a] += s[b]; s[d] = rotl32(s[d] ^ s[a], 8);                                               \
        s[c] += s[d]; s[b] = rotl32(s[b] ^ s[c], 7);                                               \
    } while (0)

static void chacha8_block(const uint32_t key[8], uint64_t ctr, uint32_t out[16]) {
    uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; ++i) init[4 + i] = key[i];
    init[12] = (uint32_t)ctr; init[13] = (uint32_t)(ctr >> 32); init[14] = 0; init[15] = 0;
    uint32_t s[16];
    memcpy(s, init, sizeof s);
    for (int r = 0; r < 4; ++r) {
        QR(0, 4, 8, 12); QR(1, 5, 9, 13); QR(2, 6, 10, 14); QR(3, 7, 11, 15);
        QR(0, 5, 10, 15); QR(1, 6, 11, 12); QR(2, 7, 8, 13); QR(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = s[i] + init[i];
}

/* rand_core::SeedableRng::seed_from_u64: PCG32 expansion of the u64 into the 32-byte key */
static void seed_from_u64(uint64_t state, uint32_t key[8]) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
}

static uint64_t RC[360];
static int rc_ready = 0;

const uint64_t *zko_poseidon_round_constants(void) {
    if (rc_ready) return RC;
    uint32_t key[8];
    seed_from_u64(0, key);
    uint32_t blk[16];
    uint64_t ctr = 0;
    int have = 0, pos = 0, n = 0;
    while (n < 360) {
        if (pos == have) { chacha8_block(key, ctr++, blk); have = 16; pos = 0; }
        uint64_t v = (uint64_t)blk[pos] | ((uint64_t)blk[pos + 1] << 32);
        pos += 2;
        /* rand 0.8 UniformInt::sample_single for range p (leading_zeros == 0): widening
         * multiply, accept when the low word <= p - 1 */
        unsigned __int128 m = (unsigned __int128)v * ZKO_P;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo <= ZKO_P - 1) RC[n++] = hi;
    }
    rc_ready = 1;
    return RC;
}

static uint64_t sbox7(uint64_t x) {
    uint64_t x2 = zko_gl_mul(x, x), x3 = zko_gl_mul(x2, x), x4 = zko_gl_mul(x2, x2);
    return zko_gl_mul(x3, x4);
}

static const uint64_t M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
static const int INNER_SHIFTS[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};

void zko_poseidon2_mds_external(uint64_t s[12]) {
    uint64_t t[12];
    for (int b = 0; b < 3; ++b)
        for (int i = 0; i < 4; ++i) {
            uint64_t acc = 0;
            for (int j = 0; j < 4; ++j) acc = zko_gl_add(acc, zko_gl_mul(M4[i][j], s[4 * b + j]));
            t[4 * b + i] = acc;
        }
    for (int i = 0; i < 4; ++i) {
        uint64_t sum = zko_gl_add(zko_gl_add(t[i], t[4 + i]), t[8 + i]);
        for (int b = 0; b < 3; ++b) s[4 * b + i] = zko_gl_add(t[4 * b + i], sum);
    }
}

void zko_poseidon2_mds_inner(uint64_t s[12]) {
    uint64_t sum = 0;
    for (int i = 0; i < 12; ++i) sum = zko_gl_add(sum, s[i]);
    for (int i = 0; i < 12; ++i)
        s[i] = zko_gl_add(sum, zko_gl_mul(s[i], 1ULL << INNER_SHIFTS[i]));
}

void zko_poseidon2_permute(uint64_t s[12]) {
    const uint64_t *rc = zko_poseidon_round_constants();
    zko_poseidon2_mds_external(s);
    for (int r = 0; r < 30; ++r) {
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; ++i) s[i] = sbox7(zko_gl_add(s[i], rc[12 * r + i]));
            zko_poseidon2_mds_external(s);
        } else {
            s[0] = sbox7(zko_gl_add(s[0], rc[12 * r]));
            zko_poseidon2_mds_inner(s);
        }
    }
}

void zko_poseidon2_permute_batch(uint64_t *states, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) zko_poseidon2_permute(states + 12 * i);
}
