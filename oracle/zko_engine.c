/* oracle/zko_engine.c — CPU ORACLE (test infrastructure, never shipped, never timed as product).
 *
 * Independent CPU interpreter + checker for a serialised zkgl scope (zk_cs_export): executes the
 * witness IR of include/zkgl_ir.h lane by lane with the oracle's own field / Poseidon2 code and
 * evaluates every gate relation, lookup tuple, copy pair and link.  It is the CPU counterpart of
 * boojum's resolver + `check_if_satisfied` (/root/reference/src/ram_permutation/mod.rs:552-556)
 * and the `cpu_baseline` ("port") leg of bench.py.  OpenMP over lanes.
 */
#include <stdlib.h>
#include <string.h>
#include "zko.h"
#include "zkgl_ir.h"

typedef struct zko_scope {
    uint32_t is_loop, n_cells, n_trace_cells, n_slots, n_copy_cols, lookup_width, n_input_words, limit, pre_words;
    uint32_t n_prog, n_consts, n_rows, n_rowconsts, n_lrows, n_copies, n_tables, n_table_words, n_links, n_carries;
    uint32_t *carries; /* 4 words each: input word, out cell, first outer cell, has_first */
    uint32_t n_stream_words;
    uint32_t *streams; /* per stream link: pa, pb, n_total, a cells[pa], b cells[pb] */
    uint32_t *prog;
    uint64_t *consts;
    zk_row_desc *rows;
    uint64_t *rowconsts;
    zk_lookup_row_desc *lrows;
    zk_copy_pair *copies;
    zk_table_desc *tables;
    uint64_t *table_words;
    zk_link *links;
} zko_scope;

static uint64_t rd64(const uint32_t *p) { return (uint64_t)p[0] | ((uint64_t)p[1] << 32); }

void zko_scope_free(zko_scope *s) {
    if (!s) return;
    free(s->prog); free(s->consts); free(s->rows); free(s->rowconsts); free(s->lrows); free(s->copies);
    free(s->tables); free(s->table_words); free(s->links); free(s->carries); free(s->streams); free(s);
}

zko_scope *zko_scope_parse(const uint32_t *w, size_t n) {
    if (n < 21 || w[0] != 0x5a4b4733u) return NULL;
    zko_scope *s = calloc(1, sizeof *s);
    s->is_loop = w[1]; s->n_cells = w[2]; s->n_trace_cells = w[3]; s->n_slots = w[4]; s->n_copy_cols = w[5];
    s->lookup_width = w[6]; s->n_input_words = w[7]; s->limit = w[8]; s->pre_words = w[9]; s->n_prog = w[10];
    s->n_consts = w[11]; s->n_rows = w[12]; s->n_rowconsts = w[13]; s->n_lrows = w[14]; s->n_copies = w[15];
    s->n_tables = w[16]; s->n_table_words = w[17]; s->n_links = w[18]; s->n_carries = w[19]; s->n_stream_words = w[20];
    const uint32_t *p = w + 21;
#define TAKE(dst, type, count, words_each, conv)                                   \
    do {                                                                          \
        s->dst = malloc(sizeof(type) * ((count) ? (count) : 1));                  \
        for (uint32_t i = 0; i < (count); ++i) { conv; p += (words_each); }       \
    } while (0)
    TAKE(prog, uint32_t, s->n_prog, 1, s->prog[i] = p[0]);
    TAKE(consts, uint64_t, s->n_consts, 2, s->consts[i] = rd64(p));
    TAKE(rows, zk_row_desc, s->n_rows, 4, (s->rows[i].kind = p[0], s->rows[i].n_instances = p[1], s->rows[i].const_off = p[2], s->rows[i].n_consts = p[3]));
    TAKE(rowconsts, uint64_t, s->n_rowconsts, 2, s->rowconsts[i] = rd64(p));
    TAKE(lrows, zk_lookup_row_desc, s->n_lrows, 2, (s->lrows[i].table = p[0], s->lrows[i].n_tuples = p[1]));
    TAKE(copies, zk_copy_pair, s->n_copies, 2, (s->copies[i].cell = p[0], s->copies[i].home = p[1]));
    TAKE(tables, zk_table_desc, s->n_tables, 9,
         (s->tables[i].word_off = p[0], s->tables[i].mult_off = p[1], s->tables[i].n_rows = p[2], s->tables[i].n_keys = p[3],
          s->tables[i].n_vals = p[4], s->tables[i].dense = p[5], s->tables[i].key_shift[0] = p[6],
          s->tables[i].key_shift[1] = p[7], s->tables[i].key_shift[2] = p[8]));
    TAKE(table_words, uint64_t, s->n_table_words, 2, s->table_words[i] = rd64(p));
    TAKE(links, zk_link, s->n_links, 4, (s->links[i].kind = p[0], s->links[i].loop_cell = p[1], s->links[i].other_cell = p[2], s->links[i].pad = 0));
    s->carries = malloc(sizeof(uint32_t) * 4 * (s->n_carries ? s->n_carries : 1));
    for (uint32_t i = 0; i < 4 * s->n_carries; ++i) s->carries[i] = *p++;
    s->streams = malloc(sizeof(uint32_t) * (s->n_stream_words ? s->n_stream_words : 1));
    for (uint32_t i = 0; i < s->n_stream_words; ++i) s->streams[i] = *p++;
#undef TAKE
    if ((size_t)(p - w) != n) { zko_scope_free(s); return NULL; }
    return s;
}

/* getters for the python side */
uint32_t zko_scope_field(const zko_scope *s, int which) {
    switch (which) {
    case 0: return s->is_loop; case 1: return s->n_cells; case 2: return s->n_trace_cells; case 3: return s->n_slots;
    case 4: return s->n_input_words; case 5: return s->limit; case 6: return s->pre_words; case 7: return s->n_prog;
    case 8: return s->n_copies; case 9: return s->n_links; case 10: return s->n_copy_cols; case 11: return s->lookup_width;
    case 12: return s->n_carries;
    default: return 0;
    }
}

/* linear scan / binary search written independently of the device code: plain linear probe over
 * the sorted table (tables are <= 2^16 rows; the oracle favours obviousness over speed) with a
 * binary search fast path validated against it in tests. */
static uint32_t table_find(const zko_scope *s, const zk_table_desc *t, const uint64_t *key) {
    const uint32_t w = t->n_keys + t->n_vals;
    const uint64_t *rows = s->table_words + t->word_off;
    uint32_t lo = 0, hi = t->n_rows;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        int c = 0;
        for (uint32_t i = 0; i < t->n_keys && !c; ++i) {
            uint64_t r = rows[(size_t)mid * w + i];
            c = (r > key[i]) - (r < key[i]);
        }
        if (!c) return mid;
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return t->n_rows;
}

typedef struct {
    const zko_scope *s;
    uint64_t *cells; size_t stride; uint32_t n_lanes;
    const uint64_t *inputs;
    const uint64_t *outer_cells; size_t outer_stride;
    const uint64_t *loop_cells; size_t loop_stride; uint32_t loop_limit;
    uint32_t *mult; uint32_t total_rows;
} run_ctx;

static uint64_t ld(const run_ctx *c, uint32_t w, uint32_t lane, uint32_t inst) {
    uint32_t kind = w & ZK_OPERAND_KIND_MASK, idx = w & ZK_OPERAND_IDX_MASK;
    if (kind == ZK_OPERAND_CONST) return c->s->consts[idx];
    if (kind == ZK_OPERAND_OUTER) return c->outer_cells[(size_t)idx * c->outer_stride + inst];
    return c->cells[(size_t)idx * c->stride + lane];
}
static void st(const run_ctx *c, const uint32_t *prog, uint32_t *pc, uint32_t lane, uint64_t v) {
    uint32_t w;
    do {
        w = prog[(*pc)++];
        c->cells[(size_t)(w & ZK_DEST_CELL_MASK) * c->stride + lane] = v;
    } while (w & ZK_DEST_MORE);
}

static uint64_t pow7(uint64_t x) {
    uint64_t x2 = zko_gl_mul(x, x), x3 = zko_gl_mul(x2, x), x4 = zko_gl_mul(x2, x2);
    return zko_gl_mul(x3, x4);
}

/* ZK_OP_KECCAK_F (include/zkgl_ir.h): Keccak-f[1600] with every intermediate of the byte-table decomposition written out, restated
 * in plain C from the output order documented in era-zkevm_circuits_amd/csrc/keccak_macro.hpp (per round: theta column xors, the five
 * rotl-by-1 + xor, the 25 state xors; rho-pi as the in-place chain; chi row by row; iota) — CPU ORACLE, test infrastructure. */
static const uint64_t KECCAK_RC_O[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_RHO_O[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KECCAK_PI_O[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
typedef struct { uint64_t *buf; size_t n; } kk_out;
static void kk_emit8(kk_out *o, uint64_t v) { for (int k = 0; k < 8; ++k) o->buf[o->n++] = (v >> (8 * k)) & 0xff; }
static uint64_t kk_rotl(kk_out *o, uint64_t a, int n) {
    n %= 64;
    int b = n % 8;
    uint64_t full = n ? (a << n) | (a >> (64 - n)) : a;
    if (!b) return full;
    for (int k = 0; k < 8; ++k) {
        uint64_t byte = (a >> (8 * k)) & 0xff;
        o->buf[o->n++] = byte & ((1u << (8 - b)) - 1);
        o->buf[o->n++] = byte >> (8 - b);
    }
    kk_emit8(o, (a << b) | (a >> (64 - b)));
    return full;
}
static void kk_keccak_f(uint64_t s[25], kk_out *o) {
    for (int rnd = 0; rnd < 24; ++rnd) {
        uint64_t cc[5], d[5];
        for (int x = 0; x < 5; ++x) {
            cc[x] = s[x];
            for (int j = 1; j < 5; ++j) { cc[x] ^= s[x + 5 * j]; kk_emit8(o, cc[x]); }
        }
        for (int x = 0; x < 5; ++x) { uint64_t r = kk_rotl(o, cc[(x + 1) % 5], 1); d[x] = cc[(x + 4) % 5] ^ r; kk_emit8(o, d[x]); }
        for (int i = 0; i < 25; ++i) { s[i] ^= d[i % 5]; kk_emit8(o, s[i]); }
        uint64_t t = s[1];
        for (int i = 0; i < 24; ++i) { int j = KECCAK_PI_O[i]; uint64_t bc = s[j]; s[j] = kk_rotl(o, t, KECCAK_RHO_O[i]); t = bc; }
        for (int y = 0; y < 5; ++y) {
            uint64_t r[5];
            for (int x = 0; x < 5; ++x) r[x] = s[x + 5 * y];
            for (int x = 0; x < 5; ++x) {
                uint64_t a = ~r[(x + 1) % 5] & r[(x + 2) % 5];
                kk_emit8(o, a);
                s[x + 5 * y] = r[x] ^ a;
                kk_emit8(o, s[x + 5 * y]);
            }
        }
        uint64_t nc = s[0] ^ KECCAK_RC_O[rnd];
        for (int k = 0; k < 8; ++k) if ((KECCAK_RC_O[rnd] >> (8 * k)) & 0xff) o->buf[o->n++] = (nc >> (8 * k)) & 0xff;
        s[0] = nc;
    }
}

/* ZK_OP_SHA256_ROUNDS (include/zkgl_ir.h): one SHA-256 compression with every intermediate of the byte-table decomposition written
 * out, restated in plain C from the output order documented in era-zkevm_circuits_amd/csrc/sha256_macro.hpp — CPU ORACLE. */
static const uint32_t SHA_K_O[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static void sh_emit4(kk_out *o, uint32_t v) { for (int k = 0; k < 4; ++k) o->buf[o->n++] = (v >> (8 * k)) & 0xff; }
static uint32_t sh_bw(kk_out *o, int t, uint32_t a, uint32_t b) { uint32_t r = t == 0 ? a ^ b : t == 1 ? a & b : ~a & b; sh_emit4(o, r); return r; }
static uint32_t sh_xor3(kk_out *o, uint32_t a, uint32_t b, uint32_t c) { return sh_bw(o, 0, sh_bw(o, 0, a, b), c); }
static void sh_split(kk_out *o, uint32_t a, int at) {
    for (int k = 0; k < 4; ++k) { uint32_t byte = (a >> (8 * k)) & 0xff; o->buf[o->n++] = byte & ((1u << at) - 1); o->buf[o->n++] = byte >> at; }
}
static uint32_t sh_rotr(kk_out *o, uint32_t a, int n) {
    uint32_t r = (a >> n) | (a << ((32 - n) & 31));
    if (n % 8) { sh_split(o, a, n % 8); sh_emit4(o, r); }
    return r;
}
static uint32_t sh_shr(kk_out *o, uint32_t a, int n) {
    int q = n / 8, b = n % 8;
    uint32_t r = a >> n;
    if (b) { sh_split(o, a, b); for (int k = 0; k < 4; ++k) if (k + q + 1 < 4) o->buf[o->n++] = (r >> (8 * k)) & 0xff; }
    return r;
}
static uint32_t sh_add(kk_out *o, const uint32_t *w, int nw, uint64_t c) {
    int T = 4 * nw + (c ? 1 : 0);
    uint64_t acc = 0;
    for (int t = 0; t < T; ++t) {
        acc += t < 4 * nw ? (uint64_t)((w[t / 4] >> (8 * (t % 4))) & 0xff) << (8 * (t % 4)) : c;
        if (t == 3 || (t > 3 && (t - 4) % 3 == 2) || (t == T - 1 && t > 3)) o->buf[o->n++] = acc;   /* one reduction gate per 4, then per 3 terms */
    }
    uint32_t low = (uint32_t)acc;
    sh_emit4(o, low);
    o->buf[o->n++] = acc >> 32;
    o->buf[o->n++] = low;
    o->buf[o->n++] = acc;
    o->buf[o->n++] = 0;
    return low;
}
static void sh_rc(kk_out *o, uint32_t w) { o->buf[o->n++] = (w ^ (w >> 8)) & 0xff; o->buf[o->n++] = ((w >> 16) ^ (w >> 24)) & 0xff; }
static void sh_compress(uint32_t st[8], const uint32_t blk[16], kk_out *o) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = blk[i];
    for (int i = 16; i < 64; ++i) {
        uint32_t a7 = sh_rotr(o, w[i - 15], 7), a18 = sh_rotr(o, w[i - 15], 18), a3 = sh_shr(o, w[i - 15], 3);
        uint32_t s0 = sh_xor3(o, a7, a18, a3);
        uint32_t b17 = sh_rotr(o, w[i - 2], 17), b19 = sh_rotr(o, w[i - 2], 19), b10 = sh_shr(o, w[i - 2], 10);
        uint32_t s1 = sh_xor3(o, b17, b19, b10);
        uint32_t t[4] = {w[i - 16], s0, w[i - 7], s1};
        w[i] = sh_add(o, t, 4, 0);
    }
    sh_rc(o, w[62]); sh_rc(o, w[63]);
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t e6 = sh_rotr(o, e, 6), e11 = sh_rotr(o, e, 11), e25 = sh_rotr(o, e, 25);
        uint32_t S1 = sh_xor3(o, e6, e11, e25);
        uint32_t ef = sh_bw(o, 1, e, f), ng = sh_bw(o, 2, e, g), ch = sh_bw(o, 0, ef, ng);
        uint32_t a2 = sh_rotr(o, a, 2), a13 = sh_rotr(o, a, 13), a22 = sh_rotr(o, a, 22);
        uint32_t S0 = sh_xor3(o, a2, a13, a22);
        uint32_t ab = sh_bw(o, 1, a, b), axb = sh_bw(o, 0, a, b), cx = sh_bw(o, 1, c, axb), maj = sh_bw(o, 0, ab, cx);
        uint32_t t1[5] = {d, h, S1, ch, w[i]}, t2[6] = {h, S1, ch, w[i], S0, maj};
        uint32_t ne = sh_add(o, t1, 5, SHA_K_O[i]);
        uint32_t na = sh_add(o, t2, 6, SHA_K_O[i]);
        h = g; g = f; f = e; e = ne; d = c; c = b; b = a; a = na;
    }
    sh_rc(o, a); sh_rc(o, e);
    uint32_t out[8] = {a, b, c, d, e, f, g, h};
    for (int i = 0; i < 8; ++i) {
        uint32_t t[2] = {st[i], out[i]};
        st[i] = sh_add(o, t, 2, 0);
        sh_rc(o, st[i]);
    }
}

/* ZK_OP_SHA256_ROUNDS with a = 1 (include/zkgl_ir.h): one SHA-256 compression over 4-bit chunks through the reference's table set (Maj4 / TriXor4 /
 * Ch4 / Split4BitChunk<1,2>, /root/reference/src/code_unpacker_sha256/mod.rs:554-566), every intermediate written out — restated in plain C from the
 * output order documented in era-zkevm_circuits_amd/csrc/sha256_macro4.hpp.  CPU ORACLE, test infrastructure. */
typedef struct { kk_out *o; uint64_t acc; uint32_t terms; uint8_t loose[320]; uint32_t n_loose; } s4_ctx;
static uint32_t s4_nib(uint32_t w, int j) { return (w >> (4 * j)) & 15u; }
static void s4_put(s4_ctx *x, uint64_t v) { x->o->buf[x->o->n++] = v; }
static uint32_t s4_from_bytes(s4_ctx *x, uint32_t b) { for (int j = 0; j < 8; ++j) s4_put(x, s4_nib(b, j)); s4_put(x, b); return b; }
static void s4_rows(s4_ctx *x, uint32_t w, int at) {   /* Split4BitChunk<at> rows of the eight chunks of w: (low, high, swapped halves) */
    for (int j = 0; j < 8; ++j) { uint32_t c = s4_nib(w, j), lo = c & ((1u << at) - 1), hi = c >> at; s4_put(x, lo); s4_put(x, hi); s4_put(x, (lo << (4 - at)) | hi); }
}
static uint32_t s4_rot(s4_ctx *x, uint32_t w, uint8_t *have, int r, int shift_only) {
    int q = r / 4, s = r % 4;
    if (s && !(*have & (1 << s))) {
        if (s == 2) s4_rows(x, w, 2);
        else {
            if (!(*have & 2)) s4_rows(x, w, 1);
            *have |= 2;
            if (s == 3)
                for (int j = 0; j < 8; ++j) {
                    uint32_t c = s4_nib(w, j), h1 = c >> 1, lo = h1 & 3u, hi = h1 >> 2;
                    s4_put(x, lo); s4_put(x, hi); s4_put(x, (lo << 2) | hi);
                    s4_put(x, 2 * lo + (c & 1u));
                }
        }
        *have |= (uint8_t)(1 << s);
    }
    uint32_t res = shift_only ? w >> r : (w >> r) | (w << (32 - r));
    if (s)
        for (int i = 0; i < 8; ++i) {
            int j = i + q;
            if (shift_only && (j >= 8 || j + 1 >= 8)) continue;   /* zero, or the bare high part: no gate */
            uint32_t hi = s4_nib(w, j % 8) >> s, lo = s4_nib(w, (j + 1) % 8) & ((1u << s) - 1);
            s4_put(x, hi + (lo << (4 - s)));
        }
    return res;
}
static uint32_t s4_tri(s4_ctx *x, int t, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t p = s4_nib(a, i), q = s4_nib(b, i), z = s4_nib(c, i);
        uint32_t v = t == 0 ? p ^ q ^ z : t == 1 ? (p & q) ^ (~p & 0xfu & z) : (p & q) ^ (p & z) ^ (q & z);
        s4_put(x, v);
        r |= v << (4 * i);
    }
    return r;
}
static void s4_term(s4_ctx *x, uint64_t v) { x->acc += v; ++x->terms; if (x->terms == 4 || (x->terms > 4 && (x->terms - 4) % 3 == 0)) s4_put(x, x->acc); }
static void s4_nibs(s4_ctx *x, uint32_t n) { for (int i = 0; i < 8; ++i) s4_term(x, (uint64_t)s4_nib(n, i) << (4 * i)); }
static uint64_t s4_end(s4_ctx *x) { if (x->terms < 4 || (x->terms - 4) % 3 != 0) s4_put(x, x->acc); return x->acc; }
static uint32_t s4_add(s4_ctx *x, uint64_t *carry) {
    uint64_t sum = s4_end(x);
    uint32_t low = (uint32_t)sum;
    for (int j = 0; j < 8; ++j) s4_put(x, s4_nib(low, j));
    s4_put(x, sum >> 32);
    s4_put(x, low & 0xffffu); s4_put(x, low & 0xfffffffu); s4_put(x, low);
    s4_put(x, ((sum >> 32) << 32) + low);
    *carry = sum >> 32;
    return low;
}
static void s4_loose(s4_ctx *x, uint32_t v) { x->loose[x->n_loose++] = (uint8_t)(v & 15u); }
static void s4_loose8(s4_ctx *x, uint32_t w) { for (int i = 0; i < 8; ++i) s4_loose(x, s4_nib(w, i)); }
static void sh4_compress(uint32_t st[8], const uint32_t blk[16], kk_out *o) {
    s4_ctx X = {o, 0, 0, {0}, 0}, *x = &X;
    uint32_t w[64]; uint8_t have[64] = {0};
    uint64_t cy;
    for (int i = 0; i < 16; ++i) w[i] = s4_from_bytes(x, blk[i]);
    s4_loose8(x, w[0]);
    for (int i = 16; i < 64; ++i) {
        uint32_t a7 = s4_rot(x, w[i - 15], &have[i - 15], 7, 0), a18 = s4_rot(x, w[i - 15], &have[i - 15], 18, 0), a3 = s4_rot(x, w[i - 15], &have[i - 15], 3, 1);
        uint32_t s0 = s4_tri(x, 0, a7, a18, a3);
        uint32_t b17 = s4_rot(x, w[i - 2], &have[i - 2], 17, 0), b19 = s4_rot(x, w[i - 2], &have[i - 2], 19, 0), b10 = s4_rot(x, w[i - 2], &have[i - 2], 10, 1);
        uint32_t s1 = s4_tri(x, 0, b17, b19, b10);
        x->acc = 0; x->terms = 0;
        s4_term(x, w[i - 16]); s4_term(x, w[i - 7]); s4_nibs(x, s0); s4_nibs(x, s1);
        w[i] = s4_add(x, &cy);
        s4_loose(x, (uint32_t)cy);
    }
    s4_loose8(x, w[62]); s4_loose8(x, w[63]);
    uint32_t s[8];
    for (int i = 0; i < 8; ++i) s[i] = s4_from_bytes(x, st[i]);
    uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    for (int i = 0; i < 64; ++i) {
        uint8_t he = 0, ha = 0;
        uint32_t e6 = s4_rot(x, e, &he, 6, 0), e11 = s4_rot(x, e, &he, 11, 0), e25 = s4_rot(x, e, &he, 25, 0);
        uint32_t S1 = s4_tri(x, 0, e6, e11, e25);
        uint32_t ch = s4_tri(x, 1, e, f, g);
        uint32_t a2 = s4_rot(x, a, &ha, 2, 0), a13 = s4_rot(x, a, &ha, 13, 0), a22 = s4_rot(x, a, &ha, 22, 0);
        uint32_t S0 = s4_tri(x, 0, a2, a13, a22);
        uint32_t mj = s4_tri(x, 2, a, b, c);
        x->acc = 0; x->terms = 0;
        s4_term(x, h); s4_term(x, w[i]); s4_term(x, SHA_K_O[i]); s4_nibs(x, S1); s4_nibs(x, ch);
        uint64_t T1 = s4_end(x), c1, c2;
        x->acc = 0; x->terms = 0;
        s4_term(x, d); s4_term(x, T1);
        uint32_t ne = s4_add(x, &c1);
        x->acc = 0; x->terms = 0;
        s4_term(x, T1); s4_nibs(x, S0); s4_nibs(x, mj);
        uint32_t na = s4_add(x, &c2);
        s4_loose(x, (uint32_t)c1); s4_loose(x, (uint32_t)c2);
        h = g; g = f; f = e; e = ne; d = c; c = b; b = a; a = na;
    }
    s4_loose8(x, a); s4_loose8(x, e);
    uint32_t out[8] = {a, b, c, d, e, f, g, h};
    for (int i = 0; i < 8; ++i) {
        x->acc = 0; x->terms = 0;
        s4_term(x, s[i]); s4_term(x, out[i]);
        uint32_t r = s4_add(x, &cy);
        s4_loose(x, (uint32_t)cy); s4_loose8(x, r);
        for (int k = 0; k < 4; ++k) s4_put(x, (r >> (8 * k)) & 0xff);
        st[i] = r;
    }
    for (uint32_t i = 0; i < x->n_loose; i += 3)
        s4_put(x, x->loose[i] ^ (i + 1 < x->n_loose ? x->loose[i + 1] : 0) ^ (i + 2 < x->n_loose ? x->loose[i + 2] : 0));
}

/* the output stream of one ZK_OP_SHA256_ROUNDS (a = 0: 8-bit tables, 1: the reference's 4-bit-chunk tables) for a state and a block given as
 * little-endian byte words; st is updated; returns the number of outputs (out: capacity >= 32768).  Used by the tests that compare the
 * device backends (host-compiled) with this restatement value by value. */
size_t zko_sha256_rounds_stream(uint32_t a, uint32_t st[8], const uint32_t blk[16], uint64_t *out) {
    kk_out o = {out, 0};
    if (a == 1) sh4_compress(st, blk, &o); else sh_compress(st, blk, &o);
    return o.n;
}

static int run_lane(const run_ctx *c, uint32_t lane, uint32_t wb, uint32_t we) {
    const zko_scope *s = c->s;
    const uint32_t *prog = s->prog;
    const uint64_t *RC = zko_poseidon_round_constants();
    uint32_t inst = s->is_loop ? lane / s->limit : lane;
    uint32_t pc = wb;
    while (pc < we) {
        uint32_t h = prog[pc++], op = h & 0xff, pa = (h >> 8) & 0xff, pb = h >> 16;
        switch (op) {
        case ZK_OP_CONST: { uint64_t v = ld(c, prog[pc++], lane, inst); st(c, prog, &pc, lane, v); } break;
        case ZK_OP_INPUT: { uint32_t w = prog[pc++]; st(c, prog, &pc, lane, c->inputs[(size_t)w * c->n_lanes + lane]); } break;
        case ZK_OP_FMA: {
            uint64_t q = ld(c, prog[pc], lane, inst), l = ld(c, prog[pc + 1], lane, inst);
            uint64_t a = ld(c, prog[pc + 2], lane, inst), b = ld(c, prog[pc + 3], lane, inst), cc = ld(c, prog[pc + 4], lane, inst);
            pc += 5;
            st(c, prog, &pc, lane, zko_gl_add(zko_gl_mul(q, zko_gl_mul(a, b)), zko_gl_mul(l, cc)));
        } break;
        case ZK_OP_LC4: {
            uint64_t r = 0;
            for (int i = 0; i < 4; ++i) r = zko_gl_add(r, zko_gl_mul(ld(c, prog[pc + i], lane, inst), ld(c, prog[pc + 4 + i], lane, inst)));
            pc += 8;
            st(c, prog, &pc, lane, r);
        } break;
        case ZK_OP_SELECT: {
            uint64_t sel = ld(c, prog[pc], lane, inst), a = ld(c, prog[pc + 1], lane, inst), b = ld(c, prog[pc + 2], lane, inst);
            pc += 3;
            st(c, prog, &pc, lane, sel ? a : b);
        } break;
        case ZK_OP_ISZERO: {
            uint64_t x = ld(c, prog[pc++], lane, inst);
            st(c, prog, &pc, lane, x == 0);
            st(c, prog, &pc, lane, zko_gl_inv(x));
        } break;
        case ZK_OP_UADD: {
            uint64_t x = ld(c, prog[pc], lane, inst), y = ld(c, prog[pc + 1], lane, inst), ci = ld(c, prog[pc + 2], lane, inst);
            pc += 3;
            uint64_t sum = x + y + ci;
            st(c, prog, &pc, lane, sum % (1ull << pa));
            st(c, prog, &pc, lane, sum >> pa);
        } break;
        case ZK_OP_USUB: {
            uint64_t x = ld(c, prog[pc], lane, inst), y = ld(c, prog[pc + 1], lane, inst), bi = ld(c, prog[pc + 2], lane, inst);
            pc += 3;
            int borrow = x < y + bi;
            st(c, prog, &pc, lane, borrow ? x + (1ull << pa) - y - bi : x - y - bi);
            st(c, prog, &pc, lane, (uint64_t)borrow);
        } break;
        case ZK_OP_DOT4: {
            uint64_t r = 0;
            for (int i = 0; i < 4; ++i) r = zko_gl_add(r, zko_gl_mul(ld(c, prog[pc + 2 * i], lane, inst), ld(c, prog[pc + 2 * i + 1], lane, inst)));
            pc += 8;
            st(c, prog, &pc, lane, r);
        } break;
        case ZK_OP_MATMUL12: {
            uint64_t v[12];
            for (int i = 0; i < 12; ++i) v[i] = ld(c, prog[pc + i], lane, inst);
            pc += 12;
            if (pa == 0) zko_poseidon2_mds_external(v); else zko_poseidon2_mds_inner(v);
            for (int i = 0; i < 12; ++i) st(c, prog, &pc, lane, v[i]);
        } break;
        case ZK_OP_SPLIT: {
            uint64_t x = ld(c, prog[pc++], lane, inst);
            for (uint32_t i = 0; i < pa; ++i) {
                st(c, prog, &pc, lane, i + 1 == pa ? x : x % (1ull << pb));
                x >>= pb;
            }
        } break;
        case ZK_OP_LOOKUP: {
            uint32_t tid = prog[pc++];
            const zk_table_desc *t = &s->tables[tid];
            uint64_t key[3] = {0, 0, 0};
            for (uint32_t i = 0; i < pa; ++i) key[i] = ld(c, prog[pc + i], lane, inst);
            pc += pa;
            uint32_t row = table_find(s, t, key);
            uint32_t w = t->n_keys + t->n_vals;
            for (uint32_t i = 0; i < pb; ++i)
                st(c, prog, &pc, lane, row < t->n_rows ? s->table_words[t->word_off + (size_t)row * w + t->n_keys + i] : 0);
            if (row < t->n_rows && c->mult) {
#pragma omp atomic
                c->mult[(size_t)inst * c->total_rows + t->mult_off + row] += 1;
            }
        } break;
        case ZK_OP_POSEIDON2: {
            uint64_t v[12];
            for (int i = 0; i < 12; ++i) v[i] = ld(c, prog[pc + i], lane, inst);
            pc += 12;
            if (pa) { /* gated: simulate_round_function(cs, state, execute) yields zeros when the flag is off */
                uint64_t execute = ld(c, prog[pc], lane, inst);
                pc += 1;
                if (!execute) { for (int i = 0; i < 12; ++i) st(c, prog, &pc, lane, 0); break; }
            }
            zko_poseidon2_permute(v);
            for (int i = 0; i < 12; ++i) st(c, prog, &pc, lane, v[i]);
        } break;
        case ZK_OP_P2_ROUNDS: {
            uint64_t v[12];
            for (int i = 0; i < 12; ++i) v[i] = ld(c, prog[pc + i], lane, inst);
            pc += 12;
            zko_poseidon2_mds_external(v);
            for (int i = 0; i < 12; ++i) st(c, prog, &pc, lane, v[i]);
            for (int r = 0; r < 30; ++r) {
                int full = r < 4 || r >= 26, n = full ? 12 : 1;
                for (int i = 0; i < n; ++i) {
                    uint64_t t = zko_gl_add(v[i], RC[12 * r + i]);
                    uint64_t x2 = zko_gl_mul(t, t), x3 = zko_gl_mul(x2, t), x4 = zko_gl_mul(x2, x2), x7 = zko_gl_mul(x3, x4);
                    st(c, prog, &pc, lane, t); st(c, prog, &pc, lane, x2); st(c, prog, &pc, lane, x3);
                    st(c, prog, &pc, lane, x4); st(c, prog, &pc, lane, x7);
                    v[i] = x7;
                }
                if (full) zko_poseidon2_mds_external(v); else zko_poseidon2_mds_inner(v);
                for (int i = 0; i < 12; ++i) st(c, prog, &pc, lane, v[i]);
            }
            (void)pow7;
        } break;
        case ZK_OP_LOOP_LAST: {
            uint32_t cell = prog[pc++];
            st(c, prog, &pc, lane, c->loop_cells[(size_t)cell * c->loop_stride + (size_t)lane * c->loop_limit + c->loop_limit - 1]);
        } break;
        case ZK_OP_U32MULADD: {
            uint64_t a = ld(c, prog[pc], lane, inst), b = ld(c, prog[pc + 1], lane, inst), cc = ld(c, prog[pc + 2], lane, inst), d = ld(c, prog[pc + 3], lane, inst);
            pc += 4;
            uint64_t r = a * b + cc + d;
            st(c, prog, &pc, lane, r & 0xffffffffull);
            st(c, prog, &pc, lane, r >> 32);
        } break;
        case ZK_OP_U8X4FMA: { /* include/zkgl_ir.h: a*b + c + d over bytes, schoolbook over the 16 byte products; wrapping u64 arithmetic
                               * like the device (defined for any operands, the gate judges them) */
            uint64_t x[16];
            for (int i = 0; i < 16; ++i) x[i] = ld(c, prog[pc + i], lane, inst);
            pc += 16;
            uint64_t cc = 0, dd = 0, low = 0, high = 0;
            for (int i = 0; i < 4; ++i) { cc += x[8 + i] << (8 * i); dd += x[12 + i] << (8 * i); }
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    uint64_t pr = x[i] * x[4 + j];
                    if (i + j < 4) low += pr << (8 * (i + j)); else high += pr << (8 * (i + j - 4));
                }
            uint64_t X = low + cc + dd, kk = X >> 32, r_lo = X & 0xffffffffull, r_hi = high + kk;
            /* the device forms r = a*b + c + d in one u64; identical whenever the operands are bytes (r_hi < 2^32).  For operands
             * that are not, reproduce its wrapping result so that traces stay comparable: */
            {
                uint64_t a = x[0] + (x[1] << 8) + (x[2] << 16) + (x[3] << 24), b = x[4] + (x[5] << 8) + (x[6] << 16) + (x[7] << 24);
                uint64_t r = a * b + cc + dd;
                uint64_t t = x[0] * b + ((x[1] * (b & 0xffffffull)) << 8) + ((x[2] * (b & 0xffffull)) << 16) + ((x[3] * (b & 0xffull)) << 24);
                uint64_t k2 = (t + cc + dd) >> 32;
                int bytes_ok = 1;
                for (int i = 0; i < 16; ++i) bytes_ok &= x[i] < 256;
                if (bytes_ok && (r != (r_lo | (r_hi << 32)) || k2 != kk)) return -1; /* the two formulations must agree on bytes */
                r_lo = r & 0xffffffffull; r_hi = r >> 32; kk = k2;
            }
            for (int i = 0; i < 4; ++i) st(c, prog, &pc, lane, (r_lo >> (8 * i)) & 0xff);
            for (int i = 0; i < 4; ++i) st(c, prog, &pc, lane, (r_hi >> (8 * i)) & 0xff);
            st(c, prog, &pc, lane, kk & 0xff);
            st(c, prog, &pc, lane, (kk >> 8) & 0xff);
        } break;
        case ZK_OP_SHA256_ROUNDS: {
            uint32_t wd[24] = {0};
            for (int j = 0; j < 96; ++j) wd[j / 4] |= (uint32_t)(ld(c, prog[pc + j], lane, inst) & 0xff) << (8 * (j % 4));
            pc += 96;
            static _Thread_local uint64_t sbuf[32768];
            kk_out o = {sbuf, 0};
            if (pa == 1) sh4_compress(wd, wd + 8, &o); else sh_compress(wd, wd + 8, &o);
            for (size_t i = 0; i < o.n; ++i) st(c, prog, &pc, lane, sbuf[i]);
        } break;
        case ZK_OP_BYTEBUF_FILL: { /* include/zkgl_ir.h: ByteBuffer::fill_with_bytes (/root/reference/src/keccak256_round_function/buffer/mod.rs:69-136)
                                     * restated over field elements, every intermediate written out in the order of its gates: x - 1 (FMA),
                                     * is-zero flag + inverse (ZeroCheck), selections, and / or / not of flags (FMA), the masked source bytes */
            enum { BUF = 192, IN = 32 };
            uint64_t bytes[BUF], input[IN], shifted[IN], place[BUF];
            for (int j = 0; j < BUF; ++j) bytes[j] = ld(c, prog[pc + j], lane, inst);
            uint64_t filled = ld(c, prog[pc + BUF], lane, inst);
            for (int j = 0; j < IN; ++j) input[j] = ld(c, prog[pc + BUF + 1 + j], lane, inst);
            const uint64_t offset = ld(c, prog[pc + BUF + 1 + IN], lane, inst), meaningful = ld(c, prog[pc + BUF + 2 + IN], lane, inst);
            pc += BUF + IN + 3;
#define BB_OUT(v) st(c, prog, &pc, lane, (v))
            for (int j = 0; j < IN; ++j) shifted[j] = input[j];
            uint64_t off = zko_gl_sub(offset, 1);
            BB_OUT(off);
            for (int i = 1; i < IN; ++i) {
                const uint64_t use = off == 0;
                BB_OUT(use); BB_OUT(zko_gl_inv(off));
                off = zko_gl_sub(off, 1);
                BB_OUT(off);
                for (int j = 0; j < IN; ++j) {
                    shifted[j] = use ? (i + j < IN ? input[i + j] : 0) : shifted[j];
                    BB_OUT(shifted[j]);
                }
            }
            const uint64_t nothing = meaningful == 0;
            BB_OUT(nothing); BB_OUT(zko_gl_inv(meaningful));
            const uint64_t marker = zko_gl_sub(1, nothing);
            BB_OUT(marker);
            uint64_t tmp = filled;
            for (int j = 0; j < BUF; ++j) {
                const uint64_t here = tmp == 0;
                BB_OUT(here); BB_OUT(zko_gl_inv(tmp));
                place[j] = zko_gl_mul(here, marker);
                BB_OUT(place[j]);
                tmp = zko_gl_sub(tmp, 1);
                BB_OUT(tmp);
            }
            uint64_t counter = meaningful, exhausted = meaningful == 0;
            BB_OUT(exhausted); BB_OUT(zko_gl_inv(meaningful));
            for (int idx = 0; idx < IN; ++idx) {
                const uint64_t live = zko_gl_sub(1, exhausted);
                BB_OUT(live);
                const uint64_t src = zko_gl_mul(shifted[idx], live);
                BB_OUT(src);
                for (int j = idx; j < BUF; ++j) {
                    bytes[j] = place[j - idx] ? src : bytes[j];
                    BB_OUT(bytes[j]);
                }
                counter = zko_gl_sub(counter, 1);
                BB_OUT(counter);
                const uint64_t done = counter == 0;
                BB_OUT(done); BB_OUT(zko_gl_inv(counter));
                const uint64_t sum = zko_gl_add(done, exhausted);
                BB_OUT(sum);
                exhausted = zko_gl_sub(sum, zko_gl_mul(done, exhausted));
                BB_OUT(exhausted);
            }
            filled = zko_gl_add(filled, meaningful);
            BB_OUT(filled);
#undef BB_OUT
        } break;
        case ZK_OP_KECCAK_F: {
            uint64_t st8[25] = {0};
            for (int j = 0; j < 200; ++j) st8[j / 8] |= (ld(c, prog[pc + j], lane, inst) & 0xff) << (8 * (j % 8));
            pc += 200;
            static _Thread_local uint64_t kbuf[40000];
            kk_out o = {kbuf, 0};
            kk_keccak_f(st8, &o);
            for (size_t i = 0; i < o.n; ++i) st(c, prog, &pc, lane, kbuf[i]);
        } break;
        case ZK_OP_NN_MULMOD: { /* A*B = q*M + r, base 2^16: schoolbook product + bit-serial restoring division */
            uint64_t prod[40] = {0}, rem[18] = {0}, mod[16];
            uint32_t nq = pa + pb - 15, np = pa + pb + 2;
            for (uint32_t i = 0; i < 16; ++i) mod[i] = ld(c, prog[pc + i], lane, inst);
            for (uint32_t i = 0; i < pa; ++i)
                for (uint32_t j = 0; j < pb; ++j)
                    prod[i + j] += ld(c, prog[pc + 16 + i], lane, inst) * ld(c, prog[pc + 16 + pa + j], lane, inst);
            pc += 16 + pa + pb;
            for (uint32_t k = 0; k + 1 < np; ++k) { prod[k + 1] += prod[k] >> 16; prod[k] &= 0xffff; }
            uint64_t quo[40] = {0};
            for (int bit = (int)np * 16 - 1; bit >= 0; --bit) {
                for (int i = 17; i > 0; --i) rem[i] = ((rem[i] << 1) | (rem[i - 1] >> 15)) & 0xffff;
                rem[0] = ((rem[0] << 1) | ((prod[bit / 16] >> (bit % 16)) & 1)) & 0xffff;
                int ge = rem[17] != 0 || rem[16] != 0;
                if (!ge) {
                    ge = 1;
                    for (int i = 15; i >= 0; --i)
                        if (rem[i] != mod[i]) { ge = rem[i] > mod[i]; break; }
                }
                if (ge) {
                    int64_t br = 0;
                    for (int i = 0; i < 18; ++i) {
                        int64_t t = (int64_t)rem[i] - (i < 16 ? (int64_t)mod[i] : 0) - br;
                        br = t < 0;
                        rem[i] = (uint64_t)(t + (br << 16));
                    }
                    quo[bit / 16] |= 1ull << (bit % 16);
                }
            }
            for (uint32_t i = 0; i < nq; ++i) st(c, prog, &pc, lane, quo[i]);
            for (uint32_t i = 0; i < 16; ++i) st(c, prog, &pc, lane, rem[i]);
        } break;
        case ZK_OP_DIVREM: {
            uint64_t x = ld(c, prog[pc++], lane, inst);
            st(c, prog, &pc, lane, x / pb);
            st(c, prog, &pc, lane, x % pb);
        } break;
        case ZK_OP_U256_MULWIDE: { /* row-wise schoolbook product over u32 limbs (the device sums column-wise) */
            uint64_t a[8], b[8], out[16] = {0};
            for (int i = 0; i < 8; ++i) a[i] = ld(c, prog[pc + i], lane, inst);
            for (int i = 0; i < 8; ++i) b[i] = ld(c, prog[pc + 8 + i], lane, inst);
            pc += 16;
            for (int i = 0; i < 8; ++i) {
                uint64_t carry = 0;
                for (int j = 0; j < 8; ++j) {
                    uint64_t t = a[i] * b[j] + out[i + j] + carry;
                    out[i + j] = t & 0xffffffffull;
                    carry = t >> 32;
                }
                out[i + 8] = carry;
            }
            for (int i = 0; i < 16; ++i) st(c, prog, &pc, lane, out[i]);
        } break;
        case ZK_OP_U256_DIVREM: { /* Knuth algorithm D over u32 limbs with 128-bit temporaries (the device divides bit-serially) */
            uint32_t a[8], b[8], q[8] = {0}, r[8] = {0};
            for (int i = 0; i < 8; ++i) a[i] = (uint32_t)ld(c, prog[pc + i], lane, inst);
            for (int i = 0; i < 8; ++i) b[i] = (uint32_t)ld(c, prog[pc + 8 + i], lane, inst);
            pc += 16;
            int n = 8;
            while (n > 0 && b[n - 1] == 0) --n;
            if (n == 0) { for (int i = 0; i < 8; ++i) r[i] = a[i]; }
            else if (n == 1) {
                uint64_t rem = 0;
                for (int i = 7; i >= 0; --i) { uint64_t cur = (rem << 32) | a[i]; q[i] = (uint32_t)(cur / b[0]); rem = cur % b[0]; }
                r[0] = (uint32_t)rem;
            } else {
                int s = __builtin_clz(b[n - 1]);
                uint32_t v[8], u[9];
                for (int i = n - 1; i > 0; --i) v[i] = (b[i] << s) | (s ? (b[i - 1] >> (32 - s)) : 0);
                v[0] = b[0] << s;
                u[8] = s ? (a[7] >> (32 - s)) : 0;
                for (int i = 7; i > 0; --i) u[i] = (a[i] << s) | (s ? (a[i - 1] >> (32 - s)) : 0);
                u[0] = a[0] << s;
                for (int j = 8 - n; j >= 0; --j) {
                    uint64_t num = ((uint64_t)u[j + n] << 32) | u[j + n - 1];
                    uint64_t qhat = num / v[n - 1], rhat = num % v[n - 1];
                    while (qhat >= (1ull << 32) || qhat * v[n - 2] > ((rhat << 32) | u[j + n - 2])) {
                        --qhat; rhat += v[n - 1];
                        if (rhat >= (1ull << 32)) break;
                    }
                    int64_t borrow = 0; uint64_t carry = 0;
                    for (int i = 0; i < n; ++i) {
                        uint64_t p = qhat * v[i] + carry;
                        carry = p >> 32;
                        int64_t t = (int64_t)u[i + j] - (int64_t)(p & 0xffffffffull) - borrow;
                        borrow = t < 0;
                        u[i + j] = (uint32_t)t;
                    }
                    int64_t t = (int64_t)u[j + n] - (int64_t)carry - borrow;
                    u[j + n] = (uint32_t)t;
                    if (t < 0) {
                        --qhat;
                        uint64_t cc = 0;
                        for (int i = 0; i < n; ++i) { uint64_t w = (uint64_t)u[i + j] + v[i] + cc; u[i + j] = (uint32_t)w; cc = w >> 32; }
                        u[j + n] += (uint32_t)cc;
                    }
                    q[j] = (uint32_t)qhat;
                }
                for (int i = 0; i < n; ++i) r[i] = (u[i] >> s) | ((s && i + 1 < n) ? (u[i + 1] << (32 - s)) : 0);
            }
            for (int i = 0; i < 8; ++i) st(c, prog, &pc, lane, q[i]);
            for (int i = 0; i < 8; ++i) st(c, prog, &pc, lane, r[i]);
        } break;
        default: return -1;
        }
    }
    return 0;
}

/* Execute words [wb, we) of the scope program for every lane. */
int zko_scope_run(const zko_scope *s, uint32_t wb, uint32_t we, uint64_t *cells, size_t stride, uint32_t n_lanes,
                  const uint64_t *inputs, const uint64_t *outer_cells, size_t outer_stride, const uint64_t *loop_cells,
                  size_t loop_stride, uint32_t loop_limit, uint32_t *mult, uint32_t total_rows) {
    run_ctx c = {s, cells, stride, n_lanes, inputs, outer_cells, outer_stride, loop_cells, loop_stride, loop_limit, mult, total_rows};
    int bad = 0;
#pragma omp parallel for schedule(static)
    for (long lane = 0; lane < (long)n_lanes; ++lane)
        if (run_lane(&c, (uint32_t)lane, wb, we)) bad = 1;
    return bad ? -1 : 0;
}

/* Sequential seeding (counterpart of k_witness_seq): iteration by iteration, carried input words
 * are taken from the previous iteration's outputs (k == 0: from the outer scope).  inputs is
 * modified in place. */
int zko_scope_run_seq(const zko_scope *s, uint64_t *cells, size_t stride, uint32_t n_instances, uint64_t *inputs,
                      const uint64_t *outer_cells, size_t outer_stride) {
    uint32_t n_lanes = n_instances * s->limit;
    run_ctx c = {s, cells, stride, n_lanes, inputs, outer_cells, outer_stride, NULL, 0, 0, NULL, 0};
    /* instances are independent chains: one thread per instance walks its iterations in order (round 3 forked a parallel region per
     * iteration; the work is the same, the 2 384 fork / joins of a main_vm pass are gone).  Threads used = min(instances, cores). */
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad)
    for (long inst = 0; inst < (long)n_instances; ++inst) {
        for (uint32_t k = 0; k < s->limit && !bad; ++k) {
            uint32_t lane = (uint32_t)inst * s->limit + k;
            for (uint32_t i = 0; i < s->n_carries; ++i) {
                const uint32_t *cr = s->carries + 4 * i;
                if (k == 0) { if (cr[3]) inputs[(size_t)cr[0] * n_lanes + lane] = outer_cells[(size_t)cr[2] * outer_stride + inst]; }
                else inputs[(size_t)cr[0] * n_lanes + lane] = cells[(size_t)cr[1] * stride + lane - 1];
            }
            if (run_lane(&c, lane, 0, s->n_prog)) bad = 1;
        }
    }
    return bad ? -1 : 0;
}

/* Lookup multiplicities of a resolved scope, counted from the lookup TUPLES of the trace (every tuple whose keys are a table row adds one
 * to that row of its instance) — the definition the prover's lookup argument uses, independent of which witness op produced the tuple
 * (a ZK_OP_LOOKUP or a macro-op such as ZK_OP_KECCAK_F).  lanes_per_instance = limit for the loop scope, 1 for the outer scope. */
void zko_scope_multiplicities(const zko_scope *s, const uint64_t *cells, size_t stride, uint32_t n_lanes, uint32_t lanes_per_instance,
                              uint32_t *mult, uint32_t total_rows) {
    const size_t NC = s->n_slots ? s->n_trace_cells / s->n_slots : 0;
    for (uint32_t lane = 0; lane < n_lanes; ++lane) {
        const uint32_t inst = lane / (lanes_per_instance ? lanes_per_instance : 1);
        for (uint32_t slot = 0; slot < s->n_slots; ++slot) {
            const zk_lookup_row_desc *lr = &s->lrows[slot];
            if (!lr->n_tuples) continue;
            const zk_table_desc *t = &s->tables[lr->table];
            for (uint32_t u = 0; u < lr->n_tuples; ++u) {
                uint32_t c0 = s->n_copy_cols + u * s->lookup_width;
                uint64_t key[3] = {0, 0, 0};
                for (uint32_t i = 0; i < t->n_keys; ++i) key[i] = cells[((size_t)slot * NC + c0 + i) * stride + lane];
                uint32_t row = table_find(s, t, key);
                if (row < t->n_rows) mult[(size_t)inst * total_rows + t->mult_off + row] += 1;
            }
        }
    }
}

static const unsigned char GW[ZK_GATE__COUNT] = {0, 1, 1, 4, 5, 4, 3, 5, 9, 24, 24, 1, 6, 5, 26};

/* Evaluate every gate relation + lookup tuple + copy pair. Returns the number of violated
 * relations; *first_key = packed (lane<<32 | slot<<12 | j<<4 | rel) of the smallest one (gates),
 * or ~0.  count_only_constraints (may be NULL) receives the number of relations evaluated. */
uint64_t zko_scope_check(const zko_scope *s, const uint64_t *cells, size_t stride, uint32_t n_lanes,
                         uint64_t *first_key, uint64_t *n_relations) {
    uint64_t bad = 0, nrel = 0, first = ~0ull;
    const size_t NC = s->n_slots ? s->n_trace_cells / s->n_slots : 0; /* cell = slot * n_columns + column */
#pragma omp parallel for schedule(static) reduction(+ : bad, nrel) reduction(min : first)
    for (long lane_ = 0; lane_ < (long)n_lanes; ++lane_) {
        uint32_t lane = (uint32_t)lane_;
#define CELL(col) cells[((size_t)slot * NC + (col)) * stride + lane]
#define FAIL(j, rel) do { ++bad; uint64_t k_ = ((uint64_t)lane << 32) | ((uint64_t)slot << 12) | (((j) & 0xff) << 4) | ((rel) & 0xf); if (k_ < first) first = k_; } while (0)
        for (uint32_t slot = 0; slot < s->n_slots; ++slot) {
            const zk_row_desc *d = &s->rows[slot];
            const uint64_t *k = s->rowconsts + d->const_off;
            uint32_t w = GW[d->kind];
            for (uint32_t j = 0; j < d->n_instances; ++j) {
                uint32_t c0 = j * w;
                switch (d->kind) {
                case ZK_GATE_CONST: ++nrel; if (CELL(c0) != k[j]) FAIL(j, 0); break;
                case ZK_GATE_BOOLEAN: { ++nrel; uint64_t v = CELL(c0); if (zko_gl_mul(v, v) != v) FAIL(j, 0); } break;
                case ZK_GATE_FMA: {
                    ++nrel;
                    uint64_t r = zko_gl_add(zko_gl_mul(k[0], zko_gl_mul(CELL(c0), CELL(c0 + 1))), zko_gl_mul(k[1], CELL(c0 + 2)));
                    if (zko_gl_sub(r, CELL(c0 + 3)) != 0) FAIL(j, 0);
                } break;
                case ZK_GATE_REDUCTION4: {
                    ++nrel;
                    uint64_t r = 0;
                    for (int i = 0; i < 4; ++i) r = zko_gl_add(r, zko_gl_mul(k[i], CELL(c0 + i)));
                    if (r != CELL(c0 + 4)) FAIL(j, 0);
                } break;
                case ZK_GATE_SELECT: {
                    ++nrel;
                    uint64_t a = CELL(c0), b = CELL(c0 + 1), sel = CELL(c0 + 2), r = CELL(c0 + 3);
                    /* s*a + (1-s)*b - r */
                    uint64_t e = zko_gl_add(zko_gl_mul(sel, a), zko_gl_mul(zko_gl_sub(1, sel), b));
                    if (e != r) FAIL(j, 0);
                } break;
                case ZK_GATE_ZEROCHECK: {
                    nrel += 2;
                    uint64_t x = CELL(c0), aux = CELL(c0 + 1), flag = CELL(c0 + 2);
                    if (zko_gl_add(zko_gl_mul(x, aux), flag) != 1) FAIL(j, 0);
                    if (zko_gl_mul(x, flag) != 0) FAIL(j, 1);
                } break;
                case ZK_GATE_UINTX_ADD: {
                    ++nrel;
                    uint64_t lhs = zko_gl_add(zko_gl_add(CELL(c0), CELL(c0 + 1)), CELL(c0 + 2));
                    uint64_t rhs = zko_gl_add(CELL(c0 + 3), zko_gl_mul(k[0], CELL(c0 + 4)));
                    if (lhs != rhs) FAIL(j, 0);
                } break;
                case ZK_GATE_DOT4: {
                    ++nrel;
                    uint64_t r = 0;
                    for (int i = 0; i < 4; ++i) r = zko_gl_add(r, zko_gl_mul(CELL(c0 + 2 * i), CELL(c0 + 2 * i + 1)));
                    if (r != CELL(c0 + 8)) FAIL(j, 0);
                } break;
                case ZK_GATE_MATMUL12_EXT:
                case ZK_GATE_MATMUL12_INT: {
                    nrel += 12;
                    uint64_t v[12];
                    for (int i = 0; i < 12; ++i) v[i] = CELL(c0 + i);
                    if (d->kind == ZK_GATE_MATMUL12_EXT) zko_poseidon2_mds_external(v); else zko_poseidon2_mds_inner(v);
                    for (int i = 0; i < 12; ++i) if (v[i] != CELL(c0 + 12 + i)) FAIL(j, i);
                } break;
                case ZK_GATE_U32_FMA: {
                    ++nrel;
                    uint64_t lhs = zko_gl_add(zko_gl_add(zko_gl_mul(CELL(c0), CELL(c0 + 1)), CELL(c0 + 2)), CELL(c0 + 3));
                    uint64_t rhs = zko_gl_add(CELL(c0 + 4), zko_gl_mul(CELL(c0 + 5), 1ull << 32));
                    if (lhs != rhs) FAIL(j, 0);
                } break;
                case ZK_GATE_U8X4_FMA: { /* both relations term by term over the 16 byte products */
                    nrel += 2;
                    uint64_t r0 = 0, r1 = 0;
                    for (int i = 0; i < 4; ++i)
                        for (int jj = 0; jj < 4; ++jj) {
                            uint64_t pr = zko_gl_mul(CELL(c0 + i), CELL(c0 + 4 + jj));
                            if (i + jj < 4) r0 = zko_gl_add(r0, zko_gl_mul(pr, 1ull << (8 * (i + jj))));
                            else r1 = zko_gl_add(r1, zko_gl_mul(pr, 1ull << (8 * (i + jj - 4))));
                        }
                    uint64_t kk = zko_gl_add(CELL(c0 + 24), zko_gl_mul(CELL(c0 + 25), 256));
                    for (int i = 0; i < 4; ++i) {
                        uint64_t sh = 1ull << (8 * i);
                        r0 = zko_gl_add(r0, zko_gl_mul(zko_gl_add(CELL(c0 + 8 + i), CELL(c0 + 12 + i)), sh));
                        r0 = zko_gl_sub(r0, zko_gl_mul(CELL(c0 + 16 + i), sh));
                        r1 = zko_gl_sub(r1, zko_gl_mul(CELL(c0 + 20 + i), sh));
                    }
                    r0 = zko_gl_sub(r0, zko_gl_mul(kk, 1ull << 32));
                    r1 = zko_gl_add(r1, kk);
                    if (r0 != 0) FAIL(j, 0);
                    if (r1 != 0) FAIL(j, 1);
                } break;
                case ZK_GATE_REDUCTION_BY_POWERS4: { /* t0 + c t1 + c^2 t2 + c^3 t3 == r, evaluated term by term */
                    ++nrel;
                    uint64_t c1 = k[0], c2 = zko_gl_mul(c1, c1), c3 = zko_gl_mul(c2, c1);
                    uint64_t r = zko_gl_add(zko_gl_add(CELL(c0), zko_gl_mul(c1, CELL(c0 + 1))),
                                            zko_gl_add(zko_gl_mul(c2, CELL(c0 + 2)), zko_gl_mul(c3, CELL(c0 + 3))));
                    if (r != CELL(c0 + 4)) FAIL(j, 0);
                } break;
                default: break;
                }
            }
            const zk_lookup_row_desc *lr = &s->lrows[slot];
            if (lr->n_tuples) {
                const zk_table_desc *t = &s->tables[lr->table];
                uint32_t tw = t->n_keys + t->n_vals;
                for (uint32_t u = 0; u < lr->n_tuples; ++u) {
                    ++nrel;
                    uint32_t c0 = s->n_copy_cols + u * s->lookup_width;
                    uint64_t key[3] = {0, 0, 0};
                    for (uint32_t i = 0; i < t->n_keys; ++i) key[i] = CELL(c0 + i);
                    uint32_t row = table_find(s, t, key);
                    int ok = row < t->n_rows;
                    for (uint32_t i = 0; ok && i < t->n_vals; ++i)
                        ok = s->table_words[t->word_off + (size_t)row * tw + t->n_keys + i] == CELL(c0 + t->n_keys + i);
                    if (!ok) FAIL(0x80 | u, 15);
                }
            }
        }
        for (uint32_t i = 0; i < s->n_copies; ++i)
            if (cells[(size_t)s->copies[i].cell * stride + lane] != cells[(size_t)s->copies[i].home * stride + lane]) {
                ++bad;
                uint64_t k_ = ((uint64_t)lane << 32) | 0xfffff000ull | (i & 0xfff);
                if (k_ < first) first = k_;
            }
#undef CELL
#undef FAIL
    }
    if (first_key) *first_key = first;
    if (n_relations) *n_relations = nrel;
    return bad;
}

/* cross-iteration / cross-scope copy constraints; returns number of violated links */
uint64_t zko_links_check(const zko_scope *loop, const uint64_t *loop_cells, size_t loop_stride, uint32_t n_lanes,
                         const uint64_t *outer_cells, size_t outer_stride) {
    uint64_t bad = 0;
    for (uint32_t off = 0; off + 3 <= loop->n_stream_words;) { /* stream links */
        const uint32_t pa = loop->streams[off], pb = loop->streams[off + 1], nt = loop->streams[off + 2];
        const uint32_t *ac = loop->streams + off + 3, *bc = ac + pa;
        for (uint32_t inst = 0; inst < n_lanes / loop->limit; ++inst)
            for (uint32_t k = 0; k < nt; ++k) {
                uint64_t va = loop_cells[(size_t)ac[k % pa] * loop_stride + (size_t)inst * loop->limit + k / pa];
                uint64_t vb = loop_cells[(size_t)bc[k % pb] * loop_stride + (size_t)inst * loop->limit + k / pb];
                if (va != vb) ++bad;
            }
        off += 3 + pa + pb;
    }
    uint32_t limit = loop->limit;
    for (uint32_t lane = 0; lane < n_lanes; ++lane) {
        uint32_t inst = lane / limit, k = lane % limit;
        for (uint32_t i = 0; i < loop->n_links; ++i) {
            const zk_link *L = &loop->links[i];
            uint64_t mine = loop_cells[(size_t)L->loop_cell * loop_stride + lane];
            if (L->kind == ZK_LINK_CARRY) {
                if (k > 0 && mine != loop_cells[(size_t)L->other_cell * loop_stride + lane - 1]) ++bad;
            } else {
                uint64_t o = outer_cells[(size_t)L->other_cell * outer_stride + inst];
                if (L->kind == ZK_LINK_FIRST) { if (k == 0 && mine != o) ++bad; }
                else if (L->kind == ZK_LINK_LAST) { if (k == limit - 1 && mine != o) ++bad; }
                else if (mine != o) ++bad;
            }
        }
    }
    return bad;
}
