// kernels_engine.hpp — the generic engine kernels: witness-IR interpreter (K1/K5/K6 ops inside
// a recorded scope), per-row gate evaluation (K7), lookup-membership check and copy-constraint
// check.  Included once by zkgl_device.hip.
//
// Mapping: lane == (instance) for the outer scope, (instance*limit + iteration) for the loop
// scope.  Every lane executes the same straight-line program (no divergence).
//
// Cell storage is TILED by lane: cells[((lane >> T) * n_cells + cell) << T | (lane & (2^T - 1))] (store_geom.hpp; T = 6: one tile
// per wavefront, T = 12: 64 wavefronts share a tile and a value of the tile is 32 KB contiguous),
// cell = slot * n_columns + column.  Every cell access of a wavefront is one coalesced 512-byte transaction whose address is
// tile_base + (cell << (T + 3)) + the wavefront's place in the tile:
// consecutive columns of a gate instance and consecutively placed rows are adjacent in DRAM, so a
// wave streams through its tile (DRAM-page and TLB locality) instead of striding by the lane count.
#pragma once
#include "../../include/zkgl_ir.h"
#include "poseidon2_device.hpp"
#include "store_geom.hpp"

// Trace stores.  Round 1 stored every cell of a variable (55 k words per VM cycle) and streamed them out non-temporally.  With
// compact traces a value is stored once, to its home cell, and is usually an operand of an op a few hundred words later: a
// cacheable store keeps it in L2 for that load (the interpreter is bound by the latency of its dependent operand loads).
#define ZKGL_STORE_ASM "buffer_store_dwordx2 %[val], %[lb], %[rs], %[addr] offen\n"
#define ZKGL_STORE_AUX 0
#ifndef ZKGL_LOOP_WAVES
#define ZKGL_LOOP_WAVES 4
#endif

namespace zke {

struct ScopeDev {
    const uint32_t* prog;       // witness program words
    uint32_t n_words;
    uint32_t n_lanes;
    const uint64_t* consts;     // constant pool
    uint64_t* cells;            // [n_tiles][n_cells][64]
    uint64_t n_cells;
    const uint64_t* inputs;     // [n_input_words][in_stride >= n_lanes]: word w of lane l at inputs[w * in_stride + l]
    uint64_t in_stride;
    // loop scope only
    const uint64_t* outer_cells;
    uint64_t outer_n_cells;
    uint32_t limit;             // iterations per instance (1 for the outer scope)
    uint32_t is_loop;
    // lookup tables
    const zk_table_desc* tables;
    const uint64_t* table_words;
    uint32_t* mult;             // [n_instances][total_table_rows]
    uint32_t total_table_rows;
    // loop cells for ZK_OP_LOOP_LAST (outer scope post phase)
    const uint64_t* loop_cells;
    uint64_t loop_n_cells;
    uint32_t loop_limit;
    // fused mode of resolve_and_check: the witness kernel is the evaluator of the gates mirrored by its ops (cs.cpp build_check_program).
    // Their relations are the ops' own field arithmetic, except SELECT: s (a - b) + b - r == 0 for r = s ? a : b fails exactly when
    // s > 1 and a != b — tested on the operands in registers, reported under the macro row (the host then names the gate).
    unsigned long long* fail;
    uint32_t defer_p2;                // 1: P2_ROUNDS writes its 12 final outputs only
    unsigned long long* p2_stats;     // nullable: {skipped, run} gated witness-only permutations, one count per wavefront
    unsigned long long* clock_probe;  // nullable: {shader clock ticks, 100 MHz ticks} of the grid's first wavefront (witness_entry2)
};

constexpr int TPB = 256;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// element offset of (cell, lane) in the tiled layout; n_cells = the store's geometry word (store_geom.hpp: a bare count = 64-lane tiles)
__device__ __forceinline__ size_t cell_off(uint64_t n_cells, uint32_t cell, uint32_t lane) {
    return zkgeom::offset(n_cells, cell, lane);
}
// one value of a store through plain pointers: `word` = slot (ordinary store) or address word (geometry word with zkgeom::NARROW)
__device__ __forceinline__ uint64_t load_value(const uint64_t* __restrict__ cells, uint64_t geom, uint32_t word, uint32_t lane) {
    if (!zkgeom::narrow(geom)) return cells[zkgeom::offset(geom, word, lane)];
    const uint8_t* __restrict__ p = reinterpret_cast<const uint8_t*>(cells) + zkgeom::narrow_byte_offset(geom, word, lane);
    return (word & zkgeom::AW_BYTE) ? (uint64_t)*p : *reinterpret_cast<const uint64_t*>(p);
}
// the pieces of the buffer-addressed fast paths: a wavefront's 64 lanes sit in ONE tile (tiles are multiples of 64 lanes);
// V# base = the tile, voffset = the lane's byte in a value of the tile, soffset = slot << (T + 3)
struct TileAddr { uint64_t* base; uint32_t lane_byte, shift; };
__device__ __forceinline__ TileAddr tile_addr(uint64_t* cells, uint64_t geom, uint32_t lane) {
    const uint32_t t = zkgeom::tile_log2(geom);
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane >> t));
    TileAddr a;
    a.base = cells + (((size_t)tile * zkgeom::slots(geom)) << t);
    a.lane_byte = (lane & ((1u << t) - 1)) * 8;
    a.shift = t + 3;
    return a;
}

// A table descriptor through the SCALAR unit.  `tables` is a plain global pointer: read as `tables[tid]` the compiler issues vector
// loads, every field lands in VGPRs, and everything that depends on the descriptor — the dense / sorted branch, the byte-table
// branch, the binary-search loop, the wait behind every gather — becomes per-lane control flow under exec masks (the strand kernel's
// lookup body: one s_waitcnt vmcnt(0) per gather, the descriptor's key shifts staged through LDS).  The table id is wave-uniform
// (it comes from a program word), so the descriptor is read from the constant address space: two scalar loads, fields in SGPRs,
// scalar branches, gathers of a group back to back.
typedef __attribute__((address_space(4))) const zk_table_desc* tdesc_ptr;
__device__ __forceinline__ zk_table_desc load_table_desc(const zk_table_desc* tables, uint32_t tid) {
    const tdesc_ptr p = (tdesc_ptr)(uintptr_t)tables + uni(tid);
    zk_table_desc t;
    t.word_off = p->word_off; t.mult_off = p->mult_off; t.n_rows = p->n_rows; t.n_keys = p->n_keys; t.n_vals = p->n_vals; t.dense = p->dense;
    t.key_shift[0] = p->key_shift[0]; t.key_shift[1] = p->key_shift[1]; t.key_shift[2] = p->key_shift[2];
    return t;
}

// locate the table row for a key tuple; returns n_rows when absent
__device__ __forceinline__ uint32_t table_find(const zk_table_desc& t, const uint64_t* __restrict__ words,
                                               const uint64_t* key) {
    const uint32_t w = t.n_keys + t.n_vals;
    const uint64_t* rows = words + (size_t)t.word_off;
    if (t.dense) {
        // full product of power-of-two key ranges, last key fastest: the row index is the packed key, and a key tuple is in
        // the table iff every key is inside its range — no key words are read
        uint64_t idx = 0;
        bool ok = true;
        uint32_t hi_bit = 31 - __clz(t.n_rows);  // n_rows = 2^(sum of the key widths)
        for (uint32_t i = 0; i < t.n_keys; ++i) {
            const uint32_t bits = hi_bit - t.key_shift[i];
            ok = ok && (key[i] >> bits) == 0;
            idx += key[i] << t.key_shift[i];
            hi_bit = t.key_shift[i];
        }
        return ok ? (uint32_t)idx : t.n_rows;
    }
    uint32_t lo = 0, hi = t.n_rows;  // rows sorted lexicographically by key tuple
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        int cmp = 0;
        for (uint32_t i = 0; i < t.n_keys && cmp == 0; ++i) {
            uint64_t r = rows[(size_t)mid * w + i];
            cmp = r < key[i] ? -1 : (r > key[i] ? 1 : 0);
        }
        if (cmp == 0) return mid;
        if (cmp < 0) lo = mid + 1; else hi = mid;
    }
    return t.n_rows;
}

// One multiplicity increment per lane with `pred`.  Lanes of a wavefront are consecutive cycles of one instance and very often
// look up the SAME row (a zero limb, a cleared flag): 64 same-address atomics serialise at the memory side — on main_vm they
// were 26 of the loop kernel's 37 ms.  The wave aggregates first: one atomic per distinct row with the number of lanes on it,
// for the first three distinct rows it meets; whatever is left (hash circuits: 64 lanes, 64 different
// table rows) goes out as plain per-lane atomics, which do not collide.
__device__ __forceinline__ void mult_add(uint32_t* mult, size_t index, bool pred) {
    uint64_t todo = __builtin_amdgcn_ballot_w64(pred);
#pragma unroll 1
    for (int round = 0; round < 3 && todo; ++round) {
        const int leader = __builtin_ctzll(todo);
        const size_t li = ((size_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(index >> 32), leader) << 32) |
                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)index, leader);
        const uint64_t same = __builtin_amdgcn_ballot_w64(pred && index == li);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(mult + li, (uint32_t)__builtin_popcountll(same));
        todo &= ~same;
    }
    if (todo & (1ull << (threadIdx.x & 63))) atomicAdd(mult + index, 1u);
}

// ------------------------------------------------------------------------------------------
// Witness interpreter.  phase_begin/phase_end delimit the word range to execute (outer scope:
// pre phase before the loop, post phase after it).
// ------------------------------------------------------------------------------------------
// Executes words [word_begin, word_end) of the scope program for one lane.  Every lane of a wave is
// at the same program position (also in the sequential seeding mode, where the wave's lanes are
// different instances at the same iteration), so op words are broadcast through readfirstlane.
//
// Program fetch: the op stream is identical for every lane, so the wave keeps a 128-word window of
// it in two VGPRs (word i of the window in lane i), refilled with ONE coalesced 256-byte load per 64
// words and prefetched one window ahead; a word is read with v_readlane (a few cycles) instead of a
// dependent scalar load (an L2 round trip per word, which made the interpreter latency-bound).
// All 64 lanes of a wave must stay alive for the window loads: surplus lanes are clamped to the last
// valid lane by the callers and recompute/re-store that lane's values (benign), `active` only
// guards the multiplicity atomics.
struct ProgWindow {
    const uint32_t* p;
    uint32_t base, w0, w1, lid;
    __device__ __forceinline__ void init(const uint32_t* prog, uint32_t start) {
        p = prog;
        lid = threadIdx.x & 63;
        base = start & ~63u;
        w0 = p[base + lid];
        w1 = p[base + 64 + lid];  // host pads the program with >= 192 zero words
    }
    __device__ __forceinline__ void advance() {
        w0 = w1;
        base += 64;
        w1 = p[base + 64 + lid];
    }
    // call where reads restart from `pc` (op boundary / sequential destination lists)
    __device__ __forceinline__ void sync(uint32_t pc) {
        while (pc - base >= 64) advance();
    }
    // idx wave-uniform, base <= idx < base + 128
    __device__ __forceinline__ uint32_t at(uint32_t idx) const {
        const uint32_t off = idx - base;
        const uint32_t src = off < 64 ? w0 : w1;
        return __builtin_amdgcn_readlane(src, off & 63);
    }
};

// Big-integer witness helper of ZK_OP_NN_MULMOD: out = q[nq] | r[16] with A*B = q*M + r over base-2^16 limbs.
// Schoolbook product with carry propagation, then Knuth algorithm D (32-bit arithmetic only).  Cold path:
// kept out of line so that the interpreter loop's register budget and code size are not affected.
__device__ __noinline__ void nn_mulmod(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb, const uint32_t* m,
                                       uint32_t nq, uint32_t* out) {
    constexpr uint32_t N = 16, MAXP = 40;
    uint32_t un[MAXP + 1], v[N];
    const uint32_t np = na + nb + 2;
    uint64_t carry = 0;
    for (uint32_t k = 0; k < np; ++k) {
        uint64_t acc = carry;
        for (uint32_t i = 0; i < na; ++i) {
            const uint32_t j = k - i;
            if (j < nb) acc += (uint64_t)a[i] * b[j];
        }
        un[k] = (uint32_t)(acc & 0xffff);
        carry = acc >> 16;
    }
    const uint32_t s = __clz(m[N - 1]) - 16;  // normalisation shift (top modulus limb is non-zero)
    for (uint32_t i = N; i-- > 0;) v[i] = ((m[i] << s) | (i ? (m[i - 1] >> (16 - s)) : 0)) & 0xffff;
    un[np] = un[np - 1] >> (16 - s);
    for (uint32_t i = np; i-- > 1;) un[i] = ((un[i] << s) | (un[i - 1] >> (16 - s))) & 0xffff;
    un[0] = (un[0] << s) & 0xffff;
    for (uint32_t jj = np - N + 1; jj-- > 0;) {
        const uint32_t num = (un[jj + N] << 16) | un[jj + N - 1];
        uint32_t qhat = num / v[N - 1], rhat = num % v[N - 1];
        while (qhat >= 65536 || qhat * v[N - 2] > ((rhat << 16) | un[jj + N - 2])) {
            --qhat;
            rhat += v[N - 1];
            if (rhat >= 65536) break;
        }
        uint32_t mc = 0;
        int32_t borrow = 0;
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t p = qhat * v[i] + mc;
            mc = p >> 16;
            int32_t t = (int32_t)un[i + jj] - (int32_t)(p & 0xffff) - borrow;
            borrow = t < 0;
            un[i + jj] = (uint32_t)t & 0xffff;
        }
        int32_t t = (int32_t)un[jj + N] - (int32_t)mc - borrow;
        un[jj + N] = (uint32_t)t & 0xffff;
        if (t < 0) {  // qhat was one too large: add the divisor back
            --qhat;
            uint32_t c = 0;
            for (uint32_t i = 0; i < N; ++i) {
                const uint32_t w = un[i + jj] + v[i] + c;
                un[i + jj] = w & 0xffff;
                c = w >> 16;
            }
            un[jj + N] = (un[jj + N] + c) & 0xffff;
        }
        if (jj < nq) out[jj] = qhat;
    }
    for (uint32_t i = 0; i < N; ++i) out[nq + i] = ((un[i] >> s) | (un[i + 1] << (16 - s))) & 0xffff;
}

// WITH_BIGINT: kernels compiled with the ZK_OP_NN_MULMOD case (an out-of-line call that costs the caller ~20 VGPRs
// and 500 B of scratch); the launcher picks them only for programs that contain the op.
// SLOTS: seeding mode (k_seed_cone).  Values live in an LDS slot store instead of trace cells: cell-kind operands
// and destinations are slot indices assigned by the host's liveness allocation, one destination word per output.
// Native hash cores of the seed-only macro-ops (ZK_OP_KECCAK_ABSORB / ZK_OP_SHA256_COMPRESS): cold code, kept small
// (rolled loops, scratch-resident arrays); only the seeding kernels instantiate them.
__device__ __noinline__ void keccak_f1600(uint64_t* A) {
    const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                             0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                             0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                             0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                             0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]
    uint64_t B[25], C[5];
#pragma unroll 1
    for (int r = 0; r < 24; ++r) {
        for (int x = 0; x < 5; ++x) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
        for (int x = 0; x < 5; ++x) {
            const uint64_t c1 = C[(x + 1) % 5], d = C[(x + 4) % 5] ^ ((c1 << 1) | (c1 >> 63));
            for (int y = 0; y < 25; y += 5) A[x + y] ^= d;
        }
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) {
                const uint64_t v = A[x + 5 * y];
                const int n = ROT[x + 5 * y];
                B[y + 5 * ((2 * x + 3 * y) % 5)] = n ? ((v << n) | (v >> (64 - n))) : v;
            }
        for (int y = 0; y < 25; y += 5)
            for (int x = 0; x < 5; ++x) A[x + y] = B[x + y] ^ (~B[(x + 1) % 5 + y] & B[(x + 2) % 5 + y]);
        A[0] ^= RC[r];
    }
}
__device__ __noinline__ void sha256_compress(uint32_t* st, const uint32_t* blk) {
    const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = blk[i];
#pragma unroll 1
    for (int i = 16; i < 64; ++i)
        w[i] = w[i - 16] + (rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
        const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// TILE_UNIFORM: the 64 threads of the wave hold the 64 lanes of ONE tile (parallel kernels: consecutive lanes), so the
// tile base is wave-uniform.  Cells are then accessed with raw buffer instructions: V# = tile base, soffset (SGPR) =
// cell * 512 B straight from the program word, voffset (VGPR, constant per thread) = lane-in-tile * 8 — no vector
// address arithmetic per load / store (`buffer_store_dwordx2 v[d], v_off, s[rsrc], s_cell offen`).  The host
// uses these kernels only when cell * 512 < 2^32 (n_cells < 2^23); larger scopes run the `_wide` variants with plain
// 64-bit global addressing.
template <bool WITH_BIGINT, bool SLOTS = false, bool TILE_UNIFORM = false, int BLOCK = TPB, bool STRANDS = false>
__device__ __forceinline__ void run_lane(const ScopeDev& sc, const uint32_t lane, const uint32_t inst, const bool active,
                                         uint32_t word_begin, uint32_t word_end, const uint32_t* prog = nullptr,
                                         uint64_t* slots = nullptr, uint32_t slot_stride = 0, const uint64_t* in_area = nullptr) {
    uint64_t* __restrict__ cells = sc.cells + cell_off(sc.n_cells, 0, lane);  // this lane's column of its tile
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t tsh = zkgeom::tile_log2(sc.n_cells);   // a value of the tile is 2^tsh lanes wide
    const TileAddr ta = TILE_UNIFORM ? tile_addr(sc.cells, sc.n_cells, lane) : TileAddr{sc.cells, (lane & 63) * 8, 9};
    const uint32_t lane_byte = ta.lane_byte, bsh = ta.shift;
    __amdgpu_buffer_rsrc_t tile_rsrc = __builtin_amdgcn_make_buffer_rsrc(ta.base, 0, -1,
        0x00020000);  // gfx9 raw buffer descriptor: stride 0, num_records 2^32-1, 32-bit data format
    // the same descriptor as four SGPRs for the hand-scheduled destination loop in st()
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 tile_rsrc4;
    {
        const uint64_t bp = (uint64_t)ta.base;
        tile_rsrc4.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)bp);
        tile_rsrc4.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(bp >> 32));
        tile_rsrc4.z = -1;
        tile_rsrc4.w = 0x00020000;
    }
    // largest group of independent ops under one header (cs.cpp emit_scope: plain programs, build_strands: strand programs).
    // The members' operands are all in registers at once: strand kernels (2 waves per SIMD anyway) take 8 / 4, the plain
    // kernels 4 / 2 so that they stay at 4-5 waves per SIMD.
    constexpr uint32_t GS = STRANDS ? 8 : 4;  // SELECT, LOOKUP
    constexpr uint32_t GF = STRANDS ? 4 : 2;  // FMA, LC4
    ProgWindow P;
    P.init(SLOTS ? prog : sc.prog, word_begin);
    __shared__ uint64_t p2s[12 * BLOCK];  // Poseidon2 state, [element][thread]

    auto ld = [&](uint32_t w) -> uint64_t {
        const uint32_t kind = w & ZK_OPERAND_KIND_MASK, idx = w & ZK_OPERAND_IDX_MASK;
        if (kind == ZK_OPERAND_CONST) return sc.consts[idx];
        if (kind == ZK_OPERAND_OUTER) return sc.outer_cells[cell_off(sc.outer_n_cells, idx, inst)];
        if constexpr (SLOTS) return slots[idx * slot_stride];
        if constexpr (TILE_UNIFORM) {
            u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(tile_rsrc, lane_byte, idx << bsh, 0);
            return (uint64_t)v.x | ((uint64_t)v.y << 32);
        }
        return cells[(size_t)idx << tsh];
    };
    uint32_t pc = word_begin;
    auto st = [&](uint64_t v) {
        if constexpr (SLOTS) {
            P.sync(pc);
            slots[P.at(pc++) * slot_stride] = v;
        } else {
            if constexpr (TILE_UNIFORM) {
                // Hand-scheduled walk of the destination words that sit in the current window half (8 instructions per word:
                // readlane, store with the cell offset shifted straight into the buffer soffset, MORE test); a list that runs
                // past the window half finishes in the generic loop below.
                P.sync(pc);
                uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pc - P.base));
                if (off < 64) {
                    u32x2 o;
                    o.x = (uint32_t)v; o.y = (uint32_t)(v >> 32);
                    uint32_t wlast, addr;
                    asm volatile(
                        ".LDST%=:\n"
                        "v_readlane_b32 %[w], %[w0], %[off]\n"
                        "s_add_u32 %[off], %[off], 1\n"
                        "s_lshl_b32 %[addr], %[w], 9\n"
                        ZKGL_STORE_ASM  // see ZKGL_NT_STORES above
                        "s_bitcmp1_b32 %[w], 31\n"
                        "s_cbranch_scc0 .LDSTX%=\n"
                        "s_cmp_lt_u32 %[off], 64\n"
                        "s_cbranch_scc1 .LDST%=\n"
                        ".LDSTX%=:\n"
                        : [off] "+s"(off), [w] "=&s"(wlast), [addr] "=&s"(addr)
                        : [w0] "v"(P.w0), [val] "v"(o), [lb] "v"(lane_byte), [rs] "s"(tile_rsrc4)
                        : "scc", "memory");
                    pc = P.base + off;
                    if (!(wlast & ZK_DEST_MORE)) return;
                }
            }
            uint32_t w;
            do {
                P.sync(pc);
                w = P.at(pc++);
                if constexpr (TILE_UNIFORM) {
                    u32x2 o;
                    o.x = (uint32_t)v; o.y = (uint32_t)(v >> 32);
                    __builtin_amdgcn_raw_buffer_store_b64(o, tile_rsrc, lane_byte, (w & ZK_DEST_CELL_MASK) << bsh, ZKGL_STORE_AUX);
                } else {
                    cells[(size_t)(w & ZK_DEST_CELL_MASK) << tsh] = v;  // 64-bit addressed scopes
                }
            } while (w & ZK_DEST_MORE);
        }
    };

    while (pc < word_end) {
        P.sync(pc);
        const uint32_t h = P.at(pc++);
        const uint32_t op = h & 0xff, pa = (h >> 8) & 0xff, pb = h >> 16;
        switch (op) {
        case ZK_OP_CONST: {
            uint64_t v = ld(P.at(pc++));
            st(v);
        } break;
        case ZK_OP_INPUT: {
            if constexpr (!SLOTS) {
                const uint32_t grp = pb + 1;  // device programs: up to 8 stream words under one header
                if (grp > 1) {
                    uint64_t v[8];
#pragma unroll
                    for (uint32_t g = 0; g < 8; ++g)
                        if (g < grp) v[g] = sc.inputs[(size_t)P.at(pc + g) * sc.in_stride + lane];
                    pc += grp;
#pragma unroll
                    for (uint32_t g = 0; g < 8; ++g)
                        if (g < grp) st(v[g]);
                    break;
                }
            }
            uint32_t w = P.at(pc++);
            if constexpr (SLOTS) st(in_area[w * slot_stride]);  // staged in LDS by the kernel prologue
            else st(sc.inputs[(size_t)w * sc.in_stride + lane]);
        } break;
        case ZK_OP_FMA: {
            if constexpr (!SLOTS) {
                const uint32_t grp = pb + 1;  // device programs: up to 4 independent ops under one header (cs.cpp emit_scope / build_strands)
                if (grp > 1) {
                    uint64_t in[GF][5];
#pragma unroll
                    for (uint32_t g = 0; g < GF; ++g)
                        if (g < grp) {
#pragma unroll
                            for (uint32_t i = 0; i < 5; ++i) in[g][i] = ld(P.at(pc + g * 5 + i));
                        }
                    pc += grp * 5;
#pragma unroll
                    for (uint32_t g = 0; g < GF; ++g)
                        if (g < grp) {
                            const uint64_t ab = gl::mul(in[g][2], in[g][3]);
                            st(gl::add(in[g][0] == 1 ? ab : gl::mul(in[g][0], ab), in[g][1] == 1 ? in[g][4] : gl::mul(in[g][1], in[g][4])));
                        }
                    break;
                }
            }
            uint64_t q = ld(P.at(pc)), l = ld(P.at(pc + 1));
            uint64_t a = ld(P.at(pc + 2)), b = ld(P.at(pc + 3)), c = ld(P.at(pc + 4));
            pc += 5;
            uint64_t ab = gl::mul(a, b);
            uint64_t r = gl::add(q == 1 ? ab : gl::mul(q, ab), l == 1 ? c : gl::mul(l, c));
            st(r);
        } break;
        case ZK_OP_LC4: {
            if constexpr (!SLOTS) {
                const uint32_t grp = pb + 1;
                if (grp > 1) {
                    uint64_t in[GF][8];
#pragma unroll
                    for (uint32_t g = 0; g < GF; ++g)
                        if (g < grp) {
#pragma unroll
                            for (uint32_t i = 0; i < 8; ++i) in[g][i] = ld(P.at(pc + g * 8 + i));
                        }
                    pc += grp * 8;
#pragma unroll
                    for (uint32_t g = 0; g < GF; ++g)
                        if (g < grp) {
                            uint64_t r = 0;
#pragma unroll
                            for (int i = 0; i < 4; ++i) r = gl::fma(in[g][i], in[g][4 + i], r);
                            st(r);
                        }
                    break;
                }
            }
            uint64_t r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r = gl::fma(ld(P.at(pc + i)), ld(P.at(pc + 4 + i)), r);
            pc += 8;
            st(r);
        } break;
        case ZK_OP_SELECT: {
            if constexpr (!SLOTS) {
                const uint32_t grp = pb + 1;
                if (grp > 1) {
                    uint64_t in[GS][3];
#pragma unroll
                    for (uint32_t g = 0; g < GS; ++g)
                        if (g < grp) {
#pragma unroll
                            for (uint32_t i = 0; i < 3; ++i) in[g][i] = ld(P.at(pc + g * 3 + i));
                        }
                    pc += grp * 3;
#pragma unroll
                    for (uint32_t g = 0; g < GS; ++g)
                        if (g < grp) st(in[g][0] ? in[g][1] : in[g][2]);
                    break;
                }
            }
            uint64_t s = ld(P.at(pc)), a = ld(P.at(pc + 1)), b = ld(P.at(pc + 2));
            pc += 3;
            st(s ? a : b);
        } break;
        case ZK_OP_ISZERO: {
            uint64_t x = ld(P.at(pc++));
            st(x == 0 ? 1ull : 0ull);
            st(gl::inv(x));
        } break;
        case ZK_OP_UADD: {
            uint64_t x = ld(P.at(pc)), y = ld(P.at(pc + 1)), ci = ld(P.at(pc + 2));
            pc += 3;
            uint64_t s = x + y + ci;  // operands < 2^32
            st(s & ((1ull << pa) - 1));
            st(s >> pa);
        } break;
        case ZK_OP_USUB: {
            uint64_t x = ld(P.at(pc)), y = ld(P.at(pc + 1)), bi = ld(P.at(pc + 2));
            pc += 3;
            uint64_t sub = y + bi;
            uint64_t borrow = x < sub ? 1 : 0;
            uint64_t d = (x + (borrow << pa)) - sub;
            st(d);
            st(borrow);
        } break;
        case ZK_OP_DOT4: {
            uint64_t r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r = gl::fma(ld(P.at(pc + 2 * i)), ld(P.at(pc + 2 * i + 1)), r);
            pc += 8;
            st(r);
        } break;
        case ZK_OP_MATMUL12: {
            uint64_t s[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) s[i] = ld(P.at(pc + i));
            pc += 12;
            if (pa == 0) p2::mds_external(s); else p2::mds_inner(s);
#pragma unroll
            for (int i = 0; i < 12; ++i) st(s[i]);
        } break;
        case ZK_OP_SPLIT: {
            uint64_t x = ld(P.at(pc++));
            for (uint32_t i = 0; i < pa; ++i) {
                st(i + 1 == pa ? x : (x & ((1ull << pb) - 1)));
                x >>= pb;
            }
        } break;
        case ZK_OP_LOOKUP: {
            const uint32_t tid = P.at(pc++);
            const zk_table_desc t = load_table_desc(sc.tables, tid);
            const uint32_t nv = pb & 0xff;
            if constexpr (!SLOTS) {
                // device programs: up to 8 independent lookups into one table under one header (cs.cpp emit_scope / build_strands);
                // all key loads first, then all table gathers, then the stores and the multiplicities
                const uint32_t grp = (pb >> 8) + 1;
                if (grp > 1) {
                    uint64_t key[GS][2];
                    uint32_t row[GS];
                    uint64_t val[GS][2];
#pragma unroll
                    for (uint32_t g = 0; g < GS; ++g)
                        if (g < grp) {
                            key[g][0] = ld(P.at(pc + g * pa));
                            key[g][1] = pa > 1 ? ld(P.at(pc + g * pa + 1)) : 0;
                        }
                    pc += grp * pa;
                    const uint8_t* __restrict__ tb = reinterpret_cast<const uint8_t*>(sc.table_words + (t.dense >> 2));
                    const uint32_t w = t.n_keys + t.n_vals;
#pragma unroll
                    for (uint32_t g = 0; g < GS; ++g)
                        if (g < grp) {
                            uint64_t k3[3] = {key[g][0], key[g][1], 0};
                            row[g] = table_find(t, sc.table_words, k3);
                            const bool found = row[g] < t.n_rows;
#pragma unroll
                            for (uint32_t i = 0; i < 2; ++i)
                                if (i < nv)
                                    val[g][i] = !found ? 0ull
                                                : (t.dense & 2u) ? (uint64_t)tb[(size_t)row[g] * t.n_vals + i]
                                                                 : sc.table_words[(size_t)t.word_off + (size_t)row[g] * w + t.n_keys + i];
                        }
#pragma unroll
                    for (uint32_t g = 0; g < GS; ++g)
                        if (g < grp) {
                            P.sync(pc);
#pragma unroll
                            for (uint32_t i = 0; i < 2; ++i)
                                if (i < nv) st(val[g][i]);
                            mult_add(sc.mult, (size_t)inst * sc.total_table_rows + t.mult_off + row[g], row[g] < t.n_rows && active && sc.mult);
                        }
                    break;
                }
            }
            uint64_t key[3] = {0, 0, 0};
            for (uint32_t i = 0; i < pa; ++i) key[i] = ld(P.at(pc + i));
            pc += pa;
            uint32_t row = table_find(t, sc.table_words, key);
            const uint32_t w = t.n_keys + t.n_vals;
            bool found = row < t.n_rows;
            if (t.dense & 2u) {  // packed byte copy of a dense byte-valued table
                const uint8_t* __restrict__ tb = reinterpret_cast<const uint8_t*>(sc.table_words + (t.dense >> 2));
                for (uint32_t i = 0; i < nv; ++i) st(found ? (uint64_t)tb[(size_t)row * t.n_vals + i] : 0ull);
            } else {
                for (uint32_t i = 0; i < nv; ++i)
                    st(found ? sc.table_words[(size_t)t.word_off + (size_t)row * w + t.n_keys + i] : 0ull);
            }
            mult_add(sc.mult, (size_t)inst * sc.total_table_rows + t.mult_off + row, found && active && sc.mult);
        } break;
        case ZK_OP_POSEIDON2:      // witness-only permutation: 12 outputs
        case ZK_OP_P2_ROUNDS: {    // in-circuit permutation: every intermediate the gates constrain is
                                   // streamed to its cells (order fixed by gadgets.cpp compute_round_function)
            // The per-lane state lives in LDS ([element][thread]: conflict-free) so that the S-box loop
            // can run over a dynamic element index WITHOUT being unrolled: a fully unrolled body made
            // this kernel ~400 KB of code and instruction-cache bound.
            const bool emit = (op == ZK_OP_P2_ROUNDS);
            uint64_t s[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) s[i] = ld(P.at(pc + i));
            pc += 12;
            bool lane_off = false;   // gated form (pa = 1): [.., execute] -> zeros where the flag is off
            if (!emit && pa != 0) { lane_off = ld(P.at(pc)) == 0; pc += 1; }
            p2::mds_external(s);
#pragma unroll
            for (int i = 0; i < 12; ++i) p2s[i * BLOCK + threadIdx.x] = s[i];
            if (emit) {
#pragma unroll 1
                for (int i = 0; i < 12; ++i) st(p2s[i * BLOCK + threadIdx.x]);
            }
#pragma unroll 1
            for (int r = 0; r < 30; ++r) {
                const bool full = (r < 4) || (r >= 26);
                const int n = full ? 12 : 1;
#pragma unroll 1
                for (int i = 0; i < n; ++i) {
                    uint64_t t = gl::add(p2s[i * BLOCK + threadIdx.x], p2::RC[12 * r + i]);
                    uint64_t x2 = gl::sqr(t), x3 = gl::mul(x2, t), x4 = gl::sqr(x2), x7 = gl::mul(x3, x4);
                    if (emit) { st(t); st(x2); st(x3); st(x4); st(x7); }
                    p2s[i * BLOCK + threadIdx.x] = x7;
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) s[i] = p2s[i * BLOCK + threadIdx.x];
                if (full) p2::mds_external(s); else p2::mds_inner(s);
#pragma unroll
                for (int i = 0; i < 12; ++i) p2s[i * BLOCK + threadIdx.x] = s[i];
                if (emit) {
#pragma unroll 1
                    for (int i = 0; i < 12; ++i) st(p2s[i * BLOCK + threadIdx.x]);
                }
            }
            if (!emit) {
#pragma unroll 1
                for (int i = 0; i < 12; ++i) st(lane_off ? 0ull : p2s[i * BLOCK + threadIdx.x]);
            }
        } break;
        case ZK_OP_LOOP_LAST: {
            uint32_t c = P.at(pc++);
            st(sc.loop_cells[cell_off(sc.loop_n_cells, c, lane * sc.loop_limit + (sc.loop_limit - 1))]);
        } break;
        case ZK_OP_U32MULADD: {
            uint64_t a = ld(P.at(pc)), b = ld(P.at(pc + 1)), c = ld(P.at(pc + 2)), d = ld(P.at(pc + 3));
            pc += 4;
            uint64_t r = a * b + c + d;  // < 2^64 for u32 operands
            st(r & 0xffffffffull);
            st(r >> 32);
        } break;
        case ZK_OP_U8X4FMA: {
            uint64_t in[16], out[10];
#pragma unroll
            for (int i = 0; i < 16; ++i) in[i] = ld(P.at(pc + i));
            pc += 16;
            gl::u8x4_fma(in, out);
#pragma unroll
            for (int i = 0; i < 10; ++i) st(out[i]);
        } break;
        case ZK_OP_NN_MULMOD: if constexpr (WITH_BIGINT) {
            uint32_t mv[16], av[17], bv[17], res[19 + 16];
            for (uint32_t i = 0; i < 16; ++i) mv[i] = (uint32_t)ld(P.at(pc + i));
            for (uint32_t i = 0; i < pa; ++i) av[i] = (uint32_t)ld(P.at(pc + 16 + i));
            for (uint32_t i = 0; i < pb; ++i) bv[i] = (uint32_t)ld(P.at(pc + 16 + pa + i));
            pc += 16 + pa + pb;
            const uint32_t nq = pa + pb - 15;
            nn_mulmod(av, pa, bv, pb, mv, nq, res);
            for (uint32_t i = 0; i < nq + 16; ++i) st(res[i]);
        } else { return; } break;
        case ZK_OP_KECCAK_ABSORB: if constexpr (SLOTS) {
            uint64_t A[25];
            for (int l = 0; l < 25; ++l) {
                P.sync(pc);
                uint64_t v = 0;
                for (int k = 0; k < 8; ++k) v |= ld(P.at(pc + k)) << (8 * k);
                A[l] = v;
                pc += 8;
            }
            for (int l = 0; l < 17; ++l) {
                P.sync(pc);
                uint64_t v = 0;
                for (int k = 0; k < 8; ++k) v |= ld(P.at(pc + k)) << (8 * k);
                A[l] ^= v;
                pc += 8;
            }
            keccak_f1600(A);
            for (int l = 0; l < 25; ++l)
                for (int k = 0; k < 8; ++k) st((A[l] >> (8 * k)) & 0xff);
        } else { return; } break;
        case ZK_OP_SHA256_COMPRESS: if constexpr (SLOTS) {
            uint32_t hs[8], blk[16];
            for (int w = 0; w < 8; ++w) {
                P.sync(pc);
                uint32_t v = 0;
                for (int k = 0; k < 4; ++k) v |= (uint32_t)ld(P.at(pc + k)) << (8 * k);
                hs[w] = v;
                pc += 4;
            }
            for (int w = 0; w < 16; ++w) {
                P.sync(pc);
                uint32_t v = 0;
                for (int k = 0; k < 4; ++k) v |= (uint32_t)ld(P.at(pc + k)) << (8 * k);
                blk[w] = v;
                pc += 4;
            }
            sha256_compress(hs, blk);
            for (int w = 0; w < 8; ++w)
                for (int k = 0; k < 4; ++k) st((hs[w] >> (8 * k)) & 0xff);
        } else { return; } break;
        case ZK_OP_DIVREM: {
            uint64_t x = ld(P.at(pc++));
            st(x / pb);
            st(x % pb);
        } break;
        case ZK_OP_U256_MULWIDE: {
            // column-wise schoolbook product, a 96-bit column accumulator (up to 8 products of 64 bits + carry)
            uint32_t a[8], b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (uint32_t)ld(P.at(pc + i));
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = (uint32_t)ld(P.at(pc + 8 + i));
            pc += 16;
            uint64_t lo = 0;
            uint32_t hi = 0, out[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int j = k - i;
                    if (j >= 0 && j < 8) {
                        const uint64_t p = (uint64_t)a[i] * b[j];
                        lo += p;
                        hi += lo < p;
                    }
                }
                out[k] = (uint32_t)lo;
                lo = (lo >> 32) | ((uint64_t)hi << 32);
                hi = 0;
            }
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                uint32_t v = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) v = (i == j) ? out[j] : v;
                st((uint64_t)v);
            }
        } break;
        case ZK_OP_U256_DIVREM: {
            // restoring shift-subtract division, 256 fixed steps in registers (two per VM cycle: Div and the right shifts)
            uint32_t a[8], b[8], r[8];
            uint32_t bnz = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (uint32_t)ld(P.at(pc + i));
#pragma unroll
            for (int i = 0; i < 8; ++i) { b[i] = (uint32_t)ld(P.at(pc + 8 + i)); bnz |= b[i]; r[i] = 0; }
            pc += 16;
            if (bnz) {
#pragma unroll 1
                for (int step = 0; step < 256; ++step) {
                    // (r, a) <<= 1 : the quotient bits enter a from the bottom as the dividend bits leave at the top
                    const uint32_t top = r[7] >> 31;  // r < b <= 2^256 - 1 before the shift; a 257-bit r is handled by `top`
#pragma unroll
                    for (int i = 7; i > 0; --i) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
                    r[0] = (r[0] << 1) | (a[7] >> 31);
#pragma unroll
                    for (int i = 7; i > 0; --i) a[i] = (a[i] << 1) | (a[i - 1] >> 31);
                    a[0] <<= 1;
                    uint32_t d[8];
                    uint32_t borrow = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint64_t t = (uint64_t)r[i] - b[i] - borrow;
                        d[i] = (uint32_t)t;
                        borrow = (uint32_t)(t >> 63);
                    }
                    if (top | (borrow ^ 1u)) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) r[i] = d[i];
                        a[0] |= 1u;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { r[i] = a[i]; a[i] = 0; }
            }
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                uint32_t v = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) v = (i == j) ? a[j] : ((i == 8 + j) ? r[j] : v);
                st((uint64_t)v);
            }
        } break;
        case ZK_OP_BARRIER: if constexpr (STRANDS) {
            // end of a dependency level: this strand's stores must be visible to the other wavefronts of the tile (same CU,
            // shared L1) before any of them starts the next level
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
        } else { return; } break;
        default:
            return;  // malformed program: host validates before upload
        }
    }
}

// The plain (one lane per thread, whole program) kernels live in kernels_engine2.hpp: scalar-decoded v2 programs.

// Strand mode: a scope with too few lanes to fill the chip (hash circuits: lanes = instances x cycles; every outer scope:
// lanes = instances) runs one 64-lane tile per BLOCK of 8 wavefronts.  Wavefront w walks strand w of the program: the ops of
// every dependency level of the op graph are dealt out over the strands by the host (cs.cpp build_strands) and a
// ZK_OP_BARRIER separates the levels.  The trace is the same, cell for cell.
#ifndef ZKGL_STRANDS_PER_TILE
#define ZKGL_STRANDS_PER_TILE 16
#endif
constexpr int STRANDS_PER_TILE = ZKGL_STRANDS_PER_TILE;
constexpr int SEED_STRANDS_PER_TILE = 8;  // device_api.hpp
struct StrandTab { uint32_t begin[STRANDS_PER_TILE], end[STRANDS_PER_TILE]; };
// (the strand kernel itself is k_witness_strands2, kernels_engine2.hpp; the round-1 form over run_lane is gone)

// Sequential seeding mode (generic, slow): thread == instance, iterations in order.  Before
// iteration k the carried input words are filled from iteration k-1's outputs (k == 0: from the
// outer scope), so a host that only has the raw witness (what the reference's closures consume)
// needs no precomputed per-iteration state.  The parallel mode afterwards reproduces the same
// trace from the seeded stream.
struct CarryDev { uint32_t word, out_cell, first_outer_cell, has_first; };
template <bool WITH_BIGINT>
__global__ __launch_bounds__(64) void k_witness_seq(ScopeDev sc, const CarryDev* carries, uint32_t n_carries,
                                                    uint64_t* inputs_rw, uint32_t n_instances) {
    uint32_t inst = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x * 64 >= n_instances) return;
    const bool active = inst < n_instances;
    inst = active ? inst : n_instances - 1;
    for (uint32_t k = 0; k < sc.limit; ++k) {
        const uint32_t lane = inst * sc.limit + k;
        for (uint32_t c = 0; c < n_carries; ++c) {
            const CarryDev cd = carries[c];
            if (k == 0) {
                if (cd.has_first)
                    inputs_rw[(size_t)cd.word * sc.in_stride + lane] = sc.outer_cells[cell_off(sc.outer_n_cells, cd.first_outer_cell, inst)];
            } else {
                inputs_rw[(size_t)cd.word * sc.in_stride + lane] = sc.cells[cell_off(sc.n_cells, cd.out_cell, lane - 1)];
            }
        }
        __threadfence();
        run_lane<WITH_BIGINT>(sc, lane, inst, active, 0, sc.n_words);
        __threadfence();
    }
}


// Cone seeding (fast path of zk_cs_seed_carried_inputs): thread group == instance, iterations in order, but only the
// backward slice of the carried outputs is executed (`seed_prog`, emitted by the host: P2_ROUNDS collapsed to the
// 12-output permutation, destinations = LDS slots from a linear-scan liveness allocation).  `lpb` instances share a
// 64-thread block (threads t >= lpb mirror lane t % lpb: identical values, benign duplicate stores), so a small batch
// spreads over many CUs and the slot store (n_slots * lpb words) fits LDS.
constexpr uint32_t SEED_LDS_WORDS = 5120;  // 40 KB slot store next to the 24 KB Poseidon2 staging array
struct SeedCarryDev { uint32_t word, out_slot, first_outer_cell, has_first; };
template <bool WITH_BIGINT>
__global__ __launch_bounds__(64) void k_seed_cone(ScopeDev sc, const uint32_t* __restrict__ seed_prog, uint32_t n_words,
                                                  const SeedCarryDev* carries, uint32_t n_carries, uint64_t* inputs_rw,
                                                  uint32_t n_instances, uint32_t lpb, uint32_t n_slots, uint32_t n_input_words) {
    __shared__ uint64_t lds[SEED_LDS_WORDS];
    if (blockIdx.x * lpb >= n_instances) return;
    uint64_t* const slot_store = lds;                    // [n_slots][lpb]
    uint64_t* const in_store = lds + n_slots * lpb;      // [n_input_words][lpb]: this iteration's input stream words
    const uint32_t l = threadIdx.x % lpb;
    const uint32_t inst = min(blockIdx.x * lpb + l, n_instances - 1);  // surplus lanes mirror the last instance
    const uint32_t ml = inst - blockIdx.x * lpb;                        // its column in the stores
    for (uint32_t k = 0; k < sc.limit; ++k) {
        // cooperative prologue (all 64 threads): stage the iteration's input words in LDS with independent loads,
        // then overwrite the carried words with the previous iteration's outputs (k == 0: the outer scope's values)
        // and publish them to the stream the parallel resolve will read.
        for (uint32_t idx = threadIdx.x; idx < n_input_words * lpb; idx += 64) {
            const uint32_t w = idx / lpb, ll = idx % lpb;
            const uint32_t li = min(blockIdx.x * lpb + ll, n_instances - 1);
            in_store[w * lpb + li - blockIdx.x * lpb] = inputs_rw[(size_t)w * sc.in_stride + (size_t)li * sc.limit + k];
        }
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < n_carries * lpb; idx += 64) {
            const SeedCarryDev cd = carries[idx / lpb];
            const uint32_t ll = idx % lpb;
            const uint32_t li = min(blockIdx.x * lpb + ll, n_instances - 1), col = li - blockIdx.x * lpb;
            if (k == 0 && !cd.has_first) continue;
            const uint64_t v = k == 0 ? sc.outer_cells[cell_off(sc.outer_n_cells, cd.first_outer_cell, li)] : slot_store[cd.out_slot * lpb + col];
            in_store[cd.word * lpb + col] = v;
            inputs_rw[(size_t)cd.word * sc.in_stride + (size_t)li * sc.limit + k] = v;
        }
        __syncthreads();
        run_lane<WITH_BIGINT, true>(sc, inst * sc.limit + k, inst, false, 0, n_words, seed_prog, slot_store + ml, lpb, in_store + ml);
        __syncthreads();
    }
}

// Strand form of the cone (cs.cpp build_seed_program, second half): the block is 8 wavefronts, wavefront w walks strand w of
// the level-ordered cone with an LDS barrier between levels — one iteration's independent sponges and decompositions run side
// by side instead of as one latency chain.
template <bool WITH_BIGINT>
__global__ __launch_bounds__(64 * SEED_STRANDS_PER_TILE) void k_seed_cone_strands(ScopeDev sc, const uint32_t* __restrict__ seed_sprog, StrandTab tab,
                                                                            const SeedCarryDev* carries, uint32_t n_carries, uint64_t* inputs_rw,
                                                                            uint32_t n_instances, uint32_t lpb, uint32_t n_slots, uint32_t n_input_words) {
    constexpr uint32_t NT = 64 * SEED_STRANDS_PER_TILE;
    __shared__ uint64_t lds[SEED_LDS_WORDS];
    if (blockIdx.x * lpb >= n_instances) return;
    uint64_t* const slot_store = lds;
    uint64_t* const in_store = lds + n_slots * lpb;
    const uint32_t w = uni(threadIdx.x >> 6);
    const uint32_t l = (threadIdx.x & 63) % lpb;
    const uint32_t inst = min(blockIdx.x * lpb + l, n_instances - 1);
    const uint32_t ml = inst - blockIdx.x * lpb;
    const uint32_t wb = tab.begin[w], we = tab.end[w];
    for (uint32_t k = 0; k < sc.limit; ++k) {
        for (uint32_t idx = threadIdx.x; idx < n_input_words * lpb; idx += NT) {
            const uint32_t wd = idx / lpb, ll = idx % lpb;
            const uint32_t li = min(blockIdx.x * lpb + ll, n_instances - 1);
            in_store[wd * lpb + li - blockIdx.x * lpb] = inputs_rw[(size_t)wd * sc.in_stride + (size_t)li * sc.limit + k];
        }
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < n_carries * lpb; idx += NT) {
            const SeedCarryDev cd = carries[idx / lpb];
            const uint32_t ll = idx % lpb;
            const uint32_t li = min(blockIdx.x * lpb + ll, n_instances - 1), col = li - blockIdx.x * lpb;
            if (k == 0 && !cd.has_first) continue;
            const uint64_t v = k == 0 ? sc.outer_cells[cell_off(sc.outer_n_cells, cd.first_outer_cell, li)] : slot_store[cd.out_slot * lpb + col];
            in_store[cd.word * lpb + col] = v;
            inputs_rw[(size_t)cd.word * sc.in_stride + (size_t)li * sc.limit + k] = v;
        }
        __syncthreads();
        run_lane<WITH_BIGINT, true, false, (int)NT, true>(sc, inst * sc.limit + k, inst, false, wb, we, seed_sprog, slot_store + ml, lpb, in_store + ml);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K7: per-row gate evaluation (check_if_satisfied counterpart,
// /root/reference/src/ram_permutation/mod.rs:556).  grid = (lane tiles, slot chunks); the gate
// kind / instance count / constants of a slot are wave-uniform, each trace cell is read once.
// A failing relation is reported through atomicMin on a packed key.
// ------------------------------------------------------------------------------------------
struct CheckDev {
    const uint64_t* cells;
    uint64_t n_cells;
    uint32_t n_cols;            // copy + lookup columns
    uint32_t n_lanes;
    uint32_t n_slots;
    const zk_row_desc* rows;
    const uint64_t* rowconsts;
    const zk_lookup_row_desc* lrows;
    uint32_t n_copy_cols;
    uint32_t lookup_width;
    const zk_table_desc* tables;
    const uint64_t* table_words;
    unsigned long long* fail;   // [0] gates/lookups, [1] copies, [2] links
    uint32_t slots_per_chunk;
    // compact check: `cells` is the variable store and alias maps a trace cell to the store slot of the variable placed there;
    // nullptr: `cells` is a materialised trace
    const uint32_t* alias;
};

__device__ __constant__ const unsigned char GATE_WIDTH[ZK_GATE__COUNT] = {0, 1, 1, 4, 5, 4, 3, 5, 9, 24, 24, 1, 6, 5, 26};

__device__ __forceinline__ void report(unsigned long long* f, uint32_t lane, uint32_t slot, uint32_t j, uint32_t rel) {
    unsigned long long key = ((unsigned long long)lane << 32) | ((unsigned long long)slot << 12) | ((j & 0xff) << 4) | (rel & 0xf);
    atomicMin(f, key);
}

template <bool ALIAS>
__device__ __forceinline__ void check_gates_body(const CheckDev& cd) {
    const uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if (lane >= cd.n_lanes) return;
    const uint32_t s0 = blockIdx.y * cd.slots_per_chunk;
    const uint32_t s1 = min(s0 + cd.slots_per_chunk, cd.n_slots);
    const uint64_t* __restrict__ cells = cd.cells + cell_off(cd.n_cells, 0, lane);
    const uint32_t tsh = zkgeom::tile_log2(cd.n_cells);
    const size_t NC = cd.n_cols;
    for (uint32_t slot = s0; slot < s1; ++slot) {
        const zk_row_desc d = cd.rows[slot];
        const uint32_t kind = uni(d.kind), ninst = uni(d.n_instances);
        const uint64_t* __restrict__ k = cd.rowconsts + uni(d.const_off);
        const uint32_t w = GATE_WIDTH[kind];
        // materialised trace: every cell is read exactly once by this kernel, non-temporal loads (-2.8 % kernel time).
        // compact trace: the cell's variable lives in its home cell (wave-uniform index from the alias map); homes are read
        // once per occurrence of the variable, so these loads stay cacheable
        auto cell = [&](uint32_t col) -> uint64_t {
            if constexpr (ALIAS) return cells[(size_t)uni(cd.alias[(size_t)slot * NC + col]) << tsh];
            else return __builtin_nontemporal_load(&cells[((size_t)slot * NC + col) << tsh]);
        };
        for (uint32_t j = 0; j < ninst; ++j) {
            const uint32_t c0 = j * w;
            switch (kind) {
            case ZK_GATE_CONST: {
                // up to n_consts constants per row: instance j is bound to constant j
                if (cell(c0) != k[j]) report(cd.fail, lane, slot, j, 0);
            } break;
            case ZK_GATE_BOOLEAN: {
                uint64_t v = cell(c0);
                if (gl::mul(v, v) != v) report(cd.fail, lane, slot, j, 0);
            } break;
            case ZK_GATE_FMA: {
                uint64_t a = cell(c0), b = cell(c0 + 1), c = cell(c0 + 2), dd = cell(c0 + 3);
                uint64_t r = gl::add(gl::mul(k[0], gl::mul(a, b)), gl::mul(k[1], c));
                if (r != dd) report(cd.fail, lane, slot, j, 0);
            } break;
            case ZK_GATE_REDUCTION4: {
                uint64_t r = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) r = gl::fma(k[i], cell(c0 + i), r);
                if (r != cell(c0 + 4)) report(cd.fail, lane, slot, j, 0);
            } break;
            case ZK_GATE_SELECT: {
                uint64_t a = cell(c0), b = cell(c0 + 1), sel = cell(c0 + 2), r = cell(c0 + 3);
                uint64_t e = gl::add(gl::mul(sel, gl::sub(a, b)), b);
                if (e != r) report(cd.fail, lane, slot, j, 0);
            } break;
            case ZK_GATE_ZEROCHECK: {
                uint64_t x = cell(c0), aux = cell(c0 + 1), flag = cell(c0 + 2);
                if (gl::mul(x, aux) != gl::sub(1, flag)) report(cd.fail, lane, slot, j, 0);
                if (gl::mul(x, flag) != 0) report(cd.fail, lane, slot, j, 1);
            } break;
            case ZK_GATE_UINTX_ADD: {
                uint64_t a = cell(c0), b = cell(c0 + 1), ci = cell(c0 + 2), c = cell(c0 + 3), co = cell(c0 + 4);
                uint64_t lhs = gl::add(gl::add(a, b), ci);
                uint64_t rhs = gl::add(c, gl::mul(k[0], co));
                if (lhs != rhs) report(cd.fail, lane, slot, j, 0);
            } break;
            case ZK_GATE_DOT4: {
                uint64_t r = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) r = gl::fma(cell(c0 + 2 * i), cell(c0 + 2 * i + 1), r);
                if (r != cell(c0 + 8)) report(cd.fail, lane, slot, j, 0);
            } break;
            case ZK_GATE_MATMUL12_EXT:
            case ZK_GATE_MATMUL12_INT: {
                uint64_t s[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) s[i] = cell(c0 + i);
                if (kind == ZK_GATE_MATMUL12_EXT) p2::mds_external(s); else p2::mds_inner(s);
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    if (s[i] != cell(c0 + 12 + i)) report(cd.fail, lane, slot, j, i);
            } break;
            case ZK_GATE_U32_FMA: {
                uint64_t a = cell(c0), b = cell(c0 + 1), c = cell(c0 + 2), dd = cell(c0 + 3), lo = cell(c0 + 4), hi = cell(c0 + 5);
                uint64_t lhs = gl::add(gl::add(gl::mul(a, b), c), dd);
                uint64_t rhs = gl::add(lo, gl::mul(hi, 1ull << 32));
                if (lhs != rhs) report(cd.fail, lane, slot, j, 0);
            } break;
            case ZK_GATE_U8X4_FMA: {
                uint64_t v[26], r0, r1;
#pragma unroll
                for (int i = 0; i < 26; ++i) v[i] = cell(c0 + i);
                gl::u8x4_relations(v, r0, r1);
                if (r0 != 0) report(cd.fail, lane, slot, j, 0);
                if (r1 != 0) report(cd.fail, lane, slot, j, 1);
            } break;
            case ZK_GATE_REDUCTION_BY_POWERS4: {  // Horner in the row constant c
                uint64_t r = cell(c0 + 3);
                r = gl::fma(r, k[0], cell(c0 + 2));
                r = gl::fma(r, k[0], cell(c0 + 1));
                r = gl::fma(r, k[0], cell(c0));
                if (r != cell(c0 + 4)) report(cd.fail, lane, slot, j, 0);
            } break;
            default: break;  // NOP, PUBLIC_INPUT: no relation
            }
        }
        // lookup tuples of this row: (keys.., values..) must be a row of the row's table
        const zk_lookup_row_desc lr = cd.lrows[slot];
        const uint32_t ntup = uni(lr.n_tuples);
        if (ntup) {
            const zk_table_desc t = load_table_desc(cd.tables, lr.table);
            const uint32_t tw = t.n_keys + t.n_vals;
            for (uint32_t u = 0; u < ntup; ++u) {
                const uint32_t c0 = cd.n_copy_cols + u * cd.lookup_width;
                uint64_t key[3] = {0, 0, 0};
                for (uint32_t i = 0; i < t.n_keys; ++i) key[i] = cell(c0 + i);
                uint32_t row = table_find(t, cd.table_words, key);
                bool ok = row < t.n_rows;
                if (t.dense & 2u) {  // packed byte copy (cs.cpp finalize)
                    const uint8_t* __restrict__ tb = reinterpret_cast<const uint8_t*>(cd.table_words + (t.dense >> 2));
                    for (uint32_t i = 0; ok && i < t.n_vals; ++i) ok = (uint64_t)tb[(size_t)row * t.n_vals + i] == cell(c0 + t.n_keys + i);
                } else {
                    for (uint32_t i = 0; ok && i < t.n_vals; ++i)
                        ok = cd.table_words[(size_t)t.word_off + (size_t)row * tw + t.n_keys + i] == cell(c0 + t.n_keys + i);
                }
                if (!ok) report(cd.fail, lane, slot, 0x80 | u, 15);
            }
        }
    }
}

__global__ __launch_bounds__(TPB) void k_check_gates(CheckDev cd) { check_gates_body<false>(cd); }
__global__ __launch_bounds__(TPB) void k_check_gates_compact(CheckDev cd) { check_gates_body<true>(cd); }

// variable store -> materialised trace: every populated trace / scratch cell receives the value of its variable (the prover-stage
// kernels, zk_cs_trace_columns and the trace readers want the full trace; the witness + check pipeline never needs it).
// pairs: {trace cell, store slot}
__global__ __launch_bounds__(TPB) void k_materialize(uint64_t* __restrict__ trace_all, uint64_t n_cells, const uint64_t* __restrict__ store_all,
                                                     uint64_t n_store, uint32_t n_lanes, const zk_copy_pair* __restrict__ pairs, uint32_t n_pairs,
                                                     uint32_t pairs_per_chunk) {
    const uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if (lane >= n_lanes) return;
    uint64_t* __restrict__ trace = trace_all + cell_off(n_cells, 0, lane);
    const uint64_t* __restrict__ store = store_all + cell_off(n_store, 0, lane);
    const uint32_t p0 = blockIdx.y * pairs_per_chunk, p1 = min(p0 + pairs_per_chunk, n_pairs);
    const uint32_t tsh = zkgeom::tile_log2(n_cells), ssh = zkgeom::tile_log2(n_store);
    for (uint32_t i = p0; i < p1; ++i) {
        const zk_copy_pair p = pairs[i];
        __builtin_nontemporal_store(store[(size_t)uni(p.home) << ssh], &trace[(size_t)uni(p.cell) << tsh]);
    }
}

// copy constraints inside a scope: every non-home cell of a variable equals the home cell
__global__ __launch_bounds__(TPB) void k_check_copies(const uint64_t* __restrict__ cells_all, uint64_t n_cells, uint32_t n_lanes,
                                                      const zk_copy_pair* __restrict__ pairs, uint32_t n_pairs,
                                                      uint32_t pairs_per_chunk, unsigned long long* fail) {
    const uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if (lane >= n_lanes) return;
    const uint64_t* __restrict__ cells = cells_all + cell_off(n_cells, 0, lane);
    const uint32_t p0 = blockIdx.y * pairs_per_chunk, p1 = min(p0 + pairs_per_chunk, n_pairs);
    const uint32_t tsh = zkgeom::tile_log2(n_cells);
    for (uint32_t i = p0; i < p1; ++i) {
        const zk_copy_pair p = pairs[i];
        // the non-home cell is read once (non-temporal, -2.2 %), the home cell again by the variable's next pair
        uint64_t a = __builtin_nontemporal_load(&cells[(size_t)uni(p.cell) << tsh]), b = cells[(size_t)uni(p.home) << tsh];
        if (a != b) atomicMin(fail + 1, ((unsigned long long)lane << 32) | i);
    }
}

// copy constraints across iterations / scopes (the hidden_fsm chain of the reference:
// /root/reference/src/ram_permutation/mod.rs:119-143,178-196)
// (NARROW: the loop store is a narrow store and the link table holds address words for the loop-scope endpoints; a template, so that <false> stays the kernel a device measured)
template <bool NARROW>
__global__ __launch_bounds__(TPB) void k_check_links_t(const uint64_t* __restrict__ loop_cells, uint64_t loop_n_cells,
                                                     uint32_t n_lanes, uint32_t limit,
                                                     const uint64_t* __restrict__ outer_cells, uint64_t outer_n_cells,
                                                     const zk_link* __restrict__ links, uint32_t n_links,
                                                     unsigned long long* fail) {
    const uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if (lane >= n_lanes) return;
    const uint32_t inst = lane / limit, k = lane % limit;
    for (uint32_t i = 0; i < n_links; ++i) {
        const zk_link L = links[i];
        const uint32_t kind = uni(L.kind);
        uint64_t mine;
        if constexpr (NARROW) mine = load_value(loop_cells, loop_n_cells, uni(L.loop_cell), lane);
        else mine = loop_cells[cell_off(loop_n_cells, uni(L.loop_cell), lane)];
        const uint32_t other = uni(L.other_cell);   // (read where the wavefront is whole: the k > 0 test below splits it)
        bool ok = true;
        if (kind == ZK_LINK_CARRY) {
            if constexpr (NARROW) { if (k > 0) ok = mine == load_value(loop_cells, loop_n_cells, other, lane - 1); }
            else { if (k > 0) ok = mine == loop_cells[cell_off(loop_n_cells, other, lane - 1)]; }
        } else {
            uint64_t o = outer_cells[cell_off(outer_n_cells, other, inst)];
            if (kind == ZK_LINK_FIRST) ok = (k != 0) || mine == o;
            else if (kind == ZK_LINK_LAST) ok = (k != limit - 1) || mine == o;
            else ok = mine == o;
        }
        if (!ok) atomicMin(fail + 2, ((unsigned long long)lane << 32) | i);
    }
}

// Stream links (include/zkgl_ir.h): thread == (instance, global index k); both copies of element k must agree.
template <bool NARROW>
__global__ __launch_bounds__(TPB) void k_check_stream_t(const uint64_t* __restrict__ loop_cells, uint64_t loop_n_cells,
                                                      uint32_t n_instances, uint32_t limit, const uint32_t* __restrict__ a_cells,
                                                      uint32_t pa, const uint32_t* __restrict__ b_cells, uint32_t pb,
                                                      uint32_t n_total, uint32_t stream_index, unsigned long long* fail) {
    const uint64_t t = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (t >= (uint64_t)n_instances * n_total) return;
    const uint32_t inst = (uint32_t)(t / n_total), k = (uint32_t)(t % n_total);
    const uint32_t lane_a = inst * limit + k / pa, lane_b = inst * limit + k / pb;
    uint64_t va, vb;
    if constexpr (NARROW) {
        va = load_value(loop_cells, loop_n_cells, a_cells[k % pa], lane_a);
        vb = load_value(loop_cells, loop_n_cells, b_cells[k % pb], lane_b);
    } else {
        va = loop_cells[cell_off(loop_n_cells, a_cells[k % pa], lane_a)];
        vb = loop_cells[cell_off(loop_n_cells, b_cells[k % pb], lane_b)];
    }
    if (va != vb) atomicMin(fail + 2, ((unsigned long long)lane_a << 32) | 0x80000000u | stream_index);
}

}  // namespace zke
