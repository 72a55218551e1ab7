// kernels_seed_wave.hpp — op-parallel cone seeding (k_seed_wave).
//
// Seeding is the sequential part of witness resolution: cycle k+1's carried state needs cycle k's.  The strand kernel
// (k_seed_cone_strands2) spends a main_vm cycle (0.76 ms) on 136 workgroup barriers, per-op scalar decode for 5 useful lanes of
// 64, and six dependent 51 us Poseidon2 permutations.  Here ONE wavefront owns an instance and its 64 lanes are the parallelism:
//   * the cone is cut into dependency levels by the host (cs.cpp build_seed_program); the ops of a level are independent, sorted
//     into segments of one op kind, and lane i of the wavefront executes op i of the segment: a level costs a few dozen
//     instructions per kind present, whatever its width;
//   * records are 16-bit (slot indices, pool indices, small parameters) and the whole program sits in LDS, shared by the up to 8
//     wavefronts (= instances) of a workgroup that walk it in lockstep; values live in a per-wavefront LDS slot store;
//   * a Poseidon2 permutation runs on 12 lanes (one state element each, 5 permutations per pass): S-boxes in parallel, the linear
//     layers through a 64-word LDS exchange buffer — a 12-wide chain instead of twelve S-boxes in a row;
//   * loop-invariant ops (pool constants, outer-scope imports) run once in a prologue and keep their slots.
// Same streams as the strand / plain cone / generic modes, word for word (tests: device-seeded state == native restatement).
#pragma once
#include "kernels_engine2.hpp"

namespace zke {

enum : uint16_t { WK_CONST = 1, WK_INPUT, WK_SELECT, WK_FMA, WK_LC4, WK_ISZERO, WK_UADD, WK_USUB, WK_DOT4, WK_SPLIT_S, WK_SPLIT_L, WK_LOOKUP, WK_P2,
                  WK_U32MULADD, WK_DIVREM, WK_U256MUL, WK_U256DIV, WK_END = 0xffff };

constexpr uint32_t SW_WAVES = 8;            // instances per workgroup
constexpr uint32_t SW_PROG_U16 = 30720;     // 60 KB of program
constexpr uint32_t SW_AREA = 1536;          // u64 words per wavefront: slots, this cycle's input words, 2 x 64 exchange words
constexpr uint32_t SW_NOSLOT = 0xffff;

// external MDS on 12 lanes: lane e = 4b + r holds x_b[r]; v[] = the 12 state words of the lane's permutation.
// out = sum_c M4[r][c] * (S_c + x_b[c]),  S_c = x_0[c] + x_1[c] + x_2[c]   (== p2::mds_external, M_E = circ(2 M4, M4, M4))
__device__ __forceinline__ uint64_t coop_mds_external(const uint64_t v[12], uint32_t e) {
    const uint32_t b = e >> 2, r = e & 3;
    // M4 rows: {5,7,1,3}, {4,6,1,1}, {1,3,5,7}, {1,1,4,6}
    const uint32_t c0 = r == 0 ? 5u : r == 1 ? 4u : 1u;
    const uint32_t c1 = r == 0 ? 7u : r == 1 ? 6u : r == 2 ? 3u : 1u;
    const uint32_t c2 = r == 0 ? 1u : r == 1 ? 1u : r == 2 ? 5u : 4u;
    const uint32_t c3 = r == 0 ? 3u : r == 1 ? 1u : r == 2 ? 7u : 6u;
    const uint32_t cf[4] = {c0, c1, c2, c3};
    uint64_t lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint64_t xb = b == 0 ? v[c] : b == 1 ? v[4 + c] : v[8 + c];
        // u = v[c] + v[4+c] + v[8+c] + xb  (< 2^66), times cf[c] (< 8)
        p2::W u{v[c], 0};
        u = p2::wadd(u, p2::W{v[4 + c], 0});
        u = p2::wadd(u, p2::W{v[8 + c], 0});
        u = p2::wadd(u, p2::W{xb, 0});
        uint64_t plo, phi;
        gl::mul_wide(u.lo, (uint64_t)cf[c], plo, phi);
        const uint32_t ph = (uint32_t)phi + u.hi * cf[c];
        const uint64_t nlo = lo + plo;
        hi += ph + (nlo < lo ? 1u : 0u);
        lo = nlo;
    }
    return gl::reduce96(lo, hi);
}
// inner MDS: out_e = sum_i v[i] + (v[e] << INNER_SHIFT[e])   (== p2::mds_inner)
__device__ __forceinline__ uint64_t coop_mds_inner(const uint64_t v[12], uint32_t e) {
    p2::W sum{v[0], 0};
#pragma unroll
    for (int i = 1; i < 12; ++i) sum = p2::wadd(sum, p2::W{v[i], 0});
    uint64_t x = v[0];
    uint32_t k = 4;  // INNER_SHIFT = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12}
#pragma unroll
    for (int i = 1; i < 12; ++i)
        if (e == (uint32_t)i) { x = v[i]; k = (uint32_t)p2::INNER_SHIFT[i]; }
    const p2::W t{x << k, k ? (uint32_t)(x >> (64 - k)) : 0u};
    const p2::W rr = p2::wadd(sum, t);
    return gl::reduce96(rr.lo, rr.hi);
}

struct SeedWaveDev {
    ScopeDev sc;
    const uint16_t* prog; uint32_t prog_u16, pro_words;
    const SeedCarryDev* carries; uint32_t n_carries;
    uint64_t* inputs_rw; uint32_t n_instances, n_slots, n_input_words;
};

__global__ __launch_bounds__(64 * SW_WAVES) void k_seed_wave(SeedWaveDev a) {
    __shared__ uint16_t prog[SW_PROG_U16];
    __shared__ uint64_t area[SW_WAVES * SW_AREA];
    const ScopeDev& sc = a.sc;
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t wv = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < a.prog_u16; i += blockDim.x) prog[i] = a.prog[i];
    uint64_t* const slots = area + wv * SW_AREA;
    uint64_t* const in_store = slots + a.n_slots;
    uint64_t* const xbuf = slots + SW_AREA - 128;
    const uint32_t inst = min(blockIdx.x * n_waves + wv, a.n_instances - 1);  // surplus wavefronts mirror the last instance (benign duplicate stores)
    const size_t lane0 = (size_t)inst * sc.limit;
    __syncthreads();

    // executes the segments from `pos` up to the end marker.  A wavefront only ever reads what IT wrote to its slot area (the
    // program is read-only), and LDS instructions of one wavefront complete in issue order: between segments a compiler-level
    // fence is enough, no workgroup barrier — the wavefronts of a workgroup drift freely and fill each other's stalls.
    auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    auto run = [&](uint32_t pos) {
        for (;;) {
            const uint32_t kind = uni(prog[pos]), count = uni(prog[pos + 1]);  // the same words for every lane: scalar control flow
            pos += 4;
            if (kind == WK_END) break;
            switch (kind) {
            case WK_CONST:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 4;
                    const uint32_t idx = (uint32_t)r[1] | ((uint32_t)r[2] << 16);
                    slots[r[0]] = r[3] ? sc.outer_cells[cell_off(sc.outer_n_cells, idx, inst)] : (uint64_t)sc.consts[idx];
                }
                pos += count * 4;
                break;
            case WK_INPUT:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 4;
                    slots[r[0]] = in_store[r[1]];
                }
                pos += count * 4;
                break;
            case WK_SELECT:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 4;
                    const uint64_t s = slots[r[0]], x = slots[r[1]], y = slots[r[2]];
                    slots[r[3]] = s ? x : y;
                }
                pos += count * 4;
                break;
            case WK_FMA:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 8;
                    const uint64_t q = sc.consts[r[0]], l = sc.consts[r[1]];
                    const uint64_t x = slots[r[2]], y = slots[r[3]], z = slots[r[4]];
                    const uint64_t xy = gl::mul(x, y);
                    slots[r[5]] = gl::add(q == 1 ? xy : gl::mul(q, xy), l == 1 ? z : gl::mul(l, z));
                }
                pos += count * 8;
                break;
            case WK_LC4:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 12;
                    uint64_t acc = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = gl::fma(sc.consts[r[j]], slots[r[4 + j]], acc);
                    slots[r[8]] = acc;
                }
                pos += count * 12;
                break;
            case WK_ISZERO:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 4;
                    const uint64_t x = slots[r[0]];
                    slots[r[1]] = x == 0 ? 1ull : 0ull;
                    if (r[3]) slots[r[2]] = x <= 1 ? x : gl::inv(x);  // the inverse is a gate witness: computed only if the cone reads it
                }
                pos += count * 4;
                break;
            case WK_UADD:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 8;
                    const uint32_t bits = r[0];
                    const uint64_t sum = slots[r[1]] + slots[r[2]] + slots[r[3]];
                    slots[r[4]] = sum & ((1ull << bits) - 1);
                    slots[r[5]] = sum >> bits;
                }
                pos += count * 8;
                break;
            case WK_USUB:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 8;
                    const uint32_t bits = r[0];
                    const uint64_t x = slots[r[1]], sub = slots[r[2]] + slots[r[3]];
                    const uint64_t borrow = x < sub ? 1 : 0;
                    slots[r[4]] = (x + (borrow << bits)) - sub;
                    slots[r[5]] = borrow;
                }
                pos += count * 8;
                break;
            case WK_DOT4:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 12;
                    uint64_t acc = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = gl::fma(slots[r[2 * j]], slots[r[2 * j + 1]], acc);
                    slots[r[8]] = acc;
                }
                pos += count * 12;
                break;
            case WK_SPLIT_S:
            case WK_SPLIT_L: {
                const uint32_t rw = kind == WK_SPLIT_S ? 12 : 68;
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * rw;
                    const uint32_t n = r[0], bits = r[1];
                    uint64_t x = slots[r[2]];
                    for (uint32_t j = 0; j < n; ++j) {
                        slots[r[3 + j]] = j + 1 == n ? x : (x & ((1ull << bits) - 1));
                        x >>= bits;
                    }
                }
                pos += count * rw;
            } break;
            case WK_LOOKUP:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 8;
                    const zk_table_desc t = sc.tables[r[0]];
                    const uint32_t nk = r[1] & 0xff, nv = r[1] >> 8;
                    const uint64_t k0 = slots[r[2]], k1 = nk > 1 ? slots[r[3]] : 0, k2 = nk > 2 ? slots[r[4]] : 0;
                    const uint32_t row = table_find3(t, sc.table_words, k0, k1, k2);
                    const bool found = row < t.n_rows;
                    const uint8_t* __restrict__ tb = reinterpret_cast<const uint8_t*>(sc.table_words + (t.dense >> 2));
                    const uint32_t w = t.n_keys + t.n_vals;
                    for (uint32_t j = 0; j < nv; ++j)
                        slots[r[5 + j]] = !found ? 0ull
                                          : (t.dense & 2u) ? (uint64_t)tb[(size_t)row * t.n_vals + j]
                                                           : sc.table_words[(size_t)t.word_off + (size_t)row * w + t.n_keys + j];
                }
                pos += count * 8;
                break;
            case WK_P2: {
                // 12 lanes per permutation, 5 permutations per pass; lanes 60..63 idle
                const uint32_t g = lane / 12, e = lane - g * 12;
                for (uint32_t base = 0; base < count; base += 5) {
                    const bool on = lane < 60 && base + g < count;
                    const uint16_t* r = prog + pos + (base + g) * 24;
                    uint64_t x = on ? slots[r[e]] : 0;
                    uint64_t v[12];
                    uint32_t flip = 0;
                    auto exchange = [&]() {  // two exchange buffers used in turn: one barrier per round
                        uint64_t* const xb = xbuf + flip * 64;
                        flip ^= 1;
                        xb[lane] = x;
                        wave_fence();
#pragma unroll
                        for (int i = 0; i < 12; ++i) v[i] = xb[min(g, 4u) * 12 + i];
                    };
                    exchange();
                    x = coop_mds_external(v, e);
#pragma unroll 1
                    for (int half = 0; half < 2; ++half) {
#pragma unroll 1
                        for (int r4 = 0; r4 < 4; ++r4) {
                            x = gl::pow7(gl::add(x, p2::RC[12 * (half * 26 + r4) + e]));
                            exchange();
                            x = coop_mds_external(v, e);
                        }
                        if (half == 0) {
#pragma unroll 1
                            for (int rr = 4; rr < 26; ++rr) {
                                const uint64_t sb = gl::pow7(gl::add(x, p2::RC[12 * rr]));  // only element 0 takes the S-box
                                x = e == 0 ? sb : x;
                                exchange();
                                x = coop_mds_inner(v, e);
                            }
                        }
                    }
                    if (on) slots[r[12 + e]] = x;
                }
                pos += count * 24;
            } break;
            case WK_U32MULADD:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 8;
                    const uint64_t v = slots[r[0]] * slots[r[1]] + slots[r[2]] + slots[r[3]];
                    slots[r[4]] = v & 0xffffffffull;
                    slots[r[5]] = v >> 32;
                }
                pos += count * 8;
                break;
            case WK_DIVREM:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 4;
                    const uint64_t x = slots[r[1]];
                    const uint32_t d = r[0];
                    slots[r[2]] = x / d;
                    slots[r[3]] = x % d;
                }
                pos += count * 4;
                break;
            case WK_U256MUL:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 32;
                    uint32_t x[8], y[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { x[j] = (uint32_t)slots[r[j]]; y[j] = (uint32_t)slots[r[8 + j]]; }
                    uint64_t lo = 0;
                    uint32_t hi = 0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int m = k - j;
                            if (m >= 0 && m < 8) {
                                const uint64_t p = (uint64_t)x[j] * y[m];
                                lo += p;
                                hi += lo < p;
                            }
                        }
                        slots[r[16 + k]] = (uint64_t)(uint32_t)lo;
                        lo = (lo >> 32) | ((uint64_t)hi << 32);
                        hi = 0;
                    }
                }
                pos += count * 32;
                break;
            case WK_U256DIV:
                for (uint32_t i = lane; i < count; i += 64) {
                    const uint16_t* r = prog + pos + i * 32;
                    uint32_t x[8], y[8], rem[8];
                    uint32_t ynz = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { x[j] = (uint32_t)slots[r[j]]; y[j] = (uint32_t)slots[r[8 + j]]; ynz |= y[j]; rem[j] = 0; }
                    if (ynz) {
#pragma unroll 1
                        for (int step = 0; step < 256; ++step) {
                            const uint32_t top = rem[7] >> 31;
#pragma unroll
                            for (int j = 7; j > 0; --j) rem[j] = (rem[j] << 1) | (rem[j - 1] >> 31);
                            rem[0] = (rem[0] << 1) | (x[7] >> 31);
#pragma unroll
                            for (int j = 7; j > 0; --j) x[j] = (x[j] << 1) | (x[j - 1] >> 31);
                            x[0] <<= 1;
                            uint32_t d[8];
                            uint32_t borrow = 0;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const uint64_t t = (uint64_t)rem[j] - y[j] - borrow;
                                d[j] = (uint32_t)t;
                                borrow = (uint32_t)(t >> 63);
                            }
                            if (top | (borrow ^ 1u)) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) rem[j] = d[j];
                                x[0] |= 1u;
                            }
                        }
                    } else {  // division by zero: q = 0, r = a (mul_div.rs:96-172)
#pragma unroll
                        for (int j = 0; j < 8; ++j) { rem[j] = x[j]; x[j] = 0; }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { slots[r[16 + j]] = (uint64_t)x[j]; slots[r[24 + j]] = (uint64_t)rem[j]; }
                }
                pos += count * 32;
                break;
            default:
                return;  // malformed program: built by the host
            }
            wave_fence();
        }
    };

    run(0);  // prologue: loop-invariant constants and outer imports
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    // the raw input words of cycle k+1 are fetched while cycle k runs (a strided gather: one cache line per word), 8 per lane
    uint64_t nxt[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t w = lane + 64 * j;
        nxt[j] = w < a.n_input_words ? a.inputs_rw[(size_t)w * sc.in_stride + lane0] : 0;
    }
    for (uint32_t k = 0; k < sc.limit; ++k) {
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t w = lane + 64 * j;
            if (w < a.n_input_words) in_store[w] = nxt[j];
        }
        if (k + 1 < sc.limit) {
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t w = lane + 64 * j;
                if (w < a.n_input_words) nxt[j] = a.inputs_rw[(size_t)w * sc.in_stride + lane0 + k + 1];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (uint32_t c = lane; c < a.n_carries; c += 64) {
            const SeedCarryDev cd = a.carries[c];
            if (k == 0 && !cd.has_first) continue;
            const uint64_t v = k == 0 ? sc.outer_cells[cell_off(sc.outer_n_cells, cd.first_outer_cell, inst)] : slots[cd.out_slot];
            in_store[cd.word] = v;
            a.inputs_rw[(size_t)cd.word * sc.in_stride + lane0 + k] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        run(a.pro_words);
    }
}

}  // namespace zke
