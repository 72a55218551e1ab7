// kernels_perm.hpp — K12: copy-permutation grand product z over the resolved trace.
//
// SURVEY.md §8f rank 3 ("copy-permutation grand product z(X)"): with the lookup accumulators (K10) and the NTT / LDE (K11) the
// third piece of the prover stage that follows satisfiability.  boojum's column chunking and its choice of cell identifiers
// are not in the tree ([EXT]); the argument is the standard one over this engine's trace:
//
//   label of a trace cell     outer scope: cell index;  loop scope, iteration k: NT_outer + k * NT_loop + cell index
//   sigma                     a permutation of the labels whose cycles are exactly the copy classes: the cells of a variable
//                             within a scope (zk_copy_pair) joined across iterations / scopes by the links (zk_link, stream
//                             links) — built on the host (cs_perm.cpp), per scope relative + overrides at the link endpoints
//   row factor                F(row) = prod over the row's populated columns of (w + beta * label + gamma) / (w + beta * sigma(label) + gamma)
//   z[0] = 1, z[r + 1] = z[r] * F(r) over the rows of one instance (loop iterations in order, then the outer scope); the copy
//   constraints hold iff z[rows] = 1 (up to the soundness error of the random beta, gamma in GF(p^2) = GF(p)[X]/(X^2 - 7))
//
// beta * label + gamma splits into a per-lane part beta * (label base of the iteration) + gamma and a per-cell part beta * cell
// (and beta * sigma_rel(cell)) that is the same for every lane (one table per scope and call, scalar loads), so a cell costs two extension-field multiplications (numerator and
// denominator running products) plus additions; only link endpoints pay for beta * sigma directly.
#pragma once
#include "kernels_lookup_arg.hpp"

namespace zkp {

using zkl::E;
using zkl::eadd;
using zkl::emul;
using zkl::einv;
using zkl::escale;
constexpr int TPB = 256;
constexpr uint32_t NONE = 0xffffffffu;

struct PermDev {
    const uint64_t* cells; uint64_t n_cells;
    uint32_t n_cols, n_lanes, n_slots, n_copy_cols, lookup_width;
    const zk_row_desc* rows; const zk_lookup_row_desc* lrows;
    const uint32_t* sigma_rel;  // per trace cell: sigma inside the scope and iteration (cell index)
    const uint32_t* ep_index;   // per trace cell: link endpoint number or NONE
    const uint64_t* ovr;        // [endpoint][lanes_per_instance] full sigma labels of the endpoints
    uint32_t lanes_per_instance;// loop: limit, outer: 1
    uint64_t label_base, label_step;  // label(k, cell) = label_base + k * label_step + cell
    const uint64_t* tb;         // per trace cell 4 words: beta * cell (a, b), beta * sigma_rel(cell) (a, b)
    E beta, gamma;
    uint32_t slots_per_chunk, n_chunks;  // a thread walks one chunk of a lane's rows (the outer scope has one lane per instance)
    uint64_t* lane_out;         // [n_lanes][n_chunks][4]: numerator (a, b), denominator (a, b) of the chunk's rows
    uint64_t* prefix;           // optional [n_slots][n_lanes][4]: running products inside the chunk after each row
    const uint32_t* slot1;      // compact batch: `cells` is the variable store and slot1[trace cell] = store slot + 1 (the trace is a view)
};

// Karatsuba form of the GF(p^2) product: three base-field multiplications instead of four (the kernel is bound by them: two
// extension products per populated cell)
__device__ __forceinline__ E emul3(E x, E y) {
    const uint64_t aa = gl::mul(x.a, y.a), bb = gl::mul(x.b, y.b);
    const uint64_t cross = gl::sub(gl::sub(gl::mul(gl::add(x.a, x.b), gl::add(y.a, y.b)), aa), bb);   // a d + b c
    return {gl::add(aa, gl::add(gl::mul_pow2(bb, 3), gl::neg(bb))), cross};
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__global__ __launch_bounds__(TPB) void k_perm_lane(PermDev d) {
    const uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if (lane >= d.n_lanes) return;
    const uint32_t k = lane % d.lanes_per_instance;
    const uint64_t* __restrict__ cells = d.cells + zkl::cell_off(d.n_cells, 0, lane);
    const uint32_t tsh = zkgeom::tile_log2(d.n_cells);
    const E A = eadd(escale(d.beta, gl::reduce(d.label_base + (uint64_t)k * d.label_step)), d.gamma);  // labels < 2^63 < p
    E num{1, 0}, den{1, 0};
    const uint32_t s0 = blockIdx.y * d.slots_per_chunk, s1 = min(s0 + d.slots_per_chunk, d.n_slots);
    for (uint32_t slot = s0; slot < s1; ++slot) {
        const zk_row_desc rd = d.rows[slot];
        const uint32_t n_gate_cols = uni(rd.n_instances) * zke::GATE_WIDTH[uni(rd.kind)];
        const uint32_t n_lookup_cols = uni(d.lrows[slot].n_tuples) * d.lookup_width;
        for (uint32_t part = 0; part < 2; ++part) {
            const uint32_t c0 = part ? d.n_copy_cols : 0, c1 = c0 + (part ? n_lookup_cols : n_gate_cols);
            for (uint32_t col = c0; col < c1; ++col) {
                const uint32_t cell = slot * d.n_cols + col;
                // a column the row descriptor counts but no variable occupies (the unused columns of a narrow lookup tuple) holds 0
                const uint32_t s1 = d.slot1 ? uni(d.slot1[cell]) : cell + 1u;
                const uint64_t w = s1 ? cells[(size_t)(s1 - 1u) << tsh] : 0ull;
                const uint64_t* __restrict__ t = d.tb + 4 * (size_t)cell;
                E tn = eadd(A, E{t[0], t[1]});
                tn.a = gl::add(tn.a, w);
                const uint32_t ep = uni(d.ep_index[cell]);
                E td = ep == NONE ? eadd(A, E{t[2], t[3]})
                                  : eadd(escale(d.beta, d.ovr[(size_t)ep * d.lanes_per_instance + k]), d.gamma);
                td.a = gl::add(td.a, w);
                num = emul3(num, tn);
                den = emul3(den, td);
            }
        }
        if (d.prefix) {
            uint64_t* p = d.prefix + ((size_t)slot * d.n_lanes + lane) * 4;
            p[0] = num.a; p[1] = num.b; p[2] = den.a; p[3] = den.b;
        }
    }
    uint64_t* o = d.lane_out + ((size_t)lane * d.n_chunks + blockIdx.y) * 4;
    o[0] = num.a; o[1] = num.b; o[2] = den.a; o[3] = den.b;
}

// tb[cell] = beta * cell, beta * sigma_rel[cell]
__global__ __launch_bounds__(TPB) void k_perm_tb(E beta, const uint32_t* __restrict__ sigma_rel, uint64_t* tb, uint32_t n) {
    const uint32_t c = blockIdx.x * TPB + threadIdx.x;
    if (c >= n) return;
    const uint32_t s = sigma_rel[c];
    tb[4 * (size_t)c] = gl::mul(beta.a, c);
    tb[4 * (size_t)c + 1] = gl::mul(beta.b, c);
    tb[4 * (size_t)c + 2] = gl::mul(beta.a, s);
    tb[4 * (size_t)c + 3] = gl::mul(beta.b, s);
}

struct Frac { E n, d; };
__device__ __forceinline__ Frac fmul(Frac x, Frac y) { return {emul(x.n, y.n), emul(x.d, y.d)}; }
__device__ __forceinline__ Frac fload(const uint64_t* p) { return {{p[0], p[1]}, {p[2], p[3]}}; }
__device__ __forceinline__ void fstore(uint64_t* p, Frac f) { p[0] = f.n.a; p[1] = f.n.b; p[2] = f.d.a; p[3] = f.d.b; }

// One block per instance: exclusive scan over the instance's `per` partial products (lane-major, chunk-minor), seeded with
// seed[inst] (the product of everything before them; null = 1); excl may be null; total[inst] = seed * all of them.
__global__ __launch_bounds__(TPB) void k_perm_scan(const uint64_t* __restrict__ part, uint32_t per, const uint64_t* __restrict__ seed,
                                                 uint64_t* excl, uint64_t* total) {
    const uint32_t inst = blockIdx.x, t = threadIdx.x;
    const uint32_t chunk = (per + TPB - 1) / TPB;
    const uint32_t k0 = min(t * chunk, per), k1 = min(k0 + chunk, per);
    Frac loc{{1, 0}, {1, 0}};
    for (uint32_t k = k0; k < k1; ++k) loc = fmul(loc, fload(part + ((size_t)inst * per + k) * 4));
    __shared__ uint64_t sc[TPB][4];
    fstore(sc[t], loc);
    __syncthreads();
    for (int off = 1; off < TPB; off <<= 1) {  // inclusive Hillis-Steele over the per-thread totals
        Frac v = fload(sc[t]);
        if ((int)t >= off) v = fmul(fload(sc[t - off]), v);
        __syncthreads();
        fstore(sc[t], v);
        __syncthreads();
    }
    const Frac sd = seed ? fload(seed + (size_t)inst * 4) : Frac{{1, 0}, {1, 0}};
    Frac run = t ? fmul(sd, fload(sc[t - 1])) : sd;
    for (uint32_t k = k0; k < k1; ++k) {
        if (excl) fstore(excl + ((size_t)inst * per + k) * 4, run);
        run = fmul(run, fload(part + ((size_t)inst * per + k) * 4));
    }
    if (t == TPB - 1) fstore(total + (size_t)inst * 4, fmul(sd, fload(sc[TPB - 1])));
}

// z[inst][row_base + k * n_slots + slot + 1] = excl(lane, chunk) * prefix(slot, lane) as one field element (2 words)
__global__ __launch_bounds__(TPB) void k_perm_z(const uint64_t* __restrict__ excl, const uint64_t* __restrict__ prefix, uint32_t n_lanes,
                                              uint32_t n_slots, uint32_t slots_per_chunk, uint32_t n_chunks, uint32_t lanes_per_instance,
                                              uint64_t row_base, uint64_t rows_per_instance, uint64_t* z) {
    const uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if (lane >= n_lanes) return;
    const uint32_t inst = lane / lanes_per_instance, k = lane % lanes_per_instance;
    const Frac ex = fload(excl + ((size_t)lane * n_chunks + blockIdx.y) * 4);
    uint64_t* zi = z + (size_t)inst * (rows_per_instance + 1) * 2;
    if (row_base == 0 && k == 0 && blockIdx.y == 0) { zi[0] = 1; zi[1] = 0; }
    const uint32_t s0 = blockIdx.y * slots_per_chunk, s1 = min(s0 + slots_per_chunk, n_slots);
    for (uint32_t slot = s0; slot < s1; ++slot) {
        const Frac f = fmul(ex, fload(prefix + ((size_t)slot * n_lanes + lane) * 4));
        const E v = emul(f.n, einv(f.d));
        const uint64_t row = row_base + (uint64_t)k * n_slots + slot + 1;
        zi[2 * row] = v.a; zi[2 * row + 1] = v.b;
    }
}

}  // namespace zkp
