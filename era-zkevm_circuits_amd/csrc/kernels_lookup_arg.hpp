// kernels_lookup_arg.hpp — K10: log-derivative lookup-argument accumulators over the resolved trace.
//
// SURVEY.md §8f rank 3 / BASELINE north star ("grand-product/lookup-argument accumulators"): what a prover computes
// from the lookup columns after witness generation.  boojum's exact polynomial form is not in the tree ([EXT]); the
// algebraic content is the standard one and is defined here for this engine's layout:
//
//   challenges beta, gamma in GF(p^2) = GF(p)[X] / (X^2 - 7)
//   tuple (c0 .. c_{W-1}) of a lookup row of table t (W = lookup width, 3 or 4; missing columns are 0):
//                                                       f = beta + sum_j gamma^j c_j + gamma^W t
//   witness side, per lane:                             A_lane = sum over every tuple of the lane of 1 / f
//   table side, per instance:                           B_inst = sum over table rows r of m[inst][r] / f(row r)
//   argument:                                           sum over the lanes of the instance of A_lane == B_inst
//
// m = the multiplicities the witness interpreter counted while resolving lookups.  One field inversion serves all
// tuples of a row (Montgomery's trick), so the witness pass is bound by reading the lookup columns (24 of 164 columns
// for the VM geometry) and the table pass by reading the multiplicities.
#pragma once
#include "gl_device.hpp"
#include "../../include/zkgl_ir.h"
#include "store_geom.hpp"

namespace zkl {

constexpr int TPB = 256;
constexpr int MAX_REPS = 32;  // lookup repetitions per row supported by the batch inversion

struct E { uint64_t a, b; };  // a + b X,  X^2 = 7
__device__ __forceinline__ E eadd(E x, E y) { return {gl::add(x.a, y.a), gl::add(x.b, y.b)}; }
__device__ __forceinline__ E emul(E x, E y) {
    uint64_t bb = gl::mul(x.b, y.b);
    uint64_t seven_bb = gl::add(gl::mul_pow2(bb, 3), gl::neg(bb));  // 7 bb = 8 bb - bb
    return {gl::add(gl::mul(x.a, y.a), seven_bb), gl::add(gl::mul(x.a, y.b), gl::mul(x.b, y.a))};
}
__device__ __forceinline__ E escale(E x, uint64_t k) { return {gl::mul(x.a, k), gl::mul(x.b, k)}; }
__device__ __forceinline__ E einv(E x) {  // (a - bX) / (a^2 - 7 b^2)
    uint64_t bb = gl::mul(x.b, x.b);
    uint64_t norm = gl::sub(gl::mul(x.a, x.a), gl::add(gl::mul_pow2(bb, 3), gl::neg(bb)));
    uint64_t ni = gl::inv(norm);
    return {gl::mul(x.a, ni), gl::mul(gl::neg(x.b), ni)};
}

struct LookupArgDev {
    const uint64_t* cells;
    uint64_t n_cells;
    uint32_t n_cols, n_lanes, n_slots, n_copy_cols, lookup_width;
    const zk_lookup_row_desc* lrows;
    E beta, g1, g2, g3, g4;  // gamma .. gamma^4
    uint64_t* acc;           // [n_lanes][2]
};

__device__ __forceinline__ size_t cell_off(uint64_t n_cells, uint32_t cell, uint32_t lane) {
    return zkgeom::offset(n_cells, cell, lane);  // n_cells = the geometry word of the store
}

// f = beta + c0 + gamma c1 + gamma^2 c2 (+ gamma^3 c3) + gamma^W t
template <typename D>
__device__ __forceinline__ E tuple_value(const D& d, uint32_t width, uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, uint32_t table) {
    E f = d.beta;
    f.a = gl::add(f.a, c0);
    f = eadd(f, escale(d.g1, c1));
    f = eadd(f, escale(d.g2, c2));
    if (width > 3) f = eadd(f, escale(d.g3, c3));
    f = eadd(f, escale(width > 3 ? d.g4 : d.g3, table));
    return f;
}

__global__ __launch_bounds__(TPB) void k_lookup_arg_witness(LookupArgDev d) {
    const uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if (lane >= d.n_lanes) return;
    const uint64_t* __restrict__ cells = d.cells + cell_off(d.n_cells, 0, lane);
    const uint32_t tsh = zkgeom::tile_log2(d.n_cells);
    E sum{0, 0};
    for (uint32_t slot = 0; slot < d.n_slots; ++slot) {
        const zk_lookup_row_desc lr = d.lrows[slot];
        const uint32_t n = lr.n_tuples;
        if (lr.table == 0xffffffffu || n == 0) continue;
        E f[MAX_REPS], pre[MAX_REPS];
        E run{1, 0};
        for (uint32_t u = 0; u < n; ++u) {  // prefix products
            const size_t c0 = (size_t)slot * d.n_cols + d.n_copy_cols + (size_t)u * d.lookup_width;
            uint64_t v0 = cells[(size_t)(c0 + 0) << tsh];
            uint64_t v1 = d.lookup_width > 1 ? cells[(size_t)(c0 + 1) << tsh] : 0;
            uint64_t v2 = d.lookup_width > 2 ? cells[(size_t)(c0 + 2) << tsh] : 0;
            uint64_t v3 = d.lookup_width > 3 ? cells[(size_t)(c0 + 3) << tsh] : 0;
            f[u] = tuple_value(d, d.lookup_width, v0, v1, v2, v3, lr.table);
            pre[u] = run;
            run = emul(run, f[u]);
        }
        E inv = einv(run);
        for (uint32_t u = n; u-- > 0;) {  // 1/f_u = inv(prod_{<=u}) * prod_{<u}
            sum = eadd(sum, emul(inv, pre[u]));
            inv = emul(inv, f[u]);
        }
    }
    d.acc[2 * (size_t)lane] = sum.a;
    d.acc[2 * (size_t)lane + 1] = sum.b;
}

struct TableArgDev {
    const zk_table_desc* tables;  // index 0 unused
    uint32_t n_tables;            // including index 0
    const uint64_t* table_words;
    uint32_t total_rows, lookup_width;
    E beta, g1, g2, g3, g4;
    uint64_t* inv_f;              // [total_rows][2]
};

// 1 / f(row) for every row of every table (instance independent)
__global__ __launch_bounds__(TPB) void k_lookup_arg_table_rows(TableArgDev d) {
    const uint32_t g = blockIdx.x * TPB + threadIdx.x;
    if (g >= d.total_rows) return;
    uint32_t t = 1;
    while (t + 1 < d.n_tables && g >= d.tables[t + 1].mult_off) ++t;  // tables are laid out in id order
    const zk_table_desc td = d.tables[t];
    const uint32_t r = g - td.mult_off, w = td.n_keys + td.n_vals;
    const uint64_t* row = d.table_words + (size_t)td.word_off + (size_t)r * w;
    E f = tuple_value(d, d.lookup_width, row[0], w > 1 ? row[1] : 0, w > 2 ? row[2] : 0, w > 3 ? row[3] : 0, t);
    E i = einv(f);
    d.inv_f[2 * (size_t)g] = i.a;
    d.inv_f[2 * (size_t)g + 1] = i.b;
}

// B_inst = sum_r m[inst][r] * inv_f[r]; one block per instance, tree reduction in LDS
__global__ __launch_bounds__(TPB) void k_lookup_arg_table_sum(const uint32_t* __restrict__ mult, const uint64_t* __restrict__ inv_f,
                                                            uint32_t total_rows, uint64_t* out /* [n_instances][2] */) {
    const uint32_t inst = blockIdx.x;
    const uint32_t* m = mult + (size_t)inst * total_rows;
    E s{0, 0};
    for (uint32_t r = threadIdx.x; r < total_rows; r += TPB) {
        const uint32_t k = m[r];
        if (k) s = eadd(s, E{gl::mul(inv_f[2 * (size_t)r], k), gl::mul(inv_f[2 * (size_t)r + 1], k)});
    }
    __shared__ uint64_t sa[TPB], sb[TPB];
    sa[threadIdx.x] = s.a; sb[threadIdx.x] = s.b;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sa[threadIdx.x] = gl::add(sa[threadIdx.x], sa[threadIdx.x + st]);
            sb[threadIdx.x] = gl::add(sb[threadIdx.x], sb[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * (size_t)inst] = sa[0]; out[2 * (size_t)inst + 1] = sb[0]; }
}

// A_inst = outer lane accumulator + sum over the instance's loop lanes; one block per instance
__global__ __launch_bounds__(TPB) void k_lookup_arg_witness_sum(const uint64_t* __restrict__ acc_outer, const uint64_t* __restrict__ acc_loop,
                                                              uint32_t limit, uint64_t* out /* [n_instances][2] */) {
    const uint32_t inst = blockIdx.x;
    E s{0, 0};
    for (uint32_t k = threadIdx.x; k < limit; k += TPB) {
        const size_t lane = (size_t)inst * limit + k;
        s = eadd(s, E{acc_loop[2 * lane], acc_loop[2 * lane + 1]});
    }
    __shared__ uint64_t sa[TPB], sb[TPB];
    sa[threadIdx.x] = s.a; sb[threadIdx.x] = s.b;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sa[threadIdx.x] = gl::add(sa[threadIdx.x], sa[threadIdx.x + st]);
            sb[threadIdx.x] = gl::add(sb[threadIdx.x], sb[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[2 * (size_t)inst] = gl::add(sa[0], acc_outer[2 * (size_t)inst]);
        out[2 * (size_t)inst + 1] = gl::add(sb[0], acc_outer[2 * (size_t)inst + 1]);
    }
}

}  // namespace zkl
