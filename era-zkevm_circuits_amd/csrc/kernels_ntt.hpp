// kernels_ntt.hpp — K11: batched Goldilocks NTT / coset LDE over device-resident polynomials.
//
// SURVEY.md §8f rank 3 ("LDE/NTT over Goldilocks"): the step a prover runs on the trace columns after satisfiability.
// boojum's transform code is not in the tree ([EXT]); the mathematical object is fixed by the field alone and is defined
// here (include/zkgl.h, zk_ntt):
//
//   omega_N = 7^((p-1)/N)        (7 generates the multiplicative group; N = 2^log_n <= 2^32)
//   forward : coefficients a[0..N) in natural order  ->  A[bitrev(k)] = sum_i a[i] (g omega_N^k)^i      (g = coset shift)
//   inverse : the inverse map (bit-reversed evaluations on g<omega_N>  ->  natural coefficients)
//
// Decomposition (Bailey's four-step, applied recursively): N = 2^r * 2^lo.  One PASS runs, for every segment of 2^seg
// consecutive elements (seg = r + lo), the 2^lo independent size-2^r transforms over the TOP r index bits entirely in LDS and
// multiplies by the inter-step twiddle omega_{2^seg}^(low index * k); the next pass treats every 2^lo chunk as a segment
// of its own.  With r <= 10 a 2^20 transform is two passes = two reads + two writes of the data (the 8 MB of one
// polynomial cannot stay on chip).  A block stages 2^(r+t) <= 8192 elements (64 KB of the 160 KB LDS): all 2^r values of
// the top bits x 2^t CONSECUTIVE low indices, so every global access is a run of 2^t * 8 >= 64 contiguous bytes; the last
// pass (lo = 0) stages 2^t whole segments, one contiguous 64 KB range.  Sub-transforms are radix-2 decimation in
// frequency (natural -> bit-reversed inside the r bits), which makes the final layout exactly bitrev_n; the inverse runs the
// passes backwards with decimation-in-time butterflies and inverse tables.
//
// Bound: 16 B of HBM traffic per element per pass against ~(r/2 + 3) field multiplications (each four v_mad_u64_u32 plus
// the 2^64 = 2^32 - 1 folding): on gfx950 the passes are VALU-bound (VALUBusy 88 %), see profiles/r1_ntt.md.
#pragma once
#include "gl_device.hpp"
#include "store_geom.hpp"

namespace zkn {

constexpr int TPB = 512;
constexpr uint32_t LOG_BLOCK_ELEMS = 13;  // 8192 elements = 64 KB (+ padding and 4 KB of twiddles) of LDS per block, 2 blocks per CU

struct PassDev {
    const uint64_t* src;
    uint64_t* dst;
    uint64_t src_stride, dst_stride;  // elements between consecutive polynomials
    uint32_t log_n, seg, r, t;        // lo = seg - r
    uint32_t dit;                     // 0: decimation in frequency (natural -> bit-reversed), 1: in time (bit-reversed -> natural)
    uint32_t coset_store;             // coset / scale factors applied when storing (inverse direction) instead of when loading
    uint32_t coset_brev;              // the coefficient side is in bit-reversed order: factor index = bitrev_n(position)
    const uint64_t* root1024;         // omega_1024^(+-j), j < 512
    const uint64_t* tw_lo;            // omega_{2^seg}^(+-j), j < 1024
    const uint64_t* tw_hi;            // omega_{2^seg}^(+-1024 j), j < max(1, 2^seg / 1024)
    const uint64_t* c_lo;             // coset / scale factor of natural index i: c_lo[i & 1023] * c_hi[i >> 10]; null = none
    const uint64_t* c_hi;
};

__device__ __forceinline__ uint32_t brev(uint32_t x, uint32_t bits) { return bits ? __brev(x) >> (32 - bits) : 0; }
// one spare word per 32: a thread that owns 8 or 16 consecutive elements then starts in a bank of its own
__device__ __forceinline__ uint32_t pad(uint32_t i) { return i + (i >> 5); }
__host__ __device__ constexpr uint32_t padded_elems(uint32_t n) { return n + (n >> 5) + 1; }

// Stages lh = g_lo .. g_lo + M - 1 of the size-2^r transforms in registers: every work item owns the 2^M elements that differ in
// those bits of h, so a group of M stages costs one LDS round trip and one barrier (r = 10 runs as 4 + 3 + 3).
template <int M, bool INV>
__device__ __forceinline__ void radix_group(uint64_t* __restrict__ s, const uint64_t* __restrict__ W, uint32_t g_lo, uint32_t r, uint32_t t,
                                            bool strided, uint32_t n_elems) {
    constexpr int R = 1 << M;
    const uint32_t n_items = n_elems >> M;
    for (uint32_t it = threadIdx.x; it < n_items; it += TPB) {
        uint32_t p, l;
        if (strided) { l = it & ((1u << t) - 1); p = it >> t; }
        else { p = it & ((1u << (r - M)) - 1); l = it >> (r - M); }
        const uint32_t low = p & ((1u << g_lo) - 1);
        const uint32_t base_h = ((p >> g_lo) << (g_lo + M)) | low;
        uint32_t addr[R];
        uint64_t v[R];
#pragma unroll
        for (int x = 0; x < R; ++x) {
            const uint32_t h = base_h | ((uint32_t)x << g_lo);
            addr[x] = pad(strided ? (h << t) | l : (l << r) | h);
            v[x] = s[addr[x]];
        }
#pragma unroll
        for (int q = 0; q < M; ++q) {
            const int lv = INV ? q : M - 1 - q;  // decimation in frequency walks the big spans first, in time the small ones
            const uint32_t lh = g_lo + lv;
#pragma unroll
            for (int x = 0; x < R; ++x) {
                if (x & (1 << lv)) continue;
                const uint32_t j = low | ((uint32_t)(x & ((1 << lv) - 1)) << g_lo);
                const uint64_t w = W[j << (9 - lh)];  // omega_{2^(lh+1)}^(+-j)
                const uint64_t a = v[x], b = v[x | (1 << lv)];
                if (!INV) {
                    v[x] = gl::add(a, b);
                    v[x | (1 << lv)] = gl::mul(gl::sub(a, b), w);
                } else {
                    const uint64_t bw = gl::mul(b, w);
                    v[x] = gl::add(a, bw);
                    v[x | (1 << lv)] = gl::sub(a, bw);
                }
            }
        }
#pragma unroll
        for (int x = 0; x < R; ++x) s[addr[x]] = v[x];
    }
}

template <bool INV>
__device__ __forceinline__ void run_groups(uint64_t* s, const uint64_t* W, uint32_t r, uint32_t t, bool strided, uint32_t n_elems) {
    const uint32_t n_groups = (r + 3) / 4, base = r / n_groups, extra = r % n_groups;  // 10 -> 4,3,3 ; 9 -> 3,3,3 ; 8 -> 4,4
    uint32_t done = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t m = base + (g < extra ? 1 : 0);
        const uint32_t g_lo = INV ? done : r - done - m;
        switch (m) {
        case 1: radix_group<1, INV>(s, W, g_lo, r, t, strided, n_elems); break;
        case 2: radix_group<2, INV>(s, W, g_lo, r, t, strided, n_elems); break;
        case 3: radix_group<3, INV>(s, W, g_lo, r, t, strided, n_elems); break;
        default: radix_group<4, INV>(s, W, g_lo, r, t, strided, n_elems); break;
        }
        done += m;
        __syncthreads();
    }
}

__global__ __launch_bounds__(TPB) void k_ntt_pass(PassDev d) {
    extern __shared__ uint64_t lds[];
    uint64_t* W = lds;           // 512 twiddles of the in-block stages
    uint64_t* s = lds + 512;     // the block's elements, padded
    const uint32_t r = d.r, t = d.t, lo = d.seg - d.r;
    const uint32_t n_elems = 1u << (r + t);
    const bool strided = lo > 0;  // lo > 0: element e = (h << t) | l ; lo == 0: e = (l << r) | h, one contiguous range
    const uint32_t tiles_per_poly = 1u << (d.log_n - r - t);
    const uint32_t poly = blockIdx.x / tiles_per_poly, tile = blockIdx.x % tiles_per_poly;
    const uint64_t* __restrict__ src = d.src + (size_t)poly * d.src_stride;
    uint64_t* __restrict__ dst = d.dst + (size_t)poly * d.dst_stride;
    uint32_t seg_base, low_base;
    if (strided) {
        const uint32_t tiles_per_seg = 1u << (lo - t);
        seg_base = (tile / tiles_per_seg) << d.seg;
        low_base = (tile % tiles_per_seg) << t;
    } else {
        seg_base = tile << (r + t);
        low_base = 0;
    }
    auto gidx = [&](uint32_t e) -> uint32_t {  // global index of block element e
        return strided ? seg_base + ((e >> t) << lo) + low_base + (e & ((1u << t) - 1)) : seg_base + e;
    };
    auto inter_twiddle = [&](uint32_t e) -> uint64_t {  // omega_{2^seg}^(+-(low index) * k), k = bitrev_r(h)
        const uint32_t h = e >> t, l = e & ((1u << t) - 1);
        const uint32_t ex = (low_base + l) * brev(h, r);  // < 2^seg <= 2^30
        return gl::mul(d.tw_lo[ex & 1023], d.tw_hi[ex >> 10]);
    };
    auto coset = [&](uint32_t pos) -> uint64_t {
        const uint32_t i = d.coset_brev ? brev(pos, d.log_n) : pos;
        return gl::mul(d.c_lo[i & 1023], d.c_hi[i >> 10]);
    };

    for (uint32_t e = threadIdx.x; e < 512; e += TPB) W[e] = d.root1024[e];
    for (uint32_t e = threadIdx.x; e < n_elems; e += TPB) {
        const uint32_t i = gidx(e);
        uint64_t v = src[i];
        if (d.c_lo && !d.coset_store) v = gl::mul(v, coset(i));  // forward direction: a[i] * g^i on the coefficient side
        if (d.dit && strided) v = gl::mul(v, inter_twiddle(e));  // the inter-step twiddle precedes decimation-in-time stages
        s[pad(e)] = v;
    }
    __syncthreads();
    if (d.dit) run_groups<true>(s, W, r, t, strided, n_elems);
    else run_groups<false>(s, W, r, t, strided, n_elems);
    for (uint32_t e = threadIdx.x; e < n_elems; e += TPB) {
        const uint32_t i = gidx(e);
        uint64_t v = s[pad(e)];
        if (!d.dit && strided) v = gl::mul(v, inter_twiddle(e));
        if (d.c_lo && d.coset_store) v = gl::mul(v, coset(i));   // inverse direction: 1/N * g^-i on the coefficient side
        dst[i] = v;
    }
}

// c_lo[j] = scale * base^j (j < 1024), c_hi[j] = base^(1024 j) (j < n_hi)
__global__ __launch_bounds__(256) void k_coset_tables(uint64_t base, uint64_t scale, uint64_t* c_lo, uint64_t* c_hi, uint32_t n_hi) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= 1024 + n_hi) return;
    uint64_t b = base;
    uint32_t ex = g;
    if (g >= 1024) {
        for (int i = 0; i < 10; ++i) b = gl::sqr(b);  // base^1024
        ex = g - 1024;
    }
    uint64_t acc = 1;
    while (ex) {
        if (ex & 1) acc = gl::mul(acc, b);
        b = gl::sqr(b);
        ex >>= 1;
    }
    if (g < 1024) c_lo[g] = gl::mul(acc, scale);
    else c_hi[g - 1024] = acc;
}

// Trace of one instance as column polynomials: out[col * stride + row], rows = loop iterations in order (row = iteration *
// loop_slots + slot) followed by the outer scope's slots, zero padded to n_rows_padded.  The wave-tiled cell storage keeps
// 64 consecutive lanes of one cell together, the column layout wants consecutive rows of one lane together: a block
// transposes 64 iterations x 64 slots of one column through LDS (512 B reads, 512 B writes).  instance = the first instance, the batch
// form writes instance i at out + (i - instance) * instance_stride.
struct ColumnsDev {
    const uint64_t* loop_cells; uint64_t loop_n_cells;
    const uint64_t* outer_cells; uint64_t outer_n_cells;
    uint32_t n_cols, loop_slots, outer_slots, limit, instance;
    uint64_t* out; uint64_t stride; uint64_t n_rows_padded;
    // compact mode (no materialised trace): *_cells = the variable stores, *_slot1[trace cell] = store slot + 1 of the variable placed
    // there, 0 for an unpopulated cell — the columns are read straight through the trace view
    const uint32_t* loop_slot1; const uint32_t* outer_slot1;
};
__device__ __forceinline__ size_t tiled(uint64_t n_cells, uint32_t cell, uint32_t lane) {
    return zkgeom::offset(n_cells, cell, lane);  // n_cells = the geometry word of the store
}
// a value of a NARROW store (store_geom.hpp: the geometry word carries zkgeom::NARROW, the view holds address words + 1)
__device__ __forceinline__ uint64_t narrow_value(const uint64_t* __restrict__ cells, uint64_t geom, uint32_t aw, uint32_t lane) {
    const uint8_t* __restrict__ p = reinterpret_cast<const uint8_t*>(cells) + zkgeom::narrow_byte_offset(geom, aw, lane);
    return (aw & zkgeom::AW_BYTE) ? (uint64_t)*p : *reinterpret_cast<const uint64_t*>(p);
}
// Batch form (round 4): a block owns one 64-lane TILE of the store (lanes = consecutive iterations, possibly of two neighbouring
// instances), 64 consecutive slots (trace rows of an iteration) and one column.  Reads are whole 512 B values of the tile (aligned:
// the block never straddles two tiles, which the per-instance form did whenever instance * limit was not a multiple of 64), writes
// are 512 B runs of a column (64 consecutive rows of one iteration).  Block order = tile-major, then slot group, then column, dealt to
// the XCDs in contiguous runs (blockIdx % 8 = XCD): the blocks in flight work on one or two tiles of the store, so the 2.1 other
// cells a variable occupies on average are found in L2 / the memory-side cache instead of being fetched from HBM again.
// Round 6: the loop store may be the NARROW store of the last fused step (the view then holds address words + 1): the columns are read from the
// one-byte / eight-byte slots directly, no widened copy of the store is made for them.
template <bool NARROW>
__global__ __launch_bounds__(256) void k_trace_columns_batch_t(ColumnsDev d, uint32_t first_tile, uint32_t n_tiles, uint32_t slot_groups, uint64_t instance_stride,
                                                             uint32_t first_instance, uint32_t n_instances) {
    __shared__ uint64_t tile[64][65];
    const uint32_t n_blocks = gridDim.x;
    // XCD-contiguous block order: the blocks an XCD receives (every 8th) cover one contiguous range of work items
    const uint32_t per_xcd = (n_blocks + 7) / 8;
    const uint32_t work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (work >= n_tiles * slot_groups * d.n_cols) return;
    const uint32_t col = work % d.n_cols, sg = (work / d.n_cols) % slot_groups, t = first_tile + work / (d.n_cols * slot_groups);
    const uint32_t s0 = sg * 64;
    const uint32_t tx = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = t * 64 + tx;
    // the view entry (trace cell -> store slot + 1) of slot s0 + tx: one coalesced load per wavefront, broadcast per slot below, so the
    // sixteen value loads of a thread do not each wait for their own index load
    uint32_t my_s1 = 0;
    {
        const uint32_t slot = s0 + tx;
        if (slot < d.loop_slots) {
            const uint32_t cell = slot * d.n_cols + col;
            my_s1 = d.loop_slot1 ? d.loop_slot1[cell] : cell + 1;
        }
    }
    const uint64_t* __restrict__ src = d.loop_cells + tiled(d.loop_n_cells, 0, lane);   // slot 0 of this lane; slot s is s << tile_log2 further
    const uint32_t tsh = zkgeom::tile_log2(d.loop_n_cells);
    uint64_t v[16];
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t s1 = __builtin_amdgcn_readlane(my_s1, w + 4 * i);   // wave-uniform
        if constexpr (NARROW) v[i] = s1 ? narrow_value(d.loop_cells, d.loop_n_cells, s1 - 1, lane) : 0;
        else v[i] = s1 ? src[(size_t)(s1 - 1) << tsh] : 0;
    }
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) tile[w + 4 * i][tx] = v[i];
    __syncthreads();
    for (uint32_t kk = w; kk < 64; kk += 4) {
        const uint32_t ln = t * 64 + kk, inst = ln / d.limit, k = ln - inst * d.limit;
        const uint32_t slot = s0 + tx;
        if (inst >= first_instance && inst < first_instance + n_instances && slot < d.loop_slots)
            d.out[(size_t)(inst - first_instance) * instance_stride + (size_t)col * d.stride + (size_t)k * d.loop_slots + slot] = tile[tx][kk];
    }
}
// the outer scope's rows and the zero padding behind them
__global__ __launch_bounds__(256) void k_trace_columns_tail(ColumnsDev d, uint64_t instance_stride) {
    const uint32_t col = blockIdx.y;
    const uint32_t instance = d.instance + blockIdx.z;
    uint64_t* __restrict__ out = d.out + (size_t)blockIdx.z * instance_stride;
    const uint64_t first = (uint64_t)d.limit * d.loop_slots;
    const uint64_t row = first + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= d.n_rows_padded) return;
    const uint64_t s = row - first;
    uint64_t v = 0;
    if (s < d.outer_slots) {
        uint32_t cell = (uint32_t)s * d.n_cols + col;
        bool populated = true;
        if (d.outer_slot1) { const uint32_t s1 = d.outer_slot1[cell]; populated = s1 != 0; cell = s1 - 1; }
        if (populated) v = d.outer_cells[tiled(d.outer_n_cells, cell, instance)];
    }
    out[(size_t)col * d.stride + row] = v;
}

}  // namespace zkn
