// kernels_primitives.hpp — hand-written gfx950 kernels for the named primitives of the hot path
// (SURVEY.md §2 kernel inventory): K1 Goldilocks column ops, K2 batched Poseidon2, K3 sponge
// chains, a9 MemoryQuery encoding, K4 permutation grand product.  Included once by
// zkgl_device.hip.  All of them are HBM- or integer-ALU-bound u64 work: no MFMA.
#pragma once
#include "poseidon2_device.hpp"

namespace zkk {

constexpr int TPB = 256;  // 4 wavefronts per workgroup

// ------------------------------------------------------------------------------------------
// K1: column arithmetic.  16 B per lane per access (ulonglong2) so that a wavefront moves 1 KiB
// per instruction; grid-stride so that ~2048 workgroups cover any n.
// ------------------------------------------------------------------------------------------
enum ColOp { COL_FMA, COL_ADD, COL_SUB, COL_MUL, COL_SELECT, COL_INV };

template <int OP>
__device__ __forceinline__ uint64_t col_apply(uint64_t a, uint64_t b, uint64_t c, uint64_t q, uint64_t l) {
    if (OP == COL_FMA) return gl::add(gl::mul(q, gl::mul(a, b)), gl::mul(l, c));
    if (OP == COL_ADD) return gl::add(a, b);
    if (OP == COL_SUB) return gl::sub(a, b);
    if (OP == COL_MUL) return gl::mul(a, b);
    if (OP == COL_SELECT) return a ? b : c;  // a = selector
    return gl::inv(a);
}

template <int OP>
__global__ __launch_bounds__(TPB) void k_col(uint64_t* __restrict__ dst, const uint64_t* __restrict__ a,
                                             const uint64_t* __restrict__ b, const uint64_t* __restrict__ c,
                                             uint64_t q, uint64_t l, size_t n) {
    size_t npair = n / 2;
    size_t tid = (size_t)blockIdx.x * TPB + threadIdx.x;
    size_t step = (size_t)gridDim.x * TPB;
    const ulonglong2* a2 = reinterpret_cast<const ulonglong2*>(a);
    const ulonglong2* b2 = reinterpret_cast<const ulonglong2*>(b);
    const ulonglong2* c2 = reinterpret_cast<const ulonglong2*>(c);
    ulonglong2* d2 = reinterpret_cast<ulonglong2*>(dst);
    for (size_t i = tid; i < npair; i += step) {
        ulonglong2 va = a2[i];
        ulonglong2 vb = (OP != COL_INV) ? b2[i] : va;
        ulonglong2 vc = (OP == COL_FMA || OP == COL_SELECT) ? c2[i] : va;
        ulonglong2 r;
        r.x = col_apply<OP>(va.x, vb.x, vc.x, q, l);
        r.y = col_apply<OP>(va.y, vb.y, vc.y, q, l);
        d2[i] = r;
    }
    if ((n & 1) && tid == 0) {
        size_t i = n - 1;
        uint64_t vb = (OP != COL_INV) ? b[i] : 0;
        uint64_t vc = (OP == COL_FMA || OP == COL_SELECT) ? c[i] : 0;
        dst[i] = col_apply<OP>(a[i], vb, vc, q, l);
    }
}

// ------------------------------------------------------------------------------------------
// K2: batched Poseidon2.  SoA form: one coalesced 512 B access per state element per wave.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_poseidon2_soa(uint64_t* __restrict__ st, size_t n, size_t stride) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    uint64_t s[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = st[j * stride + i];
    p2::permute(s);
#pragma unroll
    for (int j = 0; j < 12; ++j) st[j * stride + i] = s[j];
}

// AoS form: a workgroup owns 256 consecutive 96-byte states (24 KiB).  The wavefronts stream
// them in with 16 B/lane coalesced loads, stage them in LDS (rows padded to 13 words so the
// per-lane ds_read_b64 of a row is bank-conflict free: 26*t mod 64 is injective on a half-wave),
// permute in registers, and stream the result back the same way.
constexpr int AOS_ROW = 13;
__global__ __launch_bounds__(TPB) void k_poseidon2_aos(uint64_t* __restrict__ st, size_t n) {
    __shared__ uint64_t lds[TPB * AOS_ROW];
    size_t base = (size_t)blockIdx.x * TPB;
    size_t cnt = n - base < (size_t)TPB ? n - base : (size_t)TPB;
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(st + base * 12);
    size_t nvec = cnt * 6;  // 96 B = 6 x 16 B
    for (size_t v = threadIdx.x; v < nvec; v += TPB) {
        ulonglong2 x = src[v];
        size_t row = v / 6, pos = v % 6;
        lds[row * AOS_ROW + 2 * pos] = x.x;
        lds[row * AOS_ROW + 2 * pos + 1] = x.y;
    }
    __syncthreads();
    if (threadIdx.x < cnt) {
        uint64_t s[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) s[j] = lds[threadIdx.x * AOS_ROW + j];
        p2::permute(s);
#pragma unroll
        for (int j = 0; j < 12; ++j) lds[threadIdx.x * AOS_ROW + j] = s[j];
    }
    __syncthreads();
    ulonglong2* dstv = reinterpret_cast<ulonglong2*>(st + base * 12);
    for (size_t v = threadIdx.x; v < nvec; v += TPB) {
        size_t row = v / 6, pos = v % 6;
        ulonglong2 x;
        x.x = lds[row * AOS_ROW + 2 * pos];
        x.y = lds[row * AOS_ROW + 2 * pos + 1];
        dstv[v] = x;
    }
}

// ------------------------------------------------------------------------------------------
// K3: sponge chains.  lane == independent sponge; sequential in the chunk index.
// commit_encoding: /root/reference/src/fsm_input_output/mod.rs:281-326
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_commit_encoding(const uint64_t* __restrict__ in, size_t len, size_t n,
                                                         uint64_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    uint64_t s[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = 0;
    s[11] = gl::reduce((uint64_t)len);  // apply_length_specialization ([EXT]: last capacity slot)
    size_t nchunks = (len + 7) / 8;
    for (size_t c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            size_t k = 8 * c + j;
            s[j] = k < len ? in[k * n + i] : 0;  // absorb with replacement, capacity kept
        }
        p2::permute(s);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j * n + i] = s[j];
}

// full-state queue push chain: /root/reference/src/main_vm/utils.rs:194-213
__global__ __launch_bounds__(64) void k_queue_full_chain(const uint64_t* __restrict__ enc, size_t nq, size_t items,
                                                         uint64_t* __restrict__ tail_io,
                                                         uint64_t* __restrict__ states_out) {
    size_t q = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    uint64_t s[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = tail_io[q * 12 + j];
    for (size_t t = 0; t < items; ++t) {
        if (states_out) {
#pragma unroll
            for (int j = 0; j < 12; ++j) states_out[(q * items + t) * 12 + j] = s[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = enc[(q * items + t) * 8 + j];
        p2::permute(s);
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) tail_io[q * 12 + j] = s[j];
}

// ------------------------------------------------------------------------------------------
// a9: MemoryQuery::encode (/root/reference/src/base_structures/memory_query/mod.rs:103-221)
// 13 columns in, 8 columns out; pure bit packing (all packed words < 2^56 < p).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void memory_query_pack(const uint64_t q[13], uint64_t e[8]) {
    const uint64_t* v = q + 5;
    e[0] = q[0];
    e[1] = q[1];
    e[2] = q[2] + (q[3] << 32) + (q[4] << 33);
    uint64_t l5 = v[5], l6 = v[6], l7 = v[7];
    e[3] = v[0] + ((l5 & 0xffffffull) << 32);
    e[4] = v[1] + (((l5 >> 24) & 0xff) << 32) + ((l6 & 0xffff) << 40);
    e[5] = v[2] + (((l6 >> 16) & 0xffff) << 32) + ((l7 & 0xff) << 48);
    e[6] = v[3] + (((l7 >> 8) & 0xffffff) << 32);
    e[7] = v[4];
}

__global__ __launch_bounds__(TPB) void k_memory_query_encode(const uint64_t* __restrict__ q, size_t n,
                                                             uint64_t* __restrict__ enc) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    uint64_t qq[13], e[8];
#pragma unroll
    for (int f = 0; f < 13; ++f) qq[f] = q[f * n + i];
    memory_query_pack(qq, e);
#pragma unroll
    for (int j = 0; j < 8; ++j) enc[j * n + i] = e[j];
}

// ------------------------------------------------------------------------------------------
// a11: ExecutionContextRecord::encode (/root/reference/src/base_structures/vm_state/saved_context.rs:111-266)
// 42 columns in (declaration order, saved_context.rs:36-66), 32 columns out; pure bit packing (< 2^57).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_execution_context_encode(const uint64_t* __restrict__ rec, size_t n,
                                                                  uint64_t* __restrict__ enc) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    auto R = [&](int f) { return rec[(size_t)f * n + i]; };
    auto W = [&](int j, uint64_t v) { enc[(size_t)j * n + i] = v; };
#pragma unroll
    for (int k = 0; k < 4; ++k) { W(k, R(19 + k)); W(4 + k, R(23 + k)); W(23 + k, R(37 + k)); }
#pragma unroll
    for (int k = 0; k < 5; ++k) { W(8 + k, R(10 + k)); W(13 + k, R(k)); W(18 + k, R(5 + k)); }
    const uint64_t seg = R(27);
    W(27, R(15) + (R(28) << 32) + (R(34) << 48) + (R(32) << 56));
    W(28, R(16) + (R(29) << 32) + (R(35) << 48) + (R(33) << 56));
    W(29, R(31) + (R(30) << 32) + (R(36) << 48) + (R(41) << 56));
    W(30, R(17) + ((seg & 0xff) << 32) + (((seg >> 8) & 0xff) << 40));
    W(31, R(18) + (((seg >> 16) & 0xff) << 32) + (((seg >> 24) & 0xff) << 40));
}

// ------------------------------------------------------------------------------------------
// K4: permutation grand product (/root/reference/src/utils.rs:81-137, one repetition).
//   factor[i] = flags[i] ? ch[L] + sum_j enc[j][i]*ch[j] : 1
//   acc[i]    = init * prod_{t<=i} factor[t]
// Three passes: (1) factors + per-tile inclusive scan (TILE items per workgroup) + tile totals,
// (2) exclusive scan of the tile totals by one workgroup, (3) scale every tile by its prefix.
// ------------------------------------------------------------------------------------------
constexpr int GP_ITEMS = 4;                 // items per lane
constexpr int GP_TILE = TPB * GP_ITEMS;     // 1024 items per workgroup

__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d) {
    unsigned lo = __shfl_up((unsigned)v, d, 64), hi = __shfl_up((unsigned)(v >> 32), d, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    unsigned lo = __shfl((unsigned)v, src, 64), hi = __shfl((unsigned)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// inclusive multiplicative scan of one value per thread across the workgroup
__device__ __forceinline__ uint64_t block_scan_mul(uint64_t v, uint64_t* wave_tot /*[4] LDS*/, uint64_t& block_total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = shfl_up64(v, d);
        if (lane >= d) v = gl::mul(v, o);
    }
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    uint64_t pre = 1;
    for (int w = 0; w < wave; ++w) pre = gl::mul(pre, wave_tot[w]);
    uint64_t tot = 1;
    for (int w = 0; w < TPB / 64; ++w) tot = gl::mul(tot, wave_tot[w]);
    block_total = tot;
    __syncthreads();
    return gl::mul(v, pre);
}

__global__ __launch_bounds__(TPB) void k_gp_local(const uint64_t* __restrict__ enc, const uint64_t* __restrict__ flags,
                                                  const uint64_t* __restrict__ ch, size_t enc_len, size_t n,
                                                  uint64_t* __restrict__ acc, uint64_t* __restrict__ tile_tot) {
    __shared__ uint64_t wave_tot[TPB / 64];
    size_t base = (size_t)blockIdx.x * GP_TILE + (size_t)threadIdx.x * GP_ITEMS;
    uint64_t f[GP_ITEMS];
#pragma unroll
    for (int k = 0; k < GP_ITEMS; ++k) {
        size_t i = base + k;
        uint64_t c = 1;
        if (i < n && flags[i]) {
            c = ch[enc_len];
            for (size_t j = 0; j < enc_len; ++j) c = gl::fma(enc[j * n + i], ch[j], c);
        }
        f[k] = c;
    }
    // lane-local inclusive scan over its GP_ITEMS consecutive items
#pragma unroll
    for (int k = 1; k < GP_ITEMS; ++k) f[k] = gl::mul(f[k], f[k - 1]);
    uint64_t total;
    uint64_t incl = block_scan_mul(f[GP_ITEMS - 1], wave_tot, total);
    // exclusive prefix of this lane = incl / own  -> recompute via shuffle-free form:
    // prefix = (inclusive of previous lane) ; obtain from LDS-less trick: incl_prev = shfl_up(incl)
    uint64_t prev = shfl_up64(incl, 1);
    int lane = threadIdx.x & 63;
    __shared__ uint64_t wave_last[TPB / 64];
    if (lane == 63) wave_last[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (lane == 0) prev = (threadIdx.x == 0) ? 1 : wave_last[(threadIdx.x >> 6) - 1];
#pragma unroll
    for (int k = 0; k < GP_ITEMS; ++k) {
        size_t i = base + k;
        if (i < n) acc[i] = gl::mul(f[k], prev);
    }
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = total;
}

// one workgroup: exclusive scan over the tile totals (sequential over chunks of TPB)
__global__ __launch_bounds__(TPB) void k_gp_tiles(uint64_t* __restrict__ tile_tot, size_t ntiles, uint64_t init) {
    __shared__ uint64_t wave_tot[TPB / 64];
    __shared__ uint64_t wave_last[TPB / 64];
    uint64_t carry = init;
    for (size_t base = 0; base < ntiles; base += TPB) {
        size_t i = base + threadIdx.x;
        uint64_t v = i < ntiles ? tile_tot[i] : 1;
        uint64_t total;
        uint64_t incl = block_scan_mul(v, wave_tot, total);
        uint64_t prev = shfl_up64(incl, 1);
        int lane = threadIdx.x & 63;
        if (lane == 63) wave_last[threadIdx.x >> 6] = incl;
        __syncthreads();
        if (lane == 0) prev = (threadIdx.x == 0) ? 1 : wave_last[(threadIdx.x >> 6) - 1];
        if (i < ntiles) tile_tot[i] = gl::mul(prev, carry);  // exclusive prefix incl. init
        carry = gl::mul(carry, total);
        __syncthreads();
    }
}

__global__ __launch_bounds__(TPB) void k_gp_apply(uint64_t* __restrict__ acc, const uint64_t* __restrict__ tile_pre,
                                                  size_t n) {
    size_t base = (size_t)blockIdx.x * GP_TILE + (size_t)threadIdx.x * GP_ITEMS;
    uint64_t pre = tile_pre[blockIdx.x];
#pragma unroll
    for (int k = 0; k < GP_ITEMS; ++k) {
        size_t i = base + k;
        if (i < n) acc[i] = gl::mul(acc[i], pre);
    }
}

}  // namespace zkk
