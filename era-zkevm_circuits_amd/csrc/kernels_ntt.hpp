// kernels_ntt.hpp — K6: batched Goldilocks NTT / coset LDE over device-resident polynomials.
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
// the 2^64 = 2^32 - 1 folding): on gfx950 the passes are VALU-bound, see profiles/.
#pragma once
#include "gl_device.hpp"

namespace zkn {

constexpr int TPB = 256;
constexpr uint32_t LOG_BLOCK_ELEMS = 13;  // 8192 elements = 64 KB LDS per block

struct PassDev {
    const uint64_t* src;
    uint64_t* dst;
    uint64_t src_stride, dst_stride;  // elements between consecutive polynomials
    uint32_t log_n, seg, r, t;        // lo = seg - r
    uint32_t inverse;
    const uint64_t* root1024;         // omega_1024^(+-j), j < 512
    const uint64_t* tw_lo;            // omega_{2^seg}^(+-j), j < 1024
    const uint64_t* tw_hi;            // omega_{2^seg}^(+-1024 j), j < max(1, 2^seg / 1024)
    const uint64_t* c_lo;             // coset / scale factor of natural index i: c_lo[i & 1023] * c_hi[i >> 10]; null = none
    const uint64_t* c_hi;
};

__device__ __forceinline__ uint32_t brev(uint32_t x, uint32_t bits) { return bits ? __brev(x) >> (32 - bits) : 0; }

__global__ __launch_bounds__(TPB) void k_ntt_pass(PassDev d) {
    extern __shared__ uint64_t s[];
    const uint32_t r = d.r, t = d.t, lo = d.seg - d.r;
    const uint32_t n_elems = 1u << (r + t);
    const bool strided = lo > 0;  // lo > 0: element e = (h << t) | l ; lo == 0: e = (l << r) | h, one contiguous range
    const uint32_t tiles_per_poly = 1u << (d.log_n - r - t);
    const uint32_t poly = blockIdx.x / tiles_per_poly, tile = blockIdx.x % tiles_per_poly;
    const uint64_t* __restrict__ src = d.src + (size_t)poly * d.src_stride;
    uint64_t* __restrict__ dst = d.dst + (size_t)poly * d.dst_stride;
    // global index of block element e
    uint32_t seg_base, low_base;
    if (strided) {
        const uint32_t tiles_per_seg = 1u << (lo - t);
        seg_base = (tile / tiles_per_seg) << d.seg;
        low_base = (tile % tiles_per_seg) << t;
    } else {
        seg_base = tile << (r + t);
        low_base = 0;
    }
    auto gidx = [&](uint32_t e) -> uint32_t {
        return strided ? seg_base + ((e >> t) << lo) + low_base + (e & ((1u << t) - 1)) : seg_base + e;
    };
    auto inter_twiddle = [&](uint32_t e) -> uint64_t {  // omega_{2^seg}^(+-(low index) * k), k = bitrev_r(h)
        const uint32_t h = e >> t, l = e & ((1u << t) - 1);
        const uint64_t ex = (uint64_t)(low_base + l) * brev(h, r);  // < 2^seg
        return gl::mul(d.tw_lo[ex & 1023], d.tw_hi[ex >> 10]);
    };
    auto coset = [&](uint32_t i) -> uint64_t { return gl::mul(d.c_lo[i & 1023], d.c_hi[i >> 10]); };

    // ---- load ----
    for (uint32_t e = threadIdx.x; e < n_elems; e += TPB) {
        const uint32_t i = gidx(e);
        uint64_t v = src[i];
        if (!d.inverse) {
            if (d.c_lo) v = gl::mul(v, coset(i));               // first forward pass: a[i] * g^i
        } else if (strided) {
            v = gl::mul(v, inter_twiddle(e));                   // undo the inter-step twiddle before the DIT stages
        }
        s[e] = v;
    }
    __syncthreads();
    // ---- r radix-2 stages over h ----
    const uint32_t sh = strided ? t : 0;          // log2 stride of h in LDS
    const uint32_t n_bf = n_elems >> 1;
    for (uint32_t st = 0; st < r; ++st) {
        const uint32_t lh = d.inverse ? st : r - 1 - st;  // log2(half): DIF walks big -> small, DIT small -> big
        for (uint32_t b = threadIdx.x; b < n_bf; b += TPB) {
            uint32_t p, l;
            if (strided) { l = b & ((1u << t) - 1); p = b >> t; }
            else { p = b & ((1u << (r - 1)) - 1); l = b >> (r - 1); }
            const uint32_t j = p & ((1u << lh) - 1);
            const uint32_t h0 = ((p >> lh) << (lh + 1)) | j;
            const uint32_t a0 = strided ? (h0 << sh) | l : (l << r) | h0;
            const uint32_t a1 = a0 + ((1u << lh) << sh);
            const uint64_t w = d.root1024[j << (9 - lh)];  // omega_{2^(lh+1)}^(+-j)
            const uint64_t u = s[a0], v = s[a1];
            if (!d.inverse) {
                s[a0] = gl::add(u, v);
                s[a1] = j ? gl::mul(gl::sub(u, v), w) : gl::sub(u, v);
            } else {
                const uint64_t vw = j ? gl::mul(v, w) : v;
                s[a0] = gl::add(u, vw);
                s[a1] = gl::sub(u, vw);
            }
        }
        __syncthreads();
    }
    // ---- store ----
    for (uint32_t e = threadIdx.x; e < n_elems; e += TPB) {
        const uint32_t i = gidx(e);
        uint64_t v = s[e];
        if (!d.inverse) {
            if (strided) v = gl::mul(v, inter_twiddle(e));
        } else if (d.c_lo) {
            v = gl::mul(v, coset(i));                           // last inverse pass: 1/N * g^-i
        }
        dst[i] = v;
    }
}

// c_lo[j] = scale * base^j (j < 1024), c_hi[j] = base^(1024 j) (j < n_hi)
__global__ __launch_bounds__(TPB) void k_coset_tables(uint64_t base, uint64_t scale, uint64_t* c_lo, uint64_t* c_hi, uint32_t n_hi) {
    const uint32_t g = blockIdx.x * TPB + threadIdx.x;
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

}  // namespace zkn
