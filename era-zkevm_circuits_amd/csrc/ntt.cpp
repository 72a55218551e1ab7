// ntt.cpp — host side of K11 (kernels_ntt.hpp): pass plans, twiddle tables, zk_ntt / zk_lde.
//
// SURVEY.md §8f rank 3.  The reference reaches this stage through boojum's prover after `into_assembly`
// (/root/reference/src/ram_permutation/mod.rs:554); nothing of it is in the tree, so the transform is defined in
// include/zkgl.h and pinned by oracle/zko_ntt.c (direct evaluation of the polynomial at the domain points).
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <vector>
#include "cs.hpp"
#include "device_api.hpp"

namespace zkgl {
namespace {

constexpr uint64_t P = 0xFFFFFFFF00000001ull;
uint64_t hmul(uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % P); }
uint64_t hpow(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    for (; e; e >>= 1, b = hmul(b, b))
        if (e & 1) r = hmul(r, b);
    return r;
}
uint64_t hinv(uint64_t a) { return hpow(a, P - 2); }

struct Tables { uint64_t *lo = nullptr, *hi = nullptr; };  // omega_{2^seg}^(+-j), omega_{2^seg}^(+-1024 j)

struct NttContext {
    std::mutex mu;
    uint64_t* root1024[2] = {nullptr, nullptr};
    std::map<std::pair<uint32_t, int>, Tables> tw;  // (seg, inverse)

    static uint64_t* upload(const std::vector<uint64_t>& v) {
        uint64_t* d = nullptr;
        if (hipMalloc((void**)&d, v.size() * 8) != hipSuccess) throw ZkError(ZK_ERR_HIP, "ntt: hipMalloc");
        if (hipMemcpy(d, v.data(), v.size() * 8, hipMemcpyHostToDevice) != hipSuccess) throw ZkError(ZK_ERR_HIP, "ntt: hipMemcpy");
        return d;
    }
    const uint64_t* roots(int inverse) {
        if (!root1024[inverse]) {
            uint64_t w = two_adic_root(10);
            if (inverse) w = hinv(w);
            std::vector<uint64_t> v(512);
            uint64_t x = 1;
            for (auto& e : v) { e = x; x = hmul(x, w); }
            root1024[inverse] = upload(v);
        }
        return root1024[inverse];
    }
    Tables twiddles(uint32_t seg, int inverse) {
        auto key = std::make_pair(seg, inverse);
        auto it = tw.find(key);
        if (it != tw.end()) return it->second;
        uint64_t w = two_adic_root(seg);
        if (inverse) w = hinv(w);
        std::vector<uint64_t> lo(1024), hi(seg > 10 ? (size_t)1 << (seg - 10) : 1);
        uint64_t x = 1;
        for (auto& e : lo) { e = x; x = hmul(x, w); }
        const uint64_t w1024 = hpow(w, 1024);
        x = 1;
        for (auto& e : hi) { e = x; x = hmul(x, w1024); }
        Tables t{upload(lo), upload(hi)};
        tw[key] = t;
        return t;
    }
};
NttContext& ctx() { static NttContext c; return c; }

struct Pass { uint32_t seg, r, t; };
// forward order; the inverse runs it backwards
std::vector<Pass> plan(uint32_t log_n) {
    std::vector<Pass> ps;
    const uint32_t k = (log_n + 9) / 10, base = log_n / k, extra = log_n % k;
    uint32_t seg = log_n;
    for (uint32_t i = 0; i < k; ++i) {
        const uint32_t r = base + (i < extra ? 1 : 0), lo = seg - r;
        uint32_t t = 13 - r;
        if (lo > 0) t = std::min(t, lo);          // strided pass: 2^t consecutive low indices
        else t = std::min(t, log_n - r);          // last pass: 2^t whole segments, one contiguous range
        ps.push_back({seg, r, t});
        seg = lo;
    }
    return ps;
}

void dev_check(int rc) { if (rc) throw ZkError(ZK_ERR_HIP, zkdev::last_hip_error()); }

// one transform over n_polys polynomials; src == dst allowed (in place).
//   inverse        : root sign and where the coset / scale factors act (coefficient side = input when forward, output when inverse)
//   natural_values : values in natural order and coefficients bit-reversed (mirrored butterfly structure)
void run(const uint64_t* src, uint64_t src_stride, uint64_t* dst, uint64_t dst_stride, uint32_t log_n, uint32_t n_polys, bool inverse,
         bool natural_values, const uint64_t* c_lo, const uint64_t* c_hi, hipStream_t st) {
    NttContext& c = ctx();
    std::vector<Pass> ps = plan(log_n);
    const size_t n = ps.size();
    const bool dit = inverse != natural_values;  // forward/bitrev values and inverse/natural values are decimation in frequency
    for (size_t q = 0; q < n; ++q) {
        const Pass& p = dit ? ps[n - 1 - q] : ps[q];
        zkdev::NttPassArgs a;
        a.src = q == 0 ? src : dst; a.src_stride = q == 0 ? src_stride : dst_stride;
        a.dst = dst; a.dst_stride = dst_stride;
        a.log_n = log_n; a.seg = p.seg; a.r = p.r; a.t = p.t;
        a.dit = dit; a.coset_store = inverse; a.coset_brev = natural_values;
        a.root1024 = c.roots(inverse);
        Tables t = c.twiddles(p.seg, inverse);
        a.tw_lo = t.lo; a.tw_hi = t.hi;
        const bool coefficient_side = inverse ? q == n - 1 : q == 0;
        a.c_lo = coefficient_side ? c_lo : nullptr; a.c_hi = coefficient_side ? c_hi : nullptr;
        dev_check(zkdev::launch_ntt_pass(a, n_polys, st));
    }
}

struct CosetTables {
    uint64_t *lo = nullptr, *hi = nullptr;
    hipStream_t st;
    CosetTables(uint32_t log_n, uint64_t base, uint64_t scale, hipStream_t s) : st(s) {
        const uint32_t n_hi = log_n > 10 ? 1u << (log_n - 10) : 1;
        if (hipMallocAsync((void**)&lo, (1024 + (size_t)n_hi) * 8, st) != hipSuccess) throw ZkError(ZK_ERR_HIP, "ntt: hipMallocAsync");
        hi = lo + 1024;
        dev_check(zkdev::launch_coset_tables(base, scale, lo, hi, n_hi, st));
    }
    ~CosetTables() { if (lo) hipFreeAsync(lo, st); }
};

}  // namespace

uint64_t two_adic_root(uint32_t log_n) {
    if (log_n > 32) throw ZkError(ZK_ERR_INVALID, "two_adic_root: log_n > 32");
    return hpow(7, (P - 1) >> log_n);
}

void ntt(uint64_t* d_data, uint32_t log_n, uint32_t n_polys, uint64_t stride, uint32_t mode, uint64_t coset_shift, void* stream) {
    if (log_n > 30) throw ZkError(ZK_ERR_INVALID, "ntt: log_n > 30");
    if (mode > 3) throw ZkError(ZK_ERR_INVALID, "ntt: unknown mode bits");
    if (coset_shift == 0 || coset_shift >= P) throw ZkError(ZK_ERR_INVALID, "ntt: coset shift must be a nonzero canonical field element");
    if (stride < ((uint64_t)1 << log_n) && n_polys > 1) throw ZkError(ZK_ERR_INVALID, "ntt: stride smaller than the polynomial");
    if (log_n == 0 || n_polys == 0) return;  // a constant is its own transform (g^0 = 1, 1/N = 1)
    const bool inverse = mode & ZK_NTT_INVERSE, natural = mode & ZK_NTT_NATURAL_VALUES;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> g(ctx().mu);
    if (!inverse) {
        if (coset_shift == 1) run(d_data, stride, d_data, stride, log_n, n_polys, false, natural, nullptr, nullptr, st);
        else {
            CosetTables ct(log_n, coset_shift, 1, st);
            run(d_data, stride, d_data, stride, log_n, n_polys, false, natural, ct.lo, ct.hi, st);
        }
    } else {
        CosetTables ct(log_n, hinv(coset_shift), hinv((uint64_t)1 << log_n), st);
        run(d_data, stride, d_data, stride, log_n, n_polys, true, natural, ct.lo, ct.hi, st);
    }
}

void lde(const uint64_t* d_coeffs, uint64_t src_stride, uint64_t* d_out, uint32_t log_n, uint32_t log_blowup, uint32_t n_polys,
         uint32_t mode, uint64_t coset_shift, void* stream) {
    if (log_n == 0 || log_n > 30 || log_blowup > 8 || log_n + log_blowup > 32) throw ZkError(ZK_ERR_INVALID, "lde: sizes out of range");
    if (mode & ~(uint32_t)ZK_NTT_NATURAL_VALUES) throw ZkError(ZK_ERR_INVALID, "lde: only ZK_NTT_NATURAL_VALUES may be set");
    if (coset_shift == 0 || coset_shift >= P) throw ZkError(ZK_ERR_INVALID, "lde: coset shift must be a nonzero canonical field element");
    if (n_polys == 0) return;
    const bool natural = mode & ZK_NTT_NATURAL_VALUES;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> g(ctx().mu);
    const uint64_t n = (uint64_t)1 << log_n, blow = (uint64_t)1 << log_blowup;
    const uint64_t eta = two_adic_root(log_n + log_blowup);
    for (uint64_t j = 0; j < blow; ++j) {
        // block j of every polynomial = its values on the coset g * eta^bitrev(j) * <omega_N>: with bit-reversed values the
        // blocks together are the size-(N * blow) transform of the zero-padded coefficients in bit-reversed order
        uint64_t c = 0;
        for (uint32_t b = 0; b < log_blowup; ++b) c |= ((j >> b) & 1) << (log_blowup - 1 - b);
        const uint64_t shift = hmul(coset_shift, hpow(eta, c));
        CosetTables ct(log_n, shift, 1, st);
        run(d_coeffs, src_stride, d_out + j * n, n * blow, log_n, n_polys, false, natural, ct.lo, ct.hi, st);
    }
}

}  // namespace zkgl
