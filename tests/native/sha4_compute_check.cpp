// tests/native/sha4_compute_check.cpp — zks4::ComputeBackend (the DEVICE's form of the 4-bit-chunk SHA-256 walk, csrc/sha256_macro4.hpp) compiled for
// the host: prints its output stream for a given state + block (hex words on the command line), or self-checks digests over random inputs.
//   sha4_compute_check stream <8 state words> <16 block words>    -> "n\n v0 v1 ...\n"   (compared with the oracle's restatement by the test)
//   sha4_compute_check <trials>                                   -> "ok <trials> trials ..." : final state == a plain software compression, count == CountBackend
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../era-zkevm_circuits_amd/csrc/sha256_macro.hpp"
#include "../../era-zkevm_circuits_amd/csrc/sha256_macro4.hpp"

struct Collect {
    std::vector<uint64_t> v;
    void one(uint64_t x) { v.push_back(x); }
};

static void soft(uint32_t h[8], const uint32_t w16[16]) {
    uint32_t w[64];
    auto ror = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    for (int i = 0; i < 16; ++i) w[i] = w16[i];
    for (int t = 16; t < 64; ++t) w[t] = w[t - 16] + (ror(w[t - 15], 7) ^ ror(w[t - 15], 18) ^ (w[t - 15] >> 3)) + w[t - 7] + (ror(w[t - 2], 17) ^ ror(w[t - 2], 19) ^ (w[t - 2] >> 10));
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int t = 0; t < 64; ++t) {
        uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + zks::K[t] + w[t];
        uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

static size_t run(uint32_t st[8], const uint32_t blk[16], std::vector<uint64_t>& out) {
    Collect c;
    zks4::ComputeBackend<Collect> be(c);
    uint32_t w[64];
    zks4::ComputeBackend<Collect>::Splits sp[64];
    zks4::compress(be, st, blk, w, sp, zks::K);
    out.swap(c.v);
    return out.size();
}

int main(int argc, char** argv) {
    if (argc == 26 && !strcmp(argv[1], "stream")) {
        uint32_t st[8], blk[16];
        for (int i = 0; i < 8; ++i) st[i] = (uint32_t)strtoul(argv[2 + i], nullptr, 16);
        for (int i = 0; i < 16; ++i) blk[i] = (uint32_t)strtoul(argv[10 + i], nullptr, 16);
        std::vector<uint64_t> out;
        run(st, blk, out);
        printf("%zu\n", out.size());
        for (uint64_t v : out) printf("%llx ", (unsigned long long)v);
        printf("\n");
        for (int i = 0; i < 8; ++i) printf("%x ", st[i]);
        printf("\n");
        return 0;
    }
    const int trials = argc > 1 ? atoi(argv[1]) : 100;
    zks4::CountBackend cb;
    int cst[8] = {0}, cblk[16] = {0}, cw[64];
    zks4::CountBackend::Splits csp[64];
    zks4::compress(cb, cst, cblk, cw, csp, zks::K);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
    for (int t = 0; t < trials; ++t) {
        uint32_t st[8], ref[8], blk[16];
        for (auto& v : st) v = t == 0 ? 0u : t == 1 ? 0xffffffffu : rnd();
        for (auto& v : blk) v = t == 0 ? 0u : t == 1 ? 0xffffffffu : rnd();
        memcpy(ref, st, sizeof ref);
        soft(ref, blk);
        std::vector<uint64_t> out;
        const size_t n = run(st, blk, out);
        if (n != cb.n) { printf("trial %d: %zu outputs, the counting backend says %u\n", t, n, cb.n); return 1; }
        if (memcmp(st, ref, sizeof ref)) { printf("trial %d: final state differs from the software compression\n", t); return 1; }
    }
    printf("ok %d trials: %u outputs per compression, final states equal the software compression\n", trials, cb.n);
    return 0;
}
