// tools/valu_rates.hip — issue cost of the integer / fp64 VALU instructions a Goldilocks multiplier can be built from, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/valu_rates.hip && /tmp/valu_rates
// One wavefront per SIMD (and 4 per SIMD) runs N instructions of one kind over 8 independent register chains; s_memtime around
// the loop gives shader cycles per wave-instruction (throughput when 4 waves share the SIMD, dependent-issue latency for the chain
// of length 1).  Why: k_witness_loop spends 55 % of its VALU time in Poseidon2 field multiplications built from v_mad_u64_u32
// (profiles/r3_loop_probe.md); the counters say one such instruction costs about as much as 10 plain ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, ADD_CO, FMA_F64, MUL_F64, ADD_F64, LSHL_B64, AND_B32, MAD_U32, FMA_F32, CVT_F64_U32, N_KINDS };
static const char* NAMES[N_KINDS] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_add_co_u32+v_addc_co_u32 (pair)", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_lshlrev_b64", "v_and_b32", "v_mad_u32_u24 (dup)", "v_fma_f32", "v_cvt_f64_u32"};

template <int K>
__global__ void k_rate(uint64_t* out, uint32_t iters, uint32_t seed) {
    uint64_t a[8];
    uint32_t b = seed | 1u, c = seed * 2654435761u | 3u;
    double d[8], e = 1.0000001 + seed, f = 0.99999 + seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (uint64_t)(threadIdx.x + i) * 0x9E3779B97F4A7C15ull + seed; d[i] = 1.0 + i + threadIdx.x; }
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
#define ONE(i)                                                                                                                      \
    if constexpr (K == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");             \
    else if constexpr (K == MUL_LO_U32) { uint32_t x = (uint32_t)a[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(c)); a[i] = x; } \
    else if constexpr (K == MUL_HI_U32) { uint32_t x = (uint32_t)a[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(c)); a[i] = x; } \
    else if constexpr (K == MAD_U32_U24 || K == MAD_U32) { uint32_t x = (uint32_t)a[i]; asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(b)); a[i] = x; } \
    else if constexpr (K == MUL_HI_U32_U24) { uint32_t x = (uint32_t)a[i]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c)); a[i] = x; } \
    else if constexpr (K == ADD_CO) { uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(b), "v"(c) : "vcc"); a[i] = lo | ((uint64_t)hi << 32); } \
    else if constexpr (K == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(e), "v"(f));                          \
    else if constexpr (K == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));                                     \
    else if constexpr (K == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(f));                                     \
    else if constexpr (K == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));                                          \
    else if constexpr (K == AND_B32) { uint32_t x = (uint32_t)a[i]; asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(c)); a[i] = x; } \
    else if constexpr (K == FMA_F32) { float x = __uint_as_float((uint32_t)a[i]); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(__uint_as_float(c))); a[i] = __float_as_uint(x); } \
    else if constexpr (K == CVT_F64_U32) { uint32_t x = (uint32_t)a[i]; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(x)); }
        REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
#undef ONE
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i] + (uint64_t)d[i];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = acc; }
}

template <int K>
static void run(uint64_t* d_out, int waves_per_simd) {
    const uint32_t iters = 4096;
    // one workgroup of 64 * waves threads per SIMD would need placement control; instead: 1 024 CU-filling blocks of 256 threads
    // x waves_per_simd (4 wavefronts of a block land on the 4 SIMDs of a CU)
    const int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL(k_rate<K>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(2 * blocks);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += (double)h[2 * i];
    const double per = sum / blocks / ((double)iters * 32);
    printf("  %-36s %d wave(s)/SIMD: %7.2f cycles per wave-instruction per wave  -> %6.2f cycles of the SIMD each\n", NAMES[K], waves_per_simd, per,
           per / waves_per_simd);
}

template <int K>
static void both(uint64_t* d) { run<K>(d, 1); run<K>(d, 4); }

int main() {
    uint64_t* d;
    hipMalloc(&d, 1 << 20);
    both<MAD_U64_U32>(d); both<MUL_LO_U32>(d); both<MUL_HI_U32>(d); both<MAD_U32_U24>(d); both<MUL_HI_U32_U24>(d); both<ADD_CO>(d);
    both<FMA_F64>(d); both<MUL_F64>(d); both<ADD_F64>(d); both<LSHL_B64>(d); both<AND_B32>(d); both<FMA_F32>(d); both<CVT_F64_U32>(d);
    return 0;
}
