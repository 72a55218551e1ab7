// bytebuf_macro.hpp — the STRUCTURE of ByteBuffer::fill_with_bytes (the keccak256 precompile's 192-byte shift-register buffer:
// /root/reference/src/keccak256_round_function/buffer/mod.rs:69-136 with trivial_mapping_function, mod.rs:100-142), written once and
// walked by
//   * the host gadget (circuits/keccak.cpp): values are variables; every primitive records its gate — and its witness op, unless the
//     fill is recorded as the macro-op ZK_OP_BYTEBUF_FILL;
//   * the device macro-op (kernels_engine2.hpp): values are small integers in registers / scratch; every primitive computes its result
//     and STREAMS OUT the same intermediates in the same order (~7.7 k values per fill; interpreted, the six fills of a cycle are
//     ~46 k ops over ~360 dependency levels);
//   * a counting backend (the number of outputs).
// Primitives a backend provides (outputs = values the trace holds, in this order):
//   sub1(x)          -> 1: x - 1                                 (FMA 1 x 1 + (p - 1) 1)
//   is_zero(x)       -> 2: flag = (x == 0), aux = x^-1 or 0; returns the flag                     (ZeroCheck)
//   select(s, a, b)  -> 1: s ? a : b                                                               (Selection)
//   band(a, b)       -> 1: a b;   bnot(a) -> 1: 1 - a;   bor(a, b) -> 2: a + b, then a + b - a b   (FMA)
//   mul(a, b)        -> 1: a b;   add(a, b) -> 1: a + b                                            (FMA)
//   zero()           -> the constant 0, no output
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define ZKB_HD __host__ __device__ __forceinline__
#else
#define ZKB_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKB_LOOP _Pragma("unroll 1")
#else
#define ZKB_LOOP
#endif

namespace zkb {

constexpr int BUF = 192, IN = 32;
constexpr int N_INPUTS = BUF + 1 + IN + 2;   // buffer bytes, filled, input bytes, offset, meaningful — the macro-op's operand order

// bytes[BUF] / filled: the buffer (updated in place); input[IN]: the 32 bytes read; offset: leading bytes to drop; meaningful: bytes to take.
// E = element type of the byte arrays, V = scalar value type (host: both variables; device: uint8_t and int32_t)
template <class B>
ZKB_HD void fill_with_bytes(B& be, typename B::E* bytes, typename B::V& filled, const typename B::E* input, typename B::V offset, typename B::V meaningful,
                            typename B::E* shifted /* [IN] scratch */, typename B::E* place /* [BUF] scratch */) {
    typedef typename B::V V;
    // shift register: drop `offset` leading bytes
    ZKB_LOOP
    for (int j = 0; j < IN; ++j) shifted[j] = input[j];
    V off = be.sub1(offset);
    ZKB_LOOP
    for (int i = 1; i < IN; ++i) {
        const V use_from_here = be.is_zero(off);
        off = be.sub1(off);
        ZKB_LOOP
        for (int j = 0; j < IN; ++j) {
            const V from = i + j < IN ? (V)input[i + j] : be.zero();
            shifted[j] = be.select(use_from_here, from, (V)shifted[j]);
        }
    }
    // "start here" markers: position `filled`, only if there is something to fill
    const V nothing = be.is_zero(meaningful);
    const V marker = be.bnot(nothing);
    V tmp = filled;
    ZKB_LOOP
    for (int j = 0; j < BUF; ++j) {
        const V here = be.is_zero(tmp);
        place[j] = be.band(here, marker);
        tmp = be.sub1(tmp);
    }
    V counter = meaningful;
    V exhausted = be.is_zero(meaningful);
    ZKB_LOOP
    for (int idx = 0; idx < IN; ++idx) {
        const V live = be.bnot(exhausted);
        const V src = be.mul((V)shifted[idx], live);
        ZKB_LOOP
        for (int j = idx; j < BUF; ++j) bytes[j] = be.select((V)place[j - idx], src, (V)bytes[j]);
        counter = be.sub1(counter);
        const V done = be.is_zero(counter);
        exhausted = be.bor(done, exhausted);
    }
    filled = be.add(filled, meaningful);
}

struct CountBackend {
    typedef int E;
    typedef int V;
    uint32_t n = 0;
    V sub1(V) { ++n; return 0; }
    V is_zero(V) { n += 2; return 0; }
    V select(V, V, V) { ++n; return 0; }
    V band(V, V) { ++n; return 0; }
    V bnot(V) { ++n; return 0; }
    V bor(V, V) { n += 2; return 0; }
    V mul(V, V) { ++n; return 0; }
    V add(V, V) { ++n; return 0; }
    V zero() { return 0; }
};
inline uint32_t n_outputs() {
    CountBackend cb;
    int bytes[BUF] = {0}, input[IN] = {0}, shifted[IN], place[BUF], filled = 0;
    fill_with_bytes(cb, bytes, filled, input, 0, 0, shifted, place);
    return cb.n;
}

// compute backend over small integers: every value of the structure is a byte, a flag or a counter in (-2^15, 2^15) for inputs in their
// ranges (bytes < 256, filled <= 192, offset < 32, meaningful <= 32 — the caller checks).  Emit receives the outputs in order as field
// elements: one(v) with v in [0, p).  inv(k) = k^-1 mod p for 0 < |k| < 4096 (the device's INV_SMALL table).
template <class Emit, class Inv>
struct ComputeBackend {
    typedef uint8_t E;
    typedef int32_t V;
    Emit& emit;
    Inv& inv;
    ZKB_HD ComputeBackend(Emit& e, Inv& i) : emit(e), inv(i) {}
    ZKB_HD static uint64_t fe(V v) { return v < 0 ? 0xFFFFFFFF00000001ull - (uint64_t)(-v) : (uint64_t)v; }
    ZKB_HD V out(V v) { emit.one(fe(v)); return v; }
    ZKB_HD V sub1(V x) { return out(x - 1); }
    ZKB_HD V is_zero(V x) {
        const V f = x == 0 ? 1 : 0;
        emit.one((uint64_t)f);
        emit.one(x == 0 ? 0ull : inv(x));
        return f;
    }
    ZKB_HD V select(V s, V a, V b) { return out(s ? a : b); }
    ZKB_HD V band(V a, V b) { return out(a * b); }
    ZKB_HD V bnot(V a) { return out(1 - a); }
    ZKB_HD V bor(V a, V b) { const V s = out(a + b); return out(s - a * b); }
    ZKB_HD V mul(V a, V b) { return out(a * b); }
    ZKB_HD V add(V a, V b) { return out(a + b); }
    ZKB_HD V zero() { return 0; }
};

}  // namespace zkb
