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

#if defined(__HIPCC__)
#define ZKB_HD __host__ __device__ __forceinline__
#else
#define ZKB_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKB_LOOP _Pragma("unroll 1")
#define ZKB_UNROLL _Pragma("unroll")
#define ZKB_ROLLED _Pragma("unroll 1")
#else
#define ZKB_LOOP
#define ZKB_UNROLL
#define ZKB_ROLLED
#endif

namespace zkb {

constexpr int BUF = 192, IN = 32;
constexpr int N_INPUTS = BUF + 1 + IN + 2;   // buffer bytes, filled, input bytes, offset, meaningful — the macro-op's operand order

// The walk.  The backend owns the four byte arrays (buffer bytes, input, shifted, place) and exposes them through accessors whose index is
// a compile-time-unrollable inner-loop variable wherever the device needs one: the device keeps them packed in registers and slides
// them by one position per OUTER iteration (next_input_shift / next_placement), so that no array is ever indexed dynamically.
//   in_at(j)        input[i + j] of the current shift iteration i (0 beyond the end)       shifted(j) / set_shifted(j, v)
//   shifted_front() shifted[idx] of the current placement iteration idx                     place_rel(j) = place[j - idx]
//   set_place(j, v) called for j = 0, 1, ... BUF - 1 in this order, once each               byte(j) / set_byte(j, v)
template <class B>
ZKB_HD void fill_with_bytes(B& be, typename B::V& filled, typename B::V offset, typename B::V meaningful) {
    typedef typename B::V V;
    // shift register: drop `offset` leading bytes (shifted starts as the input)
    V off = be.sub1(offset);
    ZKB_LOOP
    for (int i = 1; i < IN; ++i) {
        const V use_from_here = be.is_zero(off);
        off = be.sub1(off);
        be.next_input_shift();   // in_at(j) is now input[i + j]
        ZKB_UNROLL
        for (int j = 0; j < IN; ++j) be.set_shifted(j, be.select(use_from_here, be.in_at(j), be.shifted(j)));
    }
    // "start here" markers: position `filled`, only if there is something to fill
    const V nothing = be.is_zero(meaningful);
    const V marker = be.bnot(nothing);
    V tmp = filled;
    ZKB_ROLLED
    for (int j = 0; j < BUF; ++j) {
        const V here = be.is_zero(tmp);
        be.set_place(j, be.band(here, marker));
        tmp = be.sub1(tmp);
    }
    V counter = meaningful;
    V exhausted = be.is_zero(meaningful);
    ZKB_LOOP
    for (int idx = 0; idx < IN; ++idx) {
        const V live = be.bnot(exhausted);
        const V src = be.mul(be.shifted_front(), live);
        ZKB_UNROLL
        for (int j = 0; j < BUF; ++j)
            if (j >= idx) be.set_byte(j, be.select(be.place_rel(j), src, be.byte(j)));
        counter = be.sub1(counter);
        const V done = be.is_zero(counter);
        exhausted = be.bor(done, exhausted);
        be.next_placement();   // shifted_front() -> shifted[idx + 1], place_rel(j) -> place[j - (idx + 1)]
    }
    filled = be.add(filled, meaningful);
}

// storage of the array-holding backends that index plainly (host gadget, counting): E = element type
template <class E>
struct PlainArrays {
    E bytes_[BUF], input_[IN], shifted_[IN], place_[BUF];
    E zero_;
    int i_ = 0, idx_ = 0;
    void load(const E* bytes, const E* input, E zero) {
        for (int j = 0; j < BUF; ++j) bytes_[j] = bytes[j];
        for (int j = 0; j < IN; ++j) input_[j] = shifted_[j] = input[j];
        zero_ = zero; i_ = 0; idx_ = 0;
    }
    void next_input_shift() { ++i_; }
    E in_at(int j) const { return i_ + j < IN ? input_[i_ + j] : zero_; }
    E shifted(int j) const { return shifted_[j]; }
    void set_shifted(int j, E v) { shifted_[j] = v; }
    void set_place(int j, E v) { place_[j] = v; }
    E shifted_front() const { return shifted_[idx_]; }
    E place_rel(int j) const { return place_[j - idx_]; }
    E byte(int j) const { return bytes_[j]; }
    void set_byte(int j, E v) { bytes_[j] = v; }
    void next_placement() { ++idx_; }
};

struct CountBackend : PlainArrays<int> {
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
};
inline uint32_t n_outputs() {
    CountBackend cb;
    int bytes[BUF] = {0}, input[IN] = {0}, filled = 0;
    cb.load(bytes, input, 0);
    fill_with_bytes(cb, filled, 0, 0);
    return cb.n;
}

// compute backend over small integers: every value of the structure is a byte, a flag or a counter in (-2^15, 2^15) for inputs in their
// ranges (bytes < 256, filled <= 192, offset < 32, meaningful <= 32 — the caller checks).  Emit receives the outputs in order as field
// elements: one(v) with v in [0, p).  inv(k) = k^-1 mod p for 0 < |k| < 4096 (the device's INV_SMALL table).
// The arrays are PACKED (four bytes per 32-bit word, the place flags one bit each) and every access is at an index that is a constant
// after unrolling: they live in registers.  The slides: next_input_shift drops the first byte of the input (zero enters at the end),
// next_placement drops the first byte of `shifted` and moves the place flags up by one position.
template <class Emit, class Inv>
struct ComputeBackend {
    typedef int32_t V;
    Emit& emit;
    Inv& inv;
    uint32_t bytes_[BUF / 4], in_[IN / 4], sh_[IN / 4], pl_[BUF / 32];
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
    // ---- the arrays
    ZKB_HD static V get8(const uint32_t* w, int j) { return (V)((w[j >> 2] >> (8 * (j & 3))) & 0xffu); }
    ZKB_HD static void put8(uint32_t* w, int j, V v) { w[j >> 2] = (w[j >> 2] & ~(0xffu << (8 * (j & 3)))) | (((uint32_t)v & 0xffu) << (8 * (j & 3))); }
    ZKB_HD void next_input_shift() {
#pragma unroll
        for (int k = 0; k < IN / 4; ++k) in_[k] = (in_[k] >> 8) | (k + 1 < IN / 4 ? in_[k + 1] << 24 : 0u);
    }
    ZKB_HD V in_at(int j) const { return get8(in_, j); }
    ZKB_HD V shifted(int j) const { return get8(sh_, j); }
    ZKB_HD void set_shifted(int j, V v) { put8(sh_, j, v); }
    // the walk sets place[0], place[1], ... place[BUF - 1] once each, in this order (a rolled loop): the flag enters at the top and the
    // flags slide down one position per call, so that after the last call bit j holds place[j] — no dynamic index
    ZKB_HD void set_place(int, V v) {
#pragma unroll
        for (int k = 0; k < BUF / 32; ++k) pl_[k] = (pl_[k] >> 1) | (k + 1 < BUF / 32 ? pl_[k + 1] << 31 : ((uint32_t)v & 1u) << 31);
    }
    ZKB_HD V shifted_front() const { return (V)(sh_[0] & 0xffu); }
    ZKB_HD V place_rel(int j) const { return (V)((pl_[j >> 5] >> (j & 31)) & 1u); }
    ZKB_HD V byte(int j) const { return get8(bytes_, j); }
    ZKB_HD void set_byte(int j, V v) { put8(bytes_, j, v); }
    ZKB_HD void next_placement() {
#pragma unroll
        for (int k = 0; k < IN / 4; ++k) sh_[k] = (sh_[k] >> 8) | (k + 1 < IN / 4 ? sh_[k + 1] << 24 : 0u);
#pragma unroll
        for (int k = BUF / 32 - 1; k >= 0; --k) pl_[k] = (pl_[k] << 1) | (k ? pl_[k - 1] >> 31 : 0u);
    }
};

}  // namespace zkb
