// circuits/vm_gadgets.hpp — the boojum gadget calls main_vm makes beyond what the queue circuits needed (SURVEY.md §8b
// frequency list: UInt16 arithmetic, mask / mask_negated, spread_into_bits, dot_product, parallel_select, from_le_bytes,
// add_no_overflow / non_widening_mul, ...).  boojum is absent from /root/reference ([EXT]): the gate decompositions are
// this engine's own, every one a witness op from the closed IR plus the gate(s) that constrain it.
#pragma once
#include "../gadgets.hpp"

namespace zkgl {

using V = zk_var;

struct VG : G {
    explicit VG(CS& cs) : G(cs) {}

    Boolean B(V v) { return Boolean{v}; }
    V c(uint64_t x) { return constant(x); }

    // ---- masks: Boolean/UIntX::mask(flag) = flag ? x : 0, mask_negated(flag) = flag ? 0 : x
    V mask(V x, Boolean f) { return mul(x, f.v); }
    V mask_negated(V x, Boolean f) { return fma(GL_P - 1, x, f.v, 1, x); }
    Boolean and_not(Boolean a, Boolean b) { return B(mask_negated(a.v, b)); }  // Boolean::mask_negated

    // ---- selection over arrays (parallel_select)
    template <size_t N>
    std::array<V, N> select_n(Boolean s, const std::array<V, N>& a, const std::array<V, N>& b) {
        std::array<V, N> r;
        for (size_t i = 0; i < N; ++i) r[i] = select(s, a[i], b[i]);
        return r;
    }
    void cond_enforce_equal(Boolean cond, V a, V b) { enforce_zero(mul(cond.v, sub(a, b))); }  // Num::conditionally_enforce_equal
    void cond_enforce_false(Boolean b, Boolean cond) { enforce_zero(mul(cond.v, b.v)); }       // Boolean::conditionally_enforce_false

    // ---- byte views
    // UInt32::decompose_into_bytes_unchecked (src/main_vm/register_input_view.rs:43): the consumers' table lookups range-check
    std::array<V, 4> bytes_unchecked(V x) {
        V b[4];
        V first = cs.alloc_vars(4);
        for (int i = 0; i < 4; ++i) b[i] = first + i;
        cs.emit_op(ZK_OP_SPLIT, 4, 8, &x, 1, b, 4, nullptr, 0);
        V vars[5] = {b[0], b[1], b[2], b[3], x};
        uint64_t k[4] = {1, 1ull << 8, 1ull << 16, 1ull << 24};
        cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
        return {b[0], b[1], b[2], b[3]};
    }
    std::array<V, 4> bytes_checked(V x) {
        auto d = decompose_into_bytes(UInt32{x});
        return {d[0].v, d[1].v, d[2].v, d[3].v};
    }
    V from_le_bytes2(V b0, V b1) { return linear_combination({{b0, 1}, {b1, 1ull << 8}}); }  // UInt16::from_le_bytes
    V from_le_bytes4(V b0, V b1, V b2, V b3) {                                              // UInt32::from_le_bytes
        return linear_combination({{b0, 1}, {b1, 1ull << 8}, {b2, 1ull << 16}, {b3, 1ull << 24}});
    }
    V low_u16(V x) {  // UInt32::low_u16: checked byte decomposition, low two bytes
        auto b = bytes_checked(x);
        return from_le_bytes2(b[0], b[1]);
    }
    // x < 2^(8 n): n byte chunks, pairwise range lookups (Num::constraint_bit_length_as_bytes / UIntX::from_variable_checked)
    void range_check_bytes(V x, int n) {
        std::vector<V> b(n);
        V first = cs.alloc_vars(n);
        for (int i = 0; i < n; ++i) b[i] = first + i;
        cs.emit_op(ZK_OP_SPLIT, n, 8, &x, 1, b.data(), n, nullptr, 0);
        std::vector<std::pair<V, uint64_t>> terms;
        for (int i = 0; i < n; ++i) terms.push_back({b[i], 1ull << (8 * i)});
        enforce_equal(linear_combination(terms), x);
        for (int i = 0; i < n; i += 2) range_check_u8_pair(b[i], i + 1 < n ? b[i + 1] : zero());
    }
    void range_check_u16(V x) { range_check_bytes(x, 2); }
    void range_check_u8(V x) { range_check_u8_pair(x, zero()); }

    // ---- UInt16 arithmetic
    std::pair<V, Boolean> u16_overflowing_add(V a, V b) {
        V outs[2] = {cs.alloc_var(), cs.alloc_var()};
        V ins[3] = {a, b, zero()};
        cs.emit_op(ZK_OP_UADD, 16, 0, ins, 3, outs, 2, nullptr, 0);
        V vars[5] = {a, b, ins[2], outs[0], outs[1]};
        uint64_t k = 1ull << 16;
        cs.place_gate(ZK_GATE_UINTX_ADD, vars, 5, &k, 1);
        cs.place_gate(ZK_GATE_BOOLEAN, &outs[1], 1, nullptr, 0);
        range_check_u16(outs[0]);
        return {outs[0], Boolean{outs[1]}};
    }
    std::pair<V, Boolean> u16_overflowing_sub(V a, V b) {
        V outs[2] = {cs.alloc_var(), cs.alloc_var()};  // diff, borrow
        V ins[3] = {a, b, zero()};
        cs.emit_op(ZK_OP_USUB, 16, 0, ins, 3, outs, 2, nullptr, 0);
        V vars[5] = {b, outs[0], ins[2], a, outs[1]};  // b + diff = a + 2^16 borrow
        uint64_t k = 1ull << 16;
        cs.place_gate(ZK_GATE_UINTX_ADD, vars, 5, &k, 1);
        cs.place_gate(ZK_GATE_BOOLEAN, &outs[1], 1, nullptr, 0);
        range_check_u16(outs[0]);
        return {outs[0], Boolean{outs[1]}};
    }
    // ---- UInt32 arithmetic
    std::pair<V, Boolean> u32_overflowing_add(V a, V b) {
        auto r = overflowing_add(UInt32{a}, UInt32{b});
        return {r.first.v, r.second};
    }
    std::pair<V, Boolean> u32_overflowing_sub(V a, V b) {
        auto r = overflowing_sub_with_borrow_in(UInt32{a}, UInt32{b}, bool_const(false));
        return {r.first.v, r.second};
    }
    V u32_add_no_overflow(V a, V b) { V r = add(a, b); range_check_u32(r); return r; }   // unsatisfiable on overflow
    V u32_sub_no_overflow(V a, V b) { V r = sub(a, b); range_check_u32(r); return r; }
    V u32_non_widening_mul(V a, V b) { V r = mul(a, b); range_check_u32(r); return r; }
    std::pair<V, V> u32_div_by_constant(V a, uint32_t k) {
        auto r = div_by_constant(UInt32{a}, k);
        return {r.first.v, r.second.v};
    }
    // unchecked 256-bit add / sub chains of the closures allocate_addition/subtraction_result_unchecked
    // (src/main_vm/opcodes/add_sub.rs:168-282): witness only, no gates
    std::pair<std::array<V, 8>, V> u256_add_witness(const std::array<V, 8>& a, const std::array<V, 8>& b) {
        std::array<V, 8> r;
        V carry = zero();
        for (int i = 0; i < 8; ++i) {
            V outs[2] = {cs.alloc_var(), cs.alloc_var()};
            V ins[3] = {a[i], b[i], carry};
            cs.emit_op(ZK_OP_UADD, 32, 0, ins, 3, outs, 2, nullptr, 0);
            r[i] = outs[0]; carry = outs[1];
        }
        return {r, carry};
    }
    std::pair<std::array<V, 8>, V> u256_sub_witness(const std::array<V, 8>& a, const std::array<V, 8>& b) {
        std::array<V, 8> r;
        V borrow = zero();
        for (int i = 0; i < 8; ++i) {
            V outs[2] = {cs.alloc_var(), cs.alloc_var()};
            V ins[3] = {a[i], b[i], borrow};
            cs.emit_op(ZK_OP_USUB, 32, 0, ins, 3, outs, 2, nullptr, 0);
            r[i] = outs[0]; borrow = outs[1];
        }
        return {r, borrow};
    }
    // allocate_mul_result_unchecked / allocate_div_result_unchecked (src/main_vm/opcodes/mul_div.rs:20-172): witness only
    std::pair<std::array<V, 8>, std::array<V, 8>> u256_wide_witness(uint32_t opcode, const std::array<V, 8>& a, const std::array<V, 8>& b) {
        V ins[16], outs[16];
        for (int i = 0; i < 8; ++i) { ins[i] = a[i]; ins[8 + i] = b[i]; }
        V first = cs.alloc_vars(16);
        for (int i = 0; i < 16; ++i) outs[i] = first + i;
        cs.emit_op(opcode, 0, 0, ins, 16, outs, 16, nullptr, 0);
        std::array<V, 8> lo, hi;
        for (int i = 0; i < 8; ++i) { lo[i] = outs[i]; hi[i] = outs[8 + i]; }
        return {lo, hi};
    }

    // ---- bit spreads
    // Num::spread_into_bits::<_, N>: N boolean variables whose weighted sum is x
    std::vector<Boolean> spread_into_bits(V x, int n) {
        std::vector<V> bits(n);
        V first = cs.alloc_vars(n);
        for (int i = 0; i < n; ++i) bits[i] = first + i;
        cs.emit_op(ZK_OP_SPLIT, n, 1, &x, 1, bits.data(), n, nullptr, 0);
        std::vector<std::pair<V, uint64_t>> terms;
        std::vector<Boolean> out;
        for (int i = 0; i < n; ++i) {
            cs.place_gate(ZK_GATE_BOOLEAN, &bits[i], 1, nullptr, 0);
            terms.push_back({bits[i], 1ull << i});
            out.push_back(Boolean{bits[i]});
        }
        enforce_equal(linear_combination(terms), x);
        return out;
    }
    // x -> (x mod 2^bits, x >> bits), relation through ONE FmaGate: 1 * one * lo + 2^bits * hi = x
    // (split_pc, src/main_vm/utils.rs:47-90; split_register_encoding_byte, decoded_opcode.rs:529-576)
    std::pair<V, V> split_low_fma(V x, int bits) {
        V o[2];
        V first = cs.alloc_vars(2);
        o[0] = first; o[1] = first + 1;
        cs.emit_op(ZK_OP_SPLIT, 2, bits, &x, 1, o, 2, nullptr, 0);
        V vars[4] = {one(), o[0], o[1], x};
        uint64_t k[2] = {1, 1ull << bits};
        cs.place_gate(ZK_GATE_FMA, vars, 4, k, 2);
        return {o[0], o[1]};
    }
    // boojum::gadgets::num::dot_product through chained DotProductGate<4> (src/main_vm/cycle.rs:204-246)
    V dot(const std::vector<V>& a, const std::vector<V>& b) {
        if (a.empty()) return zero();
        size_t pos = 0;
        V acc = ZK_VAR_NONE;
        while (pos < a.size()) {
            V x[4], y[4];
            int n = 0;
            if (acc != ZK_VAR_NONE) { x[n] = acc; y[n] = one(); ++n; }
            while (n < 4 && pos < a.size()) { x[n] = a[pos]; y[n] = b[pos]; ++n; ++pos; }
            while (n < 4) { x[n] = zero(); y[n] = zero(); ++n; }
            acc = dot4(x, y);
        }
        return acc;
    }
    Boolean all_zero(const std::array<V, 8>& limbs) {  // all_limbs_are_zero, src/main_vm/opcodes/mul_div.rs:174-182
        std::vector<Boolean> z;
        for (auto l : limbs) z.push_back(is_zero(l));
        return multi_and(z);
    }
};

}  // namespace zkgl
