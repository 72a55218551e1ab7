// gadgets.cpp — see gadgets.hpp.  Gate decompositions are this engine's own (boojum [EXT]).
#include "gadgets.hpp"
#include "poseidon_consts.hpp"

namespace zkgl {

namespace {
void gate_fma(CS& cs, uint64_t q, uint64_t l, zk_var a, zk_var b, zk_var c, zk_var d) {
    zk_var vars[4] = {a, b, c, d};
    uint64_t k[2] = {q, l};
    cs.place_gate(ZK_GATE_FMA, vars, 4, k, 2);
}
}  // namespace

zk_var G::next_input() {
    return cs.input(cs.in_loop() ? cs.loop_input_words() : cs.outer_input_words());
}
Num G::alloc_num() { return {next_input()}; }
Boolean G::alloc_bool() {
    zk_var v = next_input();
    cs.place_gate(ZK_GATE_BOOLEAN, &v, 1, nullptr, 0);
    return {v};
}
UInt32 G::alloc_u32_unchecked() { return {next_input()}; }
UInt32 G::alloc_u32_checked() {
    zk_var v = next_input();
    range_check_u32(v);
    return {v};
}
UInt256 G::alloc_u256_checked() {
    UInt256 r;
    for (auto& l : r.inner) l = alloc_u32_checked();
    return r;
}

zk_var G::fma(uint64_t q, zk_var a, zk_var b, uint64_t l, zk_var c) {
    zk_var d = cs.alloc_var();
    zk_var ins[3] = {a, b, c};
    uint64_t imm[2] = {q, l};
    cs.emit_op(ZK_OP_FMA, 0, 0, ins, 3, &d, 1, imm, 2);
    gate_fma(cs, q, l, a, b, c, d);
    return d;
}

// Num::linear_combination via ReductionGate<F,4> (src/base_structures/memory_query/mod.rs:113-130):
// first gate folds 4 terms, every further gate folds the running result + 3 terms.
zk_var G::linear_combination(const std::vector<std::pair<zk_var, uint64_t>>& terms) {
    if (terms.empty()) return zero();
    size_t pos = 0;
    zk_var acc = ZK_VAR_NONE;
    while (pos < terms.size() || acc == ZK_VAR_NONE) {
        zk_var t[4];
        uint64_t k[4];
        int n = 0;
        if (acc != ZK_VAR_NONE) { t[n] = acc; k[n] = 1; ++n; }
        while (n < 4 && pos < terms.size()) { t[n] = terms[pos].first; k[n] = terms[pos].second; ++n; ++pos; }
        while (n < 4) { t[n] = zero(); k[n] = 0; ++n; }
        zk_var r = cs.alloc_var();
        cs.emit_op(ZK_OP_LC4, 0, 0, t, 4, &r, 1, k, 4);
        zk_var vars[5] = {t[0], t[1], t[2], t[3], r};
        cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
        acc = r;
    }
    return acc;
}

void G::enforce_equal(zk_var a, zk_var b) {
    if (a == b) return;
    gate_fma(cs, 1, 0, a, one(), a, b);  // 1*a*1 + 0*a - b == 0
}

zk_var G::dot4(const zk_var a[4], const zk_var b[4]) {
    zk_var ins[8], vars[9];
    for (int i = 0; i < 4; ++i) { ins[2 * i] = a[i]; ins[2 * i + 1] = b[i]; }
    zk_var r = cs.alloc_var();
    cs.emit_op(ZK_OP_DOT4, 0, 0, ins, 8, &r, 1, nullptr, 0);
    for (int i = 0; i < 8; ++i) vars[i] = ins[i];
    vars[8] = r;
    cs.place_gate(ZK_GATE_DOT4, vars, 9, nullptr, 0);
    return r;
}

Boolean G::multi_and(const std::vector<Boolean>& v) {
    if (v.empty()) return bool_const(true);
    Boolean acc = v[0];
    for (size_t i = 1; i < v.size(); ++i) acc = b_and(acc, v[i]);
    return acc;
}
Boolean G::multi_or(const std::vector<Boolean>& v) {
    if (v.empty()) return bool_const(false);
    Boolean acc = v[0];
    for (size_t i = 1; i < v.size(); ++i) acc = b_or(acc, v[i]);
    return acc;
}
void G::conditionally_enforce_true(Boolean b, Boolean cond) {
    gate_fma(cs, 1, GL_P - 1, cond.v, b.v, cond.v, zero());  // cond*b - cond == 0
}

zk_var G::select(Boolean s, zk_var a, zk_var b) {
    if (a == b) return a;
    zk_var r = cs.alloc_var();
    zk_var ins[3] = {s.v, a, b};
    cs.emit_op(ZK_OP_SELECT, 0, 0, ins, 3, &r, 1, nullptr, 0);
    zk_var vars[4] = {a, b, s.v, r};
    cs.place_gate(ZK_GATE_SELECT, vars, 4, nullptr, 0);
    return r;
}
UInt256 G::select(Boolean s, const UInt256& a, const UInt256& b) {
    UInt256 r;
    for (int i = 0; i < 8; ++i) r.inner[i] = select(s, a.inner[i], b.inner[i]);
    return r;
}

Boolean G::is_zero(zk_var x) {
    zk_var outs[2];
    outs[0] = cs.alloc_var();  // flag
    outs[1] = cs.alloc_var();  // aux = x^-1 or 0
    cs.emit_op(ZK_OP_ISZERO, 0, 0, &x, 1, outs, 2, nullptr, 0);
    zk_var vars[3] = {x, outs[1], outs[0]};
    cs.place_gate(ZK_GATE_ZEROCHECK, vars, 3, nullptr, 0);
    return {outs[0]};
}
Boolean G::equals(const UInt256& a, const UInt256& b) {
    std::vector<Boolean> eq;
    for (int i = 0; i < 8; ++i) eq.push_back(equals(a.inner[i].v, b.inner[i].v));
    return multi_and(eq);
}

uint32_t G::xor8_table() { return cs.table_id(TABLE_XOR8); }

void G::range_check_u8_pair(zk_var a, zk_var b) {
    if (!cs.has_table(TABLE_XOR8) && cs.has_table(TABLE_TRIXOR4)) {
        // the reference's width-4 table set (code_unpacker_sha256/mod.rs:554-566 adds no 8-bit table): a byte is two 4-bit chunks,
        // x = lo + 16 hi, each chunk a key of a TriXor4 lookup (three chunks per lookup)
        std::vector<zk_var> nib;
        std::vector<zk_var> bytes = {a};
        if (b != a) bytes.push_back(b);
        for (zk_var x : bytes) {
            zk_var first = cs.alloc_vars(2);
            zk_var parts[2] = {first, first + 1};
            cs.emit_op(ZK_OP_SPLIT, 2, 4, &x, 1, parts, 2, nullptr, 0);
            zk_var vars[5] = {parts[0], parts[1], zero(), zero(), x};
            uint64_t k[4] = {1, 16, 0, 0};
            cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
            nib.push_back(parts[0]); nib.push_back(parts[1]);
        }
        const uint32_t t = cs.table_id(TABLE_TRIXOR4);
        for (size_t i = 0; i < nib.size(); i += 3) {
            zk_var keys[3] = {nib[i], i + 1 < nib.size() ? nib[i + 1] : zero(), i + 2 < nib.size() ? nib[i + 2] : zero()}, val;
            cs.lookup(t, keys, 3, &val, 1);
        }
        return;
    }
    zk_var keys[2] = {a, b}, val;
    cs.lookup(xor8_table(), keys, 2, &val, 1);
}

std::array<UInt8, 4> G::decompose_into_bytes(UInt32 x) {
    zk_var b[4];
    zk_var first = cs.alloc_vars(4);
    for (int i = 0; i < 4; ++i) b[i] = first + i;
    cs.emit_op(ZK_OP_SPLIT, 4, 8, &x.v, 1, b, 4, nullptr, 0);
    zk_var vars[5] = {b[0], b[1], b[2], b[3], x.v};
    uint64_t k[4] = {1, 1ull << 8, 1ull << 16, 1ull << 24};
    cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
    range_check_u8_pair(b[0], b[1]);
    range_check_u8_pair(b[2], b[3]);
    return {UInt8{b[0]}, UInt8{b[1]}, UInt8{b[2]}, UInt8{b[3]}};
}
void G::range_check_u32(zk_var x) { (void)decompose_into_bytes(UInt32{x}); }

std::pair<UInt32, Boolean> G::overflowing_sub_with_borrow_in(UInt32 a, UInt32 b, Boolean bin) {
    zk_var outs[2] = {cs.alloc_var(), cs.alloc_var()};  // diff, borrow
    zk_var ins[3] = {a.v, b.v, bin.v};
    cs.emit_op(ZK_OP_USUB, 32, 0, ins, 3, outs, 2, nullptr, 0);
    // b + diff + borrow_in = a + 2^32 * borrow_out
    zk_var vars[5] = {b.v, outs[0], bin.v, a.v, outs[1]};
    uint64_t k = 1ull << 32;
    cs.place_gate(ZK_GATE_UINTX_ADD, vars, 5, &k, 1);
    cs.place_gate(ZK_GATE_BOOLEAN, &outs[1], 1, nullptr, 0);
    range_check_u32(outs[0]);
    return {UInt32{outs[0]}, Boolean{outs[1]}};
}

std::pair<UInt32, Boolean> G::overflowing_add(UInt32 a, UInt32 b) {
    zk_var outs[2] = {cs.alloc_var(), cs.alloc_var()};
    zk_var ins[3] = {a.v, b.v, zero()};
    cs.emit_op(ZK_OP_UADD, 32, 0, ins, 3, outs, 2, nullptr, 0);
    zk_var vars[5] = {a.v, b.v, ins[2], outs[0], outs[1]};
    uint64_t k = 1ull << 32;
    cs.place_gate(ZK_GATE_UINTX_ADD, vars, 5, &k, 1);
    cs.place_gate(ZK_GATE_BOOLEAN, &outs[1], 1, nullptr, 0);
    range_check_u32(outs[0]);
    return {UInt32{outs[0]}, Boolean{outs[1]}};
}

std::pair<UInt32, UInt32> G::div_by_constant(UInt32 a, uint32_t c) {
    if (c == 0 || c > 256) throw ZkError(ZK_ERR_INVALID, "div_by_constant: divisor must be 1..256");
    zk_var outs[2] = {cs.alloc_var(), cs.alloc_var()};  // quotient, remainder
    cs.emit_op(ZK_OP_DIVREM, 0, c, &a.v, 1, outs, 2, nullptr, 0);
    enforce_equal(linear_combination({{outs[0], c}, {outs[1], 1}}), a.v);
    range_check_u32(outs[0]);
    range_check_u8_pair(outs[1], linear_combination({{one(), c - 1}, {outs[1], GL_P - 1}}));  // 0 <= r <= c-1
    return {UInt32{outs[0]}, UInt32{outs[1]}};
}

std::pair<UInt8, Boolean> G::overflowing_sub_u8(UInt8 a, UInt8 b) {
    zk_var outs[2] = {cs.alloc_var(), cs.alloc_var()};  // diff, borrow
    zk_var ins[3] = {a.v, b.v, zero()};
    cs.emit_op(ZK_OP_USUB, 8, 0, ins, 3, outs, 2, nullptr, 0);
    zk_var vars[5] = {b.v, outs[0], ins[2], a.v, outs[1]};  // b + diff + 0 = a + 2^8 * borrow
    uint64_t k = 1ull << 8;
    cs.place_gate(ZK_GATE_UINTX_ADD, vars, 5, &k, 1);
    cs.place_gate(ZK_GATE_BOOLEAN, &outs[1], 1, nullptr, 0);
    range_check_u8_pair(outs[0], outs[0]);
    return {UInt8{outs[0]}, Boolean{outs[1]}};
}

std::pair<UInt32, UInt32> G::u32_fma_with_carry(UInt32 a, UInt32 b, UInt32 c, UInt32 d) {
    zk_var outs[2] = {cs.alloc_var(), cs.alloc_var()};
    zk_var ins[4] = {a.v, b.v, c.v, d.v};
    cs.emit_op(ZK_OP_U32MULADD, 0, 0, ins, 4, outs, 2, nullptr, 0);
    zk_var vars[6] = {a.v, b.v, c.v, d.v, outs[0], outs[1]};
    cs.place_gate(ZK_GATE_U32_FMA, vars, 6, nullptr, 0);
    range_check_u32(outs[0]);
    range_check_u32(outs[1]);
    return {UInt32{outs[0]}, UInt32{outs[1]}};
}

std::pair<G::Bytes4, G::Bytes4> G::u8x4_fma_with_carry(const Bytes4& a, const Bytes4& b, const Bytes4& c, const Bytes4& d) {
    zk_var vars[26];
    for (int i = 0; i < 4; ++i) { vars[i] = a[i]; vars[4 + i] = b[i]; vars[8 + i] = c[i]; vars[12 + i] = d[i]; }
    zk_var first = cs.alloc_vars(10);
    for (int i = 0; i < 10; ++i) vars[16 + i] = first + i;
    cs.emit_op(ZK_OP_U8X4FMA, 0, 0, vars, 16, vars + 16, 10, nullptr, 0);
    cs.place_gate(ZK_GATE_U8X4_FMA, vars, 26, nullptr, 0);
    for (int i = 0; i < 10; i += 2) range_check_u8_pair(vars[16 + i], vars[17 + i]);   // lo, hi and the two carry bytes
    return {Bytes4{vars[16], vars[17], vars[18], vars[19]}, Bytes4{vars[20], vars[21], vars[22], vars[23]}};
}

std::vector<zk_var> G::lookup(uint32_t table_id, const std::vector<zk_var>& keys, uint32_t n_vals) {
    std::vector<zk_var> vals(n_vals);
    cs.lookup(table_id, keys.data(), (uint32_t)keys.size(), vals.data(), n_vals);
    return vals;
}

// ---------------------------------------------------------------- Poseidon2 in circuit
// Layout per permutation (3104 cells): MatrixMultiplicationGate(M_E) on the input, then per round
// for every active lane i: t = x_i + rc (FMA q=1,l=1 with the `one` and rc constant variables),
// x2 = t*t, x3 = x2*t, x4 = x2*x2, x7 = x3*x4 (4 FMA gates), then one MatrixMultiplicationGate
// (M_E after full rounds, M_I after partial rounds).  Output order == ZK_OP_P2_ROUNDS order.
std::array<zk_var, 12> G::compute_round_function(const std::array<zk_var, 12>& state) {
    const uint64_t* RC = poseidon_round_constants();
    std::vector<zk_var> produced;  // macro-op output order
    auto matmul = [&](uint32_t matrix, const std::array<zk_var, 12>& in) {
        std::array<zk_var, 12> out;
        zk_var first = cs.alloc_vars(12);
        zk_var vars[24];
        for (int i = 0; i < 12; ++i) { out[i] = first + i; vars[i] = in[i]; vars[12 + i] = out[i]; }
        if (!use_poseidon_macro_op) cs.emit_op(ZK_OP_MATMUL12, matrix, 0, in.data(), 12, out.data(), 12, nullptr, 0);
        cs.place_gate(matrix == 0 ? ZK_GATE_MATMUL12_EXT : ZK_GATE_MATMUL12_INT, vars, 24, nullptr, 0);
        for (int i = 0; i < 12; ++i) produced.push_back(out[i]);
        return out;
    };
    auto fma_step = [&](uint64_t q, uint64_t l, zk_var a, zk_var b, zk_var c) {
        zk_var d = cs.alloc_var();
        if (!use_poseidon_macro_op) {
            zk_var ins[3] = {a, b, c};
            uint64_t imm[2] = {q, l};
            cs.emit_op(ZK_OP_FMA, 0, 0, ins, 3, &d, 1, imm, 2);
        }
        gate_fma(cs, q, l, a, b, c, d);
        produced.push_back(d);
        return d;
    };
    zk_var one_v = one();
    std::array<zk_var, 12> cur = matmul(0, state);
    for (int r = 0; r < 30; ++r) {
        const bool full = r < 4 || r >= 26;
        const int n = full ? 12 : 1;
        for (int i = 0; i < n; ++i) {
            zk_var rc = constant(RC[12 * r + i]);
            zk_var t = fma_step(1, 1, cur[i], one_v, rc);
            zk_var x2 = fma_step(1, 0, t, t, t);
            zk_var x3 = fma_step(1, 0, x2, t, t);
            zk_var x4 = fma_step(1, 0, x2, x2, x2);
            zk_var x7 = fma_step(1, 0, x3, x4, x3);
            cur[i] = x7;
        }
        cur = matmul(full ? 0 : 1, cur);
    }
    if (use_poseidon_macro_op)
        cs.emit_op(ZK_OP_P2_ROUNDS, 0, 0, state.data(), 12, produced.data(), (uint32_t)produced.size(), nullptr, 0);
    return cur;
}

std::array<zk_var, 12> G::simulate_round_function(const std::array<zk_var, 12>& state) {
    std::array<zk_var, 12> out;
    zk_var first = cs.alloc_vars(12);
    for (int i = 0; i < 12; ++i) out[i] = first + i;
    cs.emit_op(ZK_OP_POSEIDON2, 0, 0, state.data(), 12, out.data(), 12, nullptr, 0);
    return out;
}
std::array<zk_var, 12> G::simulate_round_function(const std::array<zk_var, 12>& state, Boolean execute) {
    std::array<zk_var, 12> out;
    zk_var first = cs.alloc_vars(12);
    for (int i = 0; i < 12; ++i) out[i] = first + i;
    zk_var ins[13];
    for (int i = 0; i < 12; ++i) ins[i] = state[i];
    ins[12] = execute.v;
    cs.emit_op(ZK_OP_POSEIDON2, 1, 0, ins, 13, out.data(), 12, nullptr, 0);
    return out;
}

std::array<zk_var, 12> G::empty_state() {
    std::array<zk_var, 12> s;
    for (auto& v : s) v = zero();
    return s;
}

std::array<Num, 4> G::commit_encoding(const std::vector<zk_var>& input) {
    std::array<zk_var, 12> state = empty_state();
    // apply_length_specialization ([EXT]: the length goes to the last capacity element)
    state[11] = constant((uint64_t)input.size());
    size_t nchunks = (input.size() + 7) / 8;
    for (size_t c = 0; c < nchunks; ++c) {
        for (size_t j = 0; j < 8; ++j) {
            size_t k = 8 * c + j;
            state[j] = k < input.size() ? input[k] : zero();  // absorb with replacement, capacity kept
        }
        state = compute_round_function(state);
    }
    return {Num{state[0]}, Num{state[1]}, Num{state[2]}, Num{state[3]}};
}

// ---------------------------------------------------------------- standard tables
void add_xor8_table(CS& cs) {
    std::vector<uint64_t> rows;
    rows.reserve(65536 * 3);
    for (uint64_t a = 0; a < 256; ++a)
        for (uint64_t b = 0; b < 256; ++b) { rows.push_back(a); rows.push_back(b); rows.push_back(a ^ b); }
    cs.add_table(TABLE_XOR8, 2, 1, rows.data(), 65536);
}
void add_and8_table(CS& cs) {
    std::vector<uint64_t> rows;
    rows.reserve(65536 * 3);
    for (uint64_t a = 0; a < 256; ++a)
        for (uint64_t b = 0; b < 256; ++b) { rows.push_back(a); rows.push_back(b); rows.push_back(a & b); }
    cs.add_table(TABLE_AND8, 2, 1, rows.data(), 65536);
}
void add_binop_table(CS& cs) {
    std::vector<uint64_t> rows;
    rows.reserve(65536 * 3);
    for (uint64_t a = 0; a < 256; ++a)
        for (uint64_t b = 0; b < 256; ++b) {
            rows.push_back(a); rows.push_back(b);
            rows.push_back((a & b) | ((a | b) << 16) | ((a ^ b) << 32));
        }
    cs.add_table(TABLE_BINOP, 2, 1, rows.data(), 65536);
}

}  // namespace zkgl
