// gadgets.hpp — host-side mirror of the boojum gadgets the reference circuits call
// (SURVEY.md §8b frequency list: Boolean::multi_and, UInt32::conditionally_select, Num::fma,
// Num::linear_combination, is_zero, overflowing_sub, decompose_into_bytes, queues, Poseidon2
// round function ...).  Every gadget records (a) the witness op(s) from the closed IR and (b) the
// gate instance(s) that constrain them, exactly once; nothing here computes field values.
// boojum itself is absent from /root/reference ([EXT]): the gate decompositions below are this
// engine's own and are documented in DESIGN.md §gadgets.
#pragma once
#include <array>
#include <utility>
#include <vector>
#include "cs.hpp"

namespace zkgl {

constexpr uint64_t GL_P = 0xFFFFFFFF00000001ull;
inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }

struct Num { zk_var v = ZK_VAR_NONE; };
struct Boolean { zk_var v = ZK_VAR_NONE; };
struct UInt8 { zk_var v = ZK_VAR_NONE; };
struct UInt32 { zk_var v = ZK_VAR_NONE; };
struct UInt256 { std::array<UInt32, 8> inner; };
template <int N>
struct QueueState {  // boojum::gadgets::queue::QueueState: head, tail.tail, tail.length
    std::array<Num, N> head, tail;
    UInt32 length;
};

class G {  // gadget context bound to one CS
  public:
    explicit G(CS& cs) : cs(cs) {}
    CS& cs;

    // ---- constants / allocation ----
    zk_var constant(uint64_t c) { return cs.alloc_constant(c); }
    zk_var one() { return constant(1); }
    zk_var zero() { return constant(0); }
    Num num_const(uint64_t c) { return {constant(c)}; }
    Boolean bool_const(bool b) { return {constant(b ? 1 : 0)}; }
    UInt32 u32_const(uint32_t c) { return {constant(c)}; }
    UInt256 u256_zero() { UInt256 r; for (auto& l : r.inner) l = u32_const(0); return r; }

    zk_var next_input();                 // next word of the current scope's input stream
    Num alloc_num();                     // Num::allocate (witness word, unconstrained)
    Boolean alloc_bool();                // Boolean::allocate: BooleanConstraintGate
    UInt32 alloc_u32_checked();          // UInt32::allocate_checked: byte split + range lookups
    UInt32 alloc_u32_unchecked();
    UInt256 alloc_u256_checked();
    template <int N> QueueState<N> alloc_queue_state();

    // ---- field arithmetic ----
    zk_var fma(uint64_t q, zk_var a, zk_var b, uint64_t l, zk_var c);  // q*a*b + l*c
    zk_var add(zk_var a, zk_var b) { return fma(1, a, one(), 1, b); }
    zk_var sub(zk_var a, zk_var b) { return fma(1, a, one(), GL_P - 1, b); }
    zk_var mul(zk_var a, zk_var b) { return fma(1, a, b, 0, a); }
    zk_var linear_combination(const std::vector<std::pair<zk_var, uint64_t>>& terms);
    void enforce_equal(zk_var a, zk_var b);
    void enforce_zero(zk_var a) { enforce_equal(a, zero()); }
    zk_var dot4(const zk_var a[4], const zk_var b[4]);

    // ---- booleans ----
    Boolean b_and(Boolean a, Boolean b) { return {mul(a.v, b.v)}; }
    Boolean b_or(Boolean a, Boolean b) { return {fma(GL_P - 1, a.v, b.v, 1, add(a.v, b.v))}; }
    Boolean negated(Boolean a) { return {fma(GL_P - 1, a.v, one(), 1, one())}; }
    Boolean multi_and(const std::vector<Boolean>& v);
    Boolean multi_or(const std::vector<Boolean>& v);
    void conditionally_enforce_true(Boolean b, Boolean cond);   // cond*b - cond == 0
    void enforce_bool_equal(Boolean a, Boolean b) { enforce_equal(a.v, b.v); }

    // ---- selection ----
    zk_var select(Boolean s, zk_var a, zk_var b);
    Num select(Boolean s, Num a, Num b) { return {select(s, a.v, b.v)}; }
    UInt32 select(Boolean s, UInt32 a, UInt32 b) { return {select(s, a.v, b.v)}; }
    Boolean select(Boolean s, Boolean a, Boolean b) { return {select(s, a.v, b.v)}; }
    UInt256 select(Boolean s, const UInt256& a, const UInt256& b);
    template <int N> QueueState<N> select(Boolean s, const QueueState<N>& a, const QueueState<N>& b);

    // ---- comparisons ----
    Boolean is_zero(zk_var x);
    Boolean equals(zk_var a, zk_var b) { return is_zero(sub(a, b)); }
    Boolean equals(const UInt256& a, const UInt256& b);

    // ---- integers ----
    std::array<UInt8, 4> decompose_into_bytes(UInt32 x);   // with range checks
    void range_check_u8_pair(zk_var a, zk_var b);
    void range_check_u32(zk_var x);
    // (a - b - borrow_in) mod 2^32, borrow_out   [UInt32::overflowing_sub_with_borrow_in]
    std::pair<UInt32, Boolean> overflowing_sub_with_borrow_in(UInt32 a, UInt32 b, Boolean borrow_in);
    std::pair<UInt32, Boolean> overflowing_add(UInt32 a, UInt32 b);
    UInt32 increment_unchecked(UInt32 a) { return {add(a.v, one())}; }
    // UInt32::div_by_constant: a = q*c + r, r < c (c <= 256), q range-checked
    std::pair<UInt32, UInt32> div_by_constant(UInt32 a, uint32_t c);
    // UInt8::overflowing_sub: (a - b) mod 2^8, borrow
    std::pair<UInt8, Boolean> overflowing_sub_u8(UInt8 a, UInt8 b);
    // a*b + c + d = lo + 2^32 hi  [UInt32::fma_with_carry, src/main_vm/opcodes/mod.rs:152-158]
    std::pair<UInt32, UInt32> u32_fma_with_carry(UInt32 a, UInt32 b, UInt32 c, UInt32 d);
    // the same over little-endian byte variables through U8x4FMAGate (the form the reference requires, opcodes/mod.rs:146): the
    // outputs are range-checked bytes; operands must be range-checked bytes already
    using Bytes4 = std::array<zk_var, 4>;
    std::pair<Bytes4, Bytes4> u8x4_fma_with_carry(const Bytes4& a, const Bytes4& b, const Bytes4& c, const Bytes4& d);

    // ---- lookups ----
    std::vector<zk_var> lookup(uint32_t table_id, const std::vector<zk_var>& keys, uint32_t n_vals);

    // ---- Poseidon2 round function (CircuitRoundFunction<F,8,12,4>) ----
    std::array<zk_var, 12> compute_round_function(const std::array<zk_var, 12>& state);
    std::array<zk_var, 12> simulate_round_function(const std::array<zk_var, 12>& state);  // witness only
    // simulate_round_function(cs, state, execute): zeros when `execute` is false (the reference passes the flag: log.rs:532, uma.rs:410)
    std::array<zk_var, 12> simulate_round_function(const std::array<zk_var, 12>& state, Boolean execute);
    std::array<zk_var, 12> empty_state();
    // commit_encoding (src/fsm_input_output/mod.rs:281-326) -> 4 commitment elements
    std::array<Num, 4> commit_encoding(const std::vector<zk_var>& input);
    bool use_poseidon_macro_op = true;  // ZK_OP_P2_ROUNDS vs decomposed primitive ops (identical trace)

    // ---- queue state helpers ----
    template <int N> void enforce_trivial_head(const QueueState<N>& q);
    template <int N> std::vector<zk_var> flatten(const QueueState<N>& q);

  private:
    uint32_t xor8_table();
};

// ---------------------------------------------------------------- templates
template <int N>
QueueState<N> G::alloc_queue_state() {
    QueueState<N> q;
    for (auto& h : q.head) h = alloc_num();
    for (auto& t : q.tail) t = alloc_num();
    q.length = alloc_u32_checked();
    return q;
}
template <int N>
QueueState<N> G::select(Boolean s, const QueueState<N>& a, const QueueState<N>& b) {
    QueueState<N> r;
    for (int i = 0; i < N; ++i) r.head[i] = select(s, a.head[i], b.head[i]);
    for (int i = 0; i < N; ++i) r.tail[i] = select(s, a.tail[i], b.tail[i]);
    r.length = select(s, a.length, b.length);
    return r;
}
template <int N>
void G::enforce_trivial_head(const QueueState<N>& q) {
    for (auto& h : q.head) enforce_zero(h.v);
}
template <int N>
std::vector<zk_var> G::flatten(const QueueState<N>& q) {
    std::vector<zk_var> o;
    for (auto& h : q.head) o.push_back(h.v);
    for (auto& t : q.tail) o.push_back(t.v);
    o.push_back(q.length.v);
    return o;
}

// markers of the standard tables (the Rust type identity in the reference)
enum TableMarker : uint32_t {
    TABLE_XOR8 = 1,       // boojum::gadgets::tables::Xor8Table (src/ram_permutation/mod.rs:500)
    TABLE_AND8 = 2,
    TABLE_BYTE_SPLIT1 = 3,
    TABLE_BINOP = 4,      // BinopTable: (a, b) -> and | or<<16 | xor<<32 (src/main_vm/opcodes/binop.rs:148-169)
    TABLE_VM_DECODE = 16, // opcode decode + price shaped table (src/tables/opcodes_decoding.rs:14-38)
    TABLE_VM_BITSHIFT = 17,
    TABLE_VM_CONDITIONAL = 18,
    TABLE_VM_REG_TO_BITMASK = 19,       // RegisterIndexToBitmaskTable (src/tables/integer_to_boolean_mask.rs:9-11)
    TABLE_VM_SUBPC_TO_BITMASK = 20,     // VMSubPCToBitmaskTable
    TABLE_VM_UMA_SHIFT_TO_BITMASK = 21, // UMAShiftToBitmaskTable
    TABLE_VM_UMA_PTR_READ_CLEANUP = 22, // UMAPtrReadCleanupTable (src/tables/uma_ptr_read_cleanup.rs:9)
    TABLE_TRIXOR4 = 49,   // boojum TriXor4Table (src/code_unpacker_sha256/mod.rs:557-558); the set lives in circuits/sha256_gadget4.hpp
};
void add_xor8_table(CS& cs);
void add_and8_table(CS& cs);
void add_binop_table(CS& cs);

}  // namespace zkgl
