// circuits/main_vm.cpp — host-side mirror of /root/reference/src/main_vm/ (the circuit BASELINE.json's metric is named after):
//   main_vm_entry_point        mod.rs:47-232             initial_bootloader_state   loading.rs:11-226
//   vm_cycle                   cycle.rs:28-795           create_prestate            pre_state.rs:71-519
//   perform_initial_decoding   decoded_opcode.rs:42-576  memory / operand helpers   utils.rs:14-522
//   opcode families            opcodes/{nop,add_sub,jump,binop,context,ptr,log,call_ret,mul_div,shifts,uma}.rs,
//                              opcodes/call_ret_impl/{near_call,far_call,ret}.rs
//   enforce_sponges            cycle.rs:732-784,937-957  add/sub + mul/div relations  opcodes/mod.rs:101-180
// recorded against the zkgl CS: `for _cycle_idx in 0..limit { state = vm_cycle(..) }` (mod.rs:102-110) is the loop scope, recorded
// once; the VmLocalState enters a cycle from the per-cycle input stream and is tied to the previous cycle's output by CARRY links.
//
// Everything `zkevm_opcode_defs` supplies (opcode table, bit positions, system_params) comes from the zk_opcode_defs blob handed
// to main_vm_configure (include/zkgl_vm.h).  The WitnessOracle (witness_oracle.rs:45-91) becomes stream words: every getter's
// answer for this cycle (zeros when `execute` is false, like the dummy answers of the trait) is a field of the loop input
// stream, every push_* / report_* call is a no-op on the device (the host already owns that data).  The layout of both streams
// is exported by zk_circuit_main_vm_layout; the carried part (the 243 VmLocalState words) is filled on the device by
// zk_cs_seed_carried_inputs from the raw oracle words.
//
// Engine-specific points: simulate_round_function takes the reference's `execute` flag (ZK_OP_POSEIDON2 a = 1: zeros
// when it is off, and a wavefront whose cycles all have it off skips the permutation); witness closures are the closed IR ops (U256 add/sub chains, ZK_OP_U256_MULWIDE /
// ZK_OP_U256_DIVREM, SPLIT); gate decompositions of boojum gadgets are this engine's own ([EXT]).
#include <cstring>
#include <string>
#include "../../../include/zkgl_vm.h"
#include "decommit_query.hpp"
#include "log_query.hpp"
#include "memory_query.hpp"
#include "vm_gadgets.hpp"

namespace zkgl {

namespace {

constexpr int NREG = ZK_VM_REGISTERS;
using W8 = std::array<V, 8>;
using S12 = std::array<V, 12>;
using S4 = std::array<V, 4>;
using A5 = std::array<V, 5>;

struct Reg { V ptr; W8 v; };
struct FlagsPort { V of, eq, gt; };

// ExecutionContextRecord (src/base_structures/vm_state/saved_context.rs:37-68); flatten order = flatten_as_variables (:279-323)
struct Ctx {
    A5 this_, caller, code_address;
    V code_page, base_page, heap_bound, aux_heap_bound;
    S4 rq_head, rq_tail;
    V rq_len;
    V pc, sp, eh, ergs;
    V is_static, is_kernel;
    V this_shard, caller_shard, code_shard;
    S4 ctx_u128;
    V is_local;
    std::vector<V> flatten() const {
        std::vector<V> o;
        for (auto x : this_) o.push_back(x);
        for (auto x : caller) o.push_back(x);
        for (auto x : code_address) o.push_back(x);
        o.insert(o.end(), {code_page, base_page, heap_bound, aux_heap_bound});
        for (auto x : rq_head) o.push_back(x);
        for (auto x : rq_tail) o.push_back(x);
        o.insert(o.end(), {rq_len, pc, sp, eh, ergs, is_static, is_kernel, this_shard, caller_shard, code_shard});
        for (auto x : ctx_u128) o.push_back(x);
        o.push_back(is_local);
        return o;
    }
    static Ctx unflatten(const V* f) {
        Ctx c;
        int n = 0;
        for (auto& x : c.this_) x = f[n++];
        for (auto& x : c.caller) x = f[n++];
        for (auto& x : c.code_address) x = f[n++];
        c.code_page = f[n++]; c.base_page = f[n++]; c.heap_bound = f[n++]; c.aux_heap_bound = f[n++];
        for (auto& x : c.rq_head) x = f[n++];
        for (auto& x : c.rq_tail) x = f[n++];
        c.rq_len = f[n++]; c.pc = f[n++]; c.sp = f[n++]; c.eh = f[n++]; c.ergs = f[n++];
        c.is_static = f[n++]; c.is_kernel = f[n++]; c.this_shard = f[n++]; c.caller_shard = f[n++]; c.code_shard = f[n++];
        for (auto& x : c.ctx_u128) x = f[n++];
        c.is_local = f[n++];
        return c;
    }
};
constexpr int CTX_WORDS = 42;

// Callstack / FullExecutionContext (src/base_structures/vm_state/callstack.rs:9-47)
struct Callstack {
    Ctx ctx;
    S4 fwd_tail;
    V fwd_len;
    V depth;
    S12 sponge;
};

// VmLocalState (src/base_structures/vm_state/mod.rs:92-109); flatten order = CSVarLengthEncodable (declaration order), 243 words
struct State {
    W8 prev_code_word;
    std::array<Reg, NREG> regs;
    FlagsPort flags;
    V timestamp, page_counter, tx_number, prev_code_page, prev_super_pc, pending_exception, ergs_per_pubdata;
    Callstack cs;
    S12 mem_tail;
    V mem_len;
    S12 dec_tail;
    V dec_len;
    S4 ctx_u128;
    std::vector<V> flatten() const {
        std::vector<V> o(prev_code_word.begin(), prev_code_word.end());
        for (auto& r : regs) { o.push_back(r.ptr); for (auto x : r.v) o.push_back(x); }
        o.insert(o.end(), {flags.of, flags.eq, flags.gt, timestamp, page_counter, tx_number, prev_code_page, prev_super_pc, pending_exception,
                           ergs_per_pubdata});
        auto cf = cs.ctx.flatten();
        o.insert(o.end(), cf.begin(), cf.end());
        for (auto x : cs.fwd_tail) o.push_back(x);
        o.push_back(cs.fwd_len); o.push_back(cs.depth);
        for (auto x : cs.sponge) o.push_back(x);
        for (auto x : mem_tail) o.push_back(x);
        o.push_back(mem_len);
        for (auto x : dec_tail) o.push_back(x);
        o.push_back(dec_len);
        for (auto x : ctx_u128) o.push_back(x);
        return o;
    }
    static State unflatten(const std::vector<V>& f) {
        State s;
        size_t n = 0;
        for (auto& x : s.prev_code_word) x = f[n++];
        for (auto& r : s.regs) { r.ptr = f[n++]; for (auto& x : r.v) x = f[n++]; }
        s.flags.of = f[n++]; s.flags.eq = f[n++]; s.flags.gt = f[n++];
        s.timestamp = f[n++]; s.page_counter = f[n++]; s.tx_number = f[n++]; s.prev_code_page = f[n++]; s.prev_super_pc = f[n++];
        s.pending_exception = f[n++]; s.ergs_per_pubdata = f[n++];
        s.cs.ctx = Ctx::unflatten(&f[n]); n += CTX_WORDS;
        for (auto& x : s.cs.fwd_tail) x = f[n++];
        s.cs.fwd_len = f[n++]; s.cs.depth = f[n++];
        for (auto& x : s.cs.sponge) x = f[n++];
        for (auto& x : s.mem_tail) x = f[n++];
        s.mem_len = f[n++];
        for (auto& x : s.dec_tail) x = f[n++];
        s.dec_len = f[n++];
        for (auto& x : s.ctx_u128) x = f[n++];
        return s;
    }
};
constexpr size_t STATE_WORDS = 8 + NREG * 9 + 3 + 7 + CTX_WORDS + 4 + 1 + 1 + 12 + 12 + 1 + 12 + 1 + 4;  // 243
// kind of every state word for allocation (VmLocalState::allocate = each field's own allocate): 0 Num, 1 Boolean, 8/16/32 UIntX
std::vector<int> state_word_kinds() {
    std::vector<int> k;
    auto rep = [&](int kind, int n) { for (int i = 0; i < n; ++i) k.push_back(kind); };
    rep(32, 8);
    for (int r = 0; r < NREG; ++r) { rep(1, 1); rep(32, 8); }
    rep(1, 3);
    rep(32, 4); rep(16, 1); rep(1, 1); rep(32, 1);
    rep(32, 15); rep(32, 4); rep(0, 8); rep(32, 1); rep(16, 3); rep(32, 1); rep(1, 2); rep(8, 3); rep(32, 4); rep(1, 1);  // saved context
    rep(0, 4); rep(32, 1); rep(32, 1); rep(0, 12);
    rep(0, 12); rep(32, 1); rep(0, 12); rep(32, 1); rep(32, 4);
    return k;
}

struct Sponge { Boolean flag; S12 init, fin; };  // (should_enforce, initial_state, final_state)

struct AddSubRelation { W8 a, b, c; V of; };
struct MulDivRelation { W8 a, b, rem, mul_low, mul_high; };

// StateDiffsAccumulator (src/main_vm/state_diffs.rs:19-99)
struct Diffs {
    struct Dst0 { bool can_write_into_memory; Boolean flag; Reg reg; };
    std::vector<Dst0> dst_0_values;
    std::vector<std::pair<Boolean, Reg>> dst_1_values;
    std::vector<std::pair<Boolean, FlagsPort>> flags;
    std::array<std::vector<std::pair<Boolean, Reg>>, NREG> specific_registers_updates;
    std::array<std::vector<Boolean>, NREG> specific_registers_zeroing;
    std::array<std::vector<Boolean>, NREG> remove_ptr_on_specific_registers;
    std::vector<Boolean> pending_exceptions;
    std::vector<std::pair<Boolean, V>> new_ergs_left_candidates, new_pc_candidates;
    std::vector<std::pair<Boolean, V>> new_tx_number, new_ergs_per_pubdata;
    std::vector<std::pair<Boolean, V>> new_heap_bounds, new_aux_heap_bounds;
    std::vector<std::pair<Boolean, S4>> context_u128_candidates;
    std::vector<std::pair<Boolean, Callstack>> callstacks;
    V memory_page_counters = ZK_VAR_NONE;
    struct Q12 { Boolean flag; V len; S12 state; };
    struct Q4 { Boolean flag; V len; S4 state; };
    std::vector<Q12> decommitment_queue_candidates, memory_queue_candidates;
    std::vector<Q4> log_queue_forward_candidates, log_queue_rollback_candidates;
    struct SpongeSet { Boolean applies; std::vector<Sponge> sponges; };
    std::vector<SpongeSet> sponge_candidates_to_run;
    std::vector<std::pair<Boolean, W8>> u32_conditional_range_checks;
    std::vector<std::pair<Boolean, std::vector<AddSubRelation>>> add_sub_relations;
    std::vector<std::pair<Boolean, std::vector<MulDivRelation>>> mul_div_relations;
};

// OpcodeBitmask + register selectors (src/main_vm/opcode_bitmask.rs:43-51; decoded_opcode.rs:30-36)
struct Decoded {
    std::vector<Boolean> type, variant, flag, src_mode, dst_mode;
    std::array<std::vector<Boolean>, 2> src_regs, dst_regs;
    V imm0, imm1;
};
struct RegView { std::array<V, 32> u8; W8 u32; V is_ptr; };  // RegisterInputView
struct MemLoc { V page, index; };
struct Common {  // CommonOpcodeState (src/main_vm/pre_state.rs:22-34)
    FlagsPort reseted_flags, current_flags;
    Decoded dec;
    Reg src0, src1;
    RegView src0_view, src1_view;
    V ts_read, ts_first, ts_second, ts_dst;
};
struct Carry {  // AfterDecodingCarryParts (:45-54)
    Boolean did_skip_cycle;
    V heap_page, aux_heap_page, next_pc, preliminary_ergs_left;
    Sponge src0_read_sponge;
    MemLoc dst0_location;
    Boolean dst0_performs_memory_access;
};

struct Layout {
    std::string* text;
    CS& cs;
    const char* scope;
    void field(const char* name, uint32_t first, uint32_t n) {
        *text += std::string(scope) + " " + name + " " + std::to_string(first) + " " + std::to_string(n) + "\n";
    }
};

class VmCircuit {
  public:
    VmCircuit(CS& cs, const zk_opcode_defs& d) : g(cs), cs(cs), D(d) {}
    void entry_point(uint32_t limit);

  private:
    VG g;
    CS& cs;
    const zk_opcode_defs& D;
    uint32_t T_DECODE = 0, T_COND = 0, T_REGMASK = 0, T_SUBPC = 0, T_UMASHIFT = 0, T_UMACLEAN = 0, T_BITSHIFT = 0, T_BINOP = 0;
    V gctx_zkporter = ZK_VAR_NONE;   // GlobalContext (loop imports)
    W8 gctx_default_aa{};

    uint32_t P(int which) const { return D.params[which]; }
    uint32_t words() { return cs.in_loop() ? cs.loop_input_words() : cs.outer_input_words(); }
    // ---- stream fields (recorded into the layout text)
    V in_num(const char* name) { lay(name, 1); return g.next_input(); }
    V in_bool(const char* name) { lay(name, 1); return g.alloc_bool().v; }
    V in_u32(const char* name) { lay(name, 1); return g.alloc_u32_checked().v; }
    template <size_t N> std::array<V, N> in_nums(const char* name) { lay(name, N); std::array<V, N> r; for (auto& x : r) x = g.next_input(); return r; }
    template <size_t N> std::array<V, N> in_u32s(const char* name) { lay(name, N); std::array<V, N> r; for (auto& x : r) x = g.alloc_u32_checked().v; return r; }
    void lay(const char* name, uint32_t n) {
        cs.input_layout += std::string(cs.in_loop() ? "loop " : "outer ") + name + " " + std::to_string(words()) + " " + std::to_string(n) + "\n";
    }
    V alloc_kind(int kind) {
        V v = g.next_input();
        if (kind == 1) cs.place_gate(ZK_GATE_BOOLEAN, &v, 1, nullptr, 0);
        else if (kind == 8) g.range_check_u8(v);
        else if (kind == 16) g.range_check_u16(v);
        else if (kind == 32) g.range_check_u32(v);
        return v;
    }

    Boolean type_bit(const Decoded& d, int family) { return d.type[family]; }
    Boolean variant_bit(const Decoded& d, int which) { return d.variant[D.variant_idx[which]]; }
    Boolean flag_bit(const Decoded& d, int which) { return d.flag[D.flag_idx[which]]; }

    Reg select(Boolean s, const Reg& a, const Reg& b) { return Reg{g.select(s, a.ptr, b.ptr), g.select_n(s, a.v, b.v)}; }
    Reg zero_reg() { Reg r; r.ptr = g.zero(); for (auto& x : r.v) x = g.zero(); return r; }
    FlagsPort select(Boolean s, const FlagsPort& a, const FlagsPort& b) { return {g.select(s, a.of, b.of), g.select(s, a.eq, b.eq), g.select(s, a.gt, b.gt)}; }
    Ctx select(Boolean s, const Ctx& a, const Ctx& b) {
        auto fa = a.flatten(), fb = b.flatten();
        std::vector<V> r(CTX_WORDS);
        for (int i = 0; i < CTX_WORDS; ++i) r[i] = g.select(s, fa[i], fb[i]);
        return Ctx::unflatten(r.data());
    }
    Callstack select(Boolean s, const Callstack& a, const Callstack& b) {
        Callstack r;
        r.ctx = select(s, a.ctx, b.ctx);
        r.fwd_tail = g.select_n(s, a.fwd_tail, b.fwd_tail);
        r.fwd_len = g.select(s, a.fwd_len, b.fwd_len);
        r.depth = g.select(s, a.depth, b.depth);
        r.sponge = g.select_n(s, a.sponge, b.sponge);
        return r;
    }
    Sponge select(Boolean s, const Sponge& a, const Sponge& b) {
        return Sponge{g.select(s, a.flag, b.flag), g.select_n(s, a.init, b.init), g.select_n(s, a.fin, b.fin)};
    }
    Ctx uninitialized_ctx() {
        std::vector<V> z(CTX_WORDS, g.zero());
        return Ctx::unflatten(z.data());
    }
    std::array<V, 8> memory_query_encode(V ts, V page, V index, V rw, V is_ptr, const W8& value) {
        MemoryQuery q;
        q.timestamp = UInt32{ts}; q.memory_page = UInt32{page}; q.index = UInt32{index}; q.rw_flag = Boolean{rw}; q.is_ptr = Boolean{is_ptr};
        for (int i = 0; i < 8; ++i) q.value.inner[i] = UInt32{value[i]};
        return encode_memory_query(g, q);
    }
    S12 absorb8(const std::array<V, 8>& enc, const S12& cap) {
        S12 s;
        for (int i = 0; i < 8; ++i) s[i] = enc[i];
        for (int i = 8; i < 12; ++i) s[i] = cap[i];
        return s;
    }
    S12 simulate(const S12& in, Boolean execute) { return g.simulate_round_function(in, execute); }
    std::array<V, 32> encode_ctx(const Ctx& c);
    std::array<V, 20> encode_log(const LogQuery& q) { return encode_log_query(g, q); }

    // pieces of the cycle
    std::pair<V, V> split_pc(V pc);
    Decoded perform_initial_decoding(const std::array<V, 2>& raw_opcode, V encoded_flags, Boolean is_kernel, Boolean is_static,
                                     Boolean callstack_is_full, V ergs_left, Boolean did_skip, V* dirty_ergs_left);
    void create_prestate(State& st, Common& common, Carry& carry);
    RegView view(const Reg& r);
    void apply_add_sub(const State&, const Common&, const Carry&, Diffs&);
    void apply_jump(const State&, const Common&, const Carry&, Diffs&);
    void apply_binop(const State&, const Common&, const Carry&, Diffs&);
    void apply_context(const State&, const Common&, const Carry&, Diffs&);
    void apply_ptr(const State&, const Common&, const Carry&, Diffs&);
    void apply_log(const State&, const Common&, const Carry&, Diffs&);
    void apply_calls_and_ret(const State&, const Common&, const Carry&, Diffs&);
    void apply_mul_div(const State&, const Common&, const Carry&, Diffs&);
    void apply_shifts(const State&, const Common&, const Carry&, Diffs&);
    void apply_uma(const State&, const Common&, const Carry&, Diffs&);
    void enforce_addition_relation(const AddSubRelation& r);
    void enforce_mul_relation(const MulDivRelation& r);
    State vm_cycle(State st);
    State initial_bootloader_state(V mem_len, const S12& mem_tail, V dec_len, const S12& dec_tail, const S4& rollback_tail);

    // call / ret pieces
    struct FatPtr { V offset, page, start, length; };
    struct CommonAbi { FatPtr fat_ptr; V upper_bound; Boolean generally_invalid, is_non_addressable; };
    struct Forwarding { Boolean use_heap, use_aux_heap, forward_fat_pointer; };
    struct FarAbi { V ergs_passed, shard_id; Boolean constructor_call, system_call; };
    FatPtr mask_into_empty(const FatPtr& p, Boolean f) { return {g.mask_negated(p.offset, f), g.mask_negated(p.page, f), g.mask_negated(p.start, f), g.mask_negated(p.length, f)}; }
    FatPtr readjust(const FatPtr& p) { return {g.zero(), p.page, g.u32_add_no_overflow(p.start, p.offset), g.u32_sub_no_overflow(p.length, p.offset)}; }
    FatPtr select(Boolean s, const FatPtr& a, const FatPtr& b) { return {g.select(s, a.offset, b.offset), g.select(s, a.page, b.page), g.select(s, a.start, b.start), g.select(s, a.length, b.length)}; }
    Reg fat_ptr_into_register(const FatPtr& p) { Reg r = zero_reg(); r.ptr = g.one(); r.v[0] = p.offset; r.v[1] = p.page; r.v[2] = p.start; r.v[3] = p.length; return r; }
};

// ExecutionContextRecord::encode — src/base_structures/vm_state/saved_context.rs:111-266
std::array<V, 32> VmCircuit::encode_ctx(const Ctx& c) {
    const uint64_t S32 = 1ull << 32, S40 = 1ull << 40, S48 = 1ull << 48, S56 = 1ull << 56;
    std::array<V, 32> v;
    for (int i = 0; i < 4; ++i) { v[i] = c.rq_head[i]; v[4 + i] = c.rq_tail[i]; }
    for (int i = 0; i < 5; ++i) { v[8 + i] = c.code_address[i]; v[13 + i] = c.this_[i]; v[18 + i] = c.caller[i]; }
    for (int i = 0; i < 4; ++i) v[23 + i] = c.ctx_u128[i];
    v[27] = g.linear_combination({{c.code_page, 1}, {c.pc, S32}, {c.this_shard, S48}, {c.is_static, S56}});
    v[28] = g.linear_combination({{c.base_page, 1}, {c.sp, S32}, {c.caller_shard, S48}, {c.is_kernel, S56}});
    v[29] = g.linear_combination({{c.ergs, 1}, {c.eh, S32}, {c.code_shard, S48}, {c.is_local, S56}});
    auto d = g.bytes_checked(c.rq_len);
    v[30] = g.linear_combination({{c.heap_bound, 1}, {d[0], S32}, {d[1], S40}});
    v[31] = g.linear_combination({{c.aux_heap_bound, 1}, {d[2], S32}, {d[3], S40}});
    return v;
}

// split_pc — src/main_vm/utils.rs:47-104: (super_pc: UInt16 checked, bitspread of sub_pc from VMSubPCToBitmaskTable)
std::pair<V, V> VmCircuit::split_pc(V pc) {
    auto [sub_pc, super_pc] = g.split_low_fma(pc, 2);
    g.range_check_u16(super_pc);
    auto vals = g.lookup(T_SUBPC, {sub_pc}, 2);
    return {super_pc, vals[0]};
}

// perform_initial_decoding — src/main_vm/decoded_opcode.rs:42-220 (with partially_decode_from_integer_and_resolve_condition
// :395-527, split_out_aux_bits :313-387, split_register_encoding_byte :529-576, reg_idx_into_bitspread :223-237)
Decoded VmCircuit::perform_initial_decoding(const std::array<V, 2>& raw_opcode, V encoded_flags, Boolean is_kernel, Boolean is_static,
                                            Boolean callstack_is_full, V ergs_left, Boolean did_skip, V* dirty_ergs_left) {
    // ---- partially_decode_from_integer_and_resolve_condition
    auto word0 = g.bytes_checked(raw_opcode[0]);
    V variant_and_cond = g.from_le_bytes2(word0[0], word0[1]);
    V variant_var, unused0, unused1, cond_var;
    {
        V o[2];
        V first = cs.alloc_vars(2);
        o[0] = first; o[1] = first + 1;
        cs.emit_op(ZK_OP_SPLIT, 2, 11, &variant_and_cond, 1, o, 2, nullptr, 0);  // OPCODES_TABLE_WIDTH = 11
        variant_var = o[0];
        V r[3];
        first = cs.alloc_vars(3);
        for (int i = 0; i < 3; ++i) r[i] = first + i;
        cs.emit_op(ZK_OP_SPLIT, 3, 1, &o[1], 1, r, 3, nullptr, 0);
        unused0 = r[0]; unused1 = r[1]; cond_var = r[2];
        cs.place_gate(ZK_GATE_BOOLEAN, &unused0, 1, nullptr, 0);
        cs.place_gate(ZK_GATE_BOOLEAN, &unused1, 1, nullptr, 0);
        // variant + 2^11 u0 + 2^12 u1 + 2^13 cond - word == 0
        V vars[5] = {variant_var, unused0, unused1, cond_var, variant_and_cond};
        uint64_t k[4] = {1, 1ull << 11, 1ull << 12, 1ull << 13};
        cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
    }
    auto dec = g.lookup(T_DECODE, {variant_var}, 2);  // (price, properties): range-checks the variant
    V opcode_cost = dec[0], opcode_properties = dec[1];
    Boolean condition = g.B(g.lookup(T_COND, {cond_var, encoded_flags}, 1)[0]);  // range-checks the 3 condition bits
    V src_regs_encoding = word0[2], dst_regs_encoding = word0[3];
    auto word1 = g.bytes_checked(raw_opcode[1]);
    V imm0 = g.from_le_bytes2(word1[0], word1[1]), imm1 = g.from_le_bytes2(word1[2], word1[3]);

    // ---- split_out_aux_bits
    const int NB = (int)D.description_bits_flattened;
    V main_props;
    std::array<Boolean, 3> aux;
    {
        const int nb = NB / 8;
        std::vector<V> by(nb + 1);
        V first = cs.alloc_vars(nb + 1);
        for (int i = 0; i <= nb; ++i) by[i] = first + i;
        cs.emit_op(ZK_OP_SPLIT, nb + 1, 8, &opcode_properties, 1, by.data(), nb + 1, nullptr, 0);  // description bytes, then the aux bits
        std::vector<std::pair<V, uint64_t>> terms;
        for (int i = 0; i < nb; ++i) terms.push_back({by[i], 1ull << (8 * i)});
        main_props = g.linear_combination(terms);
        for (int i = 0; i < nb; i += 2) g.range_check_u8_pair(by[i], i + 1 < nb ? by[i + 1] : g.zero());  // constraint_bit_length_as_bytes
        V r[3];
        first = cs.alloc_vars(3);
        for (int i = 0; i < 3; ++i) r[i] = first + i;
        cs.emit_op(ZK_OP_SPLIT, 3, 1, &by[nb], 1, r, 3, nullptr, 0);
        for (int i = 0; i < 3; ++i) { cs.place_gate(ZK_GATE_BOOLEAN, &r[i], 1, nullptr, 0); aux[i] = g.B(r[i]); }
        V vars[5] = {main_props, r[0], r[1], r[2], opcode_properties};
        uint64_t k[4] = {1, 1ull << NB, 1ull << (NB + 1), 1ull << (NB + 2)};
        cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
    }
    // ---- fast exceptions
    V masked_ergs_cost = g.mask_negated(opcode_cost, did_skip);
    auto [ergs_after, out_of_ergs] = g.u32_overflowing_sub(ergs_left, masked_ergs_cost);
    *dirty_ergs_left = g.mask_negated(ergs_after, out_of_ergs);
    Boolean requires_kernel = aux[D.aux_kernel_mode], can_static = aux[D.aux_static_ok], explicit_panic = aux[D.aux_explicit_panic];
    Boolean kernel_mode_exception = g.b_and(requires_kernel, g.negated(is_kernel));
    Boolean write_in_static_exception = g.b_and(is_static, g.negated(can_static));
    Boolean mask_into_panic = g.multi_or({explicit_panic, out_of_ergs, kernel_mode_exception, write_in_static_exception, callstack_is_full});
    const uint64_t props_mask = (1ull << NB) - 1;
    V props = g.select(mask_into_panic, g.c(D.panic_bitspread & props_mask), main_props);
    Boolean mask_into_nop = g.b_and(g.negated(mask_into_panic), g.negated(condition));
    props = g.select(mask_into_nop, g.c(D.nop_bitspread & props_mask), props);
    Boolean mask_any = g.b_or(mask_into_nop, mask_into_panic);

    const int NBITS = (int)(D.type_bits + D.variant_bits + D.flag_bits + D.src_mode_bits + D.dst_mode_bits);  // 38
    // the selected value has only the meaningful bits set (table rows, NOP / PANIC constants): spread exactly those
    auto bits = g.spread_into_bits(props, NBITS);
    V src_enc = g.mask_negated(src_regs_encoding, mask_any), dst_enc = g.mask_negated(dst_regs_encoding, mask_any);
    auto [src0_idx, src1_idx] = g.split_low_fma(src_enc, 4);
    auto [dst0_idx, dst1_idx] = g.split_low_fma(dst_enc, 4);
    auto reg_mask = [&](V idx) { return g.spread_into_bits(g.lookup(T_REGMASK, {idx}, 2)[0], NREG); };

    Decoded d;
    int off = 0;
    auto take = [&](std::vector<Boolean>& dst, uint32_t n) { dst.assign(bits.begin() + off, bits.begin() + off + n); off += (int)n; };
    take(d.type, D.type_bits); take(d.variant, D.variant_bits); take(d.flag, D.flag_bits); take(d.src_mode, D.src_mode_bits); take(d.dst_mode, D.dst_mode_bits);
    d.src_regs[0] = reg_mask(src0_idx); d.src_regs[1] = reg_mask(src1_idx);
    d.dst_regs[0] = reg_mask(dst0_idx); d.dst_regs[1] = reg_mask(dst1_idx);
    d.imm0 = imm0; d.imm1 = imm1;
    return d;
}

RegView VmCircuit::view(const Reg& r) {  // RegisterInputView::from_input_value, src/main_vm/register_input_view.rs:27-53
    RegView v;
    for (int i = 0; i < 8; ++i) {
        auto b = g.bytes_unchecked(r.v[i]);
        for (int k = 0; k < 4; ++k) v.u8[4 * i + k] = b[k];
    }
    v.u32 = r.v;
    v.is_ptr = r.ptr;
    return v;
}

// create_prestate — src/main_vm/pre_state.rs:71-519
void VmCircuit::create_prestate(State& st, Common& common, Carry& carry) {
    Boolean should_skip_cycle = g.is_zero(st.cs.depth);  // callstack.is_empty
    Boolean pending_exception = g.B(st.pending_exception);
    Boolean execute_cycle = g.negated(should_skip_cycle);
    Boolean should_try_to_read_opcode = g.and_not(execute_cycle, pending_exception);
    st.pending_exception = g.and_not(pending_exception, pending_exception).v;  // take down the flag

    V current_pc = st.cs.ctx.pc;
    V pc_plus_one = g.u16_overflowing_add(current_pc, g.c(1)).first;
    auto [super_pc, subpc_spread] = split_pc(current_pc);
    V code_page = st.cs.ctx.code_page;
    // should_read_memory — utils.rs:107-120
    Boolean can_skip = g.b_and(g.equals(st.prev_code_page, code_page), g.equals(super_pc, st.prev_super_pc));
    Boolean should_read_opcode = g.b_and(should_try_to_read_opcode, g.negated(can_skip));

    V ts_read = st.timestamp;
    V ts_first = g.add(ts_read, g.one()), ts_second = g.add(ts_first, g.one()), ts_dst = g.add(ts_second, g.one());
    V next_cycle_timestamp = g.select(should_skip_cycle, st.timestamp, g.add(ts_dst, g.one()));

    // may_be_read_memory_for_code — utils.rs:129-233: the opcode-fetch permutation is computed in circuit
    W8 code_word = in_u32s<8>("code_word");
    {
        auto enc = memory_query_encode(ts_read, code_page, super_pc, g.zero(), g.zero(), code_word);
        S12 fin = g.compute_round_function(absorb8(enc, st.mem_tail));
        st.mem_len = g.select(should_read_opcode, g.add(st.mem_len, g.one()), st.mem_len);
        st.mem_tail = g.select_n(should_read_opcode, fin, st.mem_tail);
    }
    code_word = g.select_n(should_read_opcode, code_word, st.prev_code_word);
    auto subpc_bitmask = g.spread_into_bits(subpc_spread, 3);
    std::array<V, 2> opcode = {code_word[6], code_word[7]};
    opcode = g.select_n(subpc_bitmask[0], std::array<V, 2>{code_word[4], code_word[5]}, opcode);
    opcode = g.select_n(subpc_bitmask[1], std::array<V, 2>{code_word[2], code_word[3]}, opcode);
    opcode = g.select_n(subpc_bitmask[2], std::array<V, 2>{code_word[0], code_word[1]}, opcode);
    // mask_into_nop / mask_into_panic — utils.rs:14-42
    opcode = g.select_n(should_skip_cycle, std::array<V, 2>{g.c(D.nop_encoding & 0xffffffffu), g.c(D.nop_encoding >> 32)}, opcode);
    opcode = g.select_n(pending_exception, std::array<V, 2>{g.c(D.panic_encoding & 0xffffffffu), g.c(D.panic_encoding >> 32)}, opcode);

    st.prev_code_word = code_word;
    st.prev_code_page = code_page;
    st.cs.ctx.pc = g.select(should_skip_cycle, st.cs.ctx.pc, pc_plus_one);
    st.prev_super_pc = g.select(should_skip_cycle, st.prev_super_pc, super_pc);
    st.timestamp = next_cycle_timestamp;

    Boolean is_kernel_mode = g.B(st.cs.ctx.is_kernel), is_static_context = g.B(st.cs.ctx.is_static);
    Boolean callstack_is_full = g.equals(st.cs.depth, g.c(P(ZK_VMP_VM_MAX_STACK_DEPTH)));
    V ergs_left = st.cs.ctx.ergs;
    // encode_flags — decoded_opcode.rs:271-301
    V encoded_flags;
    {
        V r = cs.alloc_var();
        V t[4] = {st.flags.of, st.flags.eq, st.flags.gt, g.zero()};
        uint64_t k[4] = {1, 2, 4, 0};
        cs.emit_op(ZK_OP_LC4, 0, 0, t, 4, &r, 1, k, 4);
        V vars[5] = {t[0], t[1], t[2], t[3], r};
        if (cs.gate_is_allowed(ZK_GATE_REDUCTION_BY_POWERS4)) { uint64_t two = 2; cs.place_gate(ZK_GATE_REDUCTION_BY_POWERS4, vars, 5, &two, 1); }
        else cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
        encoded_flags = r;
    }
    V dirty_ergs_left;
    Decoded dec = perform_initial_decoding(opcode, encoded_flags, is_kernel_mode, is_static_context, callstack_is_full, ergs_left, should_skip_cycle,
                                           &dirty_ergs_left);
    st.cs.ctx.ergs = dirty_ergs_left;
    g.enforce_equal(type_bit(dec, ZK_VMF_INVALID).v, g.zero());  // masked: INVALID never reaches the opcodes

    // ---- source operands
    Reg draft_src0 = zero_reg(), src1_register = zero_reg();
    for (int r = 0; r < NREG; ++r) draft_src0 = select(dec.src_regs[0][r], st.regs[r], draft_src0);
    V src0_reg_lowest = g.low_u16(draft_src0.v[0]);
    for (int r = 0; r < NREG; ++r) src1_register = select(dec.src_regs[1][r], st.regs[r], src1_register);
    V current_dst0_reg_low = g.zero();
    for (int r = 0; r < NREG; ++r) current_dst0_reg_low = g.select(dec.dst_regs[0][r], st.regs[r].v[0], current_dst0_reg_low);
    V dst0_reg_lowest = g.low_u16(current_dst0_reg_low);

    V current_sp = st.cs.ctx.sp, base_page = st.cs.ctx.base_page;
    V stack_page = g.add(base_page, g.one()), heap_page = g.add(stack_page, g.one()), aux_heap_page = g.add(heap_page, g.one());
    Boolean is_nop = type_bit(dec, ZK_VMF_NOP), not_nop = g.negated(is_nop);

    // resolve_memory_region_and_index_for_source — utils.rs:237-305
    MemLoc loc_src0;
    V new_sp_after_src0;
    Boolean should_read_memory_for_src0;
    {
        Boolean use_code = dec.src_mode[ZK_VMM_CODE_PAGE], use_abs = dec.src_mode[ZK_VMM_ABSOLUTE_STACK],
                use_rel = dec.src_mode[ZK_VMM_STACK_OFFSET], use_pp = dec.src_mode[ZK_VMM_STACK_PUSH_POP];
        Boolean absolute_mode = g.b_or(use_code, use_abs);
        V index_for_absolute = g.u16_overflowing_add(src0_reg_lowest, dec.imm0).first;
        V index_for_relative = g.u16_overflowing_sub(current_sp, index_for_absolute).first;
        Boolean use_stack = g.multi_or({use_abs, use_rel, use_pp});
        Boolean did_read = g.b_and(g.b_or(use_stack, use_code), not_nop);
        loc_src0.page = g.select(use_stack, stack_page, code_page);
        loc_src0.index = g.select(absolute_mode, index_for_absolute, index_for_relative);
        new_sp_after_src0 = g.select(use_pp, index_for_relative, current_sp);
        should_read_memory_for_src0 = did_read;
    }
    // resolve_memory_region_and_index_for_dest — utils.rs:307-384
    MemLoc loc_dst0;
    V new_sp;
    Boolean should_write_memory_for_dst0;
    {
        Boolean use_abs = dec.dst_mode[ZK_VMM_ABSOLUTE_STACK], use_rel = dec.dst_mode[ZK_VMM_STACK_OFFSET], use_pp = dec.dst_mode[ZK_VMM_STACK_PUSH_POP];
        V index_for_absolute = g.u16_overflowing_add(dst0_reg_lowest, dec.imm1).first;
        V index_for_relative_with_push = g.u16_overflowing_add(new_sp_after_src0, index_for_absolute).first;
        V index_for_relative = g.u16_overflowing_sub(new_sp_after_src0, index_for_absolute).first;
        Boolean did_write = g.b_and(g.multi_or({use_abs, use_rel, use_pp}), not_nop);
        V somewhat_relative = g.select(use_pp, new_sp_after_src0, index_for_relative);
        loc_dst0.page = stack_page;
        loc_dst0.index = g.select(use_abs, index_for_absolute, somewhat_relative);
        new_sp = g.select(use_pp, index_for_relative_with_push, new_sp_after_src0);
        should_write_memory_for_dst0 = did_write;
    }
    st.cs.ctx.sp = new_sp;

    // may_be_read_memory_for_source_operand — utils.rs:388-522 (witness-only permutation, enforced among the 8 sponges)
    Reg src0_from_mem;
    Sponge src0_sponge;
    {
        src0_from_mem.v = in_u32s<8>("src0_read_value");
        src0_from_mem.ptr = in_bool("src0_read_is_ptr");
        auto enc = memory_query_encode(ts_read, loc_src0.page, loc_src0.index, g.zero(), src0_from_mem.ptr, src0_from_mem.v);
        src0_sponge.init = absorb8(enc, st.mem_tail);
        S12 simulated = simulate(src0_sponge.init, should_read_memory_for_src0);
        src0_sponge.fin = g.select_n(should_read_memory_for_src0, simulated, st.mem_tail);
        src0_sponge.flag = should_read_memory_for_src0;
        st.mem_len = g.select(should_read_memory_for_src0, g.add(st.mem_len, g.one()), st.mem_len);
        st.mem_tail = src0_sponge.fin;
    }
    Reg src0 = select(dec.src_mode[ZK_VMM_REG_ONLY], draft_src0, src0_from_mem);
    Reg imm_as_reg = zero_reg();
    imm_as_reg.v[0] = dec.imm0;
    src0 = select(dec.src_mode[ZK_VMM_IMM16], imm_as_reg, src0);

    Boolean swap_operands;
    {
        Boolean is_assymmetric = g.multi_or({type_bit(dec, ZK_VMF_SUB), type_bit(dec, ZK_VMF_DIV), type_bit(dec, ZK_VMF_SHIFT)});
        Boolean t0 = g.b_and(is_assymmetric, flag_bit(dec, ZK_VMFL_SWAP_ARITH));
        Boolean t1 = g.b_and(type_bit(dec, ZK_VMF_PTR), flag_bit(dec, ZK_VMFL_SWAP_PTR));
        swap_operands = g.b_or(t0, t1);
    }
    Reg sel_src0 = src0, sel_src1 = src1_register;
    src0 = select(swap_operands, sel_src1, sel_src0);
    Reg src1 = select(swap_operands, sel_src0, sel_src1);
    Boolean not_kernel_mode = g.negated(is_kernel_mode);
    {
        Boolean keeps_ptr = g.multi_or({type_bit(dec, ZK_VMF_RET), type_bit(dec, ZK_VMF_PTR), type_bit(dec, ZK_VMF_UMA), type_bit(dec, ZK_VMF_FAR_CALL)});
        Boolean erase0 = g.multi_and({g.B(src0.ptr), g.negated(keeps_ptr), not_kernel_mode});
        Boolean erase1 = g.b_and(g.B(src1.ptr), not_kernel_mode);
        // conditionally_erase_fat_pointer_data — src/base_structures/register/mod.rs:74-84
        src0.ptr = g.mask_negated(src0.ptr, erase0); src0.v[1] = g.mask_negated(src0.v[1], erase0); src0.v[2] = g.mask_negated(src0.v[2], erase0);
        src1.ptr = g.mask_negated(src1.ptr, erase1); src1.v[1] = g.mask_negated(src1.v[1], erase1); src1.v[2] = g.mask_negated(src1.v[2], erase1);
    }
    common.reseted_flags = {g.zero(), g.zero(), g.zero()};
    common.current_flags = st.flags;
    common.dec = dec;
    common.src0 = src0; common.src1 = src1;
    common.src0_view = view(src0); common.src1_view = view(src1);
    common.ts_read = ts_read; common.ts_first = ts_first; common.ts_second = ts_second; common.ts_dst = ts_dst;
    carry.did_skip_cycle = should_skip_cycle;
    carry.next_pc = pc_plus_one;
    carry.src0_read_sponge = src0_sponge;
    carry.dst0_location = loc_dst0;
    carry.dst0_performs_memory_access = should_write_memory_for_dst0;
    carry.preliminary_ergs_left = dirty_ergs_left;
    carry.heap_page = heap_page; carry.aux_heap_page = aux_heap_page;
}

// apply_add_sub — src/main_vm/opcodes/add_sub.rs:8-166
void VmCircuit::apply_add_sub(const State&, const Common& cm, const Carry&, Diffs& df) {
    auto [add_res, of] = g.u256_add_witness(cm.src0_view.u32, cm.src1_view.u32);
    auto [sub_res, uf] = g.u256_sub_witness(cm.src0_view.u32, cm.src1_view.u32);
    Boolean apply_add = type_bit(cm.dec, ZK_VMF_ADD), apply_sub = type_bit(cm.dec, ZK_VMF_SUB);
    W8 result = g.select_n(apply_add, add_res, sub_res);
    AddSubRelation rel;
    rel.a = cm.src1_view.u32;
    rel.b = g.select_n(apply_add, cm.src0_view.u32, sub_res);
    rel.c = g.select_n(apply_add, add_res, cm.src0_view.u32);
    rel.of = g.select(apply_add, of, uf);
    Boolean result_is_zero = g.all_zero(result);
    Boolean gt = g.negated(g.b_or(g.B(rel.of), result_is_zero));
    Boolean apply_any = g.b_or(apply_add, apply_sub);
    Reg dst0 = zero_reg();
    dst0.v = result;
    Boolean update_flags = g.b_and(apply_any, flag_bit(cm.dec, ZK_VMFL_SET_FLAGS));
    df.dst_0_values.push_back({D.can_write_dst0_into_memory[ZK_VMF_ADD] != 0, apply_any, dst0});
    df.flags.push_back({update_flags, FlagsPort{rel.of, result_is_zero.v, gt.v}});
    df.u32_conditional_range_checks.push_back({apply_any, result});
    df.add_sub_relations.push_back({apply_any, {rel}});
}

// apply_jump — src/main_vm/opcodes/jump.rs:3-38
void VmCircuit::apply_jump(const State&, const Common& cm, const Carry&, Diffs& df) {
    V jump_dst = g.from_le_bytes2(cm.src0_view.u8[0], cm.src0_view.u8[1]);
    df.new_pc_candidates.push_back({type_bit(cm.dec, ZK_VMF_JUMP), jump_dst});
}

// apply_binop + get_binop_subresults — src/main_vm/opcodes/binop.rs:14-244
void VmCircuit::apply_binop(const State&, const Common& cm, const Carry&, Diffs& df) {
    Boolean should_apply = type_bit(cm.dec, ZK_VMF_BINOP);
    Boolean is_and = variant_bit(cm.dec, ZK_VMV_BINOP_AND), is_or = variant_bit(cm.dec, ZK_VMV_BINOP_OR);
    std::array<V, 32> and_r, or_r, xor_r;
    std::array<V, 32> composite;
    for (int i = 0; i < 32; ++i) composite[i] = g.lookup(T_BINOP, {cm.src0_view.u8[i], cm.src1_view.u8[i]}, 1)[0];
    std::array<V, 96> all;
    for (int i = 0; i < 32; ++i) {
        V chunk[3];
        V first = cs.alloc_vars(3);
        for (int k = 0; k < 3; ++k) chunk[k] = first + k;
        cs.emit_op(ZK_OP_SPLIT, 3, 16, &composite[i], 1, chunk, 3, nullptr, 0);
        for (int k = 0; k < 3; ++k) all[3 * i + k] = chunk[k];
    }
    for (int i = 0; i < 96; i += 2) (void)g.lookup(T_BINOP, {all[i], all[i + 1]}, 1);  // the table as a range check of every chunk
    for (int i = 0; i < 32; ++i) {
        V vars[5] = {all[3 * i], all[3 * i + 1], all[3 * i + 2], g.zero(), composite[i]};
        if (cs.gate_is_allowed(ZK_GATE_REDUCTION4)) { uint64_t k[4] = {1, 1ull << 16, 1ull << 32, 0}; cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4); }
        else { uint64_t k = 1ull << 16; cs.place_gate(ZK_GATE_REDUCTION_BY_POWERS4, vars, 5, &k, 1); }
        and_r[i] = all[3 * i]; or_r[i] = all[3 * i + 1]; xor_r[i] = all[3 * i + 2];
    }
    W8 and_c, or_c, xor_c;
    for (int i = 0; i < 8; ++i) {
        and_c[i] = g.from_le_bytes4(and_r[4 * i], and_r[4 * i + 1], and_r[4 * i + 2], and_r[4 * i + 3]);
        or_c[i] = g.from_le_bytes4(or_r[4 * i], or_r[4 * i + 1], or_r[4 * i + 2], or_r[4 * i + 3]);
        xor_c[i] = g.from_le_bytes4(xor_r[4 * i], xor_r[4 * i + 1], xor_r[4 * i + 2], xor_r[4 * i + 3]);
    }
    W8 result = g.select_n(is_and, and_c, xor_c);
    result = g.select_n(is_or, or_c, result);
    Boolean result_is_zero = g.all_zero(result);
    Reg dst0 = zero_reg();
    dst0.v = result;
    df.dst_0_values.push_back({D.can_write_dst0_into_memory[ZK_VMF_BINOP] != 0, should_apply, dst0});
    df.flags.push_back({g.b_and(should_apply, flag_bit(cm.dec, ZK_VMFL_SET_FLAGS)), FlagsPort{g.zero(), result_is_zero.v, g.zero()}});
}

// apply_context — src/main_vm/opcodes/context.rs:7-307
void VmCircuit::apply_context(const State& st, const Common& cm, const Carry& cr, Diffs& df) {
    Boolean should_apply = type_bit(cm.dec, ZK_VMF_CONTEXT);
    Boolean is_this = variant_bit(cm.dec, ZK_VMV_CTX_THIS), is_caller = variant_bit(cm.dec, ZK_VMV_CTX_CALLER),
            is_code_address = variant_bit(cm.dec, ZK_VMV_CTX_CODE_ADDRESS), is_meta = variant_bit(cm.dec, ZK_VMV_CTX_META),
            is_ergs_left = variant_bit(cm.dec, ZK_VMV_CTX_ERGS_LEFT), is_get_u128 = variant_bit(cm.dec, ZK_VMV_CTX_GET_CONTEXT_U128),
            is_set_u128 = variant_bit(cm.dec, ZK_VMV_CTX_SET_CONTEXT_U128), is_set_pubdata = variant_bit(cm.dec, ZK_VMV_CTX_SET_ERGS_PER_PUBDATA),
            is_inc_tx = variant_bit(cm.dec, ZK_VMV_CTX_INC_TX_NUMBER);
    Boolean write_to_context = g.b_and(should_apply, is_set_u128), set_pubdata_ergs = g.b_and(should_apply, is_set_pubdata),
            increment_tx_counter = g.b_and(should_apply, is_inc_tx);
    Boolean write_like = g.negated(g.multi_or({is_set_u128, is_set_pubdata, is_inc_tx}));
    Boolean write_to_dst0 = g.b_and(should_apply, write_like);
    V incremented_tx_number = g.u32_overflowing_add(st.tx_number, g.c(1)).first;
    const Ctx& c = st.cs.ctx;
    V zero = g.zero();
    V meta_highest = g.from_le_bytes4(c.this_shard, c.caller_shard, c.code_shard, zero);
    W8 meta = {st.ergs_per_pubdata, zero, c.heap_bound, c.aux_heap_bound, zero, zero, zero, meta_highest};
    V low_u32 = g.select(is_ergs_left, cr.preliminary_ergs_left, c.sp);
    S4 r128 = {low_u32, zero, zero, zero};
    r128 = g.select_n(is_get_u128, c.ctx_u128, r128);
    A5 r160 = {r128[0], r128[1], r128[2], r128[3], zero};
    r160 = g.select_n(is_this, c.this_, r160);
    r160 = g.select_n(is_caller, c.caller, r160);
    r160 = g.select_n(is_code_address, c.code_address, r160);
    W8 r256 = {r160[0], r160[1], r160[2], r160[3], r160[4], zero, zero, zero};
    r256 = g.select_n(is_meta, meta, r256);
    Reg dst0 = zero_reg();
    dst0.v = r256;
    df.dst_0_values.push_back({D.can_write_dst0_into_memory[ZK_VMF_CONTEXT] != 0, write_to_dst0, dst0});
    df.context_u128_candidates.push_back({write_to_context, S4{cm.src0_view.u32[0], cm.src0_view.u32[1], cm.src0_view.u32[2], cm.src0_view.u32[3]}});
    df.new_tx_number.push_back({increment_tx_counter, incremented_tx_number});
    df.new_ergs_per_pubdata.push_back({set_pubdata_ergs, cm.src0_view.u32[0]});
}

// apply_ptr — src/main_vm/opcodes/ptr.rs:6-183
void VmCircuit::apply_ptr(const State&, const Common& cm, const Carry&, Diffs& df) {
    Boolean should_apply = type_bit(cm.dec, ZK_VMF_PTR);
    Boolean v_add = variant_bit(cm.dec, ZK_VMV_PTR_ADD), v_sub = variant_bit(cm.dec, ZK_VMV_PTR_SUB), v_pack = variant_bit(cm.dec, ZK_VMV_PTR_PACK),
            v_shrink = variant_bit(cm.dec, ZK_VMV_PTR_SHRINK);
    const RegView &s0 = cm.src0_view, &s1 = cm.src1_view;
    Boolean args_types_are_invalid = g.negated(g.b_and(g.B(s0.is_ptr), g.negated(g.B(s1.is_ptr))));
    std::vector<Boolean> lz;
    for (int i = 0; i < 8; ++i) lz.push_back(g.is_zero(s1.u32[i]));
    Boolean src1_32_to_256_is_zero = g.multi_and(std::vector<Boolean>(lz.begin() + 1, lz.end()));
    Boolean src1_0_to_128_is_zero = g.multi_and(std::vector<Boolean>(lz.begin(), lz.begin() + 4));
    Boolean too_large_offset = g.b_and(g.negated(src1_32_to_256_is_zero), g.b_or(v_add, v_sub));
    Boolean dirty_value_for_pack = g.b_and(g.negated(src1_0_to_128_is_zero), v_pack);
    auto [res_add, of] = g.u32_overflowing_add(s0.u32[0], s1.u32[0]);
    Boolean overflow_panic_if_add = g.b_and(v_add, of);
    auto [res_sub, uf] = g.u32_overflowing_sub(s0.u32[0], s1.u32[0]);
    Boolean underflow_panic_if_sub = g.b_and(v_sub, uf);
    auto [res_shrink, uf2] = g.u32_overflowing_sub(s0.u32[3], s1.u32[0]);
    Boolean underflow_panic_if_shrink = g.b_and(v_shrink, uf2);
    Boolean any_potential_panic = g.multi_or({args_types_are_invalid, too_large_offset, dirty_value_for_pack, overflow_panic_if_add,
                                              underflow_panic_if_sub, underflow_panic_if_shrink});
    Boolean should_panic = g.b_and(should_apply, any_potential_panic);
    Boolean should_update_register = g.b_and(should_apply, g.negated(any_potential_panic));
    V low = g.select(v_add, res_add, s0.u32[0]);
    low = g.select(v_sub, res_sub, low);
    V bits_96_128_if_shrink = g.select(v_shrink, res_shrink, s0.u32[3]);
    S4 highest_128 = g.select_n(v_pack, S4{s1.u32[4], s1.u32[5], s1.u32[6], s1.u32[7]}, S4{s0.u32[4], s0.u32[5], s0.u32[6], s0.u32[7]});
    V lowest32 = g.select(v_pack, s0.u32[0], low);
    V bits_96_128 = g.select(v_pack, s0.u32[3], bits_96_128_if_shrink);
    Reg dst0;
    dst0.ptr = s0.is_ptr;
    dst0.v = {lowest32, s0.u32[1], s0.u32[2], bits_96_128, highest_128[0], highest_128[1], highest_128[2], highest_128[3]};
    df.dst_0_values.push_back({D.can_write_dst0_into_memory[ZK_VMF_PTR] != 0, should_update_register, dst0});
    df.pending_exceptions.push_back(should_panic);
}

// apply_log + construct_hash_relations_for_log_and_new_queue_states — src/main_vm/opcodes/log.rs:16-671
void VmCircuit::apply_log(const State& st, const Common& cm, const Carry& cr, Diffs& df) {
    Boolean should_apply = type_bit(cm.dec, ZK_VMF_LOG);
    Boolean is_storage_read = variant_bit(cm.dec, ZK_VMV_LOG_STORAGE_READ), is_storage_write = variant_bit(cm.dec, ZK_VMV_LOG_STORAGE_WRITE),
            is_event = variant_bit(cm.dec, ZK_VMV_LOG_EVENT), is_l1_message = variant_bit(cm.dec, ZK_VMV_LOG_TO_L1),
            is_precompile = variant_bit(cm.dec, ZK_VMV_LOG_PRECOMPILE_CALL);
    const Ctx& c = st.cs.ctx;
    W8 key = cm.src0_view.u32, written_value = cm.src1_view.u32;
    Boolean should_swap_read_page = g.b_and(g.is_zero(key[4]), is_precompile), should_swap_write_page = g.b_and(g.is_zero(key[5]), is_precompile);
    key[4] = g.select(should_swap_read_page, cr.heap_page, key[4]);
    key[5] = g.select(should_swap_write_page, cr.heap_page, key[5]);
    Boolean is_rollup = g.is_zero(c.this_shard);
    Boolean write_to_rollup = g.b_and(is_rollup, is_storage_write);
    V ergs_to_burn_for_l1_message = g.u32_non_widening_mul(st.ergs_per_pubdata, g.c(P(ZK_VMP_L1_MESSAGE_PUBDATA_BYTES)));
    V ergs_to_burn_for_precompile_call = cm.src1_view.u32[0];
    Boolean is_storage_access = g.b_or(is_storage_read, is_storage_write);
    Boolean is_revertable = g.negated(g.b_or(is_storage_read, is_precompile));
    V aux_byte = g.linear_combination({{is_storage_access.v, P(ZK_VMP_STORAGE_AUX_BYTE)}, {is_event.v, P(ZK_VMP_EVENT_AUX_BYTE)},
                                       {is_l1_message.v, P(ZK_VMP_L1_MESSAGE_AUX_BYTE)}, {is_precompile.v, P(ZK_VMP_PRECOMPILE_AUX_BYTE)}});
    LogQuery log;
    for (int i = 0; i < 5; ++i) log.address[i] = UInt32{c.this_[i]};
    for (int i = 0; i < 8; ++i) { log.key.inner[i] = UInt32{key[i]}; log.read_value.inner[i] = UInt32{g.zero()}; log.written_value.inner[i] = UInt32{written_value[i]}; }
    log.rw_flag = is_revertable; log.aux_byte = UInt8{aux_byte}; log.rollback = g.bool_const(false);
    log.is_service = flag_bit(cm.dec, ZK_VMFL_FIRST_MESSAGE);
    log.shard_id = UInt8{c.this_shard}; log.tx_number_in_block = UInt32{st.tx_number}; log.timestamp = UInt32{cm.ts_first};

    V pubdata_refund = in_u32("log_pubdata_refund");  // oracle.get_refunds
    V net_cost = g.u32_sub_no_overflow(g.c(P(ZK_VMP_INITIAL_STORAGE_WRITE_PUBDATA_BYTES)), pubdata_refund);
    V ergs_to_burn_for_rollup_storage_write = g.u32_non_widening_mul(st.ergs_per_pubdata, net_cost);
    V ergs_to_burn = g.select(write_to_rollup, ergs_to_burn_for_rollup_storage_write, g.zero());
    ergs_to_burn = g.select(is_precompile, ergs_to_burn_for_precompile_call, ergs_to_burn);
    ergs_to_burn = g.select(is_l1_message, ergs_to_burn_for_l1_message, ergs_to_burn);
    auto [ergs_rem, not_enough_ergs] = g.u32_overflowing_sub(cr.preliminary_ergs_left, ergs_to_burn);
    V ergs_remaining = g.mask_negated(ergs_rem, not_enough_ergs);
    Boolean have_enough_ergs = g.negated(not_enough_ergs);
    Boolean execute_either = g.b_and(should_apply, have_enough_ergs);

    W8 read_value_w = in_u32s<8>("log_storage_read_value");  // oracle.get_storage_read_witness
    W8 zero8;
    for (auto& x : zero8) x = g.zero();
    W8 read_value = g.select_n(is_storage_access, read_value_w, zero8);
    for (int i = 0; i < 8; ++i) log.read_value.inner[i] = UInt32{read_value[i]};
    for (int i = 0; i < 8; ++i) log.written_value.inner[i] = UInt32{g.select(log.rw_flag, log.written_value.inner[i].v, read_value[i])};
    auto packed_forward = encode_log(log);
    auto packed_rollback = packed_forward;
    packed_rollback[19] = g.one();  // update_packing_for_rollback (ROLLBACK_PACKING_FLAG_VARIABLE_IDX)
    Boolean execute_rollback = g.b_and(execute_either, is_revertable);
    S4 prev_revert_head = in_nums<4>("log_rollback_queue_prev_head");  // oracle.get_rollback_queue_witness

    // construct_hash_relations_for_log_and_new_queue_states
    S12 empty = g.empty_state();
    auto chunk = [&](const std::array<V, 20>& enc, int k) { std::array<V, 8> e; for (int i = 0; i < 8; ++i) e[i] = enc[8 * k + i]; return e; };
    S12 r0_init = absorb8(chunk(packed_forward, 0), empty), r0_fin = simulate(r0_init, execute_either);
    S12 r1_init = absorb8(chunk(packed_forward, 1), r0_fin), r1_fin = simulate(r1_init, execute_either);
    std::array<V, 8> e2f, e2r;
    for (int i = 0; i < 4; ++i) { e2f[i] = packed_forward[16 + i]; e2f[4 + i] = st.cs.fwd_tail[i]; e2r[i] = packed_rollback[16 + i]; e2r[4 + i] = prev_revert_head[i]; }
    S12 r2f_init = absorb8(e2f, r1_fin), r2f_fin = simulate(r2f_init, execute_either);
    S12 r2r_init = absorb8(e2r, r1_fin), r2r_fin = simulate(r2r_init, execute_rollback);
    S4 new_fwd_cand = {r2f_fin[0], r2f_fin[1], r2f_fin[2], r2f_fin[3]}, sim_rollback_head = {r2r_fin[0], r2r_fin[1], r2r_fin[2], r2r_fin[3]};
    S4 new_forward_tail = g.select_n(execute_either, new_fwd_cand, st.cs.fwd_tail);
    S4 new_rollback_head = g.select_n(execute_rollback, prev_revert_head, c.rq_head);
    for (int i = 0; i < 4; ++i) g.cond_enforce_equal(execute_rollback, sim_rollback_head[i], c.rq_head[i]);
    std::vector<Sponge> relations = {{execute_either, r0_init, r0_fin}, {execute_either, r1_init, r1_fin}, {execute_either, r2f_init, r2f_fin},
                                     {execute_rollback, r2r_init, r2r_fin}};

    W8 precompile_call_result = zero8;
    precompile_call_result[0] = have_enough_ergs.v;
    Reg dst0 = zero_reg();
    dst0.v = g.select_n(is_storage_read, read_value, precompile_call_result);
    V new_fwd_len = g.select(execute_either, g.add(st.cs.fwd_len, g.one()), st.cs.fwd_len);
    V new_revert_len = g.select(execute_rollback, g.add(c.rq_len, g.one()), c.rq_len);
    Boolean should_update_dst0 = g.b_and(g.b_or(is_storage_read, is_precompile), should_apply);
    df.dst_0_values.push_back({D.can_write_dst0_into_memory[ZK_VMF_LOG] != 0, should_update_dst0, dst0});
    df.log_queue_forward_candidates.push_back({should_apply, new_fwd_len, new_forward_tail});
    df.log_queue_rollback_candidates.push_back({should_apply, new_revert_len, new_rollback_head});
    df.new_ergs_left_candidates.push_back({should_apply, ergs_remaining});
    df.sponge_candidates_to_run.push_back({should_apply, relations});
}

// apply_mul_div — src/main_vm/opcodes/mul_div.rs:199-417
void VmCircuit::apply_mul_div(const State&, const Common& cm, const Carry&, Diffs& df) {
    Boolean apply_mul = type_bit(cm.dec, ZK_VMF_MUL), apply_div = type_bit(cm.dec, ZK_VMF_DIV);
    const W8 &s0 = cm.src0_view.u32, &s1 = cm.src1_view.u32;
    auto [mul_low, mul_high] = g.u256_wide_witness(ZK_OP_U256_MULWIDE, s0, s1);
    auto [quotient, remainder] = g.u256_wide_witness(ZK_OP_U256_DIVREM, s0, s1);
    W8 result_0 = g.select_n(apply_mul, mul_low, quotient), result_1 = g.select_n(apply_mul, mul_high, remainder);
    W8 zero8;
    for (auto& x : zero8) x = g.zero();
    MulDivRelation rel;
    rel.rem = g.select_n(apply_mul, zero8, remainder);
    rel.a = g.select_n(apply_mul, s0, quotient);
    rel.b = s1;
    rel.mul_low = g.select_n(apply_mul, mul_low, s0);
    rel.mul_high = g.select_n(apply_mul, mul_high, zero8);
    Boolean high_is_zero = g.all_zero(mul_high), low_is_zero = g.all_zero(mul_low);
    Boolean of_mul = g.negated(high_is_zero), eq_mul = low_is_zero, gt_mul = g.b_and(g.negated(of_mul), g.negated(eq_mul));
    Boolean divisor_is_zero = g.all_zero(s1), divisor_is_non_zero = g.negated(divisor_is_zero);
    Boolean quotient_is_zero = g.all_zero(quotient), remainder_is_zero = g.all_zero(remainder);
    auto [sub_res, rem_lt_div] = g.u256_sub_witness(remainder, s1);
    AddSubRelation arel{s1, sub_res, remainder, rem_lt_div};
    g.conditionally_enforce_true(g.B(rem_lt_div), divisor_is_non_zero);
    g.conditionally_enforce_true(quotient_is_zero, divisor_is_zero);
    Boolean mask_remainder_into_zero = g.b_and(apply_div, divisor_is_zero);
    for (auto& x : result_1) x = g.mask_negated(x, mask_remainder_into_zero);
    Boolean of_div = divisor_is_zero, eq_div = g.b_and(g.negated(divisor_is_zero), quotient_is_zero),
            gt_div = g.b_and(g.negated(divisor_is_zero), remainder_is_zero);
    FlagsPort fl{g.select(apply_mul, of_mul.v, of_div.v), g.select(apply_mul, eq_mul.v, eq_div.v), g.select(apply_mul, gt_mul.v, gt_div.v)};
    Boolean apply_any = g.b_or(apply_mul, apply_div);
    Reg dst0 = zero_reg(), dst1 = zero_reg();
    dst0.v = result_0; dst1.v = result_1;
    df.dst_0_values.push_back({D.can_write_dst0_into_memory[ZK_VMF_MUL] != 0, apply_any, dst0});
    df.dst_1_values.push_back({apply_any, dst1});
    df.flags.push_back({g.b_and(apply_any, flag_bit(cm.dec, ZK_VMFL_SET_FLAGS)), fl});
    df.u32_conditional_range_checks.push_back({apply_any, sub_res});
    df.add_sub_relations.push_back({apply_any, {arel}});
    df.mul_div_relations.push_back({apply_any, {rel}});
}

// apply_shifts + get_shift_constant — src/main_vm/opcodes/shifts.rs:8-221
void VmCircuit::apply_shifts(const State&, const Common& cm, const Carry&, Diffs& df) {
    Boolean should_apply = type_bit(cm.dec, ZK_VMF_SHIFT);
    Boolean is_rol = variant_bit(cm.dec, ZK_VMV_SHIFT_ROL), is_ror = variant_bit(cm.dec, ZK_VMV_SHIFT_ROR), is_shr = variant_bit(cm.dec, ZK_VMV_SHIFT_SHR);
    Boolean is_cyclic = g.b_or(is_rol, is_ror), is_right = g.b_or(is_ror, is_shr);
    const W8& reg = cm.src0_view.u32;
    V shift = cm.src1_view.u8[0];
    Boolean shift_is_zero = g.is_zero(shift);
    V inverted_shift = g.sub(g.c(256), shift);
    Boolean change_flag = g.b_and(is_ror, g.negated(shift_is_zero));
    V full_shift = g.select(change_flag, inverted_shift, shift);
    W8 full_shift_limbs;
    for (int idx = 0; idx < 4; ++idx) {
        V key = g.add(full_shift, g.c((uint64_t)idx << 8));
        auto ab = g.lookup(T_BITSHIFT, {key}, 2);
        full_shift_limbs[2 * idx] = ab[0]; full_shift_limbs[2 * idx + 1] = ab[1];
    }
    Boolean is_right_shift = g.b_and(is_right, g.negated(is_cyclic));
    auto [rshift_q, rshift_r] = g.u256_wide_witness(ZK_OP_U256_DIVREM, reg, full_shift_limbs);
    Boolean apply_left_shift = g.b_and(should_apply, g.negated(is_right_shift));
    auto [lshift_low, lshift_high] = g.u256_wide_witness(ZK_OP_U256_MULWIDE, reg, full_shift_limbs);
    W8 zero8;
    for (auto& x : zero8) x = g.zero();
    MulDivRelation rel;
    rel.rem = g.select_n(apply_left_shift, zero8, rshift_r);
    rel.a = g.select_n(apply_left_shift, reg, rshift_q);
    rel.b = full_shift_limbs;
    rel.mul_low = g.select_n(apply_left_shift, lshift_low, reg);
    rel.mul_high = g.select_n(apply_left_shift, lshift_high, zero8);
    auto [sub_res, rem_lt_div] = g.u256_sub_witness(rshift_r, full_shift_limbs);
    g.conditionally_enforce_true(g.B(rem_lt_div), is_right_shift);
    AddSubRelation arel{full_shift_limbs, sub_res, rshift_r, rem_lt_div};
    W8 temp = g.select_n(is_right_shift, rshift_q, lshift_low), final_result;
    for (int i = 0; i < 8; ++i) final_result[i] = g.fma(1, lshift_high[i], is_cyclic.v, 1, temp[i]);  // of * is_cyclic + limb
    Boolean res_is_zero = g.all_zero(final_result);
    Reg dst0 = zero_reg();
    dst0.v = final_result;
    df.dst_0_values.push_back({D.can_write_dst0_into_memory[ZK_VMF_SHIFT] != 0, should_apply, dst0});
    df.flags.push_back({g.b_and(should_apply, flag_bit(cm.dec, ZK_VMFL_SET_FLAGS)), FlagsPort{g.zero(), res_is_zero.v, g.zero()}});
    df.u32_conditional_range_checks.push_back({should_apply, sub_res});
    df.add_sub_relations.push_back({should_apply, {arel}});
    df.mul_div_relations.push_back({should_apply, {rel}});
}

// apply_uma + QuasiFatPtrInUMA::parse_and_validate — src/main_vm/opcodes/uma.rs:18-1103
void VmCircuit::apply_uma(const State& st, const Common& cm, const Carry& cr, Diffs& df) {
    Boolean should_apply = type_bit(cm.dec, ZK_VMF_UMA);
    Boolean is_heap_read = variant_bit(cm.dec, ZK_VMV_UMA_HEAP_READ), is_heap_write = variant_bit(cm.dec, ZK_VMV_UMA_HEAP_WRITE),
            is_aux_read = variant_bit(cm.dec, ZK_VMV_UMA_AUX_HEAP_READ), is_aux_write = variant_bit(cm.dec, ZK_VMV_UMA_AUX_HEAP_WRITE),
            is_fat_ptr_read = variant_bit(cm.dec, ZK_VMV_UMA_FAT_PTR_READ);
    Boolean increment_offset = flag_bit(cm.dec, ZK_VMFL_UMA_INCREMENT);
    Boolean access_heap = g.b_or(is_heap_read, is_heap_write), access_aux_heap = g.b_or(is_aux_read, is_aux_write);
    const RegView& s0 = cm.src0_view;
    Boolean not_a_ptr_when_expected = g.multi_and({should_apply, is_fat_ptr_read, g.negated(g.B(s0.is_ptr))});
    // ---- QuasiFatPtrInUMA::parse_and_validate (uma.rs:1003-1086)
    V offset = s0.u32[0], page = s0.u32[1], start = s0.u32[2], length = s0.u32[3];
    Boolean offset_is_strictly_in_slice = g.u32_overflowing_sub(offset, length).second;
    Boolean skip_if_legitimate_fat_ptr = g.b_and(g.negated(offset_is_strictly_in_slice), is_fat_ptr_read);
    V formal_start = g.mask(start, is_fat_ptr_read);
    V absolute_address = g.u32_overflowing_add(formal_start, offset).first;
    auto [incremented_offset, is_non_addressable0] = g.u32_overflowing_add(offset, g.c(32));
    Boolean is_non_addressable = g.b_or(is_non_addressable0, g.equals(incremented_offset, g.c(0xffffffffu)));
    Boolean q_should_set_panic = g.b_or(not_a_ptr_when_expected, is_non_addressable);
    Boolean q_skip_memory_access = g.multi_or({not_a_ptr_when_expected, skip_if_legitimate_fat_ptr, is_non_addressable});
    auto [bytes_oob0, uf_oob] = g.u32_overflowing_sub(incremented_offset, length);
    V bytes_oob = g.mask_negated(g.mask_negated(bytes_oob0, q_skip_memory_access), uf_oob);
    V bytes_to_cleanup_out_of_bounds = g.u32_div_by_constant(bytes_oob, 32).second;
    // ---- growth
    V max_accessed = incremented_offset;
    V heap_bound = st.cs.ctx.heap_bound, aux_heap_bound = st.cs.ctx.aux_heap_bound;
    V heap_max_accessed = g.mask(max_accessed, access_heap);
    auto [heap_growth0, uf_h] = g.u32_overflowing_sub(heap_max_accessed, heap_bound);
    V heap_growth = g.mask_negated(heap_growth0, uf_h);
    V new_heap_upper_bound = g.select(uf_h, heap_bound, heap_max_accessed);
    Boolean grow_heap = g.b_and(access_heap, should_apply);
    V aux_max_accessed = g.mask(max_accessed, access_aux_heap);
    auto [aux_growth0, uf_a] = g.u32_overflowing_sub(aux_max_accessed, aux_heap_bound);
    V aux_growth = g.mask_negated(aux_growth0, uf_a);
    V new_aux_heap_upper_bound = g.select(uf_a, aux_heap_bound, aux_max_accessed);
    Boolean grow_aux_heap = g.b_and(access_aux_heap, should_apply);
    V growth_cost = g.mask(heap_growth, access_heap);
    growth_cost = g.select(access_aux_heap, aux_growth, growth_cost);
    std::vector<Boolean> top_zero;
    for (int i = 1; i < 8; ++i) top_zero.push_back(g.is_zero(s0.u32[i]));
    Boolean top_bits_are_non_zero = g.negated(g.multi_and(top_zero));
    Boolean heap_access_like = g.b_or(access_heap, access_aux_heap);
    Boolean exception_heap_deref_out_of_bounds = g.b_and(heap_access_like, g.b_or(top_bits_are_non_zero, is_non_addressable));
    growth_cost = g.select(exception_heap_deref_out_of_bounds, g.c(0xffffffffu), growth_cost);
    auto [ergs_after, uf_e] = g.u32_overflowing_sub(cr.preliminary_ergs_left, growth_cost);
    Boolean set_panic = g.multi_or({q_should_set_panic, uf_e, exception_heap_deref_out_of_bounds});
    V ergs_left_after_growth = g.mask_negated(ergs_after, uf_e);
    Boolean should_skip_memory_ops = g.b_or(q_skip_memory_access, set_panic);
    Boolean is_read_access = g.multi_or({is_heap_read, is_aux_read, is_fat_ptr_read}), is_write_access = g.b_or(is_heap_write, is_aux_write);

    auto [cell_idx, unalignment] = g.u32_div_by_constant(absolute_address, 32);
    Boolean access_is_unaligned = g.negated(g.is_zero(unalignment));
    V mem_page = g.select(access_heap, cr.heap_page, page);
    mem_page = g.select(access_aux_heap, cr.aux_heap_page, mem_page);
    V a_cell_idx = cell_idx, b_cell_idx = g.u32_overflowing_add(a_cell_idx, g.c(1)).first;
    Boolean do_not_skip_memory_access = g.negated(should_skip_memory_ops);
    Boolean is_unaligned_read = g.multi_and({should_apply, access_is_unaligned, do_not_skip_memory_access});
    Boolean should_read_a_cell = g.b_and(should_apply, do_not_skip_memory_access), should_read_b_cell = is_unaligned_read;
    W8 value_a = in_u32s<8>("uma_read_a"), value_b = in_u32s<8>("uma_read_b");
    for (auto& x : value_a) x = g.mask(x, should_read_a_cell);
    for (auto& x : value_b) x = g.mask(x, should_read_b_cell);

    std::vector<Sponge> relations;
    S12 tail = st.mem_tail;
    V len = st.mem_len;
    auto queue_op = [&](V ts, V index, V rw, const W8& value, Boolean execute) {
        auto enc = memory_query_encode(ts, mem_page, index, rw, g.zero(), value);
        S12 init = absorb8(enc, tail), fin = simulate(init, execute);
        relations.push_back({execute, init, fin});
        tail = g.select_n(execute, fin, tail);
        len = g.select(execute, g.add(len, g.one()), len);
    };
    queue_op(cm.ts_read, a_cell_idx, g.zero(), value_a, should_read_a_cell);
    queue_op(cm.ts_read, b_cell_idx, g.zero(), value_b, should_read_b_cell);

    auto unalignment_bit_mask = g.spread_into_bits(g.lookup(T_UMASHIFT, {unalignment}, 2)[0], 32);
    // UInt256::to_be_bytes: checked byte decomposition, most significant byte first
    std::array<V, 64> bytes_array;
    auto to_be = [&](const W8& w, int off) {
        for (int i = 0; i < 8; ++i) {
            auto b = g.bytes_checked(w[7 - i]);
            for (int k = 0; k < 4; ++k) bytes_array[off + 4 * i + k] = b[3 - k];
        }
    };
    to_be(value_a, 0);
    to_be(value_b, 32);
    std::array<V, 32> selected_word;
    for (auto& x : selected_word) x = g.zero();
    for (int idx = 0; idx < 32; ++idx)
        for (int k = 0; k < 32; ++k) selected_word[k] = g.select(unalignment_bit_mask[idx], bytes_array[idx + k], selected_word[k]);
    V bytes_to_cleanup_if_ptr_read = g.mask(bytes_to_cleanup_out_of_bounds, is_fat_ptr_read);
    auto cleanup_mask = g.spread_into_bits(g.lookup(T_UMACLEAN, {bytes_to_cleanup_if_ptr_read}, 2)[0], 32);
    for (int k = 0; k < 32; ++k) selected_word[k] = g.mask(selected_word[k], cleanup_mask[31 - k]);

    Boolean execute_write = g.multi_and({should_apply, is_write_access, do_not_skip_memory_access});
    Boolean execute_unaligned_write = g.b_and(execute_write, access_is_unaligned);
    std::array<V, 32> written_value_bytes;
    for (int k = 0; k < 32; ++k) written_value_bytes[k] = cm.src1_view.u8[31 - k];
    std::array<V, 64> written_bytes_buffer = bytes_array;
    for (int idx = 0; idx < 32; ++idx)
        for (int k = 0; k < 32; ++k) written_bytes_buffer[idx + k] = g.select(unalignment_bit_mask[idx], written_value_bytes[k], written_bytes_buffer[idx + k]);
    auto from_be_words = [&](const V* bytes) {  // UInt32::from_be_bytes per 4-byte group, limbs least significant first
        W8 w;
        for (int i = 0; i < 8; ++i) w[7 - i] = g.from_le_bytes4(bytes[4 * i + 3], bytes[4 * i + 2], bytes[4 * i + 1], bytes[4 * i]);
        return w;
    };
    W8 a_new_value = from_be_words(&written_bytes_buffer[0]), b_new_value = from_be_words(&written_bytes_buffer[32]);
    queue_op(cm.ts_dst, a_cell_idx, g.one(), a_new_value, execute_write);
    queue_op(cm.ts_dst, b_cell_idx, g.one(), b_new_value, execute_unaligned_write);

    Reg read_value_as_register = zero_reg();
    read_value_as_register.v = from_be_words(selected_word.data());
    Reg incremented_src0 = cm.src0;
    incremented_src0.v[0] = incremented_offset;
    Boolean is_write_access_and_increment = g.b_and(is_write_access, increment_offset);
    Boolean update_dst0 = g.b_or(is_read_access, is_write_access_and_increment);
    Boolean apply_any = g.b_and(should_apply, g.negated(set_panic));
    Boolean should_update_dst0 = g.b_and(apply_any, update_dst0);
    Reg dst0_value = select(is_write_access_and_increment, incremented_src0, read_value_as_register);
    Boolean should_update_dst1 = g.multi_and({apply_any, is_read_access, increment_offset});
    df.dst_0_values.push_back({D.can_write_dst0_into_memory[ZK_VMF_UMA] != 0, should_update_dst0, dst0_value});
    df.dst_1_values.push_back({should_update_dst1, incremented_src0});
    df.pending_exceptions.push_back(g.b_and(should_apply, set_panic));
    df.new_heap_bounds.push_back({grow_heap, new_heap_upper_bound});
    df.new_aux_heap_bounds.push_back({grow_aux_heap, new_aux_heap_upper_bound});
    df.new_ergs_left_candidates.push_back({should_apply, ergs_left_after_growth});
    df.sponge_candidates_to_run.push_back({apply_any, relations});
    df.memory_queue_candidates.push_back({should_apply, len, tail});
}

// apply_calls_and_ret (src/main_vm/opcodes/call_ret.rs:24-512) with callstack_candidate_for_near_call (call_ret_impl/near_call.rs:32-184),
// callstack_candidate_for_far_call (far_call.rs:268-1603) and callstack_candidate_for_ret (ret.rs:29-479)
void VmCircuit::apply_calls_and_ret(const State& st, const Common& cm, const Carry& cr, Diffs& df) {
    const RegView& s0v = cm.src0_view;
    V zero = g.zero();
    // ---- compute_shared_abi_parts (call_ret_impl/mod.rs:39-86)
    FarAbi far_abi;
    far_abi.ergs_passed = s0v.u32[6];
    far_abi.shard_id = s0v.u8[P(ZK_VMP_FAR_CALL_SHARD_ID_BYTE_IDX)];
    far_abi.constructor_call = g.negated(g.is_zero(s0v.u8[P(ZK_VMP_FAR_CALL_CONSTRUCTOR_CALL_BYTE_IDX)]));
    far_abi.system_call = g.negated(g.is_zero(s0v.u8[P(ZK_VMP_FAR_CALL_SYSTEM_CALL_BYTE_IDX)]));
    V forwarding_mode_byte = s0v.u8[P(ZK_VMP_FAR_CALL_FORWARDING_MODE_BYTE_IDX)];
    Forwarding fwd;
    fwd.use_aux_heap = g.equals(forwarding_mode_byte, g.c(P(ZK_VMP_FORWARD_USE_AUX_HEAP)));
    fwd.forward_fat_pointer = g.equals(forwarding_mode_byte, g.c(P(ZK_VMP_FORWARD_FAT_POINTER)));
    fwd.use_heap = g.negated(g.b_or(fwd.use_aux_heap, fwd.forward_fat_pointer));
    Boolean do_not_forward_ptr = g.negated(fwd.forward_fat_pointer);
    CommonAbi abi;
    {   // FatPtrInABI::parse_and_validate (far_call.rs:153-203)
        V offset = s0v.u32[0], page = s0v.u32[1], start = s0v.u32[2], length = s0v.u32[3];
        Boolean non_zero_offset_if_should_be_fresh = g.b_and(g.negated(g.is_zero(offset)), do_not_forward_ptr);
        auto [end_non_inclusive, slice_u32_range_overflow] = g.u32_overflowing_add(start, length);
        Boolean is_invalid_as_slice = g.u32_overflowing_sub(length, offset).second;
        Boolean ptr_is_invalid = g.multi_or({non_zero_offset_if_should_be_fresh, slice_u32_range_overflow, is_invalid_as_slice});
        abi.fat_ptr = mask_into_empty(FatPtr{offset, page, start, length}, ptr_is_invalid);
        abi.upper_bound = end_non_inclusive;
        abi.generally_invalid = ptr_is_invalid;
        abi.is_non_addressable = slice_u32_range_overflow;
    }
    const Ctx& cur = st.cs.ctx;

    // ================= near call =================
    Boolean apply_near_call = type_bit(cm.dec, ZK_VMF_NEAR_CALL);
    Ctx near_old = cur, near_new;
    {
        near_old.pc = cr.next_pc;
        near_new = near_old;
        S4 rollback_tail = in_nums<4>("near_call_rollback_queue_tail");  // oracle.get_rollback_queue_tail_witness_for_call
        near_new.rq_tail = rollback_tail; near_new.rq_head = rollback_tail; near_new.rq_len = zero;
        V ergs_passed_abi = s0v.u32[0];
        Boolean pass_all_ergs = g.is_zero(ergs_passed_abi);
        V ergs_to_pass = g.select(pass_all_ergs, cr.preliminary_ergs_left, ergs_passed_abi);
        auto [remaining_for_this_context, uf] = g.u32_overflowing_sub(cr.preliminary_ergs_left, ergs_to_pass);
        V remaining_ergs_if_pass = g.select(uf, zero, remaining_for_this_context);
        V passed_ergs_if_pass = g.select(uf, cr.preliminary_ergs_left, ergs_to_pass);
        near_old.ergs = remaining_ergs_if_pass;
        near_new.ergs = passed_ergs_if_pass;
        near_new.pc = cm.dec.imm0;
        near_new.eh = cm.dec.imm1;
        near_new.is_local = g.one();
    }

    // ================= far call =================
    Boolean apply_far_call = type_bit(cm.dec, ZK_VMF_FAR_CALL);
    Ctx far_old = cur, far_new = uninitialized_ctx();
    S12 new_decommittment_queue_tail;
    V new_decommittment_queue_len, far_new_forward_queue_len, new_memory_pages_counter;
    S4 far_new_forward_queue_tail;
    std::vector<Sponge> far_sponges;
    Boolean far_pending_exception;
    Reg far_new_r1, far_new_r2;
    Boolean far_cleanup_register;
    {
        Boolean execute = apply_far_call;
        Boolean is_delegated_call = variant_bit(cm.dec, ZK_VMV_FAR_DELEGATE), is_mimic_call = variant_bit(cm.dec, ZK_VMV_FAR_MIMIC);
        Boolean is_kernel_mode = g.B(cur.is_kernel);
        far_old.pc = cr.next_pc;
        far_new.heap_bound = g.c(P(ZK_VMP_NEW_FRAME_MEMORY_STIPEND));
        far_new.aux_heap_bound = g.c(P(ZK_VMP_NEW_FRAME_MEMORY_STIPEND));
        const Reg& implicit = st.regs[P(ZK_VMP_CALL_IMPLICIT_PARAMETER_REG_IDX)];
        A5 caller_address_for_mimic = {implicit.v[0], implicit.v[1], implicit.v[2], implicit.v[3], implicit.v[4]};
        const W8& s1 = cm.src1_view.u32;
        A5 destination_address = {s1[0], s1[1], s1[2], s1[3], s1[4]};
        Boolean is_static_call = flag_bit(cm.dec, ZK_VMFL_FAR_CALL_STATIC), is_call_shard = flag_bit(cm.dec, ZK_VMFL_FAR_CALL_SHARD);
        V caller_shard_id = far_old.this_shard;
        V destination_shard = g.select(is_call_shard, far_abi.shard_id, caller_shard_id);
        Boolean target_is_zkporter = g.negated(g.is_zero(destination_shard));
        V destination_16_32 = g.from_le_bytes2(cm.src1_view.u8[2], cm.src1_view.u8[3]);
        Boolean target_is_kernel = g.multi_and({g.is_zero(destination_16_32), g.is_zero(s1[1]), g.is_zero(s1[2]), g.is_zero(s1[3]), g.is_zero(s1[4])});
        far_abi.constructor_call = g.b_and(far_abi.constructor_call, is_kernel_mode);
        far_abi.system_call = g.b_and(far_abi.system_call, target_is_kernel);
        V timestamp_for_decommit = cm.ts_first;
        V default_target_memory_page = st.page_counter, new_base_page = st.page_counter;
        V counter_inc = g.u32_add_no_overflow(st.page_counter, g.c(P(ZK_VMP_NEW_MEMORY_PAGES_PER_FAR_CALL)));
        new_memory_pages_counter = g.select(execute, counter_inc, st.page_counter);

        // ---- may_be_read_code_hash (far_call.rs:1104-1280)
        Boolean zkporter_is_available = g.B(gctx_zkporter);
        Boolean target_is_porter_and_its_available = g.b_and(target_is_zkporter, zkporter_is_available);
        Boolean can_read = g.b_or(g.negated(target_is_zkporter), target_is_porter_and_its_available);
        Boolean should_read = g.b_and(execute, can_read);
        Boolean needs_porter_mask = g.b_and(target_is_zkporter, g.negated(zkporter_is_available));
        LogQuery log;
        log.address[0] = UInt32{g.c(P(ZK_VMP_DEPLOYER_SYSTEM_CONTRACT_ADDRESS_LOW))};
        for (int i = 1; i < 5; ++i) log.address[i] = UInt32{zero};
        for (int i = 0; i < 8; ++i) log.key.inner[i] = UInt32{i < 5 ? destination_address[i] : zero};
        log.rw_flag = g.bool_const(false); log.aux_byte = UInt8{g.c(P(ZK_VMP_STORAGE_AUX_BYTE))}; log.rollback = g.bool_const(false);
        log.is_service = g.bool_const(false); log.shard_id = UInt8{destination_shard}; log.tx_number_in_block = UInt32{st.tx_number};
        log.timestamp = UInt32{timestamp_for_decommit};
        W8 code_hash_from_storage = in_u32s<8>("far_call_code_hash_read_value");  // oracle.get_storage_read_witness
        for (int i = 0; i < 8; ++i) { log.read_value.inner[i] = UInt32{code_hash_from_storage[i]}; log.written_value.inner[i] = UInt32{code_hash_from_storage[i]}; }
        W8 bytecode_hash = code_hash_from_storage;
        Boolean bytecode_is_empty = g.all_zero(bytecode_hash);
        Boolean mask_for_default_aa = g.multi_and({should_read, bytecode_is_empty, g.negated(target_is_kernel)});
        bytecode_hash = g.select_n(mask_for_default_aa, gctx_default_aa, bytecode_hash);
        W8 zero8;
        for (auto& x : zero8) x = zero;
        bytecode_hash = g.select_n(needs_porter_mask, zero8, bytecode_hash);
        Boolean t0 = g.b_and(bytecode_is_empty, g.negated(mask_for_default_aa));
        Boolean bytecode_hash_is_trivial = g.multi_or({t0, needs_porter_mask, g.negated(should_read)});
        {   // construct_hash_relations_code_hash_read (far_call.rs:1282-1416)
            auto enc = encode_log(log);
            S12 empty = g.empty_state();
            auto chunk = [&](int k) { std::array<V, 8> e; for (int i = 0; i < 8; ++i) e[i] = enc[8 * k + i]; return e; };
            S12 r0_init = absorb8(chunk(0), empty), r0_fin = simulate(r0_init, should_read);
            S12 r1_init = absorb8(chunk(1), r0_fin), r1_fin = simulate(r1_init, should_read);
            std::array<V, 8> e2;
            for (int i = 0; i < 4; ++i) { e2[i] = enc[16 + i]; e2[4 + i] = st.cs.fwd_tail[i]; }
            S12 r2_init = absorb8(e2, r1_fin), r2_fin = simulate(r2_init, should_read);
            far_new_forward_queue_len = g.select(should_read, g.add(st.cs.fwd_len, g.one()), st.cs.fwd_len);
            far_sponges.push_back({should_read, r0_init, r0_fin});
            far_sponges.push_back({should_read, r1_init, r1_fin});
            far_sponges.push_back({should_read, r2_init, r2_fin});
            far_new_forward_queue_tail = g.select_n(should_read, S4{r2_fin[0], r2_fin[1], r2_fin[2], r2_fin[3]}, st.cs.fwd_tail);
        }
        V target_code_memory_page = g.select(bytecode_hash_is_trivial, zero, default_target_memory_page);
        auto upper = g.bytes_checked(bytecode_hash[7]);
        V version_byte = upper[3];
        V code_hash_version_byte = g.c(P(ZK_VMP_CODE_HASH_VERSION_BYTE));
        Boolean versioned_byte_is_invalid = g.negated(g.equals(version_byte, code_hash_version_byte));
        V marker_byte = upper[2];
        Boolean is_normal_call_marker = g.is_zero(marker_byte);
        Boolean is_constructor_call_marker = g.equals(marker_byte, g.c(P(ZK_VMP_CODE_YET_CONSTRUCTED_MARKER)));
        Boolean unknown_marker = g.negated(g.b_or(is_normal_call_marker, is_constructor_call_marker));
        Boolean code_format_exception = g.b_or(versioned_byte_is_invalid, unknown_marker);
        Boolean can_call_normally = g.b_and(is_normal_call_marker, g.negated(far_abi.constructor_call));
        Boolean can_call_constructor = g.b_and(is_constructor_call_marker, far_abi.constructor_call);
        Boolean can_call_code = g.b_or(can_call_normally, can_call_constructor);
        V at_rest_top_word = g.from_le_bytes4(upper[0], upper[1], g.c(P(ZK_VMP_CODE_AT_REST_MARKER)), code_hash_version_byte);
        W8 at_storage_format = bytecode_hash;
        at_storage_format[7] = at_rest_top_word;
        W8 masked_value_if_mask = g.select_n(target_is_kernel, zero8, gctx_default_aa);
        W8 masked_bytecode_hash = g.select_n(can_call_code, at_storage_format, masked_value_if_mask);
        auto masked_upper = g.bytes_checked(masked_bytecode_hash[7]);
        V code_hash_length_in_words = g.mask_negated(g.from_le_bytes2(masked_upper[0], masked_upper[1]), code_format_exception);
        Boolean call_now_in_construction_kernel = g.b_and(g.negated(can_call_code), target_is_kernel);
        Boolean src0_is_integer = g.negated(g.B(s0v.is_ptr));
        Boolean fat_ptr_expected_exception = g.b_and(fwd.forward_fat_pointer, src0_is_integer);
        Boolean exceptions_collapsed = g.multi_or({code_format_exception, call_now_in_construction_kernel, fat_ptr_expected_exception,
                                                   abi.generally_invalid, abi.is_non_addressable});
        FatPtr fat_ptr_adjusted_if_forward = readjust(abi.fat_ptr);
        V page = g.select(fwd.use_heap, cr.heap_page, cr.aux_heap_page);
        FatPtr fat_ptr_for_heaps{zero, page, abi.fat_ptr.start, abi.fat_ptr.length};
        FatPtr final_fat_ptr = mask_into_empty(select(fwd.forward_fat_pointer, fat_ptr_adjusted_if_forward, fat_ptr_for_heaps), exceptions_collapsed);
        V upper_bound = g.mask_negated(abi.upper_bound, exceptions_collapsed);
        Boolean penalize_heap_overflow = g.b_and(abi.is_non_addressable, do_not_forward_ptr);
        upper_bound = g.select(penalize_heap_overflow, g.c(0xffffffffu), upper_bound);
        V heap_max_accessed = g.mask(upper_bound, fwd.use_heap);
        V heap_bound = far_old.heap_bound;
        auto [heap_growth0, uf_h] = g.u32_overflowing_sub(heap_max_accessed, heap_bound);
        V heap_growth = g.mask_negated(heap_growth0, uf_h);
        V new_heap_upper_bound = g.select(uf_h, heap_bound, heap_max_accessed);
        Boolean grow_heap = g.b_and(fwd.use_heap, execute);
        V aux_max_accessed = g.mask(upper_bound, fwd.use_aux_heap);
        V aux_heap_bound = far_old.aux_heap_bound;
        auto [aux_growth0, uf_a] = g.u32_overflowing_sub(aux_max_accessed, aux_heap_bound);
        V aux_growth = g.mask_negated(aux_growth0, uf_a);
        V new_aux_heap_upper_bound = g.select(uf_a, aux_heap_bound, aux_max_accessed);
        Boolean grow_aux_heap = g.b_and(fwd.use_aux_heap, execute);
        V growth_cost = g.mask(heap_growth, grow_heap);
        growth_cost = g.select(grow_aux_heap, aux_growth, growth_cost);
        auto [ergs_after_growth0, uf_g] = g.u32_overflowing_sub(cr.preliminary_ergs_left, growth_cost);
        V ergs_left_after_growth = g.mask_negated(ergs_after_growth0, uf_g);
        far_old.heap_bound = g.select(grow_heap, new_heap_upper_bound, far_old.heap_bound);
        far_old.aux_heap_bound = g.select(grow_aux_heap, new_aux_heap_upper_bound, far_old.aux_heap_bound);
        V callee_stipend = zero;  // FORCED_ERGS_FOR_MSG_VALUE_SIMUALTOR == false (far_call.rs:28,734-736)
        auto [ergs_after_extra0, uf_x] = g.u32_overflowing_sub(ergs_left_after_growth, callee_stipend);
        V ergs_left_after_extra_costs = g.mask_negated(ergs_after_extra0, uf_x);
        callee_stipend = g.mask_negated(callee_stipend, uf_x);
        Boolean exception = g.multi_or({exceptions_collapsed, uf_g, uf_x});
        Boolean should_decommit0 = g.b_and(execute, g.negated(exception));
        target_code_memory_page = g.mask(target_code_memory_page, should_decommit0);

        // ---- add_to_decommittment_queue (far_call.rs:1418-1603)
        V code_memory_page, ergs_remaining_after_decommit;
        Boolean not_enough_ergs_to_decommit;
        {
            V cost_of_decommittment = g.u32_non_widening_mul(g.c(P(ZK_VMP_ERGS_PER_CODE_WORD_DECOMMITTMENT)), code_hash_length_in_words);
            auto [ergs_after_decommit_may_be, uf] = g.u32_overflowing_sub(ergs_left_after_extra_costs, cost_of_decommittment);
            not_enough_ergs_to_decommit = uf;
            Boolean should_decommit = g.b_and(should_decommit0, g.negated(uf));
            V ergs_rem = g.select(should_decommit, ergs_after_decommit_may_be, ergs_left_after_extra_costs);
            V suggested_page = in_u32("far_call_decommit_suggested_page");  // oracle.get_decommittment_request_suggested_page
            Boolean is_first = g.equals(target_code_memory_page, suggested_page);
            DecommitQuery dq;
            for (int i = 0; i < 8; ++i) dq.code_hash.inner[i] = UInt32{masked_bytecode_hash[i]};
            dq.page = UInt32{suggested_page}; dq.is_first = is_first; dq.timestamp = UInt32{timestamp_for_decommit};
            Boolean refund = g.b_and(should_decommit, g.negated(is_first));
            ergs_remaining_after_decommit = g.select(refund, ergs_left_after_extra_costs, ergs_rem);
            auto enc = encode_decommit_query(g, dq);
            S12 init = absorb8(enc, st.dec_tail), fin = simulate(init, should_decommit);
            far_sponges.push_back({should_decommit, init, fin});
            new_decommittment_queue_tail = g.select_n(should_decommit, fin, st.dec_tail);
            new_decommittment_queue_len = g.select(should_decommit, g.add(st.dec_len, g.one()), st.dec_len);
            code_memory_page = g.select(should_decommit, suggested_page, g.c(P(ZK_VMP_UNMAPPED_PAGE)));
        }
        far_pending_exception = g.b_or(exception, not_enough_ergs_to_decommit);
        S4 rollback_tail = in_nums<4>("far_call_rollback_queue_tail");  // oracle.get_rollback_queue_tail_witness_for_call
        far_new.rq_tail = rollback_tail; far_new.rq_head = rollback_tail; far_new.rq_len = zero;
        // 63/64 rule
        V preliminary_ergs_left = ergs_remaining_after_decommit;
        V ergs_div_by_64 = g.u32_div_by_constant(preliminary_ergs_left, 64).first;
        V max_passable = g.mul(ergs_div_by_64, g.c(63));
        V leftover = g.sub(preliminary_ergs_left, max_passable);
        auto [remaining_from_max_passable, uf_p] = g.u32_overflowing_sub(max_passable, far_abi.ergs_passed);
        V leftover_and_remaining_if_no_uf = g.u32_overflowing_add(leftover, remaining_from_max_passable).first;
        V ergs_to_pass = g.select(uf_p, max_passable, far_abi.ergs_passed);
        V remaining_for_this_context = g.select(uf_p, leftover, leftover_and_remaining_if_no_uf);
        V passed_ergs_if_pass = g.u32_add_no_overflow(ergs_to_pass, callee_stipend);
        far_old.ergs = remaining_for_this_context;
        V new_this_shard_id = g.select(is_delegated_call, caller_shard_id, destination_shard);
        A5 this_for_next = g.select_n(is_delegated_call, far_old.this_, destination_address);
        A5 caller_for_next = g.select_n(is_delegated_call, far_old.caller, far_old.this_);
        caller_for_next = g.select_n(is_mimic_call, caller_address_for_mimic, caller_for_next);
        Boolean next_is_static = g.b_or(is_static_call, g.B(far_old.is_static));
        far_new.ergs = passed_ergs_if_pass;
        far_new.pc = zero;
        far_new.eh = cm.dec.imm0;
        far_new.is_static = next_is_static.v;
        far_new.is_kernel = g.select(is_delegated_call, far_old.is_kernel, target_is_kernel.v);
        far_new.code_shard = destination_shard;
        far_new.code_address = destination_address;
        far_new.this_shard = new_this_shard_id;
        far_new.this_ = this_for_next;
        far_new.caller = caller_for_next;
        far_new.caller_shard = caller_shard_id;
        far_new.code_page = code_memory_page;
        far_new.base_page = new_base_page;
        far_new.ctx_u128 = g.select_n(is_delegated_call, far_old.ctx_u128, st.ctx_u128);
        far_new.is_local = zero;
        far_new_r1 = fat_ptr_into_register(final_fat_ptr);
        far_new_r2 = zero_reg();
        far_new_r2.v[0] = g.fma(1, far_abi.constructor_call.v, g.one(), 2, far_abi.system_call.v);
        far_cleanup_register = g.b_and(execute, g.negated(far_abi.system_call));
    }

    // ================= ret =================
    Boolean apply_ret = type_bit(cm.dec, ZK_VMF_RET);
    Ctx ret_new, originally_popped;
    S12 previous_callstack_state;
    S4 ret_new_forward_queue_tail;
    V ret_new_forward_queue_len;
    Boolean ret_is_panic, did_return_from_far_call;
    Reg ret_new_r1;
    Boolean update_specific_registers_on_ret;
    {
        Boolean execute = apply_ret;
        Boolean is_ret_ok = variant_bit(cm.dec, ZK_VMV_RET_OK), is_ret_revert = variant_bit(cm.dec, ZK_VMV_RET_REVERT),
                is_ret_panic = variant_bit(cm.dec, ZK_VMV_RET_PANIC);
        Boolean is_local_frame = g.B(cur.is_local);
        Reg src0 = cm.src0;  // conditionally_erase on panic
        src0.ptr = g.mask_negated(src0.ptr, is_ret_panic);
        for (auto& x : src0.v) x = g.mask_negated(x, is_ret_panic);
        Boolean is_to_label = flag_bit(cm.dec, ZK_VMFL_RET_TO_LABEL);
        V label_pc = cm.dec.imm0;
        {   // oracle.get_callstack_witness: ExecutionContextRecord::create_without_value applies every field's constraints
            lay("ret_popped_context", CTX_WORDS);
            const int kinds[CTX_WORDS] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 0, 0, 0, 0, 0, 0, 0, 0, 32,
                                          16, 16, 16, 32, 1, 1, 8, 8, 8, 32, 32, 32, 32, 1};
            V f[CTX_WORDS];
            for (int i = 0; i < CTX_WORDS; ++i) f[i] = alloc_kind(kinds[i]);
            ret_new = Ctx::unflatten(f);
            previous_callstack_state = in_nums<12>("ret_previous_callstack_state");
        }
        originally_popped = ret_new;
        Boolean forward_fat_pointer = fwd.forward_fat_pointer;
        Boolean src0_is_integer = g.negated(g.B(src0.ptr));
        Boolean is_far_return = g.negated(is_local_frame);
        Boolean fat_ptr_expected_exception = g.multi_and({forward_fat_pointer, src0_is_integer, is_far_return});
        Boolean uf_page = g.u32_overflowing_sub(abi.fat_ptr.page, cur.base_page).second;
        Boolean non_unidirectional_forwarding = g.b_and(forward_fat_pointer, uf_page);
        Boolean exceptions_collapsed = g.multi_or({fat_ptr_expected_exception, non_unidirectional_forwarding, is_ret_panic});
        FatPtr fat_ptr = mask_into_empty(abi.fat_ptr, exceptions_collapsed);
        FatPtr fat_ptr_adjusted_if_forward = readjust(fat_ptr);
        V page = g.select(fwd.use_heap, cr.heap_page, cr.aux_heap_page);
        FatPtr fat_ptr_for_heaps{zero, page, fat_ptr.start, fat_ptr.length};
        fat_ptr = select(forward_fat_pointer, fat_ptr_adjusted_if_forward, fat_ptr_for_heaps);
        V upper_bound = g.mask_negated(abi.upper_bound, exceptions_collapsed);
        Boolean penalize_heap_overflow = g.b_and(abi.is_non_addressable, do_not_forward_ptr);
        upper_bound = g.select(penalize_heap_overflow, g.c(0xffffffffu), upper_bound);
        V heap_bound = cur.heap_bound, aux_heap_bound = cur.aux_heap_bound;
        auto [heap_growth0, uf_h] = g.u32_overflowing_sub(g.mask(upper_bound, fwd.use_heap), heap_bound);
        V heap_growth = g.mask_negated(heap_growth0, uf_h);
        Boolean grow_heap = g.multi_and({fwd.use_heap, execute, is_far_return});
        auto [aux_growth0, uf_a] = g.u32_overflowing_sub(g.mask(upper_bound, fwd.use_aux_heap), aux_heap_bound);
        V aux_growth = g.mask_negated(aux_growth0, uf_a);
        Boolean grow_aux_heap = g.multi_and({fwd.use_aux_heap, execute, is_far_return});
        V growth_cost = g.mask(heap_growth, grow_heap);
        growth_cost = g.select(grow_aux_heap, aux_growth, growth_cost);
        auto [ergs_after_growth0, uf_g] = g.u32_overflowing_sub(cr.preliminary_ergs_left, growth_cost);
        V ergs_left_after_growth = g.mask_negated(ergs_after_growth0, uf_g);
        ergs_left_after_growth = g.select(is_local_frame, cr.preliminary_ergs_left, ergs_left_after_growth);
        Boolean non_local_frame_panic = g.b_and(g.multi_or({exceptions_collapsed, uf_g, is_ret_panic}), is_far_return);
        FatPtr final_fat_ptr = mask_into_empty(fat_ptr, non_local_frame_panic);
        ret_new.ergs = g.u32_add_no_overflow(ergs_left_after_growth, ret_new.ergs);
        ret_new.heap_bound = g.select(is_local_frame, heap_bound, ret_new.heap_bound);
        ret_new.aux_heap_bound = g.select(is_local_frame, aux_heap_bound, ret_new.aux_heap_bound);
        Boolean should_perform_revert = g.multi_or({is_ret_revert, is_ret_panic, non_local_frame_panic});
        Boolean perform_revert = g.b_and(execute, should_perform_revert);
        for (int i = 0; i < 4; ++i) g.cond_enforce_equal(perform_revert, cur.rq_head[i], st.cs.fwd_tail[i]);
        V new_forward_queue_len_if_revert = g.u32_add_no_overflow(st.cs.fwd_len, cur.rq_len);
        Boolean should_perform_ret_ok = g.multi_and({execute, is_ret_ok, g.negated(non_local_frame_panic)});
        for (int i = 0; i < 4; ++i) g.cond_enforce_equal(should_perform_ret_ok, ret_new.rq_head[i], cur.rq_tail[i]);
        V new_rollback_queue_len_if_ok = g.u32_add_no_overflow(ret_new.rq_len, cur.rq_len);
        ret_new_forward_queue_tail = g.select_n(should_perform_revert, cur.rq_tail, st.cs.fwd_tail);
        ret_new_forward_queue_len = g.select(should_perform_revert, new_forward_queue_len_if_revert, st.cs.fwd_len);
        ret_new.rq_head = g.select_n(should_perform_ret_ok, cur.rq_head, ret_new.rq_head);
        ret_new.rq_len = g.select(should_perform_ret_ok, new_rollback_queue_len_if_ok, ret_new.rq_len);
        Boolean should_use_label = g.b_and(is_to_label, is_local_frame);
        V ok_ret_pc = g.select(should_use_label, label_pc, ret_new.pc);
        V eh_pc = g.select(should_use_label, label_pc, cur.eh);
        ret_new.pc = g.select(perform_revert, eh_pc, ok_ret_pc);
        ret_new_r1 = fat_ptr_into_register(final_fat_ptr);
        update_specific_registers_on_ret = g.b_and(execute, is_far_return);
        ret_is_panic = g.b_or(is_ret_panic, non_local_frame_panic);
        did_return_from_far_call = is_far_return;
    }

    // ================= merge (call_ret.rs:119-512) =================
    Boolean is_call_like = g.b_or(apply_near_call, apply_far_call);
    Boolean apply_any = g.b_or(is_call_like, apply_ret);
    Boolean is_ret_panic_if_apply = g.b_and(ret_is_panic, apply_ret);
    Boolean pending_exception_if_far_call = g.b_and(far_pending_exception, apply_far_call);
    Boolean is_far_return = g.b_and(apply_ret, did_return_from_far_call);
    Boolean reset_context_value = g.b_or(is_far_return, apply_far_call);
    Ctx new_callstack_entry = select(apply_far_call, far_new, near_new);
    new_callstack_entry = select(apply_ret, ret_new, new_callstack_entry);
    Ctx old_callstack_entry = select(apply_far_call, far_old, near_old);
    old_callstack_entry = select(apply_ret, originally_popped, old_callstack_entry);
    S12 current_state = g.select_n(apply_ret, previous_callstack_state, st.cs.sponge);
    auto enc = encode_ctx(old_callstack_entry);
    std::vector<Sponge> common_relations;
    for (int r = 0; r < 4; ++r) {
        std::array<V, 8> e;
        for (int i = 0; i < 8; ++i) e[i] = enc[8 * r + i];
        S12 init = absorb8(e, current_state), fin = simulate(init, apply_any);
        common_relations.push_back({apply_any, init, fin});
        current_state = fin;
    }
    for (int i = 0; i < 12; ++i) g.cond_enforce_equal(apply_ret, current_state[i], st.cs.sponge[i]);
    S12 new_callstack_state = g.select_n(apply_ret, previous_callstack_state, current_state);
    V depth_increased = g.add(st.cs.depth, g.one());
    auto [depth_decreased, uf_d] = g.u32_overflowing_sub(st.cs.depth, g.c(1));
    g.cond_enforce_false(uf_d, apply_ret);
    Callstack new_callstack;
    new_callstack.ctx = new_callstack_entry;
    new_callstack.fwd_tail = g.select_n(apply_ret, ret_new_forward_queue_tail, far_new_forward_queue_tail);
    new_callstack.fwd_len = g.select(apply_ret, ret_new_forward_queue_len, far_new_forward_queue_len);
    new_callstack.depth = g.select(apply_ret, depth_decreased, depth_increased);
    new_callstack.sponge = new_callstack_state;
    for (auto& s : far_sponges) common_relations.push_back(s);
    FlagsPort new_flags = cm.reseted_flags;
    new_flags.of = is_ret_panic_if_apply.v;
    df.sponge_candidates_to_run.push_back({apply_any, common_relations});
    df.flags.push_back({apply_any, new_flags});
    // specific register updates: far call r1 / r2, ret r1
    df.specific_registers_updates[0].push_back({apply_far_call, far_new_r1});
    df.specific_registers_updates[1].push_back({apply_far_call, far_new_r2});
    df.specific_registers_updates[0].push_back({update_specific_registers_on_ret, ret_new_r1});
    const int abi0 = (int)P(ZK_VMP_CALL_SYSTEM_ABI_REGISTERS_BEGIN), abi1 = (int)P(ZK_VMP_CALL_SYSTEM_ABI_REGISTERS_END);
    const int res0 = (int)P(ZK_VMP_CALL_RESERVED_RANGE_BEGIN), res1 = (int)P(ZK_VMP_CALL_RESERVED_RANGE_END);
    const int implicit = (int)P(ZK_VMP_CALL_IMPLICIT_PARAMETER_REG_IDX);
    for (int r = 0; r < NREG; ++r) {
        const bool in_abi = r >= abi0 && r < abi1, in_res = r >= res0 && r < res1, is_imp = r == implicit;
        // zeroing: far call first, then ret (call_ret.rs:452-465)
        if (is_imp || in_res) df.specific_registers_zeroing[r].push_back(apply_far_call);
        else if (in_abi) df.specific_registers_zeroing[r].push_back(far_cleanup_register);
        if (r >= 1) df.specific_registers_zeroing[r].push_back(update_specific_registers_on_ret);
        if (in_abi || in_res || is_imp) df.remove_ptr_on_specific_registers[r].push_back(apply_far_call);
        if (r >= 1) df.remove_ptr_on_specific_registers[r].push_back(update_specific_registers_on_ret);
    }
    df.pending_exceptions.push_back(pending_exception_if_far_call);
    df.callstacks.push_back({apply_any, new_callstack});
    df.memory_page_counters = new_memory_pages_counter;
    df.context_u128_candidates.push_back({reset_context_value, S4{zero, zero, zero, zero}});
    df.decommitment_queue_candidates.push_back({apply_far_call, new_decommittment_queue_len, new_decommittment_queue_tail});
}

// enforce_addition_relation — src/main_vm/opcodes/mod.rs:101-126
void VmCircuit::enforce_addition_relation(const AddSubRelation& r) {
    V carry = g.zero();
    for (int i = 0; i < 8; ++i) {  // UIntXAddGate::<32>::enforce_add_relation_compute_carry: a + b + cin = c + 2^32 cout
        V cout = cs.alloc_var();
        V t = cs.alloc_var();  // (a + b + cin - c) / 2^32 computed through the integer sum
        V ins[3] = {r.a[i], r.b[i], carry};
        V outs[2] = {t, cout};
        cs.emit_op(ZK_OP_UADD, 32, 0, ins, 3, outs, 2, nullptr, 0);  // t is the low word (== c for a satisfied relation)
        V vars[5] = {r.a[i], r.b[i], carry, r.c[i], cout};
        uint64_t k = 1ull << 32;
        cs.place_gate(ZK_GATE_UINTX_ADD, vars, 5, &k, 1);
        cs.place_gate(ZK_GATE_BOOLEAN, &cout, 1, nullptr, 0);
        carry = cout;
    }
    g.enforce_equal(carry, r.of);
}

// enforce_mul_relation — src/main_vm/opcodes/mod.rs:130-180: a * b + rem = mul_low + 2^256 mul_high through 64 UInt32::fma_with_carry
void VmCircuit::enforce_mul_relation(const MulDivRelation& r) {
    if (cs.gate_is_allowed(ZK_GATE_U8X4_FMA)) {   // the reference's only branch (`if cs.gate_is_allowed::<U8x4FMAGate>()`, mod.rs:146)
        using B4 = G::Bytes4;
        const B4 Z = {g.zero(), g.zero(), g.zero(), g.zero()};
        std::array<B4, 8> ab, bb;
        std::array<B4, 16> partial;
        for (int i = 0; i < 8; ++i) {   // "fields a, b and rem will be range checked" (mod.rs:127): checked byte decompositions
            ab[i] = g.bytes_checked(r.a[i]); bb[i] = g.bytes_checked(r.b[i]);
            partial[i] = g.bytes_checked(r.rem[i]); partial[8 + i] = Z;
        }
        for (int a_idx = 0; a_idx < 8; ++a_idx) {
            B4 overflow = Z;
            for (int b_idx = 0; b_idx < 8; ++b_idx) {
                auto lh = g.u8x4_fma_with_carry(ab[a_idx], bb[b_idx], partial[a_idx + b_idx], overflow);
                partial[a_idx + b_idx] = lh.first;
                overflow = lh.second;
            }
            // end of chain: partial_result[a_idx + 8] is still the zero it was initialised with when the reference adds the
            // overflow to it (add_no_overflow(0, x)): the sum is the overflow word itself
            partial[a_idx + 8] = overflow;
        }
        for (int i = 0; i < 16; ++i) {   // Num::enforce_equal(partial_result[i], mul_low / mul_high[i]): the bytes recompose to the given word
            V vars[5] = {partial[i][0], partial[i][1], partial[i][2], partial[i][3], i < 8 ? r.mul_low[i] : r.mul_high[i - 8]};
            uint64_t k[4] = {1, 1ull << 8, 1ull << 16, 1ull << 24};
            cs.place_gate(ZK_GATE_REDUCTION4, vars, 5, k, 4);
        }
        return;
    }
    // engines configured without U8x4FMAGate (zk_circuit_main_vm_configure_flags: ZK_VM_CFG_U32_FMA_ROLE): one-relation u32 gates
    std::array<V, 16> partial;
    for (int i = 0; i < 8; ++i) { partial[i] = r.rem[i]; partial[8 + i] = g.zero(); }
    for (int a_idx = 0; a_idx < 8; ++a_idx) {
        V overflow = g.zero();
        for (int b_idx = 0; b_idx < 8; ++b_idx) {
            auto lh = g.u32_fma_with_carry(UInt32{r.a[a_idx]}, UInt32{r.b[b_idx]}, UInt32{partial[a_idx + b_idx]}, UInt32{overflow});
            partial[a_idx + b_idx] = lh.first.v;
            overflow = lh.second.v;
        }
        partial[a_idx + 8] = g.u32_add_no_overflow(partial[a_idx + 8], overflow);
    }
    for (int i = 0; i < 8; ++i) { g.enforce_equal(partial[i], r.mul_low[i]); g.enforce_equal(partial[8 + i], r.mul_high[i]); }
}

// vm_cycle — src/main_vm/cycle.rs:28-795
State VmCircuit::vm_cycle(State st) {
    Common cm;
    Carry cr;
    create_prestate(st, cm, cr);
    const State& draft = st;
    Diffs df;
    // apply_nop: nothing to record (opcodes/nop.rs)
    apply_add_sub(draft, cm, cr, df);
    apply_jump(draft, cm, cr, df);
    apply_binop(draft, cm, cr, df);
    apply_context(draft, cm, cr, df);
    apply_ptr(draft, cm, cr, df);
    apply_log(draft, cm, cr, df);
    apply_calls_and_ret(draft, cm, cr, df);
    apply_mul_div(draft, cm, cr, df);
    apply_shifts(draft, cm, cr, df);
    apply_uma(draft, cm, cr, df);

    State ns = draft;
    std::vector<Boolean> write_dst0_bools, reg_only_bools;
    for (auto& el : df.dst_0_values) (el.can_write_into_memory ? write_dst0_bools : reg_only_bools).push_back(el.flag);
    Boolean dst0_update_potentially_to_memory = g.multi_or(write_dst0_bools);
    Boolean can_update_dst0_as_register_only = g.multi_or(reg_only_bools);
    auto dot_over = [&](auto&& flag_of, auto&& value_of, size_t n) {
        std::vector<V> a, b;
        for (size_t i = 0; i < n; ++i) { a.push_back(flag_of(i)); b.push_back(value_of(i)); }
        return g.dot(a, b);
    };
    Reg dst0, dst1;
    dst0.ptr = dot_over([&](size_t i) { return df.dst_0_values[i].flag.v; }, [&](size_t i) { return df.dst_0_values[i].reg.ptr; }, df.dst_0_values.size());
    for (int l = 0; l < 8; ++l)
        dst0.v[l] = dot_over([&](size_t i) { return df.dst_0_values[i].flag.v; }, [&](size_t i) { return df.dst_0_values[i].reg.v[l]; }, df.dst_0_values.size());
    dst1.ptr = dot_over([&](size_t i) { return df.dst_1_values[i].first.v; }, [&](size_t i) { return df.dst_1_values[i].second.ptr; }, df.dst_1_values.size());
    for (int l = 0; l < 8; ++l)
        dst1.v[l] = dot_over([&](size_t i) { return df.dst_1_values[i].first.v; }, [&](size_t i) { return df.dst_1_values[i].second.v[l]; }, df.dst_1_values.size());

    Boolean perform_dst0_memory_write_update = g.b_and(cr.dst0_performs_memory_access, dst0_update_potentially_to_memory);
    // may_be_write_memory — cycle.rs:799-935 (on the prestate's memory queue: UMA never writes dst0 to memory)
    Sponge dst0_write_sponge;
    {
        auto enc = memory_query_encode(cm.ts_dst, cr.dst0_location.page, cr.dst0_location.index, g.one(), dst0.ptr, dst0.v);
        dst0_write_sponge.init = absorb8(enc, draft.mem_tail);
        dst0_write_sponge.fin = simulate(dst0_write_sponge.init, perform_dst0_memory_write_update);
        dst0_write_sponge.flag = perform_dst0_memory_write_update;
        ns.mem_len = g.select(perform_dst0_memory_write_update, g.add(draft.mem_len, g.one()), draft.mem_len);
        ns.mem_tail = g.select_n(perform_dst0_memory_write_update, dst0_write_sponge.fin, draft.mem_tail);
    }
    Boolean t = g.b_and(g.negated(cr.dst0_performs_memory_access), dst0_update_potentially_to_memory);
    Boolean dst0_update_register = g.b_or(can_update_dst0_as_register_only, t);
    W8 zero8;
    for (auto& x : zero8) x = g.zero();
    for (int idx = 0; idx < NREG; ++idx) {
        Boolean write_as_dst0 = g.b_and(dst0_update_register, cm.dec.dst_regs[0][idx]);
        Boolean write_as_dst1 = cm.dec.dst_regs[1][idx];
        std::vector<Boolean> apply_ptr_update_as_dst0 = {write_as_dst0};
        std::vector<std::pair<Boolean, V>> is_ptr_as_dst0 = {{write_as_dst0, dst0.ptr}};
        std::vector<std::pair<Boolean, W8>> value_as_dst0 = {{write_as_dst0, dst0.v}};
        for (auto& su : df.specific_registers_updates[idx]) {
            apply_ptr_update_as_dst0.push_back(su.first);
            is_ptr_as_dst0.push_back({su.first, su.second.ptr});
            value_as_dst0.push_back({su.first, su.second.v});
        }
        if (!df.remove_ptr_on_specific_registers[idx].empty()) {
            Boolean remove_ptr_marker = g.multi_or(df.remove_ptr_on_specific_registers[idx]);
            apply_ptr_update_as_dst0.push_back(remove_ptr_marker);
            is_ptr_as_dst0.push_back({remove_ptr_marker, g.zero()});
        }
        if (!df.specific_registers_zeroing[idx].empty()) value_as_dst0.push_back({g.multi_or(df.specific_registers_zeroing[idx]), zero8});
        Boolean any_ptr_update_as_dst0 = g.multi_or(apply_ptr_update_as_dst0);
        V is_ptr0 = dot_over([&](size_t i) { return is_ptr_as_dst0[i].first.v; }, [&](size_t i) { return is_ptr_as_dst0[i].second; }, is_ptr_as_dst0.size());
        ns.regs[idx].ptr = g.select(any_ptr_update_as_dst0, is_ptr0, ns.regs[idx].ptr);
        V is_ptr1 = g.dot({write_as_dst1.v}, {dst1.ptr});
        ns.regs[idx].ptr = g.select(write_as_dst1, is_ptr1, ns.regs[idx].ptr);
        for (auto& fv : value_as_dst0) ns.regs[idx].v = g.select_n(fv.first, fv.second, ns.regs[idx].v);
        ns.regs[idx].v = g.select_n(write_as_dst1, dst1.v, ns.regs[idx].v);
    }
    for (auto& c : df.new_pc_candidates) ns.cs.ctx.pc = g.select(c.first, c.second, ns.cs.ctx.pc);
    for (auto& c : df.new_ergs_left_candidates) ns.cs.ctx.ergs = g.select(c.first, c.second, ns.cs.ctx.ergs);
    for (auto& c : df.new_ergs_per_pubdata) ns.ergs_per_pubdata = g.select(c.first, c.second, ns.ergs_per_pubdata);
    for (auto& c : df.new_tx_number) ns.tx_number = g.select(c.first, c.second, ns.tx_number);
    ns.page_counter = df.memory_page_counters;
    for (auto& c : df.context_u128_candidates) ns.ctx_u128 = g.select_n(c.first, c.second, ns.ctx_u128);
    for (auto& c : df.new_heap_bounds) ns.cs.ctx.heap_bound = g.select(c.first, c.second, ns.cs.ctx.heap_bound);
    for (auto& c : df.new_aux_heap_bounds) ns.cs.ctx.aux_heap_bound = g.select(c.first, c.second, ns.cs.ctx.aux_heap_bound);
    for (auto& c : df.memory_queue_candidates) { ns.mem_len = g.select(c.flag, c.len, ns.mem_len); ns.mem_tail = g.select_n(c.flag, c.state, ns.mem_tail); }
    for (auto& c : df.decommitment_queue_candidates) { ns.dec_len = g.select(c.flag, c.len, ns.dec_len); ns.dec_tail = g.select_n(c.flag, c.state, ns.dec_tail); }
    for (auto& c : df.log_queue_forward_candidates) { ns.cs.fwd_len = g.select(c.flag, c.len, ns.cs.fwd_len); ns.cs.fwd_tail = g.select_n(c.flag, c.state, ns.cs.fwd_tail); }
    for (auto& c : df.log_queue_rollback_candidates) { ns.cs.ctx.rq_len = g.select(c.flag, c.len, ns.cs.ctx.rq_len); ns.cs.ctx.rq_head = g.select_n(c.flag, c.state, ns.cs.ctx.rq_head); }
    for (auto& c : df.flags) ns.flags = select(c.first, c.second, ns.flags);
    for (auto& c : df.callstacks) ns.cs = select(c.first, c.second, ns.cs);
    ns.pending_exception = g.multi_or(df.pending_exceptions).v;

    // conditional u32 range checks (cycle.rs:618-629)
    {
        W8 to_enforce = df.u32_conditional_range_checks.back().second;
        for (size_t i = 0; i + 1 < df.u32_conditional_range_checks.size(); ++i)
            to_enforce = g.select_n(df.u32_conditional_range_checks[i].first, df.u32_conditional_range_checks[i].second, to_enforce);
        for (auto x : to_enforce) g.range_check_u32(x);
    }
    // add/sub relation (MAX_ADD_SUB_RELATIONS_PER_CYCLE = 1) and mul/div relations (MAX = 3; every opcode contributes at most one)
    {
        AddSubRelation sel = df.add_sub_relations.back().second[0];
        for (size_t i = 0; i + 1 < df.add_sub_relations.size(); ++i) {
            const auto& [flag, v] = df.add_sub_relations[i];
            sel = AddSubRelation{g.select_n(flag, v[0].a, sel.a), g.select_n(flag, v[0].b, sel.b), g.select_n(flag, v[0].c, sel.c), g.select(flag, v[0].of, sel.of)};
        }
        enforce_addition_relation(sel);
        MulDivRelation msel = df.mul_div_relations.back().second[0];
        for (size_t i = 0; i + 1 < df.mul_div_relations.size(); ++i) {
            const auto& [flag, v] = df.mul_div_relations[i];
            msel = MulDivRelation{g.select_n(flag, v[0].a, msel.a), g.select_n(flag, v[0].b, msel.b), g.select_n(flag, v[0].rem, msel.rem),
                                  g.select_n(flag, v[0].mul_low, msel.mul_low), g.select_n(flag, v[0].mul_high, msel.mul_high)};
        }
        enforce_mul_relation(msel);
    }
    // sponges (cycle.rs:670-784): candidates are popped from the END of every opcode's list
    std::vector<Sponge> selected;
    auto pop_round = [&](Sponge* first_candidate) {
        bool have = first_candidate != nullptr;
        Sponge sel = have ? *first_candidate : Sponge{};
        for (auto& set : df.sponge_candidates_to_run) {
            if (set.sponges.empty()) continue;
            Sponge formal = set.sponges.back();
            set.sponges.pop_back();
            if (have) sel = select(set.applies, formal, sel);
            else { formal.flag = g.b_and(formal.flag, set.applies); sel = formal; have = true; }
        }
        if (!have) throw ZkError(ZK_ERR_INVALID, "main_vm: non-trivial sponge expected");
        selected.push_back(sel);
    };
    Sponge first = cr.src0_read_sponge, second = dst0_write_sponge;
    pop_round(&first);
    pop_round(&second);
    for (int i = 2; i < 8; ++i) pop_round(nullptr);
    for (auto& set : df.sponge_candidates_to_run)
        if (!set.sponges.empty()) throw ZkError(ZK_ERR_INVALID, "main_vm: more than MAX_SPONGES_PER_CYCLE sponges");
    for (auto& s : selected) {  // enforce_sponges — cycle.rs:937-957
        S12 true_final = g.compute_round_function(s.init);
        for (int i = 0; i < 12; ++i) g.cond_enforce_equal(s.flag, true_final[i], s.fin[i]);
    }
    return ns;
}

// initial_bootloader_state — src/main_vm/loading.rs:11-226
State VmCircuit::initial_bootloader_state(V mem_len, const S12& mem_tail, V dec_len, const S12& dec_tail, const S4& rollback_tail) {
    V zero = g.zero(), one = g.one();
    Ctx ctx = uninitialized_ctx();
    ctx.base_page = g.c(P(ZK_VMP_BOOTLOADER_BASE_PAGE));
    ctx.code_page = g.c(P(ZK_VMP_BOOTLOADER_CODE_PAGE));
    ctx.pc = zero;
    ctx.eh = g.c(P(ZK_VMP_INITIAL_FRAME_FORMAL_EH_LOCATION));
    ctx.ergs = g.c(P(ZK_VMP_VM_INITIAL_FRAME_ERGS));
    A5 bootloader_address = {g.c(P(ZK_VMP_BOOTLOADER_FORMAL_ADDRESS_LOW)), zero, zero, zero, zero};
    ctx.code_address = bootloader_address; ctx.this_ = bootloader_address;
    ctx.rq_tail = rollback_tail; ctx.rq_head = rollback_tail;
    ctx.is_kernel = one;
    ctx.heap_bound = g.c(P(ZK_VMP_BOOTLOADER_MAX_MEMORY)); ctx.aux_heap_bound = g.c(P(ZK_VMP_BOOTLOADER_MAX_MEMORY));
    Ctx empty_entry = uninitialized_ctx();
    empty_entry.rq_tail = rollback_tail; empty_entry.rq_head = rollback_tail; empty_entry.is_kernel = one;
    auto enc = encode_ctx(empty_entry);
    S12 current = g.empty_state();
    for (int r = 0; r < 4; ++r) {
        std::array<V, 8> e;
        for (int i = 0; i < 8; ++i) e[i] = enc[8 * r + i];
        current = g.compute_round_function(absorb8(e, current));
    }
    std::vector<V> z(STATE_WORDS, zero);
    State s = State::unflatten(z);
    s.mem_len = mem_len; s.mem_tail = mem_tail; s.dec_len = dec_len; s.dec_tail = dec_tail;
    s.cs.ctx = ctx; s.cs.depth = one; s.cs.sponge = current;
    s.timestamp = g.c(P(ZK_VMP_STARTING_TIMESTAMP));
    s.page_counter = g.c(P(ZK_VMP_STARTING_BASE_PAGE));
    // r1: formal empty fat pointer into the bootloader calldata page (FatPointer::to_u256: offset | page << 32 | start << 64 | length << 96)
    s.regs[0].ptr = one;
    s.regs[0].v[1] = g.c(P(ZK_VMP_BOOTLOADER_CALLDATA_PAGE));
    return s;
}

// main_vm_entry_point — src/main_vm/mod.rs:47-232
void VmCircuit::entry_point(uint32_t limit) {
    T_DECODE = cs.table_id(TABLE_VM_DECODE); T_COND = cs.table_id(TABLE_VM_CONDITIONAL); T_REGMASK = cs.table_id(TABLE_VM_REG_TO_BITMASK);
    T_SUBPC = cs.table_id(TABLE_VM_SUBPC_TO_BITMASK); T_UMASHIFT = cs.table_id(TABLE_VM_UMA_SHIFT_TO_BITMASK);
    T_UMACLEAN = cs.table_id(TABLE_VM_UMA_PTR_READ_CLEANUP); T_BITSHIFT = cs.table_id(TABLE_VM_BITSHIFT); T_BINOP = cs.table_id(TABLE_BINOP);
    cs.input_layout.clear();
    // ---- VmCircuitInputOutput::alloc_ignoring_outputs (src/fsm_input_output/mod.rs:73-98)
    Boolean start_flag = g.B(in_bool("start_flag"));
    S4 rollback_queue_tail_for_block = in_nums<4>("rollback_queue_tail_for_block");
    S12 memory_queue_initial_tail = in_nums<12>("memory_queue_initial_tail");
    V memory_queue_initial_length = in_u32("memory_queue_initial_length");
    S12 decommitment_queue_initial_tail = in_nums<12>("decommitment_queue_initial_tail");
    V decommitment_queue_initial_length = in_u32("decommitment_queue_initial_length");
    V zkporter_is_available = in_bool("zkporter_is_available");
    W8 default_aa_code_hash = in_u32s<8>("default_aa_code_hash");
    lay("hidden_fsm_input", STATE_WORDS);
    std::vector<V> fsm_in(STATE_WORDS);
    {
        auto kinds = state_word_kinds();
        for (size_t i = 0; i < STATE_WORDS; ++i) fsm_in[i] = alloc_kind(kinds[i]);
    }
    std::vector<V> observable_input(rollback_queue_tail_for_block.begin(), rollback_queue_tail_for_block.end());
    for (auto x : memory_queue_initial_tail) observable_input.push_back(x);
    observable_input.push_back(memory_queue_initial_length);
    for (auto x : decommitment_queue_initial_tail) observable_input.push_back(x);
    observable_input.push_back(decommitment_queue_initial_length);
    observable_input.push_back(zkporter_is_available);
    for (auto x : default_aa_code_hash) observable_input.push_back(x);

    State bootloader = initial_bootloader_state(memory_queue_initial_length, memory_queue_initial_tail, decommitment_queue_initial_length,
                                                decommitment_queue_initial_tail, rollback_queue_tail_for_block);
    std::vector<V> boot_flat = bootloader.flatten(), state0(STATE_WORDS);
    for (size_t i = 0; i < STATE_WORDS; ++i) state0[i] = g.select(start_flag, boot_flat[i], fsm_in[i]);

    // commitments that do not depend on the loop: side phase, overlapped with the loop kernel
    cs.side_begin();
    auto c_obs_in = g.commit_encoding(observable_input);
    auto c_fsm_in = g.commit_encoding(fsm_in);

    // ---- the cycles
    cs.loop_begin(limit);
    lay("state", STATE_WORDS);
    std::vector<V> in_flat(STATE_WORDS);
    for (size_t i = 0; i < STATE_WORDS; ++i) {
        in_flat[i] = g.next_input();
        cs.link(ZK_LINK_FIRST, in_flat[i], state0[i]);
    }
    gctx_zkporter = cs.loop_import(zkporter_is_available);
    for (int i = 0; i < 8; ++i) gctx_default_aa[i] = cs.loop_import(default_aa_code_hash[i]);
    State next = vm_cycle(State::unflatten(in_flat));
    std::vector<V> out_flat = next.flatten();
    for (size_t i = 0; i < STATE_WORDS; ++i) cs.link(ZK_LINK_CARRY, in_flat[i], out_flat[i]);
    cs.loop_end();

    // ---- epilogue (mod.rs:112-231)
    std::vector<V> fin(STATE_WORDS);
    for (size_t i = 0; i < STATE_WORDS; ++i) fin[i] = cs.loop_last(out_flat[i]);
    State final_state = State::unflatten(fin);
    Boolean done = g.is_zero(final_state.cs.depth);
    Boolean bootloader_exited_successfully = g.is_zero(final_state.cs.ctx.pc);
    g.conditionally_enforce_true(bootloader_exited_successfully, done);
    Boolean completion_flag = done;
    V zero = g.zero();
    for (int i = 0; i < 4; ++i) g.cond_enforce_equal(completion_flag, final_state.cs.fwd_tail[i], final_state.cs.ctx.rq_head[i]);
    // VmOutputData: log_queue_final_state (4+4+1), memory_queue_final_state (12+12+1), decommitment_queue_final_state (12+12+1);
    // heads are the placeholder zeros, tails selected against the empty state by the completion flag
    std::vector<V> observable_output;
    auto out_queue = [&](const V* tail, int n, V len) {
        for (int i = 0; i < n; ++i) observable_output.push_back(zero);
        for (int i = 0; i < n; ++i) observable_output.push_back(g.select(completion_flag, tail[i], zero));
        observable_output.push_back(g.select(completion_flag, len, zero));
    };
    out_queue(final_state.cs.fwd_tail.data(), 4, final_state.cs.fwd_len);
    out_queue(final_state.mem_tail.data(), 12, final_state.mem_len);
    out_queue(final_state.dec_tail.data(), 12, final_state.dec_len);

    // ClosedFormInputCompactForm::from_full_form (src/fsm_input_output/mod.rs:178-253)
    auto c_obs_out = g.commit_encoding(observable_output);
    auto c_fsm_out = g.commit_encoding(fin);
    Num zero_num = g.num_const(0);
    std::vector<V> compact = {start_flag.v, completion_flag.v};
    for (int i = 0; i < 4; ++i) compact.push_back(c_obs_in[i].v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completion_flag, c_obs_out[i], zero_num).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(start_flag, zero_num, c_fsm_in[i]).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completion_flag, zero_num, c_fsm_out[i]).v);
    auto input_commitment = g.commit_encoding(compact);
    for (auto& el : input_commitment) cs.place_gate(ZK_GATE_PUBLIC_INPUT, &el.v, 1, nullptr, 0);
    // structured_input.hook_compare_witness (mod.rs:218): the groups a host compares with its own closed-form input
    cs.hooks["hidden_fsm_output"] = fin;
    cs.hooks["observable_output"] = observable_output;
    cs.native_seed_kind = 1;  // the carried VmLocalState has a native walker + chain kernels (vm_native.hpp, kernels_vm_seed.hpp)
}

}  // namespace

// geometry check + gate set + tables (src/main_vm/cycle.rs:959-966; tables: src/tables/*.rs)
void main_vm_configure(CS& cs, const zk_opcode_defs& d, uint32_t flags) {
    if (d.type_bits != ZK_VMF__COUNT || d.variant_bits > 16 || d.flag_bits > 4 || d.src_mode_bits != ZK_VMM__COUNT || d.dst_mode_bits != 4 ||
        d.description_bits_flattened % 8 || d.description_bits_flattened + d.aux_bits > 56 || d.aux_bits != 3 ||
        d.type_bits + d.variant_bits + d.flag_bits + d.src_mode_bits + d.dst_mode_bits > d.description_bits_flattened)
        throw ZkError(ZK_ERR_INVALID, "main_vm_configure: opcode-defs blob has an unsupported shape");
    cs.allow_lookup(3, 8, true);
    for (uint32_t k = 1; k < ZK_GATE__COUNT; ++k)
        if (!(k == ZK_GATE_U8X4_FMA && (flags & ZK_VM_CFG_U32_FMA_ROLE))) cs.allow_gate(k);
    add_xor8_table(cs);
    add_binop_table(cs);
    {   // create_opcodes_decoding_and_pricing_table — src/tables/opcodes_decoding.rs:14-38: [opcode, price, properties]
        std::vector<uint64_t> rows;
        for (uint64_t x = 0; x < ZK_VM_OPCODE_TABLE_ROWS; ++x) { rows.push_back(x); rows.push_back(d.prices[x]); rows.push_back(d.props[x]); }
        cs.add_table(TABLE_VM_DECODE, 1, 2, rows.data(), ZK_VM_OPCODE_TABLE_ROWS);
    }
    {   // create_conditionals_resolution_table — src/tables/conditional.rs:21-58: [condition, flags(of | eq << 1 | gt << 2), resolution]
        std::vector<uint64_t> rows;
        for (int cnd = 0; cnd < ZK_VMC__COUNT; ++cnd)
            for (uint64_t i = 0; i < 8; ++i) {
                const bool of = i & 1, eq = i & 2, gt = i & 4;
                bool res = false;
                switch (cnd) {
                case ZK_VMC_ALWAYS: res = true; break;
                case ZK_VMC_LT: res = of; break;
                case ZK_VMC_EQ: res = eq; break;
                case ZK_VMC_GT: res = gt; break;
                case ZK_VMC_GE: res = gt || eq; break;
                case ZK_VMC_LE: res = of || eq; break;
                case ZK_VMC_NE: res = !eq; break;
                case ZK_VMC_GT_OR_LT: res = gt || of; break;
                }
                rows.push_back(d.condition_idx[cnd]); rows.push_back(i); rows.push_back(res);
            }
        cs.add_table(TABLE_VM_CONDITIONAL, 2, 1, rows.data(), 64);
    }
    auto int_to_bitmask = [&](uint32_t marker, int num_bits) {  // create_integer_to_bitmask_table — integer_to_boolean_mask.rs:21-45
        std::vector<uint64_t> rows;
        for (uint64_t a = 0; a < (1ull << num_bits); ++a) { rows.push_back(a); rows.push_back(a == 0 ? 0 : 1ull << (a - 1)); rows.push_back(0); }
        cs.add_table(marker, 1, 2, rows.data(), 1u << num_bits);
    };
    int_to_bitmask(TABLE_VM_REG_TO_BITMASK, 4);   // REGISTER_ENCODING_BITS
    int_to_bitmask(TABLE_VM_SUBPC_TO_BITMASK, 2); // create_subpc_bitmask_table :68-70
    {   // create_integer_set_ith_bit_table(5) — integer_to_boolean_mask.rs:47-66 (UMAShiftToBitmaskTable: 32 unalignments)
        std::vector<uint64_t> rows;
        for (uint64_t a = 0; a < 32; ++a) { rows.push_back(a); rows.push_back(1ull << a); rows.push_back(0); }
        cs.add_table(TABLE_VM_UMA_SHIFT_TO_BITMASK, 1, 2, rows.data(), 32);
    }
    {   // create_uma_ptr_read_bitmask_table — src/tables/uma_ptr_read_cleanup.rs:11-40
        std::vector<uint64_t> rows;
        const uint64_t FULL = (1ull << 32) - 1;
        for (uint64_t a = 0; a < 32; ++a) { rows.push_back(a); rows.push_back(a == 0 ? FULL : FULL - ((1ull << a) - 1)); rows.push_back(0); }
        cs.add_table(TABLE_VM_UMA_PTR_READ_CLEANUP, 1, 2, rows.data(), 32);
    }
    {   // create_shift_to_num_converter_table — src/tables/bitshift.rs:12-40: key = shift + (idx << 8), two 32-bit limbs of 1 << shift per row
        std::vector<uint64_t> rows;
        for (uint64_t shift = 0; shift < 256; ++shift)
            for (uint64_t idx = 0; idx < 4; ++idx) {
                auto limb = [&](uint64_t l) -> uint64_t { return shift / 32 == l ? 1ull << (shift % 32) : 0; };
                rows.push_back(shift + (idx << 8)); rows.push_back(limb(2 * idx)); rows.push_back(limb(2 * idx + 1));
            }
        cs.add_table(TABLE_VM_BITSHIFT, 1, 2, rows.data(), 1024);
    }
    cs.circuit_blob.assign((const uint8_t*)&d, (const uint8_t*)&d + sizeof d);
}

void main_vm_entry_point(CS& cs, uint32_t limit) {
    if (cs.circuit_blob.size() != sizeof(zk_opcode_defs)) throw ZkError(ZK_ERR_INVALID, "main_vm: call zk_circuit_main_vm_configure first");
    static thread_local zk_opcode_defs defs;
    std::memcpy(&defs, cs.circuit_blob.data(), sizeof defs);
    VmCircuit vm(cs, defs);
    vm.entry_point(limit);
}

}  // namespace zkgl
