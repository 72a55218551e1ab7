// circuits/ram_permutation.cpp — host-side mirror of
// /root/reference/src/ram_permutation/mod.rs (ram_permutation_entry_point :31-210,
// partial_accumulate_inner :212-382, long_equals :384-392) recorded against the zkgl CS.
//
// The reference unrolls `for _cycle in 0..limit`; here the body is recorded ONCE in the loop
// scope.  Loop-carried state (queue heads/lengths, grand-product accumulators, previous keys,
// the non-deterministic-write counter, the `_cycle == 0` flag) enters every iteration from the
// per-iteration input stream and is tied to the previous iteration's outputs by CARRY links
// (copy constraints), which is what lets all iterations run as independent GPU lanes.
//
// INPUT STREAM LAYOUT:
//   outer scope, per instance (121 words, allocation order of alloc_ignoring_outputs,
//   src/fsm_input_output/mod.rs:73-98):
//     [0] start_flag
//     [1..26) unsorted_queue_initial_state {head[12], tail[12], length}
//     [26..51) sorted_queue_initial_state
//     [51] non_deterministic_bootloader_memory_snapshot_length
//     [52..56) lhs_accumulator[2], rhs_accumulator[2]
//     [56..81) current_unsorted_queue_state   [81..106) current_sorted_queue_state
//     [106..109) previous_sorting_key  [109..111) previous_full_key  [111..119) previous_value
//     [119] previous_is_ptr   [120] num_nondeterministic_writes
//   loop scope, per iteration (72 words):
//     [0] is_first_cycle
//     [1..13) unsorted head   [13] unsorted length   [14..26) sorted head   [26] sorted length
//     [27..29) lhs  [29..31) rhs  [31] num_nondeterministic_writes
//     [32..35) previous_sorting_key  [35..37) previous_full_key  [37..45) previous_value  [45] previous_is_ptr
//     [46..59) unsorted item {ts, page, index, rw, is_ptr, value[8]}   [59..72) sorted item
#include "../gadgets.hpp"

namespace zkgl {

namespace {

constexpr uint32_t BOOTLOADER_HEAP_PAGE = 10;  // zkevm_opcode_defs [EXT]: heap_page_from_base(BOOTLOADER_BASE_PAGE = 8)
constexpr int REPS = 2;                        // DEFAULT_NUM_PERMUTATION_ARGUMENT_REPETITIONS (src/lib.rs:39)
constexpr int ENC = 8;                         // MEMORY_QUERY_PACKED_WIDTH

struct MemoryQuery {
    UInt32 timestamp, memory_page, index;
    Boolean rw_flag, is_ptr;
    UInt256 value;
};

MemoryQuery allocate_memory_query(G& g) {  // CSAllocatable derive: every field's own `allocate`
    MemoryQuery q;
    q.timestamp = g.alloc_u32_checked();
    q.memory_page = g.alloc_u32_checked();
    q.index = g.alloc_u32_checked();
    q.rw_flag = g.alloc_bool();
    q.is_ptr = g.alloc_bool();
    q.value = g.alloc_u256_checked();
    return q;
}

// MemoryQuery::encode — src/base_structures/memory_query/mod.rs:103-221
std::array<zk_var, ENC> encode_memory_query(G& g, const MemoryQuery& q) {
    const uint64_t S32 = 1ull << 32, S33 = 1ull << 33, S40 = 1ull << 40, S48 = 1ull << 48;
    zk_var v0 = q.timestamp.v, v1 = q.memory_page.v;
    zk_var v2 = g.linear_combination({{q.index.v, 1}, {q.rw_flag.v, S32}, {q.is_ptr.v, S33}});
    auto d5 = g.decompose_into_bytes(q.value.inner[5]);
    auto d6 = g.decompose_into_bytes(q.value.inner[6]);
    auto d7 = g.decompose_into_bytes(q.value.inner[7]);
    zk_var v3 = g.linear_combination({{q.value.inner[0].v, 1}, {d5[0].v, S32}, {d5[1].v, S40}, {d5[2].v, S48}});
    zk_var v4 = g.linear_combination({{q.value.inner[1].v, 1}, {d5[3].v, S32}, {d6[0].v, S40}, {d6[1].v, S48}});
    zk_var v5 = g.linear_combination({{q.value.inner[2].v, 1}, {d6[2].v, S32}, {d6[3].v, S40}, {d7[0].v, S48}});
    zk_var v6 = g.linear_combination({{q.value.inner[3].v, 1}, {d7[1].v, S32}, {d7[2].v, S40}, {d7[3].v, S48}});
    zk_var v7 = q.value.inner[4].v;
    return {v0, v1, v2, v3, v4, v5, v6, v7};
}

// unpacked_long_comparison — src/storage_validity_by_grand_product/mod.rs:925-944
template <size_t N>
std::pair<Boolean, Boolean> unpacked_long_comparison(G& g, const std::array<UInt32, N>& a, const std::array<UInt32, N>& b) {
    Boolean borrow = g.bool_const(false);
    std::vector<Boolean> equals;
    for (size_t i = 0; i < N; ++i) {
        auto [diff, nb] = g.overflowing_sub_with_borrow_in(b[i], a[i], borrow);
        borrow = nb;
        equals.push_back(g.is_zero(diff.v));
    }
    return {g.multi_and(equals), borrow};
}

// long_equals — src/ram_permutation/mod.rs:384-392
template <size_t N>
Boolean long_equals(G& g, const std::array<UInt32, N>& a, const std::array<UInt32, N>& b) {
    std::vector<Boolean> eq;
    for (size_t i = 0; i < N; ++i) eq.push_back(g.equals(a[i].v, b[i].v));
    return g.multi_and(eq);
}

// accumulate_grand_products — src/utils.rs:81-137
void accumulate_grand_products(G& g, std::array<Num, REPS>& lhs, std::array<Num, REPS>& rhs,
                               const std::array<std::array<zk_var, ENC + 1>, REPS>& ch,
                               const std::array<zk_var, ENC>& lhs_enc, const std::array<zk_var, ENC>& rhs_enc,
                               Boolean should_accumulate) {
    for (int r = 0; r < REPS; ++r) {
        zk_var lc = ch[r][ENC], rc = ch[r][ENC];
        for (int i = 0; i < ENC; ++i) {
            lc = g.fma(1, lhs_enc[i], ch[r][i], 1, lc);  // Num::fma(el, challenge, 1, contribution, 1)
            rc = g.fma(1, rhs_enc[i], ch[r][i], 1, rc);
        }
        zk_var new_lhs = g.mul(lhs[r].v, lc), new_rhs = g.mul(rhs[r].v, rc);
        lhs[r] = g.select(should_accumulate, Num{new_lhs}, lhs[r]);
        rhs[r] = g.select(should_accumulate, Num{new_rhs}, rhs[r]);
    }
}

// Num::conditionally_enforce_equal: cond * (a - b) == 0
void conditionally_enforce_equal(G& g, Boolean cond, zk_var a, zk_var b) {
    zk_var d = g.sub(a, b);
    g.enforce_zero(g.mul(cond.v, d));
}

}  // namespace

// The CS the reference test builds: src/ram_permutation/mod.rs:419-501
void ram_permutation_configure(CS& cs) {
    cs.allow_lookup(3, 8, true);
    for (uint32_t k : {ZK_GATE_CONST, ZK_GATE_FMA, ZK_GATE_REDUCTION4, ZK_GATE_BOOLEAN, ZK_GATE_UINTX_ADD, ZK_GATE_SELECT,
                       ZK_GATE_ZEROCHECK, ZK_GATE_DOT4, ZK_GATE_MATMUL12_EXT, ZK_GATE_MATMUL12_INT, ZK_GATE_NOP,
                       ZK_GATE_PUBLIC_INPUT})
        cs.allow_gate(k);
    add_xor8_table(cs);
}

void ram_permutation_entry_point(CS& cs, uint32_t limit) {
    G g(cs);
    // ---- alloc_ignoring_outputs (mod.rs:50-51) ----
    Boolean start_flag = g.alloc_bool();
    auto obs_unsorted = g.alloc_queue_state<12>();
    auto obs_sorted = g.alloc_queue_state<12>();
    UInt32 obs_nondet_len = g.alloc_u32_checked();
    std::array<Num, REPS> fsm_lhs, fsm_rhs;
    for (auto& x : fsm_lhs) x = g.alloc_num();
    for (auto& x : fsm_rhs) x = g.alloc_num();
    auto fsm_unsorted = g.alloc_queue_state<12>();
    auto fsm_sorted = g.alloc_queue_state<12>();
    std::array<UInt32, 3> fsm_prev_sorting_key;
    for (auto& x : fsm_prev_sorting_key) x = g.alloc_u32_checked();
    std::array<UInt32, 2> fsm_prev_full_key;
    for (auto& x : fsm_prev_full_key) x = g.alloc_u32_checked();
    UInt256 fsm_prev_value = g.alloc_u256_checked();
    Boolean fsm_prev_is_ptr = g.alloc_bool();
    UInt32 fsm_nondet = g.alloc_u32_checked();

    // passthrough must be trivial (mod.rs:58-60, 86-88)
    g.enforce_trivial_head(obs_unsorted);
    g.enforce_trivial_head(obs_sorted);
    auto unsorted_state = g.select(start_flag, obs_unsorted, fsm_unsorted);  // mod.rs:62-67
    auto sorted_state = g.select(start_flag, obs_sorted, fsm_sorted);

    // ---- produce_fs_challenges (mod.rs:111-116 -> src/utils.rs:12-78) ----
    std::array<std::array<zk_var, ENC + 1>, REPS> challenges;
    {
        std::vector<zk_var> fs_input;
        for (auto& t : obs_unsorted.tail) fs_input.push_back(t.v);
        fs_input.push_back(obs_unsorted.length.v);
        for (auto& t : obs_sorted.tail) fs_input.push_back(t.v);
        fs_input.push_back(obs_sorted.length.v);
        std::array<zk_var, 12> state = g.empty_state();
        state[11] = g.constant(fs_input.size());  // apply_length_specialization
        size_t nchunks = (fs_input.size() + 7) / 8;
        for (size_t c = 0; c < nchunks; ++c) {
            for (size_t j = 0; j < 8; ++j) {
                size_t k = 8 * c + j;
                state[j] = k < fs_input.size() ? fs_input[k] : g.zero();
            }
            state = g.compute_round_function(state);
        }
        int can_take = 8;
        for (int r = 0; r < REPS; ++r) {
            challenges[r][0] = g.one();
            for (int i = 1; i < ENC + 1; ++i) {
                if (can_take == 0) { state = g.compute_round_function(state); can_take = 8; }
                challenges[r][i] = state[8 - can_take];
                --can_take;
            }
        }
    }

    // native seeding (kernels_queue_seed.hpp): with the queue heads given by the host (the witness's previous tails) the rest of the
    // carried state is scans over the cycles; the kernel reads the challenges from the outer store
    cs.native_seed_kind = 2;
    cs.native_seed_param = BOOTLOADER_HEAP_PAGE;
    cs.native_seed_outer_vars.clear();
    for (int r = 0; r < REPS; ++r)
        for (int i = 1; i <= ENC; ++i) cs.native_seed_outer_vars.push_back(challenges[r][i]);
    Num num_one = g.num_const(1);
    std::array<Num, REPS> lhs0, rhs0;
    for (int r = 0; r < REPS; ++r) {  // mod.rs:118-130
        lhs0[r] = g.select(start_flag, num_one, fsm_lhs[r]);
        rhs0[r] = g.select(start_flag, num_one, fsm_rhs[r]);
    }
    UInt32 nondet0 = g.select(start_flag, g.u32_const(0), fsm_nondet);  // mod.rs:132-138
    Boolean not_start = g.negated(start_flag);
    // partial_accumulate_inner prologue (mod.rs:238-243)
    g.enforce_equal(unsorted_state.length.v, sorted_state.length.v);
    zk_var outer_one = g.one();

    // commitments of observable_input / hidden_fsm_input (from_full_form, src/fsm_input_output/mod.rs:196-201) do not
    // depend on the loop: side phase, overlapped with the loop kernel
    cs.side_begin();
    std::vector<zk_var> obs_in;
    for (auto v : g.flatten(obs_unsorted)) obs_in.push_back(v);
    for (auto v : g.flatten(obs_sorted)) obs_in.push_back(v);
    obs_in.push_back(obs_nondet_len.v);
    std::vector<zk_var> fsm_in;
    for (auto& x : fsm_lhs) fsm_in.push_back(x.v);
    for (auto& x : fsm_rhs) fsm_in.push_back(x.v);
    for (auto v : g.flatten(fsm_unsorted)) fsm_in.push_back(v);
    for (auto v : g.flatten(fsm_sorted)) fsm_in.push_back(v);
    for (auto& x : fsm_prev_sorting_key) fsm_in.push_back(x.v);
    for (auto& x : fsm_prev_full_key) fsm_in.push_back(x.v);
    for (auto& x : fsm_prev_value.inner) fsm_in.push_back(x.v);
    fsm_in.push_back(fsm_prev_is_ptr.v);
    fsm_in.push_back(fsm_nondet.v);
    auto c_obs_in = g.commit_encoding(obs_in);
    auto c_fsm_in = g.commit_encoding(fsm_in);

    // =========================== loop body (mod.rs:246-381), recorded once ===========================
    cs.loop_begin(limit);
    auto carry_in = [&](zk_var init_outer) {  // state entering the iteration
        zk_var v = g.next_input();
        cs.link(ZK_LINK_FIRST, v, init_outer);
        return v;
    };
    Boolean is_first{carry_in(outer_one)};
    std::array<zk_var, 12> u_head, s_head;
    for (int i = 0; i < 12; ++i) u_head[i] = carry_in(unsorted_state.head[i].v);
    UInt32 u_len{carry_in(unsorted_state.length.v)};
    for (int i = 0; i < 12; ++i) s_head[i] = carry_in(sorted_state.head[i].v);
    UInt32 s_len{carry_in(sorted_state.length.v)};
    std::array<Num, REPS> lhs, rhs;
    for (int r = 0; r < REPS; ++r) lhs[r] = Num{carry_in(lhs0[r].v)};
    for (int r = 0; r < REPS; ++r) rhs[r] = Num{carry_in(rhs0[r].v)};
    UInt32 nondet{carry_in(nondet0.v)};
    std::array<UInt32, 3> prev_sorting_key;
    for (int i = 0; i < 3; ++i) prev_sorting_key[i] = UInt32{carry_in(fsm_prev_sorting_key[i].v)};
    std::array<UInt32, 2> prev_full_key;
    for (int i = 0; i < 2; ++i) prev_full_key[i] = UInt32{carry_in(fsm_prev_full_key[i].v)};
    UInt256 prev_value;
    for (int i = 0; i < 8; ++i) prev_value.inner[i] = UInt32{carry_in(fsm_prev_value.inner[i].v)};
    Boolean prev_is_ptr{carry_in(fsm_prev_is_ptr.v)};
    const std::array<zk_var, 46> state_in = [&] {
        std::array<zk_var, 46> a{};
        int n = 0;
        a[n++] = is_first.v;
        for (auto v : u_head) a[n++] = v;
        a[n++] = u_len.v;
        for (auto v : s_head) a[n++] = v;
        a[n++] = s_len.v;
        for (auto& x : lhs) a[n++] = x.v;
        for (auto& x : rhs) a[n++] = x.v;
        a[n++] = nondet.v;
        for (auto& x : prev_sorting_key) a[n++] = x.v;
        for (auto& x : prev_full_key) a[n++] = x.v;
        for (auto& x : prev_value.inner) a[n++] = x.v;
        a[n++] = prev_is_ptr.v;
        return a;
    }();

    // loop-invariant values computed by the outer scope
    std::array<std::array<zk_var, ENC + 1>, REPS> ch;
    for (int r = 0; r < REPS; ++r)
        for (int i = 0; i <= ENC; ++i) ch[r][i] = (i == 0) ? g.one() : cs.loop_import(challenges[r][i]);
    Boolean is_start{cs.loop_import(start_flag.v)};
    // `_cycle == 0` specialisation (mod.rs:308-313, 335-357) folded into effective flags
    Boolean is_start_eff = g.b_and(is_start, is_first);
    Boolean not_start_eff = g.negated(is_start_eff);
    UInt32 bootloader_heap_page = g.u32_const(BOOTLOADER_HEAP_PAGE);
    UInt256 uint256_zero = g.u256_zero();

    Boolean unsorted_is_empty = g.is_zero(u_len.v);
    Boolean sorted_is_empty = g.is_zero(s_len.v);
    g.enforce_bool_equal(unsorted_is_empty, sorted_is_empty);
    Boolean can_pop = g.negated(unsorted_is_empty);

    // FullStateCircuitQueue::pop_front (boojum [EXT]; push rule pinned by src/main_vm/utils.rs:194-213)
    auto pop_front = [&](std::array<zk_var, 12>& head, UInt32& len, MemoryQuery& item) {
        item = allocate_memory_query(g);
        auto enc = encode_memory_query(g, item);
        std::array<zk_var, 12> st;
        for (int i = 0; i < 8; ++i) st[i] = enc[i];
        for (int i = 8; i < 12; ++i) st[i] = head[i];
        auto nh = g.compute_round_function(st);
        for (int i = 0; i < 12; ++i) head[i] = g.select(can_pop, nh[i], head[i]);
        UInt32 dec{g.sub(len.v, g.one())};
        len = g.select(can_pop, dec, len);
        return enc;
    };
    MemoryQuery unsorted_item, sorted_item;
    auto unsorted_enc = pop_front(u_head, u_len, unsorted_item);
    auto sorted_enc = pop_front(s_head, s_len, sorted_item);

    {  // non-deterministic writes (mod.rs:259-290)
        Boolean ts_is_zero = g.is_zero(sorted_item.timestamp.v);
        Boolean page_is_bootloader_heap = g.equals(sorted_item.memory_page.v, bootloader_heap_page.v);
        Boolean not_ptr = g.negated(sorted_item.is_ptr);
        Boolean is_nondet_write = g.multi_and({can_pop, ts_is_zero, page_is_bootloader_heap, sorted_item.rw_flag, not_ptr});
        UInt32 inc = g.increment_unchecked(nondet);
        nondet = g.select(is_nondet_write, inc, nondet);
    }
    {  // RAM ordering (mod.rs:292-364)
        std::array<UInt32, 3> sorting_key = {sorted_item.timestamp, sorted_item.index, sorted_item.memory_page};
        std::array<UInt32, 2> comparison_key = {sorted_item.index, sorted_item.memory_page};
        auto [keys_equal, previous_key_is_smaller] = unpacked_long_comparison(g, sorting_key, prev_sorting_key);
        (void)keys_equal;
        Boolean should_enforce_order = g.b_and(can_pop, not_start_eff);
        g.conditionally_enforce_true(previous_key_is_smaller, should_enforce_order);

        Boolean same_memory_cell = long_equals(g, comparison_key, prev_full_key);
        Boolean value_equal = g.equals(sorted_item.value, prev_value);
        Boolean not_same_cell = g.negated(same_memory_cell);
        Boolean rw_flag = sorted_item.rw_flag;
        Boolean not_rw_flag = g.negated(rw_flag);
        Boolean value_is_zero = g.equals(sorted_item.value, uint256_zero);
        Boolean not_ptr = g.negated(sorted_item.is_ptr);
        Boolean is_zero = g.b_and(value_is_zero, not_ptr);
        Boolean ptr_equality = g.equals(prev_is_ptr.v, sorted_item.is_ptr.v);
        Boolean value_and_ptr_equal = g.b_and(value_equal, ptr_equality);

        Boolean read_uninit_if_continue = g.multi_and({not_start_eff, not_same_cell, not_rw_flag});
        Boolean read_uninit_at_start = g.b_and(is_start_eff, not_rw_flag);
        Boolean should_enforce = g.b_or(read_uninit_if_continue, read_uninit_at_start);
        g.conditionally_enforce_true(is_zero, should_enforce);
        Boolean check_equality = g.multi_and({same_memory_cell, not_rw_flag, not_start_eff});
        g.conditionally_enforce_true(value_and_ptr_equal, check_equality);

        prev_sorting_key = sorting_key;
        prev_full_key = comparison_key;
        prev_value = sorted_item.value;
        prev_is_ptr = sorted_item.is_ptr;
    }
    accumulate_grand_products(g, lhs, rhs, ch, unsorted_enc, sorted_enc, can_pop);

    // state leaving the iteration, in the same order as state_in
    std::array<zk_var, 46> state_out{};
    {
        int n = 0;
        state_out[n++] = g.zero();  // is_first of the next cycle
        for (auto v : u_head) state_out[n++] = v;
        state_out[n++] = u_len.v;
        for (auto v : s_head) state_out[n++] = v;
        state_out[n++] = s_len.v;
        for (auto& x : lhs) state_out[n++] = x.v;
        for (auto& x : rhs) state_out[n++] = x.v;
        state_out[n++] = nondet.v;
        for (auto& x : prev_sorting_key) state_out[n++] = x.v;
        for (auto& x : prev_full_key) state_out[n++] = x.v;
        for (auto& x : prev_value.inner) state_out[n++] = x.v;
        state_out[n++] = prev_is_ptr.v;
    }
    for (int i = 0; i < 46; ++i) cs.link(ZK_LINK_CARRY, state_in[i], state_out[i]);
    cs.loop_end();
    // =========================== epilogue (mod.rs:161-209) ===========================
    std::array<zk_var, 46> fin;
    for (int i = 0; i < 46; ++i) fin[i] = cs.loop_last(state_out[i]);
    QueueState<12> unsorted_final = unsorted_state, sorted_final = sorted_state;
    {
        int n = 1;
        for (int i = 0; i < 12; ++i) unsorted_final.head[i] = Num{fin[n++]};
        unsorted_final.length = UInt32{fin[n++]};
        for (int i = 0; i < 12; ++i) sorted_final.head[i] = Num{fin[n++]};
        sorted_final.length = UInt32{fin[n++]};
    }
    std::array<Num, REPS> lhs_f = {Num{fin[27]}, Num{fin[28]}}, rhs_f = {Num{fin[29]}, Num{fin[30]}};
    UInt32 nondet_f{fin[31]};

    // enforce_consistency: an empty queue has head == tail
    auto enforce_consistency = [&](const QueueState<12>& q) {
        Boolean is_empty = g.is_zero(q.length.v);
        for (int i = 0; i < 12; ++i) conditionally_enforce_equal(g, is_empty, q.head[i].v, q.tail[i].v);
    };
    enforce_consistency(unsorted_final);
    enforce_consistency(sorted_final);
    Boolean completed = g.is_zero(unsorted_final.length.v);
    for (int r = 0; r < REPS; ++r) conditionally_enforce_equal(g, completed, lhs_f[r].v, rhs_f[r].v);
    Boolean nondet_equal = g.equals(nondet_f.v, obs_nondet_len.v);
    g.conditionally_enforce_true(nondet_equal, completed);

    // hidden_fsm_output in RamPermutationFSMInputOutput field order (src/ram_permutation/input.rs:49-62)
    std::vector<zk_var> fsm_out;
    for (auto& x : lhs_f) fsm_out.push_back(x.v);
    for (auto& x : rhs_f) fsm_out.push_back(x.v);
    for (auto v : g.flatten(unsorted_final)) fsm_out.push_back(v);
    for (auto v : g.flatten(sorted_final)) fsm_out.push_back(v);
    for (int i = 32; i < 46; ++i) fsm_out.push_back(fin[i]);  // prev sorting key, full key, value, is_ptr
    fsm_out.push_back(nondet_f.v);

    // ClosedFormInputCompactForm::from_full_form (src/fsm_input_output/mod.rs:178-253); c_obs_in / c_fsm_in: side phase
    auto c_obs_out = g.commit_encoding({});  // observable_output = ()
    cs.hooks["hidden_fsm_output"] = fsm_out;  // zk_cs_hook_compare_witness (src/fsm_input_output/mod.rs:102-133)
    auto c_fsm_out = g.commit_encoding(fsm_out);
    Num zero_num = g.num_const(0);
    std::vector<zk_var> compact = {start_flag.v, completed.v};
    for (int i = 0; i < 4; ++i) compact.push_back(c_obs_in[i].v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completed, c_obs_out[i], zero_num).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(start_flag, zero_num, c_fsm_in[i]).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completed, zero_num, c_fsm_out[i]).v);
    auto input_commitment = g.commit_encoding(compact);
    for (auto& el : input_commitment) cs.place_gate(ZK_GATE_PUBLIC_INPUT, &el.v, 1, nullptr, 0);
    (void)not_start;
}

}  // namespace zkgl
