// circuits/log_sorter.cpp — host-side mirror of /root/reference/src/log_sorter/mod.rs:
//   sort_and_deduplicate_events_entry_point :34-232
//   repack_and_prove_events_rollbacks_inner :234-441 (loop body :283-403, finalisation :405-436)
// (events / L2->L1 message sorter with rollback collapse), loop body recorded once.
//
// INPUT STREAMS
//   outer (87 words): start_flag | initial_log_queue_state[9], intermediate_sorted_queue_state[9]
//     | hidden_fsm_input in EventsDeduplicatorFSMInputOutput order (input.rs:28-36): lhs[2], rhs[2],
//       initial_unsorted_queue_state[9], intermediate_sorted_queue_state[9], final_result_queue_state[9],
//       previous_key, previous_item[36]
//   loop (129 words): carried[57] = previous_is_trivial, lhs[2], rhs[2], unsorted head[4]+len,
//       sorted head[4]+len, result tail[4]+len, previous_key, previous_item[36]
//     | unsorted LogQuery[36] | sorted LogQuery[36]
#include "log_query.hpp"

namespace zkgl {

namespace {
constexpr size_t ENC = 20;  // LOG_QUERY_PACKED_WIDTH

std::vector<zk_var> flatten_query(const LogQuery& q) {  // flatten_as_variables_impl order
    std::vector<zk_var> o;
    for (auto& l : q.address) o.push_back(l.v);
    for (auto& l : q.key.inner) o.push_back(l.v);
    for (auto& l : q.read_value.inner) o.push_back(l.v);
    for (auto& l : q.written_value.inner) o.push_back(l.v);
    o.push_back(q.aux_byte.v); o.push_back(q.rw_flag.v); o.push_back(q.rollback.v); o.push_back(q.is_service.v);
    o.push_back(q.shard_id.v); o.push_back(q.tx_number_in_block.v); o.push_back(q.timestamp.v);
    return o;
}
LogQuery unflatten_query(const std::vector<zk_var>& f, size_t off) {
    LogQuery q;
    size_t n = off;
    for (auto& l : q.address) l = UInt32{f[n++]};
    for (auto& l : q.key.inner) l = UInt32{f[n++]};
    for (auto& l : q.read_value.inner) l = UInt32{f[n++]};
    for (auto& l : q.written_value.inner) l = UInt32{f[n++]};
    q.aux_byte = UInt8{f[n++]}; q.rw_flag = Boolean{f[n++]}; q.rollback = Boolean{f[n++]}; q.is_service = Boolean{f[n++]};
    q.shard_id = UInt8{f[n++]}; q.tx_number_in_block = UInt32{f[n++]}; q.timestamp = UInt32{f[n++]};
    return q;
}
// the cleaned-up record pushed to the result queue (mod.rs:372-386, 419-433)
LogQuery query_to_add(G& g, const LogQuery& prev) {
    LogQuery q;
    q.address = prev.address; q.key = prev.key; q.read_value = g.u256_zero(); q.written_value = prev.written_value;
    q.rw_flag = g.bool_const(false); q.aux_byte = UInt8{g.zero()}; q.rollback = g.bool_const(false);
    q.is_service = prev.is_service; q.shard_id = prev.shard_id; q.tx_number_in_block = prev.tx_number_in_block;
    q.timestamp = g.u32_const(0);
    return q;
}
}  // namespace

void log_sorter_configure(CS& cs) {  // the reference test's CS: mod.rs:495-583
    cs.allow_lookup(3, 8, true);
    for (uint32_t k : {ZK_GATE_CONST, ZK_GATE_FMA, ZK_GATE_REDUCTION4, ZK_GATE_BOOLEAN, ZK_GATE_UINTX_ADD, ZK_GATE_SELECT,
                       ZK_GATE_ZEROCHECK, ZK_GATE_DOT4, ZK_GATE_MATMUL12_EXT, ZK_GATE_MATMUL12_INT, ZK_GATE_NOP,
                       ZK_GATE_PUBLIC_INPUT})
        cs.allow_gate(k);
    add_xor8_table(cs);
}

void sort_and_deduplicate_events_entry_point(CS& cs, uint32_t limit) {
    G g(cs);
    Boolean start_flag = g.alloc_bool();
    Queue4 obs_unsorted = alloc_queue4(g), obs_sorted = alloc_queue4(g);
    std::array<Num, 2> fsm_lhs, fsm_rhs;
    for (auto& x : fsm_lhs) x = g.alloc_num();
    for (auto& x : fsm_rhs) x = g.alloc_num();
    Queue4 fsm_unsorted = alloc_queue4(g), fsm_sorted = alloc_queue4(g), fsm_result = alloc_queue4(g);
    UInt32 fsm_prev_key = g.alloc_u32_checked();
    LogQuery fsm_prev_item = allocate_log_query(g);

    for (auto h : obs_unsorted.head) g.enforce_zero(h);
    for (auto h : obs_sorted.head) g.enforce_zero(h);
    Queue4 unsorted0 = select_queue4(g, start_flag, obs_unsorted, fsm_unsorted);
    Queue4 sorted0 = select_queue4(g, start_flag, obs_sorted, fsm_sorted);
    Queue4 empty_q;
    for (auto& h : empty_q.head) h = g.zero();
    for (auto& t : empty_q.tail) t = g.zero();
    empty_q.length = g.u32_const(0);
    Queue4 result0 = select_queue4(g, start_flag, empty_q, fsm_result);

    std::vector<zk_var> fs_input(obs_unsorted.tail.begin(), obs_unsorted.tail.end());
    fs_input.push_back(obs_unsorted.length.v);
    fs_input.insert(fs_input.end(), obs_sorted.tail.begin(), obs_sorted.tail.end());
    fs_input.push_back(obs_sorted.length.v);
    auto challenges = produce_fs_challenges<ENC + 1>(g, fs_input);
    // native seeding (kernels_queue_seed.hpp) once the host packer has walked the integer state: the accumulators are scans that read the challenges
    cs.native_seed_kind = 6;
    cs.native_seed_outer_vars.clear();
    for (int r = 0; r < 2; ++r)
        for (size_t i = 1; i <= ENC; ++i) cs.native_seed_outer_vars.push_back(challenges[r][i]);

    Num one = g.num_const(1);
    std::array<Num, 2> lhs0, rhs0;
    for (int r = 0; r < 2; ++r) { lhs0[r] = g.select(start_flag, one, fsm_lhs[r]); rhs0[r] = g.select(start_flag, one, fsm_rhs[r]); }
    UInt32 prev_key0 = g.select(start_flag, g.u32_const(0), fsm_prev_key);
    // LogQuery::conditionally_select(start_flag, placeholder, fsm previous_item) (mod.rs:159-168)
    std::vector<zk_var> fsm_prev_flat = flatten_query(fsm_prev_item), prev_item0(36);
    for (int i = 0; i < 36; ++i) prev_item0[i] = g.select(start_flag, g.zero(), fsm_prev_flat[i]);
    // inner prologue (mod.rs:264-278)
    Boolean no_work = g.is_zero(unsorted0.length.v);
    Boolean prev_is_trivial0 = g.multi_or({no_work, start_flag});
    g.enforce_equal(unsorted0.length.v, sorted0.length.v);

    // =========================== loop body (mod.rs:283-403) ===========================
    cs.loop_begin(limit);
    std::vector<zk_var> state_in, state_out;
    auto carry_in = [&](zk_var init_outer) {
        zk_var v = g.next_input();
        cs.link(ZK_LINK_FIRST, v, init_outer);
        state_in.push_back(v);
        return v;
    };
    Boolean prev_is_trivial{carry_in(prev_is_trivial0.v)};
    std::array<Num, 2> lhs, rhs;
    for (int r = 0; r < 2; ++r) lhs[r] = Num{carry_in(lhs0[r].v)};
    for (int r = 0; r < 2; ++r) rhs[r] = Num{carry_in(rhs0[r].v)};
    std::array<zk_var, 4> u_head, s_head, r_tail;
    for (int i = 0; i < 4; ++i) u_head[i] = carry_in(unsorted0.head[i]);
    UInt32 u_len{carry_in(unsorted0.length.v)};
    for (int i = 0; i < 4; ++i) s_head[i] = carry_in(sorted0.head[i]);
    UInt32 s_len{carry_in(sorted0.length.v)};
    for (int i = 0; i < 4; ++i) r_tail[i] = carry_in(result0.tail[i]);
    UInt32 r_len{carry_in(result0.length.v)};
    UInt32 prev_key{carry_in(prev_key0.v)};
    std::vector<zk_var> prev_flat(36);
    for (int i = 0; i < 36; ++i) prev_flat[i] = carry_in(prev_item0[i]);
    LogQuery prev_item = unflatten_query(prev_flat, 0);

    std::array<std::array<zk_var, ENC + 1>, 2> ch;
    for (int r = 0; r < 2; ++r)
        for (size_t i = 0; i <= ENC; ++i) ch[r][i] = i == 0 ? g.one() : cs.loop_import(challenges[r][i]);

    Boolean original_is_empty = g.is_zero(u_len.v), sorted_is_empty = g.is_zero(s_len.v);
    g.enforce_bool_equal(original_is_empty, sorted_is_empty);
    Boolean should_pop = g.negated(original_is_empty);
    Boolean is_trivial = original_is_empty;
    LogQuery unsorted_item = allocate_log_query(g);
    auto original_encoding = encode_log_query(g, unsorted_item);
    queue4_pop(g, u_head, u_len, original_encoding, should_pop);
    LogQuery sorted_item = allocate_log_query(g);
    auto sorted_encoding = encode_log_query(g, sorted_item);
    queue4_pop(g, s_head, s_len, sorted_encoding, should_pop);
    g.conditionally_enforce_true(unsorted_item.rw_flag, should_pop);
    accumulate_grand_products<ENC>(g, lhs, rhs, ch, original_encoding, sorted_encoding, should_pop);
    {
        g.conditionally_enforce_true(sorted_item.rw_flag, should_pop);
        UInt32 sorting_key = sorted_item.timestamp;
        std::array<UInt32, 1> a = {prev_key}, b = {sorting_key};
        auto [keys_are_equal, new_key_is_smaller] = unpacked_long_comparison(g, a, b);
        conditionally_enforce_false(g, new_key_is_smaller, should_pop);
        Boolean same_log = keys_are_equal;
        Boolean same_nontrivial_log = g.multi_and({should_pop, same_log});
        Boolean may_be_different_log = g.negated(same_log);
        Boolean different_nontrivial_log = g.multi_and({should_pop, may_be_different_log});
        g.conditionally_enforce_true(g.negated(sorted_item.rollback), different_nontrivial_log);
        g.conditionally_enforce_true(sorted_item.rollback, same_nontrivial_log);
        Boolean body_keys_equal = g.equals(sorted_item.key, prev_item.key);
        Boolean values_are_equal = g.equals(sorted_item.written_value, prev_item.written_value);
        Boolean same_body = g.multi_and({body_keys_equal, values_are_equal});
        Boolean previous_is_non_trivial = g.negated(prev_is_trivial);
        g.conditionally_enforce_true(same_body, g.multi_and({same_log, previous_is_non_trivial}));
        Boolean previous_item_is_not_rollback = g.negated(prev_item.rollback);
        Boolean maybe_add_to_queue = g.b_or(may_be_different_log, is_trivial);
        Boolean add_to_the_queue = g.multi_and({previous_is_non_trivial, maybe_add_to_queue, previous_item_is_not_rollback});
        queue4_push(g, r_tail, r_len, encode_log_query(g, query_to_add(g, prev_item)), add_to_the_queue);
        prev_is_trivial = is_trivial;
        prev_item = sorted_item;
        prev_key = sorting_key;
    }
    state_out.push_back(prev_is_trivial.v);
    for (auto& x : lhs) state_out.push_back(x.v);
    for (auto& x : rhs) state_out.push_back(x.v);
    for (auto v : u_head) state_out.push_back(v);
    state_out.push_back(u_len.v);
    for (auto v : s_head) state_out.push_back(v);
    state_out.push_back(s_len.v);
    for (auto v : r_tail) state_out.push_back(v);
    state_out.push_back(r_len.v);
    state_out.push_back(prev_key.v);
    for (auto v : flatten_query(prev_item)) state_out.push_back(v);
    if (state_in.size() != 57 || state_out.size() != 57) throw ZkError(ZK_ERR_INVALID, "log_sorter: carried state size");
    for (size_t i = 0; i < state_in.size(); ++i) cs.link(ZK_LINK_CARRY, state_in[i], state_out[i]);
    cs.loop_end();

    // =========================== finalisation (mod.rs:405-440) + epilogue (mod.rs:186-231) ===========================
    std::vector<zk_var> fin;
    for (auto v : state_out) fin.push_back(cs.loop_last(v));
    size_t n = 0;
    Boolean f_prev_trivial{fin[n++]};
    std::array<Num, 2> lhs_f = {Num{fin[n]}, Num{fin[n + 1]}}, rhs_f = {Num{fin[n + 2]}, Num{fin[n + 3]}};
    n += 4;
    Queue4 unsorted_f = unsorted0, sorted_f = sorted0, result_f = result0;
    for (int i = 0; i < 4; ++i) unsorted_f.head[i] = fin[n++];
    unsorted_f.length = UInt32{fin[n++]};
    for (int i = 0; i < 4; ++i) sorted_f.head[i] = fin[n++];
    sorted_f.length = UInt32{fin[n++]};
    for (int i = 0; i < 4; ++i) result_f.tail[i] = fin[n++];
    result_f.length = UInt32{fin[n++]};
    UInt32 prev_key_f{fin[n++]};
    LogQuery prev_item_f = unflatten_query(fin, n);
    {
        Boolean now_empty = g.is_zero(unsorted_f.length.v);
        Boolean add = g.multi_and({g.negated(f_prev_trivial), g.negated(prev_item_f.rollback), now_empty});
        queue4_push(g, result_f.tail, result_f.length, encode_log_query(g, query_to_add(g, prev_item_f)), add);
    }
    queue4_enforce_consistency(g, unsorted_f);
    queue4_enforce_consistency(g, sorted_f);
    Boolean unsorted_is_empty = g.is_zero(unsorted_f.length.v), sorted_is_empty_f = g.is_zero(sorted_f.length.v);
    g.enforce_bool_equal(unsorted_is_empty, sorted_is_empty_f);
    Boolean completed = g.is_zero(unsorted_f.length.v);
    for (int r = 0; r < 2; ++r) conditionally_enforce_equal(g, completed, lhs_f[r].v, rhs_f[r].v);

    auto app = [](std::vector<zk_var>& dst, const std::vector<zk_var>& src) { dst.insert(dst.end(), src.begin(), src.end()); };
    std::vector<zk_var> fsm_out = {lhs_f[0].v, lhs_f[1].v, rhs_f[0].v, rhs_f[1].v};
    app(fsm_out, unsorted_f.flatten()); app(fsm_out, sorted_f.flatten()); app(fsm_out, result_f.flatten());
    fsm_out.push_back(prev_key_f.v);
    app(fsm_out, flatten_query(prev_item_f));
    std::vector<zk_var> fsm_in = {fsm_lhs[0].v, fsm_lhs[1].v, fsm_rhs[0].v, fsm_rhs[1].v};
    app(fsm_in, fsm_unsorted.flatten()); app(fsm_in, fsm_sorted.flatten()); app(fsm_in, fsm_result.flatten());
    fsm_in.push_back(fsm_prev_key.v);
    app(fsm_in, fsm_prev_flat);
    std::vector<zk_var> obs_in = obs_unsorted.flatten();
    app(obs_in, obs_sorted.flatten());
    std::vector<zk_var> obs_out = select_queue4(g, completed, result_f, empty_q).flatten();

    auto c_obs_in = g.commit_encoding(obs_in), c_obs_out = g.commit_encoding(obs_out);
    auto c_fsm_in = g.commit_encoding(fsm_in), c_fsm_out = g.commit_encoding(fsm_out);
    Num zero_num = g.num_const(0);
    std::vector<zk_var> compact = {start_flag.v, completed.v};
    for (int i = 0; i < 4; ++i) compact.push_back(c_obs_in[i].v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completed, c_obs_out[i], zero_num).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(start_flag, zero_num, c_fsm_in[i]).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completed, zero_num, c_fsm_out[i]).v);
    auto commitment = g.commit_encoding(compact);
    for (auto& el : commitment) cs.place_gate(ZK_GATE_PUBLIC_INPUT, &el.v, 1, nullptr, 0);
}

}  // namespace zkgl
