// circuits/sort_decommits.cpp — host-side mirror of /root/reference/src/sort_decommittment_requests/mod.rs:
//   sort_and_deduplicate_code_decommittments_entry_point :40-222, ..._inner :224-372, concatenate_key :374-391;
//   DecommitQuery + encode: src/base_structures/decommit_query/mod.rs:22-113; FSM structs: input.rs:25-95.
// Two full-state queues (original, sorted by (code_hash, timestamp)) are popped in lock step under the grand-product
// permutation argument; runs of equal code hashes collapse into one record pushed to the result queue.
//
// INPUT STREAMS
//   outer (151 words): start_flag | observable_input: initial_queue_state[25], sorted_queue_initial_state[25]
//     | hidden_fsm_input (input.rs:25-34): initial_queue_state[25], sorted_queue_state[25], final_queue_state[25], lhs[2], rhs[2],
//       previous_packed_key[9], first_encountered_timestamp, previous_record {code_hash[8], page, is_first, timestamp}
//   loop (87 words): carried[65] = previous_item_is_trivial, lhs[2], rhs[2], original head[12]+len, sorted head[12]+len,
//       result tail[12]+len, previous_packed_key[9], first_encountered_timestamp, previous_record[11]
//     | original DecommitQuery[11] | sorted DecommitQuery[11]
#include "decommit_query.hpp"
#include "log_query.hpp"
#include "memory_query.hpp"

namespace zkgl {

void log_sorter_configure(CS& cs);

namespace {
constexpr int REPS = 2, ENC = 8, KEY = 9, CARRIED = 65;

void enforce_full_queue_consistency(G& g, const QueueState<12>& q) {  // empty => head == tail
    Boolean is_empty = g.is_zero(q.length.v);
    for (int i = 0; i < 12; ++i) conditionally_enforce_equal(g, is_empty, q.head[i].v, q.tail[i].v);
}
}  // namespace

void sort_decommits_configure(CS& cs) { log_sorter_configure(cs); }  // the reference test's CS (mod.rs:400-480)

void sort_and_deduplicate_code_decommittments_entry_point(CS& cs, uint32_t limit) {
    G g(cs);
    Boolean start_flag = g.alloc_bool();
    auto obs_initial = g.alloc_queue_state<12>();
    auto obs_sorted = g.alloc_queue_state<12>();
    auto f_initial = g.alloc_queue_state<12>();
    auto f_sorted = g.alloc_queue_state<12>();
    auto f_final = g.alloc_queue_state<12>();
    std::array<Num, REPS> f_lhs, f_rhs;
    for (auto& x : f_lhs) x = g.alloc_num();
    for (auto& x : f_rhs) x = g.alloc_num();
    std::array<UInt32, KEY> f_prev_key;
    for (auto& x : f_prev_key) x = g.alloc_u32_checked();
    UInt32 f_first_ts = g.alloc_u32_checked();
    DecommitQuery f_prev_record = allocate_decommit_query(g);

    auto initial = g.select(start_flag, obs_initial, f_initial);  // mod.rs:64-73
    g.enforce_trivial_head(obs_initial);
    g.enforce_trivial_head(obs_sorted);
    auto sorted = g.select(start_flag, obs_sorted, f_sorted);
    QueueState<12> empty;
    for (auto& h : empty.head) h = g.num_const(0);
    for (auto& t : empty.tail) t = g.num_const(0);
    empty.length = g.u32_const(0);
    auto result = g.select(start_flag, empty, f_final);

    std::vector<zk_var> fs_input;
    for (auto& t : obs_initial.tail) fs_input.push_back(t.v);
    fs_input.push_back(obs_initial.length.v);
    for (auto& t : obs_sorted.tail) fs_input.push_back(t.v);
    fs_input.push_back(obs_sorted.length.v);
    auto challenges = produce_fs_challenges<ENC + 1>(g, fs_input);
    // native seeding (kernels_queue_seed.hpp: k_decommit_seed) once the host packer has written the integer state and the queue states
    // of every cycle (zk_pack_sort_decommits_witness_tails): the accumulators are scans that read the challenges
    cs.native_seed_kind = 8;
    cs.native_seed_outer_vars.clear();
    for (int r = 0; r < REPS; ++r)
        for (int i = 1; i <= ENC; ++i) cs.native_seed_outer_vars.push_back(challenges[r][i]);

    Num one_num = g.num_const(1);
    std::array<Num, REPS> lhs0, rhs0;
    for (int r = 0; r < REPS; ++r) {
        lhs0[r] = g.select(start_flag, one_num, f_lhs[r]);
        rhs0[r] = g.select(start_flag, one_num, f_rhs[r]);
    }
    Boolean not_start = g.negated(start_flag);
    auto masked = [&](zk_var v) { return g.mul(v, not_start.v); };  // select(start_flag, zero / false, v)
    std::vector<zk_var> prev_record0, f_rec_flat = f_prev_record.flatten();
    for (auto v : f_rec_flat) prev_record0.push_back(masked(v));
    std::array<zk_var, KEY> prev_key0;
    for (int i = 0; i < KEY; ++i) prev_key0[i] = masked(f_prev_key[i].v);
    zk_var first_ts0 = masked(f_first_ts.v);
    // inner prologue (mod.rs:252-262)
    g.enforce_equal(initial.length.v, sorted.length.v);
    Boolean no_work = g.is_zero(initial.length.v);
    Boolean prev_trivial0 = g.b_or(no_work, start_flag);

    cs.side_begin();
    std::vector<zk_var> obs_in = g.flatten(obs_initial);
    for (auto v : g.flatten(obs_sorted)) obs_in.push_back(v);
    std::vector<zk_var> fsm_in = g.flatten(f_initial);
    for (auto v : g.flatten(f_sorted)) fsm_in.push_back(v);
    for (auto v : g.flatten(f_final)) fsm_in.push_back(v);
    for (auto& x : f_lhs) fsm_in.push_back(x.v);
    for (auto& x : f_rhs) fsm_in.push_back(x.v);
    for (auto& x : f_prev_key) fsm_in.push_back(x.v);
    fsm_in.push_back(f_first_ts.v);
    for (auto v : f_rec_flat) fsm_in.push_back(v);
    auto c_obs_in = g.commit_encoding(obs_in);
    auto c_fsm_in = g.commit_encoding(fsm_in);

    std::array<zk_var, CARRIED> init{};
    {
        int n = 0;
        init[n++] = prev_trivial0.v;
        for (auto& x : lhs0) init[n++] = x.v;
        for (auto& x : rhs0) init[n++] = x.v;
        for (auto& h : initial.head) init[n++] = h.v;
        init[n++] = initial.length.v;
        for (auto& h : sorted.head) init[n++] = h.v;
        init[n++] = sorted.length.v;
        for (auto& t : result.tail) init[n++] = t.v;
        init[n++] = result.length.v;
        for (auto v : prev_key0) init[n++] = v;
        init[n++] = first_ts0;
        for (auto v : prev_record0) init[n++] = v;
    }

    // =========================== loop body (mod.rs:264-345), recorded once ===========================
    cs.loop_begin(limit);
    std::array<zk_var, CARRIED> in{}, out{};
    for (int i = 0; i < CARRIED; ++i) {
        in[i] = g.next_input();
        cs.link(ZK_LINK_FIRST, in[i], init[i]);
    }
    Boolean prev_trivial{in[0]};
    std::array<Num, REPS> lhs = {Num{in[1]}, Num{in[2]}}, rhs = {Num{in[3]}, Num{in[4]}};
    std::array<zk_var, 12> o_head, s_head, r_tail;
    for (int i = 0; i < 12; ++i) { o_head[i] = in[5 + i]; s_head[i] = in[18 + i]; r_tail[i] = in[31 + i]; }
    UInt32 o_len{in[17]}, s_len{in[30]}, r_len{in[43]};
    std::array<UInt32, KEY> prev_key;
    for (int i = 0; i < KEY; ++i) prev_key[i] = UInt32{in[44 + i]};
    UInt32 first_ts{in[53]};
    DecommitQuery prev_record = unflatten_decommit_query(&in[54]);
    std::array<std::array<zk_var, ENC + 1>, REPS> ch;
    for (int r = 0; r < REPS; ++r)
        for (int i = 0; i <= ENC; ++i) ch[r][i] = (i == 0) ? g.one() : cs.loop_import(challenges[r][i]);

    Boolean original_is_empty = g.is_zero(o_len.v);
    Boolean sorted_is_empty = g.is_zero(s_len.v);
    g.enforce_bool_equal(original_is_empty, sorted_is_empty);
    Boolean should_pop = g.negated(original_is_empty);
    Boolean is_trivial = original_is_empty;
    DecommitQuery original_item = allocate_decommit_query(g);
    auto original_enc = encode_decommit_query(g, original_item);
    full_queue_pop(g, o_head, o_len, original_enc, should_pop);
    DecommitQuery sorted_item = allocate_decommit_query(g);
    auto sorted_enc = encode_decommit_query(g, sorted_item);
    full_queue_pop(g, s_head, s_len, sorted_enc, should_pop);
    accumulate_grand_products<ENC>(g, lhs, rhs, ch, original_enc, sorted_enc, should_pop);

    std::array<UInt32, KEY> packed_key;  // concatenate_key (mod.rs:374-391): timestamp is the least significant limb
    packed_key[0] = sorted_item.timestamp;
    for (int i = 0; i < 8; ++i) packed_key[1 + i] = sorted_item.code_hash.inner[i];
    auto [keys_equal, new_key_is_greater] = unpacked_long_comparison(g, packed_key, prev_key);
    (void)keys_equal;
    g.conditionally_enforce_true(new_key_is_greater, should_pop);
    Boolean same_hash = g.equals(prev_record.code_hash, sorted_item.code_hash);
    Boolean different_hash = g.negated(same_hash);
    g.conditionally_enforce_true(sorted_item.is_first, g.b_and(different_hash, should_pop));
    Boolean prev_non_trivial = g.negated(prev_trivial);
    conditionally_enforce_equal(g, g.b_and(same_hash, prev_non_trivial), sorted_item.page.v, prev_record.page.v);
    Boolean add_to_the_queue = g.b_and(prev_non_trivial, different_hash);
    {
        DecommitQuery record_to_add = prev_record;
        record_to_add.is_first = g.bool_const(true);
        record_to_add.timestamp = first_ts;
        full_queue_push(g, r_tail, r_len, encode_decommit_query(g, record_to_add), add_to_the_queue);
    }
    prev_trivial = is_trivial;
    first_ts = g.select(same_hash, first_ts, sorted_item.timestamp);
    prev_record = sorted_item;
    prev_key = packed_key;

    {
        int n = 0;
        out[n++] = prev_trivial.v;
        for (auto& x : lhs) out[n++] = x.v;
        for (auto& x : rhs) out[n++] = x.v;
        for (auto v : o_head) out[n++] = v;
        out[n++] = o_len.v;
        for (auto v : s_head) out[n++] = v;
        out[n++] = s_len.v;
        for (auto v : r_tail) out[n++] = v;
        out[n++] = r_len.v;
        for (auto& x : prev_key) out[n++] = x.v;
        out[n++] = first_ts.v;
        for (auto v : prev_record.flatten()) out[n++] = v;
    }
    for (int i = 0; i < CARRIED; ++i) cs.link(ZK_LINK_CARRY, in[i], out[i]);
    cs.loop_end();

    // =========================== epilogue (mod.rs:347-371, 168-221) ===========================
    std::array<zk_var, CARRIED> fin;
    for (int i = 0; i < CARRIED; ++i) fin[i] = cs.loop_last(out[i]);
    QueueState<12> initial_f = initial, sorted_f = sorted, result_f = result;
    for (int i = 0; i < 12; ++i) { initial_f.head[i] = Num{fin[5 + i]}; sorted_f.head[i] = Num{fin[18 + i]}; }
    initial_f.length = UInt32{fin[17]};
    sorted_f.length = UInt32{fin[30]};
    std::array<zk_var, 12> rt;
    for (int i = 0; i < 12; ++i) rt[i] = fin[31 + i];
    UInt32 rl{fin[43]};
    Boolean completed = g.is_zero(initial_f.length.v);
    g.enforce_bool_equal(completed, g.is_zero(sorted_f.length.v));
    DecommitQuery last_record = unflatten_decommit_query(&fin[54]);
    {
        Boolean add = g.b_and(g.negated(Boolean{fin[0]}), completed);
        DecommitQuery record_to_add = last_record;
        record_to_add.is_first = g.bool_const(true);
        record_to_add.timestamp = UInt32{fin[53]};
        full_queue_push(g, rt, rl, encode_decommit_query(g, record_to_add), add);
    }
    for (int i = 0; i < 12; ++i) result_f.tail[i] = Num{rt[i]};
    result_f.length = rl;
    enforce_full_queue_consistency(g, initial_f);
    enforce_full_queue_consistency(g, sorted_f);
    for (int r = 0; r < REPS; ++r) conditionally_enforce_equal(g, completed, fin[1 + r], fin[3 + r]);

    Num zero_num = g.num_const(0);
    std::vector<zk_var> obs_out;
    for (auto v : g.flatten(result_f)) obs_out.push_back(g.select(completed, v, zero_num.v));
    std::vector<zk_var> fsm_out = g.flatten(initial_f);
    for (auto v : g.flatten(sorted_f)) fsm_out.push_back(v);
    for (auto v : g.flatten(result_f)) fsm_out.push_back(v);
    for (int i = 1; i <= 4; ++i) fsm_out.push_back(fin[i]);
    for (int i = 44; i < 65; ++i) fsm_out.push_back(fin[i]);  // previous_packed_key, first_encountered_timestamp, previous_record
    auto c_obs_out = g.commit_encoding(obs_out);
    auto c_fsm_out = g.commit_encoding(fsm_out);
    std::vector<zk_var> compact = {start_flag.v, completed.v};
    for (int i = 0; i < 4; ++i) compact.push_back(c_obs_in[i].v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completed, c_obs_out[i], zero_num).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(start_flag, zero_num, c_fsm_in[i]).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completed, zero_num, c_fsm_out[i]).v);
    auto input_commitment = g.commit_encoding(compact);
    for (auto& el : input_commitment) cs.place_gate(ZK_GATE_PUBLIC_INPUT, &el.v, 1, nullptr, 0);
}

}  // namespace zkgl
