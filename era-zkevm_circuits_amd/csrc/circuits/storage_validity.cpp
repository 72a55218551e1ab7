// circuits/storage_validity.cpp — host-side mirror of
// /root/reference/src/storage_validity_by_grand_product/mod.rs:
//   sort_and_deduplicate_storage_access_entry_point :166-506
//   sort_and_deduplicate_storage_access_inner       :510-897 (loop body :584-833, finalisation :836-880)
//   TimestampedStorageLogRecord encoding            :72-95,  concatenate_key :899-920
// recorded against the zkgl CS with the loop body recorded once (see ram_permutation.cpp for the
// carried-state convention).  `enforce_permutation = false` reproduces what the reference's only test
// exercises (`sort_and_deduplicate_storage_access_inner` on the fixture of test_input.rs, whose sorted
// side is NOT a permutation of the unsorted side: mod.rs:1034-1135).
//
// INPUT STREAMS
//   outer (97 words): start_flag | shard_id_to_process, unsorted_log_queue_state[9], intermediate_sorted_queue_state[9]
//     | hidden_fsm_input in StorageDeduplicatorFSMInputOutput order (input.rs:37-52, 77 words)
//   loop (140 words): carried[67] = is_first, previous_item_is_trivial, lhs[2], rhs[2], cycle_idx,
//     unsorted head[4]+len, intermediate head[4]+len, final tail[4]+len, previous_packed_key[13], previous_key[8],
//     previous_address[5], previous_timestamp, this_cell_has_explicit_read_and_rollback_depth_zero,
//     this_cell_base_value[8], this_cell_current_value[8], this_cell_current_depth
//     | unsorted LogQuery[36] | sorted TimestampedStorageLogRecord[37] (record, timestamp)
#include "log_query.hpp"

namespace zkgl {

namespace {
constexpr size_t ENC = 20;       // TIMESTAMPED_STORAGE_LOG_ENCODING_LEN
constexpr size_t PACKED_KEY = 13;  // PACKED_KEY_LENGTH (input.rs:28)

// TimestampedStorageLogRecord::append_timestamp_to_raw_query_encoding (mod.rs:72-95)
std::array<zk_var, ENC> append_timestamp(G& g, std::array<zk_var, ENC> enc, UInt32 ts) {
    enc[19] = g.linear_combination({{enc[19], 1}, {ts.v, 1ull << 8}});
    return enc;
}

struct CellState {  // the per-cell generation-aware memory (mod.rs:43-52)
    Boolean has_read_at_depth_zero;
    UInt256 base_value, current_value;
    UInt32 depth;
};

// the LogQuery pushed to the final queue when a cell is finished (mod.rs:697-709, 849-861)
LogQuery final_query(G& g, const std::array<UInt32, 5>& address, const UInt256& key, const CellState& c, Boolean should_write,
                     UInt8 shard_id) {
    LogQuery q;
    q.address = address; q.key = key; q.read_value = c.base_value; q.written_value = c.current_value;
    q.rw_flag = should_write; q.aux_byte = UInt8{g.zero()}; q.rollback = g.bool_const(false); q.is_service = g.bool_const(false);
    q.shard_id = shard_id; q.tx_number_in_block = g.u32_const(0); q.timestamp = g.u32_const(0);
    return q;
}
}  // namespace

void storage_validity_configure(CS& cs) {  // the reference test's CS: mod.rs:955-1031 (geometry 100/0/8/4, xor8 table)
    cs.allow_lookup(3, 8, true);
    for (uint32_t k : {ZK_GATE_CONST, ZK_GATE_FMA, ZK_GATE_REDUCTION4, ZK_GATE_BOOLEAN, ZK_GATE_UINTX_ADD, ZK_GATE_SELECT,
                       ZK_GATE_ZEROCHECK, ZK_GATE_DOT4, ZK_GATE_MATMUL12_EXT, ZK_GATE_MATMUL12_INT, ZK_GATE_NOP,
                       ZK_GATE_PUBLIC_INPUT})
        cs.allow_gate(k);
    add_xor8_table(cs);
}

void sort_and_deduplicate_storage_access_entry_point(CS& cs, uint32_t limit, bool enforce_permutation) {
    G g(cs);
    // ---------------- alloc_ignoring_outputs ----------------
    Boolean start_flag = g.alloc_bool();
    UInt8 shard_id = alloc_u8_checked(g);
    Queue4 obs_unsorted = alloc_queue4(g), obs_sorted = alloc_queue4(g);
    std::array<Num, 2> fsm_lhs, fsm_rhs;
    for (auto& x : fsm_lhs) x = g.alloc_num();
    for (auto& x : fsm_rhs) x = g.alloc_num();
    Queue4 fsm_unsorted = alloc_queue4(g), fsm_sorted = alloc_queue4(g), fsm_final = alloc_queue4(g);
    UInt32 fsm_cycle_idx = g.alloc_u32_checked();
    std::array<UInt32, PACKED_KEY> fsm_prev_packed_key;
    for (auto& x : fsm_prev_packed_key) x = g.alloc_u32_checked();
    UInt256 fsm_prev_key = g.alloc_u256_checked();
    std::array<UInt32, 5> fsm_prev_address;
    for (auto& x : fsm_prev_address) x = g.alloc_u32_checked();
    UInt32 fsm_prev_timestamp = g.alloc_u32_checked();
    CellState fsm_cell;
    fsm_cell.has_read_at_depth_zero = g.alloc_bool();
    fsm_cell.base_value = g.alloc_u256_checked();
    fsm_cell.current_value = g.alloc_u256_checked();
    fsm_cell.depth = g.alloc_u32_checked();

    for (auto h : obs_unsorted.head) g.enforce_zero(h);  // passthrough must be trivial (mod.rs:213, 286)
    for (auto h : obs_sorted.head) g.enforce_zero(h);
    Queue4 unsorted0 = select_queue4(g, start_flag, obs_unsorted, fsm_unsorted);
    Queue4 sorted0 = select_queue4(g, start_flag, obs_sorted, fsm_sorted);
    Queue4 empty_q;
    for (auto& h : empty_q.head) h = g.zero();
    for (auto& t : empty_q.tail) t = g.zero();
    empty_q.length = g.u32_const(0);
    Queue4 final0 = select_queue4(g, start_flag, empty_q, fsm_final);

    // produce_fs_challenges over the OBSERVABLE tails (mod.rs:335-353)
    std::vector<zk_var> fs_input(obs_unsorted.tail.begin(), obs_unsorted.tail.end());
    fs_input.push_back(obs_unsorted.length.v);
    fs_input.insert(fs_input.end(), obs_sorted.tail.begin(), obs_sorted.tail.end());
    fs_input.push_back(obs_sorted.length.v);
    auto challenges = produce_fs_challenges<ENC + 1>(g, fs_input);
    // native seeding (kernels_queue_seed.hpp) once the host packer has walked the integer state: the accumulators are scans that read the challenges
    cs.native_seed_kind = 5;
    cs.native_seed_outer_vars.clear();
    for (int r = 0; r < 2; ++r)
        for (size_t i = 1; i <= ENC; ++i) cs.native_seed_outer_vars.push_back(challenges[r][i]);
    cs.native_seed_outer_vars.push_back(shard_id.v);

    Num one = g.num_const(1);
    std::array<Num, 2> lhs0, rhs0;
    for (int r = 0; r < 2; ++r) { lhs0[r] = g.select(start_flag, one, fsm_lhs[r]); rhs0[r] = g.select(start_flag, one, fsm_rhs[r]); }
    UInt32 zero_u32 = g.u32_const(0);
    std::array<UInt32, PACKED_KEY> prev_packed_key0;
    for (size_t i = 0; i < PACKED_KEY; ++i) prev_packed_key0[i] = g.select(start_flag, zero_u32, fsm_prev_packed_key[i]);
    UInt32 cycle_idx0 = g.select(start_flag, zero_u32, fsm_cycle_idx);
    // inner prologue (mod.rs:562-576)
    g.enforce_equal(unsorted0.length.v, sorted0.length.v);
    Boolean no_work = g.is_zero(unsorted0.length.v);
    Boolean prev_item_is_trivial0 = g.b_or(no_work, start_flag);
    zk_var outer_one = g.one();

    // =========================== loop body (mod.rs:584-833), recorded once ===========================
    cs.loop_begin(limit);
    std::vector<zk_var> state_in, state_out;
    auto carry_in = [&](zk_var init_outer) {
        zk_var v = g.next_input();
        cs.link(ZK_LINK_FIRST, v, init_outer);
        state_in.push_back(v);
        return v;
    };
    Boolean is_first{carry_in(outer_one)};
    Boolean prev_item_is_trivial{carry_in(prev_item_is_trivial0.v)};
    std::array<Num, 2> lhs, rhs;
    for (int r = 0; r < 2; ++r) lhs[r] = Num{carry_in(lhs0[r].v)};
    for (int r = 0; r < 2; ++r) rhs[r] = Num{carry_in(rhs0[r].v)};
    UInt32 cycle_idx{carry_in(cycle_idx0.v)};
    std::array<zk_var, 4> u_head, s_head, f_tail;
    for (int i = 0; i < 4; ++i) u_head[i] = carry_in(unsorted0.head[i]);
    UInt32 u_len{carry_in(unsorted0.length.v)};
    for (int i = 0; i < 4; ++i) s_head[i] = carry_in(sorted0.head[i]);
    UInt32 s_len{carry_in(sorted0.length.v)};
    for (int i = 0; i < 4; ++i) f_tail[i] = carry_in(final0.tail[i]);
    UInt32 f_len{carry_in(final0.length.v)};
    std::array<UInt32, PACKED_KEY> prev_packed_key;
    for (size_t i = 0; i < PACKED_KEY; ++i) prev_packed_key[i] = UInt32{carry_in(prev_packed_key0[i].v)};
    UInt256 prev_key;
    for (int i = 0; i < 8; ++i) prev_key.inner[i] = UInt32{carry_in(fsm_prev_key.inner[i].v)};
    std::array<UInt32, 5> prev_address;
    for (int i = 0; i < 5; ++i) prev_address[i] = UInt32{carry_in(fsm_prev_address[i].v)};
    UInt32 prev_timestamp{carry_in(fsm_prev_timestamp.v)};
    CellState cell;
    cell.has_read_at_depth_zero = Boolean{carry_in(fsm_cell.has_read_at_depth_zero.v)};
    for (int i = 0; i < 8; ++i) cell.base_value.inner[i] = UInt32{carry_in(fsm_cell.base_value.inner[i].v)};
    for (int i = 0; i < 8; ++i) cell.current_value.inner[i] = UInt32{carry_in(fsm_cell.current_value.inner[i].v)};
    cell.depth = UInt32{carry_in(fsm_cell.depth.v)};

    std::array<std::array<zk_var, ENC + 1>, 2> ch;
    for (int r = 0; r < 2; ++r)
        for (size_t i = 0; i <= ENC; ++i) ch[r][i] = i == 0 ? g.one() : cs.loop_import(challenges[r][i]);
    Boolean is_start{cs.loop_import(start_flag.v)};
    UInt8 shard_id_l{cs.loop_import(shard_id.v)};

    UInt32 original_timestamp = cycle_idx;
    cycle_idx = g.increment_unchecked(cycle_idx);
    Boolean original_is_empty = g.is_zero(u_len.v), sorted_is_empty = g.is_zero(s_len.v);
    g.enforce_bool_equal(original_is_empty, sorted_is_empty);
    Boolean original_is_not_empty = g.negated(original_is_empty), sorted_is_not_empty = g.negated(sorted_is_empty);
    Boolean should_pop = g.multi_and({original_is_not_empty, sorted_is_not_empty});
    Boolean item_is_trivial = original_is_empty;

    LogQuery original_item = allocate_log_query(g);
    auto original_encoding = encode_log_query(g, original_item);
    queue4_pop(g, u_head, u_len, original_encoding, should_pop);
    LogQuery record = allocate_log_query(g);
    UInt32 timestamp = g.alloc_u32_checked();
    auto sorted_encoding = append_timestamp(g, encode_log_query(g, record), timestamp);
    queue4_pop(g, s_head, s_len, sorted_encoding, should_pop);
    auto extended_original_encoding = append_timestamp(g, original_encoding, original_timestamp);

    Boolean shard_id_is_valid = g.equals(shard_id_l.v, record.shard_id.v);
    g.conditionally_enforce_true(shard_id_is_valid, should_pop);
    accumulate_grand_products<ENC>(g, lhs, rhs, ch, extended_original_encoding, sorted_encoding, should_pop);

    // concatenate_key: LE packing so comparison is subtraction (mod.rs:899-920)
    std::array<UInt32, PACKED_KEY> packed_key;
    for (int i = 0; i < 8; ++i) packed_key[i] = record.key.inner[i];
    for (int i = 0; i < 5; ++i) packed_key[8 + i] = record.address[i];
    auto [keys_are_equal, previous_key_is_greater] = unpacked_long_comparison(g, prev_packed_key, packed_key);
    Boolean not_item_is_trivial = g.negated(item_is_trivial);
    conditionally_enforce_false(g, previous_key_is_greater, not_item_is_trivial);
    auto [ts_diff, previous_timestamp_is_less] = g.overflowing_sub_with_borrow_in(prev_timestamp, timestamp, g.bool_const(false));
    (void)ts_diff;
    Boolean must_enforce = g.b_and(keys_are_equal, not_item_is_trivial);
    g.conditionally_enforce_true(previous_timestamp_is_less, must_enforce);

    {  // if new cell (mod.rs:664-768)
        Boolean not_keys_are_equal = g.negated(keys_are_equal);
        Boolean enforce_first = g.multi_and({is_start, is_first, should_pop});  // `if _cycle == 0` (mod.rs:666-670)
        g.conditionally_enforce_true(not_keys_are_equal, enforce_first);
        Boolean value_is_unchanged = g.equals(cell.current_value, cell.base_value);
        Boolean current_depth_is_zero = g.is_zero(cell.depth.v);
        Boolean unchanged_but_not_by_rollback = g.b_and(value_is_unchanged, g.negated(current_depth_is_zero));
        Boolean issue_protective_read = g.b_or(cell.has_read_at_depth_zero, unchanged_but_not_by_rollback);
        Boolean should_write = g.negated(value_is_unchanged);
        LogQuery query = final_query(g, prev_address, prev_key, cell, should_write, shard_id_l);
        Boolean should_update = g.b_or(issue_protective_read, should_write);
        Boolean should_push = g.b_and(g.negated(prev_item_is_trivial), g.b_and(not_keys_are_equal, should_update));
        queue4_push(g, f_tail, f_len, encode_log_query(g, query), should_push);

        Boolean new_non_trivial_cell = g.b_and(g.negated(item_is_trivial), not_keys_are_equal);
        UInt256 meaningful_value = g.select(record.rw_flag, record.written_value, record.read_value);
        cell.base_value = g.select(new_non_trivial_cell, record.read_value, cell.base_value);
        cell.current_value = g.select(new_non_trivial_cell, meaningful_value, cell.current_value);
        UInt32 depth_for_new_cell = g.select(record.rw_flag, g.u32_const(1), g.u32_const(0));
        cell.depth = g.select(new_non_trivial_cell, depth_for_new_cell, cell.depth);
        cell.has_read_at_depth_zero = g.select(new_non_trivial_cell, g.negated(record.rw_flag), cell.has_read_at_depth_zero);
    }
    {  // if same cell - update (mod.rs:771-826)
        Boolean not_rw_flag = g.negated(record.rw_flag);
        Boolean non_trivial_and_same_cell = g.b_and(g.negated(item_is_trivial), keys_are_equal);
        Boolean non_trivial_read_of_same_cell = g.b_and(non_trivial_and_same_cell, not_rw_flag);
        Boolean non_trivial_write_of_same_cell = g.b_and(non_trivial_and_same_cell, record.rw_flag);
        Boolean write_no_rollback = g.b_and(non_trivial_write_of_same_cell, g.negated(record.rollback));
        Boolean write_rollback = g.b_and(non_trivial_write_of_same_cell, record.rollback);
        cell.depth = g.select(write_no_rollback, g.increment_unchecked(cell.depth), cell.depth);
        cell.depth = g.select(write_rollback, UInt32{g.sub(cell.depth.v, g.one())}, cell.depth);
        Boolean read_is_equal_to_current = g.equals(cell.current_value, record.read_value);
        Boolean check_read_consistency = g.multi_or({non_trivial_read_of_same_cell, write_no_rollback});
        g.conditionally_enforce_true(read_is_equal_to_current, check_read_consistency);
        cell.current_value = g.select(write_no_rollback, record.written_value, cell.current_value);
        cell.current_value = g.select(write_rollback, record.read_value, cell.current_value);
        Boolean read_at_depth_zero_of_same_cell = g.b_and(g.is_zero(cell.depth.v), non_trivial_read_of_same_cell);
        cell.base_value = g.select(read_at_depth_zero_of_same_cell, record.read_value, cell.base_value);
        cell.has_read_at_depth_zero = g.select(read_at_depth_zero_of_same_cell, g.bool_const(true), cell.has_read_at_depth_zero);
    }
    // always update counters (mod.rs:828-833)
    prev_address = record.address;
    prev_key = record.key;
    prev_item_is_trivial = item_is_trivial;
    prev_timestamp = timestamp;
    prev_packed_key = packed_key;

    state_out.push_back(g.zero());  // is_first of the next cycle
    state_out.push_back(prev_item_is_trivial.v);
    for (auto& x : lhs) state_out.push_back(x.v);
    for (auto& x : rhs) state_out.push_back(x.v);
    state_out.push_back(cycle_idx.v);
    for (auto v : u_head) state_out.push_back(v);
    state_out.push_back(u_len.v);
    for (auto v : s_head) state_out.push_back(v);
    state_out.push_back(s_len.v);
    for (auto v : f_tail) state_out.push_back(v);
    state_out.push_back(f_len.v);
    for (auto& x : prev_packed_key) state_out.push_back(x.v);
    for (auto& x : prev_key.inner) state_out.push_back(x.v);
    for (auto& x : prev_address) state_out.push_back(x.v);
    state_out.push_back(prev_timestamp.v);
    state_out.push_back(cell.has_read_at_depth_zero.v);
    for (auto& x : cell.base_value.inner) state_out.push_back(x.v);
    for (auto& x : cell.current_value.inner) state_out.push_back(x.v);
    state_out.push_back(cell.depth.v);
    if (state_in.size() != 67 || state_out.size() != 67) throw ZkError(ZK_ERR_INVALID, "storage_validity: carried state size");
    for (size_t i = 0; i < state_in.size(); ++i) cs.link(ZK_LINK_CARRY, state_in[i], state_out[i]);
    cs.loop_end();

    // =========================== finalisation (mod.rs:836-880) + entry point epilogue (mod.rs:417-505) ===========================
    std::vector<zk_var> fin;
    for (auto v : state_out) fin.push_back(cs.loop_last(v));
    size_t n = 1;
    Boolean f_prev_trivial{fin[n++]};
    std::array<Num, 2> lhs_f = {Num{fin[n]}, Num{fin[n + 1]}}, rhs_f = {Num{fin[n + 2]}, Num{fin[n + 3]}};
    n += 4;
    UInt32 cycle_idx_f{fin[n++]};
    Queue4 unsorted_f = unsorted0, sorted_f = sorted0, final_f = final0;
    for (int i = 0; i < 4; ++i) unsorted_f.head[i] = fin[n++];
    unsorted_f.length = UInt32{fin[n++]};
    for (int i = 0; i < 4; ++i) sorted_f.head[i] = fin[n++];
    sorted_f.length = UInt32{fin[n++]};
    for (int i = 0; i < 4; ++i) final_f.tail[i] = fin[n++];
    final_f.length = UInt32{fin[n++]};
    std::array<UInt32, PACKED_KEY> ppk_f;
    for (auto& x : ppk_f) x = UInt32{fin[n++]};
    UInt256 pk_f;
    for (auto& x : pk_f.inner) x = UInt32{fin[n++]};
    std::array<UInt32, 5> pa_f;
    for (auto& x : pa_f) x = UInt32{fin[n++]};
    UInt32 pts_f{fin[n++]};
    CellState cell_f;
    cell_f.has_read_at_depth_zero = Boolean{fin[n++]};
    for (auto& x : cell_f.base_value.inner) x = UInt32{fin[n++]};
    for (auto& x : cell_f.current_value.inner) x = UInt32{fin[n++]};
    cell_f.depth = UInt32{fin[n++]};
    {
        Boolean queues_exhausted = g.is_zero(unsorted_f.length.v);
        Boolean value_is_unchanged = g.equals(cell_f.current_value, cell_f.base_value);
        Boolean unchanged_but_not_by_rollback = g.b_and(value_is_unchanged, g.negated(g.is_zero(cell_f.depth.v)));
        Boolean issue_protective_read = g.b_or(cell_f.has_read_at_depth_zero, unchanged_but_not_by_rollback);
        Boolean should_write = g.negated(value_is_unchanged);
        LogQuery query = final_query(g, pa_f, pk_f, cell_f, should_write, shard_id);
        Boolean should_update = g.b_or(issue_protective_read, should_write);
        Boolean should_push = g.b_and(g.negated(f_prev_trivial), g.b_and(should_update, queues_exhausted));
        queue4_push(g, final_f.tail, final_f.length, encode_log_query(g, query), should_push);
        cell_f.has_read_at_depth_zero = g.select(queues_exhausted, g.bool_const(false), cell_f.has_read_at_depth_zero);
    }
    queue4_enforce_consistency(g, unsorted_f);
    queue4_enforce_consistency(g, sorted_f);
    Boolean unsorted_is_empty = g.is_zero(unsorted_f.length.v), sorted_is_empty_f = g.is_zero(sorted_f.length.v);
    g.enforce_bool_equal(unsorted_is_empty, sorted_is_empty_f);
    Boolean completed = g.b_and(unsorted_is_empty, sorted_is_empty_f);
    if (enforce_permutation)
        for (int r = 0; r < 2; ++r) conditionally_enforce_equal(g, completed, lhs_f[r].v, rhs_f[r].v);

    // hidden_fsm_output in StorageDeduplicatorFSMInputOutput order
    std::vector<zk_var> fsm_out = {lhs_f[0].v, lhs_f[1].v, rhs_f[0].v, rhs_f[1].v};
    auto app = [](std::vector<zk_var>& dst, const std::vector<zk_var>& src) { dst.insert(dst.end(), src.begin(), src.end()); };
    app(fsm_out, unsorted_f.flatten()); app(fsm_out, sorted_f.flatten()); app(fsm_out, final_f.flatten());
    fsm_out.push_back(cycle_idx_f.v);
    for (auto& x : ppk_f) fsm_out.push_back(x.v);
    for (auto& x : pk_f.inner) fsm_out.push_back(x.v);
    for (auto& x : pa_f) fsm_out.push_back(x.v);
    fsm_out.push_back(pts_f.v);
    fsm_out.push_back(cell_f.has_read_at_depth_zero.v);
    for (auto& x : cell_f.base_value.inner) fsm_out.push_back(x.v);
    for (auto& x : cell_f.current_value.inner) fsm_out.push_back(x.v);
    fsm_out.push_back(cell_f.depth.v);
    std::vector<zk_var> fsm_in = {fsm_lhs[0].v, fsm_lhs[1].v, fsm_rhs[0].v, fsm_rhs[1].v};
    app(fsm_in, fsm_unsorted.flatten()); app(fsm_in, fsm_sorted.flatten()); app(fsm_in, fsm_final.flatten());
    fsm_in.push_back(fsm_cycle_idx.v);
    for (auto& x : fsm_prev_packed_key) fsm_in.push_back(x.v);
    for (auto& x : fsm_prev_key.inner) fsm_in.push_back(x.v);
    for (auto& x : fsm_prev_address) fsm_in.push_back(x.v);
    fsm_in.push_back(fsm_prev_timestamp.v);
    fsm_in.push_back(fsm_cell.has_read_at_depth_zero.v);
    for (auto& x : fsm_cell.base_value.inner) fsm_in.push_back(x.v);
    for (auto& x : fsm_cell.current_value.inner) fsm_in.push_back(x.v);
    fsm_in.push_back(fsm_cell.depth.v);
    std::vector<zk_var> obs_in = {shard_id.v};
    app(obs_in, obs_unsorted.flatten()); app(obs_in, obs_sorted.flatten());
    // observable output: the final sorted queue once completed, else empty (mod.rs:470-487)
    Queue4 obs_out_q = select_queue4(g, completed, final_f, empty_q);
    std::vector<zk_var> obs_out = obs_out_q.flatten();

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
