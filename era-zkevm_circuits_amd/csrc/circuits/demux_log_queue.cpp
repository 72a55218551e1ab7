// circuits/demux_log_queue.cpp — host-side mirror of /root/reference/src/demux_log_queue/mod.rs:
//   demultiplex_storage_logs_enty_point :38-232, demultiplex_storage_logs_inner :247-396,
//   push_with_optimize :401-442, check_if_bitmask_and_if_empty :444-455; structs: input.rs:25-110.
// One LogQuery is popped per cycle and pushed to exactly one of six queues (rollup storage, events, L1 messages,
// keccak / sha256 / ecrecover precompile calls) chosen by aux_byte, shard and address.
//
// INPUT STREAMS
//   outer (73 words): start_flag | observable_input.initial_log_queue_state[9]
//     | hidden_fsm_input (LogDemuxerFSMInputOutput, input.rs:25-33): initial_log_queue_state[9], storage[9], events[9],
//       l1messages[9], keccak256[9], sha256[9], ecrecover[9]
//   loop (71 words): carried[35] = initial queue head[4] + length | 6 x (output queue tail[4] + length) in the order
//       storage, events, l1messages, keccak256, sha256, ecrecover | popped LogQuery[36]
//
// [EXT] zkevm_opcode_defs v1.4.1: STORAGE/EVENT/L1_MESSAGE/PRECOMPILE_AUX_BYTE = 0/1/2/3; formal precompile
// addresses keccak256 0x8010, sha256 0x02, ecrecover 0x01.
#include "log_query.hpp"

namespace zkgl {

void log_sorter_configure(CS& cs);

namespace {
constexpr int NQ = 6;
constexpr uint32_t AUX_STORAGE = 0, AUX_EVENT = 1, AUX_L1_MESSAGE = 2, AUX_PRECOMPILE = 3;
constexpr uint32_t ADDR_KECCAK = 0x8010, ADDR_SHA256 = 0x02, ADDR_ECRECOVER = 0x01;
constexpr int CARRIED = 5 + NQ * 5;

Boolean address_equals(G& g, const std::array<UInt32, 5>& a, uint32_t constant) {  // UInt160::equals
    std::vector<Boolean> eq;
    for (int i = 0; i < 5; ++i) eq.push_back(g.equals(a[i].v, g.constant(i == 0 ? constant : 0)));
    return g.multi_and(eq);
}
}  // namespace

void demux_log_queue_configure(CS& cs) { log_sorter_configure(cs); }  // same CS as the sibling queue circuits (mod.rs:480-560)

void demultiplex_storage_logs_entry_point(CS& cs, uint32_t limit) {
    G g(cs);
    Boolean start_flag = g.alloc_bool();
    Queue4 obs_initial = alloc_queue4(g);
    Queue4 f_initial = alloc_queue4(g);
    std::array<Queue4, NQ> f_out;
    for (auto& q : f_out) q = alloc_queue4(g);

    for (auto h : obs_initial.head) g.enforce_zero(h);  // enforce_trivial_head (mod.rs:57-60)
    Queue4 initial = select_queue4(g, start_flag, obs_initial, f_initial);
    Queue4 empty;
    for (auto& h : empty.head) h = g.zero();
    for (auto& t : empty.tail) t = g.zero();
    empty.length = g.u32_const(0);
    std::array<Queue4, NQ> outq;
    for (int i = 0; i < NQ; ++i) outq[i] = select_queue4(g, start_flag, empty, f_out[i]);

    cs.side_begin();
    std::vector<zk_var> obs_in = obs_initial.flatten();
    std::vector<zk_var> fsm_in = f_initial.flatten();
    for (auto& q : f_out)
        for (auto v : q.flatten()) fsm_in.push_back(v);
    auto c_obs_in = g.commit_encoding(obs_in);
    auto c_fsm_in = g.commit_encoding(fsm_in);

    std::array<zk_var, CARRIED> init{};
    {
        int n = 0;
        for (auto v : initial.head) init[n++] = v;
        init[n++] = initial.length.v;
        for (auto& q : outq) {
            for (auto v : q.tail) init[n++] = v;
            init[n++] = q.length.v;
        }
    }

    // =========================== loop body (mod.rs:278-393), recorded once ===========================
    cs.loop_begin(limit);
    std::array<zk_var, CARRIED> in{}, out{};
    for (int i = 0; i < CARRIED; ++i) {
        in[i] = g.next_input();
        cs.link(ZK_LINK_FIRST, in[i], init[i]);
    }
    std::array<zk_var, 4> head = {in[0], in[1], in[2], in[3]};
    UInt32 len{in[4]};
    std::array<std::array<zk_var, 4>, NQ> tails;
    std::array<UInt32, NQ> lens;
    for (int i = 0; i < NQ; ++i) {
        for (int k = 0; k < 4; ++k) tails[i][k] = in[5 + 5 * i + k];
        lens[i] = UInt32{in[5 + 5 * i + 4]};
    }

    Boolean queue_is_empty = g.is_zero(len.v);
    Boolean execute = g.negated(queue_is_empty);
    LogQuery popped = allocate_log_query(g);
    auto enc = encode_log_query(g, popped);
    queue4_pop(g, head, len, enc, execute);

    Boolean is_storage_aux = g.equals(popped.aux_byte.v, g.constant(AUX_STORAGE));
    Boolean is_event_aux = g.equals(popped.aux_byte.v, g.constant(AUX_EVENT));
    Boolean is_l1_message_aux = g.equals(popped.aux_byte.v, g.constant(AUX_L1_MESSAGE));
    Boolean is_precompile_aux = g.equals(popped.aux_byte.v, g.constant(AUX_PRECOMPILE));
    Boolean is_keccak_address = address_equals(g, popped.address, ADDR_KECCAK);
    Boolean is_sha256_address = address_equals(g, popped.address, ADDR_SHA256);
    Boolean is_ecrecover_address = address_equals(g, popped.address, ADDR_ECRECOVER);
    Boolean is_rollup_shard = g.is_zero(popped.shard_id.v);
    Boolean execute_rollup_storage = g.multi_and({is_storage_aux, is_rollup_shard, execute});
    Boolean execute_porter_storage = g.multi_and({is_storage_aux, g.negated(is_rollup_shard), execute});
    g.enforce_zero(execute_porter_storage.v);
    std::array<Boolean, NQ> bitmask = {execute_rollup_storage,
                                       g.b_and(is_event_aux, execute),
                                       g.b_and(is_l1_message_aux, execute),
                                       g.multi_and({is_precompile_aux, is_keccak_address, execute}),
                                       g.multi_and({is_precompile_aux, is_sha256_address, execute}),
                                       g.multi_and({is_precompile_aux, is_ecrecover_address, execute})};

    // push_with_optimize (mod.rs:401-442): select the one target state, push once, scatter tail/length back
    std::array<zk_var, 4> sel_tail = tails[0];
    UInt32 sel_len = lens[0];
    for (int i = 1; i < NQ; ++i) {
        for (int k = 0; k < 4; ++k) sel_tail[k] = g.select(bitmask[i], tails[i][k], sel_tail[k]);
        sel_len = g.select(bitmask[i], lens[i], sel_len);
    }
    queue4_push(g, sel_tail, sel_len, enc, g.bool_const(true));
    for (int i = 0; i < NQ; ++i) {
        for (int k = 0; k < 4; ++k) tails[i][k] = g.select(bitmask[i], sel_tail[k], tails[i][k]);
        lens[i] = g.select(bitmask[i], sel_len, lens[i]);
    }
    // exactly one aux-byte class (mod.rs:383-391)
    zk_var lc = g.linear_combination({{is_storage_aux.v, 1}, {is_event_aux.v, 1}, {is_l1_message_aux.v, 1}, {is_precompile_aux.v, 1}});
    Boolean is_bitmask = g.equals(lc, g.one());
    g.conditionally_enforce_true(is_bitmask, execute);

    {
        int n = 0;
        for (auto v : head) out[n++] = v;
        out[n++] = len.v;
        for (int i = 0; i < NQ; ++i) {
            for (auto v : tails[i]) out[n++] = v;
            out[n++] = lens[i].v;
        }
    }
    for (int i = 0; i < CARRIED; ++i) cs.link(ZK_LINK_CARRY, in[i], out[i]);
    cs.loop_end();

    // =========================== epilogue (mod.rs:395, 117-231) ===========================
    std::array<zk_var, CARRIED> fin;
    for (int i = 0; i < CARRIED; ++i) fin[i] = cs.loop_last(out[i]);
    Queue4 initial_final = initial;
    for (int k = 0; k < 4; ++k) initial_final.head[k] = fin[k];
    initial_final.length = UInt32{fin[4]};
    queue4_enforce_consistency(g, initial_final);
    std::array<Queue4, NQ> out_final = outq;
    for (int i = 0; i < NQ; ++i) {
        for (int k = 0; k < 4; ++k) out_final[i].tail[k] = fin[5 + 5 * i + k];
        out_final[i].length = UInt32{fin[5 + 5 * i + 4]};
    }
    Boolean completed = g.is_zero(initial_final.length.v);

    Num zero_num = g.num_const(0);
    std::vector<zk_var> obs_out, fsm_out = initial_final.flatten();
    for (auto& q : out_final)
        for (auto v : q.flatten()) {
            fsm_out.push_back(v);
            obs_out.push_back(g.select(completed, v, zero_num.v));  // placeholder (all zero) until completed
        }
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
