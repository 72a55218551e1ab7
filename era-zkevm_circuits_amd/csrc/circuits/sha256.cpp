// circuits/sha256.cpp — SHA-256 compression function over byte variables through 8-bit lookup tables
// (kernel K8 of SURVEY.md §2) and the SHA-256 chain over pre-padded 64-byte blocks.
//
// Reference surface: the compression step of `sha256_precompile_inner`
// (/root/reference/src/sha256_round_function/mod.rs:271-285: 16 big-endian u32 words of two memory reads,
// then boojum's `round_function_over_uint32` [EXT] on the `[UInt32; 8]` state whose initial value are the
// SHA-256 IVs, src/sha256_round_function/input.rs:41).  The precompile FSM around it (request queue, 2 memory
// reads + 1 write per cycle: mod.rs:88-340) is `sha256_round_function_entry_point` at the end of this file.
//
// boojum decomposes words into 4-bit chunks with width-4 tables (Maj4/Ch4/TriXor4, [EXT]); this engine keeps
// its lookup width of 3 and works on bytes: rotations/shifts = ByteSplitTable splits + ReductionGates,
// xor/and/andn = 8-bit tables, maj = (a&b) ^ (c&(a^b)), modular additions = one field sum re-split into
// 4 bytes + a small carry (sound because every byte is range-checked by the table lookups that consume it).
//
// INPUT STREAMS: outer none; loop 96 words = carried state[32] (word w little-endian bytes at 4w..4w+3) | block[64].
#include "../gadgets.hpp"
#include "sha256_gadget4.hpp"
#include "log_query.hpp"
#include "memory_query.hpp"

namespace zkgl {

void keccak_configure(CS& cs);

using namespace sha256_gadget;

void sha256_configure(CS& cs) {
    keccak_configure(cs);  // gate set, xor8, andn8, ByteSplit<1..7>
    add_and8_table(cs);
}
// The reference's own configuration of the SHA circuits (/root/reference/src/code_unpacker_sha256/mod.rs:484-566): lookup width 4 x 8
// repetitions, the gate set of its test, and exactly the five width-4 tables — no 8-bit table.  The circuits recorded into such a CS
// use the 4-bit-chunk compression (sha256_gadget4.hpp) and range-check bytes through TriXor4.
void sha256_configure_reference_tables(CS& cs) {
    cs.allow_lookup(4, 8, true);
    for (uint32_t k : {ZK_GATE_CONST, ZK_GATE_FMA, ZK_GATE_REDUCTION4, ZK_GATE_BOOLEAN, ZK_GATE_UINTX_ADD, ZK_GATE_SELECT,
                       ZK_GATE_ZEROCHECK, ZK_GATE_DOT4, ZK_GATE_MATMUL12_EXT, ZK_GATE_MATMUL12_INT, ZK_GATE_NOP,
                       ZK_GATE_PUBLIC_INPUT})
        cs.allow_gate(k);
    sha256_gadget4::add_reference_sha_tables(cs);
}

// FSM entry point: see the block comment above sha256_round_function_entry_point below.
// SHA-256 over `n_blocks` pre-padded 64-byte blocks; public inputs = the 32 digest bytes (big-endian word order).
void sha256_blocks_entry_point(CS& cs, uint32_t n_blocks) {
    G g(cs);
    std::array<zk_var, 32> iv_bytes;
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k) iv_bytes[4 * w + k] = g.constant((SHA_IV[w] >> (8 * k)) & 0xff);
    cs.loop_begin(n_blocks);
    sha256_gadget4::AnySha s(g);
    std::array<Word, 8> st;
    std::vector<zk_var> state_in, state_out;
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k) {
            zk_var v = g.next_input();
            cs.link(ZK_LINK_FIRST, v, iv_bytes[4 * w + k]);
            state_in.push_back(v);
            st[w][k] = v;
        }
    std::array<Word, 16> block;
    for (int w = 0; w < 16; ++w) {
        zk_var be[4];
        for (int k = 0; k < 4; ++k) be[k] = g.next_input();          // message bytes in stream order (big-endian words)
        g.range_check_u8_pair(be[0], be[1]);
        g.range_check_u8_pair(be[2], be[3]);
        for (int k = 0; k < 4; ++k) block[w][k] = be[3 - k];
    }
    s.compress_with_hint(st, block);
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k) state_out.push_back(st[w][k]);
    for (size_t i = 0; i < 32; ++i) cs.link(ZK_LINK_CARRY, state_in[i], state_out[i]);
    cs.loop_end();
    for (int w = 0; w < 8; ++w)
        for (int k = 3; k >= 0; --k) {  // digest bytes: big-endian within each word
            zk_var d = cs.loop_last(state_out[4 * w + k]);
            cs.place_gate(ZK_GATE_PUBLIC_INPUT, &d, 1, nullptr, 0);
        }
}

// One SHA-256 compression as a gadget call of a circuit recorded through the C ABI (zk_gadget_sha256_compress): round_function_over_uint32
// (/root/reference/src/sha256_round_function/mod.rs:271-285) on byte variables of the current scope.  state[4 w + k] = byte k (little-endian)
// of working word w; block[4 w + k] = byte k (little-endian) of message word w.  The table set of the CS picks the decomposition.
void sha256_compress_gadget(CS& cs, zk_var* state, const zk_var* block) {
    G g(cs);
    sha256_gadget4::AnySha s(g);
    std::array<Word, 8> st;
    std::array<Word, 16> bw;
    for (int w = 0; w < 8; ++w) for (int k = 0; k < 4; ++k) st[w][k] = state[4 * w + k];
    for (int w = 0; w < 16; ++w) for (int k = 0; k < 4; ++k) bw[w][k] = block[4 * w + k];
    s.compress_with_hint(st, bw);
    for (int w = 0; w < 8; ++w) for (int k = 0; k < 4; ++k) state[4 * w + k] = st[w][k];
}

// =====================================================================================================
// sha256_round_function_entry_point — host-side mirror of
// /root/reference/src/sha256_round_function/mod.rs:347-468 (entry point) and :88-340 (sha256_precompile_inner),
// FSM structs: src/sha256_round_function/input.rs:20-60, precompile IO: src/base_structures/precompile_input_outputs/mod.rs:23-50.
//
// Per cycle: conditional pop of a precompile request (LogQuery) from the 4-element-tail request queue, two
// memory reads and one conditional memory write pushed to the full-state memory queue, one SHA-256
// compression.  Loop-carried state enters through 60 INPUT words tied by CARRY links.
//
// INPUT STREAMS
//   outer, per instance (87 words, alloc_ignoring_outputs order):
//     [0] start_flag
//     [1..10)  observable_input.initial_log_queue_state {head[4], tail[4], length}
//     [10..35) observable_input.initial_memory_queue_state {head[12], tail[12], length}
//     [35..38) fsm: read_precompile_call, read_words_for_round, completed
//     [38..46) fsm: sha256_inner_state[8] (u32)   [46] timestamp_to_use_for_read  [47] timestamp_to_use_for_write
//     [48..53) fsm: input_page, input_offset, output_page, output_offset, num_rounds
//     [53..62) fsm: log_queue_state   [62..87) fsm: memory_queue_state
//   loop, per cycle (112 words):
//     [0..3) read_precompile_call, read_words_for_round, completed   [3..35) sha256 state bytes (word w LE at 3+4w)
//     [35] ts_read [36] ts_write [37..42) input_page, input_offset, output_page, output_offset, num_rounds
//     [42..46) request queue head  [46] request queue length  [47..59) memory queue tail  [59] memory queue length
//     [60..96) popped LogQuery (36 words, zeros when nothing is popped)
//     [96..104) first read value (u32 limbs, LE)  [104..112) second read value
//
// [EXT] zkevm_opcode_defs v1.4.1 constants: PRECOMPILE_AUX_BYTE = 3, SHA256_ROUND_FUNCTION_PRECOMPILE_FORMAL_ADDRESS = 0x02.
// [EXT] boojum `round_function_over_uint32` returns the little-endian bytes of every new state word, so
//       `write_word.inner[7 - i]` (mod.rs:296-303) is state word i and the written U256 is the big-endian digest.
namespace {
constexpr uint32_t PRECOMPILE_AUX_BYTE = 3;
constexpr uint32_t SHA256_PRECOMPILE_ADDRESS = 0x02;
constexpr int SHA_FSM_CARRIED = 60;
}  // namespace

void sha256_round_function_entry_point(CS& cs, uint32_t limit) {
    G g(cs);
    // ---- alloc_ignoring_outputs (mod.rs:367-370) ----
    Boolean start_flag = g.alloc_bool();
    Queue4 obs_req = alloc_queue4(g);
    auto obs_mem = g.alloc_queue_state<12>();
    Boolean f_rpc = g.alloc_bool(), f_rwfr = g.alloc_bool(), f_completed = g.alloc_bool();
    std::array<UInt32, 8> f_state;
    std::array<std::array<UInt8, 4>, 8> f_state_bytes;
    for (int w = 0; w < 8; ++w) {
        f_state[w] = UInt32{g.next_input()};
        f_state_bytes[w] = g.decompose_into_bytes(f_state[w]);  // == allocate_checked, keeps the bytes
    }
    UInt32 f_ts_read = g.alloc_u32_checked(), f_ts_write = g.alloc_u32_checked();
    std::array<UInt32, 5> f_params;  // input_page, input_offset, output_page, output_offset, num_rounds
    for (auto& x : f_params) x = g.alloc_u32_checked();
    Queue4 f_req = alloc_queue4(g);
    auto f_mem = g.alloc_queue_state<12>();

    // mod.rs:374-403
    for (auto h : obs_req.head) g.enforce_zero(h);
    g.enforce_trivial_head(obs_mem);
    Queue4 req_state = select_queue4(g, start_flag, obs_req, f_req);
    auto mem_state = g.select(start_flag, obs_mem, f_mem);

    // starting FSM state (mod.rs:415-423) and the `can_finish_immediatelly` masking (mod.rs:120-137)
    Boolean b_true = g.bool_const(true), b_false = g.bool_const(false);
    UInt32 zero_u32 = g.u32_const(0);
    Boolean rpc0 = g.select(start_flag, b_true, f_rpc);
    Boolean rwfr0 = g.select(start_flag, b_false, f_rwfr);
    Boolean completed0 = g.select(start_flag, b_false, f_completed);
    std::array<zk_var, 32> state0;
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k)
            state0[4 * w + k] = g.select(start_flag, g.constant((SHA_IV[w] >> (8 * k)) & 0xff), f_state_bytes[w][k].v);
    UInt32 ts_read0 = g.select(start_flag, zero_u32, f_ts_read), ts_write0 = g.select(start_flag, zero_u32, f_ts_write);
    std::array<UInt32, 5> params0;
    for (int i = 0; i < 5; ++i) params0[i] = g.select(start_flag, zero_u32, f_params[i]);
    Boolean input_queue_is_empty = g.is_zero(req_state.length.v);
    Boolean can_finish = g.b_and(rpc0, input_queue_is_empty);
    Boolean not_can_finish = g.negated(can_finish);
    rpc0 = g.b_and(rpc0, not_can_finish);
    rwfr0 = g.b_and(rwfr0, not_can_finish);
    completed0 = g.b_or(completed0, can_finish);

    // commitments of observable_input / hidden_fsm_input do not depend on the loop: side phase
    cs.side_begin();
    std::vector<zk_var> obs_in = obs_req.flatten();
    for (auto v : g.flatten(obs_mem)) obs_in.push_back(v);
    std::vector<zk_var> fsm_in = {f_rpc.v, f_rwfr.v, f_completed.v};
    for (auto& x : f_state) fsm_in.push_back(x.v);
    fsm_in.push_back(f_ts_read.v);
    fsm_in.push_back(f_ts_write.v);
    for (auto& x : f_params) fsm_in.push_back(x.v);
    for (auto v : f_req.flatten()) fsm_in.push_back(v);
    for (auto v : g.flatten(f_mem)) fsm_in.push_back(v);
    auto c_obs_in = g.commit_encoding(obs_in);
    auto c_fsm_in = g.commit_encoding(fsm_in);

    // initial values of the carried words, in loop stream order
    std::array<zk_var, SHA_FSM_CARRIED> init{};
    {
        int n = 0;
        init[n++] = rpc0.v; init[n++] = rwfr0.v; init[n++] = completed0.v;
        for (auto v : state0) init[n++] = v;
        init[n++] = ts_read0.v; init[n++] = ts_write0.v;
        for (auto& x : params0) init[n++] = x.v;
        for (auto v : req_state.head) init[n++] = v;
        init[n++] = req_state.length.v;
        for (auto& t : mem_state.tail) init[n++] = t.v;
        init[n++] = mem_state.length.v;
    }

    // =========================== loop body (mod.rs:139-331), recorded once ===========================
    cs.native_seed_kind = 4;  // the carried FSM state has a native walker (kernels_fsm_seed.hpp)
    cs.loop_begin(limit);
    sha256_gadget4::AnySha s(g);
    std::array<zk_var, SHA_FSM_CARRIED> in{}, out{};
    for (int i = 0; i < SHA_FSM_CARRIED; ++i) {
        in[i] = g.next_input();
        cs.link(ZK_LINK_FIRST, in[i], init[i]);
    }
    Boolean rpc{in[0]}, rwfr{in[1]}, completed{in[2]};
    std::array<Word, 8> st;
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k) st[w][k] = in[3 + 4 * w + k];
    UInt32 ts_read{in[35]}, ts_write{in[36]};
    UInt32 input_page{in[37]}, input_offset{in[38]}, output_page{in[39]}, output_offset{in[40]}, num_rounds{in[41]};
    std::array<zk_var, 4> req_head = {in[42], in[43], in[44], in[45]};
    UInt32 req_len{in[46]};
    std::array<zk_var, 12> mem_tail;
    for (int i = 0; i < 12; ++i) mem_tail[i] = in[47 + i];
    UInt32 mem_len{in[59]};
    Boolean l_false = g.bool_const(false), l_true = g.bool_const(true);

    // pop the request (mod.rs:147-170)
    Boolean req_empty = g.is_zero(req_len.v);
    conditionally_enforce_false(g, req_empty, rpc);
    LogQuery call = allocate_log_query(g);
    auto call_enc = encode_log_query(g, call);
    queue4_pop(g, req_head, req_len, call_enc, rpc);
    conditionally_enforce_equal(g, rpc, call.aux_byte.v, g.constant(PRECOMPILE_AUX_BYTE));
    for (int i = 0; i < 5; ++i)
        conditionally_enforce_equal(g, rpc, call.address[i].v, g.constant(i == 0 ? SHA256_PRECOMPILE_ADDRESS : 0));
    // Sha256PrecompileCallParams::from_encoding (mod.rs:62-80) + selects (mod.rs:174-203)
    input_offset = g.select(rpc, call.key.inner[0], input_offset);
    output_offset = g.select(rpc, call.key.inner[2], output_offset);
    input_page = g.select(rpc, call.key.inner[4], input_page);
    output_page = g.select(rpc, call.key.inner[5], output_page);
    num_rounds = g.select(rpc, call.key.inner[6], num_rounds);
    ts_read = g.select(rpc, call.timestamp, ts_read);
    ts_write = g.select(rpc, g.increment_unchecked(ts_read), ts_write);
    Boolean reset_buffer = g.b_or(rpc, completed);
    rwfr = g.b_or(rpc, rwfr);

    // two memory reads (mod.rs:212-258)
    Boolean zero_rounds_left = g.is_zero(num_rounds.v);
    Boolean should_read = g.negated(zero_rounds_left);
    std::array<Word, 16> block;
    for (int r = 0; r < 2; ++r) {
        MemoryQuery q;
        q.timestamp = ts_read; q.memory_page = input_page; q.index = input_offset;
        q.rw_flag = l_false; q.is_ptr = l_false;
        std::array<std::array<UInt8, 4>, 8> vb;
        for (int i = 0; i < 8; ++i) {
            q.value.inner[i] = UInt32{g.next_input()};
            vb[i] = g.decompose_into_bytes(q.value.inner[i]);
        }
        input_offset = g.select(rwfr, g.increment_unchecked(input_offset), input_offset);
        auto enc = encode_memory_query_with_bytes(g, q, vb[5], vb[6], vb[7]);
        full_queue_push(g, mem_tail, mem_len, enc, should_read);
        // memory is big-endian: message word j = limb 7-j (to_be_bytes + from_be_bytes per 4-byte chunk)
        for (int j = 0; j < 8; ++j)
            for (int k = 0; k < 4; ++k) block[8 * r + j][k] = vb[7 - j][k].v;
    }
    num_rounds = g.select(rwfr, UInt32{g.sub(num_rounds.v, g.one())}, num_rounds);

    // absorb (mod.rs:271-285)
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k) st[w][k] = g.select(reset_buffer, g.constant((SHA_IV[w] >> (8 * k)) & 0xff), st[w][k]);
    s.compress_with_hint(st, block);

    // conditional write of the digest (mod.rs:287-315)
    Boolean no_rounds_left = g.is_zero(num_rounds.v);
    Boolean write_result = g.b_and(rwfr, no_rounds_left);
    {
        MemoryQuery q;
        q.timestamp = ts_write; q.memory_page = output_page; q.index = output_offset;
        q.rw_flag = l_true; q.is_ptr = l_false;
        std::array<std::array<UInt8, 4>, 8> vb;
        for (int i = 0; i < 8; ++i) {
            const Word& w = st[7 - i];
            q.value.inner[i] = UInt32{g.linear_combination({{w[0], 1}, {w[1], 1ull << 8}, {w[2], 1ull << 16}, {w[3], 1ull << 24}})};
            for (int k = 0; k < 4; ++k) vb[i][k] = UInt8{w[k]};
        }
        auto enc = encode_memory_query_with_bytes(g, q, vb[5], vb[6], vb[7]);
        full_queue_push(g, mem_tail, mem_len, enc, write_result);
    }

    // FSM update (mod.rs:319-331)
    Boolean input_is_empty = g.is_zero(req_len.v);
    Boolean nothing_left = g.b_and(write_result, input_is_empty);
    Boolean process_next = g.b_and(write_result, g.negated(input_is_empty));
    rpc = process_next;
    completed = g.b_or(nothing_left, completed);
    rwfr = g.negated(g.b_or(rpc, completed));

    {
        int n = 0;
        out[n++] = rpc.v; out[n++] = rwfr.v; out[n++] = completed.v;
        for (int w = 0; w < 8; ++w)
            for (int k = 0; k < 4; ++k) out[n++] = st[w][k];
        out[n++] = ts_read.v; out[n++] = ts_write.v;
        out[n++] = input_page.v; out[n++] = input_offset.v; out[n++] = output_page.v; out[n++] = output_offset.v;
        out[n++] = num_rounds.v;
        for (auto v : req_head) out[n++] = v;
        out[n++] = req_len.v;
        for (auto v : mem_tail) out[n++] = v;
        out[n++] = mem_len.v;
    }
    for (int i = 0; i < SHA_FSM_CARRIED; ++i) cs.link(ZK_LINK_CARRY, in[i], out[i]);
    cs.loop_end();

    // =========================== epilogue (mod.rs:333-336, 436-467) ===========================
    std::array<zk_var, SHA_FSM_CARRIED> fin;
    for (int i = 0; i < SHA_FSM_CARRIED; ++i) fin[i] = cs.loop_last(out[i]);
    Queue4 req_final = req_state;
    for (int i = 0; i < 4; ++i) req_final.head[i] = fin[42 + i];
    req_final.length = UInt32{fin[46]};
    queue4_enforce_consistency(g, req_final);
    auto mem_final = mem_state;
    for (int i = 0; i < 12; ++i) mem_final.tail[i] = Num{fin[47 + i]};
    mem_final.length = UInt32{fin[59]};
    Boolean done{fin[2]};

    // observable_output.final_memory_state = select(done, final, placeholder)
    Num zero_num = g.num_const(0);
    std::vector<zk_var> obs_out;
    for (auto v : g.flatten(mem_final)) obs_out.push_back(g.select(done, v, zero_num.v));
    std::vector<zk_var> fsm_out = {fin[0], fin[1], fin[2]};
    for (int w = 0; w < 8; ++w)
        fsm_out.push_back(g.linear_combination({{fin[3 + 4 * w], 1}, {fin[4 + 4 * w], 1ull << 8}, {fin[5 + 4 * w], 1ull << 16}, {fin[6 + 4 * w], 1ull << 24}}));
    for (int i = 35; i < 42; ++i) fsm_out.push_back(fin[i]);
    for (auto v : req_final.flatten()) fsm_out.push_back(v);
    for (auto v : g.flatten(mem_final)) fsm_out.push_back(v);

    // ClosedFormInputCompactForm::from_full_form (src/fsm_input_output/mod.rs:178-253)
    auto c_obs_out = g.commit_encoding(obs_out);
    auto c_fsm_out = g.commit_encoding(fsm_out);
    std::vector<zk_var> compact = {start_flag.v, done.v};
    for (int i = 0; i < 4; ++i) compact.push_back(c_obs_in[i].v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(done, c_obs_out[i], zero_num).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(start_flag, zero_num, c_fsm_in[i]).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(done, zero_num, c_fsm_out[i]).v);
    auto input_commitment = g.commit_encoding(compact);
    for (auto& el : input_commitment) cs.place_gate(ZK_GATE_PUBLIC_INPUT, &el.v, 1, nullptr, 0);
}

}  // namespace zkgl
