// circuits/code_unpacker.cpp — host-side mirror of /root/reference/src/code_unpacker_sha256/mod.rs:
//   unpack_code_into_memory_entry_point :33-142, unpack_code_into_memory_inner :144-442, decompose_uint32_to_uint16s :444-455;
//   FSM structs: input.rs:23-80.
// Per cycle: conditional pop of a DecommitQuery (full-state queue), up to two bytecode words written to the memory queue,
// one SHA-256 compression over them (padding + bit length in the last round), comparison of the 224 low bits of the
// digest with the versioned code hash of the request.
//
// INPUT STREAMS
//   outer (125 words): start_flag | observable_input: sorted_requests_queue_initial_state[25], memory_queue_initial_state[25]
//     | hidden_fsm_input: internal_fsm {sha256_inner_state[8], hash_to_compare_against[8], current_index, current_page,
//       timestamp, num_rounds_left, length_in_bits, state_get_from_queue, state_decommit, finished},
//       decommittment_requests_queue_state[25], memory_queue_state[25]
//   loop (101 words): carried[74] = sha256 state bytes[32] (word w LE at 4w) | hash_to_compare_against[8] | current_index,
//       current_page, timestamp, num_rounds_left, length_in_bits | state_get_from_queue, state_decommit, finished
//       | requests queue head[12] + length | memory queue tail[12] + length
//     | popped DecommitQuery[11] | code word 0 [8 u32 limbs LE] | code word 1 [8]
//
// [EXT] zkevm_opcode_defs: ContractCodeSha256::VERSION_BYTE = 0x01 (pinned by the reference fixture: the code hash of
// mod.rs:662-667 starts with 0x0100 0x0021 and continues with sha256(bytecode)[4..32]).
#include "decommit_query.hpp"
#include "log_query.hpp"
#include "memory_query.hpp"
#include "sha256_gadget4.hpp"

namespace zkgl {

void sha256_configure(CS& cs);

using namespace sha256_gadget;

namespace {
constexpr uint32_t VERSIONED_HASH_TOP_16_BITS = 0x01 << 8;
constexpr int CARRIED = 74;
}  // namespace

void code_unpacker_configure(CS& cs) { sha256_configure(cs); }

void unpack_code_into_memory_entry_point(CS& cs, uint32_t limit) {
    G g(cs);
    Boolean start_flag = g.alloc_bool();
    auto obs_req = g.alloc_queue_state<12>();
    auto obs_mem = g.alloc_queue_state<12>();
    std::array<UInt32, 8> f_state;
    std::array<std::array<UInt8, 4>, 8> f_state_bytes;
    for (int w = 0; w < 8; ++w) {
        f_state[w] = UInt32{g.next_input()};
        f_state_bytes[w] = g.decompose_into_bytes(f_state[w]);
    }
    UInt256 f_hash = g.alloc_u256_checked();
    UInt32 f_index = g.alloc_u32_checked(), f_page = g.alloc_u32_checked(), f_timestamp = g.alloc_u32_checked();
    UInt32 f_rounds_left = g.alloc_u32_checked();  // UInt16 in the reference; a u32 range check is the weaker bound
    UInt32 f_length_in_bits = g.alloc_u32_checked();
    Boolean f_get = g.alloc_bool(), f_decommit = g.alloc_bool(), f_finished = g.alloc_bool();
    auto f_req = g.alloc_queue_state<12>();
    auto f_mem = g.alloc_queue_state<12>();

    auto req_state = g.select(start_flag, obs_req, f_req);  // mod.rs:57-80
    auto mem_state = g.select(start_flag, obs_mem, f_mem);
    // starting FSM = placeholder (all zero, also the SHA-256 state) with state_get_from_queue = true (mod.rs:82-90)
    Boolean not_start = g.negated(start_flag);
    auto masked = [&](zk_var v) { return g.mul(v, not_start.v); };

    // values entering the first cycle: computed in the PRE phase (the seeding pass and the FIRST links read them)
    std::array<zk_var, CARRIED> init{};
    {
        int n = 0;
        for (int w = 0; w < 8; ++w)
            for (int k = 0; k < 4; ++k) init[n++] = masked(f_state_bytes[w][k].v);
        for (auto& x : f_hash.inner) init[n++] = masked(x.v);
        for (zk_var v : {f_index.v, f_page.v, f_timestamp.v, f_rounds_left.v, f_length_in_bits.v}) init[n++] = masked(v);
        init[n++] = g.select(start_flag, g.bool_const(true), f_get).v;
        init[n++] = masked(f_decommit.v);
        init[n++] = masked(f_finished.v);
        for (auto& h : req_state.head) init[n++] = h.v;
        init[n++] = req_state.length.v;
        for (auto& t : mem_state.tail) init[n++] = t.v;
        init[n++] = mem_state.length.v;
    }

    cs.side_begin();
    // the commitment follows CodeDecommitterInputData's field order (input.rs:80-83): memory queue first, then the requests queue
    // (the input STREAM keeps requests first: an engine-defined layout)
    std::vector<zk_var> obs_in = g.flatten(obs_mem);
    for (auto v : g.flatten(obs_req)) obs_in.push_back(v);
    std::vector<zk_var> fsm_in;
    for (auto& x : f_state) fsm_in.push_back(x.v);
    for (auto& x : f_hash.inner) fsm_in.push_back(x.v);
    for (zk_var v : {f_index.v, f_page.v, f_timestamp.v, f_rounds_left.v, f_length_in_bits.v, f_get.v, f_decommit.v, f_finished.v}) fsm_in.push_back(v);
    for (auto v : g.flatten(f_req)) fsm_in.push_back(v);
    for (auto v : g.flatten(f_mem)) fsm_in.push_back(v);
    auto c_obs_in = g.commit_encoding(obs_in);
    auto c_fsm_in = g.commit_encoding(fsm_in);

    // =========================== loop body (mod.rs:181-438), recorded once ===========================
    cs.loop_begin(limit);
    sha256_gadget4::AnySha s(g);
    std::array<zk_var, CARRIED> in{}, out{};
    for (int i = 0; i < CARRIED; ++i) {
        in[i] = g.next_input();
        cs.link(ZK_LINK_FIRST, in[i], init[i]);
    }
    std::array<Word, 8> st;
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k) st[w][k] = in[4 * w + k];
    UInt256 hash_to_compare;
    for (int i = 0; i < 8; ++i) hash_to_compare.inner[i] = UInt32{in[32 + i]};
    UInt32 current_index{in[40]}, current_page{in[41]}, timestamp{in[42]}, num_rounds_left{in[43]}, length_in_bits{in[44]};
    Boolean get_from_queue{in[45]}, state_decommit{in[46]}, finished{in[47]};
    std::array<zk_var, 12> req_head, mem_tail;
    for (int i = 0; i < 12; ++i) { req_head[i] = in[48 + i]; mem_tail[i] = in[61 + i]; }
    UInt32 req_len{in[60]}, mem_len{in[73]};
    Boolean l_false = g.bool_const(false), l_true = g.bool_const(true);
    UInt32 zero_u32 = g.u32_const(0);

    // pop the request (mod.rs:182-197)
    conditionally_enforce_false(g, g.is_zero(req_len.v), get_from_queue);
    DecommitQuery request = allocate_decommit_query(g);
    full_queue_pop(g, req_head, req_len, encode_decommit_query(g, request), get_from_queue);
    auto top = g.decompose_into_bytes(request.code_hash.inner[7]);  // decompose_uint32_to_uint16s
    zk_var chunk0 = g.linear_combination({{top[0].v, 1}, {top[1].v, 1ull << 8}});
    zk_var chunk1 = g.linear_combination({{top[2].v, 1}, {top[3].v, 1ull << 8}});
    g.conditionally_enforce_true(g.equals(chunk1, g.constant(VERSIONED_HASH_TOP_16_BITS)), get_from_queue);
    zk_var length_in_words = g.select(get_from_queue, chunk0, g.one());
    // (length_in_words + 1) / 2 must be an integer below 2^16: the bytecode length in words is odd (mod.rs:206-209)
    zk_var length_in_rounds = g.fma((GL_P + 1) / 2, g.add(length_in_words, g.one()), g.one(), 0, length_in_words);
    {
        zk_var first = g.cs.alloc_vars(2);
        zk_var parts[2] = {first, first + 1};
        g.cs.emit_op(ZK_OP_SPLIT, 2, 8, &length_in_rounds, 1, parts, 2, nullptr, 0);  // UInt16::from_variable_checked
        g.enforce_equal(g.linear_combination({{parts[0], 1}, {parts[1], 1ull << 8}}), length_in_rounds);
        g.range_check_u8_pair(parts[0], parts[1]);
    }
    zk_var length_in_bits_may_be = g.fma(32 * 8, length_in_words, g.one(), 0, length_in_words);
    num_rounds_left = g.select(get_from_queue, UInt32{length_in_rounds}, num_rounds_left);
    length_in_bits = g.select(get_from_queue, UInt32{length_in_bits_may_be}, length_in_bits);
    timestamp = g.select(get_from_queue, request.timestamp, timestamp);
    current_page = g.select(get_from_queue, request.page, current_page);
    for (int i = 0; i < 8; ++i)
        hash_to_compare.inner[i] = g.select(get_from_queue, i == 7 ? zero_u32 : request.code_hash.inner[i], hash_to_compare.inner[i]);
    current_index = g.select(get_from_queue, zero_u32, current_index);
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k) st[w][k] = g.select(get_from_queue, g.constant((SHA_IV[w] >> (8 * k)) & 0xff), st[w][k]);
    state_decommit = g.b_or(state_decommit, get_from_queue);

    num_rounds_left = g.select(state_decommit, UInt32{g.sub(num_rounds_left.v, g.one())}, num_rounds_left);
    Boolean last_round = g.is_zero(num_rounds_left.v);
    Boolean finalize = g.b_and(last_round, state_decommit);
    Boolean process_second_word = g.b_and(g.negated(last_round), state_decommit);

    // two code words -> memory queue writes and the SHA-256 block (mod.rs:277-372)
    std::array<Word, 16> block;
    const Boolean push_flags[2] = {state_decommit, process_second_word};
    for (int r = 0; r < 2; ++r) {
        MemoryQuery q;
        q.timestamp = timestamp; q.memory_page = current_page; q.index = current_index;
        q.rw_flag = l_true; q.is_ptr = l_false;
        std::array<std::array<UInt8, 4>, 8> vb;
        for (int i = 0; i < 8; ++i) {
            q.value.inner[i] = UInt32{g.next_input()};
            vb[i] = g.decompose_into_bytes(q.value.inner[i]);
        }
        current_index = g.select(push_flags[r], g.increment_unchecked(current_index), current_index);
        full_queue_push(g, mem_tail, mem_len, encode_memory_query_with_bytes(g, q, vb[5], vb[6], vb[7]), push_flags[r]);
        for (int j = 0; j < 8; ++j)  // to_be_bytes + from_be_bytes per 4-byte chunk: message word j = limb 7-j
            for (int k = 0; k < 4; ++k) block[8 * r + j][k] = vb[7 - j][k].v;
    }
    auto len_bytes = g.decompose_into_bytes(length_in_bits);
    for (int j = 0; j < 8; ++j)
        for (int k = 0; k < 4; ++k) {
            zk_var pad = j == 0 ? g.constant(k == 3 ? 0x80 : 0) : (j == 7 ? len_bytes[k].v : g.zero());  // 1 << 31, zeros, bit length
            block[8 + j][k] = g.select(finalize, pad, block[8 + j][k]);
        }
    std::array<Word, 8> new_state = st;
    s.compress_with_hint(new_state, block);
    for (int w = 0; w < 8; ++w)
        for (int k = 0; k < 4; ++k) st[w][k] = g.select(state_decommit, new_state[w][k], st[w][k]);
    // digest words 1..7 against the versioned hash (mod.rs:381-407)
    for (int i = 0; i < 7; ++i) {
        const Word& w = new_state[7 - i];
        zk_var word = g.linear_combination({{w[0], 1}, {w[1], 1ull << 8}, {w[2], 1ull << 16}, {w[3], 1ull << 24}});
        conditionally_enforce_equal(g, finalize, word, hash_to_compare.inner[i].v);
    }
    conditionally_enforce_equal(g, finalize, g.zero(), hash_to_compare.inner[7].v);

    Boolean is_empty = g.is_zero(req_len.v);
    finished = g.b_or(finished, g.b_and(is_empty, finalize));
    get_from_queue = g.b_and(g.negated(is_empty), finalize);
    state_decommit = process_second_word;

    {
        int n = 0;
        for (int w = 0; w < 8; ++w)
            for (int k = 0; k < 4; ++k) out[n++] = st[w][k];
        for (auto& x : hash_to_compare.inner) out[n++] = x.v;
        for (zk_var v : {current_index.v, current_page.v, timestamp.v, num_rounds_left.v, length_in_bits.v, get_from_queue.v, state_decommit.v, finished.v})
            out[n++] = v;
        for (auto v : req_head) out[n++] = v;
        out[n++] = req_len.v;
        for (auto v : mem_tail) out[n++] = v;
        out[n++] = mem_len.v;
    }
    for (int i = 0; i < CARRIED; ++i) cs.link(ZK_LINK_CARRY, in[i], out[i]);
    cs.loop_end();

    // =========================== epilogue (mod.rs:440, 106-141) ===========================
    std::array<zk_var, CARRIED> fin;
    for (int i = 0; i < CARRIED; ++i) fin[i] = cs.loop_last(out[i]);
    auto req_final = req_state;
    for (int i = 0; i < 12; ++i) req_final.head[i] = Num{fin[48 + i]};
    req_final.length = UInt32{fin[60]};
    {
        Boolean empty = g.is_zero(req_final.length.v);  // enforce_consistency
        for (int i = 0; i < 12; ++i) conditionally_enforce_equal(g, empty, req_final.head[i].v, req_final.tail[i].v);
    }
    auto mem_final = mem_state;
    for (int i = 0; i < 12; ++i) mem_final.tail[i] = Num{fin[61 + i]};
    mem_final.length = UInt32{fin[73]};
    Boolean done{fin[47]};

    Num zero_num = g.num_const(0);
    std::vector<zk_var> obs_out;
    for (auto v : g.flatten(mem_final)) obs_out.push_back(g.select(done, v, zero_num.v));
    std::vector<zk_var> fsm_out;
    for (int w = 0; w < 8; ++w)
        fsm_out.push_back(g.linear_combination({{fin[4 * w], 1}, {fin[4 * w + 1], 1ull << 8}, {fin[4 * w + 2], 1ull << 16}, {fin[4 * w + 3], 1ull << 24}}));
    for (int i = 32; i < 48; ++i) fsm_out.push_back(fin[i]);
    for (auto v : g.flatten(req_final)) fsm_out.push_back(v);
    for (auto v : g.flatten(mem_final)) fsm_out.push_back(v);
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
