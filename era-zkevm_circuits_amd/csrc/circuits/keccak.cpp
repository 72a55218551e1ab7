// circuits/keccak.cpp — Keccak-f[1600] over byte variables through 8-bit lookup tables (kernel K8 of
// SURVEY.md §2) and the Keccak-256 sponge over pre-padded 136-byte blocks.
//
// Reference surface: `keccak256_absorb_and_run_permutation`
// (/root/reference/src/keccak256_round_function/mod.rs:796-838: xor the 136-byte block into the state
// `[[[UInt8; 8]; 5]; 5]`, then boojum's `keccak_256_round_function` [EXT]) and the whole-message
// `keccak256(cs, &bytes)` gadget used by eip_4844 (src/eip_4844/mod.rs:156-163, 207, 229-237).
// The precompile FSM around it (request queue, unaligned memory reads, ByteBuffer:
// src/keccak256_round_function/mod.rs:155-670) is `keccak256_round_function_entry_point` at the end of this file.
//
// boojum's own decomposition is absent; this one uses, per round, theta 160+40+200 xor lookups and 40
// bit-rotation splits, rho/pi <= 192 splits, chi 200 andn + 200 xor lookups, iota <= 8 xor lookups
// (~1030 lookups + ~230 ReductionGates per round, 24 rounds).  One loop iteration = one block, the
// 200-byte sponge state is the carried state, so a message of n blocks runs as n GPU lanes.
//
// INPUT STREAMS: outer none; loop 336 words = carried state[200] (byte (x,y,k) at 8*(x+5y)+k) | block[136].
#include "../gadgets.hpp"
#include "keccak_gadget.hpp"
#include "log_query.hpp"
#include "memory_query.hpp"
#include <cstdlib>
#include "../bytebuf_macro.hpp"

namespace zkgl {

void keccak_configure_with(CS& cs, uint32_t lookup_repetitions);
void keccak_configure(CS& cs) {  // geometry as the reference keccak tests: 100/0/8/4 (src/keccak256_round_function/mod.rs:847-852)
    keccak_configure_with(cs, 8);
}
void keccak_configure_with(CS& cs, uint32_t lookup_repetitions) {
    cs.allow_lookup(3, lookup_repetitions, true);
    for (uint32_t k : {ZK_GATE_CONST, ZK_GATE_FMA, ZK_GATE_REDUCTION4, ZK_GATE_BOOLEAN, ZK_GATE_UINTX_ADD, ZK_GATE_SELECT,
                       ZK_GATE_ZEROCHECK, ZK_GATE_DOT4, ZK_GATE_MATMUL12_EXT, ZK_GATE_MATMUL12_INT, ZK_GATE_NOP,
                       ZK_GATE_PUBLIC_INPUT})
        cs.allow_gate(k);
    add_xor8_table(cs);
    {
        std::vector<uint64_t> rows;
        rows.reserve(65536 * 3);
        for (uint64_t a = 0; a < 256; ++a)
            for (uint64_t b = 0; b < 256; ++b) { rows.push_back(a); rows.push_back(b); rows.push_back((~a & 0xff) & b); }
        cs.add_table(TABLE_ANDN8, 2, 1, rows.data(), 65536);
    }
    for (int k = 1; k < 8; ++k) {  // ByteSplitTable<k> (src/keccak256_round_function/mod.rs:953-966)
        std::vector<uint64_t> rows;
        for (uint64_t a = 0; a < 256; ++a) { rows.push_back(a); rows.push_back(a & ((1u << k) - 1)); rows.push_back(a >> k); }
        cs.add_table(TABLE_SPLIT_BASE + k, 1, 2, rows.data(), 256);
    }
}

// Keccak-256 over `n_blocks` pre-padded 136-byte blocks; public inputs = the 32 digest bytes.
void keccak256_blocks_entry_point(CS& cs, uint32_t n_blocks) {
    G g(cs);
    zk_var outer_zero = g.zero();
    cs.loop_begin(n_blocks);
    K k(g);
    std::array<Lane, 25> s;
    std::vector<zk_var> state_in, state_out;
    for (int i = 0; i < 25; ++i)
        for (int b = 0; b < 8; ++b) {
            zk_var v = g.next_input();  // carried sponge state: every byte is an output of a lookup in the previous block
            cs.link(ZK_LINK_FIRST, v, outer_zero);
            state_in.push_back(v);
            s[i][b] = v;
        }
    // absorb: state[0..136) ^= block (keccak256_absorb_and_run_permutation, mod.rs:803-817); the xor lookup
    // range-checks both the carried byte and the fresh input byte
    std::array<zk_var, 136> block;
    for (auto& b : block) b = g.next_input();
    // capacity bytes of the carried state are range-checked through a pair lookup (they enter theta's xor anyway,
    // but only as the first key: make the check explicit)
    for (int j = 136; j < 200; j += 2) g.range_check_u8_pair(s[j / 8][j % 8], s[(j + 1) / 8][(j + 1) % 8]);
    k.absorb_and_permute(s, block.data());
    for (int i = 0; i < 25; ++i)
        for (int b = 0; b < 8; ++b) state_out.push_back(s[i][b]);
    for (size_t i = 0; i < 200; ++i) cs.link(ZK_LINK_CARRY, state_in[i], state_out[i]);
    cs.loop_end();
    for (int j = 0; j < 32; ++j) {
        zk_var d = cs.loop_last(state_out[j]);
        cs.place_gate(ZK_GATE_PUBLIC_INPUT, &d, 1, nullptr, 0);
    }
}

// Keccak-f[1600] as a gadget call of a circuit recorded through the C ABI (zk_gadget_keccak_f1600): the permutation the crate reaches
// through boojum's keccak256 round function (/root/reference/src/keccak256_round_function/mod.rs:796-838), on 200 byte variables of the
// current scope; state[8 (x + 5 y) + k] = byte k (little-endian) of lane (x, y).
void keccak_f1600_gadget(CS& cs, zk_var* state) {
    G g(cs);
    K k(g);
    std::array<Lane, 25> s;
    for (int i = 0; i < 25; ++i)
        for (int b = 0; b < 8; ++b) s[i][b] = state[8 * i + b];
    k.permutation(s);
    for (int i = 0; i < 25; ++i)
        for (int b = 0; b < 8; ++b) state[8 * i + b] = s[i][b];
}

// =====================================================================================================
// keccak256_round_function_entry_point — host-side mirror of
// /root/reference/src/keccak256_round_function/mod.rs:672-794 (entry point), :155-670 (keccak256_precompile_inner),
// :100-142 (trivial_mapping_function), buffer/mod.rs:42-163 (ByteBuffer), input.rs:20-80 (FSM structs).
//
// Per cycle: conditional pop of a precompile request, 6 conditional unaligned memory reads that feed a
// 192-byte shift-register buffer, 136 bytes consumed, padding, one Keccak-f[1600], conditional write of
// the digest.  Loop-carried state enters through 423 INPUT words tied by CARRY links.
//
// INPUT STREAMS
//   outer, per instance (474 words, alloc_ignoring_outputs order):
//     [0] start_flag  [1..10) initial_log_queue_state  [10..35) initial_memory_queue_state
//     [35..39) fsm: read_precompile_call, read_unaligned_words_for_round, padding_round, completed
//     [39..239) fsm: keccak_internal_state[i][j][k] (i-major; lane x=i, y=j)
//     [239] ts_read [240] ts_write
//     [241..247) input_page, input_memory_byte_offset, input_memory_byte_length, output_page, output_word_offset,
//                needs_full_padding_round
//     [247..439) buffer.bytes  [439] buffer.filled   [440..449) log_queue_state   [449..474) memory_queue_state
//   loop, per cycle (507 words):
//     [0..4) the 4 flags   [4..204) keccak state (byte k of lane x+5y at 4+8(x+5y)+k)   [204] ts_read [205] ts_write
//     [206..212) params (same order as above)   [212..404) buffer.bytes  [404] buffer.filled
//     [405..409) request queue head [409] length   [410..422) memory queue tail [422] length
//     [423..459) popped LogQuery (zeros when nothing is popped)   [459+8r..467+8r) read value r (u32 limbs, LE), r < 6
//
// [EXT] zkevm_opcode_defs v1.4.1: PRECOMPILE_AUX_BYTE = 3, KECCAK256_ROUND_FUNCTION_PRECOMPILE_FORMAL_ADDRESS = 0x8010.
namespace {
constexpr uint32_t PRECOMPILE_AUX_BYTE = 3;
constexpr uint32_t KECCAK_PRECOMPILE_ADDRESS = 0x8010;
constexpr int RATE = 136, BUF = 192, READS = 6, KF_CARRIED = 423;

struct ByteBuffer {
    std::array<zk_var, BUF> bytes;
    zk_var filled;
};

zk_var masked(G& g, zk_var x, Boolean b) { return g.mul(x, b.v); }

// ByteBuffer::can_fill_bytes — buffer/mod.rs:42-55
Boolean can_fill_bytes(G& g, const ByteBuffer& buf, zk_var bytes_to_fill) {
    zk_var next_filled = g.add(buf.filled, bytes_to_fill);  // add_no_overflow: the sum must stay a byte
    g.range_check_u8_pair(next_filled, next_filled);
    auto [diff, uf] = g.overflowing_sub_u8(UInt8{g.constant(BUF)}, UInt8{next_filled});
    (void)diff;
    return g.negated(uf);
}

// ByteBuffer::fill_with_bytes with trivial_mapping_function — buffer/mod.rs:69-136, mod.rs:100-142.  The structure is
// zkb::fill_with_bytes (bytebuf_macro.hpp); this is its HOST backend: every primitive records its gate, and its witness op unless the
// whole fill is recorded as ONE macro-op (ZKGL_BYTEBUF_MACRO=1: ZK_OP_BYTEBUF_FILL, whose outputs are the pre-allocated variables the
// walk then constrains — same variables, same gates, same cells as the op-by-op form).
struct BufBackend : zkb::PlainArrays<zk_var> {
    typedef zk_var V;
    G& g;
    zk_var macro_next = ZK_VAR_NONE;
    explicit BufBackend(G& g) : g(g) {}
    V fma(uint64_t q, V a, V b, uint64_t l, V c) {
        if (macro_next == ZK_VAR_NONE) return g.fma(q, a, b, l, c);
        const V d = macro_next++;
        zk_var vars[4] = {a, b, c, d};
        uint64_t k[2] = {q, l};
        g.cs.place_gate(ZK_GATE_FMA, vars, 4, k, 2);
        return d;
    }
    V sub1(V x) { return fma(1, x, g.one(), GL_P - 1, g.one()); }
    V add(V a, V b) { return fma(1, a, g.one(), 1, b); }
    V mul(V a, V b) { return fma(1, a, b, 0, a); }
    V band(V a, V b) { return mul(a, b); }
    V bnot(V a) { return fma(GL_P - 1, a, g.one(), 1, g.one()); }
    V bor(V a, V b) { const V s = add(a, b); return fma(GL_P - 1, a, b, 1, s); }
    V is_zero(V x) {
        if (macro_next == ZK_VAR_NONE) return g.is_zero(x).v;
        const V flag = macro_next++, aux = macro_next++;
        zk_var vars[3] = {x, aux, flag};
        g.cs.place_gate(ZK_GATE_ZEROCHECK, vars, 3, nullptr, 0);
        return flag;
    }
    V select(V s, V a, V b) {
        if (a == b) throw ZkError(ZK_ERR_INVALID, "internal: ByteBuffer select over one variable (the macro-op's output count assumes none)");
        if (macro_next == ZK_VAR_NONE) return g.select(Boolean{s}, a, b);
        const V r = macro_next++;
        zk_var vars[4] = {a, b, s, r};
        g.cs.place_gate(ZK_GATE_SELECT, vars, 4, nullptr, 0);
        return r;
    }
};

void fill_with_bytes(G& g, ByteBuffer& buf, const std::array<zk_var, 32>& input, zk_var offset, zk_var meaningful) {
    BufBackend be(g);
    (void)g.one(); (void)g.zero();   // the constants exist before the macro-op's outputs are allocated
    const char* e = getenv("ZKGL_BYTEBUF_MACRO");
    const bool use_macro = e && e[0] == '1';
    zk_var first = ZK_VAR_NONE;
    uint32_t n = 0;
    if (use_macro) {
        n = zkb::n_outputs();
        std::vector<zk_var> ins(buf.bytes.begin(), buf.bytes.end());
        ins.push_back(buf.filled);
        ins.insert(ins.end(), input.begin(), input.end());
        ins.push_back(offset); ins.push_back(meaningful);
        first = g.cs.alloc_vars(n);
        g.cs.emit_macro_op(ZK_OP_BYTEBUF_FILL, ins.data(), (uint32_t)ins.size(), first, n);
        be.macro_next = first;
    }
    be.load(buf.bytes.data(), input.data(), g.zero());
    zkb::fill_with_bytes(be, buf.filled, offset, meaningful);
    if (use_macro) g.cs.end_macro_op();
    for (int j = 0; j < BUF; ++j) buf.bytes[j] = be.byte(j);
    if (use_macro && be.macro_next != first + n) throw ZkError(ZK_ERR_INVALID, "internal: the ByteBuffer gadget and its macro-op disagree on the output count");
    g.range_check_u8_pair(buf.filled, g.sub(g.constant(BUF), buf.filled));  // filled <= capacity
}

// ByteBuffer::consume::<136>(allow_partial = true) — buffer/mod.rs:138-162
std::array<zk_var, RATE> consume(G& g, ByteBuffer& buf) {
    auto [leftover, uf] = g.overflowing_sub_u8(UInt8{buf.filled}, UInt8{g.constant(RATE)});
    buf.filled = masked(g, leftover.v, g.negated(uf));
    std::array<zk_var, RATE> out;
    for (int j = 0; j < RATE; ++j) out[j] = buf.bytes[j];
    zk_var zero = g.zero();
    for (int j = 0; j < BUF; ++j) buf.bytes[j] = j + RATE < BUF ? buf.bytes[j + RATE] : zero;
    return out;
}
}  // namespace

void keccak256_round_function_entry_point(CS& cs, uint32_t limit) {
    G g(cs);
    // ---- alloc_ignoring_outputs ----
    Boolean start_flag = g.alloc_bool();
    Queue4 obs_req = alloc_queue4(g);
    auto obs_mem = g.alloc_queue_state<12>();
    std::array<Boolean, 4> f_flags;
    for (auto& b : f_flags) b = g.alloc_bool();
    auto alloc_bytes = [&](zk_var* dst, int n) {  // UInt8::allocate x n: pair range checks
        for (int i = 0; i < n; ++i) dst[i] = g.next_input();
        for (int i = 0; i + 1 < n; i += 2) g.range_check_u8_pair(dst[i], dst[i + 1]);
        if (n & 1) g.range_check_u8_pair(dst[n - 1], dst[n - 1]);
    };
    std::array<zk_var, 200> f_state_ref;  // reference order: [i][j][k], lane x=i, y=j
    alloc_bytes(f_state_ref.data(), 200);
    UInt32 f_ts_read = g.alloc_u32_checked(), f_ts_write = g.alloc_u32_checked();
    std::array<UInt32, 5> f_params;
    for (auto& x : f_params) x = g.alloc_u32_checked();
    Boolean f_needs_full = g.alloc_bool();
    std::array<zk_var, BUF + 1> f_buffer;  // bytes, filled
    alloc_bytes(f_buffer.data(), BUF + 1);
    Queue4 f_req = alloc_queue4(g);
    auto f_mem = g.alloc_queue_state<12>();

    for (auto h : obs_req.head) g.enforce_zero(h);
    g.enforce_trivial_head(obs_mem);
    Queue4 req_state = select_queue4(g, start_flag, obs_req, f_req);
    auto mem_state = g.select(start_flag, obs_mem, f_mem);

    // starting FSM state (placeholder with read_precompile_call = true) and the `can_finish_immediatelly` masking
    zk_var zero = g.zero();
    Boolean b_true = g.bool_const(true), b_false = g.bool_const(false);
    Boolean rpc0 = g.select(start_flag, b_true, f_flags[0]);
    Boolean ruw0 = g.select(start_flag, b_false, f_flags[1]);
    Boolean padding0 = g.select(start_flag, b_false, f_flags[2]);
    Boolean completed0 = g.select(start_flag, b_false, f_flags[3]);
    Boolean not_start = g.negated(start_flag);
    std::array<zk_var, 200> state0;  // loop order: x + 5y
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j)
            for (int k = 0; k < 8; ++k) state0[8 * (i + 5 * j) + k] = masked(g, f_state_ref[8 * (5 * i + j) + k], not_start);
    zk_var ts_read0 = masked(g, f_ts_read.v, not_start), ts_write0 = masked(g, f_ts_write.v, not_start);
    std::array<zk_var, 6> params0;
    for (int i = 0; i < 5; ++i) params0[i] = masked(g, f_params[i].v, not_start);
    params0[5] = masked(g, f_needs_full.v, not_start);
    std::array<zk_var, BUF + 1> buffer0;
    for (int i = 0; i <= BUF; ++i) buffer0[i] = masked(g, f_buffer[i], not_start);
    Boolean can_finish = g.b_and(rpc0, g.is_zero(req_state.length.v));
    Boolean not_can_finish = g.negated(can_finish);
    rpc0 = g.b_and(rpc0, not_can_finish);
    ruw0 = g.b_and(ruw0, not_can_finish);
    completed0 = g.b_or(completed0, can_finish);

    cs.side_begin();
    std::vector<zk_var> obs_in = obs_req.flatten();
    for (auto v : g.flatten(obs_mem)) obs_in.push_back(v);
    std::vector<zk_var> fsm_in;
    for (auto& b : f_flags) fsm_in.push_back(b.v);
    for (auto v : f_state_ref) fsm_in.push_back(v);
    fsm_in.push_back(f_ts_read.v);
    fsm_in.push_back(f_ts_write.v);
    for (auto& x : f_params) fsm_in.push_back(x.v);
    fsm_in.push_back(f_needs_full.v);
    for (auto v : f_buffer) fsm_in.push_back(v);
    for (auto v : f_req.flatten()) fsm_in.push_back(v);
    for (auto v : g.flatten(f_mem)) fsm_in.push_back(v);
    auto c_obs_in = g.commit_encoding(obs_in);
    auto c_fsm_in = g.commit_encoding(fsm_in);

    std::array<zk_var, KF_CARRIED> init{};
    {
        int n = 0;
        init[n++] = rpc0.v; init[n++] = ruw0.v; init[n++] = padding0.v; init[n++] = completed0.v;
        for (auto v : state0) init[n++] = v;
        init[n++] = ts_read0; init[n++] = ts_write0;
        for (auto v : params0) init[n++] = v;
        for (auto v : buffer0) init[n++] = v;
        for (auto v : req_state.head) init[n++] = v;
        init[n++] = req_state.length.v;
        for (auto& t : mem_state.tail) init[n++] = t.v;
        init[n++] = mem_state.length.v;
    }

    // =========================== loop body (mod.rs:228-667), recorded once ===========================
    cs.native_seed_kind = 3;  // the carried FSM state has a native walker (kernels_fsm_seed.hpp)
    cs.loop_begin(limit);
    K kk(g);
    std::array<zk_var, KF_CARRIED> in{}, out{};
    for (int i = 0; i < KF_CARRIED; ++i) {
        in[i] = g.next_input();
        cs.link(ZK_LINK_FIRST, in[i], init[i]);
    }
    Boolean rpc{in[0]}, ruw{in[1]}, padding_round{in[2]}, completed{in[3]};
    std::array<Lane, 25> st;
    for (int l = 0; l < 25; ++l)
        for (int k = 0; k < 8; ++k) st[l][k] = in[4 + 8 * l + k];
    UInt32 ts_read{in[204]}, ts_write{in[205]};
    UInt32 input_page{in[206]}, byte_offset{in[207]}, byte_length{in[208]}, output_page{in[209]}, output_word_offset{in[210]};
    Boolean needs_full_padding_round{in[211]};
    ByteBuffer buf;
    for (int j = 0; j < BUF; ++j) buf.bytes[j] = in[212 + j];
    buf.filled = in[404];
    std::array<zk_var, 4> req_head = {in[405], in[406], in[407], in[408]};
    UInt32 req_len{in[409]};
    std::array<zk_var, 12> mem_tail;
    for (int i = 0; i < 12; ++i) mem_tail[i] = in[410 + i];
    UInt32 mem_len{in[422]};
    Boolean l_false = g.bool_const(false), l_true = g.bool_const(true);
    zk_var l_one = g.one();

    // pop the request (mod.rs:262-283)
    Boolean req_empty = g.is_zero(req_len.v);
    conditionally_enforce_false(g, req_empty, rpc);
    LogQuery call = allocate_log_query(g);
    auto call_enc = encode_log_query(g, call);
    queue4_pop(g, req_head, req_len, call_enc, rpc);
    conditionally_enforce_equal(g, rpc, call.aux_byte.v, g.constant(PRECOMPILE_AUX_BYTE));
    for (int i = 0; i < 5; ++i)
        conditionally_enforce_equal(g, rpc, call.address[i].v, g.constant(i == 0 ? KECCAK_PRECOMPILE_ADDRESS : 0));
    // Keccak256PrecompileCallParams::from_encoding (mod.rs:68-90)
    UInt32 call_byte_length = call.key.inner[1];
    Boolean call_needs_full = g.is_zero(g.div_by_constant(call_byte_length, RATE).second.v);
    byte_offset = g.select(rpc, call.key.inner[0], byte_offset);
    byte_length = g.select(rpc, call_byte_length, byte_length);
    output_word_offset = g.select(rpc, call.key.inner[2], output_word_offset);
    input_page = g.select(rpc, call.key.inner[4], input_page);
    output_page = g.select(rpc, call.key.inner[5], output_page);
    needs_full_padding_round = g.select(rpc, call_needs_full, needs_full_padding_round);
    ts_read = g.select(rpc, call.timestamp, ts_read);
    ts_write = g.select(rpc, g.increment_unchecked(ts_read), ts_write);

    // mod.rs:318-350
    Boolean reset_buffer = g.b_or(rpc, completed);
    Boolean new_request_is_zero_length = g.is_zero(call_byte_length.v);
    Boolean have_read_zero_length_call = g.b_and(rpc, new_request_is_zero_length);
    Boolean have_read_non_zero_length_call = g.b_and(rpc, g.negated(new_request_is_zero_length));
    ruw = g.b_or(ruw, have_read_non_zero_length_call);
    padding_round = g.b_or(padding_round, have_read_zero_length_call);
    Boolean keep = g.negated(reset_buffer);
    for (auto& b : buf.bytes) b = masked(g, b, keep);
    buf.filled = masked(g, buf.filled, keep);
    for (auto& lane : st)
        for (auto& b : lane) b = masked(g, b, keep);

    // six conditional unaligned reads (mod.rs:392-495)
    for (int r = 0; r < READS; ++r) {
        auto [aligned_index, unalignment] = g.div_by_constant(byte_offset, 32);
        zk_var at_most = g.sub(g.constant(32), unalignment.v);
        auto [diff, uf] = g.overflowing_sub_with_borrow_in(byte_length, UInt32{at_most}, l_false);
        (void)diff;
        UInt32 meaningful = g.select(uf, byte_length, UInt32{at_most});
        Boolean have_something = g.negated(g.is_zero(meaningful.v));
        Boolean enough_space = can_fill_bytes(g, buf, meaningful.v);
        Boolean should_read = g.multi_and({have_something, enough_space, ruw});
        MemoryQuery q;
        q.timestamp = ts_read; q.memory_page = input_page; q.index = aligned_index;
        q.rw_flag = l_false; q.is_ptr = l_false;
        std::array<std::array<UInt8, 4>, 8> vb;
        for (int i = 0; i < 8; ++i) {
            q.value.inner[i] = UInt32{g.next_input()};
            vb[i] = g.decompose_into_bytes(q.value.inner[i]);
        }
        auto enc = encode_memory_query_with_bytes(g, q, vb[5], vb[6], vb[7]);
        full_queue_push(g, mem_tail, mem_len, enc, should_read);
        zk_var new_offset = g.add(byte_offset.v, meaningful.v), new_length = g.sub(byte_length.v, meaningful.v);
        g.range_check_u32(new_offset);  // add_no_overflow / sub_no_overflow
        g.range_check_u32(new_length);
        byte_offset = g.select(should_read, UInt32{new_offset}, byte_offset);
        byte_length = g.select(should_read, UInt32{new_length}, byte_length);
        zk_var bytes_to_fill = masked(g, meaningful.v, should_read);
        std::array<zk_var, 32> be;  // value.to_be_bytes()
        for (int m = 0; m < 32; ++m) be[m] = vb[7 - m / 4][3 - m % 4].v;
        fill_with_bytes(g, buf, be, unalignment.v, bytes_to_fill);
    }

    // padding and the permutation (mod.rs:497-590)
    Boolean zero_bytes_left = g.is_zero(byte_length.v);
    zk_var currently_filled = buf.filled;
    Boolean do_one_byte_of_padding = g.equals(currently_filled, g.constant(RATE - 1));
    auto input = consume(g, buf);
    Boolean buffer_now_empty = g.is_zero(buf.filled);
    Boolean apply_padding = g.multi_and({zero_bytes_left, buffer_now_empty, ruw, g.negated(needs_full_padding_round)});
    {
        zk_var tmp = currently_filled, pad_constant = g.constant(0x01);
        for (int j = 0; j < RATE - 1; ++j) {
            Boolean pad_this_byte = g.is_zero(tmp);
            input[j] = g.select(g.b_and(apply_padding, pad_this_byte), pad_constant, input[j]);
            tmp = g.sub(tmp, l_one);
        }
        zk_var last = g.select(do_one_byte_of_padding, g.constant(0x81), g.constant(0x80));
        input[RATE - 1] = g.select(apply_padding, last, input[RATE - 1]);
        for (int j = 0; j < RATE; ++j) {
            uint64_t full = j == 0 ? 0x01 : (j == RATE - 1 ? 0x80 : 0x00);
            input[j] = g.select(padding_round, g.constant(full), input[j]);
        }
    }
    // keccak256_absorb_and_run_permutation (mod.rs:796-838)
    for (int j = RATE; j < 200; j += 2) g.range_check_u8_pair(st[j / 8][j % 8], st[(j + 1) / 8][(j + 1) % 8]);
    kk.absorb_and_permute(st, input.data());

    // conditional write of the digest (mod.rs:592-627): UInt256::from_be_bytes(squeezed)
    Boolean write_result = g.b_or(apply_padding, padding_round);
    {
        MemoryQuery q;
        q.timestamp = ts_write; q.memory_page = output_page; q.index = output_word_offset;
        q.rw_flag = l_true; q.is_ptr = l_false;
        std::array<std::array<UInt8, 4>, 8> vb;
        for (int i = 0; i < 8; ++i) {  // limb 7-i holds squeezed[4i..4i+4] big-endian
            zk_var sq[4];
            for (int k = 0; k < 4; ++k) sq[k] = st[(4 * i + k) / 8][(4 * i + k) % 8];
            q.value.inner[7 - i] = UInt32{g.linear_combination({{sq[3], 1}, {sq[2], 1ull << 8}, {sq[1], 1ull << 16}, {sq[0], 1ull << 24}})};
            for (int k = 0; k < 4; ++k) vb[7 - i][k] = UInt8{sq[3 - k]};
        }
        auto enc = encode_memory_query_with_bytes(g, q, vb[5], vb[6], vb[7]);
        full_queue_push(g, mem_tail, mem_len, enc, write_result);
    }

    // FSM update (mod.rs:631-664)
    Boolean input_is_empty = g.is_zero(req_len.v);
    Boolean nothing_left = g.b_and(write_result, input_is_empty);
    Boolean process_next = g.b_and(write_result, g.negated(input_is_empty));
    rpc = process_next;
    completed = g.b_or(nothing_left, completed);
    Boolean needs_full_padding = g.multi_and({ruw, zero_bytes_left, buffer_now_empty, needs_full_padding_round});
    padding_round = needs_full_padding;
    ruw = g.negated(g.multi_or({rpc, padding_round, completed}));

    {
        int n = 0;
        out[n++] = rpc.v; out[n++] = ruw.v; out[n++] = padding_round.v; out[n++] = completed.v;
        for (auto& lane : st)
            for (auto b : lane) out[n++] = b;
        out[n++] = ts_read.v; out[n++] = ts_write.v;
        out[n++] = input_page.v; out[n++] = byte_offset.v; out[n++] = byte_length.v; out[n++] = output_page.v;
        out[n++] = output_word_offset.v; out[n++] = needs_full_padding_round.v;
        for (auto b : buf.bytes) out[n++] = b;
        out[n++] = buf.filled;
        for (auto v : req_head) out[n++] = v;
        out[n++] = req_len.v;
        for (auto v : mem_tail) out[n++] = v;
        out[n++] = mem_len.v;
    }
    for (int i = 0; i < KF_CARRIED; ++i) cs.link(ZK_LINK_CARRY, in[i], out[i]);
    cs.loop_end();

    // =========================== epilogue (mod.rs:669, 745-793) ===========================
    std::array<zk_var, KF_CARRIED> fin;
    for (int i = 0; i < KF_CARRIED; ++i) fin[i] = cs.loop_last(out[i]);
    Queue4 req_final = req_state;
    for (int i = 0; i < 4; ++i) req_final.head[i] = fin[405 + i];
    req_final.length = UInt32{fin[409]};
    queue4_enforce_consistency(g, req_final);
    auto mem_final = mem_state;
    for (int i = 0; i < 12; ++i) mem_final.tail[i] = Num{fin[410 + i]};
    mem_final.length = UInt32{fin[422]};
    Boolean done{fin[3]};

    Num zero_num = g.num_const(0);
    std::vector<zk_var> obs_out;
    for (auto v : g.flatten(mem_final)) obs_out.push_back(g.select(done, v, zero_num.v));
    std::vector<zk_var> fsm_out = {fin[0], fin[1], fin[2], fin[3]};
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j)
            for (int k = 0; k < 8; ++k) fsm_out.push_back(fin[4 + 8 * (i + 5 * j) + k]);
    for (int i = 204; i < 405; ++i) fsm_out.push_back(fin[i]);  // timestamps, params, buffer
    for (auto v : req_final.flatten()) fsm_out.push_back(v);
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
    (void)zero;
}

}  // namespace zkgl
