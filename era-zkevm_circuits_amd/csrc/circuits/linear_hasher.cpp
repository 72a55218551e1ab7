// circuits/linear_hasher.cpp — host-side mirror of /root/reference/src/linear_hasher/mod.rs:35-212 (linear_hasher_entry_point)
// and LogQuery::into_bytes (src/base_structures/log_query/mod.rs:645-686): Keccak-256 of the 88-byte serialisations of every
// L2->L1 message log of a queue.
//
// The reference unrolls `limit` cycles with a *static* byte buffer: every cycle appends 88 bytes, a 136-byte block is
// absorbed whenever the buffer holds one, and a padded "last round" absorb is conditionally applied after every cycle.
// 17 cycles x 88 B = 11 blocks x 136 B, so the buffer is empty again after every 17 cycles: the loop body recorded here is
// one such period (17 pops, 11 conditional absorbs, 17 conditional last-round absorbs) and `limit` must be a multiple of 17.
//
// INPUT STREAMS
//   outer (10 words): start_flag | observable_input.queue_state {head[4], tail[4], length}
//   loop (818 words per 17-cycle period): carried[206] = keccak state[200] (byte k of lane x+5y at 8(x+5y)+k) | queue head[4],
//       length | done   then 17 x popped LogQuery[36]
#include "keccak_gadget.hpp"
#include "log_query.hpp"

namespace zkgl {

void keccak_configure(CS& cs);

namespace {
constexpr int RATE = 136, PERIOD = 17, CARRIED = 206;  // 88 bytes per serialised query

// keccak256_conditionally_absorb_and_run_permutation (boojum [EXT]): state <- cond ? f(state ^ block) : state
void conditionally_absorb(G& g, K& k, Boolean cond, std::array<Lane, 25>& st, const std::array<zk_var, RATE>& block) {
    std::array<Lane, 25> next = st;
    k.absorb_and_permute(next, block.data());
    for (int l = 0; l < 25; ++l)
        for (int b = 0; b < 8; ++b) st[l][b] = g.select(cond, next[l][b], st[l][b]);
}

// LogQuery::into_bytes — src/base_structures/log_query/mod.rs:647-686
std::vector<zk_var> into_bytes(G& g, const LogQuery& q) {
    std::vector<zk_var> out = {q.shard_id.v, q.is_service.v};
    auto be = [&](UInt32 x) {
        auto b = g.decompose_into_bytes(x);
        return std::array<zk_var, 4>{b[3].v, b[2].v, b[1].v, b[0].v};
    };
    auto tx = be(q.tx_number_in_block);
    g.enforce_zero(tx[0]);  // "we truncated, so let's enforce that those were unused"
    g.enforce_zero(tx[1]);
    out.push_back(tx[2]);
    out.push_back(tx[3]);
    for (int i = 4; i >= 0; --i)
        for (auto v : be(q.address[i])) out.push_back(v);
    for (int i = 7; i >= 0; --i)
        for (auto v : be(q.key.inner[i])) out.push_back(v);
    for (int i = 7; i >= 0; --i)
        for (auto v : be(q.written_value.inner[i])) out.push_back(v);
    return out;
}
}  // namespace

void linear_hasher_configure(CS& cs) { keccak_configure(cs); }

void linear_hasher_entry_point(CS& cs, uint32_t limit) {
    if (limit == 0 || limit % PERIOD) throw ZkError(ZK_ERR_INVALID, "linear_hasher: limit must be a positive multiple of 17 (88 B per cycle vs 136 B blocks)");
    G g(cs);
    Boolean start_flag = g.alloc_bool();
    Queue4 queue = alloc_queue4(g);
    g.enforce_equal(start_flag.v, g.one());                 // mod.rs:62-63
    for (auto h : queue.head) g.enforce_zero(h);            // enforce_trivial_head
    Boolean no_work = g.is_zero(queue.length.v);            // `done = queue.is_empty(); no_work = done`
    zk_var outer_zero = g.zero();
    cs.side_begin();
    auto c_obs_in = g.commit_encoding(queue.flatten());

    cs.loop_begin(limit / PERIOD);
    K k(g);
    std::vector<zk_var> in, out;
    auto carry_in = [&](zk_var first) {
        zk_var v = g.next_input();
        cs.link(ZK_LINK_FIRST, v, first);
        in.push_back(v);
        return v;
    };
    std::array<Lane, 25> st;
    for (auto& lane : st)
        for (auto& b : lane) b = carry_in(outer_zero);
    std::array<zk_var, 4> head;
    for (int i = 0; i < 4; ++i) head[i] = carry_in(queue.head[i]);
    UInt32 len{carry_in(queue.length.v)};
    Boolean done{carry_in(no_work.v)};
    for (int j = RATE; j < 200; j += 2) g.range_check_u8_pair(st[j / 8][j % 8], st[(j + 1) / 8][(j + 1) % 8]);

    std::vector<zk_var> buffer;
    for (int c = 0; c < PERIOD; ++c) {
        Boolean should_pop = g.negated(g.is_zero(len.v));
        LogQuery q = allocate_log_query(g);
        queue4_pop(g, head, len, encode_log_query(g, q), should_pop);
        Boolean is_last_serialization = g.b_and(should_pop, g.is_zero(len.v));
        for (auto v : into_bytes(g, q)) buffer.push_back(v);
        Boolean continue_to_absorb = g.negated(done);
        if (buffer.size() >= (size_t)RATE) {
            std::array<zk_var, RATE> block;
            for (int j = 0; j < RATE; ++j) block[j] = buffer[j];
            buffer.erase(buffer.begin(), buffer.begin() + RATE);
            conditionally_absorb(g, k, continue_to_absorb, st, block);
        }
        {
            Boolean absorb_as_last_round = g.b_and(continue_to_absorb, is_last_serialization);
            std::array<zk_var, RATE> last;
            const size_t tail = buffer.size();
            for (size_t j = 0; j < (size_t)RATE; ++j) last[j] = j < tail ? buffer[j] : g.zero();
            if (tail == (size_t)RATE - 1) last[tail] = g.constant(0x81);
            else { last[tail] = g.constant(0x01); last[RATE - 1] = g.constant(0x80); }
            conditionally_absorb(g, k, absorb_as_last_round, st, last);
        }
        done = g.b_or(done, is_last_serialization);
    }
    if (!buffer.empty()) throw ZkError(ZK_ERR_INVALID, "linear_hasher: period does not drain the buffer");
    for (auto& lane : st)
        for (auto b : lane) out.push_back(b);
    for (auto v : head) out.push_back(v);
    out.push_back(len.v);
    out.push_back(done.v);
    for (int i = 0; i < CARRIED; ++i) cs.link(ZK_LINK_CARRY, in[i], out[i]);
    cs.loop_end();

    // epilogue (mod.rs:168-211)
    Queue4 fin = queue;
    for (int i = 0; i < 4; ++i) fin.head[i] = cs.loop_last(out[200 + i]);
    fin.length = UInt32{cs.loop_last(out[204])};
    queue4_enforce_consistency(g, fin);
    Boolean completed = g.is_zero(fin.length.v);
    g.enforce_equal(completed.v, g.one());
    static const uint8_t EMPTY_HASH[32] = {0xc5, 0xd2, 0x46, 0x01, 0x86, 0xf7, 0x23, 0x3c, 0x92, 0x7e, 0x7d, 0xb2, 0xdc, 0xc7, 0x03, 0xc0,
                                           0xe5, 0x00, 0xb6, 0x53, 0xca, 0x82, 0x27, 0x3b, 0x7b, 0xfa, 0xd8, 0x04, 0x5d, 0x85, 0xa4, 0x70};
    std::vector<zk_var> obs_out;
    for (int j = 0; j < 32; ++j) obs_out.push_back(g.select(no_work, g.constant(EMPTY_HASH[j]), cs.loop_last(out[j])));
    Num zero_num = g.num_const(0);
    auto c_obs_out = g.commit_encoding(obs_out);
    auto c_fsm_in = g.commit_encoding({});
    auto c_fsm_out = g.commit_encoding({});
    std::vector<zk_var> compact = {start_flag.v, completed.v};
    for (int i = 0; i < 4; ++i) compact.push_back(c_obs_in[i].v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completed, c_obs_out[i], zero_num).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(start_flag, zero_num, c_fsm_in[i]).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(completed, zero_num, c_fsm_out[i]).v);
    auto input_commitment = g.commit_encoding(compact);
    for (auto& el : input_commitment) cs.place_gate(ZK_GATE_PUBLIC_INPUT, &el.v, 1, nullptr, 0);
}

}  // namespace zkgl
