// circuits/log_query.hpp — LogQuery, its 20-element encoding and the 4-element-tail CircuitQueue ops,
// shared by storage_validity_by_grand_product and log_sorter.
//   LogQuery / flatten order : /root/reference/src/base_structures/log_query/mod.rs:21-34, 60-99
//   LogQuery::encode          : mod.rs:121-517  (36 variables -> 20 field elements)
//   queue push rule           : /root/reference/src/main_vm/opcodes/log.rs:508-609 (3 absorb rounds from the
//                               EMPTY sponge state, tail mixed into the third round, new tail = first 4)
//   pop_front                 : boojum [EXT]; by symmetry with push the head advances along the same chain
#pragma once
#include "../gadgets.hpp"

namespace zkgl {

struct LogQuery {
    std::array<UInt32, 5> address;  // UInt160
    UInt256 key, read_value, written_value;
    UInt8 aux_byte;
    Boolean rw_flag, rollback, is_service;
    UInt8 shard_id;
    UInt32 tx_number_in_block, timestamp;
};

struct Queue4 {  // QueueState<F, QUEUE_STATE_WIDTH = 4>
    std::array<zk_var, 4> head, tail;
    UInt32 length;
    std::vector<zk_var> flatten() const {
        std::vector<zk_var> o(head.begin(), head.end());
        o.insert(o.end(), tail.begin(), tail.end());
        o.push_back(length.v);
        return o;
    }
};

inline UInt8 alloc_u8_checked(G& g) {
    zk_var v = g.next_input();
    g.range_check_u8_pair(v, g.zero());
    return {v};
}

inline Queue4 alloc_queue4(G& g) {
    Queue4 q;
    for (auto& h : q.head) h = g.alloc_num().v;
    for (auto& t : q.tail) t = g.alloc_num().v;
    q.length = g.alloc_u32_checked();
    return q;
}

inline Queue4 select_queue4(G& g, Boolean s, const Queue4& a, const Queue4& b) {
    Queue4 r;
    for (int i = 0; i < 4; ++i) r.head[i] = g.select(s, a.head[i], b.head[i]);
    for (int i = 0; i < 4; ++i) r.tail[i] = g.select(s, a.tail[i], b.tail[i]);
    r.length = g.select(s, a.length, b.length);
    return r;
}

// CSAllocatable derive: every field's own `allocate`, in declaration order (= flatten order, 36 words)
inline LogQuery allocate_log_query(G& g) {
    LogQuery q;
    for (auto& l : q.address) l = g.alloc_u32_checked();
    q.key = g.alloc_u256_checked();
    q.read_value = g.alloc_u256_checked();
    q.written_value = g.alloc_u256_checked();
    q.aux_byte = alloc_u8_checked(g);
    q.rw_flag = g.alloc_bool();
    q.rollback = g.alloc_bool();
    q.is_service = g.alloc_bool();
    q.shard_id = alloc_u8_checked(g);
    q.tx_number_in_block = g.alloc_u32_checked();
    q.timestamp = g.alloc_u32_checked();
    return q;
}

// LogQuery::encode (mod.rs:121-517): the 32 key bytes followed by the 20 address bytes are spread, three
// per element, over read_value[0..8], written_value[0..8], timestamp and tx_number_in_block.
inline std::array<zk_var, 20> encode_log_query(G& g, const LogQuery& q) {
    const uint64_t S32 = 1ull << 32, S40 = 1ull << 40, S48 = 1ull << 48;
    std::vector<zk_var> bytes;  // key_bytes[0..8][0..4] then address_bytes[0..5][0..4]
    for (int i = 0; i < 8; ++i)
        for (auto& b : g.decompose_into_bytes(q.key.inner[i])) bytes.push_back(b.v);
    for (int i = 0; i < 5; ++i)
        for (auto& b : g.decompose_into_bytes(q.address[i])) bytes.push_back(b.v);
    std::array<zk_var, 20> v;
    for (int i = 0; i < 17; ++i) {
        zk_var base = i < 8 ? q.read_value.inner[i].v : (i < 16 ? q.written_value.inner[i - 8].v : q.timestamp.v);
        v[i] = g.linear_combination({{base, 1}, {bytes[3 * i], S32}, {bytes[3 * i + 1], S40}, {bytes[3 * i + 2], S48}});
    }
    v[17] = g.linear_combination({{q.tx_number_in_block.v, 1}, {bytes[51], S32}, {q.aux_byte.v, S40}, {q.shard_id.v, S48}});
    v[18] = g.linear_combination({{q.rw_flag.v, 1}, {q.is_service.v, 2}});
    v[19] = q.rollback.v;
    return v;
}

// three absorb-with-replacement rounds of a 20-element encoding into a 4-element queue state
inline std::array<zk_var, 4> absorb_encoding20(G& g, const std::array<zk_var, 20>& enc, const std::array<zk_var, 4>& state) {
    std::array<zk_var, 12> s = g.empty_state();
    for (int i = 0; i < 8; ++i) s[i] = enc[i];
    s = g.compute_round_function(s);
    for (int i = 0; i < 8; ++i) s[i] = enc[8 + i];
    s = g.compute_round_function(s);
    for (int i = 0; i < 4; ++i) s[i] = enc[16 + i];
    for (int i = 0; i < 4; ++i) s[4 + i] = state[i];
    s = g.compute_round_function(s);
    return {s[0], s[1], s[2], s[3]};
}

// CircuitQueue::pop_front body once the item is allocated and encoded
inline void queue4_pop(G& g, std::array<zk_var, 4>& head, UInt32& length, const std::array<zk_var, 20>& enc, Boolean execute) {
    auto nh = absorb_encoding20(g, enc, head);
    for (int i = 0; i < 4; ++i) head[i] = g.select(execute, nh[i], head[i]);
    length = g.select(execute, UInt32{g.sub(length.v, g.one())}, length);
}
// CircuitQueue::push
inline void queue4_push(G& g, std::array<zk_var, 4>& tail, UInt32& length, const std::array<zk_var, 20>& enc, Boolean execute) {
    auto nt = absorb_encoding20(g, enc, tail);
    for (int i = 0; i < 4; ++i) tail[i] = g.select(execute, nt[i], tail[i]);
    length = g.select(execute, UInt32{g.add(length.v, g.one())}, length);
}

inline void conditionally_enforce_equal(G& g, Boolean cond, zk_var a, zk_var b) {  // cond * (a - b) == 0
    g.enforce_zero(g.mul(cond.v, g.sub(a, b)));
}
inline void conditionally_enforce_false(G& g, Boolean b, Boolean cond) {  // cond * b == 0
    g.enforce_zero(g.mul(cond.v, b.v));
}
inline void queue4_enforce_consistency(G& g, const Queue4& q) {  // empty queue => head == tail
    Boolean is_empty = g.is_zero(q.length.v);
    for (int i = 0; i < 4; ++i) conditionally_enforce_equal(g, is_empty, q.head[i], q.tail[i]);
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

// produce_fs_challenges — src/utils.rs:12-78 (generic over the input length / challenge count)
template <size_t NCH>
std::array<std::array<zk_var, NCH>, 2> produce_fs_challenges(G& g, const std::vector<zk_var>& fs_input) {
    std::array<zk_var, 12> state = g.empty_state();
    state[11] = g.constant(fs_input.size());
    size_t nchunks = (fs_input.size() + 7) / 8;
    for (size_t c = 0; c < nchunks; ++c) {
        for (size_t j = 0; j < 8; ++j) {
            size_t k = 8 * c + j;
            state[j] = k < fs_input.size() ? fs_input[k] : g.zero();
        }
        state = g.compute_round_function(state);
    }
    std::array<std::array<zk_var, NCH>, 2> out;
    int can_take = 8;
    for (int r = 0; r < 2; ++r) {
        out[r][0] = g.one();
        for (size_t i = 1; i < NCH; ++i) {
            if (can_take == 0) { state = g.compute_round_function(state); can_take = 8; }
            out[r][i] = state[8 - can_take];
            --can_take;
        }
    }
    return out;
}

// accumulate_grand_products — src/utils.rs:81-137
template <size_t ENC>
void accumulate_grand_products(G& g, std::array<Num, 2>& lhs, std::array<Num, 2>& rhs,
                               const std::array<std::array<zk_var, ENC + 1>, 2>& ch, const std::array<zk_var, ENC>& lhs_enc,
                               const std::array<zk_var, ENC>& rhs_enc, Boolean should_accumulate) {
    for (int r = 0; r < 2; ++r) {
        zk_var lc = ch[r][ENC], rc = ch[r][ENC];
        for (size_t i = 0; i < ENC; ++i) {
            lc = g.fma(1, lhs_enc[i], ch[r][i], 1, lc);
            rc = g.fma(1, rhs_enc[i], ch[r][i], 1, rc);
        }
        zk_var new_lhs = g.mul(lhs[r].v, lc), new_rhs = g.mul(rhs[r].v, rc);
        lhs[r] = g.select(should_accumulate, Num{new_lhs}, lhs[r]);
        rhs[r] = g.select(should_accumulate, Num{new_rhs}, rhs[r]);
    }
}

}  // namespace zkgl
