// circuits/memory_query.hpp — MemoryQuery, its 8-element encoding and the full-state (12-element tail)
// memory queue ops shared by ram_permutation and the precompile circuits.
//   MemoryQuery / flatten order : /root/reference/src/base_structures/memory_query/mod.rs:17-30
//   MemoryQuery::encode          : mod.rs:103-221 (13 variables -> 8 field elements)
//   full-state queue push rule   : /root/reference/src/main_vm/utils.rs:194-213 (encoding overwrites the
//                                  rate part of the WHOLE 12-element tail, one permutation, new tail = state)
#pragma once
#include "../gadgets.hpp"

namespace zkgl {

constexpr int MEMORY_QUERY_PACKED_WIDTH = 8;

struct MemoryQuery {
    UInt32 timestamp, memory_page, index;
    Boolean rw_flag, is_ptr;
    UInt256 value;
};

inline MemoryQuery allocate_memory_query(G& g) {  // CSAllocatable derive: every field's own `allocate`
    MemoryQuery q;
    q.timestamp = g.alloc_u32_checked();
    q.memory_page = g.alloc_u32_checked();
    q.index = g.alloc_u32_checked();
    q.rw_flag = g.alloc_bool();
    q.is_ptr = g.alloc_bool();
    q.value = g.alloc_u256_checked();
    return q;
}

// encode with the byte decompositions of value limbs 5, 6, 7 supplied by the caller (mod.rs:131-210)
inline std::array<zk_var, 8> encode_memory_query_with_bytes(G& g, const MemoryQuery& q, const std::array<UInt8, 4>& d5,
                                                            const std::array<UInt8, 4>& d6, const std::array<UInt8, 4>& d7) {
    const uint64_t S32 = 1ull << 32, S33 = 1ull << 33, S40 = 1ull << 40, S48 = 1ull << 48;
    zk_var v0 = q.timestamp.v, v1 = q.memory_page.v;
    zk_var v2 = g.linear_combination({{q.index.v, 1}, {q.rw_flag.v, S32}, {q.is_ptr.v, S33}});
    zk_var v3 = g.linear_combination({{q.value.inner[0].v, 1}, {d5[0].v, S32}, {d5[1].v, S40}, {d5[2].v, S48}});
    zk_var v4 = g.linear_combination({{q.value.inner[1].v, 1}, {d5[3].v, S32}, {d6[0].v, S40}, {d6[1].v, S48}});
    zk_var v5 = g.linear_combination({{q.value.inner[2].v, 1}, {d6[2].v, S32}, {d6[3].v, S40}, {d7[0].v, S48}});
    zk_var v6 = g.linear_combination({{q.value.inner[3].v, 1}, {d7[1].v, S32}, {d7[2].v, S40}, {d7[3].v, S48}});
    zk_var v7 = q.value.inner[4].v;
    return {v0, v1, v2, v3, v4, v5, v6, v7};
}

// MemoryQuery::encode — src/base_structures/memory_query/mod.rs:103-221
inline std::array<zk_var, 8> encode_memory_query(G& g, const MemoryQuery& q) {
    auto d5 = g.decompose_into_bytes(q.value.inner[5]);
    auto d6 = g.decompose_into_bytes(q.value.inner[6]);
    auto d7 = g.decompose_into_bytes(q.value.inner[7]);
    return encode_memory_query_with_bytes(g, q, d5, d6, d7);
}

// FullStateCircuitQueue::pop_front once the item is allocated and encoded (boojum [EXT]; by symmetry with push the
// head advances along the same chain): head <- round_function(enc | head[8..12]) when `execute`, length -= execute
inline void full_queue_pop(G& g, std::array<zk_var, 12>& head, UInt32& length, const std::array<zk_var, 8>& enc, Boolean execute) {
    std::array<zk_var, 12> st;
    for (int i = 0; i < 8; ++i) st[i] = enc[i];
    for (int i = 8; i < 12; ++i) st[i] = head[i];
    auto nh = g.compute_round_function(st);
    for (int i = 0; i < 12; ++i) head[i] = g.select(execute, nh[i], head[i]);
    length = g.select(execute, UInt32{g.sub(length.v, g.one())}, length);
}

// FullStateCircuitQueue::push: tail <- round_function(enc | tail[8..12]) when `execute`, length += execute
inline void full_queue_push(G& g, std::array<zk_var, 12>& tail, UInt32& length, const std::array<zk_var, 8>& enc, Boolean execute) {
    std::array<zk_var, 12> st;
    for (int i = 0; i < 8; ++i) st[i] = enc[i];
    for (int i = 8; i < 12; ++i) st[i] = tail[i];
    auto nt = g.compute_round_function(st);
    for (int i = 0; i < 12; ++i) tail[i] = g.select(execute, nt[i], tail[i]);
    length = UInt32{g.add(length.v, execute.v)};
}

}  // namespace zkgl
