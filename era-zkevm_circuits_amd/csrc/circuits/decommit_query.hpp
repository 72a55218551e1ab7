// circuits/decommit_query.hpp — DecommitQuery and its 8-element encoding
// (/root/reference/src/base_structures/decommit_query/mod.rs:22-113, flatten order :137-155), shared by
// sort_decommits.cpp and code_unpacker.cpp.
#pragma once
#include "../gadgets.hpp"

namespace zkgl {

struct DecommitQuery {
    UInt256 code_hash;
    UInt32 page;
    Boolean is_first;
    UInt32 timestamp;
    std::vector<zk_var> flatten() const {
        std::vector<zk_var> o;
        for (auto& l : code_hash.inner) o.push_back(l.v);
        o.push_back(page.v); o.push_back(is_first.v); o.push_back(timestamp.v);
        return o;
    }
};
inline DecommitQuery allocate_decommit_query(G& g) {
    DecommitQuery q;
    q.code_hash = g.alloc_u256_checked();
    q.page = g.alloc_u32_checked();
    q.is_first = g.alloc_bool();
    q.timestamp = g.alloc_u32_checked();
    return q;
}
inline DecommitQuery unflatten_decommit_query(const zk_var* f) {
    DecommitQuery q;
    for (int i = 0; i < 8; ++i) q.code_hash.inner[i] = UInt32{f[i]};
    q.page = UInt32{f[8]}; q.is_first = Boolean{f[9]}; q.timestamp = UInt32{f[10]};
    return q;
}
// DecommitQuery::encode — src/base_structures/decommit_query/mod.rs:33-113
inline std::array<zk_var, 8> encode_decommit_query(G& g, const DecommitQuery& q) {
    const uint64_t S32 = 1ull << 32, S40 = 1ull << 40, S48 = 1ull << 48;
    auto p = g.decompose_into_bytes(q.page);
    auto t = g.decompose_into_bytes(q.timestamp);
    zk_var v0 = g.linear_combination({{q.code_hash.inner[0].v, 1}, {p[0].v, S32}, {p[1].v, S40}, {p[2].v, S48}});
    zk_var v1 = g.linear_combination({{q.code_hash.inner[1].v, 1}, {p[3].v, S32}, {t[0].v, S40}, {t[1].v, S48}});
    zk_var v2 = g.linear_combination({{q.code_hash.inner[2].v, 1}, {t[2].v, S32}, {t[3].v, S40}, {q.is_first.v, S48}});
    return {v0, v1, v2, q.code_hash.inner[3].v, q.code_hash.inner[4].v, q.code_hash.inner[5].v, q.code_hash.inner[6].v,
            q.code_hash.inner[7].v};
}
}  // namespace zkgl
