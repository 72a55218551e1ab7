// kernels_queue_seed.hpp — seeding of a queue circuit's carried state WITHOUT a chain: ram_permutation.
//
// The reference's queue witnesses carry (element, previous tail) pairs (src/ram_permutation/input.rs:103-116): the state of the queue
// before every push, i.e. the head the sorting circuit holds before it pops that element.  With the heads taken from there
// (zk_pack_ram_witness writes them, zk_cs_set_seed_given declares them) nothing sequential is left in partial_accumulate_inner
// (src/ram_permutation/mod.rs:212-382):
//   queue lengths                  len0 - min(cycle, len0)
//   lhs / rhs grand products       multiplicative prefix scans of  ch[8] + sum_i enc_i(element) * ch[i]  over the popping cycles
//                                  (accumulate_grand_products, src/utils.rs:81-137)
//   num_nondeterministic_writes    additive prefix scan of the bootloader-heap predicate (mod.rs:259-290)
//   previous key / value / is_ptr  the sorted element of the cycle before (mod.rs:323-326)
// One workgroup per instance, a thread owns a run of consecutive cycles: local products, a workgroup scan, the words.
// Same 22 words per cycle as the cone kernels (tests/test_witness_pack.py: == native restatement).
#pragma once
#include "gl_device.hpp"
#include "vm_native.hpp"
#include "kernels_vm_seed.hpp"

namespace zkq {

using vmn::u32;
using vmn::u64;

struct RamSeedDev {
    u64* loop; u64 in_stride; u32 limit, n_instances;
    const u64* outer_store; u64 outer_n_store;
    const u32* state0_slot;   // [46] outer store slot behind the FIRST link of every carried word
    const u32* ch_slot;       // [2][8] outer store slots of challenges[r][1..8] (challenges[r][0] == 1)
    u32 bootloader_heap_page;
};
constexpr u32 RAM_CARRIED = 46, RAM_ITEM_U = 46, RAM_ITEM_S = 59;

__device__ __forceinline__ u64 ov(const RamSeedDev& a, u32 inst, u32 slot) {
    return a.outer_store[zkgeom::offset(a.outer_n_store, slot, inst)];
}
struct Item { u32 ts, page, index, rw, is_ptr; vmn::U256 value; };
__device__ __forceinline__ Item load_item(const u64* col, u64 stride, u32 first) {
    Item q;
    q.ts = (u32)col[(u64)first * stride]; q.page = (u32)col[(u64)(first + 1) * stride]; q.index = (u32)col[(u64)(first + 2) * stride];
    q.rw = (u32)col[(u64)(first + 3) * stride]; q.is_ptr = (u32)col[(u64)(first + 4) * stride];
#pragma unroll
    for (int i = 0; i < 8; ++i) q.value.l[i] = (u32)col[(u64)(first + 5 + i) * stride];
    return q;
}
// ch[8] + sum_i enc_i * ch[i], ch[0] = 1
__device__ __forceinline__ u64 term(const Item& q, const u64 ch[9]) {
    u64 enc[8];
    vmn::memory_query_encode(enc, q.ts, q.page, q.index, q.rw, q.is_ptr, q.value);
    u64 t = gl::add(ch[8], enc[0]);
#pragma unroll
    for (int i = 1; i < 8; ++i) t = gl::fma(enc[i], ch[i], t);
    return t;
}

__global__ __launch_bounds__(256) void k_ram_seed(RamSeedDev a) {
    __shared__ u64 part[2][4][256];
    __shared__ u32 cnt[2][256];
    const u32 inst = blockIdx.x, t = threadIdx.x;
    const u64 lane0 = (u64)inst * a.limit;
    const u32 n0 = (u32)ov(a, inst, a.state0_slot[13]);             // elements left in both queues
    const u64 sn0 = ov(a, inst, a.state0_slot[26]);
    u64 ch[2][9];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        ch[r][0] = 1;
#pragma unroll
        for (int i = 1; i <= 8; ++i) ch[r][i] = ov(a, inst, a.ch_slot[r * 8 + i - 1]);
    }
    const u32 per = (a.limit + 255) / 256;
    const u32 c_begin = min(t * per, a.limit), c_end = min(c_begin + per, a.limit);
    // ---- pass 1: this thread's product of terms and count of non-deterministic writes over its popping cycles
    u64 p[4] = {1, 1, 1, 1};
    u32 k = 0;
    for (u32 c = c_begin; c < c_end && c < n0; ++c) {
        const u64* col = a.loop + lane0 + c;
        const Item qu = load_item(col, a.in_stride, RAM_ITEM_U), qs = load_item(col, a.in_stride, RAM_ITEM_S);
        p[0] = gl::mul(p[0], term(qu, ch[0])); p[1] = gl::mul(p[1], term(qu, ch[1]));
        p[2] = gl::mul(p[2], term(qs, ch[0])); p[3] = gl::mul(p[3], term(qs, ch[1]));
        k += (qs.ts == 0 && qs.page == a.bootloader_heap_page && qs.rw && !qs.is_ptr) ? 1u : 0u;
    }
    // ---- inclusive scan over the 256 threads (Hillis-Steele, double-buffered)
    int cur = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) part[0][j][t] = p[j];
    cnt[0][t] = k;
    __syncthreads();
    for (u32 d = 1; d < 256; d <<= 1) {
        const int nxt = cur ^ 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) part[nxt][j][t] = t >= d ? gl::mul(part[cur][j][t], part[cur][j][t - d]) : part[cur][j][t];
        cnt[nxt][t] = t >= d ? cnt[cur][t] + cnt[cur][t - d] : cnt[cur][t];
        cur = nxt;
        __syncthreads();
    }
    // exclusive prefix of this thread, times the instance's initial accumulators
    u64 acc[4];
    const u32 acc_word[4] = {27, 28, 29, 30};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u64 init = ov(a, inst, a.state0_slot[acc_word[j]]);
        acc[j] = t ? gl::mul(init, part[cur][j][t - 1]) : init;
    }
    u64 nondet = gl::add(ov(a, inst, a.state0_slot[31]), (u64)(t ? cnt[cur][t - 1] : 0));
    // ---- pass 2: the words of every cycle of the run (cycle 0: the outer scope's words verbatim)
    Item prev;   // the sorted element of cycle c - 1
    if (c_begin > 0 && c_begin < a.limit) prev = load_item(a.loop + lane0 + c_begin - 1, a.in_stride, RAM_ITEM_S);
    for (u32 c = c_begin; c < c_end; ++c) {
        u64* col = a.loop + lane0 + c;
        auto put = [&](u32 w, u64 v) { col[(u64)w * a.in_stride] = v; };
        const u32 popped = min(c, n0);
        if (c == 0) {
            for (u32 w = 0; w < RAM_CARRIED; ++w)
                if (!((w >= 1 && w < 13) || (w >= 14 && w < 26))) put(w, ov(a, inst, a.state0_slot[w]));
        } else {
            put(0, 0);
            put(13, (u64)(n0 - popped));
            put(26, gl::sub(sn0, (u64)popped));
            put(27, acc[0]); put(28, acc[1]); put(29, acc[2]); put(30, acc[3]);
            put(31, nondet);
            put(32, prev.ts); put(33, prev.index); put(34, prev.page);
            put(35, prev.index); put(36, prev.page);
#pragma unroll
            for (int i = 0; i < 8; ++i) put(37 + i, prev.value.l[i]);
            put(45, prev.is_ptr);
        }
        const Item qu = load_item(col, a.in_stride, RAM_ITEM_U), qs = load_item(col, a.in_stride, RAM_ITEM_S);
        if (c < n0) {
            acc[0] = gl::mul(acc[0], term(qu, ch[0])); acc[1] = gl::mul(acc[1], term(qu, ch[1]));
            acc[2] = gl::mul(acc[2], term(qs, ch[0])); acc[3] = gl::mul(acc[3], term(qs, ch[1]));
            if (qs.ts == 0 && qs.page == a.bootloader_heap_page && qs.rw && !qs.is_ptr) nondet = gl::add(nondet, 1);
        }
        prev = qs;
    }
}


// ------------------------------------------------------------------------------------------------ the LogQuery sorters
// storage_validity_by_grand_product (KIND 0, reference mod.rs:508-880) and log_sorter (KIND 1, mod.rs:246-441).  The host packer has
// walked the integer state (zk_pack_storage_witness / zk_pack_log_sorter_witness with previous tails: heads, lengths, previous
// item, cell state); left here: the two grand-product accumulators per repetition (a multiplicative scan over the popping cycles of
// ch[20] + sum_i enc_i * ch[i], the unsorted encoding of storage extended by the cycle index, mod.rs:560-580) and, when the host did
// not supply it, the output queue's tail (a Poseidon2 chain over the pushes: k_tail4_chain).
struct LogqSeedDev {
    u64* loop; u64 in_stride; u32 limit, n_instances;
    const u64* outer_store; u64 outer_n_store;
    const u32* state0_slot;   // [carried] outer store slot behind the FIRST link of every carried word
    const u32* ch_slot;       // [2][20] outer store slots of challenges[r][1..20] (challenges[r][0] == 1); storage: [40] = shard_id_to_process
};
template <int KIND> struct LogqLayout;
template <> struct LogqLayout<0> {   // storage_validity: 67 carried, unsorted item, sorted record, its timestamp
    static constexpr u32 CARRIED = 67, ACC = 2, CYCLE_IDX = 6, U_LEN = 11, S_LEN = 16, OUT_TAIL = 17, OUT_LEN = 21, ITEM_U = 67, ITEM_S = 103, STS = 139;
    static constexpr u32 PREV_KEY = 35, PREV_ADDR = 43, BASE = 50, CUR = 58;
};
template <> struct LogqLayout<1> {   // log_sorter: 57 carried, two items
    static constexpr u32 CARRIED = 57, ACC = 1, U_LEN = 9, S_LEN = 14, OUT_TAIL = 15, OUT_LEN = 19, ITEM_U = 57, ITEM_S = 93, PREV_ITEM = 21;
};
__device__ __forceinline__ void load_logq(vmn::LogQ& q, const u64* col, u64 stride, u32 first) {
#pragma unroll
    for (int i = 0; i < 5; ++i) q.address[i] = (u32)col[(u64)(first + i) * stride];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        q.key.l[i] = (u32)col[(u64)(first + 5 + i) * stride]; q.read_value.l[i] = (u32)col[(u64)(first + 13 + i) * stride];
        q.written_value.l[i] = (u32)col[(u64)(first + 21 + i) * stride];
    }
    q.aux_byte = (u32)col[(u64)(first + 29) * stride]; q.rw_flag = (u32)col[(u64)(first + 30) * stride]; q.rollback = (u32)col[(u64)(first + 31) * stride];
    q.is_service = (u32)col[(u64)(first + 32) * stride]; q.shard_id = (u32)col[(u64)(first + 33) * stride]; q.tx_number = (u32)col[(u64)(first + 34) * stride];
    q.timestamp = (u32)col[(u64)(first + 35) * stride];
}
// the four factors of one cycle: (unsorted, sorted) x (repetition 0, 1); ch in LDS
template <int KIND>
__device__ __forceinline__ void logq_terms(u64 out[4], const u64* col, u64 stride, const u64 (*ch)[21]) {
    using L = LogqLayout<KIND>;
    vmn::LogQ q;
    u64 enc[20];
    load_logq(q, col, stride, L::ITEM_U);
    vmn::log_query_encode(enc, q);
    if constexpr (KIND == 0) enc[19] = gl::add(enc[19], (u64)(u32)col[(u64)L::CYCLE_IDX * stride] << 8);   // extended by the position in the unsorted queue
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        u64 t = ch[r][20];
#pragma unroll
        for (int i = 0; i < 20; ++i) t = gl::fma(enc[i], ch[r][i], t);
        out[r] = t;
    }
    load_logq(q, col, stride, L::ITEM_S);
    vmn::log_query_encode(enc, q);
    if constexpr (KIND == 0) enc[19] = gl::add(enc[19], (u64)(u32)col[(u64)L::STS * stride] << 8);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        u64 t = ch[r][20];
#pragma unroll
        for (int i = 0; i < 20; ++i) t = gl::fma(enc[i], ch[r][i], t);
        out[2 + r] = t;
    }
}
template <int KIND>
__device__ __forceinline__ bool logq_pops(const u64* col, u64 stride) {
    using L = LogqLayout<KIND>;
    const bool u = col[(u64)L::U_LEN * stride] != 0;
    if constexpr (KIND == 0) return u && col[(u64)L::S_LEN * stride] != 0;
    return u;
}

__device__ __forceinline__ u64 ovq(const LogqSeedDev& a, u32 inst, u32 slot) {
    return a.outer_store[zkgeom::offset(a.outer_n_store, slot, inst)];
}

// one workgroup per instance; a thread owns a run of consecutive cycles: local products, a workgroup scan, the four words per cycle
constexpr u32 LOGQ_TPB = 1024;
template <int KIND>
__global__ __launch_bounds__(LOGQ_TPB) void k_logq_seed(LogqSeedDev a) {
    using L = LogqLayout<KIND>;
    __shared__ u64 part[2][4][LOGQ_TPB];
    __shared__ u64 ch[2][21];
    const u32 inst = blockIdx.x, t = threadIdx.x;
    const u64 lane0 = (u64)inst * a.limit;
    if (t < 40) ch[t / 20][1 + t % 20] = ovq(a, inst, a.ch_slot[t]);
    if (t < 2) ch[t][0] = 1;
    __syncthreads();
    const u32 per = (a.limit + LOGQ_TPB - 1) / LOGQ_TPB;
    const u32 c_begin = min(t * per, a.limit), c_end = min(c_begin + per, a.limit);
    u64 p[4] = {1, 1, 1, 1};
    for (u32 c = c_begin; c < c_end; ++c) {
        const u64* col = a.loop + lane0 + c;
        if (!logq_pops<KIND>(col, a.in_stride)) continue;
        u64 f[4];
        logq_terms<KIND>(f, col, a.in_stride, ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = gl::mul(p[j], f[j]);
    }
    int cur = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) part[0][j][t] = p[j];
    __syncthreads();
    for (u32 d = 1; d < LOGQ_TPB; d <<= 1) {
        const int nxt = cur ^ 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) part[nxt][j][t] = t >= d ? gl::mul(part[cur][j][t], part[cur][j][t - d]) : part[cur][j][t];
        cur = nxt;
        __syncthreads();
    }
    u64 acc[4];   // lhs[0], lhs[1], rhs[0], rhs[1] = carried words ACC .. ACC + 3, before cycle c_begin
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u64 init = ovq(a, inst, a.state0_slot[L::ACC + j]);
        acc[j] = t ? gl::mul(init, part[cur][j][t - 1]) : init;
    }
    for (u32 c = c_begin; c < c_end; ++c) {
        u64* col = a.loop + lane0 + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) col[(u64)(L::ACC + j) * a.in_stride] = acc[j];
        if (!logq_pops<KIND>(col, a.in_stride)) continue;
        u64 f[4];
        logq_terms<KIND>(f, col, a.in_stride, ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = gl::mul(acc[j], f[j]);
    }
}


// The output queue's tail when the host did not supply it: one wavefront per instance walks the cycles 64 at a time (the length word
// of consecutive cycles is contiguous), and for every push (length grows into the next cycle) hashes the pushed item — built from
// the carried words of the pushing cycle: storage_validity's final_query(previous address / key, cell base / current value,
// should_write) (mod.rs:700-760), log_sorter's cleaned previous item (mod.rs:372-386) — onto the 4-word tail (three permutations on a
// DPP row).  chain length x ~30 us: the one sequential piece these circuits keep.
template <int KIND>
__global__ __launch_bounds__(64) void k_tail4_chain(LogqSeedDev a) {
    using L = LogqLayout<KIND>;
    __shared__ u64 enc[20];
    const u32 inst = blockIdx.x, lane = threadIdx.x;
    const u32 e = lane & 15, el = e < 12 ? e : 0;
    u64 rcf[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { rcf[r] = p2::RC[12 * r + el]; rcf[4 + r] = p2::RC[12 * (26 + r) + el]; }
    u64* const col0 = a.loop + (u64)inst * a.limit;
    u64 tail[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tail[i] = ovq(a, inst, a.state0_slot[L::OUT_TAIL + i]);
    u32 shard = 0;
    if constexpr (KIND == 0) shard = (u32)ovq(a, inst, a.ch_slot[40]);
    for (u32 c0 = 0; c0 < a.limit; c0 += 64) {
        const u32 c = c0 + lane;
        const bool valid = c < a.limit;
        const u64 len_c = valid ? col0[c + (u64)L::OUT_LEN * a.in_stride] : 0;
        const u64 len_n = c + 1 < a.limit ? col0[c + 1 + (u64)L::OUT_LEN * a.in_stride] : len_c;
        u64 mine[4] = {tail[0], tail[1], tail[2], tail[3]};
        unsigned long long mask = __ballot(valid && len_n != len_c);
        while (mask) {
            const u32 b = (u32)__builtin_ctzll(mask);
            mask &= mask - 1;
            const u64* col = col0 + c0 + b;
            if (lane == 0) {
                vmn::LogQ q;
                if constexpr (KIND == 0) {
                    bool differ = false;
                    for (int i = 0; i < 5; ++i) q.address[i] = (u32)col[(u64)(L::PREV_ADDR + i) * a.in_stride];
                    for (int i = 0; i < 8; ++i) {
                        q.key.l[i] = (u32)col[(u64)(L::PREV_KEY + i) * a.in_stride];
                        q.read_value.l[i] = (u32)col[(u64)(L::BASE + i) * a.in_stride];
                        q.written_value.l[i] = (u32)col[(u64)(L::CUR + i) * a.in_stride];
                        differ |= q.read_value.l[i] != q.written_value.l[i];
                    }
                    q.aux_byte = 0; q.rw_flag = differ; q.rollback = 0; q.is_service = 0; q.shard_id = shard; q.tx_number = 0; q.timestamp = 0;
                } else {
                    load_logq(q, col, a.in_stride, L::PREV_ITEM);
                    for (int i = 0; i < 8; ++i) q.read_value.l[i] = 0;
                    q.aux_byte = 0; q.rw_flag = 0; q.rollback = 0; q.timestamp = 0;
                }
                u64 en[20];
                vmn::log_query_encode(en, q);
                for (int i = 0; i < 20; ++i) enc[i] = en[i];
            }
            __syncthreads();
            u64 s = e < 8 ? enc[e] : 0;
            s = zkvm::row_permute(s, e, rcf);
            if (e < 8) s = enc[8 + e];
            s = zkvm::row_permute(e < 12 ? s : 0, e, rcf);
            u64 old = tail[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) old = e == (u32)(4 + i) ? tail[i] : old;
            if (e < 4) s = enc[16 + e];
            else if (e < 8) s = old;
            s = zkvm::row_permute(e < 12 ? s : 0, e, rcf);
#pragma unroll
            for (int i = 0; i < 4; ++i) tail[i] = __shfl(s, i);
            if (lane > b) { mine[0] = tail[0]; mine[1] = tail[1]; mine[2] = tail[2]; mine[3] = tail[3]; }
            __syncthreads();
        }
        if (valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) col0[c + (u64)(L::OUT_TAIL + i) * a.in_stride] = mine[i];
        }
    }
}

// sort_decommittment_requests with the integer state and the queue states written by the host packer (zk_pack_sort_decommits_witness_tails):
// left here are the two grand-product accumulators per repetition — a multiplicative scan over the popping cycles of
// ch[r][8] + sum_i enc_i * ch[r][i] over DecommitQuery::encode of the unsorted / the sorted element (mod.rs:232-260, decommit_query/mod.rs:33-113).
// Carried layout (circuits/sort_decommits.cpp): 0 previous_trivial, 1-2 lhs, 3-4 rhs, 17 unsorted length, items at 65 and 76.
struct DecommitLayout { static constexpr u32 ACC = 1, U_LEN = 17, ITEM_U = 65, ITEM_S = 76; };
__device__ __forceinline__ void decommit_encode(u64 enc[8], const u64* col, u64 stride, u32 first) {
    u32 h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (u32)col[(u64)(first + i) * stride];
    const u32 page = (u32)col[(u64)(first + 8) * stride], is_first = (u32)col[(u64)(first + 9) * stride], ts = (u32)col[(u64)(first + 10) * stride];
    enc[0] = (u64)h[0] | ((u64)(page & 0xffffffu) << 32);
    enc[1] = (u64)h[1] | ((u64)(page >> 24) << 32) | ((u64)(ts & 0xffffu) << 40);
    enc[2] = (u64)h[2] | ((u64)(ts >> 16) << 32) | ((u64)(is_first & 1u) << 48);
#pragma unroll
    for (int i = 3; i < 8; ++i) enc[i] = h[i];
}
__device__ __forceinline__ void decommit_terms(u64 out[4], const u64* col, u64 stride, const u64 (*ch)[9]) {
    u64 enc[8];
    decommit_encode(enc, col, stride, DecommitLayout::ITEM_U);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        u64 t = ch[r][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t = gl::fma(enc[i], ch[r][i], t);
        out[r] = t;
    }
    decommit_encode(enc, col, stride, DecommitLayout::ITEM_S);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        u64 t = ch[r][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t = gl::fma(enc[i], ch[r][i], t);
        out[2 + r] = t;
    }
}
__global__ __launch_bounds__(LOGQ_TPB) void k_decommit_seed(LogqSeedDev a) {
    using L = DecommitLayout;
    __shared__ u64 part[2][4][LOGQ_TPB];
    __shared__ u64 ch[2][9];
    const u32 inst = blockIdx.x, t = threadIdx.x;
    const u64 lane0 = (u64)inst * a.limit;
    if (t < 16) ch[t / 8][1 + t % 8] = ovq(a, inst, a.ch_slot[t]);
    if (t < 2) ch[t][0] = 1;
    __syncthreads();
    const u32 per = (a.limit + LOGQ_TPB - 1) / LOGQ_TPB;
    const u32 c_begin = min(t * per, a.limit), c_end = min(c_begin + per, a.limit);
    u64 p[4] = {1, 1, 1, 1};
    for (u32 c = c_begin; c < c_end; ++c) {
        const u64* col = a.loop + lane0 + c;
        if (col[(u64)L::U_LEN * a.in_stride] == 0) continue;
        u64 f[4];
        decommit_terms(f, col, a.in_stride, ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = gl::mul(p[j], f[j]);
    }
    int cur = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) part[0][j][t] = p[j];
    __syncthreads();
    for (u32 d = 1; d < LOGQ_TPB; d <<= 1) {
        const int nxt = cur ^ 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) part[nxt][j][t] = t >= d ? gl::mul(part[cur][j][t], part[cur][j][t - d]) : part[cur][j][t];
        cur = nxt;
        __syncthreads();
    }
    u64 acc[4];   // lhs[0], lhs[1], rhs[0], rhs[1] before cycle c_begin
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u64 init = ovq(a, inst, a.state0_slot[L::ACC + j]);
        acc[j] = t ? gl::mul(init, part[cur][j][t - 1]) : init;
    }
    for (u32 c = c_begin; c < c_end; ++c) {
        u64* col = a.loop + lane0 + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) col[(u64)(L::ACC + j) * a.in_stride] = acc[j];
        if (col[(u64)L::U_LEN * a.in_stride] == 0) continue;
        u64 f[4];
        decommit_terms(f, col, a.in_stride, ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = gl::mul(acc[j], f[j]);
    }
}

}  // namespace zkq
