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
    return a.outer_store[((u64)(inst >> 6) * a.outer_n_store + slot) * 64 + (inst & 63)];
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

}  // namespace zkq
