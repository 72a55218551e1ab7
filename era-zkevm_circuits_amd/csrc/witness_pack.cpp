// witness_pack.cpp — product-side witness packers + the bincode decoder of include/zkgl_witness.h.  Host-only code (no GPU):
// structs -> the input-stream words in the order the recorded circuits allocate them (circuits/ram_permutation.cpp).
#include <cstring>
#include <string>
#include "../../include/zkgl.h"
#include "../../include/zkgl_witness.h"
#include "../../include/zkgl_vm.h"

namespace zkgl { void set_last_error(const std::string& m); }

namespace {
int bad(int code, const char* m) { zkgl::set_last_error(m); return code; }

struct Cursor {
    const uint8_t* p; size_t n, at = 0; bool ok = true;
    bool need(size_t k) { if (!ok || n - at < k) { ok = false; return false; } return true; }
    uint8_t u8() { if (!need(1)) return 0; return p[at++]; }
    uint16_t u16() { if (!need(2)) return 0; uint16_t v; std::memcpy(&v, p + at, 2); at += 2; return v; }
    uint32_t u32() { if (!need(4)) return 0; uint32_t v; std::memcpy(&v, p + at, 4); at += 4; return v; }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v; std::memcpy(&v, p + at, 8); at += 8; return v; }
    bool boolean() { const uint8_t b = u8(); if (b > 1) ok = false; return b == 1; }
    uint64_t field() { const uint64_t v = u64(); if (v >= 0xFFFFFFFF00000001ull) ok = false; return v; }
    // U256 as impl-serde writes it: a string "0x" + hex digits without leading zeros
    void u256(uint32_t limbs[8]) {
        std::memset(limbs, 0, 32);
        const uint64_t len = u64();
        if (!ok || len < 3 || len > 66 || !need((size_t)len)) { ok = false; return; }
        if (p[at] != '0' || p[at + 1] != 'x') { ok = false; return; }
        const size_t digits = (size_t)len - 2;
        for (size_t i = 0; i < digits; ++i) {
            const uint8_t c = p[at + 2 + i];
            uint32_t d;
            if (c >= '0' && c <= '9') d = c - '0';
            else if (c >= 'a' && c <= 'f') d = 10 + c - 'a';
            else if (c >= 'A' && c <= 'F') d = 10 + c - 'A';
            else { ok = false; return; }
            const size_t nib = digits - 1 - i;  // nibble position, least significant = 0
            limbs[nib / 8] |= d << (4 * (nib % 8));
        }
        at += (size_t)len;
    }
    // H160 as impl-serde's fixed-hash writes it: the string "0x" + 40 hex digits (leading zeros kept)
    void h160(uint32_t limbs[5]) {
        std::memset(limbs, 0, 20);
        const uint64_t len = u64();
        if (!ok || len != 42 || !need(42)) { ok = false; return; }
        if (p[at] != '0' || p[at + 1] != 'x') { ok = false; return; }
        for (size_t i = 0; i < 40; ++i) {
            const uint8_t c = p[at + 2 + i];
            uint32_t d;
            if (c >= '0' && c <= '9') d = c - '0';
            else if (c >= 'a' && c <= 'f') d = 10 + c - 'a';
            else if (c >= 'A' && c <= 'F') d = 10 + c - 'A';
            else { ok = false; return; }
            const size_t nib = 39 - i;
            limbs[nib / 8] |= d << (4 * (nib % 8));
        }
        at += 42;
    }
    void queue_state4(zk_queue_state_witness& q) {
        for (auto& x : q.head) x = field();
        for (auto& x : q.tail) x = field();
        q.length = u32();
    }
    void log_query(zk_log_query_witness& q) {   // LogQuery field order (log_query/mod.rs:23-35)
        h160(q.address); u256(q.key); u256(q.read_value); u256(q.written_value);
        q.aux_byte = u8(); q.rw_flag = boolean(); q.rollback = boolean(); q.is_service = boolean(); q.shard_id = u8();
        q.tx_number_in_block = u32(); q.timestamp = u32();
    }
    // CircuitQueueRawWitness<LogQuery, 4, 20>: u64 count, then (item, previous tail[4]) pairs; the tails are not consumed by the circuits
    bool log_queue(zk_log_query_witness* buf, uint32_t cap, uint32_t& n_out, int& err, uint64_t (*tails)[4] = nullptr) {
        const uint64_t n = u64();
        if (!ok) return false;
        if (n > cap || (n && !buf)) { err = ZK_ERR_CAPACITY; return false; }
        for (uint64_t i = 0; i < n && ok; ++i) {
            log_query(buf[i]);
            for (int t = 0; t < 4; ++t) { const uint64_t v = field(); if (tails) tails[i][t] = v; }   // the queue's tail before this element was pushed
        }
        n_out = (uint32_t)n;
        return ok;
    }
    void queue_state(zk_full_queue_state_witness& q) {
        for (auto& x : q.head) x = field();
        for (auto& x : q.tail) x = field();
        q.length = u32();
    }
    void memory_query(zk_memory_query_witness& m) {
        m.timestamp = u32(); m.memory_page = u32(); m.index = u32();
        m.rw_flag = boolean(); m.is_ptr = boolean();
        u256(m.value);
    }
    void ram_fsm(zk_ram_fsm_witness& f) {
        for (auto& x : f.lhs_accumulator) x = field();
        for (auto& x : f.rhs_accumulator) x = field();
        queue_state(f.current_unsorted_queue_state);
        queue_state(f.current_sorted_queue_state);
        for (auto& x : f.previous_sorting_key) x = u32();
        for (auto& x : f.previous_full_key) x = u32();
        u256(f.previous_value);
        f.previous_is_ptr = boolean();
        f.num_nondeterministic_writes = u32();
    }
};

void put_queue_state(uint64_t* dst, size_t stride, size_t& w, const zk_full_queue_state_witness& q) {
    for (auto x : q.head) dst[(w++) * stride] = x;
    for (auto x : q.tail) dst[(w++) * stride] = x;
    dst[(w++) * stride] = q.length;
}

// strided word writer: word k of the instance at dst[k * stride]
struct Out {
    uint64_t* dst; size_t stride; size_t k = 0;
    void w(uint64_t v) { dst[(k++) * stride] = v; }
    template <class T, size_t N> void arr(const T (&a)[N]) { for (auto x : a) w(x); }
    void qstate(const zk_queue_state_witness& q) { arr(q.head); arr(q.tail); w(q.length); }
    void log_query(const zk_log_query_witness* q) {  // LogQuery field order, 36 words; nullptr: the zero item
        if (!q) { for (int i = 0; i < 36; ++i) w(0); return; }
        arr(q->address); arr(q->key); arr(q->read_value); arr(q->written_value);
        w(q->aux_byte); w(q->rw_flag ? 1 : 0); w(q->rollback ? 1 : 0); w(q->is_service ? 1 : 0); w(q->shard_id); w(q->tx_number_in_block); w(q->timestamp);
    }
};
}  // namespace

extern "C" {

int zk_pack_ram_witness(const zk_ram_permutation_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_ram_witness: bad argument");
    if (w->n_unsorted != w->n_sorted) return bad(ZK_ERR_INVALID, "zk_pack_ram_witness: the two queue witnesses differ in length");
    if (w->n_unsorted > limit) return bad(ZK_ERR_INVALID, "zk_pack_ram_witness: more queue elements than cycles");
    if (w->n_unsorted && (!w->unsorted_queue_witness || !w->sorted_queue_witness)) return bad(ZK_ERR_INVALID, "zk_pack_ram_witness: null queue witness");
    // ---- outer scope: start_flag, observable input, hidden FSM input (circuits/ram_permutation.cpp allocation order)
    {
        uint64_t* o = outer_words + instance;
        const size_t st = batch;
        size_t k = 0;
        o[(k++) * st] = w->start_flag ? 1 : 0;
        put_queue_state(o, st, k, w->unsorted_queue_initial_state);
        put_queue_state(o, st, k, w->sorted_queue_initial_state);
        o[(k++) * st] = w->non_deterministic_bootloader_memory_snapshot_length;
        const zk_ram_fsm_witness& f = w->hidden_fsm_input;
        for (auto x : f.lhs_accumulator) o[(k++) * st] = x;
        for (auto x : f.rhs_accumulator) o[(k++) * st] = x;
        put_queue_state(o, st, k, f.current_unsorted_queue_state);
        put_queue_state(o, st, k, f.current_sorted_queue_state);
        for (auto x : f.previous_sorting_key) o[(k++) * st] = x;
        for (auto x : f.previous_full_key) o[(k++) * st] = x;
        for (auto x : f.previous_value) o[(k++) * st] = x;
        o[(k++) * st] = f.previous_is_ptr ? 1 : 0;
        o[(k++) * st] = f.num_nondeterministic_writes;
        if (k != ZK_RAM_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: ram outer layout");
    }
    // ---- loop scope: 46 carried words (device seeding), then the popped unsorted and sorted elements of the cycle
    const size_t lanes = (size_t)batch * limit;
    for (uint32_t c = 0; c < limit; ++c) {
        uint64_t* l = loop_words + (size_t)instance * limit + c;
        size_t k = 0;
        for (; k < 46; ++k) l[k * lanes] = 0;
        for (int side = 0; side < 2; ++side) {
            const zk_memory_query_witness* q = side == 0 ? w->unsorted_queue_witness : w->sorted_queue_witness;
            if (c < w->n_unsorted) {
                const zk_memory_query_witness& m = q[c];
                l[(k++) * lanes] = m.timestamp; l[(k++) * lanes] = m.memory_page; l[(k++) * lanes] = m.index;
                l[(k++) * lanes] = m.rw_flag ? 1 : 0; l[(k++) * lanes] = m.is_ptr ? 1 : 0;
                for (auto x : m.value) l[(k++) * lanes] = x;
            } else {
                for (int i = 0; i < 13; ++i) l[(k++) * lanes] = 0;
            }
        }
        if (k != ZK_RAM_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: ram loop layout");
        // queue heads from the witness's previous tails: head before popping element c = the state before element c was pushed; past
        // the last element the head has reached the tail of the state the circuit starts from (observable input, or the FSM input)
        if (w->unsorted_previous_tails && w->sorted_previous_tails) {
            const zk_full_queue_state_witness& qu = w->start_flag ? w->unsorted_queue_initial_state : w->hidden_fsm_input.current_unsorted_queue_state;
            const zk_full_queue_state_witness& qs = w->start_flag ? w->sorted_queue_initial_state : w->hidden_fsm_input.current_sorted_queue_state;
            for (int i = 0; i < 12; ++i) {
                l[(1 + i) * lanes] = c < w->n_unsorted ? w->unsorted_previous_tails[c][i] : qu.tail[i];
                l[(14 + i) * lanes] = c < w->n_sorted ? w->sorted_previous_tails[c][i] : qs.tail[i];
            }
        }
    }
    return ZK_OK;
}

int zk_decode_ram_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_ram_permutation_witness* out, zk_memory_query_witness* unsorted_buf,
                                  uint32_t unsorted_cap, zk_memory_query_witness* sorted_buf, uint32_t sorted_cap, size_t* consumed) {
    return zk_decode_ram_witness_bincode_tails(bytes, n_bytes, out, unsorted_buf, unsorted_cap, sorted_buf, sorted_cap, nullptr, nullptr, consumed);
}

void zk_ram_head_words(uint32_t words[ZK_RAM_HEAD_WORDS]) {   // circuits/ram_permutation.cpp: [1..13) unsorted head, [14..26) sorted head
    for (uint32_t i = 0; i < 12; ++i) { words[i] = 1 + i; words[12 + i] = 14 + i; }
}

int zk_decode_ram_witness_bincode_tails(const uint8_t* bytes, size_t n_bytes, zk_ram_permutation_witness* out, zk_memory_query_witness* unsorted_buf,
                                        uint32_t unsorted_cap, zk_memory_query_witness* sorted_buf, uint32_t sorted_cap, uint64_t (*unsorted_tails)[12],
                                        uint64_t (*sorted_tails)[12], size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_ram_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    // closed_form_input: start_flag, completion_flag, observable_input, observable_output = () (nothing), hidden_fsm_input, hidden_fsm_output
    out->start_flag = c.boolean();
    out->completion_flag = c.boolean();
    c.queue_state(out->unsorted_queue_initial_state);
    c.queue_state(out->sorted_queue_initial_state);
    out->non_deterministic_bootloader_memory_snapshot_length = c.u32();
    c.ram_fsm(out->hidden_fsm_input);
    c.ram_fsm(out->hidden_fsm_output);
    for (int side = 0; side < 2 && c.ok; ++side) {
        const uint64_t n = c.u64();  // VecDeque length
        zk_memory_query_witness* buf = side == 0 ? unsorted_buf : sorted_buf;
        const uint32_t cap = side == 0 ? unsorted_cap : sorted_cap;
        if (!c.ok) break;
        if (n > cap || (n && !buf)) return bad(ZK_ERR_CAPACITY, "zk_decode_ram_witness_bincode: queue witness longer than the caller's buffer");
        for (uint64_t i = 0; i < n && c.ok; ++i) {
            c.memory_query(buf[i]);
            uint64_t (*tails)[12] = side == 0 ? unsorted_tails : sorted_tails;
            for (int t = 0; t < 12; ++t) {   // the queue state before this push = the head before its pop
                const uint64_t v = c.field();
                if (tails) tails[i][t] = v;
            }
        }
        if (side == 0) { out->unsorted_queue_witness = buf; out->n_unsorted = (uint32_t)n; out->unsorted_previous_tails = unsorted_tails; }
        else { out->sorted_queue_witness = buf; out->n_sorted = (uint32_t)n; out->sorted_previous_tails = sorted_tails; }
    }
    if (!c.ok) return bad(ZK_ERR_INVALID, "zk_decode_ram_witness_bincode: truncated or malformed input");
    if (consumed) *consumed = c.at;
    return ZK_OK;
}


int zk_pack_storage_witness(const zk_storage_validity_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_storage_witness: bad argument");
    if (w->n_unsorted != w->n_sorted) return bad(ZK_ERR_INVALID, "zk_pack_storage_witness: the two queue witnesses differ in length");
    if (w->n_unsorted > limit) return bad(ZK_ERR_INVALID, "zk_pack_storage_witness: more queue elements than cycles");
    if (w->n_unsorted && (!w->unsorted_queue_witness || !w->intermediate_sorted_queue_witness)) return bad(ZK_ERR_INVALID, "zk_pack_storage_witness: null queue witness");
    Out o{outer_words + instance, batch};
    o.w(w->start_flag ? 1 : 0);
    o.w(w->shard_id_to_process); o.qstate(w->unsorted_log_queue_state); o.qstate(w->intermediate_sorted_queue_state);
    const zk_storage_fsm_witness& f = w->hidden_fsm_input;  // StorageDeduplicatorFSMInputOutput order (input.rs:37-52)
    o.arr(f.lhs_accumulator); o.arr(f.rhs_accumulator);
    o.qstate(f.current_unsorted_queue_state); o.qstate(f.current_intermediate_sorted_queue_state); o.qstate(f.current_final_sorted_queue_state);
    o.w(f.cycle_idx); o.arr(f.previous_packed_key); o.arr(f.previous_key); o.arr(f.previous_address); o.w(f.previous_timestamp);
    o.w(f.this_cell_has_explicit_read_and_rollback_depth_zero ? 1 : 0); o.arr(f.this_cell_base_value); o.arr(f.this_cell_current_value); o.w(f.this_cell_current_depth);
    if (o.k != ZK_STORAGE_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: storage outer layout");
    const size_t lanes = (size_t)batch * limit;
    // With the previous tails: the integer carried state of every cycle, walked here (sort_and_deduplicate_storage_access_inner,
    // mod.rs:508-880; initial selection :230-330 = circuits/storage_validity.cpp).  Left to the device: the grand-product accumulators
    // (words 2..5: they need the Fiat-Shamir challenges) and, without output_tails, the final queue's tail (words 17..20).
    const bool walk = w->unsorted_previous_tails && w->sorted_previous_tails;
    const zk_log_query_witness zero_item{};
    const bool start = w->start_flag != 0;
    const zk_queue_state_witness& u0 = start ? w->unsorted_log_queue_state : f.current_unsorted_queue_state;
    const zk_queue_state_witness& s0 = start ? w->intermediate_sorted_queue_state : f.current_intermediate_sorted_queue_state;
    uint32_t u_len = u0.length, s_len = s0.length, f_len = start ? 0 : f.current_final_sorted_queue_state.length, popped = 0, pushed = 0;
    uint64_t f_tail[4];
    for (int i = 0; i < 4; ++i) f_tail[i] = start ? 0 : f.current_final_sorted_queue_state.tail[i];
    uint32_t cycle_idx = start ? 0 : f.cycle_idx, prev_ts = f.previous_timestamp, depth = f.this_cell_current_depth;
    uint32_t prev_packed[13], prev_key[8], prev_addr[5], base[8], cur[8];
    for (int i = 0; i < 13; ++i) prev_packed[i] = start ? 0 : f.previous_packed_key[i];
    std::memcpy(prev_key, f.previous_key, sizeof prev_key); std::memcpy(prev_addr, f.previous_address, sizeof prev_addr);
    std::memcpy(base, f.this_cell_base_value, sizeof base); std::memcpy(cur, f.this_cell_current_value, sizeof cur);
    bool has_read = f.this_cell_has_explicit_read_and_rollback_depth_zero != 0, prev_trivial = u0.length == 0 || start;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        const zk_log_query_witness& rec = c < w->n_sorted ? w->intermediate_sorted_queue_witness[c].record : zero_item;
        const uint32_t sts = c < w->n_sorted ? w->intermediate_sorted_queue_witness[c].timestamp : 0;
        if (!walk) for (int i = 0; i < 67; ++i) l.w(0);
        else {
            l.w(c == 0 ? 1 : 0); l.w(prev_trivial ? 1 : 0);
            for (int i = 0; i < 4; ++i) l.w(0);
            l.w(cycle_idx);
            for (int i = 0; i < 4; ++i) l.w(popped < w->n_unsorted ? w->unsorted_previous_tails[popped][i] : u0.tail[i]);
            l.w(u_len);
            for (int i = 0; i < 4; ++i) l.w(popped < w->n_sorted ? w->sorted_previous_tails[popped][i] : s0.tail[i]);
            l.w(s_len);
            for (int i = 0; i < 4; ++i) l.w(w->output_tails ? f_tail[i] : 0);
            l.w(f_len);
            l.arr(prev_packed); l.arr(prev_key); l.arr(prev_addr); l.w(prev_ts); l.w(has_read ? 1 : 0); l.arr(base); l.arr(cur); l.w(depth);
            // ---- the cycle
            cycle_idx += 1;
            const bool should_pop = u_len != 0 && s_len != 0, trivial = !should_pop;
            if (should_pop) { u_len -= 1; s_len -= 1; popped += 1; }
            uint32_t packed[13];
            std::memcpy(packed, rec.key, 32); std::memcpy(packed + 8, rec.address, 20);
            const bool keys_equal = std::memcmp(packed, prev_packed, sizeof packed) == 0;
            const bool unchanged = std::memcmp(cur, base, sizeof cur) == 0;
            const bool issue_read = has_read || (unchanged && depth != 0), should_write = !unchanged;
            if (!prev_trivial && !keys_equal && (issue_read || should_write)) {
                if (w->output_tails) {
                    if (pushed >= w->n_output_tails) return bad(ZK_ERR_INVALID, "zk_pack_storage_witness: fewer output tails than the instance pushes");
                    for (int i = 0; i < 4; ++i) f_tail[i] = w->output_tails[pushed][i];
                }
                pushed += 1; f_len += 1;
            }
            const bool new_cell = !trivial && !keys_equal, same_cell = !trivial && keys_equal, rw = rec.rw_flag != 0, rb = rec.rollback != 0;
            if (new_cell) {
                std::memcpy(base, rec.read_value, sizeof base);
                std::memcpy(cur, rw ? rec.written_value : rec.read_value, sizeof cur);
                depth = rw ? 1 : 0; has_read = !rw;
            }
            const bool read_same = same_cell && !rw, w_norb = same_cell && rw && !rb, w_rb = same_cell && rw && rb;
            if (w_norb) { depth += 1; std::memcpy(cur, rec.written_value, sizeof cur); }
            if (w_rb) { depth -= 1; std::memcpy(cur, rec.read_value, sizeof cur); }
            if (depth == 0 && read_same) { std::memcpy(base, rec.read_value, sizeof base); has_read = true; }
            std::memcpy(prev_addr, rec.address, sizeof prev_addr); std::memcpy(prev_key, rec.key, sizeof prev_key);
            std::memcpy(prev_packed, packed, sizeof packed);
            prev_trivial = trivial; prev_ts = sts;
        }
        l.log_query(c < w->n_unsorted ? &w->unsorted_queue_witness[c] : nullptr);
        if (c < w->n_sorted) { l.log_query(&w->intermediate_sorted_queue_witness[c].record); l.w(w->intermediate_sorted_queue_witness[c].timestamp); }
        else { l.log_query(nullptr); l.w(0); }
        if (l.k != ZK_STORAGE_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: storage loop layout");
    }
    return ZK_OK;
}

uint32_t zk_storage_given_words(const zk_storage_validity_witness* w, uint32_t words[67]) {
    if (!w || !words || !w->unsorted_previous_tails || !w->sorted_previous_tails) return 0;
    uint32_t n = 0;
    for (uint32_t i = 0; i < 67; ++i) {
        if (i >= 2 && i < 6) continue;                          // lhs / rhs accumulators
        if (i >= 17 && i < 21 && !w->output_tails) continue;    // final queue tail
        words[n++] = i;
    }
    return n;
}
uint32_t zk_log_sorter_given_words(const zk_log_sorter_witness* w, uint32_t words[57]) {
    if (!w || !words || !w->initial_previous_tails || !w->sorted_previous_tails) return 0;
    uint32_t n = 0;
    for (uint32_t i = 0; i < 57; ++i) {
        if (i >= 1 && i < 5) continue;
        if (i >= 15 && i < 19 && !w->output_tails) continue;
        words[n++] = i;
    }
    return n;
}

int zk_pack_log_sorter_witness(const zk_log_sorter_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_log_sorter_witness: bad argument");
    if (w->n_initial != w->n_sorted) return bad(ZK_ERR_INVALID, "zk_pack_log_sorter_witness: the two queue witnesses differ in length");
    if (w->n_initial > limit) return bad(ZK_ERR_INVALID, "zk_pack_log_sorter_witness: more queue elements than cycles");
    if (w->n_initial && (!w->initial_queue_witness || !w->intermediate_sorted_queue_witness)) return bad(ZK_ERR_INVALID, "zk_pack_log_sorter_witness: null queue witness");
    Out o{outer_words + instance, batch};
    o.w(w->start_flag ? 1 : 0);
    o.qstate(w->initial_log_queue_state); o.qstate(w->intermediate_sorted_queue_state);
    const zk_log_sorter_fsm_witness& f = w->hidden_fsm_input;  // EventsDeduplicatorFSMInputOutput order (input.rs:28-36)
    o.arr(f.lhs_accumulator); o.arr(f.rhs_accumulator);
    o.qstate(f.initial_unsorted_queue_state); o.qstate(f.intermediate_sorted_queue_state); o.qstate(f.final_result_queue_state);
    o.w(f.previous_key); o.log_query(&f.previous_item);
    if (o.k != ZK_LOG_SORTER_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: log_sorter outer layout");
    const size_t lanes = (size_t)batch * limit;
    // with the previous tails: the integer carried state (repack_and_prove_events_rollbacks_inner, mod.rs:246-441), as in zk_pack_storage_witness
    const bool walk = w->initial_previous_tails && w->sorted_previous_tails;
    const zk_log_query_witness zero_item{};
    const bool start = w->start_flag != 0;
    const zk_queue_state_witness& u0 = start ? w->initial_log_queue_state : f.initial_unsorted_queue_state;
    const zk_queue_state_witness& s0 = start ? w->intermediate_sorted_queue_state : f.intermediate_sorted_queue_state;
    uint32_t u_len = u0.length, s_len = s0.length, r_len = start ? 0 : f.final_result_queue_state.length, popped = 0, pushed = 0;
    uint64_t r_tail[4];
    for (int i = 0; i < 4; ++i) r_tail[i] = start ? 0 : f.final_result_queue_state.tail[i];
    uint32_t prev_key = start ? 0 : f.previous_key;
    zk_log_query_witness prev_item = start ? zero_item : f.previous_item;
    bool prev_trivial = u0.length == 0 || start;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        if (!walk) for (int i = 0; i < 57; ++i) l.w(0);
        else {
            const zk_log_query_witness& sq = c < w->n_sorted ? w->intermediate_sorted_queue_witness[c] : zero_item;
            l.w(prev_trivial ? 1 : 0);
            for (int i = 0; i < 4; ++i) l.w(0);
            for (int i = 0; i < 4; ++i) l.w(popped < w->n_initial ? w->initial_previous_tails[popped][i] : u0.tail[i]);
            l.w(u_len);
            for (int i = 0; i < 4; ++i) l.w(popped < w->n_sorted ? w->sorted_previous_tails[popped][i] : s0.tail[i]);
            l.w(s_len);
            for (int i = 0; i < 4; ++i) l.w(w->output_tails ? r_tail[i] : 0);
            l.w(r_len);
            l.w(prev_key); l.log_query(&prev_item);
            const bool should_pop = u_len != 0, trivial = !should_pop;
            if (should_pop) { u_len -= 1; s_len -= 1; popped += 1; }
            const bool same_log = sq.timestamp == prev_key;
            if (!prev_trivial && (!same_log || trivial) && !prev_item.rollback) {
                if (w->output_tails) {
                    if (pushed >= w->n_output_tails) return bad(ZK_ERR_INVALID, "zk_pack_log_sorter_witness: fewer output tails than the instance pushes");
                    for (int i = 0; i < 4; ++i) r_tail[i] = w->output_tails[pushed][i];
                }
                pushed += 1; r_len += 1;
            }
            prev_trivial = trivial; prev_item = sq; prev_key = sq.timestamp;
        }
        l.log_query(c < w->n_initial ? &w->initial_queue_witness[c] : nullptr);
        l.log_query(c < w->n_sorted ? &w->intermediate_sorted_queue_witness[c] : nullptr);
        if (l.k != ZK_LOG_SORTER_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: log_sorter loop layout");
    }
    return ZK_OK;
}


int zk_eip4844_stream_shape(uint32_t n_chunks, uint32_t* n_iterations, uint32_t* loop_words) {
    if (!n_chunks || !n_iterations || !loop_words) return bad(ZK_ERR_INVALID, "zk_eip4844_stream_shape: bad argument");
    const uint64_t n_bytes = 31ull * n_chunks;
    const uint32_t n_blocks = (uint32_t)(n_bytes / 136 + 1);               // the padding always opens a block (circuits/eip4844.cpp)
    const uint32_t cpi = (n_chunks + n_blocks - 1) / n_blocks;               // Horner steps per iteration
    *n_iterations = n_blocks;
    *loop_words = 217 + 136 + 31 * cpi;
    return ZK_OK;
}

int zk_pack_eip4844_witness(const zk_eip4844_witness* w, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || instance >= batch || !w->n_chunks || !w->data_chunks) return bad(ZK_ERR_INVALID, "zk_pack_eip4844_witness: bad argument");
    uint32_t n_it = 0, lw = 0;
    zk_eip4844_stream_shape(w->n_chunks, &n_it, &lw);
    const uint32_t cpi = (lw - 217 - 136) / 31;
    const uint64_t n_bytes = 31ull * w->n_chunks;
    Out o{outer_words + instance, batch};
    o.arr(w->versioned_hash); o.arr(w->linear_hash_output);
    const size_t lanes = (size_t)batch * n_it;
    for (uint32_t t = 0; t < n_it; ++t) {
        Out l{loop_words + (size_t)instance * n_it + t, lanes};
        for (int i = 0; i < 217; ++i) l.w(0);
        for (uint32_t j = 0; j < 136; ++j) { const uint64_t at = 136ull * t + j; l.w(at < n_bytes ? w->data_chunks[at] : 0); }
        for (uint32_t j = 0; j < 31 * cpi; ++j) { const uint64_t at = 31ull * cpi * t + j; l.w(at < n_bytes ? w->data_chunks[at] : 0); }
        if (l.k != lw) return bad(ZK_ERR_INVALID, "internal: eip_4844 loop layout");
    }
    return ZK_OK;
}


int zk_pack_sha256_witness(const zk_sha256_round_function_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_sha256_witness: bad argument");
    if ((w->n_requests && !w->requests_queue_witness) || (w->n_reads && !w->memory_reads_witness)) return bad(ZK_ERR_INVALID, "zk_pack_sha256_witness: null witness array");
    auto full = [](Out& o, const zk_full_queue_state_witness& q) { o.arr(q.head); o.arr(q.tail); o.w(q.length); };
    Out o{outer_words + instance, batch};
    o.w(w->start_flag ? 1 : 0);
    o.qstate(w->initial_log_queue_state); full(o, w->initial_memory_queue_state);
    const zk_sha256_fsm_witness& f = w->hidden_fsm_input;
    o.w(f.read_precompile_call ? 1 : 0); o.w(f.read_words_for_round ? 1 : 0); o.w(f.completed ? 1 : 0);
    o.arr(f.sha256_inner_state); o.w(f.timestamp_to_use_for_read); o.w(f.timestamp_to_use_for_write);
    o.w(f.input_page); o.w(f.input_offset); o.w(f.output_page); o.w(f.output_offset); o.w(f.num_rounds);
    o.qstate(f.log_queue_state); full(o, f.memory_queue_state);
    if (o.k != ZK_SHA256_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: sha256 outer layout");
    // the FSM's schedule (mod.rs:120-137 can_finish_immediatelly, :150-181 request pop, :199-255 reads, :380-418 next flags)
    bool rpc, rwfr, completed;
    uint64_t num_rounds, req_len;
    if (w->start_flag) { rpc = true; rwfr = false; completed = false; num_rounds = 0; req_len = w->initial_log_queue_state.length; }
    else { rpc = f.read_precompile_call; rwfr = f.read_words_for_round; completed = f.completed; num_rounds = f.num_rounds; req_len = f.log_queue_state.length; }
    if (rpc && req_len == 0) { rpc = false; rwfr = false; completed = true; }
    uint32_t next_req = 0, next_read = 0;
    const size_t lanes = (size_t)batch * limit;
    const uint64_t P = 0xFFFFFFFF00000001ull;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        for (int i = 0; i < 60; ++i) l.w(0);
        const zk_log_query_witness* call = nullptr;
        if (rpc && req_len != 0) {
            if (next_req >= w->n_requests) return bad(ZK_ERR_INVALID, "zk_pack_sha256_witness: the request queue witness is shorter than its length");
            call = &w->requests_queue_witness[next_req++];
            --req_len;
            num_rounds = call->key[6];   // precompile call ABI: the number of rounds is limb 6 of the key
        }
        l.log_query(call);
        rwfr = rpc || rwfr;
        rpc = false;
        const bool should_read = num_rounds != 0;
        for (int r = 0; r < 2; ++r) {
            if (should_read && next_read < w->n_reads) { for (int i = 0; i < 8; ++i) l.w(w->memory_reads_witness[next_read][i]); ++next_read; }
            else for (int i = 0; i < 8; ++i) l.w(0);
        }
        if (rwfr) num_rounds = num_rounds ? num_rounds - 1 : P - 1;   // the circuit's field subtraction
        const bool write_result = rwfr && num_rounds == 0, input_is_empty = req_len == 0;
        rpc = write_result && !input_is_empty;
        completed = (write_result && input_is_empty) || completed;
        rwfr = !(rpc || completed);
        if (l.k != ZK_SHA256_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: sha256 loop layout");
    }
    return ZK_OK;
}


int zk_pack_keccak_witness(const zk_keccak_round_function_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_keccak_witness: bad argument");
    if ((w->n_requests && !w->requests_queue_witness) || (w->n_reads && !w->memory_reads_witness)) return bad(ZK_ERR_INVALID, "zk_pack_keccak_witness: null witness array");
    constexpr uint32_t RATE = 136, BUF = 192, READS = 6;
    auto full = [](Out& o, const zk_full_queue_state_witness& q) { o.arr(q.head); o.arr(q.tail); o.w(q.length); };
    Out o{outer_words + instance, batch};
    o.w(w->start_flag ? 1 : 0);
    o.qstate(w->initial_log_queue_state); full(o, w->initial_memory_queue_state);
    const zk_keccak_fsm_witness& f = w->hidden_fsm_input;
    o.w(f.read_precompile_call ? 1 : 0); o.w(f.read_unaligned_words_for_round ? 1 : 0); o.w(f.padding_round ? 1 : 0); o.w(f.completed ? 1 : 0);
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) for (int k = 0; k < 8; ++k) o.w(f.keccak_internal_state[i][j][k]);
    o.w(f.timestamp_to_use_for_read); o.w(f.timestamp_to_use_for_write);
    o.w(f.input_page); o.w(f.input_memory_byte_offset); o.w(f.input_memory_byte_length); o.w(f.output_page); o.w(f.output_word_offset);
    o.w(f.needs_full_padding_round ? 1 : 0);
    o.arr(f.buffer_bytes); o.w(f.buffer_filled);
    o.qstate(f.log_queue_state); full(o, f.memory_queue_state);
    if (o.k != ZK_KECCAK_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: keccak outer layout");
    // the schedule: which cycle pops a request, which of its six read slots fetch a word (mod.rs:200-213, :228-297, :300-420, :690-740)
    bool rpc, ruw, padding_round, completed, needs_full;
    uint64_t byte_offset, byte_length, filled, req_len;
    if (w->start_flag) { rpc = true; ruw = padding_round = completed = needs_full = false; byte_offset = byte_length = filled = 0; req_len = w->initial_log_queue_state.length; }
    else {
        rpc = f.read_precompile_call; ruw = f.read_unaligned_words_for_round; padding_round = f.padding_round; completed = f.completed;
        needs_full = f.needs_full_padding_round; byte_offset = f.input_memory_byte_offset; byte_length = f.input_memory_byte_length;
        filled = f.buffer_filled; req_len = f.log_queue_state.length;
    }
    if (rpc && req_len == 0) { rpc = false; ruw = false; completed = true; }
    uint32_t next_req = 0, next_read = 0;
    const size_t lanes = (size_t)batch * limit;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        for (int i = 0; i < 423; ++i) l.w(0);
        const zk_log_query_witness* call = nullptr;
        if (rpc && req_len != 0) {
            if (next_req >= w->n_requests) return bad(ZK_ERR_INVALID, "zk_pack_keccak_witness: the request queue witness is shorter than its length");
            call = &w->requests_queue_witness[next_req++];
            --req_len;
        }
        l.log_query(call);
        const uint64_t call_length = call ? call->key[1] : 0;
        if (rpc) {  // precompile call ABI in the key limbs: byte offset, byte length (an absent call reads as zeros, as in the circuit)
            byte_offset = call ? call->key[0] : 0; byte_length = call_length;
            needs_full = call_length % RATE == 0;
        }
        const bool reset_buffer = rpc || completed;
        if (rpc && call_length == 0) padding_round = true;
        if (rpc && call_length != 0) ruw = true;
        rpc = false;
        if (reset_buffer) filled = 0;
        for (uint32_t r = 0; r < READS; ++r) {
            const uint64_t unalignment = byte_offset % 32, at_most = 32 - unalignment;
            const uint64_t meaningful = byte_length < at_most ? byte_length : at_most;
            const bool should_read = meaningful != 0 && filled + meaningful <= BUF && ruw;
            if (should_read && next_read < w->n_reads) { for (int i = 0; i < 8; ++i) l.w(w->memory_reads_witness[next_read][i]); ++next_read; }
            else for (int i = 0; i < 8; ++i) l.w(0);
            if (should_read) { byte_offset += meaningful; byte_length -= meaningful; filled += meaningful; }
        }
        const bool zero_bytes_left = byte_length == 0;
        filled = filled >= RATE ? filled - RATE : 0;
        const bool buffer_now_empty = filled == 0;
        const bool apply_padding = zero_bytes_left && buffer_now_empty && ruw && !needs_full;
        const bool write_result = apply_padding || padding_round, input_is_empty = req_len == 0;
        rpc = write_result && !input_is_empty;
        completed = (write_result && input_is_empty) || completed;
        padding_round = ruw && zero_bytes_left && buffer_now_empty && needs_full;
        ruw = !(rpc || padding_round || completed);
        if (l.k != ZK_KECCAK_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: keccak loop layout");
    }
    return ZK_OK;
}


int zk_pack_demux_witness(const zk_demux_log_queue_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_demux_witness: bad argument");
    if (w->n_initial > limit) return bad(ZK_ERR_INVALID, "zk_pack_demux_witness: more queue elements than cycles");
    if (w->n_initial && !w->initial_queue_witness) return bad(ZK_ERR_INVALID, "zk_pack_demux_witness: null queue witness");
    Out o{outer_words + instance, batch};
    o.w(w->start_flag ? 1 : 0);
    o.qstate(w->initial_log_queue_state);
    const zk_demux_fsm_witness& f = w->hidden_fsm_input;   // LogDemuxerFSMInputOutput order (input.rs:26-34)
    o.qstate(f.initial_log_queue_state);
    for (const auto& q : f.output_queue_states) o.qstate(q);
    if (o.k != ZK_DEMUX_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: demux outer layout");
    const size_t lanes = (size_t)batch * limit;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        for (int i = 0; i < 35; ++i) l.w(0);
        l.log_query(c < w->n_initial ? &w->initial_queue_witness[c] : nullptr);
        if (l.k != ZK_DEMUX_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: demux loop layout");
    }
    return ZK_OK;
}

// With the states the reference's witnesses already hold nothing is left to derive on the device: the input queue's witness is a
// VecDeque of (LogQuery, previous tail) (input.rs:118-121: CircuitQueueRawWitness) — the head the circuit holds before it pops that
// element — and each of the six output queues is the input queue of a later circuit whose witness holds the same pairs, i.e. the
// tail after every push here.  The lengths are counts.  Every one of the 35 carried words of every cycle is then a function of the
// witness alone (zk_demux_given_words = all of them): zk_cs_seed_* has no kernel to run, the circuit's own queue constraints and CARRY
// links judge the words.
int zk_pack_demux_witness_tails(const zk_demux_log_queue_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words,
                                const uint64_t* input_previous_tails, const uint64_t* output_tails) {
    if (!input_previous_tails || !output_tails) return bad(ZK_ERR_INVALID, "zk_pack_demux_witness_tails: null tails (use zk_pack_demux_witness and device seeding)");
    if (int rc = zk_pack_demux_witness(w, limit, instance, batch, outer_words, loop_words)) return rc;
    // the state the first cycle starts from: observable input (empty output queues) on the first instance, the FSM input otherwise
    const zk_queue_state_witness& in0 = w->start_flag ? w->initial_log_queue_state : w->hidden_fsm_input.initial_log_queue_state;
    uint64_t head[4], out_tail[6][4];
    uint32_t len = in0.length, out_len[6];
    for (int k = 0; k < 4; ++k) head[k] = in0.head[k];
    for (int q = 0; q < 6; ++q) {
        for (int k = 0; k < 4; ++k) out_tail[q][k] = w->start_flag ? 0 : w->hidden_fsm_input.output_queue_states[q].tail[k];
        out_len[q] = w->start_flag ? 0 : w->hidden_fsm_input.output_queue_states[q].length;
    }
    if (len > w->n_initial && len > limit) { /* more elements than this instance's witness holds: the remaining cycles pop what is there */ }
    const size_t lanes = (size_t)batch * limit;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        const bool pops = len != 0 && c < w->n_initial;
        if (pops) for (int k = 0; k < 4; ++k) head[k] = input_previous_tails[4 * (size_t)c + k];   // the head before this pop, from the witness
        l.arr(head); l.w(len);
        for (int q = 0; q < 6; ++q) { l.arr(out_tail[q]); l.w(out_len[q]); }
        if (pops) {
            const zk_log_query_witness& it = w->initial_queue_witness[c];
            // the head after the pop = the previous tail of the next element, or the queue's tail when this was the last one
            if (c + 1 < w->n_initial && len > 1) for (int k = 0; k < 4; ++k) head[k] = input_previous_tails[4 * (size_t)(c + 1) + k];
            else for (int k = 0; k < 4; ++k) head[k] = in0.tail[k];
            --len;
            bool hi_zero = true;
            for (int i = 1; i < 5; ++i) hi_zero &= it.address[i] == 0;
            int q = -1;   // mod.rs:300-356 (aux byte, shard, formal precompile addresses [EXT] zkevm_opcode_defs)
            if (it.aux_byte == 0 && it.shard_id == 0) q = 0;
            else if (it.aux_byte == 1) q = 1;
            else if (it.aux_byte == 2) q = 2;
            else if (it.aux_byte == 3 && hi_zero && it.address[0] == 0x8010) q = 3;
            else if (it.aux_byte == 3 && hi_zero && it.address[0] == 0x02) q = 4;
            else if (it.aux_byte == 3 && hi_zero && it.address[0] == 0x01) q = 5;
            if (q >= 0) {
                for (int k = 0; k < 4; ++k) out_tail[q][k] = output_tails[4 * (size_t)c + k];
                ++out_len[q];
            }
        }
    }
    return ZK_OK;
}
uint32_t zk_demux_given_words(uint32_t words[35]) {
    for (uint32_t i = 0; i < 35; ++i) words[i] = i;
    return 35;
}

namespace {
void put_decommit(Out& o, const zk_decommit_query_witness* q) {   // DecommitQuery field order, 11 words; nullptr: the zero item
    if (!q) { for (int i = 0; i < 11; ++i) o.w(0); return; }
    o.arr(q->code_hash); o.w(q->page); o.w(q->is_first ? 1 : 0); o.w(q->timestamp);
}
void put_full(Out& o, const zk_full_queue_state_witness& q) { o.arr(q.head); o.arr(q.tail); o.w(q.length); }
}  // namespace

int zk_pack_sort_decommits_witness(const zk_sort_decommits_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_sort_decommits_witness: bad argument");
    if (w->n_initial != w->n_sorted) return bad(ZK_ERR_INVALID, "zk_pack_sort_decommits_witness: the two queue witnesses differ in length");
    if (w->n_initial > limit) return bad(ZK_ERR_INVALID, "zk_pack_sort_decommits_witness: more queue elements than cycles");
    if (w->n_initial && (!w->initial_queue_witness || !w->sorted_queue_witness)) return bad(ZK_ERR_INVALID, "zk_pack_sort_decommits_witness: null queue witness");
    Out o{outer_words + instance, batch};
    o.w(w->start_flag ? 1 : 0);
    put_full(o, w->initial_queue_state); put_full(o, w->sorted_queue_initial_state);
    const zk_sort_decommits_fsm_witness& f = w->hidden_fsm_input;   // CodeDecommittmentsDeduplicatorFSMInputOutput order (input.rs:26-37)
    put_full(o, f.initial_queue_state); put_full(o, f.sorted_queue_state); put_full(o, f.final_queue_state);
    o.arr(f.lhs_accumulator); o.arr(f.rhs_accumulator); o.arr(f.previous_packed_key); o.w(f.first_encountered_timestamp);
    put_decommit(o, &f.previous_record);
    if (o.k != ZK_SORT_DECOMMITS_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: sort_decommits outer layout");
    const size_t lanes = (size_t)batch * limit;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        for (int i = 0; i < 65; ++i) l.w(0);
        put_decommit(l, c < w->n_initial ? &w->initial_queue_witness[c] : nullptr);
        put_decommit(l, c < w->n_sorted ? &w->sorted_queue_witness[c] : nullptr);
        if (l.k != ZK_SORT_DECOMMITS_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: sort_decommits loop layout");
    }
    return ZK_OK;
}

// The deduplicator with the queue states its own and its neighbour's witnesses hold: both input queues' witnesses are VecDeques of
// (DecommitQuery, previous tail) (input.rs:110-124: FullStateCircuitQueueRawWitness) — the 12-word heads before each pop — and the
// result queue is the decommitter's requests queue, whose witness holds the same pairs, i.e. the tail after every push here
// (`result_tails[k]` = the tail after the k-th push of this instance's loop).  The integer state (previous key / record, first
// timestamp, lengths, previous_trivial: mod.rs:264-345) is walked here; the four grand-product words (1..4) are the device's scans
// (k_decommit_seed) because they need the Fiat-Shamir challenges the circuit derives.
int zk_pack_sort_decommits_witness_tails(const zk_sort_decommits_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words,
                                         const uint64_t* initial_previous_tails, const uint64_t* sorted_previous_tails, const uint64_t* result_tails, uint32_t n_result_tails) {
    if (w && w->n_initial && (!initial_previous_tails || !sorted_previous_tails))
        return bad(ZK_ERR_INVALID, "zk_pack_sort_decommits_witness_tails: null tails (use zk_pack_sort_decommits_witness and device seeding)");
    if (n_result_tails && !result_tails) return bad(ZK_ERR_INVALID, "zk_pack_sort_decommits_witness_tails: null result tails");
    if (int rc = zk_pack_sort_decommits_witness(w, limit, instance, batch, outer_words, loop_words)) return rc;
    const zk_sort_decommits_fsm_witness& f = w->hidden_fsm_input;
    const zk_full_queue_state_witness& oq = w->start_flag ? w->initial_queue_state : f.initial_queue_state;
    const zk_full_queue_state_witness& sq0 = w->start_flag ? w->sorted_queue_initial_state : f.sorted_queue_state;
    uint64_t o_head[12], s_head[12], r_tail[12] = {0};
    uint64_t o_len = oq.length, s_len = sq0.length, r_len = 0;
    uint32_t prev_key[9] = {0}, first_ts = 0;
    zk_decommit_query_witness prev_record{};
    for (int k = 0; k < 12; ++k) { o_head[k] = oq.head[k]; s_head[k] = sq0.head[k]; }
    if (!w->start_flag) {
        for (int k = 0; k < 12; ++k) r_tail[k] = f.final_queue_state.tail[k];
        r_len = f.final_queue_state.length;
        for (int i = 0; i < 9; ++i) prev_key[i] = f.previous_packed_key[i];
        first_ts = f.first_encountered_timestamp; prev_record = f.previous_record;
    }
    bool prev_trivial = o_len == 0 || w->start_flag;
    uint32_t next = 0, next_result = 0;
    const zk_decommit_query_witness zero{};
    const size_t lanes = (size_t)batch * limit;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        const bool pops = o_len != 0 && next < w->n_initial;
        if (pops) for (int k = 0; k < 12; ++k) { o_head[k] = initial_previous_tails[12 * (size_t)next + k]; s_head[k] = sorted_previous_tails[12 * (size_t)next + k]; }
        l.w(prev_trivial ? 1 : 0);
        for (int i = 0; i < 4; ++i) l.w(0);   // lhs / rhs: k_decommit_seed
        l.arr(o_head); l.w(o_len); l.arr(s_head); l.w(s_len); l.arr(r_tail); l.w(r_len);
        l.arr(prev_key); l.w(first_ts); put_decommit(l, &prev_record);
        if (l.k != 65) return bad(ZK_ERR_INVALID, "internal: sort_decommits carried layout");
        const zk_decommit_query_witness& s = pops ? w->sorted_queue_witness[next] : zero;
        if (pops) {
            ++next; --o_len; if (s_len) --s_len;
            // the heads after the pop: the previous tails of the next elements, the queues' tails once empty, else what the FSM hands on
            for (int k = 0; k < 12; ++k) {
                o_head[k] = next < w->n_initial ? initial_previous_tails[12 * (size_t)next + k] : o_len == 0 ? oq.tail[k] : w->hidden_fsm_output.initial_queue_state.head[k];
                s_head[k] = next < w->n_sorted ? sorted_previous_tails[12 * (size_t)next + k] : s_len == 0 ? sq0.tail[k] : w->hidden_fsm_output.sorted_queue_state.head[k];
            }
        }
        bool same_hash = true;
        for (int i = 0; i < 8; ++i) same_hash &= prev_record.code_hash[i] == s.code_hash[i];
        if (!prev_trivial && !same_hash) {   // the previous record was the last of its hash: it goes to the result queue (mod.rs:318-333)
            if (next_result >= n_result_tails) return bad(ZK_ERR_INVALID, "zk_pack_sort_decommits_witness_tails: fewer result tails than pushes");
            for (int k = 0; k < 12; ++k) r_tail[k] = result_tails[12 * (size_t)next_result + k];
            ++next_result; ++r_len;
        }
        prev_trivial = !pops;
        if (!same_hash) first_ts = s.timestamp;
        prev_record = s; prev_record.is_first = s.is_first ? 1 : 0;
        prev_key[0] = s.timestamp;
        for (int i = 0; i < 8; ++i) prev_key[1 + i] = s.code_hash[i];
    }
    return ZK_OK;
}
uint32_t zk_sort_decommits_given_words(uint32_t words[65]) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < 65; ++i)
        if (!(i >= 1 && i <= 4)) words[n++] = i;
    return n;
}

int zk_pack_code_unpacker_witness(const zk_code_unpacker_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_code_unpacker_witness: bad argument");
    if ((w->n_requests && !w->sorted_requests_queue_witness) || (w->n_code_words && !w->code_words)) return bad(ZK_ERR_INVALID, "zk_pack_code_unpacker_witness: null witness array");
    Out o{outer_words + instance, batch};
    o.w(w->start_flag ? 1 : 0);
    put_full(o, w->sorted_requests_queue_initial_state); put_full(o, w->memory_queue_initial_state);   // the circuit's stream order
    const zk_code_unpacker_fsm_witness& f = w->hidden_fsm_input;   // CodeDecommittmentFSM (input.rs:23-34), then the two queue states (:61-65)
    o.arr(f.sha256_inner_state); o.arr(f.hash_to_compare_against);
    o.w(f.current_index); o.w(f.current_page); o.w(f.timestamp); o.w(f.num_rounds_left); o.w(f.length_in_bits);
    o.w(f.state_get_from_queue ? 1 : 0); o.w(f.state_decommit ? 1 : 0); o.w(f.finished ? 1 : 0);
    put_full(o, f.decommittment_requests_queue_state); put_full(o, f.memory_queue_state);
    if (o.k != ZK_CODE_UNPACKER_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: code_unpacker outer layout");
    // the FSM's schedule (mod.rs:167-255 pop and parameters, :257-300 the two words of a round, :402-430 next flags)
    bool get, decommit, finished;
    uint64_t rounds_left, req_len;
    if (w->start_flag) { get = true; decommit = false; finished = false; rounds_left = 0; req_len = w->sorted_requests_queue_initial_state.length; }
    else { get = f.state_get_from_queue; decommit = f.state_decommit; finished = f.finished; rounds_left = f.num_rounds_left; req_len = f.decommittment_requests_queue_state.length; }
    uint32_t next_req = 0, next_word = 0;
    const size_t lanes = (size_t)batch * limit;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        for (int i = 0; i < 74; ++i) l.w(0);
        const zk_decommit_query_witness* req = nullptr;
        if (get) {
            if (req_len != 0) {
                if (next_req >= w->n_requests) return bad(ZK_ERR_INVALID, "zk_pack_code_unpacker_witness: the request queue witness is shorter than its length");
                req = &w->sorted_requests_queue_witness[next_req++];
                --req_len;
            }
            const uint32_t top = req ? req->code_hash[7] : 0;   // versioned hash: 0x01 0x00 | length in words (u16)
            rounds_left = ((uint64_t)(top & 0xffff) + 1) / 2;
        }
        put_decommit(l, req);
        decommit = decommit || get;
        get = false;
        if (decommit) rounds_left = (rounds_left - 1) & 0xffff;   // UInt16 subtraction in the circuit
        const bool last_round = rounds_left == 0, finalize = last_round && decommit, second = !last_round && decommit;
        for (int r = 0; r < 2; ++r) {
            const bool take = r == 0 ? decommit : second;
            if (take && next_word < w->n_code_words) { for (int i = 0; i < 8; ++i) l.w(w->code_words[next_word][i]); ++next_word; }
            else for (int i = 0; i < 8; ++i) l.w(0);
        }
        const bool is_empty = req_len == 0;
        finished = finished || (is_empty && finalize);
        get = !is_empty && finalize;
        decommit = second;
        if (l.k != ZK_CODE_UNPACKER_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: code_unpacker loop layout");
    }
    (void)finished;
    return ZK_OK;
}

namespace {
// FIPS 180-4 compression on the host: the packer with tails walks the decommitter's SHA-256 state natively (one compression per cycle)
void sha256_compress_host(uint32_t st[8], const uint32_t block[16]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
        0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
        0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
        0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
        0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    auto rr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = block[i];
    for (int i = 16; i < 64; ++i)
        w[i] = w[i - 16] + (rr(w[i - 15], 7) ^ rr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (rr(w[i - 2], 17) ^ rr(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        const uint32_t t1 = h + (rr(e, 6) ^ rr(e, 11) ^ rr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        const uint32_t t2 = (rr(a, 2) ^ rr(a, 13) ^ rr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
}  // namespace

// The decommitter with the queue states its neighbours' witnesses hold: the requests queue's witness is a VecDeque of (DecommitQuery,
// previous tail) (input.rs:134-140: FullStateCircuitQueueRawWitness) — the 12-word head before each pop — and the memory queue it
// writes is the RAM permutation's unsorted queue, whose witness holds (MemoryQuery, previous tail) for every element, i.e. the tail
// after every push here (`memory_tails[k]` = the tail after the k-th code word of this instance went in).  Everything else the loop
// carries is integer state: the FSM's scalars and the SHA-256 inner state, one native compression per cycle (mod.rs:302-400).  All 74
// carried words of every cycle are written: nothing to seed on the device.
int zk_pack_code_unpacker_witness_tails(const zk_code_unpacker_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words,
                                        const uint64_t* request_previous_tails, const uint64_t* memory_tails) {
    if ((w && w->n_requests && !request_previous_tails) || (w && w->n_code_words && !memory_tails))
        return bad(ZK_ERR_INVALID, "zk_pack_code_unpacker_witness_tails: null tails (use zk_pack_code_unpacker_witness and device seeding)");
    if (int rc = zk_pack_code_unpacker_witness(w, limit, instance, batch, outer_words, loop_words)) return rc;
    const zk_code_unpacker_fsm_witness& f = w->hidden_fsm_input;
    static const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint32_t state[8] = {0}, hash_cmp[8] = {0}, index = 0, page = 0, timestamp = 0, length_in_bits = 0;
    uint64_t rounds_left = 0, req_len, mem_len;
    bool get, decommit, finished;
    uint64_t req_head[12], req_tail[12], mem_tail[12];
    const zk_full_queue_state_witness& rq = w->start_flag ? w->sorted_requests_queue_initial_state : f.decommittment_requests_queue_state;
    const zk_full_queue_state_witness& mq = w->start_flag ? w->memory_queue_initial_state : f.memory_queue_state;
    for (int k = 0; k < 12; ++k) { req_head[k] = rq.head[k]; req_tail[k] = rq.tail[k]; mem_tail[k] = mq.tail[k]; }
    req_len = rq.length; mem_len = mq.length;
    if (w->start_flag) { get = true; decommit = false; finished = false; }
    else {
        for (int i = 0; i < 8; ++i) { state[i] = f.sha256_inner_state[i]; hash_cmp[i] = f.hash_to_compare_against[i]; }
        index = f.current_index; page = f.current_page; timestamp = f.timestamp; rounds_left = f.num_rounds_left; length_in_bits = f.length_in_bits;
        get = f.state_get_from_queue; decommit = f.state_decommit; finished = f.finished;
    }
    uint32_t next_req = 0, next_word = 0;
    const size_t lanes = (size_t)batch * limit;
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        const bool pops = get && req_len != 0;
        if (pops) for (int k = 0; k < 12; ++k) req_head[k] = request_previous_tails[12 * (size_t)next_req + k];   // the head before this pop
        for (int i = 0; i < 8; ++i) for (int k = 0; k < 4; ++k) l.w((state[i] >> (8 * k)) & 0xff);
        l.arr(hash_cmp);
        l.w(index); l.w(page); l.w(timestamp); l.w(rounds_left); l.w(length_in_bits); l.w(get ? 1 : 0); l.w(decommit ? 1 : 0); l.w(finished ? 1 : 0);
        l.arr(req_head); l.w(req_len); l.arr(mem_tail); l.w(mem_len);
        if (l.k != 74) return bad(ZK_ERR_INVALID, "internal: code_unpacker carried layout");
        if (get) {
            const zk_decommit_query_witness* req = nullptr;
            if (pops) {
                req = &w->sorted_requests_queue_witness[next_req++];
                --req_len;
                // the head after the pop: the previous tail of the next request, the queue's tail once it is empty, else what the FSM hands on
                if (next_req < w->n_requests) for (int k = 0; k < 12; ++k) req_head[k] = request_previous_tails[12 * (size_t)next_req + k];
                else if (req_len == 0) for (int k = 0; k < 12; ++k) req_head[k] = req_tail[k];
                else for (int k = 0; k < 12; ++k) req_head[k] = w->hidden_fsm_output.decommittment_requests_queue_state.head[k];
            }
            const uint32_t top = req ? req->code_hash[7] : 0;
            const uint32_t length_in_words = top & 0xffff;
            rounds_left = ((uint64_t)length_in_words + 1) / 2;
            length_in_bits = length_in_words * 256;
            timestamp = req ? req->timestamp : 0; page = req ? req->page : 0;
            for (int i = 0; i < 7; ++i) hash_cmp[i] = req ? req->code_hash[i] : 0;
            hash_cmp[7] = 0;
            index = 0;
            for (int i = 0; i < 8; ++i) state[i] = IV[i];
        }
        decommit = decommit || get;
        get = false;
        if (decommit) rounds_left = (rounds_left - 1) & 0xffff;
        const bool last_round = rounds_left == 0, finalize = last_round && decommit, second = !last_round && decommit;
        uint32_t block[16] = {0};
        for (int r = 0; r < 2; ++r) {
            const bool take = r == 0 ? decommit : second;
            if (!take) continue;
            if (next_word < w->n_code_words) {
                for (int i = 0; i < 8; ++i) block[8 * r + i] = w->code_words[next_word][7 - i];
                for (int k = 0; k < 12; ++k) mem_tail[k] = memory_tails[12 * (size_t)next_word + k];
                ++next_word;
            } else return bad(ZK_ERR_INVALID, "zk_pack_code_unpacker_witness_tails: the code witness is shorter than the schedule (no tail for the push of a zero word)");
            ++mem_len; ++index;
        }
        if (finalize) { block[8] = 0x80000000u; for (int i = 9; i < 15; ++i) block[i] = 0; block[15] = length_in_bits; }
        if (decommit) sha256_compress_host(state, block);
        const bool is_empty = req_len == 0;
        finished = finished || (is_empty && finalize);
        get = !is_empty && finalize;
        decommit = second;
    }
    return ZK_OK;
}
// The SHA-256 precompile FSM with the queue states its neighbours' witnesses hold: the request queue's witness is a VecDeque of
// (LogQuery, previous tail) (input.rs:85-89: CircuitQueueRawWitness) — the 4-word head before each pop — and the memory queue it
// writes (two reads per round, one digest write per call) is the RAM permutation's unsorted queue, whose witness holds the tail
// after every push here.  The rest of the 60 carried words is integer state: flags, call parameters, timestamps and the SHA-256
// inner state, one native compression per cycle (mod.rs:199-340).  Nothing is left to seed: the chain of 2 790 dependent
// permutations per instance that the device pass spends 38 ms on is data the host already has.
int zk_pack_sha256_witness_tails(const zk_sha256_round_function_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words,
                                 const uint64_t* request_previous_tails, const uint64_t* memory_tails, uint32_t n_memory_tails) {
    if ((w && w->n_requests && !request_previous_tails) || (n_memory_tails && !memory_tails))
        return bad(ZK_ERR_INVALID, "zk_pack_sha256_witness_tails: null tails (use zk_pack_sha256_witness and device seeding)");
    if (int rc = zk_pack_sha256_witness(w, limit, instance, batch, outer_words, loop_words)) return rc;
    static const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const zk_sha256_fsm_witness& f = w->hidden_fsm_input;
    const uint64_t P = 0xFFFFFFFF00000001ull;
    bool rpc, rwfr, completed;
    uint32_t state[8];
    uint64_t ts_read = 0, ts_write = 0, input_page = 0, input_offset = 0, output_page = 0, output_offset = 0, num_rounds = 0;
    const zk_queue_state_witness& rq = w->start_flag ? w->initial_log_queue_state : f.log_queue_state;
    const zk_full_queue_state_witness& mq = w->start_flag ? w->initial_memory_queue_state : f.memory_queue_state;
    uint64_t req_head[4], req_len = rq.length, mem_tail[12], mem_len = mq.length;
    for (int k = 0; k < 4; ++k) req_head[k] = rq.head[k];
    for (int k = 0; k < 12; ++k) mem_tail[k] = mq.tail[k];
    if (w->start_flag) { rpc = true; rwfr = false; completed = false; for (int i = 0; i < 8; ++i) state[i] = IV[i]; }
    else {
        rpc = f.read_precompile_call; rwfr = f.read_words_for_round; completed = f.completed;
        for (int i = 0; i < 8; ++i) state[i] = f.sha256_inner_state[i];
        ts_read = f.timestamp_to_use_for_read; ts_write = f.timestamp_to_use_for_write;
        input_page = f.input_page; input_offset = f.input_offset; output_page = f.output_page; output_offset = f.output_offset; num_rounds = f.num_rounds;
    }
    if (rpc && req_len == 0) { rpc = false; rwfr = false; completed = true; }   // can_finish_immediatelly (mod.rs:120-137)
    uint32_t next_req = 0, next_read = 0, next_push = 0;
    const size_t lanes = (size_t)batch * limit;
    auto push = [&]() -> bool {
        if (next_push >= n_memory_tails) return false;
        for (int k = 0; k < 12; ++k) mem_tail[k] = memory_tails[12 * (size_t)next_push + k];
        ++next_push; ++mem_len;
        return true;
    };
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        const bool pops = rpc && req_len != 0;
        if (pops) for (int k = 0; k < 4; ++k) req_head[k] = request_previous_tails[4 * (size_t)next_req + k];   // the head before this pop
        l.w(rpc ? 1 : 0); l.w(rwfr ? 1 : 0); l.w(completed ? 1 : 0);
        for (int i = 0; i < 8; ++i) for (int k = 0; k < 4; ++k) l.w((state[i] >> (8 * k)) & 0xff);
        l.w(ts_read); l.w(ts_write); l.w(input_page); l.w(input_offset); l.w(output_page); l.w(output_offset); l.w(num_rounds);
        l.arr(req_head); l.w(req_len); l.arr(mem_tail); l.w(mem_len);
        if (l.k != 60) return bad(ZK_ERR_INVALID, "internal: sha256 carried layout");
        if (rpc) {
            const zk_log_query_witness* call = nullptr;
            if (pops) {
                call = &w->requests_queue_witness[next_req++];
                --req_len;
                for (int k = 0; k < 4; ++k)
                    req_head[k] = next_req < w->n_requests ? request_previous_tails[4 * (size_t)next_req + k] : req_len == 0 ? rq.tail[k] : w->hidden_fsm_output.log_queue_state.head[k];
            }
            input_offset = call ? call->key[0] : 0; output_offset = call ? call->key[2] : 0; input_page = call ? call->key[4] : 0;
            output_page = call ? call->key[5] : 0; num_rounds = call ? call->key[6] : 0;
            ts_read = call ? call->timestamp : 0;
            ts_write = ts_read + 1;
        }
        const bool reset_buffer = rpc || completed;
        rwfr = rpc || rwfr;
        rpc = false;
        const bool should_read = num_rounds != 0;
        uint32_t block[16] = {0};
        for (int r = 0; r < 2; ++r) {
            if (should_read && next_read < w->n_reads) { for (int i = 0; i < 8; ++i) block[8 * r + i] = w->memory_reads_witness[next_read][7 - i]; ++next_read; }
            if (should_read && !push()) return bad(ZK_ERR_INVALID, "zk_pack_sha256_witness_tails: fewer memory tails than pushes");
            if (rwfr) input_offset += 1;
        }
        if (rwfr) num_rounds = num_rounds ? num_rounds - 1 : P - 1;
        if (reset_buffer) for (int i = 0; i < 8; ++i) state[i] = IV[i];
        sha256_compress_host(state, block);
        const bool write_result = rwfr && num_rounds == 0;
        if (write_result && !push()) return bad(ZK_ERR_INVALID, "zk_pack_sha256_witness_tails: fewer memory tails than pushes");
        const bool input_is_empty = req_len == 0;
        rpc = write_result && !input_is_empty;
        completed = (write_result && input_is_empty) || completed;
        rwfr = !(rpc || completed);
    }
    return ZK_OK;
}
namespace {
void keccak_f1600_host(uint64_t s[25]) {   // FIPS 202
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
        0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
        0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int RHO[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PI[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    auto rotl = [](uint64_t x, int n) { return (x << n) | (x >> (64 - n)); };
    for (int r = 0; r < 24; ++r) {
        uint64_t c[5];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) { const uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1); for (int y = 0; y < 25; y += 5) s[y + x] ^= d; }
        uint64_t t = s[1];
        for (int i = 0; i < 24; ++i) { const int j = PI[i]; const uint64_t b = s[j]; s[j] = rotl(t, RHO[i]); t = b; }
        for (int y = 0; y < 25; y += 5) {
            uint64_t row[5];
            for (int x = 0; x < 5; ++x) row[x] = s[y + x];
            for (int x = 0; x < 5; ++x) s[y + x] = row[x] ^ (~row[(x + 1) % 5] & row[(x + 2) % 5]);
        }
        s[0] ^= RC[r];
    }
}
}  // namespace

namespace {
// BLS12-381 scalar field arithmetic for the packer that walks eip_4844's Horner recurrence: 256-bit little-endian limbs, product by
// schoolbook, reduction by binary long division (4 096 reductions per blob: speed is irrelevant here)
struct U256 { uint64_t l[4]; };
const U256 BLS_FR = {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};
bool ge256(const U256& a, const U256& b) { for (int i = 3; i >= 0; --i) if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; return true; }
void sub256(U256& a, const U256& b) { unsigned __int128 br = 0; for (int i = 0; i < 4; ++i) { const unsigned __int128 t = (unsigned __int128)a.l[i] - b.l[i] - br; a.l[i] = (uint64_t)t; br = (t >> 64) & 1; } }
U256 mulmod_fr(const U256& a, const U256& b) {
    uint64_t prod[8] = {0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 carry = 0;
        for (int j = 0; j < 4; ++j) { const unsigned __int128 t = (unsigned __int128)a.l[i] * b.l[j] + prod[i + j] + carry; prod[i + j] = (uint64_t)t; carry = t >> 64; }
        prod[i + 4] = (uint64_t)carry;
    }
    U256 r = {{0, 0, 0, 0}};
    for (int bit = 511; bit >= 0; --bit) {   // r < FR < 2^255 before the shift
        for (int i = 3; i > 0; --i) r.l[i] = (r.l[i] << 1) | (r.l[i - 1] >> 63);
        r.l[0] = (r.l[0] << 1) | ((prod[bit / 64] >> (bit % 64)) & 1);
        if (ge256(r, BLS_FR)) sub256(r, BLS_FR);
    }
    return r;
}
// a * b * 2^-256 mod FR (CIOS Montgomery, a < 2^256, b < FR): with b = z * 2^256 mod FR this is a * z mod FR — the 4 096 Horner steps of a
// blob cost ~0.2 ms instead of the 11 ms of the bit-serial reduction
U256 montmul_fr(const U256& a, const U256& b) {
    static uint64_t ninv = 0;   // -FR^-1 mod 2^64
    if (!ninv) { uint64_t x = 1; for (int i = 0; i < 6; ++i) x *= 2 - BLS_FR.l[0] * x; ninv = 0 - x; }
    uint64_t t[6] = {0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; ++j) { const unsigned __int128 u = (unsigned __int128)a.l[i] * b.l[j] + t[j] + c; t[j] = (uint64_t)u; c = u >> 64; }
        unsigned __int128 u = (unsigned __int128)t[4] + c; t[4] = (uint64_t)u; t[5] = (uint64_t)(u >> 64);
        const uint64_t m = t[0] * ninv;
        c = ((unsigned __int128)m * BLS_FR.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { const unsigned __int128 v = (unsigned __int128)m * BLS_FR.l[j] + t[j] + c; t[j - 1] = (uint64_t)v; c = v >> 64; }
        u = (unsigned __int128)t[4] + c; t[3] = (uint64_t)u; t[4] = t[5] + (uint64_t)(u >> 64); t[5] = 0;
    }
    U256 r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || ge256(r, BLS_FR)) sub256(r, BLS_FR);
    return r;
}
void keccak256_host(const uint8_t* data, size_t n, uint8_t out[32]) {
    uint64_t st[25] = {0};
    size_t at = 0;
    for (;;) {
        const size_t take = n - at < 136 ? n - at : 136;
        uint8_t block[136] = {0};
        for (size_t j = 0; j < take; ++j) block[j] = data[at + j];
        const bool last = take < 136;
        if (last) { block[take] |= 0x01; block[135] |= 0x80; }
        for (int j = 0; j < 136; ++j) st[j / 8] ^= (uint64_t)block[j] << (8 * (j % 8));
        keccak_f1600_host(st);
        at += take;
        if (last) break;
    }
    for (int j = 0; j < 32; ++j) out[j] = (uint8_t)(st[j / 8] >> (8 * (j % 8)));
}
}  // namespace

// eip_4844 with every carried word written by the host: the sponge state before each of the blob's Keccak blocks and the opening limbs
// before each iteration's Horner steps (lazy limb-wise addition of the chunk, then x z mod the BLS12-381 scalar field: mod.rs:150-235),
// z = the last 16 bytes of keccak256(linear_hash_output | versioned_hash).  The reference computes exactly these values out of circuit
// on the CPU (its test, mod.rs:595-683); the device pass this replaces walks ~940 dependent Keccak-f per blob.
int zk_pack_eip4844_witness_full(const zk_eip4844_witness* w, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (int rc = zk_pack_eip4844_witness(w, instance, batch, outer_words, loop_words)) return rc;
    uint32_t n_it = 0, lw = 0;
    zk_eip4844_stream_shape(w->n_chunks, &n_it, &lw);
    const uint32_t cpi = (lw - 217 - 136) / 31;
    const uint64_t n_bytes = 31ull * w->n_chunks;
    uint8_t zin[64], zh[32];
    for (int i = 0; i < 32; ++i) { zin[i] = w->linear_hash_output[i]; zin[32 + i] = w->versioned_hash[i]; }
    keccak256_host(zin, 64, zh);
    U256 z = {{0, 0, 0, 0}};   // big-endian integer of the last 16 bytes
    for (int i = 0; i < 16; ++i) z.l[(15 - i) / 8] |= (uint64_t)zh[16 + i] << (8 * ((15 - i) % 8));
    // z in Montgomery form: z * 2^256 mod FR = (z * (2^128 mod FR)) * (2^128 mod FR), by the bit-serial routine (once per blob)
    const U256 two128 = {{0, 0, 1, 0}};
    const U256 zM = mulmod_fr(mulmod_fr(z, two128), two128);
    uint64_t state[25] = {0}, opening[16] = {0};
    const size_t lanes = (size_t)batch * n_it;
    for (uint32_t t = 0; t < n_it; ++t) {
        Out l{loop_words + (size_t)instance * n_it + t, lanes};
        for (int lane = 0; lane < 25; ++lane) for (int k = 0; k < 8; ++k) l.w((state[lane] >> (8 * k)) & 0xff);
        l.arr(opening); l.w(t);
        if (l.k != 217) return bad(ZK_ERR_INVALID, "internal: eip_4844 carried layout");
        for (uint32_t c = 0; c < cpi; ++c) {
            const uint64_t idx = (uint64_t)cpi * t + c;
            if (idx >= w->n_chunks) continue;
            const uint8_t* ch = w->data_chunks + 31 * idx;   // little-endian integer of 31 bytes: 16-bit limbs
            for (int i = 0; i < 16; ++i) opening[i] += (uint64_t)ch[2 * i] | (2 * i + 1 < 31 ? (uint64_t)ch[2 * i + 1] << 8 : 0);
            if (idx != (uint64_t)w->n_chunks - 1) {
                U256 v = {{0, 0, 0, 0}};
                unsigned __int128 acc = 0;
                for (int i = 0; i < 16; ++i) {   // limbs of at most 17 bits -> the integer, carries propagated
                    acc += (unsigned __int128)opening[i] << (16 * (i % 4));
                    if (i % 4 == 3) { v.l[i / 4] = (uint64_t)acc; acc >>= 64; }
                }
                const U256 r = montmul_fr(v, zM);
                for (int i = 0; i < 16; ++i) opening[i] = (r.l[i / 4] >> (16 * (i % 4))) & 0xffff;
            }
        }
        for (uint32_t j = 0; j < 136; ++j) {
            const uint64_t at = 136ull * t + j;
            uint8_t b = at < n_bytes ? w->data_chunks[at] : 0;
            if (at == n_bytes) b |= 0x01;
            if (at == 136ull * n_it - 1) b |= 0x80;
            state[j / 8] ^= (uint64_t)b << (8 * (j % 8));
        }
        keccak_f1600_host(state);
    }
    return ZK_OK;
}
uint32_t zk_eip4844_given_words(uint32_t words[217]) {
    for (uint32_t i = 0; i < 217; ++i) words[i] = i;
    return 217;
}

// The Keccak-256 precompile FSM with the queue states its neighbours' witnesses hold (as zk_pack_sha256_witness_tails): request queue
// heads from the witness's previous tails, memory queue tails from the RAM permutation's witness (up to six reads and one digest write
// per cycle, in that order).  The rest of the 423 carried words is integer state walked here: flags, call parameters, the 192-byte
// ByteBuffer (fill_with_bytes / consume: buffer/mod.rs:69-163) and the sponge state, one native Keccak-f per cycle (mod.rs:497-590).
int zk_pack_keccak_witness_tails(const zk_keccak_round_function_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words,
                                 const uint64_t* request_previous_tails, const uint64_t* memory_tails, uint32_t n_memory_tails) {
    if ((w && w->n_requests && !request_previous_tails) || (n_memory_tails && !memory_tails))
        return bad(ZK_ERR_INVALID, "zk_pack_keccak_witness_tails: null tails (use zk_pack_keccak_witness and device seeding)");
    if (int rc = zk_pack_keccak_witness(w, limit, instance, batch, outer_words, loop_words)) return rc;
    constexpr uint32_t RATE = 136, BUF = 192, READS = 6;
    const zk_keccak_fsm_witness& f = w->hidden_fsm_input;
    bool rpc, ruw, padding_round, completed, needs_full = false;
    uint64_t state[25] = {0};
    uint8_t buffer[BUF] = {0};
    uint64_t ts_read = 0, ts_write = 0, input_page = 0, byte_offset = 0, byte_length = 0, output_page = 0, output_word_offset = 0, filled = 0;
    const zk_queue_state_witness& rq = w->start_flag ? w->initial_log_queue_state : f.log_queue_state;
    const zk_full_queue_state_witness& mq = w->start_flag ? w->initial_memory_queue_state : f.memory_queue_state;
    uint64_t req_head[4], req_len = rq.length, mem_tail[12], mem_len = mq.length;
    for (int k = 0; k < 4; ++k) req_head[k] = rq.head[k];
    for (int k = 0; k < 12; ++k) mem_tail[k] = mq.tail[k];
    if (w->start_flag) { rpc = true; ruw = padding_round = completed = false; }
    else {
        rpc = f.read_precompile_call; ruw = f.read_unaligned_words_for_round; padding_round = f.padding_round; completed = f.completed;
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) for (int k = 0; k < 8; ++k) state[i + 5 * j] |= (uint64_t)f.keccak_internal_state[i][j][k] << (8 * k);
        ts_read = f.timestamp_to_use_for_read; ts_write = f.timestamp_to_use_for_write;
        input_page = f.input_page; byte_offset = f.input_memory_byte_offset; byte_length = f.input_memory_byte_length;
        output_page = f.output_page; output_word_offset = f.output_word_offset; needs_full = f.needs_full_padding_round;
        for (uint32_t j = 0; j < BUF; ++j) buffer[j] = f.buffer_bytes[j];
        filled = f.buffer_filled;
    }
    if (rpc && req_len == 0) { rpc = false; ruw = false; completed = true; }   // can_finish_immediatelly (mod.rs:200-213)
    uint32_t next_req = 0, next_read = 0, next_push = 0;
    const size_t lanes = (size_t)batch * limit;
    auto push = [&]() -> bool {
        if (next_push >= n_memory_tails) return false;
        for (int k = 0; k < 12; ++k) mem_tail[k] = memory_tails[12 * (size_t)next_push + k];
        ++next_push; ++mem_len;
        return true;
    };
    for (uint32_t c = 0; c < limit; ++c) {
        Out l{loop_words + (size_t)instance * limit + c, lanes};
        const bool pops = rpc && req_len != 0;
        if (pops) for (int k = 0; k < 4; ++k) req_head[k] = request_previous_tails[4 * (size_t)next_req + k];
        l.w(rpc ? 1 : 0); l.w(ruw ? 1 : 0); l.w(padding_round ? 1 : 0); l.w(completed ? 1 : 0);
        for (int lane = 0; lane < 25; ++lane) for (int k = 0; k < 8; ++k) l.w((state[lane] >> (8 * k)) & 0xff);
        l.w(ts_read); l.w(ts_write); l.w(input_page); l.w(byte_offset); l.w(byte_length); l.w(output_page); l.w(output_word_offset); l.w(needs_full ? 1 : 0);
        l.arr(buffer); l.w(filled);
        l.arr(req_head); l.w(req_len); l.arr(mem_tail); l.w(mem_len);
        if (l.k != 423) return bad(ZK_ERR_INVALID, "internal: keccak carried layout");
        const zk_log_query_witness* call = nullptr;
        if (pops) {
            call = &w->requests_queue_witness[next_req++];
            --req_len;
            for (int k = 0; k < 4; ++k)
                req_head[k] = next_req < w->n_requests ? request_previous_tails[4 * (size_t)next_req + k] : req_len == 0 ? rq.tail[k] : w->hidden_fsm_output.log_queue_state.head[k];
        }
        const uint64_t call_length = call ? call->key[1] : 0;
        if (rpc) {
            byte_offset = call ? call->key[0] : 0; byte_length = call_length; output_word_offset = call ? call->key[2] : 0;
            input_page = call ? call->key[4] : 0; output_page = call ? call->key[5] : 0;
            needs_full = call_length % RATE == 0;
            ts_read = call ? call->timestamp : 0;
            ts_write = ts_read + 1;
        }
        const bool reset_buffer = rpc || completed;
        if (rpc && call_length == 0) padding_round = true;
        if (rpc && call_length != 0) ruw = true;
        rpc = false;
        if (reset_buffer) { for (auto& b : buffer) b = 0; filled = 0; for (auto& x : state) x = 0; }
        for (uint32_t r = 0; r < READS; ++r) {
            const uint64_t unalignment = byte_offset % 32, at_most = 32 - unalignment;
            const uint64_t meaningful = byte_length < at_most ? byte_length : at_most;
            const bool should_read = meaningful != 0 && filled + meaningful <= BUF && ruw;
            uint8_t be[32] = {0};   // value.to_be_bytes()
            if (should_read && next_read < w->n_reads) {
                for (int m = 0; m < 32; ++m) be[m] = (uint8_t)(w->memory_reads_witness[next_read][7 - m / 4] >> (8 * (3 - m % 4)));
                ++next_read;
            }
            if (should_read) {
                if (!push()) return bad(ZK_ERR_INVALID, "zk_pack_keccak_witness_tails: fewer memory tails than pushes");
                byte_offset += meaningful; byte_length -= meaningful;
                for (uint64_t idx = 0; idx < meaningful; ++idx)   // fill_with_bytes: `meaningful` bytes from `unalignment` on, at position `filled`
                    if (filled + idx < BUF) buffer[filled + idx] = unalignment + idx < 32 ? be[unalignment + idx] : 0;
                filled += meaningful;
            }
        }
        const bool zero_bytes_left = byte_length == 0;
        const uint64_t currently_filled = filled;
        uint8_t block[RATE];
        for (uint32_t j = 0; j < RATE; ++j) block[j] = buffer[j];
        for (uint32_t j = 0; j < BUF; ++j) buffer[j] = j + RATE < BUF ? buffer[j + RATE] : 0;   // consume::<136>
        filled = filled >= RATE ? filled - RATE : 0;
        const bool buffer_now_empty = filled == 0;
        const bool apply_padding = zero_bytes_left && buffer_now_empty && ruw && !needs_full;
        if (apply_padding) {
            if (currently_filled < RATE - 1) block[currently_filled] = 0x01;
            block[RATE - 1] = currently_filled == RATE - 1 ? 0x81 : 0x80;
        }
        if (padding_round) { for (auto& b : block) b = 0; block[0] = 0x01; block[RATE - 1] = 0x80; }
        for (uint32_t j = 0; j < RATE; ++j) state[j / 8] ^= (uint64_t)block[j] << (8 * (j % 8));
        keccak_f1600_host(state);
        const bool write_result = apply_padding || padding_round;
        if (write_result && !push()) return bad(ZK_ERR_INVALID, "zk_pack_keccak_witness_tails: fewer memory tails than pushes");
        const bool input_is_empty = req_len == 0;
        rpc = write_result && !input_is_empty;
        completed = (write_result && input_is_empty) || completed;
        padding_round = ruw && zero_bytes_left && buffer_now_empty && needs_full;
        ruw = !(rpc || padding_round || completed);
    }
    return ZK_OK;
}
uint32_t zk_keccak_given_words(uint32_t words[423]) {
    for (uint32_t i = 0; i < 423; ++i) words[i] = i;
    return 423;
}

uint32_t zk_sha256_given_words(uint32_t words[60]) {
    for (uint32_t i = 0; i < 60; ++i) words[i] = i;
    return 60;
}

uint32_t zk_code_unpacker_given_words(uint32_t words[74]) {
    for (uint32_t i = 0; i < 74; ++i) words[i] = i;
    return 74;
}

int zk_pack_linear_hasher_witness(const zk_linear_hasher_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words) {
    if (!w || !outer_words || !loop_words || !limit || instance >= batch) return bad(ZK_ERR_INVALID, "zk_pack_linear_hasher_witness: bad argument");
    if (limit % ZK_LINEAR_HASHER_PERIOD) return bad(ZK_ERR_INVALID, "zk_pack_linear_hasher_witness: limit must be a multiple of 17");
    if (w->n_queue > limit) return bad(ZK_ERR_INVALID, "zk_pack_linear_hasher_witness: more queue elements than cycles");
    if (w->n_queue && !w->queue_witness) return bad(ZK_ERR_INVALID, "zk_pack_linear_hasher_witness: null queue witness");
    Out o{outer_words + instance, batch};
    o.w(w->start_flag ? 1 : 0);
    o.qstate(w->queue_state);
    if (o.k != ZK_LINEAR_HASHER_OUTER_WORDS) return bad(ZK_ERR_INVALID, "internal: linear_hasher outer layout");
    const uint32_t iters = limit / ZK_LINEAR_HASHER_PERIOD;
    const size_t lanes = (size_t)batch * iters;
    for (uint32_t it = 0; it < iters; ++it) {
        Out l{loop_words + (size_t)instance * iters + it, lanes};
        for (int i = 0; i < 206; ++i) l.w(0);
        for (uint32_t c = 0; c < ZK_LINEAR_HASHER_PERIOD; ++c) {
            const uint32_t idx = it * ZK_LINEAR_HASHER_PERIOD + c;
            l.log_query(idx < w->n_queue ? &w->queue_witness[idx] : nullptr);
        }
        if (l.k != ZK_LINEAR_HASHER_LOOP_WORDS) return bad(ZK_ERR_INVALID, "internal: linear_hasher loop layout");
    }
    return ZK_OK;
}


// linear_hasher with the queue heads of its witness (the previous tail beside every element, input.rs:71-80) and the sponge walked on
// the host: the 206 carried words of every 17-cycle period (Keccak state, queue head, length, done flag) — LogQuery::into_bytes
// (log_query/mod.rs:645-686: 88 bytes) against 136-byte blocks, padding after the last element (mod.rs:120-200).
int zk_pack_linear_hasher_witness_tails(const zk_linear_hasher_witness* w, uint32_t limit, uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words,
                                        const uint64_t* queue_previous_tails) {
    if (w && w->n_queue && !queue_previous_tails) return bad(ZK_ERR_INVALID, "zk_pack_linear_hasher_witness_tails: null tails (use zk_pack_linear_hasher_witness and device seeding)");
    if (int rc = zk_pack_linear_hasher_witness(w, limit, instance, batch, outer_words, loop_words)) return rc;
    constexpr uint32_t RATE = 136, MSG = 88;
    const uint32_t iters = limit / ZK_LINEAR_HASHER_PERIOD;
    const size_t lanes = (size_t)batch * iters;
    uint64_t state[25] = {0}, head[4], length = w->queue_state.length;
    for (int k = 0; k < 4; ++k) head[k] = w->queue_state.head[k];
    bool done = length == 0;
    uint8_t buffer[RATE + MSG];
    uint32_t fill = 0, next = 0;
    auto absorb = [&](const uint8_t* block) { for (uint32_t j = 0; j < RATE; ++j) state[j / 8] ^= (uint64_t)block[j] << (8 * (j % 8)); keccak_f1600_host(state); };
    for (uint32_t it = 0; it < iters; ++it) {
        Out l{loop_words + (size_t)instance * iters + it, lanes};
        if (length != 0 && next < w->n_queue) for (int k = 0; k < 4; ++k) head[k] = queue_previous_tails[4 * (size_t)next + k];   // the head before this period's first pop
        for (int lane = 0; lane < 25; ++lane) for (int k = 0; k < 8; ++k) l.w((state[lane] >> (8 * k)) & 0xff);
        l.arr(head); l.w(length); l.w(done ? 1 : 0);
        if (l.k != 206) return bad(ZK_ERR_INVALID, "internal: linear_hasher carried layout");
        for (uint32_t c = 0; c < ZK_LINEAR_HASHER_PERIOD; ++c) {
            const bool should_pop = length != 0 && next < w->n_queue;
            uint8_t msg[MSG] = {0};
            if (should_pop) {
                const zk_log_query_witness& q = w->queue_witness[next++];
                --length;
                for (int k = 0; k < 4; ++k) head[k] = next < w->n_queue ? queue_previous_tails[4 * (size_t)next + k] : w->queue_state.tail[k];
                uint32_t at = 0;
                msg[at++] = q.shard_id; msg[at++] = q.is_service ? 1 : 0;
                msg[at++] = (uint8_t)(q.tx_number_in_block >> 8); msg[at++] = (uint8_t)q.tx_number_in_block;
                auto be = [&](const uint32_t* limbs, int n) { for (int i = n - 1; i >= 0; --i) for (int sh = 24; sh >= 0; sh -= 8) msg[at++] = (uint8_t)(limbs[i] >> sh); };
                be(q.address, 5); be(q.key, 8); be(q.written_value, 8);
            }
            const bool is_last = should_pop && length == 0;
            for (uint32_t j = 0; j < MSG; ++j) buffer[fill + j] = msg[j];
            fill += MSG;
            const bool cont = !done;
            if (fill >= RATE) {
                if (cont) absorb(buffer);
                for (uint32_t j = RATE; j < fill; ++j) buffer[j - RATE] = buffer[j];
                fill -= RATE;
            }
            if (cont && is_last) {
                uint8_t last[RATE] = {0};
                for (uint32_t j = 0; j < fill; ++j) last[j] = buffer[j];
                if (fill == RATE - 1) last[fill] = 0x81;
                else { last[fill] = 0x01; last[RATE - 1] = 0x80; }
                absorb(last);
            }
            done = done || is_last;
        }
    }
    return ZK_OK;
}
uint32_t zk_linear_hasher_given_words(uint32_t words[206]) {
    for (uint32_t i = 0; i < 206; ++i) words[i] = i;
    return 206;
}

namespace {
void storage_fsm(Cursor& c, zk_storage_fsm_witness& f) {   // StorageDeduplicatorFSMInputOutput (input.rs:37-52)
    for (auto& x : f.lhs_accumulator) x = c.field();
    for (auto& x : f.rhs_accumulator) x = c.field();
    c.queue_state4(f.current_unsorted_queue_state); c.queue_state4(f.current_intermediate_sorted_queue_state); c.queue_state4(f.current_final_sorted_queue_state);
    f.cycle_idx = c.u32();
    for (auto& x : f.previous_packed_key) x = c.u32();
    c.u256(f.previous_key); c.h160(f.previous_address); f.previous_timestamp = c.u32();
    f.this_cell_has_explicit_read_and_rollback_depth_zero = c.boolean();
    c.u256(f.this_cell_base_value); c.u256(f.this_cell_current_value); f.this_cell_current_depth = c.u32();
}
void log_sorter_fsm(Cursor& c, zk_log_sorter_fsm_witness& f) {   // EventsDeduplicatorFSMInputOutput (input.rs:28-36)
    for (auto& x : f.lhs_accumulator) x = c.field();
    for (auto& x : f.rhs_accumulator) x = c.field();
    c.queue_state4(f.initial_unsorted_queue_state); c.queue_state4(f.intermediate_sorted_queue_state); c.queue_state4(f.final_result_queue_state);
    f.previous_key = c.u32();
    c.log_query(f.previous_item);
}
int finish(Cursor& c, int err, size_t* consumed, const char* what) {
    if (err != ZK_OK) return bad(err, what);
    if (!c.ok) return bad(ZK_ERR_INVALID, what);
    if (consumed) *consumed = c.at;
    return ZK_OK;
}
}  // namespace

int zk_decode_storage_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_storage_validity_witness* out, zk_log_query_witness* unsorted_buf, uint32_t unsorted_cap,
                                      zk_timestamped_log_record_witness* sorted_buf, uint32_t sorted_cap, size_t* consumed) {
    return zk_decode_storage_witness_bincode_tails(bytes, n_bytes, out, unsorted_buf, unsorted_cap, sorted_buf, sorted_cap, nullptr, nullptr, consumed);
}
int zk_decode_storage_witness_bincode_tails(const uint8_t* bytes, size_t n_bytes, zk_storage_validity_witness* out, zk_log_query_witness* unsorted_buf, uint32_t unsorted_cap,
                                            zk_timestamped_log_record_witness* sorted_buf, uint32_t sorted_cap, uint64_t (*unsorted_tails)[4], uint64_t (*sorted_tails)[4],
                                            size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_storage_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    int err = ZK_OK;
    out->start_flag = c.boolean(); out->completion_flag = c.boolean();
    out->shard_id_to_process = c.u8(); c.queue_state4(out->unsorted_log_queue_state); c.queue_state4(out->intermediate_sorted_queue_state);
    zk_queue_state_witness final_sorted;   // observable_output.final_sorted_queue_state: read, not needed by the packer
    c.queue_state4(final_sorted);
    storage_fsm(c, out->hidden_fsm_input); storage_fsm(c, out->hidden_fsm_output);
    if (c.log_queue(unsorted_buf, unsorted_cap, out->n_unsorted, err, unsorted_tails)) {
        out->unsorted_queue_witness = unsorted_buf;
        out->unsorted_previous_tails = unsorted_tails;
        const uint64_t n = c.u64();   // CircuitQueueRawWitness<TimestampedStorageLogRecord, 4, ..>
        if (c.ok && (n > sorted_cap || (n && !sorted_buf))) err = ZK_ERR_CAPACITY;
        else {
            for (uint64_t i = 0; i < n && c.ok; ++i) {
                c.log_query(sorted_buf[i].record); sorted_buf[i].timestamp = c.u32();
                for (int t = 0; t < 4; ++t) { const uint64_t v = c.field(); if (sorted_tails) sorted_tails[i][t] = v; }
            }
            out->intermediate_sorted_queue_witness = sorted_buf; out->n_sorted = (uint32_t)n;
            out->sorted_previous_tails = sorted_tails;
        }
    }
    return finish(c, err, consumed, "zk_decode_storage_witness_bincode: truncated, malformed or longer than the caller's buffers");
}

int zk_decode_log_sorter_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_log_sorter_witness* out, zk_log_query_witness* initial_buf, uint32_t initial_cap,
                                         zk_log_query_witness* sorted_buf, uint32_t sorted_cap, size_t* consumed) {
    return zk_decode_log_sorter_witness_bincode_tails(bytes, n_bytes, out, initial_buf, initial_cap, sorted_buf, sorted_cap, nullptr, nullptr, consumed);
}
int zk_decode_log_sorter_witness_bincode_tails(const uint8_t* bytes, size_t n_bytes, zk_log_sorter_witness* out, zk_log_query_witness* initial_buf, uint32_t initial_cap,
                                               zk_log_query_witness* sorted_buf, uint32_t sorted_cap, uint64_t (*initial_tails)[4], uint64_t (*sorted_tails)[4],
                                               size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_log_sorter_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    int err = ZK_OK;
    out->start_flag = c.boolean(); out->completion_flag = c.boolean();
    c.queue_state4(out->initial_log_queue_state); c.queue_state4(out->intermediate_sorted_queue_state);
    zk_queue_state_witness final_queue;   // observable_output.final_queue_state
    c.queue_state4(final_queue);
    log_sorter_fsm(c, out->hidden_fsm_input); log_sorter_fsm(c, out->hidden_fsm_output);
    if (c.log_queue(initial_buf, initial_cap, out->n_initial, err, initial_tails)) {
        out->initial_queue_witness = initial_buf;
        out->initial_previous_tails = initial_tails;
        if (c.log_queue(sorted_buf, sorted_cap, out->n_sorted, err, sorted_tails)) { out->intermediate_sorted_queue_witness = sorted_buf; out->sorted_previous_tails = sorted_tails; }
    }
    return finish(c, err, consumed, "zk_decode_log_sorter_witness_bincode: truncated, malformed or longer than the caller's buffers");
}

int zk_decode_demux_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_demux_log_queue_witness* out, zk_log_query_witness* initial_buf, uint32_t initial_cap,
                                    size_t* consumed) {
    return zk_decode_demux_witness_bincode_tails(bytes, n_bytes, out, initial_buf, initial_cap, nullptr, consumed);
}
int zk_decode_demux_witness_bincode_tails(const uint8_t* bytes, size_t n_bytes, zk_demux_log_queue_witness* out, zk_log_query_witness* initial_buf, uint32_t initial_cap,
                                          uint64_t (*initial_tails)[4], size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_demux_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    int err = ZK_OK;
    out->start_flag = c.boolean(); out->completion_flag = c.boolean();
    c.queue_state4(out->initial_log_queue_state);
    zk_queue_state_witness outq;   // observable_output: the six output queue states (LogDemuxerOutputData, input.rs:77-84)
    for (int i = 0; i < 6; ++i) c.queue_state4(outq);
    for (zk_demux_fsm_witness* f : {&out->hidden_fsm_input, &out->hidden_fsm_output}) {
        c.queue_state4(f->initial_log_queue_state);
        for (auto& q : f->output_queue_states) c.queue_state4(q);
    }
    if (c.log_queue(initial_buf, initial_cap, out->n_initial, err, initial_tails)) out->initial_queue_witness = initial_buf;
    return finish(c, err, consumed, "zk_decode_demux_witness_bincode: truncated, malformed or longer than the caller's buffer");
}

int zk_decode_linear_hasher_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_linear_hasher_witness* out, zk_log_query_witness* queue_buf, uint32_t queue_cap,
                                            size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_linear_hasher_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    int err = ZK_OK;
    out->start_flag = c.boolean(); out->completion_flag = c.boolean();
    c.queue_state4(out->queue_state);
    for (int i = 0; i < 32; ++i) c.u8();   // observable_output.keccak256_hash; hidden FSM states are () : nothing on the wire
    if (c.log_queue(queue_buf, queue_cap, out->n_queue, err)) out->queue_witness = queue_buf;
    return finish(c, err, consumed, "zk_decode_linear_hasher_witness_bincode: truncated, malformed or longer than the caller's buffer");
}


namespace {
void decommit_query(Cursor& c, zk_decommit_query_witness& q) {   // DecommitQuery field order (decommit_query/mod.rs:22-29)
    c.u256(q.code_hash); q.page = c.u32(); q.is_first = c.boolean(); q.timestamp = c.u32();
}
bool decommit_queue(Cursor& c, zk_decommit_query_witness* buf, uint32_t cap, uint32_t& n_out, int& err, uint64_t (*tails)[12] = nullptr) {
    const uint64_t n = c.u64();
    if (!c.ok) return false;
    if (n > cap || (n && !buf)) { err = ZK_ERR_CAPACITY; return false; }
    for (uint64_t i = 0; i < n && c.ok; ++i) { decommit_query(c, buf[i]); for (int t = 0; t < 12; ++t) { const uint64_t v = c.field(); if (tails) tails[i][t] = v; } }
    n_out = (uint32_t)n;
    return c.ok;
}
bool u256_seq(Cursor& c, uint32_t (*buf)[8], uint32_t cap, uint32_t& n_out, int& err) {   // VecDeque<U256> / Vec<U256>, appended at n_out
    const uint64_t n = c.u64();
    if (!c.ok) return false;
    if (n > (uint64_t)cap - n_out || (n && !buf)) { err = ZK_ERR_CAPACITY; return false; }
    for (uint64_t i = 0; i < n && c.ok; ++i) c.u256(buf[n_out + i]);
    n_out += (uint32_t)n;
    return c.ok;
}
void sha256_fsm(Cursor& c, zk_sha256_fsm_witness& f) {   // Sha256RoundFunctionFSMInputOutput (input.rs:24-32, 53-57; call params mod.rs:44-50)
    f.read_precompile_call = c.boolean(); f.read_words_for_round = c.boolean(); f.completed = c.boolean();
    for (auto& x : f.sha256_inner_state) x = c.u32();
    f.timestamp_to_use_for_read = c.u32(); f.timestamp_to_use_for_write = c.u32();
    f.input_page = c.u32(); f.input_offset = c.u32(); f.output_page = c.u32(); f.output_offset = c.u32(); f.num_rounds = c.u32();
    c.queue_state4(f.log_queue_state); c.queue_state(f.memory_queue_state);
}
void keccak_fsm(Cursor& c, zk_keccak_fsm_witness& f) {   // Keccak256RoundFunctionFSMInputOutput (input.rs:29-39, 63-67; call params mod.rs:48-55)
    f.read_precompile_call = c.boolean(); f.read_unaligned_words_for_round = c.boolean(); f.padding_round = c.boolean(); f.completed = c.boolean();
    for (auto& i : f.keccak_internal_state) for (auto& j : i) for (auto& k : j) k = c.u8();
    f.timestamp_to_use_for_read = c.u32(); f.timestamp_to_use_for_write = c.u32();
    f.input_page = c.u32(); f.input_memory_byte_offset = c.u32(); f.input_memory_byte_length = c.u32(); f.output_page = c.u32(); f.output_word_offset = c.u32();
    f.needs_full_padding_round = c.boolean();
    for (auto& b : f.buffer_bytes) b = c.u8();
    f.buffer_filled = c.u8();
    c.queue_state4(f.log_queue_state); c.queue_state(f.memory_queue_state);
}
void sort_decommits_fsm(Cursor& c, zk_sort_decommits_fsm_witness& f) {   // CodeDecommittmentsDeduplicatorFSMInputOutput (input.rs:26-37)
    c.queue_state(f.initial_queue_state); c.queue_state(f.sorted_queue_state); c.queue_state(f.final_queue_state);
    for (auto& x : f.lhs_accumulator) x = c.field();
    for (auto& x : f.rhs_accumulator) x = c.field();
    for (auto& x : f.previous_packed_key) x = c.u32();
    f.first_encountered_timestamp = c.u32();
    decommit_query(c, f.previous_record);
}
void code_unpacker_fsm(Cursor& c, zk_code_unpacker_fsm_witness& f) {   // CodeDecommitterFSMInputOutput (input.rs:23-34, 61-65)
    for (auto& x : f.sha256_inner_state) x = c.u32();
    c.u256(f.hash_to_compare_against);
    f.current_index = c.u32(); f.current_page = c.u32(); f.timestamp = c.u32();
    if (c.need(2)) { uint16_t v; std::memcpy(&v, c.p + c.at, 2); c.at += 2; f.num_rounds_left = v; }
    f.length_in_bits = c.u32();
    f.state_get_from_queue = c.boolean(); f.state_decommit = c.boolean(); f.finished = c.boolean();
    c.queue_state(f.decommittment_requests_queue_state); c.queue_state(f.memory_queue_state);
}
}  // namespace

int zk_decode_sha256_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_sha256_round_function_witness* out, zk_log_query_witness* requests_buf, uint32_t requests_cap,
                                     uint32_t (*reads_buf)[8], uint32_t reads_cap, size_t* consumed) {
    return zk_decode_sha256_witness_bincode_tails(bytes, n_bytes, out, requests_buf, requests_cap, reads_buf, reads_cap, nullptr, consumed);
}
int zk_decode_sha256_witness_bincode_tails(const uint8_t* bytes, size_t n_bytes, zk_sha256_round_function_witness* out, zk_log_query_witness* requests_buf, uint32_t requests_cap,
                                           uint32_t (*reads_buf)[8], uint32_t reads_cap, uint64_t (*request_tails)[4], size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_sha256_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    int err = ZK_OK;
    out->start_flag = c.boolean(); out->completion_flag = c.boolean();
    c.queue_state4(out->initial_log_queue_state); c.queue_state(out->initial_memory_queue_state);
    zk_full_queue_state_witness final_memory;   // observable_output.final_memory_state
    c.queue_state(final_memory);
    sha256_fsm(c, out->hidden_fsm_input); sha256_fsm(c, out->hidden_fsm_output);
    if (c.log_queue(requests_buf, requests_cap, out->n_requests, err, request_tails)) {
        out->requests_queue_witness = requests_buf;
        if (u256_seq(c, reads_buf, reads_cap, out->n_reads, err)) out->memory_reads_witness = reads_buf;
    }
    return finish(c, err, consumed, "zk_decode_sha256_witness_bincode: truncated, malformed or longer than the caller's buffers");
}

int zk_decode_keccak_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_keccak_round_function_witness* out, zk_log_query_witness* requests_buf, uint32_t requests_cap,
                                     uint32_t (*reads_buf)[8], uint32_t reads_cap, size_t* consumed) {
    return zk_decode_keccak_witness_bincode_tails(bytes, n_bytes, out, requests_buf, requests_cap, reads_buf, reads_cap, nullptr, consumed);
}
int zk_decode_keccak_witness_bincode_tails(const uint8_t* bytes, size_t n_bytes, zk_keccak_round_function_witness* out, zk_log_query_witness* requests_buf, uint32_t requests_cap,
                                           uint32_t (*reads_buf)[8], uint32_t reads_cap, uint64_t (*request_tails)[4], size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_keccak_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    int err = ZK_OK;
    out->start_flag = c.boolean(); out->completion_flag = c.boolean();
    c.queue_state4(out->initial_log_queue_state); c.queue_state(out->initial_memory_queue_state);
    zk_full_queue_state_witness final_memory;
    c.queue_state(final_memory);
    keccak_fsm(c, out->hidden_fsm_input); keccak_fsm(c, out->hidden_fsm_output);
    if (c.log_queue(requests_buf, requests_cap, out->n_requests, err, request_tails)) {
        out->requests_queue_witness = requests_buf;
        if (u256_seq(c, reads_buf, reads_cap, out->n_reads, err)) out->memory_reads_witness = reads_buf;
    }
    return finish(c, err, consumed, "zk_decode_keccak_witness_bincode: truncated, malformed or longer than the caller's buffers");
}

int zk_decode_sort_decommits_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_sort_decommits_witness* out, zk_decommit_query_witness* initial_buf, uint32_t initial_cap,
                                             zk_decommit_query_witness* sorted_buf, uint32_t sorted_cap, size_t* consumed) {
    return zk_decode_sort_decommits_witness_bincode_tails(bytes, n_bytes, out, initial_buf, initial_cap, sorted_buf, sorted_cap, nullptr, nullptr, consumed);
}
int zk_decode_sort_decommits_witness_bincode_tails(const uint8_t* bytes, size_t n_bytes, zk_sort_decommits_witness* out, zk_decommit_query_witness* initial_buf,
                                                   uint32_t initial_cap, zk_decommit_query_witness* sorted_buf, uint32_t sorted_cap, uint64_t (*initial_tails)[12],
                                                   uint64_t (*sorted_tails)[12], size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_sort_decommits_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    int err = ZK_OK;
    out->start_flag = c.boolean(); out->completion_flag = c.boolean();
    c.queue_state(out->initial_queue_state); c.queue_state(out->sorted_queue_initial_state);
    zk_full_queue_state_witness final_queue;   // observable_output.final_queue_state
    c.queue_state(final_queue);
    sort_decommits_fsm(c, out->hidden_fsm_input); sort_decommits_fsm(c, out->hidden_fsm_output);
    if (decommit_queue(c, initial_buf, initial_cap, out->n_initial, err, initial_tails)) {
        out->initial_queue_witness = initial_buf;
        if (decommit_queue(c, sorted_buf, sorted_cap, out->n_sorted, err, sorted_tails)) out->sorted_queue_witness = sorted_buf;
    }
    return finish(c, err, consumed, "zk_decode_sort_decommits_witness_bincode: truncated, malformed or longer than the caller's buffers");
}

int zk_decode_code_unpacker_witness_bincode(const uint8_t* bytes, size_t n_bytes, zk_code_unpacker_witness* out, zk_decommit_query_witness* requests_buf, uint32_t requests_cap,
                                            uint32_t (*words_buf)[8], uint32_t words_cap, size_t* consumed) {
    return zk_decode_code_unpacker_witness_bincode_tails(bytes, n_bytes, out, requests_buf, requests_cap, words_buf, words_cap, nullptr, consumed);
}
int zk_decode_code_unpacker_witness_bincode_tails(const uint8_t* bytes, size_t n_bytes, zk_code_unpacker_witness* out, zk_decommit_query_witness* requests_buf,
                                                  uint32_t requests_cap, uint32_t (*words_buf)[8], uint32_t words_cap, uint64_t (*request_tails)[12], size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_code_unpacker_witness_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    int err = ZK_OK;
    out->start_flag = c.boolean(); out->completion_flag = c.boolean();
    c.queue_state(out->memory_queue_initial_state); c.queue_state(out->sorted_requests_queue_initial_state);   // CodeDecommitterInputData order (input.rs:80-83)
    zk_full_queue_state_witness final_memory;   // observable_output.memory_queue_final_state
    c.queue_state(final_memory);
    code_unpacker_fsm(c, out->hidden_fsm_input); code_unpacker_fsm(c, out->hidden_fsm_output);
    if (decommit_queue(c, requests_buf, requests_cap, out->n_requests, err, request_tails)) {
        out->sorted_requests_queue_witness = requests_buf;
        const uint64_t n_codes = c.u64();   // Vec<Vec<U256>>
        bool good = c.ok;
        for (uint64_t k = 0; k < n_codes && good; ++k) good = u256_seq(c, words_buf, words_cap, out->n_code_words, err);
        if (good) out->code_words = words_buf;
    }
    return finish(c, err, consumed, "zk_decode_code_unpacker_witness_bincode: truncated, malformed or longer than the caller's buffers");
}

}  // extern "C"


// ---- VmCircuitInputOutputWitness = ClosedFormInputWitness<VmLocalState, VmInputData, VmOutputData> (circuit_inputs/main_vm.rs:7-62), the
// closed-form half of VmCircuitWitness (its witness_oracle half is a host-defined type: the FIFOs of zk_vm_witness_oracle).
namespace {
// VmLocalStateWitness in serde's derive order (src/base_structures/vm_state/mod.rs:92-109) -> the 243 flattened words
void vm_local_state(Cursor& c, uint64_t* o) {
    uint32_t l8[8], l5[5];
    int n = 0;
    auto put256 = [&] { c.u256(l8); for (int i = 0; i < 8; ++i) o[n++] = l8[i]; };
    put256();                                                   // previous_code_word: UInt256 -> U256
    for (int r = 0; r < ZK_VM_REGISTERS; ++r) { o[n++] = c.boolean(); put256(); }   // registers[15]: VMRegister { is_pointer, value }
    for (int i = 0; i < 3; ++i) o[n++] = c.boolean();           // flags: overflow_or_less_than, equal, greater_than
    for (int i = 0; i < 4; ++i) o[n++] = c.u32();               // timestamp, memory_page_counter, tx_number_in_block, previous_code_page
    o[n++] = c.u16();                                           // previous_super_pc: UInt16
    o[n++] = c.boolean();                                       // pending_exception
    o[n++] = c.u32();                                           // ergs_per_pubdata_byte
    // callstack.current_context.saved_context: ExecutionContextRecord (saved_context.rs:37-68)
    for (int a = 0; a < 3; ++a) { c.h160(l5); for (int i = 0; i < 5; ++i) o[n++] = l5[i]; }   // this, caller, code_address: UInt160 -> Address
    for (int i = 0; i < 4; ++i) o[n++] = c.u32();               // code_page, base_page, heap_upper_bound, aux_heap_upper_bound
    for (int i = 0; i < 8; ++i) o[n++] = c.field();             // reverted_queue_head[4], reverted_queue_tail[4]
    o[n++] = c.u32();                                           // reverted_queue_segment_len
    for (int i = 0; i < 3; ++i) o[n++] = c.u16();               // pc, sp, exception_handler_loc
    o[n++] = c.u32();                                           // ergs_remaining
    for (int i = 0; i < 2; ++i) o[n++] = c.boolean();           // is_static_execution, is_kernel_mode
    for (int i = 0; i < 3; ++i) o[n++] = c.u8();                // this_shard_id, caller_shard_id, code_shard_id
    for (int i = 0; i < 4; ++i) o[n++] = c.u32();               // context_u128_value_composite
    o[n++] = c.boolean();                                       // is_local_call
    for (int i = 0; i < 4; ++i) o[n++] = c.field();             // current_context.log_queue_forward_tail
    o[n++] = c.u32();                                           // current_context.log_queue_forward_part_length
    o[n++] = c.u32();                                           // callstack.context_stack_depth
    for (int i = 0; i < 12; ++i) o[n++] = c.field();            // callstack.stack_sponge_state
    for (int i = 0; i < 12; ++i) o[n++] = c.field();            // memory_queue_state
    o[n++] = c.u32();                                           // memory_queue_length
    for (int i = 0; i < 12; ++i) o[n++] = c.field();            // code_decommittment_queue_state
    o[n++] = c.u32();                                           // code_decommittment_queue_length
    for (int i = 0; i < 4; ++i) o[n++] = c.u32();               // context_composite_u128
    if (n != 243) c.ok = false;
}
}  // namespace

int zk_decode_vm_closed_form_input_bincode(const uint8_t* bytes, size_t n_bytes, zk_vm_closed_form_input* out, zk_vm_closed_form_rest* rest, size_t* consumed) {
    if (!bytes || !out) return bad(ZK_ERR_INVALID, "zk_decode_vm_closed_form_input_bincode: null argument");
    Cursor c{bytes, n_bytes};
    std::memset(out, 0, sizeof *out);
    zk_vm_closed_form_rest tmp;
    zk_vm_closed_form_rest& r = rest ? *rest : tmp;
    std::memset(&r, 0, sizeof r);
    out->start_flag = c.boolean();
    r.completion_flag = c.boolean();
    // observable_input: VmInputData (circuit_inputs/main_vm.rs:9-17)
    for (auto& x : out->rollback_queue_tail_for_block) x = c.field();
    for (auto& x : out->memory_queue_initial_tail) x = c.field();           // QueueTailStateWitness { tail, length }
    out->memory_queue_initial_length = c.u32();
    for (auto& x : out->decommitment_queue_initial_tail) x = c.field();
    out->decommitment_queue_initial_length = c.u32();
    out->zkporter_is_available = c.boolean();                                // per_block_context: GlobalContext (vm_state/mod.rs:157-160)
    c.u256(out->default_aa_code_hash);
    // observable_output: VmOutputData (:32-38): three QueueStateWitness { head, tail { tail, length } }
    for (auto& x : r.log_queue_final_state.head) x = c.field();
    for (auto& x : r.log_queue_final_state.tail) x = c.field();
    r.log_queue_final_state.length = c.u32();
    for (zk_full_queue_state_witness* q : {&r.memory_queue_final_state, &r.decommitment_queue_final_state}) {
        for (auto& x : q->head) x = c.field();
        for (auto& x : q->tail) x = c.field();
        q->length = c.u32();
    }
    vm_local_state(c, out->hidden_fsm_input);
    vm_local_state(c, r.hidden_fsm_output);
    if (!c.ok) return bad(ZK_ERR_INVALID, "zk_decode_vm_closed_form_input_bincode: truncated or malformed input");
    if (consumed) *consumed = c.at;
    return ZK_OK;
}
