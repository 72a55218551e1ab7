// vm_pack.cpp — product-side input path of main_vm: zk_pack_main_vm_witness (include/zkgl_vm.h).
//
// VmCircuitWitness = { closed_form_input, witness_oracle } (/root/reference/src/fsm_input_output/circuit_inputs/main_vm.rs:64-71).
// The oracle's getters (src/main_vm/witness_oracle.rs:45-91) are FIFOs consumed only under `execute`; the recorded circuit reads every
// getter's answer from a fixed word of the cycle's stream column.  Which cycle pops which FIFO depends on the VM state, so the
// packer runs the native walker (vm_native.hpp: decode + the one opcode family that applies, no hashing) and writes each answer
// into the cycle that asked for it.  With ZK_VM_PACK_FILL_STATE it also runs the four Poseidon2 chains on the host and writes the
// VmLocalState of every cycle — the same words zk_cs_seed_stream derives on the device (kernels_vm_seed.hpp).
#include <cstring>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <string>
#include <vector>
#include "../../include/zkgl.h"
#include "../../include/zkgl_vm.h"
#include "cs.hpp"
#include "poseidon_consts.hpp"
#include "vm_native.hpp"

namespace zkgl { CS* cs_of(zk_cs* h); }
namespace zkgl { void set_last_error(const std::string& m); }

namespace {

using vmn::u32;
using vmn::u64;
constexpr u64 P = 0xFFFFFFFF00000001ull;

// ---- host Poseidon2 (Goldilocks, t = 12): same structure as poseidon2_device.hpp (M_E = circ(2 M4, M4, M4), M_I = J + diag(2^k))
inline u64 gl_reduce128(unsigned __int128 x) {
    const u64 lo = (u64)x, hi = (u64)(x >> 64);
    const u64 hi_hi = hi >> 32, hi_lo = hi & 0xffffffffull;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= 0xffffffffull;
    const u64 t1 = (hi_lo << 32) - hi_lo;
    u64 t2 = t0 + t1;
    if (t2 < t1) t2 += 0xffffffffull;
    return t2 >= P ? t2 - P : t2;
}
inline u64 gl_mul(u64 a, u64 b) { return gl_reduce128((unsigned __int128)a * b); }
inline u64 gl_add(u64 a, u64 b) { return gl_reduce128((unsigned __int128)a + b); }
inline u64 gl_pow7(u64 x) { const u64 x2 = gl_mul(x, x), x3 = gl_mul(x2, x), x4 = gl_mul(x2, x2); return gl_mul(x3, x4); }
void mds_external(u64 s[12]) {
    static const u32 M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
    unsigned __int128 t[12];
    for (int b = 0; b < 3; ++b)
        for (int r = 0; r < 4; ++r) {
            unsigned __int128 acc = 0;
            for (int c = 0; c < 4; ++c) acc += (unsigned __int128)s[4 * b + c] * M4[r][c];
            t[4 * b + r] = acc;
        }
    for (int r = 0; r < 4; ++r) {
        const unsigned __int128 sum = t[r] + t[4 + r] + t[8 + r];
        for (int b = 0; b < 3; ++b) s[4 * b + r] = gl_reduce128(t[4 * b + r] + sum);
    }
}
void mds_inner(u64 s[12]) {
    static const int SHIFT[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};
    unsigned __int128 sum = 0;
    for (int i = 0; i < 12; ++i) sum += s[i];
    for (int i = 0; i < 12; ++i) s[i] = gl_reduce128(sum + ((unsigned __int128)s[i] << SHIFT[i]));
}
thread_local size_t g_host_permutations = 0;
void poseidon2(u64 s[12]) {
    ++g_host_permutations;
    const u64* RC = zkgl::poseidon_round_constants();
    mds_external(s);
    for (int r = 0; r < 30; ++r) {
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; ++i) s[i] = gl_pow7(gl_add(s[i], RC[12 * r + i]));
            mds_external(s);
        } else {
            s[0] = gl_pow7(gl_add(s[0], RC[12 * r]));
            mds_inner(s);
        }
    }
}

struct Chains { u64 mem[12], dec[12], fwd[4], sponge[12]; };

// n words to a destination this core will not read again: non-temporal 8-byte stores (no read-for-ownership of the destination lines) for
// the WHOLE 64-byte lines of the run; the partial lines at its ends take plain stores.  A staging array that is not line-aligned (a numpy
// array: 16 bytes off) makes every 512-byte run start and end inside a line whose other part belongs to the neighbouring tile: streamed,
// those lines leave the write-combining buffers half-filled — measured here 0.65 ms per instance instead of 0.13 (round 6: the figure
// round 5 published, 0.98 ms per instance per core, was that case; pinned staging memory is page-aligned).
inline void stream_out(u64* dst, const u64* src, uint32_t n) {
#if defined(__x86_64__)
    uint32_t i = 0;
    const uint32_t head = (uint32_t)(((64u - (uint32_t)((uintptr_t)dst & 63u)) & 63u) / 8u);   // (dst is 8-byte aligned)
    for (; i < n && i < head; ++i) dst[i] = src[i];
    const uint32_t whole = i + ((n - i) & ~7u);
    for (; i < whole; ++i) _mm_stream_si64((long long*)(dst + i), (long long)src[i]);
    for (; i < n; ++i) dst[i] = src[i];
#else
    std::memcpy(dst, src, (size_t)n * sizeof(u64));
#endif
}
inline void stream_fence() {
#if defined(__x86_64__)
    _mm_sfence();
#endif
}

// Env of the walker: FIFOs in, raw stream words + (optionally) hash chains out
struct PackEnv {
    const zk_vm_witness_oracle* o;
    zk_vm_pack_report* rep;
    zkgl::CS* cs;
    u64* col;       // loop_words + instance * limit + cycle; word w at col[w * stride]
    u64 stride;
    u32 w_code_word, w_src0_value, w_src0_is_ptr, w_refund, w_log_read, w_log_prev_head, w_near_tail, w_far_code_hash, w_far_page, w_far_tail, w_ret_ctx,
        w_ret_state, w_uma_a, w_uma_b;
    bool chains_on;
    zk_vm_queue_states* qs = nullptr;   // ZK_VM_PACK_RECORD_STATES / ZK_VM_PACK_STATES_FROM_WITNESS
    int qs_mode = 0;                    // 0: hash, 1: hash and record, 2: read the tails from the witness's queue states
    bool qs_overflow = false;
    Chains ch;
    // a chain step whose result the witness generator holds: read it, or compute it (and record it for a fixture)
    template <int W, class F>
    void chained(u64* tail, u64 (*arr)[W], size_t n, size_t& used, F&& hash) {
        if (qs_mode == 2) {
            if (used >= n) { rep->underflow = 1; return; }
            for (int i = 0; i < W; ++i) tail[i] = arr[used][i];
            ++used;
            return;
        }
        hash();
        if (qs_mode == 1) {
            if (used >= n) { qs_overflow = true; return; }
            for (int i = 0; i < W; ++i) arr[used][i] = tail[i];
            ++used;
        }
    }
    void put(u32 w, u64 v) { col[(u64)w * stride] = v; }
    void opcode_row(const vmn::Defs& D, u32 variant, u32& price, u64& props) { price = D.prices[variant]; props = D.props[variant]; }
    void mem_read(bool exec, u32 w_value, int w_ptr, vmn::U256& v, u32* is_ptr) {
        v = vmn::u256_zero();
        if (is_ptr) *is_ptr = 0;
        if (!exec) return;
        if (rep->used_memory_reads >= o->n_memory_reads) { rep->underflow = 1; return; }
        const zk_vm_memory_witness& m = o->memory_reads[rep->used_memory_reads++];
        for (int i = 0; i < 8; ++i) { v.l[i] = m.value[i]; put(w_value + i, m.value[i]); }
        if (is_ptr) { *is_ptr = m.is_ptr ? 1 : 0; put((u32)w_ptr, *is_ptr); }
    }
    void code_word(bool exec, vmn::U256& v) { mem_read(exec, w_code_word, -1, v, nullptr); }
    void src0(bool exec, vmn::U256& v, u32& is_ptr) { mem_read(exec, w_src0_value, (int)w_src0_is_ptr, v, &is_ptr); }
    void uma_read(int which, bool exec, vmn::U256& v) { mem_read(exec, which ? w_uma_b : w_uma_a, -1, v, nullptr); }
    u32 refund(bool exec) {
        if (!exec) return 0;
        if (rep->used_refunds >= o->n_refunds) { rep->underflow = 1; return 0; }
        const u32 r = o->refunds[rep->used_refunds++];
        put(w_refund, r);
        return r;
    }
    void storage_read(bool exec, u32 w, vmn::U256& v) {
        v = vmn::u256_zero();
        if (!exec) return;
        if (rep->used_storage_reads >= o->n_storage_reads) { rep->underflow = 1; return; }
        const uint32_t* s = o->storage_reads[rep->used_storage_reads++];
        for (int i = 0; i < 8; ++i) { v.l[i] = s[i]; put(w + i, s[i]); }
    }
    void log_read(bool exec, vmn::U256& v) { storage_read(exec, w_log_read, v); }
    void far_code_hash(bool exec, vmn::U256& v) { storage_read(exec, w_far_code_hash, v); }
    void four(bool exec, const uint64_t (*q)[4], size_t n, size_t& used, u32 w, u64 out[4]) {
        for (int i = 0; i < 4; ++i) out[i] = 0;
        if (!exec) return;
        if (used >= n) { rep->underflow = 1; return; }
        for (int i = 0; i < 4; ++i) { out[i] = q[used][i]; put(w + i, out[i]); }
        ++used;
    }
    void log_prev_head(bool exec, u64 out[4]) { four(exec, o->rollback_queue_witness, o->n_rollback_queue_witness, rep->used_rollback_queue_witness, w_log_prev_head, out); }
    void near_call_tail(bool exec, u64 out[4]) { four(exec, o->rollback_tails_for_call, o->n_rollback_tails_for_call, rep->used_rollback_tails_for_call, w_near_tail, out); }
    void far_call_tail(bool exec, u64 out[4]) { four(exec, o->rollback_tails_for_call, o->n_rollback_tails_for_call, rep->used_rollback_tails_for_call, w_far_tail, out); }
    u32 far_decommit_page(bool exec) {
        if (!exec) return 0;
        if (rep->used_decommit_pages >= o->n_decommit_pages) { rep->underflow = 1; return 0; }
        const u32 p = o->decommit_pages[rep->used_decommit_pages++];
        put(w_far_page, p);
        return p;
    }
    void ret_pop(bool exec, u64 ctx42[42], u64 state[12]) {
        for (int i = 0; i < 42; ++i) ctx42[i] = 0;
        for (int i = 0; i < 12; ++i) state[i] = 0;
        if (!exec) return;
        if (rep->used_callstack >= o->n_callstack) { rep->underflow = 1; return; }
        const zk_vm_callstack_witness& c = o->callstack[rep->used_callstack++];
        for (int i = 0; i < 42; ++i) { ctx42[i] = c.context[i]; put(w_ret_ctx + i, c.context[i]); }
        for (int i = 0; i < 12; ++i) { state[i] = c.state[i]; put(w_ret_state + i, c.state[i]); }
    }
    // ---- the chains (src/main_vm/utils.rs:194-213; opcodes/log.rs:508-609; opcodes/call_ret.rs:170-270)
    void push12(u64 tail[12], const u64 enc[8]) {
        u64 s[12];
        for (int i = 0; i < 8; ++i) s[i] = enc[i];
        for (int i = 8; i < 12; ++i) s[i] = tail[i];
        poseidon2(s);
        std::memcpy(tail, s, sizeof s);
    }
    void mem_push(const u64 enc[8]) {
        if (!chains_on) return;
        if (!qs) { push12(ch.mem, enc); return; }
        chained<12>(ch.mem, qs->memory_tails, qs->n_memory_tails, qs->used_memory_tails, [&] { push12(ch.mem, enc); });
    }
    void dec_push(const u64 enc[8]) {
        if (!chains_on) return;
        if (!qs) { push12(ch.dec, enc); return; }
        chained<12>(ch.dec, qs->decommit_tails, qs->n_decommit_tails, qs->used_decommit_tails, [&] { push12(ch.dec, enc); });
    }
    void fwd_hash(const u64 enc[20]) {
        u64 s[12] = {0};
        for (int i = 0; i < 8; ++i) s[i] = enc[i];
        poseidon2(s);
        for (int i = 0; i < 8; ++i) s[i] = enc[8 + i];
        poseidon2(s);
        for (int i = 0; i < 4; ++i) { s[i] = enc[16 + i]; s[4 + i] = ch.fwd[i]; }
        poseidon2(s);
        for (int i = 0; i < 4; ++i) ch.fwd[i] = s[i];
    }
    void fwd_push(const u64 enc[20]) {
        if (!chains_on) return;
        if (!qs) { fwd_hash(enc); return; }
        chained<4>(ch.fwd, qs->log_forward_tails, qs->n_log_forward_tails, qs->used_log_forward_tails, [&] { fwd_hash(enc); });
    }
    void fwd_set(const u64 v[4]) { for (int i = 0; i < 4; ++i) ch.fwd[i] = v[i]; }
    void sponge_push(const u64 enc[32]) {
        if (!chains_on) return;
        for (int r = 0; r < 4; ++r) {
            for (int i = 0; i < 8; ++i) ch.sponge[i] = enc[8 * r + i];
            poseidon2(ch.sponge);
        }
    }
    void sponge_set(const u64 v[12]) { for (int i = 0; i < 12; ++i) ch.sponge[i] = v[i]; }
};

// initial_bootloader_state — src/main_vm/loading.rs:11-226 (the callstack sponge of the formal empty frame needs 4 permutations:
// computed only when the chains are wanted)
void bootloader_state(const zk_opcode_defs& d, const zk_vm_closed_form_input& in, vmn::State& s, Chains& ch, bool chains_on) {
    u64 z[vmn::STATE_WORDS] = {0};
    vmn::state_unflatten(s, [&](int w) { return z[w]; });
    vmn::Ctx& c = s.ctx;
    c.base_page = d.params[ZK_VMP_BOOTLOADER_BASE_PAGE]; c.code_page = d.params[ZK_VMP_BOOTLOADER_CODE_PAGE];
    c.eh = d.params[ZK_VMP_INITIAL_FRAME_FORMAL_EH_LOCATION]; c.ergs = d.params[ZK_VMP_VM_INITIAL_FRAME_ERGS];
    c.code_address[0] = c.this_[0] = d.params[ZK_VMP_BOOTLOADER_FORMAL_ADDRESS_LOW];
    for (int i = 0; i < 4; ++i) c.rq_tail[i] = c.rq_head[i] = in.rollback_queue_tail_for_block[i];
    c.is_kernel = 1;
    c.heap_bound = c.aux_heap_bound = d.params[ZK_VMP_BOOTLOADER_MAX_MEMORY];
    s.depth = 1;
    s.mem_len = in.memory_queue_initial_length; s.dec_len = in.decommitment_queue_initial_length;
    s.timestamp = d.params[ZK_VMP_STARTING_TIMESTAMP]; s.page_counter = d.params[ZK_VMP_STARTING_BASE_PAGE];
    s.regs[0].ptr = 1; s.regs[0].v.l[1] = d.params[ZK_VMP_BOOTLOADER_CALLDATA_PAGE];
    std::memset(&ch, 0, sizeof ch);
    std::memcpy(ch.mem, in.memory_queue_initial_tail, sizeof ch.mem);
    std::memcpy(ch.dec, in.decommitment_queue_initial_tail, sizeof ch.dec);
    if (chains_on) {
        vmn::Ctx empty;
        vmn::ctx_unflatten(empty, z);
        for (int i = 0; i < 4; ++i) empty.rq_tail[i] = empty.rq_head[i] = in.rollback_queue_tail_for_block[i];
        empty.is_kernel = 1;
        u64 enc[32];
        vmn::ctx_encode(enc, empty);
        for (int r = 0; r < 4; ++r) {
            for (int i = 0; i < 8; ++i) ch.sponge[i] = enc[8 * r + i];
            poseidon2(ch.sponge);
        }
    }
}

void write_state(const vmn::State& s, const Chains& ch, bool chains_on, u64* col, u64 stride) {
    vmn::state_flatten(s, [&](int w, u64 v) { col[(u64)w * stride] = v; });
    if (!chains_on) return;
    for (int i = 0; i < 4; ++i) col[(u64)(vmn::SW_FWD_TAIL + i) * stride] = ch.fwd[i];
    for (int i = 0; i < 12; ++i) {
        col[(u64)(vmn::SW_SPONGE + i) * stride] = ch.sponge[i];
        col[(u64)(vmn::SW_MEM_TAIL + i) * stride] = ch.mem[i];
        col[(u64)(vmn::SW_DEC_TAIL + i) * stride] = ch.dec[i];
    }
}

}  // namespace

extern "C" int zk_pack_main_vm_witness(zk_cs* h, const zk_vm_closed_form_input* in, const zk_vm_witness_oracle* oracle, uint32_t instance, uint32_t batch,
                                       uint64_t* outer_words, uint64_t* loop_words, uint32_t flags, zk_vm_pack_report* report) {
    if (flags & (ZK_VM_PACK_RECORD_STATES | ZK_VM_PACK_STATES_FROM_WITNESS)) { zkgl::set_last_error("zk_pack_main_vm_witness: the queue-state flags need zk_pack_main_vm_witness_states"); return (int)ZK_ERR_INVALID; }
    return zk_pack_main_vm_witness_states(h, in, oracle, nullptr, instance, batch, outer_words, loop_words, flags, report);
}

extern "C" int zk_pack_main_vm_witness_states(zk_cs* h, const zk_vm_closed_form_input* in, const zk_vm_witness_oracle* oracle, zk_vm_queue_states* states,
                                              uint32_t instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words, uint32_t flags, zk_vm_pack_report* report) {
    auto bad = [](const char* m) { zkgl::set_last_error(m); return (int)ZK_ERR_INVALID; };
    if (!h || !zkgl::cs_of(h) || !in || !oracle || !outer_words || !loop_words || !report) return bad("zk_pack_main_vm_witness: null argument");
    const bool rec = (flags & ZK_VM_PACK_RECORD_STATES) != 0, from = (flags & ZK_VM_PACK_STATES_FROM_WITNESS) != 0;
    if ((rec || from) && !states) return bad("zk_pack_main_vm_witness_states: the queue-state flags need `states`");
    if (rec && from) return bad("zk_pack_main_vm_witness_states: RECORD_STATES and STATES_FROM_WITNESS exclude each other");
    if (rec && !(flags & ZK_VM_PACK_FILL_STATE)) return bad("zk_pack_main_vm_witness_states: RECORD_STATES needs FILL_STATE");
    if (from) flags |= ZK_VM_PACK_FILL_STATE;   // reading the states means writing the whole VmLocalState
    if (states && (rec || from)) {
        if ((states->n_memory_tails && !states->memory_tails) || (states->n_decommit_tails && !states->decommit_tails) || (states->n_log_forward_tails && !states->log_forward_tails))
            return bad("zk_pack_main_vm_witness_states: null state array");
        states->used_memory_tails = states->used_decommit_tails = states->used_log_forward_tails = 0;
        states->host_permutations = 0;
    }
    g_host_permutations = 0;
    zkgl::CS& cs = *zkgl::cs_of(h);
    if (cs.native_seed_kind != 1 || cs.circuit_blob.size() != sizeof(zk_opcode_defs) || !cs.limit()) return bad("zk_pack_main_vm_witness: not a recorded main_vm circuit");
    if (instance >= batch) return bad("zk_pack_main_vm_witness: instance >= batch");
    zk_opcode_defs defs;
    std::memcpy(&defs, cs.circuit_blob.data(), sizeof defs);
    const uint32_t limit = cs.limit();
    const bool chains_on = (flags & ZK_VM_PACK_FILL_STATE) != 0;
    const bool oracle_only = (flags & ZK_VM_PACK_ORACLE_WORDS_ONLY) != 0;
    if (oracle_only && from) return bad("zk_pack_main_vm_witness: ORACLE_WORDS_ONLY excludes STATES_FROM_WITNESS (reading the states means writing the state rows, which this form does not ship)");
    if (oracle_only && chains_on) return bad("zk_pack_main_vm_witness: ORACLE_WORDS_ONLY leaves the state rows to the device seeder; it excludes FILL_STATE");
    if (oracle_only && cs.loop_input_words() <= (uint32_t)vmn::STATE_WORDS) return bad("zk_pack_main_vm_witness: ORACLE_WORDS_ONLY needs a loop layout with oracle rows behind the VmLocalState");
    if (oracle_only && cs.layout_word("loop", "state") != 0) return bad("zk_pack_main_vm_witness: the recorded layout does not start with the VmLocalState");
    std::memset(report, 0, sizeof *report);

    PackEnv env;
    env.o = oracle; env.rep = report; env.cs = &cs; env.chains_on = chains_on;
    if (rec || from) { env.qs = states; env.qs_mode = rec ? 1 : 2; }
    struct { const char* name; u32* dst; } fields[] = {
        {"code_word", &env.w_code_word}, {"src0_read_value", &env.w_src0_value}, {"src0_read_is_ptr", &env.w_src0_is_ptr},
        {"log_pubdata_refund", &env.w_refund}, {"log_storage_read_value", &env.w_log_read}, {"log_rollback_queue_prev_head", &env.w_log_prev_head},
        {"near_call_rollback_queue_tail", &env.w_near_tail}, {"far_call_code_hash_read_value", &env.w_far_code_hash},
        {"far_call_decommit_suggested_page", &env.w_far_page}, {"far_call_rollback_queue_tail", &env.w_far_tail}, {"ret_popped_context", &env.w_ret_ctx},
        {"ret_previous_callstack_state", &env.w_ret_state}, {"uma_read_a", &env.w_uma_a}, {"uma_read_b", &env.w_uma_b}};
    for (auto& f : fields) {
        *f.dst = cs.layout_word("loop", f.name);
        if (*f.dst == UINT32_MAX) return bad("zk_pack_main_vm_witness: the recorded layout lacks an oracle field");
        // ORACLE_WORDS_ONLY ships rows [STATE_WORDS, n) only: an oracle field recorded in front of them would be dropped silently
        if (oracle_only && *f.dst < (uint32_t)vmn::STATE_WORDS) return bad("zk_pack_main_vm_witness: ORACLE_WORDS_ONLY needs every oracle field behind the VmLocalState rows of the recorded layout");
    }
    // ---- outer stream: VmCircuitInputOutput::alloc_ignoring_outputs order (circuits/main_vm.cpp entry_point)
    const char* last_name = nullptr; uint32_t last_w = 0;   // (one layout lookup per field, not per word: hidden_fsm_input is 243 words)
    auto outer = [&](const char* name, uint32_t i, u64 v) {
        if (name != last_name) {
            last_w = cs.layout_word("outer", name);
            if (last_w == UINT32_MAX) throw zkgl::ZkError(ZK_ERR_INVALID, std::string("main_vm layout lacks ") + name);
            last_name = name;
        }
        outer_words[(u64)(last_w + i) * batch + instance] = v;
    };
    try {
        outer("start_flag", 0, in->start_flag ? 1 : 0);
        for (int i = 0; i < 4; ++i) outer("rollback_queue_tail_for_block", i, in->rollback_queue_tail_for_block[i]);
        for (int i = 0; i < 12; ++i) outer("memory_queue_initial_tail", i, in->memory_queue_initial_tail[i]);
        outer("memory_queue_initial_length", 0, in->memory_queue_initial_length);
        for (int i = 0; i < 12; ++i) outer("decommitment_queue_initial_tail", i, in->decommitment_queue_initial_tail[i]);
        outer("decommitment_queue_initial_length", 0, in->decommitment_queue_initial_length);
        outer("zkporter_is_available", 0, in->zkporter_is_available ? 1 : 0);
        for (int i = 0; i < 8; ++i) outer("default_aa_code_hash", i, in->default_aa_code_hash[i]);
        for (int i = 0; i < vmn::STATE_WORDS; ++i) outer("hidden_fsm_input", i, in->start_flag ? 0 : in->hidden_fsm_input[i]);
    } catch (const zkgl::ZkError& e) {
        return bad(e.what());
    }
    // ---- the cycles
    vmn::Defs D;
    vmn::defs_prepare(D, &defs, &defs);
    vmn::Gctx G;
    G.zkporter_is_available = in->zkporter_is_available ? 1 : 0;
    for (int i = 0; i < 8; ++i) G.default_aa_code_hash.l[i] = in->default_aa_code_hash[i];
    vmn::State st;
    if (in->start_flag) bootloader_state(defs, *in, st, env.ch, chains_on);
    else {
        vmn::state_unflatten(st, [&](int w) { return in->hidden_fsm_input[w]; });
        for (int i = 0; i < 4; ++i) env.ch.fwd[i] = in->hidden_fsm_input[vmn::SW_FWD_TAIL + i];
        for (int i = 0; i < 12; ++i) {
            env.ch.sponge[i] = in->hidden_fsm_input[vmn::SW_SPONGE + i];
            env.ch.mem[i] = in->hidden_fsm_input[vmn::SW_MEM_TAIL + i];
            env.ch.dec[i] = in->hidden_fsm_input[vmn::SW_DEC_TAIL + i];
        }
    }
    // The stream is word-major (loop_words[w][instance][cycle]): a cycle's words are `batch * limit` apart, so writing cycle by cycle is one
    // cache line and one page per word (round 4: 5.5 ms per instance, all of it store misses).  The walker writes into a TILE of TC cycles
    // (word w of cycle c at tile[w * TC + c]: 360 words x 64 cycles = 184 KB, cache-resident) and the tile leaves as one contiguous run of
    // TC words per stream word.
    // The tile leaves with non-temporal stores: the staging array is written once and read by the DMA engine, never by this core — plain
    // stores would first READ every line they fill.  ZK_VM_PACK_ORACLE_WORDS_ONLY: the 243 VmLocalState rows are not part of `loop_words` at
    // all (the device seeder writes every one of them for every cycle): the array starts at row 243 — 117 of 360 rows written, and
    // copied to the device in one piece (rows are contiguous in the word-major stream).
    const uint32_t n_loop_words = cs.loop_input_words();
    const uint32_t w_first = oracle_only ? (uint32_t)vmn::STATE_WORDS : 0u;
    constexpr uint32_t TC = 64;
    static thread_local std::vector<u64> tile;   // (184 KB: above malloc's mmap threshold — a fresh mapping per instance otherwise)
    tile.resize((size_t)n_loop_words * TC);
    env.stride = TC;
    u64* const dst0 = loop_words + (u64)instance * limit;
    const u64 dst_stride = (u64)batch * limit;
    for (uint32_t c0 = 0; c0 < limit; c0 += TC) {
        const uint32_t nc = std::min(TC, limit - c0);
        std::memset(tile.data() + (size_t)w_first * TC, 0, (tile.size() - (size_t)w_first * TC) * sizeof(u64));
        for (uint32_t c = 0; c < nc; ++c) {
            u64* const col = tile.data() + c;
            if (chains_on) write_state(st, env.ch, true, col, TC);
            env.col = col;
            vmn::vm_cycle(D, G, st, env);
        }
        for (uint32_t w = w_first; w < n_loop_words; ++w) stream_out(dst0 + (u64)(w - w_first) * dst_stride + c0, tile.data() + (size_t)w * TC, nc);
    }
    stream_fence();
    u64 fin[vmn::STATE_WORDS] = {0};
    write_state(st, env.ch, true, fin, 1);
    std::memcpy(report->final_state, fin, sizeof fin);
    if (states && (rec || from)) states->host_permutations = g_host_permutations;
    if (env.qs_overflow) { zkgl::set_last_error("zk_pack_main_vm_witness_states: a state array is too small for this chunk"); return (int)ZK_ERR_CAPACITY; }
    return ZK_OK;
}
