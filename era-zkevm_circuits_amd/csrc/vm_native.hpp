// vm_native.hpp — value-level walker of the main_vm cycle (host AND device code).
//
// The recorded circuit (circuits/main_vm.cpp) evaluates all eleven opcode families of /root/reference/src/main_vm every cycle and
// merges them by selects: right for a SIMT trace, wasteful for the one thing that is sequential — the VmLocalState a cycle hands
// to the next one (`state = vm_cycle(state, ..)`, src/main_vm/mod.rs:102-110).  This file computes that state the way a VM does:
// decode once, run the ONE family that applies, in 32-bit integer arithmetic.  Two users:
//   * the seeding kernels (kernels_vm_seed.hpp): phase A = this walker, one thread per circuit instance, writes the non-hash words
//     of every cycle's VmLocalState and the absorb events of the four Poseidon2 chains (memory queue, decommit queue, forward log
//     queue, callstack sponge); phase B runs the chains (no interpreter, no cone);
//   * the host packer zk_pack_main_vm_witness (vm_pack.cpp): the reference's WitnessOracle getters (src/main_vm/witness_oracle.rs:45-91)
//     answer only under `execute`, so placing an answer at its cycle needs this walk.
// Citations per block; the structure follows create_prestate (pre_state.rs:71-519), perform_initial_decoding
// (decoded_opcode.rs:42-220) and opcodes/*.rs.  Everything zkevm_opcode_defs supplies comes from the zk_opcode_defs blob.
// Values that the circuit range-checks are carried as u32; the walker never fails: on inputs no satisfiable trace has, it produces
// SOME state and the circuit's own links / gates report the trace.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/zkgl_vm.h"

#if defined(__HIP__)
#define VMN_HD __host__ __device__ inline
#else
#define VMN_HD inline
#endif

namespace vmn {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int NREG = ZK_VM_REGISTERS;
constexpr int STATE_WORDS = 243;
constexpr int CTX_WORDS = 42;
// word offsets inside the flattened VmLocalState (src/base_structures/vm_state/mod.rs:92-109, declaration order)
enum : int {
    SW_PREV_CODE_WORD = 0, SW_REGS = 8, SW_FLAGS = 143, SW_TIMESTAMP = 146, SW_PAGE_COUNTER = 147, SW_TX_NUMBER = 148, SW_PREV_CODE_PAGE = 149,
    SW_PREV_SUPER_PC = 150, SW_PENDING = 151, SW_ERGS_PER_PUBDATA = 152, SW_CTX = 153, SW_FWD_TAIL = 195, SW_FWD_LEN = 199, SW_DEPTH = 200,
    SW_SPONGE = 201, SW_MEM_TAIL = 213, SW_MEM_LEN = 225, SW_DEC_TAIL = 226, SW_DEC_LEN = 238, SW_CTX_U128 = 239
};
// true for the words phase B (the Poseidon2 chains) owns
VMN_HD bool is_chain_word(int w) {
    return (w >= SW_FWD_TAIL && w < SW_FWD_TAIL + 4) || (w >= SW_SPONGE && w < SW_SPONGE + 12) || (w >= SW_MEM_TAIL && w < SW_MEM_TAIL + 12) ||
           (w >= SW_DEC_TAIL && w < SW_DEC_TAIL + 12);
}

struct U256 { u32 l[8]; };
struct Reg { u32 ptr; U256 v; };
// ExecutionContextRecord — src/base_structures/vm_state/saved_context.rs:37-68
struct Ctx {
    u32 this_[5], caller[5], code_address[5];
    u32 code_page, base_page, heap_bound, aux_heap_bound;
    u64 rq_head[4], rq_tail[4];
    u32 rq_len, pc, sp, eh, ergs, is_static, is_kernel, this_shard, caller_shard, code_shard;
    u32 ctx_u128[4];
    u32 is_local;
};
// VmLocalState without the four hash-chain states
struct State {
    U256 prev_code_word;
    Reg regs[NREG];
    u32 of, eq, gt;
    u32 timestamp, page_counter, tx_number, prev_code_page, prev_super_pc, pending_exception, ergs_per_pubdata;
    Ctx ctx;
    u32 fwd_len, depth, mem_len, dec_len;
    u32 ctx_u128[4];
    u32 last_family;  // bookkeeping (profiling / tests): the family the last cycle applied; not a state word
};

// the blob as the walker wants it: the two 2048-row tables by pointer (host or device memory, matching the caller), everything
// else BY VALUE — as a kernel argument these become scalar loads of the kernarg segment instead of dependent global loads
struct DefsSmall {
    u32 type_bits, variant_bits, flag_bits, src_mode_bits, dst_mode_bits, description_bits_flattened, aux_bits;
    u32 aux_kernel_mode, aux_static_ok, aux_explicit_panic;
    u32 variant_idx[ZK_VMV__COUNT], flag_idx[ZK_VMFL__COUNT], condition_idx[ZK_VMC__COUNT], can_write_dst0_into_memory[ZK_VMF__COUNT];
    u64 nop_encoding, panic_encoding, nop_bitspread, panic_bitspread;
    u32 params[ZK_VMP__COUNT];
};
struct Defs {
    DefsSmall s;
    const u64* props;
    const u32* prices;
    u32 variant_bit0, flag_bit0, src_bit0, dst_bit0, aux_bit0;
    u64 props_mask;
};
// GlobalContext (src/base_structures/vm_state/mod.rs: per_block_context), per instance
struct Gctx { u32 zkporter_is_available; U256 default_aa_code_hash; };

// host_view: the blob in host memory; tables: where props / prices live for the code that will walk (the same blob, or its device copy)
inline void defs_prepare(Defs& D, const zk_opcode_defs* host_view, const zk_opcode_defs* tables) {
    const zk_opcode_defs& d = *host_view;
    DefsSmall& t = D.s;
    t.type_bits = d.type_bits; t.variant_bits = d.variant_bits; t.flag_bits = d.flag_bits; t.src_mode_bits = d.src_mode_bits;
    t.dst_mode_bits = d.dst_mode_bits; t.description_bits_flattened = d.description_bits_flattened; t.aux_bits = d.aux_bits;
    t.aux_kernel_mode = d.aux_kernel_mode; t.aux_static_ok = d.aux_static_ok; t.aux_explicit_panic = d.aux_explicit_panic;
    for (int i = 0; i < ZK_VMV__COUNT; ++i) t.variant_idx[i] = d.variant_idx[i];
    for (int i = 0; i < ZK_VMFL__COUNT; ++i) t.flag_idx[i] = d.flag_idx[i];
    for (int i = 0; i < ZK_VMC__COUNT; ++i) t.condition_idx[i] = d.condition_idx[i];
    for (int i = 0; i < ZK_VMF__COUNT; ++i) t.can_write_dst0_into_memory[i] = d.can_write_dst0_into_memory[i];
    t.nop_encoding = d.nop_encoding; t.panic_encoding = d.panic_encoding; t.nop_bitspread = d.nop_bitspread; t.panic_bitspread = d.panic_bitspread;
    for (int i = 0; i < ZK_VMP__COUNT; ++i) t.params[i] = d.params[i];
    D.props = tables->props; D.prices = tables->prices;
    D.variant_bit0 = d.type_bits;
    D.flag_bit0 = D.variant_bit0 + d.variant_bits;
    D.src_bit0 = D.flag_bit0 + d.flag_bits;
    D.dst_bit0 = D.src_bit0 + d.src_mode_bits;
    D.aux_bit0 = d.description_bits_flattened;
    D.props_mask = (1ull << d.description_bits_flattened) - 1;
}

// ------------------------------------------------------------------------------------------------ flatten / unflatten
VMN_HD void ctx_flatten(const Ctx& c, u64* o) {  // flatten_as_variables, saved_context.rs:279-323
    int n = 0;
    for (int i = 0; i < 5; ++i) o[n++] = c.this_[i];
    for (int i = 0; i < 5; ++i) o[n++] = c.caller[i];
    for (int i = 0; i < 5; ++i) o[n++] = c.code_address[i];
    o[n++] = c.code_page; o[n++] = c.base_page; o[n++] = c.heap_bound; o[n++] = c.aux_heap_bound;
    for (int i = 0; i < 4; ++i) o[n++] = c.rq_head[i];
    for (int i = 0; i < 4; ++i) o[n++] = c.rq_tail[i];
    o[n++] = c.rq_len; o[n++] = c.pc; o[n++] = c.sp; o[n++] = c.eh; o[n++] = c.ergs; o[n++] = c.is_static; o[n++] = c.is_kernel;
    o[n++] = c.this_shard; o[n++] = c.caller_shard; o[n++] = c.code_shard;
    for (int i = 0; i < 4; ++i) o[n++] = c.ctx_u128[i];
    o[n++] = c.is_local;
}
VMN_HD void ctx_unflatten(Ctx& c, const u64* f) {
    int n = 0;
    for (int i = 0; i < 5; ++i) c.this_[i] = (u32)f[n++];
    for (int i = 0; i < 5; ++i) c.caller[i] = (u32)f[n++];
    for (int i = 0; i < 5; ++i) c.code_address[i] = (u32)f[n++];
    c.code_page = (u32)f[n++]; c.base_page = (u32)f[n++]; c.heap_bound = (u32)f[n++]; c.aux_heap_bound = (u32)f[n++];
    for (int i = 0; i < 4; ++i) c.rq_head[i] = f[n++];
    for (int i = 0; i < 4; ++i) c.rq_tail[i] = f[n++];
    c.rq_len = (u32)f[n++]; c.pc = (u32)f[n++]; c.sp = (u32)f[n++]; c.eh = (u32)f[n++]; c.ergs = (u32)f[n++];
    c.is_static = (u32)f[n++]; c.is_kernel = (u32)f[n++]; c.this_shard = (u32)f[n++]; c.caller_shard = (u32)f[n++]; c.code_shard = (u32)f[n++];
    for (int i = 0; i < 4; ++i) c.ctx_u128[i] = (u32)f[n++];
    c.is_local = (u32)f[n++];
}
static_assert(offsetof(State, regs) == 4 * SW_REGS && offsetof(State, of) == 4 * SW_FLAGS && offsetof(State, timestamp) == 4 * SW_TIMESTAMP &&
              offsetof(State, ergs_per_pubdata) == 4 * SW_ERGS_PER_PUBDATA && sizeof(Reg) == 36, "State: the words before the context are u32 in flatten order");
static_assert(offsetof(Ctx, aux_heap_bound) == 4 * 18 && offsetof(Ctx, pc) == offsetof(Ctx, rq_len) + 4 && offsetof(Ctx, is_local) == offsetof(Ctx, rq_len) + 4 * 14,
              "Ctx: u32 runs in flatten order around the two u64 queue states");
// word w of the flattened VmLocalState read straight from the struct (chain words: 0) — the struct is the walker's working state, so
// nothing has to be flattened per cycle; every lane of a wavefront can fetch some of the words
VMN_HD u64 state_word(const State& s, int w) {
    const char* const base = (const char*)&s;
    if (w < SW_CTX) return *(const u32*)(base + 4 * w);     // prev_code_word, regs, flags, scalars: u32 words in declaration order
    if (w < SW_CTX + 19) return *(const u32*)((const char*)&s.ctx + 4 * (w - SW_CTX));
    if (w < SW_CTX + 23) return s.ctx.rq_head[w - (SW_CTX + 19)];
    if (w < SW_CTX + 27) return s.ctx.rq_tail[w - (SW_CTX + 23)];
    if (w < SW_CTX + CTX_WORDS) return *(const u32*)((const char*)&s.ctx.rq_len + 4 * (w - (SW_CTX + 27)));
    if (w == SW_FWD_LEN) return s.fwd_len;
    if (w == SW_DEPTH) return s.depth;
    if (w == SW_MEM_LEN) return s.mem_len;
    if (w == SW_DEC_LEN) return s.dec_len;
    if (w >= SW_CTX_U128 && w < SW_CTX_U128 + 4) return s.ctx_u128[w - SW_CTX_U128];
    return 0;
}
// every non-chain word of the state through `put(word index, value)`
template <class Put>
VMN_HD void state_flatten(const State& s, Put&& put) {
    for (int i = 0; i < 8; ++i) put(SW_PREV_CODE_WORD + i, (u64)s.prev_code_word.l[i]);
    for (int r = 0; r < NREG; ++r) {
        put(SW_REGS + 9 * r, (u64)s.regs[r].ptr);
        for (int i = 0; i < 8; ++i) put(SW_REGS + 9 * r + 1 + i, (u64)s.regs[r].v.l[i]);
    }
    put(SW_FLAGS, (u64)s.of); put(SW_FLAGS + 1, (u64)s.eq); put(SW_FLAGS + 2, (u64)s.gt);
    put(SW_TIMESTAMP, (u64)s.timestamp); put(SW_PAGE_COUNTER, (u64)s.page_counter); put(SW_TX_NUMBER, (u64)s.tx_number);
    put(SW_PREV_CODE_PAGE, (u64)s.prev_code_page); put(SW_PREV_SUPER_PC, (u64)s.prev_super_pc); put(SW_PENDING, (u64)s.pending_exception);
    put(SW_ERGS_PER_PUBDATA, (u64)s.ergs_per_pubdata);
    u64 cf[CTX_WORDS];
    ctx_flatten(s.ctx, cf);
    for (int i = 0; i < CTX_WORDS; ++i) put(SW_CTX + i, cf[i]);
    put(SW_FWD_LEN, (u64)s.fwd_len); put(SW_DEPTH, (u64)s.depth); put(SW_MEM_LEN, (u64)s.mem_len); put(SW_DEC_LEN, (u64)s.dec_len);
    for (int i = 0; i < 4; ++i) put(SW_CTX_U128 + i, (u64)s.ctx_u128[i]);
}
template <class Get>
VMN_HD void state_unflatten(State& s, Get&& get) {
    for (int i = 0; i < 8; ++i) s.prev_code_word.l[i] = (u32)get(SW_PREV_CODE_WORD + i);
    for (int r = 0; r < NREG; ++r) {
        s.regs[r].ptr = (u32)get(SW_REGS + 9 * r);
        for (int i = 0; i < 8; ++i) s.regs[r].v.l[i] = (u32)get(SW_REGS + 9 * r + 1 + i);
    }
    s.of = (u32)get(SW_FLAGS); s.eq = (u32)get(SW_FLAGS + 1); s.gt = (u32)get(SW_FLAGS + 2);
    s.timestamp = (u32)get(SW_TIMESTAMP); s.page_counter = (u32)get(SW_PAGE_COUNTER); s.tx_number = (u32)get(SW_TX_NUMBER);
    s.prev_code_page = (u32)get(SW_PREV_CODE_PAGE); s.prev_super_pc = (u32)get(SW_PREV_SUPER_PC); s.pending_exception = (u32)get(SW_PENDING);
    s.ergs_per_pubdata = (u32)get(SW_ERGS_PER_PUBDATA);
    u64 cf[CTX_WORDS];
    for (int i = 0; i < CTX_WORDS; ++i) cf[i] = get(SW_CTX + i);
    ctx_unflatten(s.ctx, cf);
    s.fwd_len = (u32)get(SW_FWD_LEN); s.depth = (u32)get(SW_DEPTH); s.mem_len = (u32)get(SW_MEM_LEN); s.dec_len = (u32)get(SW_DEC_LEN);
    for (int i = 0; i < 4; ++i) s.ctx_u128[i] = (u32)get(SW_CTX_U128 + i);
}

// ------------------------------------------------------------------------------------------------ encodings
// MemoryQuery::encode — src/base_structures/memory_query/mod.rs:103-221
VMN_HD void memory_query_encode(u64 enc[8], u32 ts, u32 page, u32 index, u32 rw, u32 is_ptr, const U256& v) {
    u32 b[12];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 4; ++k) b[4 * i + k] = (v.l[5 + i] >> (8 * k)) & 0xff;
    enc[0] = ts; enc[1] = page;
    enc[2] = (u64)index + ((u64)rw << 32) + ((u64)is_ptr << 33);
    for (int i = 0; i < 4; ++i) enc[3 + i] = (u64)v.l[i] + ((u64)b[3 * i] << 32) + ((u64)b[3 * i + 1] << 40) + ((u64)b[3 * i + 2] << 48);
    enc[7] = v.l[4];
}
// DecommitQuery::encode — src/base_structures/decommit_query/mod.rs:33-113
VMN_HD void decommit_query_encode(u64 enc[8], const U256& h, u32 page, u32 is_first, u32 ts) {
    u32 p[4], t[4];
    for (int k = 0; k < 4; ++k) { p[k] = (page >> (8 * k)) & 0xff; t[k] = (ts >> (8 * k)) & 0xff; }
    enc[0] = (u64)h.l[0] + ((u64)p[0] << 32) + ((u64)p[1] << 40) + ((u64)p[2] << 48);
    enc[1] = (u64)h.l[1] + ((u64)p[3] << 32) + ((u64)t[0] << 40) + ((u64)t[1] << 48);
    enc[2] = (u64)h.l[2] + ((u64)t[2] << 32) + ((u64)t[3] << 40) + ((u64)is_first << 48);
    for (int i = 3; i < 8; ++i) enc[i] = h.l[i];
}
// LogQuery::encode — src/base_structures/log_query/mod.rs:121-517
struct LogQ { u32 address[5]; U256 key, read_value, written_value; u32 aux_byte, rw_flag, rollback, is_service, shard_id, tx_number, timestamp; };
VMN_HD void log_query_encode(u64 enc[20], const LogQ& q) {
    u32 bs[52];
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 4; ++k) bs[4 * i + k] = (q.key.l[i] >> (8 * k)) & 0xff;
    for (int i = 0; i < 5; ++i)
        for (int k = 0; k < 4; ++k) bs[32 + 4 * i + k] = (q.address[i] >> (8 * k)) & 0xff;
    for (int i = 0; i < 17; ++i) {
        const u32 base = i < 8 ? q.read_value.l[i] : i < 16 ? q.written_value.l[i - 8] : q.timestamp;
        enc[i] = (u64)base + ((u64)bs[3 * i] << 32) + ((u64)bs[3 * i + 1] << 40) + ((u64)bs[3 * i + 2] << 48);
    }
    enc[17] = (u64)q.tx_number + ((u64)bs[51] << 32) + ((u64)q.aux_byte << 40) + ((u64)q.shard_id << 48);
    enc[18] = (u64)q.rw_flag + 2 * (u64)q.is_service;
    enc[19] = q.rollback;
}
// ExecutionContextRecord::encode — saved_context.rs:111-266
VMN_HD void ctx_encode(u64 v[32], const Ctx& c) {
    for (int i = 0; i < 4; ++i) { v[i] = c.rq_head[i]; v[4 + i] = c.rq_tail[i]; }
    for (int i = 0; i < 5; ++i) { v[8 + i] = c.code_address[i]; v[13 + i] = c.this_[i]; v[18 + i] = c.caller[i]; }
    for (int i = 0; i < 4; ++i) v[23 + i] = c.ctx_u128[i];
    u32 d[4];
    for (int k = 0; k < 4; ++k) d[k] = (c.rq_len >> (8 * k)) & 0xff;
    v[27] = (u64)c.code_page + ((u64)c.pc << 32) + ((u64)c.this_shard << 48) + ((u64)c.is_static << 56);
    v[28] = (u64)c.base_page + ((u64)c.sp << 32) + ((u64)c.caller_shard << 48) + ((u64)c.is_kernel << 56);
    v[29] = (u64)c.ergs + ((u64)c.eh << 32) + ((u64)c.code_shard << 48) + ((u64)c.is_local << 56);
    v[30] = (u64)c.heap_bound + ((u64)d[0] << 32) + ((u64)d[1] << 40);
    v[31] = (u64)c.aux_heap_bound + ((u64)d[2] << 32) + ((u64)d[3] << 40);
}

// ------------------------------------------------------------------------------------------------ U256 helpers
VMN_HD bool u256_is_zero(const U256& a) { u32 o = 0; for (int i = 0; i < 8; ++i) o |= a.l[i]; return o == 0; }
VMN_HD U256 u256_zero() { U256 z; for (int i = 0; i < 8; ++i) z.l[i] = 0; return z; }
VMN_HD u32 u256_add(U256& r, const U256& a, const U256& b) {
    u64 c = 0;
    for (int i = 0; i < 8; ++i) { c += (u64)a.l[i] + b.l[i]; r.l[i] = (u32)c; c >>= 32; }
    return (u32)c;
}
VMN_HD u32 u256_sub(U256& r, const U256& a, const U256& b) {
    u64 br = 0;
    for (int i = 0; i < 8; ++i) { const u64 d = (u64)a.l[i] - b.l[i] - br; r.l[i] = (u32)d; br = (d >> 32) & 1; }
    return (u32)br;
}
VMN_HD void u256_mul_wide(u32 out[16], const U256& a, const U256& b) {
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (int i = 0; i < 8; ++i) {
        u64 carry = 0;
        for (int j = 0; j < 8; ++j) {
            const u64 t = (u64)a.l[i] * b.l[j] + out[i + j] + carry;
            out[i + j] = (u32)t;
            carry = t >> 32;
        }
        out[i + 8] = (u32)carry;
    }
}
// q, r = divmod(a, b), b != 0 — Knuth algorithm D in base 2^32
// (operands by value: the limbs are indexed dynamically, and only this function's copies may live in addressable memory)
VMN_HD void u256_divrem(U256& q, U256& r, const U256 a, const U256 b) {
    q = u256_zero(); r = u256_zero();
    int n = 8;
    while (n > 0 && b.l[n - 1] == 0) --n;
    if (n == 0) { r = a; return; }
    int m = 8;
    while (m > 0 && a.l[m - 1] == 0) --m;
    if (m < n) { r = a; return; }
    if (n == 1) {
        u64 rem = 0;
        for (int i = m - 1; i >= 0; --i) { const u64 cur = (rem << 32) | a.l[i]; q.l[i] = (u32)(cur / b.l[0]); rem = cur % b.l[0]; }
        r.l[0] = (u32)rem;
        return;
    }
    int s = 0;
    { u32 top = b.l[n - 1]; while (!(top & 0x80000000u)) { top <<= 1; ++s; } }
    u32 vn[8], un[9];
    for (int i = n - 1; i > 0; --i) vn[i] = s ? (b.l[i] << s) | (b.l[i - 1] >> (32 - s)) : b.l[i];
    vn[0] = b.l[0] << s;
    un[m] = s ? a.l[m - 1] >> (32 - s) : 0;
    for (int i = m - 1; i > 0; --i) un[i] = s ? (a.l[i] << s) | (a.l[i - 1] >> (32 - s)) : a.l[i];
    un[0] = a.l[0] << s;
    for (int j = m - n; j >= 0; --j) {
        const u64 num = ((u64)un[j + n] << 32) | un[j + n - 1];
        u64 qhat = num / vn[n - 1], rhat = num % vn[n - 1];
        while (qhat >= (1ull << 32) || qhat * vn[n - 2] > ((rhat << 32) | un[j + n - 2])) {
            --qhat; rhat += vn[n - 1];
            if (rhat >= (1ull << 32)) break;
        }
        int64_t borrow = 0;
        for (int i = 0; i < n; ++i) {
            const u64 p = qhat * vn[i];
            const int64_t t = (int64_t)un[i + j] - borrow - (int64_t)(p & 0xffffffffull);
            un[i + j] = (u32)t;
            borrow = (int64_t)(p >> 32) - (t >> 32);
        }
        const int64_t t = (int64_t)un[j + n] - borrow;
        un[j + n] = (u32)t;
        if (t < 0) {
            --qhat;
            u64 c = 0;
            for (int i = 0; i < n; ++i) { c += (u64)un[i + j] + vn[i]; un[i + j] = (u32)c; c >>= 32; }
            un[j + n] += (u32)c;
        }
        q.l[j] = (u32)qhat;
    }
    for (int i = 0; i < n - 1; ++i) r.l[i] = s ? (un[i] >> s) | (un[i + 1] << (32 - s)) : un[i];
    r.l[n - 1] = un[n - 1] >> s;
}
// No dynamic indexing into U256 values anywhere on the common path: an indexed local lives in scratch memory on the GPU and drags
// every use of the operand (src0 / src1 / dst0: all families) with it.  Picks are select chains, shifts are barrel shifters.
VMN_HD u32 u256_byte(const U256& a, u32 byte_idx) {
    const u32 li = (byte_idx >> 2) & 7;
    u32 limb = 0;
    for (u32 k = 0; k < 8; ++k) limb = li == k ? a.l[k] : limb;
    return (limb >> (8 * (byte_idx & 3))) & 0xff;
}
// x <<= sh / x >>= sh over N little-endian u32 limbs, sh < 32 N
template <int N>
VMN_HD void limbs_shl(u32* x, u32 sh) {
    const u32 ws = sh >> 5, b = sh & 31;
    for (int k = 1; k < N; k <<= 1) {
        const bool on = (ws & (u32)k) != 0;
        for (int i = N - 1; i >= 0; --i) { const u32 src = i - k >= 0 ? x[i - k] : 0; x[i] = on ? src : x[i]; }
    }
    for (int i = N - 1; i >= 0; --i) { const u32 below = i ? x[i - 1] : 0; x[i] = b ? (x[i] << b) | (below >> (32 - b)) : x[i]; }
}
template <int N>
VMN_HD void limbs_shr(u32* x, u32 sh) {
    const u32 ws = sh >> 5, b = sh & 31;
    for (int k = 1; k < N; k <<= 1) {
        const bool on = (ws & (u32)k) != 0;
        for (int i = 0; i < N; ++i) { const u32 src = i + k < N ? x[i + k] : 0; x[i] = on ? src : x[i]; }
    }
    for (int i = 0; i < N; ++i) { const u32 above = i + 1 < N ? x[i + 1] : 0; x[i] = b ? (x[i] >> b) | (above << (32 - b)) : x[i]; }
}

// c ? a : b on aggregates, member by member (a C++ `?:` on struct lvalues is a pointer select + copy: both sides become addressable memory)
VMN_HD U256 sel256(bool c, const U256& a, const U256& b) { U256 r; for (int i = 0; i < 8; ++i) r.l[i] = c ? a.l[i] : b.l[i]; return r; }
VMN_HD Reg sel_reg(bool c, const Reg& a, const Reg& b) { Reg r; r.ptr = c ? a.ptr : b.ptr; r.v = sel256(c, a.v, b.v); return r; }
struct FatPtr { u32 offset, page, start, length; };
VMN_HD FatPtr sel_fp(bool c, const FatPtr& a, const FatPtr& b) { return FatPtr{c ? a.offset : b.offset, c ? a.page : b.page, c ? a.start : b.start, c ? a.length : b.length}; }
VMN_HD FatPtr fp_readjust(const FatPtr& p) { return FatPtr{0, p.page, p.start + p.offset, p.length - p.offset}; }

// ------------------------------------------------------------------------------------------------ the cycle
// Env supplies the WitnessOracle answers and takes the hash-chain events:
//   void opcode_row(const Defs&, u32 variant, u32& price, u64& props)   the decode table row (where the caller keeps the table)
//   void code_word(bool exec, U256&)                       get_memory_witness_for_read (utils.rs:170)
//   void src0(bool exec, U256&, u32& is_ptr)               get_memory_witness_for_read (utils.rs:434)
//   u32  refund(bool exec)                                 get_refunds (log.rs:235)
//   void log_read(bool exec, U256&)                        get_storage_read_witness (log.rs:305)
//   void log_prev_head(bool exec, u64[4])                  get_rollback_queue_witness (log.rs:356)
//   void near_call_tail(bool exec, u64[4])                 get_rollback_queue_tail_witness_for_call (near_call.rs:86)
//   void far_code_hash(bool exec, U256&)                   get_storage_read_witness (far_call.rs:1217)
//   u32  far_decommit_page(bool exec)                      get_decommittment_request_suggested_page (far_call.rs:1518)
//   void far_call_tail(bool exec, u64[4])                  get_rollback_queue_tail_witness_for_call (far_call.rs:835)
//   void ret_pop(bool exec, u64 ctx42[42], u64 state[12])  get_callstack_witness (ret.rs:152)
//   void uma_read(int which, bool exec, U256&)             get_memory_witness_for_read (uma.rs:308,341)
//   void mem_push(const u64[8]) / dec_push(const u64[8]) / fwd_push(const u64[20]) / fwd_set(const u64[4]) /
//   sponge_push(const u64[32]) / sponge_set(const u64[12])    the chains' events, in order
template <class Env>
VMN_HD void vm_cycle(const Defs& D, const Gctx& G, State& st, Env& env) {
    const DefsSmall& d = D.s;
    Ctx& c = st.ctx;
    // ---------------- create_prestate (pre_state.rs:88-221)
    const bool skip = st.depth == 0;
    const bool pending = st.pending_exception != 0;
    const bool should_try_read = !skip && !pending;
    st.pending_exception = 0;
    const u32 pc = c.pc & 0xffff;
    const u32 pc_plus_one = (pc + 1) & 0xffff;
    const u32 super_pc = pc >> 2, sub_pc = pc & 3;
    const bool should_read_opcode = should_try_read && !(st.prev_code_page == c.code_page && super_pc == st.prev_super_pc);
    const u32 ts0 = st.timestamp, ts1 = ts0 + 1, ts3 = ts0 + 3;
    U256 code_word;
    env.code_word(should_read_opcode, code_word);
    if (should_read_opcode) {
        u64 enc[8];
        memory_query_encode(enc, ts0, c.code_page, super_pc, 0, 0, code_word);
        env.mem_push(enc);
        st.mem_len += 1;
    } else {
        code_word = st.prev_code_word;
    }
    u64 opcode = 0;  // four opcodes per word, the first in the most significant 8 bytes (pre_state.rs:184-206)
    for (u32 k = 0; k < 4; ++k) opcode = sub_pc == k ? (((u64)code_word.l[2 * (3 - k) + 1] << 32) | code_word.l[2 * (3 - k)]) : opcode;
    if (skip) opcode = d.nop_encoding;
    if (pending) opcode = d.panic_encoding;
    st.prev_code_word = code_word;
    st.prev_code_page = c.code_page;
    if (!skip) { c.pc = pc_plus_one; st.prev_super_pc = super_pc; st.timestamp = ts0 + 4; }
    const bool is_kernel = c.is_kernel != 0, is_static = c.is_static != 0;
    const bool callstack_is_full = st.depth == d.params[ZK_VMP_VM_MAX_STACK_DEPTH];
    // ---------------- perform_initial_decoding (decoded_opcode.rs:42-220)
    const u32 variant = (u32)(opcode & 0x7ff), cond_key = (u32)((opcode >> 13) & 7);
    u32 src_byte = (u32)((opcode >> 16) & 0xff), dst_byte = (u32)((opcode >> 24) & 0xff);
    const u32 imm0 = (u32)((opcode >> 32) & 0xffff), imm1 = (u32)((opcode >> 48) & 0xffff);
    u32 price;
    u64 props_full;
    env.opcode_row(D, variant, price, props_full);  // OPCODES_PRICES / OPCODES_PROPS_INTEGER_BITMASKS row (src/tables/opcodes_decoding.rs:14-38)
    bool condition = false;
    u32 cond_kind = 0xff;  // constant indices only: a dynamically indexed kernel-argument array would push the whole argument into scratch
    for (int k = 0; k < ZK_VMC__COUNT; ++k)
        if (d.condition_idx[k] == cond_key) cond_kind = (u32)k;
    switch (cond_kind) {  // src/tables/conditional.rs:35-44
    case ZK_VMC_ALWAYS: condition = true; break;
    case ZK_VMC_LT: condition = st.of; break;
    case ZK_VMC_EQ: condition = st.eq; break;
    case ZK_VMC_GT: condition = st.gt; break;
    case ZK_VMC_GE: condition = st.gt || st.eq; break;
    case ZK_VMC_LE: condition = st.of || st.eq; break;
    case ZK_VMC_NE: condition = !st.eq; break;
    case ZK_VMC_GT_OR_LT: condition = st.gt || st.of; break;
    default: break;
    }
    const u32 aux = (u32)(props_full >> D.aux_bit0);
    const bool requires_kernel = (aux >> d.aux_kernel_mode) & 1, can_static = (aux >> d.aux_static_ok) & 1, explicit_panic = (aux >> d.aux_explicit_panic) & 1;
    const u32 cost = skip ? 0 : price;
    const bool out_of_ergs = c.ergs < cost;
    const u32 ergs_left = out_of_ergs ? 0 : c.ergs - cost;
    const bool mask_into_panic = explicit_panic || out_of_ergs || (requires_kernel && !is_kernel) || (is_static && !can_static) || callstack_is_full;
    const bool mask_into_nop = !mask_into_panic && !condition;
    u64 props = props_full & D.props_mask;
    if (mask_into_panic) props = d.panic_bitspread & D.props_mask;
    else if (mask_into_nop) props = d.nop_bitspread & D.props_mask;
    if (mask_into_panic || mask_into_nop) { src_byte = 0; dst_byte = 0; }
    u32 fam = ZK_VMF_NOP;
    {
        const u32 type_bits = (u32)(props & ((1u << d.type_bits) - 1));
        for (u32 i = 0; i < d.type_bits; ++i)
            if ((type_bits >> i) & 1) { fam = i; break; }
        if (type_bits == 0 || fam == ZK_VMF_INVALID) fam = ZK_VMF_NOP;  // no satisfiable trace decodes to this
    }
    auto var = [&](int which) -> bool { return (props >> (D.variant_bit0 + d.variant_idx[which])) & 1; };
    auto flag = [&](int which) -> bool { return (props >> (D.flag_bit0 + d.flag_idx[which])) & 1; };
    auto src_mode = [&](int m) -> bool { return (props >> (D.src_bit0 + m)) & 1; };
    auto dst_mode = [&](int m) -> bool { return (props >> (D.dst_bit0 + m)) & 1; };
    const u32 src0_idx = src_byte & 15, src1_idx = src_byte >> 4, dst0_idx = dst_byte & 15, dst1_idx = dst_byte >> 4;
    c.ergs = ergs_left;
    const u32 preliminary_ergs_left = ergs_left;
    Reg zero_reg;
    zero_reg.ptr = 0; zero_reg.v = u256_zero();
    // (no `cond ? st.regs[i] : zero_reg`: a select between an LDS object and a local one makes both addressable)
    Reg draft_src0 = zero_reg, src1_register = zero_reg;
    if (src0_idx) draft_src0 = st.regs[src0_idx - 1];
    if (src1_idx) src1_register = st.regs[src1_idx - 1];
    const u32 src0_reg_lowest = draft_src0.v.l[0] & 0xffff;
    const u32 dst0_reg_lowest = (dst0_idx ? st.regs[dst0_idx - 1].v.l[0] : 0) & 0xffff;
    const u32 stack_page = c.base_page + 1, heap_page = c.base_page + 2, aux_heap_page = c.base_page + 3;
    const bool is_nop = fam == ZK_VMF_NOP;
    // resolve_memory_region_and_index_for_source / _for_dest (utils.rs:237-384)
    const bool use_code = src_mode(ZK_VMM_CODE_PAGE), use_abs = src_mode(ZK_VMM_ABSOLUTE_STACK), use_rel = src_mode(ZK_VMM_STACK_OFFSET),
               use_pp = src_mode(ZK_VMM_STACK_PUSH_POP);
    const u32 idx_abs = (src0_reg_lowest + imm0) & 0xffff;
    const u32 idx_rel = (c.sp - idx_abs) & 0xffff;
    const bool use_stack = use_abs || use_rel || use_pp;
    const bool should_read_src0 = (use_stack || use_code) && !is_nop;
    const u32 src0_page = use_stack ? stack_page : c.code_page;
    const u32 src0_index = (use_code || use_abs) ? idx_abs : idx_rel;
    const u32 sp_after_src0 = use_pp ? idx_rel : c.sp;
    const bool d_abs = dst_mode(ZK_VMM_ABSOLUTE_STACK), d_rel = dst_mode(ZK_VMM_STACK_OFFSET), d_pp = dst_mode(ZK_VMM_STACK_PUSH_POP);
    const u32 didx_abs = (dst0_reg_lowest + imm1) & 0xffff;
    const u32 didx_rel_push = (sp_after_src0 + didx_abs) & 0xffff, didx_rel = (sp_after_src0 - didx_abs) & 0xffff;
    const bool dst0_in_memory = (d_abs || d_rel || d_pp) && !is_nop;
    const u32 dst0_page = stack_page;
    const u32 dst0_index = d_abs ? didx_abs : (d_pp ? sp_after_src0 : didx_rel);
    c.sp = d_pp ? didx_rel_push : sp_after_src0;
    // may_be_read_memory_for_source_operand (utils.rs:388-522)
    Reg src0_from_mem;
    env.src0(should_read_src0, src0_from_mem.v, src0_from_mem.ptr);
    if (should_read_src0) {
        u64 enc[8];
        memory_query_encode(enc, ts0, src0_page, src0_index, 0, src0_from_mem.ptr, src0_from_mem.v);
        env.mem_push(enc);
        st.mem_len += 1;
    }
    Reg src0 = sel_reg(src_mode(ZK_VMM_REG_ONLY), draft_src0, src0_from_mem);
    if (src_mode(ZK_VMM_IMM16)) { src0 = zero_reg; src0.v.l[0] = imm0; }
    Reg src1 = src1_register;
    const bool swap = ((fam == ZK_VMF_SUB || fam == ZK_VMF_DIV || fam == ZK_VMF_SHIFT) && flag(ZK_VMFL_SWAP_ARITH)) || (fam == ZK_VMF_PTR && flag(ZK_VMFL_SWAP_PTR));
    { const Reg a0 = src0, a1 = src1; src0 = sel_reg(swap, a1, a0); src1 = sel_reg(swap, a0, a1); }
    {   // conditionally_erase_fat_pointer_data (pre_state.rs:417-452; register/mod.rs:74-84)
        const bool keeps_ptr = fam == ZK_VMF_RET || fam == ZK_VMF_PTR || fam == ZK_VMF_UMA || fam == ZK_VMF_FAR_CALL;
        const bool erase0 = src0.ptr && !keeps_ptr && !is_kernel, erase1 = src1.ptr && !is_kernel;
        src0.ptr = erase0 ? 0 : src0.ptr; src0.v.l[1] = erase0 ? 0 : src0.v.l[1]; src0.v.l[2] = erase0 ? 0 : src0.v.l[2];
        src1.ptr = erase1 ? 0 : src1.ptr; src1.v.l[1] = erase1 ? 0 : src1.v.l[1]; src1.v.l[2] = erase1 ? 0 : src1.v.l[2];
    }
    const U256 &s0 = src0.v, &s1 = src1.v;
    const bool s0p = src0.ptr != 0, s1p = src1.ptr != 0;

    // ---------------- the opcode that applies
    bool have_dst0 = false, have_dst1 = false, dst0_may_go_to_memory = false;
    Reg dst0 = zero_reg, dst1 = zero_reg;
    bool have_flags = false;
    u32 nf_of = 0, nf_eq = 0, nf_gt = 0;
    bool have_pc = false, have_ergs = false;
    u32 new_pc = 0, new_ergs = 0, pend = 0;
    bool uma_owns_memory_queue = false;
    const bool set_flags = flag(ZK_VMFL_SET_FLAGS);
    auto can_mem = [&](int f) -> bool { return d.can_write_dst0_into_memory[f] != 0; };

    switch (fam) {
    case ZK_VMF_ADD: case ZK_VMF_SUB: {  // add_sub.rs:8-166
        u32 o;
        if (fam == ZK_VMF_ADD) o = u256_add(dst0.v, s0, s1);
        else o = u256_sub(dst0.v, s0, s1);
        have_dst0 = true; dst0_may_go_to_memory = can_mem(ZK_VMF_ADD);
        if (set_flags) { have_flags = true; const bool z = u256_is_zero(dst0.v); nf_of = o; nf_eq = z; nf_gt = !(o || z); }
    } break;
    case ZK_VMF_JUMP:  // jump.rs:3-38
        have_pc = true; new_pc = s0.l[0] & 0xffff;
        break;
    case ZK_VMF_BINOP: {  // binop.rs:14-121
        const bool is_and = var(ZK_VMV_BINOP_AND), is_or = var(ZK_VMV_BINOP_OR);
        for (int i = 0; i < 8; ++i) dst0.v.l[i] = is_or ? (s0.l[i] | s1.l[i]) : is_and ? (s0.l[i] & s1.l[i]) : (s0.l[i] ^ s1.l[i]);
        have_dst0 = true; dst0_may_go_to_memory = can_mem(ZK_VMF_BINOP);
        if (set_flags) { have_flags = true; nf_eq = u256_is_zero(dst0.v); }
    } break;
    case ZK_VMF_CONTEXT: {  // context.rs:7-307
        if (var(ZK_VMV_CTX_SET_CONTEXT_U128)) { for (int i = 0; i < 4; ++i) st.ctx_u128[i] = s0.l[i]; }
        else if (var(ZK_VMV_CTX_SET_ERGS_PER_PUBDATA)) st.ergs_per_pubdata = s0.l[0];
        else if (var(ZK_VMV_CTX_INC_TX_NUMBER)) st.tx_number = st.tx_number + 1;
        else {
            // priority of the selects in the circuit: meta > code_address > caller > this > get_u128 > (ergs_left | sp)
            if (var(ZK_VMV_CTX_META)) {
                dst0.v.l[0] = st.ergs_per_pubdata; dst0.v.l[2] = c.heap_bound; dst0.v.l[3] = c.aux_heap_bound;
                dst0.v.l[7] = c.this_shard | (c.caller_shard << 8) | (c.code_shard << 16);
            } else if (var(ZK_VMV_CTX_CODE_ADDRESS)) { for (int i = 0; i < 5; ++i) dst0.v.l[i] = c.code_address[i]; }
            else if (var(ZK_VMV_CTX_CALLER)) { for (int i = 0; i < 5; ++i) dst0.v.l[i] = c.caller[i]; }
            else if (var(ZK_VMV_CTX_THIS)) { for (int i = 0; i < 5; ++i) dst0.v.l[i] = c.this_[i]; }
            else if (var(ZK_VMV_CTX_GET_CONTEXT_U128)) { for (int i = 0; i < 4; ++i) dst0.v.l[i] = c.ctx_u128[i]; }
            else if (var(ZK_VMV_CTX_ERGS_LEFT)) dst0.v.l[0] = preliminary_ergs_left;
            else dst0.v.l[0] = c.sp;
            have_dst0 = true; dst0_may_go_to_memory = can_mem(ZK_VMF_CONTEXT);
        }
    } break;
    case ZK_VMF_PTR: {  // ptr.rs:6-183
        const bool is_add = var(ZK_VMV_PTR_ADD), is_sub = var(ZK_VMV_PTR_SUB), is_pack = var(ZK_VMV_PTR_PACK), is_shrink = var(ZK_VMV_PTR_SHRINK);
        bool panic = !(s0p && !s1p);
        bool hi_nonzero = false, lo128_nonzero = false;
        for (int i = 1; i < 8; ++i) hi_nonzero |= s1.l[i] != 0;
        for (int i = 0; i < 4; ++i) lo128_nonzero |= s1.l[i] != 0;
        panic = panic || ((is_add || is_sub) && hi_nonzero) || (is_pack && lo128_nonzero);
        const u32 ra = s0.l[0] + s1.l[0]; const bool oa = ra < s0.l[0];
        const u32 rs = s0.l[0] - s1.l[0]; const bool us = s0.l[0] < s1.l[0];
        const u32 rk = s0.l[3] - s1.l[0]; const bool uk = s0.l[3] < s1.l[0];
        panic = panic || (is_add && oa) || (is_sub && us) || (is_shrink && uk);
        if (panic) pend = 1;
        else {
            dst0.ptr = src0.ptr;
            dst0.v = s0;
            if (is_add) dst0.v.l[0] = ra;
            if (is_sub) dst0.v.l[0] = rs;
            if (is_shrink) dst0.v.l[3] = rk;
            if (is_pack) { dst0.v = s0; for (int i = 4; i < 8; ++i) dst0.v.l[i] = s1.l[i]; }
            have_dst0 = true; dst0_may_go_to_memory = can_mem(ZK_VMF_PTR);
        }
    } break;
    case ZK_VMF_MUL: case ZK_VMF_DIV: {  // mul_div.rs:199-417
        if (fam == ZK_VMF_MUL) {
            u32 w[16];
            u256_mul_wide(w, s0, s1);
            bool lo_zero = true, hi_zero = true;
            for (int i = 0; i < 8; ++i) { dst0.v.l[i] = w[i]; dst1.v.l[i] = w[8 + i]; lo_zero &= w[i] == 0; hi_zero &= w[8 + i] == 0; }
            nf_of = !hi_zero; nf_eq = lo_zero; nf_gt = hi_zero && !lo_zero;
        } else {
            const bool dz = u256_is_zero(s1);
            U256 q, r;
            if (dz) { q = u256_zero(); r = u256_zero(); }
            else u256_divrem(q, r, s0, s1);
            dst0.v = q; dst1.v = r;
            nf_of = dz; nf_eq = !dz && u256_is_zero(q); nf_gt = !dz && u256_is_zero(r);
        }
        have_dst0 = true; dst0_may_go_to_memory = can_mem(ZK_VMF_MUL); have_dst1 = true;
        if (set_flags) have_flags = true;
    } break;
    case ZK_VMF_SHIFT: {  // shifts.rs:8-198
        const u32 shift = s1.l[0] & 0xff;
        const bool is_rol = var(ZK_VMV_SHIFT_ROL), is_ror = var(ZK_VMV_SHIFT_ROR), is_shr = var(ZK_VMV_SHIFT_SHR);
        const u32 full_shift = (is_ror && shift) ? 256 - shift : shift;
        const bool is_cyclic = is_rol || is_ror, is_right = is_ror || is_shr;
        U256 r = s0;
        if (is_right && !is_cyclic) {
            limbs_shr<8>(r.l, full_shift);
        } else {
            u32 w[16];
            for (int i = 0; i < 16; ++i) w[i] = i < 8 ? s0.l[i] : 0;
            limbs_shl<16>(w, full_shift);
            for (int i = 0; i < 8; ++i) r.l[i] = w[i] + (is_cyclic ? w[8 + i] : 0);  // disjoint bit ranges: + == |
        }
        dst0.v = r;
        have_dst0 = true; dst0_may_go_to_memory = can_mem(ZK_VMF_SHIFT);
        if (set_flags) { have_flags = true; nf_eq = u256_is_zero(r); }
    } break;
    case ZK_VMF_LOG: {  // log.rs:16-463
        const bool is_read = var(ZK_VMV_LOG_STORAGE_READ), is_write = var(ZK_VMV_LOG_STORAGE_WRITE), is_event = var(ZK_VMV_LOG_EVENT),
                   is_l1 = var(ZK_VMV_LOG_TO_L1), is_pre = var(ZK_VMV_LOG_PRECOMPILE_CALL);
        LogQ q;
        q.key = s0;
        if (is_pre && q.key.l[4] == 0) q.key.l[4] = heap_page;
        if (is_pre && q.key.l[5] == 0) q.key.l[5] = heap_page;
        const bool is_storage = is_read || is_write;
        const bool is_revertable = !(is_read || is_pre);
        q.aux_byte = (is_storage ? d.params[ZK_VMP_STORAGE_AUX_BYTE] : 0) + (is_event ? d.params[ZK_VMP_EVENT_AUX_BYTE] : 0) +
                     (is_l1 ? d.params[ZK_VMP_L1_MESSAGE_AUX_BYTE] : 0) + (is_pre ? d.params[ZK_VMP_PRECOMPILE_AUX_BYTE] : 0);
        const u32 refund = env.refund(true);
        u32 burn = 0;
        if (is_write && c.this_shard == 0) burn = st.ergs_per_pubdata * (d.params[ZK_VMP_INITIAL_STORAGE_WRITE_PUBDATA_BYTES] - refund);
        if (is_pre) burn = s1.l[0];
        if (is_l1) burn = st.ergs_per_pubdata * d.params[ZK_VMP_L1_MESSAGE_PUBDATA_BYTES];
        const bool not_enough = preliminary_ergs_left < burn;
        const u32 ergs_rem = not_enough ? 0 : preliminary_ergs_left - burn;
        const bool execute = !not_enough;
        U256 read_w;
        env.log_read(execute && is_storage, read_w);
        const U256 read_value = sel256(is_storage, read_w, u256_zero());
        for (int i = 0; i < 5; ++i) q.address[i] = c.this_[i];
        q.read_value = read_value;
        q.written_value = sel256(is_revertable, s1, read_value);
        q.rw_flag = is_revertable; q.rollback = 0; q.is_service = flag(ZK_VMFL_FIRST_MESSAGE); q.shard_id = c.this_shard;
        q.tx_number = st.tx_number; q.timestamp = ts1;
        const bool execute_rollback = execute && is_revertable;
        u64 prev_head[4];
        env.log_prev_head(execute_rollback, prev_head);
        if (execute) {
            u64 enc[20];
            log_query_encode(enc, q);
            env.fwd_push(enc);
            st.fwd_len += 1;
        }
        if (execute_rollback) {
            for (int i = 0; i < 4; ++i) c.rq_head[i] = prev_head[i];
            c.rq_len += 1;
        }
        have_ergs = true; new_ergs = ergs_rem;
        if (is_read) { have_dst0 = true; dst0.v = read_value; }
        else if (is_pre) { have_dst0 = true; dst0.v.l[0] = execute; }
        dst0_may_go_to_memory = can_mem(ZK_VMF_LOG);
    } break;
    case ZK_VMF_UMA: {  // uma.rs:18-990
        const bool is_hr = var(ZK_VMV_UMA_HEAP_READ), is_hw = var(ZK_VMV_UMA_HEAP_WRITE), is_ar = var(ZK_VMV_UMA_AUX_HEAP_READ),
                   is_aw = var(ZK_VMV_UMA_AUX_HEAP_WRITE), is_fp = var(ZK_VMV_UMA_FAT_PTR_READ);
        const bool inc = flag(ZK_VMFL_UMA_INCREMENT);
        const bool access_heap = is_hr || is_hw, access_aux = is_ar || is_aw;
        const bool not_a_ptr = is_fp && !s0p;
        const u32 offset = s0.l[0], page = s0.l[1], start = s0.l[2], length = s0.l[3];
        const bool skip_legit = is_fp && !(offset < length);
        const u32 formal_start = is_fp ? start : 0;
        const u32 absolute_address = formal_start + offset;
        const u32 incremented_offset = offset + 32;
        const bool non_addr = incremented_offset < offset || incremented_offset == 0xffffffffu;
        const bool q_panic = not_a_ptr || non_addr;
        const bool q_skip = not_a_ptr || skip_legit || non_addr;
        const bool ufb = incremented_offset < length;
        const u32 boob = (q_skip || ufb) ? 0 : incremented_offset - length;
        const u32 bytes_to_cleanup = boob % 32;
        u32 growth = 0;
        u32 new_heap_bound = c.heap_bound, new_aux_bound = c.aux_heap_bound;
        if (access_heap) {
            const bool uf = incremented_offset < c.heap_bound;
            growth = uf ? 0 : incremented_offset - c.heap_bound;
            new_heap_bound = uf ? c.heap_bound : incremented_offset;
        }
        if (access_aux) {
            const bool uf = incremented_offset < c.aux_heap_bound;
            growth = uf ? 0 : incremented_offset - c.aux_heap_bound;
            new_aux_bound = uf ? c.aux_heap_bound : incremented_offset;
        }
        bool top_nonzero = false;
        for (int i = 1; i < 8; ++i) top_nonzero |= s0.l[i] != 0;
        const bool oob = (access_heap || access_aux) && (top_nonzero || non_addr);
        if (oob) growth = 0xffffffffu;
        const bool ufe = preliminary_ergs_left < growth;
        const u32 ergs_after = ufe ? 0 : preliminary_ergs_left - growth;
        const bool set_panic = q_panic || ufe || oob;
        const bool skip_mem = q_skip || set_panic;
        const u32 cell = absolute_address / 32, unalign = absolute_address % 32;
        const u32 mem_page = access_heap ? heap_page : (access_aux ? aux_heap_page : page);
        const u32 cell_b = cell + 1;
        const bool read_a = !skip_mem, read_b = !skip_mem && unalign != 0;
        U256 va, vb;
        env.uma_read(0, read_a, va);
        env.uma_read(1, read_b, vb);
        if (!read_a) va = u256_zero();
        if (!read_b) vb = u256_zero();
        u64 enc[8];
        if (read_a) { memory_query_encode(enc, ts0, mem_page, cell, 0, 0, va); env.mem_push(enc); st.mem_len += 1; }
        if (read_b) { memory_query_encode(enc, ts0, mem_page, cell_b, 0, 0, vb); env.mem_push(enc); st.mem_len += 1; }
        // the 64-byte big-endian window [va | vb] as one 512-bit integer X = va * 2^256 + vb: the word at byte `unalign` is
        // (X >> 8 (32 - unalign)) mod 2^256
        const u32 window_shift = 8 * (32 - unalign);  // 8 .. 256
        u32 X[16];
        for (int i = 0; i < 8; ++i) { X[i] = vb.l[i]; X[8 + i] = va.l[i]; }
        U256 read_value;
        {
            u32 t[16];
            for (int i = 0; i < 16; ++i) t[i] = X[i];
            limbs_shr<16>(t, window_shift);
            for (int i = 0; i < 8; ++i) read_value.l[i] = t[i];
        }
        {   // fat-pointer reads past the slice: the lowest `nclean` bytes are zeroed (src/tables/uma_ptr_read_cleanup.rs)
            const u32 nbits = 8 * (is_fp ? bytes_to_cleanup : 0);
            for (u32 i = 0; i < 8; ++i) {
                const u32 m = 32 * (i + 1) <= nbits ? 0u : (32 * i >= nbits ? 0xffffffffu : ~((1u << (nbits - 32 * i)) - 1u));
                read_value.l[i] &= m;
            }
        }
        const bool is_write_access = is_hw || is_aw;
        const bool exec_write = is_write_access && !skip_mem;
        if (exec_write) {
            u32 wv[16], wm[16];
            for (int i = 0; i < 16; ++i) { wv[i] = i < 8 ? s1.l[i] : 0; wm[i] = i < 8 ? 0xffffffffu : 0; }
            limbs_shl<16>(wv, window_shift);
            limbs_shl<16>(wm, window_shift);
            U256 na, nb;
            for (int i = 0; i < 8; ++i) { nb.l[i] = (X[i] & ~wm[i]) | wv[i]; na.l[i] = (X[8 + i] & ~wm[8 + i]) | wv[8 + i]; }
            memory_query_encode(enc, ts3, mem_page, cell, 1, 0, na); env.mem_push(enc); st.mem_len += 1;
            if (unalign) { memory_query_encode(enc, ts3, mem_page, cell_b, 1, 0, nb); env.mem_push(enc); st.mem_len += 1; }
        }
        Reg incremented_src0 = src0;
        incremented_src0.v.l[0] = incremented_offset;
        if (set_panic) pend = 1;
        else {
            if (is_write_access && inc) { have_dst0 = true; dst0 = incremented_src0; }
            else if (!is_write_access) { have_dst0 = true; dst0.ptr = 0; dst0.v = read_value; }
            if (!is_write_access && inc) { have_dst1 = true; dst1 = incremented_src0; }
        }
        dst0_may_go_to_memory = can_mem(ZK_VMF_UMA);
        if (access_heap) c.heap_bound = new_heap_bound;
        if (access_aux) c.aux_heap_bound = new_aux_bound;
        have_ergs = true; new_ergs = ergs_after;
        uma_owns_memory_queue = true;  // memory_queue_candidates is applied after the dst0 write (cycle.rs state-diff order)
    } break;
    case ZK_VMF_NEAR_CALL: case ZK_VMF_FAR_CALL: case ZK_VMF_RET: {  // call_ret.rs:24-512 + call_ret_impl/*
        const u32 fwd_byte = u256_byte(s0, d.params[ZK_VMP_FAR_CALL_FORWARDING_MODE_BYTE_IDX]);
        const bool use_aux_heap = fwd_byte == d.params[ZK_VMP_FORWARD_USE_AUX_HEAP], forward_fat_pointer = fwd_byte == d.params[ZK_VMP_FORWARD_FAT_POINTER];
        const bool use_heap = !(use_aux_heap || forward_fat_pointer);
        const u32 offset = s0.l[0], page = s0.l[1], start = s0.l[2], length = s0.l[3];
        const u32 end_non_inclusive = start + length;
        const bool range_overflow = end_non_inclusive < start;
        const bool ptr_invalid = (offset != 0 && !forward_fat_pointer) || range_overflow || length < offset;
        const FatPtr fp = sel_fp(ptr_invalid, FatPtr{0, 0, 0, 0}, FatPtr{offset, page, start, length});
        const u32 upper_bound_abi = end_non_inclusive;
        Ctx old_ctx = c, new_ctx;
        bool fwd_is_set = false;
        u64 fwd_set_to[4] = {0, 0, 0, 0};
        u32 ret_panic_flag = 0;
        u64 prev_sponge[12];
        const bool apply_ret = fam == ZK_VMF_RET;
        if (fam == ZK_VMF_NEAR_CALL) {  // near_call.rs:32-184
            old_ctx.pc = pc_plus_one;
            new_ctx = old_ctx;
            u64 tail[4];
            env.near_call_tail(true, tail);
            for (int i = 0; i < 4; ++i) { new_ctx.rq_tail[i] = tail[i]; new_ctx.rq_head[i] = tail[i]; }
            new_ctx.rq_len = 0;
            const u32 passed_abi = s0.l[0];
            const u32 to_pass = passed_abi == 0 ? preliminary_ergs_left : passed_abi;
            const bool uf = preliminary_ergs_left < to_pass;
            old_ctx.ergs = uf ? 0 : preliminary_ergs_left - to_pass;
            new_ctx.ergs = uf ? preliminary_ergs_left : to_pass;
            new_ctx.pc = imm0; new_ctx.eh = imm1; new_ctx.is_local = 1;
        } else if (fam == ZK_VMF_FAR_CALL) {  // far_call.rs:268-1603
            const bool is_delegate = var(ZK_VMV_FAR_DELEGATE), is_mimic = var(ZK_VMV_FAR_MIMIC);
            old_ctx.pc = pc_plus_one;
            {
                u64 z[CTX_WORDS];
                for (int i = 0; i < CTX_WORDS; ++i) z[i] = 0;
                ctx_unflatten(new_ctx, z);
            }
            new_ctx.heap_bound = d.params[ZK_VMP_NEW_FRAME_MEMORY_STIPEND];
            new_ctx.aux_heap_bound = d.params[ZK_VMP_NEW_FRAME_MEMORY_STIPEND];
            const Reg& implicit = st.regs[d.params[ZK_VMP_CALL_IMPLICIT_PARAMETER_REG_IDX] % NREG];
            const bool is_static_call = flag(ZK_VMFL_FAR_CALL_STATIC), is_call_shard = flag(ZK_VMFL_FAR_CALL_SHARD);
            const u32 abi_shard = u256_byte(s0, d.params[ZK_VMP_FAR_CALL_SHARD_ID_BYTE_IDX]);
            bool ctor = u256_byte(s0, d.params[ZK_VMP_FAR_CALL_CONSTRUCTOR_CALL_BYTE_IDX]) != 0;
            bool syscall = u256_byte(s0, d.params[ZK_VMP_FAR_CALL_SYSTEM_CALL_BYTE_IDX]) != 0;
            const u32 caller_shard = c.this_shard;
            const u32 dest_shard = is_call_shard ? abi_shard : caller_shard;
            const bool target_is_zkporter = dest_shard != 0;
            const bool target_is_kernel = (s1.l[0] >> 16) == 0 && s1.l[1] == 0 && s1.l[2] == 0 && s1.l[3] == 0 && s1.l[4] == 0;
            ctor = ctor && c.is_kernel;
            syscall = syscall && target_is_kernel;
            const u32 default_page = st.page_counter;
            st.page_counter = st.page_counter + d.params[ZK_VMP_NEW_MEMORY_PAGES_PER_FAR_CALL];
            // may_be_read_code_hash (far_call.rs:1104-1280)
            const bool zkporter_ok = G.zkporter_is_available != 0;
            const bool should_read = !target_is_zkporter || zkporter_ok;
            const bool needs_porter_mask = target_is_zkporter && !zkporter_ok;
            U256 code_hash;
            env.far_code_hash(should_read, code_hash);
            if (should_read) {
                LogQ q;
                q.address[0] = d.params[ZK_VMP_DEPLOYER_SYSTEM_CONTRACT_ADDRESS_LOW];
                for (int i = 1; i < 5; ++i) q.address[i] = 0;
                q.key = u256_zero();
                for (int i = 0; i < 5; ++i) q.key.l[i] = s1.l[i];
                q.read_value = code_hash; q.written_value = code_hash;
                q.aux_byte = d.params[ZK_VMP_STORAGE_AUX_BYTE]; q.rw_flag = 0; q.rollback = 0; q.is_service = 0; q.shard_id = dest_shard;
                q.tx_number = st.tx_number; q.timestamp = ts1;
                u64 enc[20];
                log_query_encode(enc, q);
                env.fwd_push(enc);
                st.fwd_len += 1;
            }
            U256 bytecode_hash = code_hash;
            const bool empty = u256_is_zero(bytecode_hash);
            const bool mask_default_aa = should_read && empty && !target_is_kernel;
            bytecode_hash = sel256(mask_default_aa, G.default_aa_code_hash, bytecode_hash);
            if (needs_porter_mask) bytecode_hash = u256_zero();
            const bool trivial = (empty && !mask_default_aa) || needs_porter_mask || !should_read;
            u32 target_page = trivial ? 0 : default_page;
            const u32 top = bytecode_hash.l[7];
            const u32 version_byte = (top >> 24) & 0xff, marker_byte = (top >> 16) & 0xff;
            const bool normal_marker = marker_byte == 0, ctor_marker = marker_byte == d.params[ZK_VMP_CODE_YET_CONSTRUCTED_MARKER];
            const bool code_format_exception = version_byte != d.params[ZK_VMP_CODE_HASH_VERSION_BYTE] || !(normal_marker || ctor_marker);
            const bool can_call_code = (normal_marker && !ctor) || (ctor_marker && ctor);
            U256 at_rest = bytecode_hash;
            at_rest.l[7] = (top & 0xffff) | (d.params[ZK_VMP_CODE_AT_REST_MARKER] << 16) | (d.params[ZK_VMP_CODE_HASH_VERSION_BYTE] << 24);
            const U256 masked_hash = sel256(can_call_code, at_rest, sel256(target_is_kernel, u256_zero(), G.default_aa_code_hash));
            const u32 code_len_words = code_format_exception ? 0 : (masked_hash.l[7] & 0xffff);
            const bool exceptions = code_format_exception || (!can_call_code && target_is_kernel) || (forward_fat_pointer && !s0p) || ptr_invalid || range_overflow;
            FatPtr final_fp = sel_fp(forward_fat_pointer, fp_readjust(fp), FatPtr{0, use_heap ? heap_page : aux_heap_page, fp.start, fp.length});
            if (exceptions) final_fp = FatPtr{0, 0, 0, 0};
            u32 upper = exceptions ? 0 : upper_bound_abi;
            if (range_overflow && !forward_fat_pointer) upper = 0xffffffffu;
            u32 growth = 0;
            if (use_heap) {
                const bool uf = upper < old_ctx.heap_bound;
                growth = uf ? 0 : upper - old_ctx.heap_bound;
                old_ctx.heap_bound = uf ? old_ctx.heap_bound : upper;
            }
            if (use_aux_heap) {
                const bool uf = upper < old_ctx.aux_heap_bound;
                growth = uf ? 0 : upper - old_ctx.aux_heap_bound;
                old_ctx.aux_heap_bound = uf ? old_ctx.aux_heap_bound : upper;
            }
            const bool ufg = preliminary_ergs_left < growth;
            const u32 ergs_after_growth = ufg ? 0 : preliminary_ergs_left - growth;
            const bool exception = exceptions || ufg;
            const bool should_decommit0 = !exception;
            target_page = should_decommit0 ? target_page : 0;
            const u32 dcost = d.params[ZK_VMP_ERGS_PER_CODE_WORD_DECOMMITTMENT] * code_len_words;
            const bool ufd = ergs_after_growth < dcost;
            const bool should_decommit = should_decommit0 && !ufd;
            u32 ergs_rem = should_decommit ? ergs_after_growth - dcost : ergs_after_growth;
            const u32 suggested = env.far_decommit_page(should_decommit);
            const bool is_first = target_page == suggested;
            if (should_decommit && !is_first) ergs_rem = ergs_after_growth;
            if (should_decommit) {
                u64 enc[8];
                decommit_query_encode(enc, masked_hash, suggested, is_first, ts1);
                env.dec_push(enc);
                st.dec_len += 1;
            }
            const u32 code_memory_page = should_decommit ? suggested : d.params[ZK_VMP_UNMAPPED_PAGE];
            pend = exception || ufd;
            u64 tail[4];
            env.far_call_tail(true, tail);
            for (int i = 0; i < 4; ++i) { new_ctx.rq_tail[i] = tail[i]; new_ctx.rq_head[i] = tail[i]; }
            new_ctx.rq_len = 0;
            const u32 max_passable = (ergs_rem / 64) * 63;
            const u32 leftover = ergs_rem - max_passable;
            const u32 passed_abi = s0.l[6];
            const bool ufp = max_passable < passed_abi;
            const u32 to_pass = ufp ? max_passable : passed_abi;
            old_ctx.ergs = ufp ? leftover : leftover + (max_passable - passed_abi);
            new_ctx.ergs = to_pass; new_ctx.pc = 0; new_ctx.eh = imm0;
            new_ctx.is_static = is_static_call || old_ctx.is_static;
            new_ctx.is_kernel = is_delegate ? old_ctx.is_kernel : (u32)target_is_kernel;
            new_ctx.code_shard = dest_shard;
            for (int i = 0; i < 5; ++i) new_ctx.code_address[i] = s1.l[i];
            new_ctx.this_shard = is_delegate ? caller_shard : dest_shard;
            for (int i = 0; i < 5; ++i) new_ctx.this_[i] = is_delegate ? old_ctx.this_[i] : s1.l[i];
            for (int i = 0; i < 5; ++i) new_ctx.caller[i] = is_mimic ? implicit.v.l[i] : (is_delegate ? old_ctx.caller[i] : old_ctx.this_[i]);
            new_ctx.caller_shard = caller_shard;
            new_ctx.code_page = code_memory_page; new_ctx.base_page = default_page;
            for (int i = 0; i < 4; ++i) new_ctx.ctx_u128[i] = is_delegate ? old_ctx.ctx_u128[i] : st.ctx_u128[i];
            new_ctx.is_local = 0;
            // registers (far_call.rs:1006-1071; call_ret.rs:419-465)
            const u32 abi0 = d.params[ZK_VMP_CALL_SYSTEM_ABI_REGISTERS_BEGIN], abi1 = d.params[ZK_VMP_CALL_SYSTEM_ABI_REGISTERS_END];
            const u32 res0 = d.params[ZK_VMP_CALL_RESERVED_RANGE_BEGIN], res1 = d.params[ZK_VMP_CALL_RESERVED_RANGE_END],
                      imp = d.params[ZK_VMP_CALL_IMPLICIT_PARAMETER_REG_IDX];
            for (u32 r = 0; r < (u32)NREG; ++r) {
                if (r >= abi0 && r < abi1) { st.regs[r].ptr = 0; if (!syscall) st.regs[r].v = u256_zero(); }
                if ((r >= res0 && r < res1) || r == imp) { st.regs[r].ptr = 0; st.regs[r].v = u256_zero(); }
            }
            st.regs[0].ptr = 1; st.regs[0].v = u256_zero();
            st.regs[0].v.l[0] = final_fp.offset; st.regs[0].v.l[1] = final_fp.page; st.regs[0].v.l[2] = final_fp.start; st.regs[0].v.l[3] = final_fp.length;
            st.regs[1].ptr = 0; st.regs[1].v = u256_zero();
            st.regs[1].v.l[0] = (u32)ctor + 2 * (u32)syscall;
            for (int i = 0; i < 4; ++i) st.ctx_u128[i] = 0;
        } else {  // ret.rs:29-479
            const bool is_revert = var(ZK_VMV_RET_REVERT), is_panic = var(ZK_VMV_RET_PANIC);
            const bool is_local = c.is_local != 0;
            const bool r0p = is_panic ? false : s0p;
            u64 popped_flat[CTX_WORDS];
            env.ret_pop(true, popped_flat, prev_sponge);
            Ctx popped;
            ctx_unflatten(popped, popped_flat);
            old_ctx = popped;
            new_ctx = popped;
            const bool is_far_return = !is_local;
            // the circuit erases src0 on panic BEFORE parsing nothing else of it: the ABI parts above were parsed from the unerased value
            const bool exc = (forward_fat_pointer && !r0p && is_far_return) || (forward_fat_pointer && fp.page < c.base_page) || is_panic;
            FatPtr fpr = sel_fp(exc, FatPtr{0, 0, 0, 0}, fp);
            fpr = sel_fp(forward_fat_pointer, fp_readjust(fpr), FatPtr{0, use_heap ? heap_page : aux_heap_page, fpr.start, fpr.length});
            u32 upper = exc ? 0 : upper_bound_abi;
            if (range_overflow && !forward_fat_pointer) upper = 0xffffffffu;
            u32 growth = 0;
            if (use_heap && is_far_return) growth = upper < c.heap_bound ? 0 : upper - c.heap_bound;
            if (use_aux_heap && is_far_return) growth = upper < c.aux_heap_bound ? 0 : upper - c.aux_heap_bound;
            const bool ufg = preliminary_ergs_left < growth;
            u32 ergs_after = ufg ? 0 : preliminary_ergs_left - growth;
            if (is_local) ergs_after = preliminary_ergs_left;
            const bool non_local_panic = (exc || ufg || is_panic) && is_far_return;
            const FatPtr final_fp = sel_fp(non_local_panic, FatPtr{0, 0, 0, 0}, fpr);
            new_ctx.ergs = ergs_after + popped.ergs;
            if (is_local) { new_ctx.heap_bound = c.heap_bound; new_ctx.aux_heap_bound = c.aux_heap_bound; }
            const bool perform_revert = is_revert || is_panic || non_local_panic;
            if (perform_revert) {
                fwd_is_set = true;
                for (int i = 0; i < 4; ++i) fwd_set_to[i] = c.rq_tail[i];
                st.fwd_len = st.fwd_len + c.rq_len;
            } else {
                for (int i = 0; i < 4; ++i) new_ctx.rq_head[i] = c.rq_head[i];
                new_ctx.rq_len = popped.rq_len + c.rq_len;
            }
            const bool use_label = flag(ZK_VMFL_RET_TO_LABEL) && is_local;
            const u32 ok_pc = use_label ? imm0 : popped.pc;
            const u32 eh_pc = use_label ? imm0 : c.eh;
            new_ctx.pc = perform_revert ? eh_pc : ok_pc;
            if (is_far_return) {
                for (int r = 0; r < NREG; ++r) { st.regs[r].ptr = 0; st.regs[r].v = u256_zero(); }
                st.regs[0].ptr = 1;
                st.regs[0].v.l[0] = final_fp.offset; st.regs[0].v.l[1] = final_fp.page; st.regs[0].v.l[2] = final_fp.start; st.regs[0].v.l[3] = final_fp.length;
                for (int i = 0; i < 4; ++i) st.ctx_u128[i] = 0;
            }
            ret_panic_flag = is_panic || non_local_panic;
        }
        // merge (call_ret.rs:119-330)
        if (apply_ret) {
            env.sponge_set(prev_sponge);
            st.depth = st.depth - 1;
        } else {
            u64 enc[32];
            ctx_encode(enc, old_ctx);
            env.sponge_push(enc);
            st.depth = st.depth + 1;
        }
        st.ctx = new_ctx;
        if (fwd_is_set) env.fwd_set(fwd_set_to);
        have_flags = true; nf_of = apply_ret ? ret_panic_flag : 0; nf_eq = 0; nf_gt = 0;
    } break;
    default: break;  // NOP
    }

    // ---------------- apply state diffs (cycle.rs:160-616)
    Ctx& nc = st.ctx;
    if (have_dst0) {
        if (dst0_may_go_to_memory && dst0_in_memory) {
            if (!uma_owns_memory_queue) {
                u64 enc[8];
                memory_query_encode(enc, ts3, dst0_page, dst0_index, 1, dst0.ptr, dst0.v);
                env.mem_push(enc);
                st.mem_len += 1;
            }
        } else if (dst0_idx) {
            st.regs[dst0_idx - 1] = dst0;
        }
    }
    if (dst1_idx) {  // written unconditionally from the (possibly empty) dot product
        if (!have_dst1) dst1 = zero_reg;
        st.regs[dst1_idx - 1] = dst1;
    }
    if (have_pc) nc.pc = new_pc;
    if (have_ergs) nc.ergs = new_ergs;
    if (have_flags) { st.of = nf_of; st.eq = nf_eq; st.gt = nf_gt; }
    st.pending_exception = pend;
    st.last_family = fam;
}

}  // namespace vmn
