// kernels_vm_seed.hpp — chain-specialised seeding of main_vm's carried state (the sequential half of witness resolution).
//
// The reference resolves `state = vm_cycle(state, ..)` (src/main_vm/mod.rs:102-110) cycle after cycle on one core.  Round 2 ran the
// recorded cone of the carried outputs through an interpreter, one wavefront per instance: 330 us per cycle.  What is sequential
// in a cycle is much less than its cone:
//   phase A  k_vm_walk    the non-hash VmLocalState (registers, flags, pc, callstack scalars ...): a few hundred integer operations
//                         per cycle once the opcode is decoded natively (vm_native.hpp) — one thread per instance, few instances
//                         per wavefront so that instances on different opcodes do not serialise each other.  It also lists, per
//                         instance, what each of the four Poseidon2 chains absorbs (memory queue: src/main_vm/utils.rs:194-213,
//                         cycle.rs:846-884, uma.rs:706-726; decommit queue: far_call.rs:1418-1603; forward log queue: log.rs:508-609;
//                         callstack sponge: call_ret.rs:170-270) and how many events precede every cycle;
//   phase B  k_vm_chains  every chain is independent of the others once phase A fixed what is absorbed: 12 lanes own one chain, one
//                         state element each — S-boxes in parallel, linear layers through an LDS exchange — and walk its events;
//                         the state after every event goes to a snapshot array;
//   phase C  k_vm_fill    lane = (instance, cycle): the chain words of the cycle's input state = snapshot[events before the cycle].
// Results are the same 243 words per cycle the cone kernels produce (tests: == native restatement, == k_seed_wave).
#pragma once
#include "kernels_seed_wave.hpp"
#include "vm_native.hpp"

namespace zkvm {

using vmn::u32;
using vmn::u64;

struct RawLayout {  // first word of every WitnessOracle field inside the loop input stream (zk_circuit_main_vm_layout)
    u32 code_word, src0_value, src0_is_ptr, refund, log_read, log_prev_head, near_tail, far_code_hash, far_page, far_tail, ret_ctx, ret_state,
        uma_a, uma_b;
};

constexpr u32 EV_MEM = 8, EV_DEC = 8, EV_FWD = 24, EV_SP = 36;  // u64 words per event record (payload, then the type for FWD / SP)
constexpr u32 FWD_TYPE = 20, SP_TYPE = 32;                      // type word: 1 = push, 2 = set
constexpr u32 MEM_EVENTS_PER_CYCLE = 6;                          // code, src0, then UMA's 2 reads + 2 writes or the dst0 write

struct SeedDev {
    vmn::Defs D;  // D.d -> device copy of the blob
    u64* loop; u64 in_stride; u32 limit, n_instances;
    RawLayout raw;
    const u64* outer_store; u64 outer_n_store;
    const u32* state0_slot;     // [243] outer store slot of the FIRST link of every state word
    const u64* outer_inputs; u64 outer_in_stride; u32 w_zkporter, w_default_aa;
    u64 *mem_ev, *dec_ev, *fwd_ev, *sp_ev;
    u64 *mem_snap, *dec_snap, *fwd_snap, *sp_snap;  // [inst][cap][12 | 12 | 4 | 12]
    u32 cap_mem, cap_one;
    uint4* counts;              // [inst][limit]: events of (mem, dec, fwd, sponge) in cycles < c
    uint4* totals;              // [inst]
    u32 lanes_per_wave;         // instances per wavefront in phase A
};

__device__ __forceinline__ u64 outer_value(const SeedDev& a, u32 inst, u32 slot) {
    return a.outer_store[((u64)(inst >> 6) * a.outer_n_store + slot) * 64 + (inst & 63)];
}

struct DevEnv {
    const u64* col; u64 stride; const RawLayout* raw;
    u64 *mem_ev, *dec_ev, *fwd_ev, *sp_ev;
    u32 n_mem = 0, n_dec = 0, n_fwd = 0, n_sp = 0;
    __device__ void load256(u32 w, bool exec, vmn::U256& o) const {
        if (exec) { for (int i = 0; i < 8; ++i) o.l[i] = (u32)col[(u64)(w + i) * stride]; }
        else o = vmn::u256_zero();
    }
    __device__ void load4(u32 w, bool exec, u64 o[4]) const { for (int i = 0; i < 4; ++i) o[i] = exec ? col[(u64)(w + i) * stride] : 0; }
    __device__ void code_word(bool exec, vmn::U256& o) { load256(raw->code_word, exec, o); }
    __device__ void src0(bool exec, vmn::U256& v, u32& is_ptr) { load256(raw->src0_value, exec, v); is_ptr = exec ? (u32)col[(u64)raw->src0_is_ptr * stride] : 0; }
    __device__ u32 refund(bool exec) { return exec ? (u32)col[(u64)raw->refund * stride] : 0; }
    __device__ void log_read(bool exec, vmn::U256& o) { load256(raw->log_read, exec, o); }
    __device__ void log_prev_head(bool exec, u64 o[4]) { load4(raw->log_prev_head, exec, o); }
    __device__ void near_call_tail(bool exec, u64 o[4]) { load4(raw->near_tail, exec, o); }
    __device__ void far_code_hash(bool exec, vmn::U256& o) { load256(raw->far_code_hash, exec, o); }
    __device__ u32 far_decommit_page(bool exec) { return exec ? (u32)col[(u64)raw->far_page * stride] : 0; }
    __device__ void far_call_tail(bool exec, u64 o[4]) { load4(raw->far_tail, exec, o); }
    __device__ void ret_pop(bool, u64 ctx42[42], u64 state[12]) {
        for (int i = 0; i < 42; ++i) ctx42[i] = col[(u64)(raw->ret_ctx + i) * stride];
        for (int i = 0; i < 12; ++i) state[i] = col[(u64)(raw->ret_state + i) * stride];
    }
    __device__ void uma_read(int which, bool exec, vmn::U256& o) { load256(which ? raw->uma_b : raw->uma_a, exec, o); }
    __device__ void mem_push(const u64 enc[8]) { u64* p = mem_ev + (u64)n_mem * EV_MEM; for (int i = 0; i < 8; ++i) p[i] = enc[i]; ++n_mem; }
    __device__ void dec_push(const u64 enc[8]) { u64* p = dec_ev + (u64)n_dec * EV_DEC; for (int i = 0; i < 8; ++i) p[i] = enc[i]; ++n_dec; }
    __device__ void fwd_push(const u64 enc[20]) { u64* p = fwd_ev + (u64)n_fwd * EV_FWD; for (int i = 0; i < 20; ++i) p[i] = enc[i]; p[FWD_TYPE] = 1; ++n_fwd; }
    __device__ void fwd_set(const u64 v[4]) { u64* p = fwd_ev + (u64)n_fwd * EV_FWD; for (int i = 0; i < 4; ++i) p[i] = v[i]; p[FWD_TYPE] = 2; ++n_fwd; }
    __device__ void sponge_push(const u64 enc[32]) { u64* p = sp_ev + (u64)n_sp * EV_SP; for (int i = 0; i < 32; ++i) p[i] = enc[i]; p[SP_TYPE] = 1; ++n_sp; }
    __device__ void sponge_set(const u64 v[12]) { u64* p = sp_ev + (u64)n_sp * EV_SP; for (int i = 0; i < 12; ++i) p[i] = v[i]; p[SP_TYPE] = 2; ++n_sp; }
};

// ---- phase A: one thread per instance; only `lanes_per_wave` lanes of a wavefront are used
__global__ __launch_bounds__(64) void k_vm_walk(SeedDev a) {
    const u32 lane = threadIdx.x;
    if (lane >= a.lanes_per_wave) return;
    const u32 inst = blockIdx.x * a.lanes_per_wave + lane;
    if (inst >= a.n_instances) return;
    vmn::Defs D = a.D;
    D.zkporter_is_available = (u32)a.outer_inputs[(u64)a.w_zkporter * a.outer_in_stride + inst];
    for (int i = 0; i < 8; ++i) D.default_aa_code_hash.l[i] = (u32)a.outer_inputs[(u64)(a.w_default_aa + i) * a.outer_in_stride + inst];
    vmn::State st;
    vmn::state_unflatten(st, [&](int w) { return outer_value(a, inst, a.state0_slot[w]); });
    const u64 lane0 = (u64)inst * a.limit;
    DevEnv env;
    env.stride = a.in_stride; env.raw = &a.raw;
    env.mem_ev = a.mem_ev + (u64)inst * a.cap_mem * EV_MEM;
    env.dec_ev = a.dec_ev + (u64)inst * a.cap_one * EV_DEC;
    env.fwd_ev = a.fwd_ev + (u64)inst * a.cap_one * EV_FWD;
    env.sp_ev = a.sp_ev + (u64)inst * a.cap_one * EV_SP;
    // cycle 0 takes the outer scope's words verbatim (chain words included: they are snapshot 0 of every chain)
    for (int w = 0; w < vmn::STATE_WORDS; ++w) a.loop[(u64)w * a.in_stride + lane0] = outer_value(a, inst, a.state0_slot[w]);
    for (u32 c = 0; c < a.limit; ++c) {
        u64* const out = a.loop + lane0 + c;
        if (c) vmn::state_flatten(st, [&](int w, u64 v) { out[(u64)w * a.in_stride] = v; });
        a.counts[lane0 + c] = make_uint4(env.n_mem, env.n_dec, env.n_fwd, env.n_sp);
        env.col = out;
        vmn::vm_cycle(D, st, env);
    }
    a.totals[inst] = make_uint4(env.n_mem, env.n_dec, env.n_fwd, env.n_sp);
}

// ---- phase B: 16-lane groups, 12 lanes = the 12 state elements of one chain
constexpr u32 CH_GROUPS = 4;   // per wavefront
struct ChainCtx {
    u64* xbuf;   // 2 x 64 words of LDS per wavefront
    u32 lane, gbase, e, flip;
    __device__ __forceinline__ void exchange(u64 x, u64 v[12]) {
        u64* const xb = xbuf + flip * 64;
        flip ^= 1;
        xb[lane] = x;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = xb[gbase + i];
    }
    // one Poseidon2 permutation; lane e of the group holds state element e (lanes 12..15 compute on junk, never read)
    __device__ __forceinline__ u64 permute(u64 x) {
        const u32 el = e < 12 ? e : 0;
        u64 v[12];
        exchange(x, v);
        x = zke::coop_mds_external(v, el);
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
#pragma unroll 1
            for (int r4 = 0; r4 < 4; ++r4) {
                x = gl::pow7(gl::add(x, p2::RC[12 * (half * 26 + r4) + el]));
                exchange(x, v);
                x = zke::coop_mds_external(v, el);
            }
            if (half == 0) {
#pragma unroll 1
                for (int rr = 4; rr < 26; ++rr) {
                    const u64 sb = gl::pow7(gl::add(x, p2::RC[12 * rr]));
                    x = el == 0 ? sb : x;
                    exchange(x, v);
                    x = zke::coop_mds_inner(v, el);
                }
            }
        }
        return x;
    }
    // value of group lane `src` (0..11)
    __device__ __forceinline__ u64 from_lane(u64 x, u32 src) {
        u64* const xb = xbuf + flip * 64;
        flip ^= 1;
        xb[lane] = x;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return xb[gbase + src];
    }
};

// grid: chain kind major (0 = memory queue first: the long pole), then instances; one group per (kind, instance)
__global__ __launch_bounds__(256) void k_vm_chains(SeedDev a) {
    __shared__ u64 xbuf_all[4 * 128];
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ChainCtx cx;
    cx.xbuf = xbuf_all + wave * 128; cx.lane = lane; cx.gbase = lane & ~15u; cx.e = lane & 15; cx.flip = 0;
    const u32 e = cx.e;
    const u64 group = ((u64)blockIdx.x * 4 + wave) * CH_GROUPS + (lane >> 4);
    const u64 total_groups = (u64)a.n_instances * 4;
    const bool live = group < total_groups;
    const u32 kind = live ? (u32)(group / a.n_instances) : 0;
    const u32 inst = live ? (u32)(group % a.n_instances) : 0;
    const uint4 tot = a.totals[inst];
    u32 n = kind == 0 ? tot.x : kind == 1 ? tot.y : kind == 2 ? tot.z : tot.w;
    if (!live) n = 0;
    // the wavefront walks max(n) events; groups that are done keep computing on their last state and store nothing
    u32 n_max = n;
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) n_max = max(n_max, (u32)__shfl_xor((int)n_max, off));
    const u32 first_word = kind == 0 ? vmn::SW_MEM_TAIL : kind == 1 ? vmn::SW_DEC_TAIL : kind == 2 ? vmn::SW_FWD_TAIL : vmn::SW_SPONGE;
    const u32 width = kind == 2 ? 4 : 12;
    u64 x = (e < width) ? outer_value(a, inst, a.state0_slot[first_word + e]) : 0;
    const u64* ev = kind == 0 ? a.mem_ev + (u64)inst * a.cap_mem * EV_MEM : kind == 1 ? a.dec_ev + (u64)inst * a.cap_one * EV_DEC
                  : kind == 2 ? a.fwd_ev + (u64)inst * a.cap_one * EV_FWD : a.sp_ev + (u64)inst * a.cap_one * EV_SP;
    u64* snap = kind == 0 ? a.mem_snap + (u64)inst * a.cap_mem * 12 : kind == 1 ? a.dec_snap + (u64)inst * a.cap_one * 12
              : kind == 2 ? a.fwd_snap + (u64)inst * a.cap_one * 4 : a.sp_snap + (u64)inst * a.cap_one * 12;
    const u32 ev_words = kind == 0 ? EV_MEM : kind == 1 ? EV_DEC : kind == 2 ? EV_FWD : EV_SP;
    // the number of permutations per event differs by kind (1, 1, 3, 4): every group runs the wavefront's maximum and keeps what it needs
    u32 kmax = kind;
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) kmax = max(kmax, (u32)__shfl_xor((int)kmax, off));
    const u32 rounds = kmax >= 3 ? 4 : kmax == 2 ? 3 : 1;
    for (u32 k = 0; k < n_max; ++k) {
        const bool on = k < n;
        const u64* p = ev + (u64)(on ? k : 0) * ev_words;
        u32 type = 1;
        if (kind == 2) type = (u32)p[FWD_TYPE];
        if (kind == 3) type = (u32)p[SP_TYPE];
        const bool is_set = on && type == 2, is_push = on && type != 2;
        u64 s = x;       // working sponge state
        u64 tail4 = 0;   // FWD: previous tail moved to lanes 4..7 for the third absorb
        if (kind == 2) tail4 = cx.from_lane(x, e >= 4 && e < 8 ? e - 4 : 0);
        for (u32 r = 0; r < rounds; ++r) {
            if (kind <= 1) { if (r == 0 && e < 8) s = p[e]; }
            else if (kind == 2) {
                if (r == 0) s = e < 8 ? p[e] : 0;                                   // empty state, no length specialisation (log.rs:510-511)
                else if (r == 1) { if (e < 8) s = p[8 + e]; }
                else if (r == 2) { if (e < 4) s = p[16 + e]; else if (e < 8) s = tail4; }
            } else { if (e < 8) s = p[8 * r + e]; }
            const bool active_round = kind <= 1 ? r == 0 : kind == 2 ? r < 3 : true;
            const u64 t = cx.permute(s);
            if (active_round) s = t;
        }
        if (is_push) x = s;
        if (is_set) x = e < width ? p[e] : 0;
        if (on && e < width) snap[(u64)k * width + e] = x;
    }
}

// ---- phase C: the chain words of every cycle's input state
__global__ __launch_bounds__(256) void k_vm_fill(SeedDev a) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 total = (u64)a.n_instances * a.limit;
    if (i >= total) return;
    const u32 inst = (u32)(i / a.limit), c = (u32)(i % a.limit);
    if (c == 0) return;  // written by phase A
    const uint4 n = a.counts[i];
    u64* const out = a.loop + i;
    auto put = [&](u32 first_word, u32 width, u32 count, const u64* snap, u32 cap) {
        for (u32 e = 0; e < width; ++e) {
            const u64 v = count ? snap[((u64)inst * cap + (count - 1)) * width + e] : outer_value(a, inst, a.state0_slot[first_word + e]);
            out[(u64)(first_word + e) * a.in_stride] = v;
        }
    };
    put(vmn::SW_MEM_TAIL, 12, n.x, a.mem_snap, a.cap_mem);
    put(vmn::SW_DEC_TAIL, 12, n.y, a.dec_snap, a.cap_one);
    put(vmn::SW_FWD_TAIL, 4, n.z, a.fwd_snap, a.cap_one);
    put(vmn::SW_SPONGE, 12, n.w, a.sp_snap, a.cap_one);
}

}  // namespace zkvm
