// kernels_vm_seed.hpp — chain-specialised seeding of main_vm's carried state (the sequential half of witness resolution).
//
// The reference resolves `state = vm_cycle(state, ..)` (src/main_vm/mod.rs:102-110) cycle after cycle on one core.  Round 2 ran the
// recorded cone of the carried outputs through an interpreter, one wavefront per instance: 330 us per cycle.  What is sequential
// in a cycle is much less than its cone:
//   phase A  k_vm_walk    the non-hash VmLocalState (registers, flags, pc, callstack scalars ...): a few hundred integer operations
//                         per cycle once the opcode is decoded natively (vm_native.hpp) — one WAVEFRONT per instance: lane 0 walks with
//                         the state in LDS, all 64 lanes prefetch the next cycle's oracle words and write the state words.  It also
//                         lists, per instance, what each of the four Poseidon2 chains absorbs (memory queue: src/main_vm/utils.rs:194-213,
//                         cycle.rs:846-884, uma.rs:706-726; decommit queue: far_call.rs:1418-1603; forward log queue: log.rs:508-609;
//                         callstack sponge: call_ret.rs:170-270) and how many events precede every cycle;
//   phase B  k_vm_chains  every chain is independent of the others once phase A fixed what is absorbed: one DPP row (12 of 16 lanes) owns
//                         one chain, one state element per lane — S-boxes in parallel, linear layers by quad_perm / row_ror moves — and
//                         walks its events; the state after every event goes to a snapshot array;
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
    vmn::Defs D;  // small fields by value (scalar loads of the kernel argument), the two 2048-row tables in device memory
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
    u32 n_loop_words;           // words per cycle of the loop stream (243 state words, then the oracle words)
    // the pass is cut into chunks of cycles so that the chains of chunk j run (second stream) while the walker is in chunk j + 1:
    u32 c0, c1, chunk;          // this launch: cycles [c0, c1), chunk index
    uint4* chunk_totals;        // [chunk][inst]: events up to the end of the chunk
    u32* saved_state;           // [inst][SAVE_WORDS]: the walker's State between chunks
};
constexpr u32 SAVE_WORDS = (sizeof(vmn::State) + 3) / 4;

__device__ __forceinline__ u64 outer_value(const SeedDev& a, u32 inst, u32 slot) {
    return a.outer_store[zkgeom::offset(a.outer_n_store, slot, inst)];
}

constexpr u32 RAW_MAX = 128;  // oracle words per cycle the walker stages in LDS (main_vm: 117)
// the decode table (2048 rows in the blob, ~1100 defined) staged in LDS: its row is the one load on every cycle's dependency chain
constexpr u32 TABLE_ROWS_LDS = 1152;
__shared__ u64 g_props[TABLE_ROWS_LDS];
__shared__ u32 g_prices[TABLE_ROWS_LDS];

// lane 0's view of the cycle: oracle answers from the LDS copy of this cycle's stream column, events to global memory
struct DevEnv {
    const u64* raw;  // LDS: word w of the loop stream at raw[w - first_raw]
    u32 first_raw;
    RawLayout lay;
    u64 *mem_ev, *dec_ev, *fwd_ev, *sp_ev;
    u32 n_mem = 0, n_dec = 0, n_fwd = 0, n_sp = 0;
    __device__ u64 word(u32 w) const { return raw[w - first_raw]; }
    __device__ void opcode_row(const vmn::Defs& D, u32 variant, u32& price, u64& props) {
        if (variant < TABLE_ROWS_LDS) { price = g_prices[variant]; props = g_props[variant]; }
        else { price = D.prices[variant]; props = D.props[variant]; }
    }
    __device__ void load256(u32 w, bool exec, vmn::U256& o) const {
        if (exec) { for (int i = 0; i < 8; ++i) o.l[i] = (u32)word(w + i); }
        else o = vmn::u256_zero();
    }
    __device__ void load4(u32 w, bool exec, u64 o[4]) const { for (int i = 0; i < 4; ++i) o[i] = exec ? word(w + i) : 0; }
    __device__ void code_word(bool exec, vmn::U256& o) { load256(lay.code_word, exec, o); }
    __device__ void src0(bool exec, vmn::U256& v, u32& is_ptr) { load256(lay.src0_value, exec, v); is_ptr = exec ? (u32)word(lay.src0_is_ptr) : 0; }
    __device__ u32 refund(bool exec) { return exec ? (u32)word(lay.refund) : 0; }
    __device__ void log_read(bool exec, vmn::U256& o) { load256(lay.log_read, exec, o); }
    __device__ void log_prev_head(bool exec, u64 o[4]) { load4(lay.log_prev_head, exec, o); }
    __device__ void near_call_tail(bool exec, u64 o[4]) { load4(lay.near_tail, exec, o); }
    __device__ void far_code_hash(bool exec, vmn::U256& o) { load256(lay.far_code_hash, exec, o); }
    __device__ u32 far_decommit_page(bool exec) { return exec ? (u32)word(lay.far_page) : 0; }
    __device__ void far_call_tail(bool exec, u64 o[4]) { load4(lay.far_tail, exec, o); }
    __device__ void ret_pop(bool, u64 ctx42[42], u64 state[12]) {
        for (int i = 0; i < 42; ++i) ctx42[i] = word(lay.ret_ctx + i);
        for (int i = 0; i < 12; ++i) state[i] = word(lay.ret_state + i);
    }
    __device__ void uma_read(int which, bool exec, vmn::U256& o) { load256(which ? lay.uma_b : lay.uma_a, exec, o); }
    __device__ void mem_push(const u64 enc[8]) { u64* p = mem_ev + (u64)n_mem * EV_MEM; for (int i = 0; i < 8; ++i) p[i] = enc[i]; ++n_mem; }
    __device__ void dec_push(const u64 enc[8]) { u64* p = dec_ev + (u64)n_dec * EV_DEC; for (int i = 0; i < 8; ++i) p[i] = enc[i]; ++n_dec; }
    __device__ void fwd_push(const u64 enc[20]) { u64* p = fwd_ev + (u64)n_fwd * EV_FWD; for (int i = 0; i < 20; ++i) p[i] = enc[i]; p[FWD_TYPE] = 1; ++n_fwd; }
    __device__ void fwd_set(const u64 v[4]) { u64* p = fwd_ev + (u64)n_fwd * EV_FWD; for (int i = 0; i < 4; ++i) p[i] = v[i]; p[FWD_TYPE] = 2; ++n_fwd; }
    __device__ void sponge_push(const u64 enc[32]) { u64* p = sp_ev + (u64)n_sp * EV_SP; for (int i = 0; i < 32; ++i) p[i] = enc[i]; p[SP_TYPE] = 1; ++n_sp; }
    __device__ void sponge_set(const u64 v[12]) { u64* p = sp_ev + (u64)n_sp * EV_SP; for (int i = 0; i < 12; ++i) p[i] = v[i]; p[SP_TYPE] = 2; ++n_sp; }
};

// ---- phase A: NI instances per wavefront (1, 2 or 4).  One lane per instance walks the VM (its state in LDS: registers are picked by
// dynamic index); the 64 / NI lanes of its group move its data: the next cycle's oracle words are fetched into LDS while the walker
// works on this one (no global load sits on the cycle's dependency chain except the opcode-table row), and the 243 state words of a
// cycle leave as coalesced-by-word stores of the group instead of 203 stores of one lane.  A lone walking lane owns its SIMD's issue
// slots: two walkers in one wavefront share every instruction both of them need (decode, operand fetch, flags, state update) and
// serialise only where their opcodes differ.
template <int NI>
__global__ __launch_bounds__(64) void k_vm_walk(SeedDev a) {
    constexpr u32 G = 64 / NI;                      // lanes per instance
    constexpr u32 RAW_PER_LANE = (RAW_MAX + G - 1) / G;
    __shared__ vmn::State st_[NI];
    __shared__ vmn::Gctx gctx_[NI];
    __shared__ u64 flat_[NI][256];
    __shared__ u64 raw_[NI][2][RAW_MAX];
    const u32 lane = threadIdx.x, sub = lane / G, sl = lane % G;
    const u32 inst = blockIdx.x * NI + sub;
    const bool live = inst < a.n_instances;
    const u64 lane0 = (u64)(live ? inst : 0) * a.limit;
    vmn::State& st = st_[sub];
    vmn::Gctx& gctx = gctx_[sub];
    u64* const flat = flat_[sub];
    const u32 first_raw = vmn::STATE_WORDS, n_raw = a.n_loop_words - vmn::STATE_WORDS;
    if (live) {
        if (a.c0 == 0) {
            // cycle 0 takes the outer scope's words verbatim (chain words included: they are snapshot 0 of every chain)
            for (u32 w = sl; w < (u32)vmn::STATE_WORDS; w += G) {
                const u64 v = outer_value(a, inst, a.state0_slot[w]);
                flat[w] = v;
                a.loop[(u64)w * a.in_stride + lane0] = v;
            }
        } else {
            for (u32 w = sl; w < SAVE_WORDS; w += G) ((u32*)&st)[w] = a.saved_state[(u64)inst * SAVE_WORDS + w];
        }
    }
    for (u32 i = lane; i < TABLE_ROWS_LDS; i += 64) { g_props[i] = a.D.props[i]; g_prices[i] = a.D.prices[i]; }
    auto fetch = [&](u32 c, u32 k) -> u64 { return (live && k < n_raw) ? a.loop[(u64)(first_raw + k) * a.in_stride + lane0 + c] : 0; };
#pragma unroll
    for (u32 r = 0; r < RAW_PER_LANE; ++r)
        if (sl + G * r < RAW_MAX) raw_[sub][a.c0 & 1][sl + G * r] = fetch(a.c0, sl + G * r);
    __syncthreads();
    const vmn::Defs& D = a.D;  // kernel argument: its small fields are scalar loads
    DevEnv env;
    const bool walker = sl == 0 && live;
    if (walker) {
        gctx.zkporter_is_available = (u32)a.outer_inputs[(u64)a.w_zkporter * a.outer_in_stride + inst];
        for (int i = 0; i < 8; ++i) gctx.default_aa_code_hash.l[i] = (u32)a.outer_inputs[(u64)(a.w_default_aa + i) * a.outer_in_stride + inst];
        if (a.c0 == 0) vmn::state_unflatten(st, [&](int w) { return flat[w]; });
        else { const uint4 t = a.chunk_totals[(u64)(a.chunk - 1) * a.n_instances + inst]; env.n_mem = t.x; env.n_dec = t.y; env.n_fwd = t.z; env.n_sp = t.w; }
        env.first_raw = first_raw; env.lay = a.raw;
        env.mem_ev = a.mem_ev + (u64)inst * a.cap_mem * EV_MEM;
        env.dec_ev = a.dec_ev + (u64)inst * a.cap_one * EV_DEC;
        env.fwd_ev = a.fwd_ev + (u64)inst * a.cap_one * EV_FWD;
        env.sp_ev = a.sp_ev + (u64)inst * a.cap_one * EV_SP;
    }
#ifdef ZKGL_VM_WALK_PROFILE
    static_assert(NI == 1 || NI == 2 || NI == 4, "");
    u64 t_walk = 0, t_flat = 0, t_io = 0, t_fam[16] = {0}, n_fam[16] = {0};
#endif
    for (u32 c = a.c0; c < a.c1; ++c) {
        const bool more = c + 1 < a.limit, more_here = c + 1 < a.c1;
        u64 p[RAW_PER_LANE];
#pragma unroll
        for (u32 r = 0; r < RAW_PER_LANE; ++r) p[r] = more_here ? fetch(c + 1, sl + G * r) : 0;
#ifdef ZKGL_VM_WALK_PROFILE
        const u64 t0 = wall_clock64();
#endif
        if (walker) {
            a.counts[lane0 + c] = make_uint4(env.n_mem, env.n_dec, env.n_fwd, env.n_sp);
            env.raw = raw_[sub][c & 1];
            vmn::vm_cycle(D, gctx, st, env);
#ifdef ZKGL_VM_WALK_PROFILE
            const u64 t1 = wall_clock64();
            t_walk += t1 - t0;
            { const u32 f = st.last_family & 15; t_fam[f] += t1 - t0; n_fam[f] += 1; }
#endif
        }
#ifdef ZKGL_VM_WALK_PROFILE
        const u64 t2 = wall_clock64();
#endif
        __syncthreads();
        if (more && live) {
            u64* const out = a.loop + lane0 + c + 1;
            for (u32 w = sl; w < (u32)vmn::STATE_WORDS; w += G) out[(u64)w * a.in_stride] = vmn::state_word(st, (int)w);  // chain words: phase C writes them
#pragma unroll
            for (u32 r = 0; r < RAW_PER_LANE; ++r)
                if (sl + G * r < RAW_MAX) raw_[sub][(c + 1) & 1][sl + G * r] = p[r];
        }
        __syncthreads();
#ifdef ZKGL_VM_WALK_PROFILE
        t_io += wall_clock64() - t2;
#endif
    }
#ifdef ZKGL_VM_WALK_PROFILE
    if (walker && inst < 8) {   // 100 MHz constant clock: 10 ns units
        printf("[walk profile] inst %u: walk %llu flatten %llu io+sync %llu (x10ns) per family (count, x10ns):", inst, (unsigned long long)t_walk, (unsigned long long)t_flat, (unsigned long long)t_io);
        for (int f = 0; f < 16; ++f) printf(" %d:(%llu,%llu)", f, (unsigned long long)n_fam[f], (unsigned long long)t_fam[f]);
        printf("\n");
    }
#endif
    if (walker) {
        const uint4 t = make_uint4(env.n_mem, env.n_dec, env.n_fwd, env.n_sp);
        a.chunk_totals[(u64)a.chunk * a.n_instances + inst] = t;
        if (a.c1 == a.limit) a.totals[inst] = t;
    }
    if (a.c1 < a.limit && live)
        for (u32 w = sl; w < SAVE_WORDS; w += G) a.saved_state[(u64)inst * SAVE_WORDS + w] = ((const u32*)&st)[w];
}

// ---- phase B: 16-lane rows (one DPP row each), 12 lanes = the 12 state elements of one chain, lanes 12..15 hold zero.
// Everything that crosses lanes is a DPP move inside the row — no LDS, no waits:
//   inner layer   out_e = S + (x_e << k_e), S = sum of the row: rotate-and-add all-reduce (ror 8, 4, 2, 1) on 96-bit sums;
//   outer layer   lane e = 4 b + r: the quad's four values by quad_perm broadcasts, y_b = M4 x_b through the 8-addition chain (every
//                 lane keeps row r), Y = sum of y over the quads (ror 8, 4), out = y + Y   (M_E = circ(2 M4, M4, M4));
//   S-box         products reduced lazily (any u64 representative), one canonical reduction per layer output.
constexpr u32 CH_GROUPS = 4;   // per wavefront
namespace dpp {
template <int CTRL> __device__ __forceinline__ u32 mov32(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ u64 mov64(u64 v) { return ((u64)mov32<CTRL>((u32)(v >> 32)) << 32) | mov32<CTRL>((u32)v); }
template <int CTRL> __device__ __forceinline__ p2::W movW(p2::W v) { p2::W r; r.lo = mov64<CTRL>(v.lo); r.hi = mov32<CTRL>(v.hi); return r; }
constexpr int ROR(int n) { return 0x120 | n; }            // lane i <- lane (i - n) mod 16
constexpr int SHR(int n) { return 0x110 | n; }            // lane i <- lane i - n (lanes < n: 0)
constexpr int QUAD(int c) { return c | (c << 2) | (c << 4) | (c << 6); }  // every lane of a quad <- lane c of the quad
}  // namespace dpp

// 128 -> 64 without the final conditional subtraction: any representative < 2^64
__device__ __forceinline__ u64 reduce128_lazy(u64 lo, u64 hi) {
    const u64 hi_hi = hi >> 32, hi_lo = hi & gl::EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= gl::EPS;
    const u64 t1 = (hi_lo << 32) - hi_lo;
    u64 t2 = t0 + t1;
    if (t2 < t1) t2 += gl::EPS;
    return t2;
}
__device__ __forceinline__ u64 mul_lazy(u64 a, u64 b) { u64 lo, hi; gl::mul_wide(a, b, lo, hi); return reduce128_lazy(lo, hi); }
// (x + rc)^7, x and rc canonical; the result is some representative < 2^64
__device__ __forceinline__ u64 sbox_lazy(u64 x, u64 rc) {
    u64 t = x + rc;
    if (t < x) t += gl::EPS;   // x + rc < 2p: the wrapped sum + EPS cannot wrap again
    const u64 t2 = mul_lazy(t, t), t3 = mul_lazy(t2, t), t4 = mul_lazy(t2, t2);
    return mul_lazy(t3, t4);
}
__device__ __forceinline__ u64 row_mds_inner(u64 x, u32 e) {
    p2::W s{x, 0};
    s = p2::wadd(s, dpp::movW<dpp::ROR(8)>(s));
    s = p2::wadd(s, dpp::movW<dpp::ROR(4)>(s));
    s = p2::wadd(s, dpp::movW<dpp::ROR(2)>(s));
    s = p2::wadd(s, dpp::movW<dpp::ROR(1)>(s));
    u32 k = 4;  // INNER_SHIFT = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12}
#pragma unroll
    for (int i = 1; i < 12; ++i) k = e == (u32)i ? (u32)p2::INNER_SHIFT[i] : k;
    const p2::W t{x << k, k ? (u32)(x >> (64 - k)) : 0u};
    const p2::W r = p2::wadd(s, t);
    return e < 12 ? gl::reduce96(r.lo, r.hi) : 0;   // lanes 12..15 stay zero: they take part in every row sum
}
__device__ __forceinline__ u64 row_mds_external(u64 x, u32 e) {
    p2::W w0{dpp::mov64<dpp::QUAD(0)>(x), 0}, w1{dpp::mov64<dpp::QUAD(1)>(x), 0}, w2{dpp::mov64<dpp::QUAD(2)>(x), 0}, w3{dpp::mov64<dpp::QUAD(3)>(x), 0};
    p2::m4w(w0, w1, w2, w3);
    const u32 r = e & 3;
    p2::W y = r == 0 ? w0 : r == 1 ? w1 : r == 2 ? w2 : w3;
    p2::W t = p2::wadd(y, dpp::movW<dpp::ROR(8)>(y));
    t = p2::wadd(t, dpp::movW<dpp::ROR(4)>(t));
    const p2::W o = p2::wadd(y, t);
    return e < 12 ? gl::reduce96(o.lo, o.hi) : 0;
}
// one permutation of the row's state; rcf[8] = this lane's constants of the 8 full rounds (lanes 12..15: anything, they are re-zeroed)
__device__ __forceinline__ u64 row_permute(u64 x, u32 e, const u64 rcf[8]) {
    const bool live = e < 12;
    x = row_mds_external(x, e);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const u64 sb = sbox_lazy(x, rcf[r]);
        x = row_mds_external(live ? sb : 0, e);
    }
    u64 rc = p2::RC[12 * 4];
#pragma unroll 1
    for (int rr = 4; rr < 26; ++rr) {
        const u64 rc_next = p2::RC[12 * (rr + 1 < 26 ? rr + 1 : 4)];  // uniform: a scalar load, a round ahead of its use
        const u64 sb = sbox_lazy(x, rc);
        const u64 canon = sb >= gl::P ? sb - gl::P : sb;   // element 0 enters the shift term: keep it canonical
        x = row_mds_inner(e == 0 ? canon : x, e);
        rc = rc_next;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const u64 sb = sbox_lazy(x, rcf[4 + r]);
        x = row_mds_external(live ? sb : 0, e);
    }
    return x;
}

// grid: chain kind major (0 = memory queue first: the long pole), then instances; one row per (kind, instance)
__global__ __launch_bounds__(256) void k_vm_chains(SeedDev a) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 e = lane & 15;
    const u32 el = e < 12 ? e : 0;
    const u64 group = ((u64)blockIdx.x * 4 + wave) * CH_GROUPS + (lane >> 4);
    const u64 total_groups = (u64)a.n_instances * 4;
    const bool live = group < total_groups;
    const u32 kind = live ? (u32)(group / a.n_instances) : 0;
    const u32 inst = live ? (u32)(group % a.n_instances) : 0;
    const uint4 tot = a.chunk_totals[(u64)a.chunk * a.n_instances + inst];
    uint4 tot0 = make_uint4(0, 0, 0, 0);
    if (a.chunk) tot0 = a.chunk_totals[(u64)(a.chunk - 1) * a.n_instances + inst];
    const u32 k0 = kind == 0 ? tot0.x : kind == 1 ? tot0.y : kind == 2 ? tot0.z : tot0.w;   // events [k0, k0 + n) belong to this chunk
    u32 n = (kind == 0 ? tot.x : kind == 1 ? tot.y : kind == 2 ? tot.z : tot.w) - k0;
    if (!live) n = 0;
    // the wavefront walks max(n) events; rows that are done keep computing and store nothing
    u32 n_max = n;
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) n_max = max(n_max, (u32)__shfl_xor((int)n_max, off));
    const u32 first_word = kind == 0 ? vmn::SW_MEM_TAIL : kind == 1 ? vmn::SW_DEC_TAIL : kind == 2 ? vmn::SW_FWD_TAIL : vmn::SW_SPONGE;
    const u32 width = kind == 2 ? 4 : 12;
    u64 x = 0;
    const u64* ev = kind == 0 ? a.mem_ev + (u64)inst * a.cap_mem * EV_MEM : kind == 1 ? a.dec_ev + (u64)inst * a.cap_one * EV_DEC
                  : kind == 2 ? a.fwd_ev + (u64)inst * a.cap_one * EV_FWD : a.sp_ev + (u64)inst * a.cap_one * EV_SP;
    u64* snap = kind == 0 ? a.mem_snap + (u64)inst * a.cap_mem * 12 : kind == 1 ? a.dec_snap + (u64)inst * a.cap_one * 12
              : kind == 2 ? a.fwd_snap + (u64)inst * a.cap_one * 4 : a.sp_snap + (u64)inst * a.cap_one * 12;
    const u32 ev_words = kind == 0 ? EV_MEM : kind == 1 ? EV_DEC : kind == 2 ? EV_FWD : EV_SP;
    if (e < width) x = k0 ? snap[(u64)(k0 - 1) * width + e] : outer_value(a, inst, a.state0_slot[first_word + e]);   // the chain's state where the chunk starts
    ev += (u64)k0 * ev_words;
    snap += (u64)k0 * width;
    // permutations per event by kind: 1, 1, 3, 4 — every row runs the wavefront's maximum and keeps what it needs
    u32 kmax = kind;
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) kmax = max(kmax, (u32)__shfl_xor((int)kmax, off));
    const u32 rounds = kmax >= 3 ? 4 : kmax == 2 ? 3 : 1;
    u64 rcf[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { rcf[r] = p2::RC[12 * r + el]; rcf[4 + r] = p2::RC[12 * (26 + r) + el]; }
    u64 pay = (n && e < 8) ? ev[e] : 0;   // first absorb of event 0
    for (u32 k = 0; k < n_max; ++k) {
        const bool on = k < n;
        const u64* p = ev + (u64)(on ? k : 0) * ev_words;
        u32 type = 1;
        if (kind == 2) type = (u32)p[FWD_TYPE];
        if (kind == 3) type = (u32)p[SP_TYPE];
        const bool is_set = on && type == 2, is_push = on && type != 2;
        // next event's first absorb: fetched while this one is hashed
        const u64 pay_next = (k + 1 < n && e < 8) ? ev[(u64)(k + 1) * ev_words + e] : 0;
        u64 s = x;                                             // working sponge state
        const u64 tail4 = dpp::mov64<dpp::SHR(4)>(x);          // FWD: previous tail on lanes 4..7 for the third absorb
        for (u32 r = 0; r < rounds; ++r) {
            if (kind <= 1) { if (r == 0 && e < 8) s = pay; }
            else if (kind == 2) {
                if (r == 0) s = e < 8 ? pay : 0;                                  // empty state, no length specialisation (log.rs:510-511)
                else if (r == 1) { if (e < 8) s = p[8 + e]; }
                else if (r == 2) { if (e < 4) s = p[16 + e]; else if (e < 8) s = tail4; }
            } else { if (e < 8) s = r == 0 ? pay : p[8 * r + e]; }
            const bool active_round = kind <= 1 ? r == 0 : kind == 2 ? r < 3 : true;
            const u64 t = row_permute(e < 12 ? s : 0, e, rcf);
            if (active_round) s = t;
        }
        if (is_push) x = e < width ? s : 0;
        if (is_set) x = e < width ? p[e] : 0;
        if (on && e < width) snap[(u64)k * width + e] = x;
        pay = pay_next;
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
