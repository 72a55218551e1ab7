// kernels_fsm_seed.hpp — native seeding of the precompile FSM circuits' carried state: keccak256_round_function, sha256_round_function.
//
// What is sequential in these circuits (reference src/keccak256_round_function/mod.rs:155-670, src/sha256_round_function/mod.rs:88-340)
// is a small state machine per cycle — flags, call parameters, the byte buffer, the hash state — plus two Poseidon2 chains: the
// request queue's head (one pop per precompile call, three permutations) and the memory queue's tail (one permutation per query).
// The recorded cone of that state is ~77 k interpreted byte-table ops per Keccak cycle; natively it is one Keccak-f / SHA-256
// compression and a few hundred integer ops.
//
// One workgroup of two wavefronts per instance, the state in LDS (scalars as words, the hash state and the byte buffer packed):
//   walker wavefront   all lanes write the cycle's carried words from the state; lane 0 steps the state machine in place and lists
//                      the queue events of the cycle (the chains never feed back into the machine)
//   hasher wavefront   stages the next cycle's raw words (the call, the memory read values), runs the previous cycle's events through
//                      the DPP-row Poseidon2 of kernels_vm_seed.hpp (16 lanes, 12 live) and writes the chain words of the cycle
// one workgroup barrier per cycle; the cycle costs max(walker, hasher) instead of their sum.
// Same words as the cone kernels (tests/test_gpu_cs.py, tests/config_timings.py: == native restatement).
#pragma once
#include "kernels_engine.hpp"
#include "kernels_vm_seed.hpp"

namespace zkf {

using vmn::u32;
using vmn::u64;

struct FsmSeedDev {
    u64* loop; u64 in_stride; u32 limit, n_instances;
    const u64* outer_store; u64 outer_n_store;
    const u32* state0_slot;   // [n carried] outer store slot behind the FIRST link of every carried word
    u32 debug;                // timing experiments (ZKGL_FSM_SEED_DEBUG): 1 = no hashing, 2 = no walking
};

constexpr u32 EV_MAX = 8;
struct Events { u32 n; u32 kind[EV_MAX]; u64 pay[EV_MAX][20]; };   // kind 0: memory queue push (8 words), 1: request queue pop (20 words)

__device__ __forceinline__ u64 ov(const FsmSeedDev& a, u32 inst, u32 slot) {
    return a.outer_store[zkgeom::offset(a.outer_n_store, slot, inst)];
}

// wavefront-scope fence + scheduling barrier: orders one wavefront's LDS traffic between its lanes (its DS instructions execute in order)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// chain[0..12) = memory queue tail, chain[12..16) = request queue head; the events of one cycle, all lanes of the hasher wavefront
__device__ __forceinline__ void run_events(u64* chain, const Events& ev, u32 lane, const u64 rcf[8]) {
    const u32 e = lane & 15;
    const bool row0 = lane < 16;
    const u32 n = ev.n;
    u64 x = e < 12 ? chain[e] : 0;                                    // memory queue tail on lanes 0..11
    u64 h = (e >= 4 && e < 8) ? chain[12 + e - 4] : 0;               // request queue head on lanes 4..7 (where the third absorb wants it)
    for (u32 k = 0; k < n; ++k) {
        if (ev.kind[k] == 0) {   // full-state queue: the encoding replaces the first 8 elements of the tail, one permutation
            x = zkvm::row_permute(e < 8 ? ev.pay[k][e] : x, e, rcf);
        } else {                  // 4-wide tail, 20-word encoding: sponge over encoding || old tail from the empty state (log_query: 3 absorbs)
            u64 s = e < 8 ? ev.pay[k][e] : 0;
            s = zkvm::row_permute(s, e, rcf);
            if (e < 8) s = ev.pay[k][8 + e];
            s = zkvm::row_permute(e < 12 ? s : 0, e, rcf);
            if (e < 4) s = ev.pay[k][16 + e];
            else if (e < 8) s = h;
            s = zkvm::row_permute(e < 12 ? s : 0, e, rcf);
            h = zkvm::dpp::mov64<zkvm::dpp::SHR(4)>(s);              // new head = elements 0..3, kept on lanes 4..7
        }
    }
    wave_sync();   // all 64 lanes have read chain[] above (with no event there is nothing else between that read and these writes)
    if (row0 && e < 12) chain[e] = x;
    if (row0 && e >= 4 && e < 8) chain[12 + e - 4] = h;
    wave_sync();   // the callers read chain[12..15] on lanes 12..15, written here by lanes 4..7 (tests/emu race detector)
}

__device__ __forceinline__ void load_call(vmn::LogQ& q, const u64* raw) {   // flattened LogQuery, 36 words (log_query/mod.rs:60-99)
    for (int i = 0; i < 5; ++i) q.address[i] = (u32)raw[i];
    for (int i = 0; i < 8; ++i) { q.key.l[i] = (u32)raw[5 + i]; q.read_value.l[i] = (u32)raw[13 + i]; q.written_value.l[i] = (u32)raw[21 + i]; }
    q.aux_byte = (u32)raw[29]; q.rw_flag = (u32)raw[30]; q.rollback = (u32)raw[31]; q.is_service = (u32)raw[32];
    q.shard_id = (u32)raw[33]; q.tx_number = (u32)raw[34]; q.timestamp = (u32)raw[35];
}
__device__ __forceinline__ void push_mem(Events& ev, u32 ts, u32 page, u32 index, u32 rw, const vmn::U256& v) {
    const u32 k = ev.n++;
    ev.kind[k] = 0;
    vmn::memory_query_encode(ev.pay[k], ts, page, index, rw, 0, v);
}
__device__ __forceinline__ void pop_request(Events& ev, const vmn::LogQ& q) {
    const u32 k = ev.n++;
    ev.kind[k] = 1;
    vmn::log_query_encode(ev.pay[k], q);
}

__constant__ const u64 KECCAK_RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
// Keccak-f[1600] on 25 register-resident lanes A[x + 5y]: the round body unrolled (constant rotations and indices), 24 trips
__device__ __forceinline__ void keccak_f(u64 (&A)[25]) {
    constexpr int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll 1
    for (int r = 0; r < 24; ++r) {
        u64 C[5], B[25];
#pragma unroll
        for (int x = 0; x < 5; ++x) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            const u64 c1 = C[(x + 1) % 5], d = C[(x + 4) % 5] ^ ((c1 << 1) | (c1 >> 63));
#pragma unroll
            for (int y = 0; y < 25; y += 5) A[x + y] ^= d;
        }
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y) {
                const u64 v = A[x + 5 * y];
                const int n = ROT[x + 5 * y];
                B[y + 5 * ((2 * x + 3 * y) % 5)] = n ? ((v << n) | (v >> ((64 - n) & 63))) : v;
            }
#pragma unroll
        for (int y = 0; y < 25; y += 5)
#pragma unroll
            for (int x = 0; x < 5; ++x) A[x + y] = B[x + y] ^ (~B[(x + 1) % 5 + y] & B[(x + 2) % 5 + y]);
        A[0] ^= KECCAK_RC[r];
    }
}

// Keccak-f[1600] on 25 lanes of one wavefront (lane = x + 5y holds A[x + 5y]), the cross-lane steps through LDS: a third of the single
// lane's latency (14 k instructions there).  sh: 25 + 5 + 25 words of this wavefront's own.  Wavefront-scope fences order the LDS
// traffic (one wavefront's DS instructions execute in order).
__device__ __forceinline__ u64 keccak_f_lanes(u64 a, u32 lane, u64* sh) {
    constexpr int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    u64* const shA = sh, * const shC = sh + 25, * const shB = sh + 30;
    const bool live = lane < 25;
    const u32 l = live ? lane : 0, x = l % 5, y = l / 5;
    u32 rot = 0;
#pragma unroll
    for (int i = 1; i < 25; ++i) rot = l == (u32)i ? (u32)ROT[i] : rot;
    const u32 dest = y + 5 * ((2 * x + 3 * y) % 5);           // rho + pi: B[y, 2x + 3y] = rot(A[x, y])
    const u32 row = 5 * y;
#pragma unroll 1
    for (int r = 0; r < 24; ++r) {
        if (live) shA[l] = a;
        wave_sync();
        if (lane < 5) shC[lane] = shA[lane] ^ shA[lane + 5] ^ shA[lane + 10] ^ shA[lane + 15] ^ shA[lane + 20];
        wave_sync();
        const u64 c1 = shC[(x + 1) % 5];
        a ^= shC[(x + 4) % 5] ^ ((c1 << 1) | (c1 >> 63));
        const u64 rr = rot ? ((a << rot) | (a >> (64 - rot))) : a;
        if (live) shB[dest] = rr;
        wave_sync();
        a = shB[row + x] ^ (~shB[row + (x + 1) % 5] & shB[row + (x + 2) % 5]);
        if (lane == 0) a ^= KECCAK_RC[r];
        wave_sync();
    }
    return a;
}

__constant__ const u32 SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
// SHA-256 compression, the message schedule as a 16-register ring, four trips of 16 unrolled rounds
__device__ __forceinline__ void sha256_compress(u32 (&st)[8], u32 (&w)[16]) {
    auto rotr = [](u32 x, int n) { return (x >> n) | (x << (32 - n)); };
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll 1
    for (int t = 0; t < 64; t += 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (t) {
                const u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
                w[i] += (rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3)) + w[(i + 9) & 15] + (rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10));
            }
            const u32 t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[t + i] + w[i];
            const u32 t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// ------------------------------------------------------------------------------------------------ keccak256_round_function
struct Keccak {
    static constexpr u32 RPC = 0, RUW = 1, PAD = 2, DONE = 3, STATE = 4, TS_READ = 204, TS_WRITE = 205, INPUT_PAGE = 206, BYTE_OFFSET = 207, BYTE_LENGTH = 208,
                         OUTPUT_PAGE = 209, OUTPUT_OFFSET = 210, NEEDS_FULL = 211, BUFFER = 212, FILLED = 404, REQ_HEAD = 405, REQ_LEN = 409, MEM_TAIL = 410, MEM_LEN = 422,
                         CARRIED = 423, CALL = 423, VALUES = 459, LOOP_WORDS = 507;
    static constexpr u32 RATE = 136, BUF = 192, READS = 6;
    // the machine: scalar words by their carried index (array ranges unused), the sponge lanes and the byte buffer packed
    struct State { u64 w[CARRIED]; u64 A[25]; alignas(8) unsigned char buf[BUF]; };
    static __device__ __forceinline__ u64 get(const State& s, u32 w) {
        if (w >= STATE && w < STATE + 200) return ((const unsigned char*)s.A)[w - STATE];   // lane i byte k at 8 i + k: the carried order (input.rs:34)
        if (w >= BUFFER && w < BUFFER + BUF) return s.buf[w - BUFFER];
        return s.w[w];
    }
    static __device__ __forceinline__ void set(State& s, u32 w, u64 v) {
        if (w >= STATE && w < STATE + 200) ((unsigned char*)s.A)[w - STATE] = (unsigned char)v;
        else if (w >= BUFFER && w < BUFFER + BUF) s.buf[w - BUFFER] = (unsigned char)v;
        else s.w[w] = v;
    }
    // keccak256_precompile_inner, one cycle (mod.rs:215-640); raw = loop words [CALL, LOOP_WORDS)
    static __device__ __noinline__ void step(State& s, const u64* raw, Events& ev) {
        u64* cw = s.w;
        ev.n = 0;
        u32 rpc = (u32)cw[RPC], ruw = (u32)cw[RUW], padding_round = (u32)cw[PAD], completed = (u32)cw[DONE];
        u32 req_len = (u32)cw[REQ_LEN];
        u32 byte_offset = (u32)cw[BYTE_OFFSET], byte_length = (u32)cw[BYTE_LENGTH], ts_read = (u32)cw[TS_READ], ts_write = (u32)cw[TS_WRITE];
        u32 input_page = (u32)cw[INPUT_PAGE], output_page = (u32)cw[OUTPUT_PAGE], output_offset = (u32)cw[OUTPUT_OFFSET], needs_full = (u32)cw[NEEDS_FULL];
        if (rpc) {
            vmn::LogQ call;
            load_call(call, raw);
            if (req_len) {
                pop_request(ev, call);
                req_len -= 1;
            }
            const u32 call_length = call.key.l[1];
            byte_offset = call.key.l[0]; byte_length = call_length; output_offset = call.key.l[2]; input_page = call.key.l[4]; output_page = call.key.l[5];
            needs_full = call_length % RATE == 0;
            ts_read = call.timestamp;
            ts_write = ts_read + 1;
            if (call_length == 0) padding_round = 1;
            else ruw = 1;
        }
        const bool reset_buffer = rpc || completed;
        rpc = 0;
        u32 filled = (u32)cw[FILLED];
        u64 A[25];
        u64* const buf8 = (u64*)s.buf;
        if (reset_buffer) {
            for (u32 i = 0; i < BUF / 8; ++i) buf8[i] = 0;
            filled = 0;
#pragma unroll
            for (int i = 0; i < 25; ++i) A[i] = 0;
        } else {
#pragma unroll
            for (int i = 0; i < 25; ++i) A[i] = s.A[i];
        }
        u32 mem_len = (u32)cw[MEM_LEN];
        for (u32 r = 0; r < READS; ++r) {
            const u32 unalignment = byte_offset % 32, aligned = byte_offset / 32;
            const u32 at_most = 32 - unalignment;
            const u32 meaningful = byte_length < at_most ? byte_length : at_most;
            const bool should_read = meaningful != 0 && filled + meaningful <= BUF && ruw;
            if (should_read) {
                vmn::U256 v;
                for (int i = 0; i < 8; ++i) v.l[i] = (u32)raw[VALUES - CALL + 8 * r + i];
                push_mem(ev, ts_read, input_page, aligned, 0, v);
                mem_len += 1;
                byte_offset += meaningful;
                byte_length -= meaningful;
                // fill_with_bytes (buffer/mod.rs:74-135): the word's big-endian bytes from `unalignment` on, `meaningful` of them, zeros up to 32
                for (u32 idx = 0; idx < 32; ++idx) {
                    const u32 j = unalignment + idx;   // big-endian byte index
                    const u32 limb = (u32)raw[VALUES - CALL + 8 * r + 7 - ((j / 4) & 7)];
                    const u32 b = (idx < meaningful && j < 32) ? (limb >> (8 * (3 - j % 4))) & 0xff : 0;
                    if (filled + idx < BUF) s.buf[filled + idx] = (unsigned char)b;
                }
                filled += meaningful;
            }
        }
        const bool zero_bytes_left = byte_length == 0;
        const u32 currently_filled = filled;
        // consume::<136> (buffer/mod.rs:137-163)
        u64 block[17];
#pragma unroll
        for (int i = 0; i < 17; ++i) block[i] = buf8[i];
#pragma unroll
        for (int i = 0; i < 7; ++i) buf8[i] = buf8[17 + i];
#pragma unroll
        for (int i = 7; i < 24; ++i) buf8[i] = 0;
        filled = filled >= RATE ? filled - RATE : 0;
        const bool buffer_now_empty = filled == 0;
        const bool apply_padding = zero_bytes_left && buffer_now_empty && ruw && !needs_full;
        if (apply_padding) {
            const u32 j = currently_filled;
#pragma unroll
            for (int i = 0; i < 17; ++i)
                if (j < RATE - 1 && (u32)i == j / 8) block[i] = (block[i] & ~(0xffULL << (8 * (j % 8)))) | (0x01ULL << (8 * (j % 8)));
            block[16] = (block[16] & ~(0xffULL << 56)) | ((currently_filled == RATE - 1 ? 0x81ULL : 0x80ULL) << 56);
        }
        if (padding_round) {
#pragma unroll
            for (int i = 0; i < 17; ++i) block[i] = 0;
            block[0] = 0x01; block[16] = 0x80ULL << 56;
        }
#pragma unroll
        for (int i = 0; i < 17; ++i) A[i] ^= block[i];
        keccak_f(A);
        const bool write_result = apply_padding || padding_round;
        if (write_result) {
            vmn::U256 d;   // digest bytes = lanes 0..3 little-endian; the value is those 32 bytes read big-endian
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d.l[7 - 2 * i] = __builtin_bswap32((u32)A[i]);
                d.l[6 - 2 * i] = __builtin_bswap32((u32)(A[i] >> 32));
            }
            push_mem(ev, ts_write, output_page, output_offset, 1, d);
            mem_len += 1;
        }
        const bool input_is_empty = req_len == 0;
        rpc = write_result && !input_is_empty;
        completed = (write_result && input_is_empty) || completed;
        padding_round = ruw && zero_bytes_left && buffer_now_empty && needs_full;
        ruw = !(rpc || padding_round || completed);
        cw[RPC] = rpc; cw[RUW] = ruw; cw[PAD] = padding_round; cw[DONE] = completed;
#pragma unroll
        for (int i = 0; i < 25; ++i) s.A[i] = A[i];
        cw[TS_READ] = ts_read; cw[TS_WRITE] = ts_write; cw[INPUT_PAGE] = input_page; cw[BYTE_OFFSET] = byte_offset; cw[BYTE_LENGTH] = byte_length;
        cw[OUTPUT_PAGE] = output_page; cw[OUTPUT_OFFSET] = output_offset; cw[NEEDS_FULL] = needs_full;
        cw[FILLED] = filled; cw[REQ_LEN] = req_len; cw[MEM_LEN] = mem_len;
    }
};

// ------------------------------------------------------------------------------------------------ sha256_round_function
struct Sha256 {
    static constexpr u32 RPC = 0, RWFR = 1, DONE = 2, STATE = 3, TS_READ = 35, TS_WRITE = 36, INPUT_PAGE = 37, INPUT_OFFSET = 38, OUTPUT_PAGE = 39, OUTPUT_OFFSET = 40,
                         NUM_ROUNDS = 41, REQ_HEAD = 42, REQ_LEN = 46, MEM_TAIL = 47, MEM_LEN = 59, CARRIED = 60, CALL = 60, VALUES = 96, LOOP_WORDS = 112;
    struct State { u64 w[CARRIED]; u32 st[8]; };
    static __device__ __forceinline__ u64 get(const State& s, u32 w) {
        if (w >= STATE && w < STATE + 32) return ((const unsigned char*)s.st)[w - STATE];   // word i byte k (little-endian) at 4 i + k
        return s.w[w];
    }
    static __device__ __forceinline__ void set(State& s, u32 w, u64 v) {
        if (w >= STATE && w < STATE + 32) ((unsigned char*)s.st)[w - STATE] = (unsigned char)v;
        else s.w[w] = v;
    }
    // sha256_precompile_inner, one cycle (mod.rs:139-340)
    static __device__ __noinline__ void step(State& s, const u64* raw, Events& ev) {
        u64* cw = s.w;
        ev.n = 0;
        u32 rpc = (u32)cw[RPC], rwfr = (u32)cw[RWFR], completed = (u32)cw[DONE];
        u32 req_len = (u32)cw[REQ_LEN];
        u32 input_page = (u32)cw[INPUT_PAGE], input_offset = (u32)cw[INPUT_OFFSET], output_page = (u32)cw[OUTPUT_PAGE], output_offset = (u32)cw[OUTPUT_OFFSET];
        u64 num_rounds = cw[NUM_ROUNDS];
        u32 ts_read = (u32)cw[TS_READ], ts_write = (u32)cw[TS_WRITE];
        if (rpc) {
            vmn::LogQ call;
            load_call(call, raw);
            if (req_len) {
                pop_request(ev, call);
                req_len -= 1;
            }
            input_offset = call.key.l[0]; output_offset = call.key.l[2]; input_page = call.key.l[4]; output_page = call.key.l[5]; num_rounds = call.key.l[6];
            ts_read = call.timestamp;
            ts_write = ts_read + 1;
        }
        const bool reset_buffer = rpc || completed;
        rwfr = rpc || rwfr;
        rpc = 0;
        const bool should_read = num_rounds != 0;
        u32 mem_len = (u32)cw[MEM_LEN];
        u32 blk[16];
#pragma unroll
        for (u32 r = 0; r < 2; ++r) {
            vmn::U256 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v.l[i] = (u32)raw[VALUES - CALL + 8 * r + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) blk[8 * r + i] = v.l[7 - i];
            if (should_read) {
                push_mem(ev, ts_read, input_page, input_offset, 0, v);
                mem_len += 1;
            }
            if (rwfr) input_offset += 1;
        }
        if (rwfr) num_rounds = gl::sub(num_rounds, 1);
        u32 st[8];
        const u32 iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = reset_buffer ? iv[i] : s.st[i];
        sha256_compress(st, blk);
        const bool write_result = rwfr && num_rounds == 0;
        if (write_result) {
            vmn::U256 d;   // state words big-endian, concatenated, read as one big-endian number
#pragma unroll
            for (int i = 0; i < 8; ++i) d.l[7 - i] = st[i];
            push_mem(ev, ts_write, output_page, output_offset, 1, d);
            mem_len += 1;
        }
        const bool input_is_empty = req_len == 0;
        rpc = write_result && !input_is_empty;
        completed = (write_result && input_is_empty) || completed;
        rwfr = !(rpc || completed);
        cw[RPC] = rpc; cw[RWFR] = rwfr; cw[DONE] = completed;
#pragma unroll
        for (int i = 0; i < 8; ++i) s.st[i] = st[i];
        cw[TS_READ] = ts_read; cw[TS_WRITE] = ts_write; cw[INPUT_PAGE] = input_page; cw[INPUT_OFFSET] = input_offset; cw[OUTPUT_PAGE] = output_page;
        cw[OUTPUT_OFFSET] = output_offset; cw[NUM_ROUNDS] = num_rounds; cw[REQ_LEN] = req_len; cw[MEM_LEN] = mem_len;
    }
};

// F: Keccak or Sha256.  Wavefront 1 walks, wavefront 0 hashes; see the header comment.
template <class F>
__global__ __launch_bounds__(128) void k_fsm_seed(FsmSeedDev a) {
    constexpr u32 CARRIED = F::CARRIED, RAW = F::LOOP_WORDS - F::CARRIED;
    __shared__ typename F::State S;
    __shared__ u64 raw[2][RAW];
    __shared__ Events ev[2];
    __shared__ u64 chain[16];
    const u32 inst = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool walker = tid >= 64;
    const u32 e = lane & 15, el = e < 12 ? e : 0;
    auto is_chain = [](u32 w) { return (w >= F::MEM_TAIL && w < F::MEM_TAIL + 12) || (w >= F::REQ_HEAD && w < F::REQ_HEAD + 4); };
    u64 rcf[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { rcf[r] = p2::RC[12 * r + el]; rcf[4 + r] = p2::RC[12 * (26 + r) + el]; }
    u64* const col0 = a.loop + (u64)inst * a.limit;
    for (u32 w = tid; w < CARRIED; w += 128) {
        const u64 v = ov(a, inst, a.state0_slot[w]);
        if (w >= F::MEM_TAIL && w < F::MEM_TAIL + 12) chain[w - F::MEM_TAIL] = v;
        else if (w >= F::REQ_HEAD && w < F::REQ_HEAD + 4) chain[12 + w - F::REQ_HEAD] = v;
        else F::set(S, w, v);
        if (is_chain(w)) col0[(u64)w * a.in_stride] = v;
    }
    for (u32 w = tid; w < RAW; w += 128) raw[0][w] = col0[(u64)(CARRIED + w) * a.in_stride];
    if (tid == 0) { ev[0].n = 0; ev[1].n = 0; }
    __syncthreads();
    for (u32 c = 0; c < a.limit; ++c) {
        u64* const col = col0 + c;
        if (walker) {
            for (u32 w = lane; w < CARRIED; w += 64)
                if (!is_chain(w)) col[(u64)w * a.in_stride] = F::get(S, w);
            wave_sync();   // every lane has read the state of cycle c before lane 0 steps it (found by the emulated device of tests/emu: the lanes are in lockstep on the hardware, the memory model does not say so)
            if (lane == 0 && c + 1 < a.limit && !(a.debug & 2)) F::step(S, raw[c & 1], ev[c & 1]);
        } else {
            u64 nxt[(RAW + 63) / 64];
            const bool more = c + 1 < a.limit;
#pragma unroll
            for (u32 k = 0; k < (RAW + 63) / 64; ++k) {
                const u32 w = lane + 64 * k;
                nxt[k] = (more && w < RAW) ? col[1 + (u64)(CARRIED + w) * a.in_stride] : 0;
            }
            if (c) {   // the events of cycle c - 1, then the chain words of cycle c
                if (!(a.debug & 1)) run_events(chain, ev[(c - 1) & 1], lane, rcf);
                if (lane < 12) col[(u64)(F::MEM_TAIL + lane) * a.in_stride] = chain[lane];
                else if (lane < 16) col[(u64)(F::REQ_HEAD + lane - 12) * a.in_stride] = chain[lane];
            }
#pragma unroll
            for (u32 k = 0; k < (RAW + 63) / 64; ++k) {
                const u32 w = lane + 64 * k;
                if (w < RAW) raw[(c + 1) & 1][w] = nxt[k];
            }
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------ eip_4844
// eip_4844_entry_point (reference src/eip_4844/mod.rs:107-260; circuits/eip4844.cpp): iteration t absorbs Keccak block t of the blob
// and performs `cpi` Horner steps of the opening y = sum chunk_i z^(n-1-i) in the BLS12-381 scalar field.  Carried: the sponge (200
// bytes), the opening (16 limbs of 16 bits, lazily added before each reduction), the counter.  The two recurrences are independent:
// one lane of wavefront 0 runs the sponge, one lane of wavefront 1 the Horner steps (Montgomery products on 9 limbs of 32 bits,
// R = 2^288, so the lazily added operand < 2^257 needs no reduction first); every GROUP cycles all 128 threads flush the snapshots
// and stage the next group's bytes.  z = the last 16 bytes of keccak256(linear_hash || versioned_hash) (mod.rs:156-175), recomputed here.
struct EipSeedDev {
    u64* loop; u64 in_stride; u32 limit, n_instances, n_chunks, cpi;
    const u64* outer_inputs; u64 outer_in_stride;   // versioned_hash[32] | linear_hash_output[32], one byte per word
};
namespace bls {
constexpr u32 NL = 9;
__constant__ const u32 R_MOD[NL] = {0x00000001, 0xffffffff, 0xfffe5bfe, 0x53bda402, 0x09a1d805, 0x3339d808, 0x299d7d48, 0x73eda753, 0};
__constant__ const u32 R2[NL] = {0xbba87a71, 0x344171c0, 0xdf7ed1ca, 0x3c0538d1, 0x7ed9fe9f, 0x4aa18ade, 0x0b4094fb, 0x63643e57, 0};   // 2^576 mod r
constexpr u32 NPRIME = 0xffffffffu;   // -r^-1 mod 2^32
// a * b * 2^-288 mod r, fully reduced; a * b < r * 2^288 (CIOS)
__device__ __forceinline__ void mont_mul(u32 (&out)[NL], const u32 (&a)[NL], const u32 (&b)[NL]) {
    u32 t[NL + 2];
#pragma unroll
    for (u32 i = 0; i < NL + 2; ++i) t[i] = 0;
#pragma unroll
    for (u32 i = 0; i < NL; ++i) {
        u64 c = 0;
#pragma unroll
        for (u32 j = 0; j < NL; ++j) { c += (u64)a[j] * b[i] + t[j]; t[j] = (u32)c; c >>= 32; }
        c += t[NL]; t[NL] = (u32)c; t[NL + 1] = (u32)(c >> 32);
        const u32 m = t[0] * NPRIME;
        c = ((u64)m * R_MOD[0] + t[0]) >> 32;
#pragma unroll
        for (u32 j = 1; j < NL; ++j) { c += (u64)m * R_MOD[j] + t[j]; t[j - 1] = (u32)c; c >>= 32; }
        c += t[NL]; t[NL - 1] = (u32)c;
        t[NL] = t[NL + 1] + (u32)(c >> 32);
    }
    // t < 2 r: one conditional subtraction
    u32 d[NL];
    u64 br = 0;
#pragma unroll
    for (u32 j = 0; j < NL; ++j) { const u64 x = (u64)t[j] - R_MOD[j] - br; d[j] = (u32)x; br = (x >> 32) & 1; }
    const bool ge = t[NL] != 0 || br == 0;
#pragma unroll
    for (u32 j = 0; j < NL; ++j) out[j] = ge ? d[j] : t[j];
}
}  // namespace bls

constexpr u32 EIP_GROUP = 16, EIP_MAX_CPI = 8;
__global__ __launch_bounds__(128) void k_eip4844_seed(EipSeedDev a) {
    constexpr u32 RATE = 136, CHUNK = 31, CARRIED = 217;
    __shared__ u64 snapA[EIP_GROUP][25];
    __shared__ u32 snapO[EIP_GROUP][16];
    __shared__ alignas(8) unsigned char blk[EIP_GROUP][RATE];
    __shared__ unsigned char chk[EIP_GROUP][CHUNK * EIP_MAX_CPI];
    __shared__ u32 zR[bls::NL];
    __shared__ u64 ksh[2][55];
    const u32 inst = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool sponge = tid < 64;
    u64* const col0 = a.loop + (u64)inst * a.limit;
    const u32 chunk_bytes = CHUNK * a.cpi;
    u64 A = 0;            // wavefront 0: lane x + 5y holds sponge lane (x, y)
    u32 opening[16];      // wavefront 1, lane 0
#pragma unroll
    for (int i = 0; i < 16; ++i) opening[i] = 0;
    if (!sponge) {        // z = the last 16 bytes of keccak256(linear_hash || versioned_hash), then z * R mod r
        u64 Z = 0;
        if (lane < 8) {
            for (u32 k = 0; k < 8; ++k) {
                const u32 j = 8 * lane + k;
                Z |= (a.outer_inputs[(u64)(j < 32 ? 32 + j : j - 32) * a.outer_in_stride + inst] & 0xff) << (8 * k);
            }
        }
        if (lane == 8) Z = 0x01;
        if (lane == 16) Z = 0x80ULL << 56;
        Z = keccak_f_lanes(Z, lane, ksh[1]);
        if (lane < 4) ksh[1][lane] = Z;
        wave_sync();
        if (lane == 0) {
            u32 z[bls::NL], r2[bls::NL], out[bls::NL];
#pragma unroll
            for (u32 i = 0; i < bls::NL; ++i) { z[i] = 0; r2[i] = bls::R2[i]; }
            for (u32 j = 0; j < 16; ++j) {   // digest bytes 16..31 big-endian: byte 31 is the least significant
                const u32 byte = (u32)(ksh[1][(31 - j) / 8] >> (8 * ((31 - j) % 8))) & 0xff;
                z[j / 4] |= byte << (8 * (j % 4));
            }
            bls::mont_mul(out, z, r2);
            for (u32 i = 0; i < bls::NL; ++i) zR[i] = out[i];
        }
    }
    for (u32 g0 = 0; g0 < a.limit; g0 += EIP_GROUP) {
        const u32 gn = min(EIP_GROUP, a.limit - g0);
        // ---- stage the group's bytes (consecutive threads: consecutive cycles)
        for (u32 i = tid; i < gn * RATE; i += 128) blk[i % gn][i / gn] = (unsigned char)col0[g0 + i % gn + (u64)(CARRIED + i / gn) * a.in_stride];
        for (u32 i = tid; i < gn * chunk_bytes; i += 128) chk[i % gn][i / gn] = (unsigned char)col0[g0 + i % gn + (u64)(CARRIED + RATE + i / gn) * a.in_stride];
        __syncthreads();
        if (sponge) {
            for (u32 k = 0; k < gn; ++k) {
                if (lane < 25) snapA[k][lane] = A;
                if (lane < 17) A ^= ((const u64*)blk[k])[lane];
                A = keccak_f_lanes(A, lane, ksh[0]);   // the last block's padding only matters after the last cycle
            }
        } else if (lane == 0) {
            u32 zr[bls::NL];
#pragma unroll
            for (u32 i = 0; i < bls::NL; ++i) zr[i] = zR[i];
            for (u32 k = 0; k < gn; ++k) {
#pragma unroll
                for (int i = 0; i < 16; ++i) snapO[k][i] = opening[i];
                for (u32 c = 0; c < a.cpi; ++c) {
                    const u32 idx = a.cpi * (g0 + k) + c;
                    if (idx >= a.n_chunks) break;
                    const unsigned char* b = chk[k] + CHUNK * c;
#pragma unroll
                    for (int i = 0; i < 15; ++i) opening[i] += (u32)b[2 * i] | ((u32)b[2 * i + 1] << 8);   // add_lazy: limb-wise, no carries
                    opening[15] += b[30];
                    if (idx + 1 == a.n_chunks) break;   // the last chunk is added without a multiplication (mod.rs:200-202)
                    u32 v[bls::NL], out[bls::NL];
                    u64 carry = 0;
#pragma unroll
                    for (u32 i = 0; i < 8; ++i) {
                        carry += (u64)opening[2 * i] + ((u64)opening[2 * i + 1] << 16);
                        v[i] = (u32)carry; carry >>= 32;
                    }
                    v[8] = (u32)carry;
                    bls::mont_mul(out, v, zr);
#pragma unroll
                    for (u32 i = 0; i < 8; ++i) { opening[2 * i] = out[i] & 0xffff; opening[2 * i + 1] = out[i] >> 16; }
                }
            }
        }
        __syncthreads();
        // ---- the carried words of the group's cycles
        for (u32 i = tid; i < gn * CARRIED; i += 128) {
            const u32 k = i % gn, w = i / gn;
            const u64 v = w < 200 ? ((const unsigned char*)snapA[k])[w] : w < 216 ? snapO[k][w - 200] : (u64)(g0 + k);
            col0[g0 + k + (u64)w * a.in_stride] = v;
        }
        __syncthreads();
    }
}

}  // namespace zkf
