// kernels_engine2.hpp — the plain witness interpreter, scalar-decoded ("v2" device programs, cs.cpp emit_scope).
//
// The first interpreter (kernels_engine.hpp run_lane) fetched program words through a VGPR window + v_readlane and
// resolved every operand through a three-way kind branch; rocprofv3 PMC on main_vm (profiles/r2_pmc_vm_loop_v1.txt)
// showed what that costs: 2.1 M instructions per wavefront for 7 073 ops, 80 % of every wave's life in s_waitcnt, and
// — from the ISA — a vmcnt(0) behind nearly EVERY operand load: the compiler could not keep the loads of a group in flight
// across the kind branches, so each of the 28 k loads per lane paid its own L2 round trip (TCP_TCC_READ_REQ_LATENCY ~ 950 clk).
//
// This interpreter is built around what the variable store (cs.cpp assign_store_slots) guarantees:
//   * Program words are read by the SCALAR unit: one s_load_dwordx16 per op (header + operands land in SGPRs; every
//     wavefront walks the same words, so they are scalar-cache / L2 hits), no readlane, no window bookkeeping.
//   * Data operands are always store slots (the recorder never puts a pool constant or an outer value in a data position;
//     the only ops that read those are ZK_OP_CONST imports and the FMA / LC4 coefficients): an operand load is ONE
//     instruction, buffer_load_dwordx2 with soffset = slot * 512 from an SGPR, branch-free, so all operand loads of a
//     group are issued back to back and waited for once.
//   * Coefficients (FMA q, l; LC4 k0..3) are pool indices read with scalar loads; "coefficient == 1" is a scalar branch.
//   * Outputs have NO destination words: an op's outputs are the next consecutive store slots (production order), the
//     kernel keeps one running SGPR byte offset; a store is buffer_store_dwordx2 + s_add.
// Same values, same slots as the strand / wide / sequential kernels that still run the v1 form.
#pragma once
#include "kernels_engine.hpp"
#include "keccak_macro.hpp"
#include "sha256_macro.hpp"
#include "sha256_macro4.hpp"
#include "bytebuf_macro.hpp"
#include <utility>

namespace zke {

typedef uint32_t u32x16_a4 __attribute__((ext_vector_type(16), aligned(4)));
typedef __attribute__((address_space(4))) const u32x16_a4* prog16_ptr;
typedef __attribute__((address_space(4))) const uint32_t* prog1_ptr;
typedef __attribute__((address_space(4))) const uint64_t* cpool_ptr;

// v2 group caps (cs.cpp group_cap must agree): members of one header
constexpr uint32_t G2_INPUT = 8, G2_SELECT = 5, G2_FMA = 3, G2_LOOKUP = 4, G2_U32MULADD = 3;
// multiplicities: wave-aggregated atomics in the interpreter when the host passes the vector (sc.mult), or the k_multiplicities pass
// after the witness kernels when it passes nullptr (cs.cpp multiplicity_mode); -DZKGL_STUB_MULT: neither (time attribution)
// ELIMINATION PROBES (time attribution: profiles/r3_loop_probe.md, tools/variants.sh): a library built with -DZKGL_EXPERIMENT=<mask> leaves parts of the
// interpreter out (WRONG results, timing only; bench.py runs it with ZKGL_STUB_RUN=1).  The product is built without: every probe is `false`
// and its branch is discarded at compile time.  This is the only preprocessor switch of the interpreter besides the two opt-in macro-op backends.
namespace probe {
#ifdef ZKGL_EXPERIMENT
constexpr uint32_t MASK = ZKGL_EXPERIMENT;
#else
constexpr uint32_t MASK = 0;
#endif
constexpr bool NO_LOADS = MASK & 1, NO_STORES = MASK & 2, NO_P2_SBOX = MASK & 4, NO_P2_LINEAR = MASK & 8, NO_FMA = MASK & 16, NO_INV = MASK & 32, NO_FIND = MASK & 64, NO_MULT = MASK & 128;
}  // namespace probe
#define ZKGL_MULT_ON (!probe::NO_MULT)
template <uint32_t N> struct GroupSize { static constexpr uint32_t value = N; };

// table row of a key tuple of <= 2 keys (the grouped lookups), no key array: a dynamically indexed array would live in scratch
__device__ __forceinline__ uint32_t table_find2(const zk_table_desc& t, const uint64_t* __restrict__ words, uint64_t k0, uint64_t k1) {
    if (t.dense) {
        // full product of power-of-two key ranges, last key fastest: row index = packed key; in the table iff every key is in range
        const uint32_t top = 31 - __clz(t.n_rows);
        const uint32_t bits0 = top - t.key_shift[0];
        bool ok = (k0 >> bits0) == 0;
        uint64_t idx = k0 << t.key_shift[0];
        if (t.n_keys > 1) {
            const uint32_t bits1 = t.key_shift[0] - t.key_shift[1];
            ok = ok && (k1 >> bits1) == 0;
            idx += k1 << t.key_shift[1];
        }
        return ok ? (uint32_t)idx : t.n_rows;
    }
    const uint32_t w = t.n_keys + t.n_vals;
    const uint64_t* rows = words + (size_t)t.word_off;
    uint32_t lo = 0, hi = t.n_rows;  // rows sorted lexicographically by key tuple
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t r0 = rows[(size_t)mid * w];
        int cmp = r0 < k0 ? -1 : (r0 > k0 ? 1 : 0);
        if (cmp == 0 && t.n_keys > 1) {
            const uint64_t r1 = rows[(size_t)mid * w + 1];
            cmp = r1 < k1 ? -1 : (r1 > k1 ? 1 : 0);
        }
        if (cmp == 0) return mid;
        if (cmp < 0) lo = mid + 1; else hi = mid;
    }
    return t.n_rows;
}

// table row of a key tuple of <= 3 keys without a key array (see table_find2)
__device__ __forceinline__ uint32_t table_find3(const zk_table_desc& t, const uint64_t* __restrict__ words, uint64_t k0, uint64_t k1, uint64_t k2) {
    if (t.n_keys <= 2) return table_find2(t, words, k0, k1);
    if (t.dense) {
        const uint32_t top = 31 - __clz(t.n_rows);
        const bool ok = (k0 >> (top - t.key_shift[0])) == 0 && (k1 >> (t.key_shift[0] - t.key_shift[1])) == 0 && (k2 >> (t.key_shift[1] - t.key_shift[2])) == 0;
        const uint64_t idx = (k0 << t.key_shift[0]) + (k1 << t.key_shift[1]) + (k2 << t.key_shift[2]);
        return ok ? (uint32_t)idx : t.n_rows;
    }
    const uint32_t w = t.n_keys + t.n_vals;
    const uint64_t* rows = words + (size_t)t.word_off;
    uint32_t lo = 0, hi = t.n_rows;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t r0 = rows[(size_t)mid * w], r1 = rows[(size_t)mid * w + 1], r2 = rows[(size_t)mid * w + 2];
        const int cmp = r0 != k0 ? (r0 < k0 ? -1 : 1) : r1 != k1 ? (r1 < k1 ? -1 : 1) : r2 != k2 ? (r2 < k2 ? -1 : 1) : 0;
        if (cmp == 0) return mid;
        if (cmp < 0) lo = mid + 1; else hi = mid;
    }
    return t.n_rows;
}

// fused mode: a gate the witness kernel evaluates itself is violated -> the macro row of the failure key (the host re-runs the
// gate-by-gate program on the stored values to name the gate)
__device__ __forceinline__ void report_fused(unsigned long long* f, uint32_t lane) {
    atomicMin(f, ((unsigned long long)lane << 32) | ((unsigned long long)0xffffeu << 12));
}

// K8, out of line: Keccak-f with the state in registers (inner loops unrolled: every lane index static), streaming its ~30 k outputs to
// consecutive store slots of the wavefront's tile.  Its own function so that its register allocation (25 lanes = 50 VGPRs + theta's
// columns) does not land on the interpreter loop; the 25 input lanes come in through scratch, the next output offset comes back.
__device__ __noinline__ uint32_t keccak_f_stream(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_byte, uint32_t dst, uint32_t bstep, const uint64_t* in25,
                                                  uint32_t share, uint32_t n_share_mask) {
    // Cooperative form (strand kernels): the macro-op sits in EVERY strand's program; each of the tile's wavefronts computes the whole
    // permutation (cheap: ~10 k VALU instructions) and stores the outputs whose running index & n_share_mask == share — a single
    // wavefront sustains only ~14 stores / us (its 30 k stores took 2.2 ms and sat on the tile's critical path), sixteen share them.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    uint64_t sl[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) sl[i] = in25[i];
    struct Emit {
        __amdgpu_buffer_rsrc_t rsrc;
        uint32_t lane_byte, d, bstep, cnt, mine, mask;
        __device__ __forceinline__ void store(uint32_t v, uint32_t at) {
            u32x2 o;
            o.x = v; o.y = 0u;
            __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, lane_byte, at, 0);
        }
        // one block = eight consecutive outputs; the strands share the work block by block
        __device__ __forceinline__ void block8(uint64_t v) {
            if ((cnt & mask) == mine) {
#pragma unroll
                for (int k = 0; k < 8; ++k) store((uint32_t)(v >> (8 * k)) & 0xffu, d + k * bstep);
            }
            d += 8 * bstep;
            ++cnt;
        }
        __device__ __forceinline__ void one(uint64_t v) {
            if ((cnt & mask) == mine) store((uint32_t)v, d);
            d += bstep;
            ++cnt;
        }
    } emit{rsrc, lane_byte, uni(dst), bstep, 0u, uni(share), uni(n_share_mask)};
    zkk::ComputeBackend<Emit> be(emit);
    zkk::keccak_f_unrolled(be, sl, zkk::RC);
    return emit.d;
}

// K8, out of line: one SHA-256 compression with every intermediate of the byte-table decomposition streamed out (zks::compress);
// cooperative like keccak_f_stream: every strand computes, strand `share` stores every (mask + 1)-th block of outputs.
__device__ __noinline__ uint32_t sha256_rounds_stream(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_byte, uint32_t dst, uint32_t bstep, const uint32_t* in24,
                                                       uint32_t share, uint32_t n_share_mask) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    struct Emit {
        __amdgpu_buffer_rsrc_t rsrc;
        uint32_t lane_byte, d, bstep, cnt, mine, mask;
        __device__ __forceinline__ void block(const uint64_t* v, int n) {
            if ((cnt & mask) == mine) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < n) {
                        u32x2 o;
                        o.x = (uint32_t)v[k]; o.y = (uint32_t)(v[k] >> 32);
                        __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, lane_byte, d + k * bstep, 0);
                    }
            }
            d += n * bstep;
            ++cnt;
        }
    } emit{rsrc, lane_byte, uni(dst), bstep, 0u, uni(share), uni(n_share_mask)};
    uint32_t st[8], blk[16], w[64];   // w is indexed dynamically: scratch
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = in24[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) blk[i] = in24[8 + i];
    zks::ComputeBackend<Emit> be(emit);
    zks::compress(be, st, blk, w, zks::K);
    return emit.d;
}

// K8, out of line: one SHA-256 compression over the REFERENCE's 4-bit-chunk table set (ZK_OP_SHA256_ROUNDS with a = 1; zks4::compress, sha256_macro4.hpp):
// 26 088 outputs; cooperative like sha256_rounds_stream — every strand computes, strand `share` stores every (mask + 1)-th run of eight outputs.
// Called by the kernels instantiated with XMACROS & X_SHA4 only (k_witness_strands2_x / k_witness_plain_x below): the kernels of every other circuit
// are compiled without it (with it inside, the hash circuits' strand kernel went from 26 to 219 spilled VGPRs — profiles/r6_resource_usage.md).
__device__ __noinline__ uint32_t sha256_rounds4_stream(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_byte, uint32_t dst, uint32_t bstep, const uint32_t* in24,
                                                        uint32_t share, uint32_t n_share_mask) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    struct Emit {
        __amdgpu_buffer_rsrc_t rsrc;
        uint32_t lane_byte, d, bstep, cnt, mine, mask;
        __device__ __forceinline__ void one(uint64_t v) {
            if (((cnt >> 3) & mask) == mine) {
                u32x2 o;
                o.x = (uint32_t)v; o.y = (uint32_t)(v >> 32);
                __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, lane_byte, d, 0);
            }
            d += bstep;
            ++cnt;
        }
    } emit{rsrc, lane_byte, uni(dst), bstep, 0u, uni(share), uni(n_share_mask)};
    uint32_t st[8], blk[16], w[64];   // w is indexed dynamically: scratch
    typename zks4::ComputeBackend<Emit>::Splits wsp[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = in24[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) blk[i] = in24[8 + i];
    zks4::ComputeBackend<Emit> be(emit);
    zks4::compress(be, st, blk, w, wsp, zks::K);
    return emit.d;
}

// ZK_OP_BYTEBUF_FILL's device backend: in the kernels instantiated with XMACROS & X_BYTEBUF only (recordings made with ZKGL_BYTEBUF_MACRO=1)
struct BytebufInv {   // k^-1 mod p for 0 < |k| < INV_SMALL_N
    __device__ __forceinline__ uint64_t operator()(int32_t k) const {
        const uint64_t r = p2::INV_SMALL[(uint32_t)(k < 0 ? -k : k) & (p2::INV_SMALL_N - 1)];
        return k < 0 ? 0xFFFFFFFF00000001ull - r : r;
    }
};
// K8, out of line: ByteBuffer::fill_with_bytes with every intermediate streamed out (zkb::fill_with_bytes, bytebuf_macro.hpp): the byte
// arrays are packed in registers (bytebuf_macro.hpp ComputeBackend), the values are small integers; cooperative like keccak_f_stream — every strand computes,
// strand `share` stores every (mask + 1)-th run of eight outputs.
__device__ __noinline__ uint32_t bytebuf_fill_stream(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_byte, uint32_t dst, uint32_t bstep, const uint32_t* packed /* 48 + 8 words */,
                                                      int32_t filled, int32_t offset, int32_t meaningful, uint32_t share, uint32_t n_share_mask) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    struct Emit {
        __amdgpu_buffer_rsrc_t rsrc;
        uint32_t lane_byte, d, bstep, cnt, mine, mask;
        __device__ __forceinline__ void one(uint64_t v) {
            if (((cnt >> 3) & mask) == mine) {
                u32x2 o;
                o.x = (uint32_t)v; o.y = (uint32_t)(v >> 32);
                __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, lane_byte, d, 0);
            }
            d += bstep;
            ++cnt;
        }
    } emit{rsrc, lane_byte, uni(dst), bstep, 0u, uni(share), uni(n_share_mask)};
    BytebufInv inv;
    zkb::ComputeBackend<Emit, BytebufInv> be(emit, inv);
#pragma unroll
    for (int k = 0; k < zkb::BUF / 4; ++k) be.bytes_[k] = packed[k];
#pragma unroll
    for (int k = 0; k < zkb::IN / 4; ++k) be.in_[k] = be.sh_[k] = packed[zkb::BUF / 4 + k];
#pragma unroll
    for (int k = 0; k < zkb::BUF / 32; ++k) be.pl_[k] = 0;
    int32_t f = filled;
    zkb::fill_with_bytes(be, f, offset, meaningful);
    return emit.d;
}

// STRANDS: the strand form (k_witness_strands2): one destination word per op behind the operands — the store slot of its first
// output (a strand's ops are not consecutive in production order) — and ZK_OP_BARRIER between the dependency levels.
// NARROW (k_witness_loop_narrow; store_geom.hpp, cs.cpp build_narrow_layout): the scope's store is a narrow store — data operand words are ADDRESS
// WORDS (first unit | class << 28), an output is a byte-class value when its bit is set in the header's class word (sc.cls, one word per header):
// buffer_load_ubyte / buffer_store_byte, 64 B per wavefront instead of 512.  A value that does not fit its byte slot is the fused mode's failure
// (the class of a variable is a bound that holds in EVERY satisfying witness: CS::bound_values); the host then repeats the step on the ordinary store.
// XMACROS: macro-op backends beyond the basic set of WITH_BIGINT (X_SHA4: ZK_OP_SHA256_ROUNDS with a = 1, the reference's 4-bit-chunk tables; X_BYTEBUF:
// ZK_OP_BYTEBUF_FILL).  Each has its own kernel instantiations, launched for the circuits that record the op: nobody else's kernel carries its registers.
constexpr int X_SHA4 = 1, X_BYTEBUF = 2;
template <bool WITH_BIGINT, bool WIDE, int BLOCK = TPB, bool STRANDS = false, bool NARROW = false, int XMACROS = 0>
__device__ __forceinline__ void run_tile2(const ScopeDev& sc, const uint32_t lane, const uint32_t inst, const bool active,
                                          uint32_t word_begin, uint32_t word_end, uint32_t slot_begin, const uint32_t* cls_words = nullptr) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    // store_geom.hpp: V# = the lane tile of this wavefront, voffset = the lane's byte in a value of the tile, soffset = slot << bsh
    const TileAddr ta = tile_addr(sc.cells, sc.n_cells, lane);
    const uint32_t lane_byte = ta.lane_byte;
    const uint32_t bsh = uni(ta.shift), tsh = bsh - 3, bstep = 1u << bsh;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ta.base, 0, -1, 0x00020000);
    uint64_t* __restrict__ wide_cells = ta.base + (lane_byte >> 3);
    const prog1_ptr prog = (prog1_ptr)(uintptr_t)sc.prog;
    const cpool_ptr cpool = (cpool_ptr)(uintptr_t)sc.consts;

    static_assert(!NARROW || (!WIDE && !STRANDS), "the narrow store is read by the plain buffer-addressed kernel only");
    const uint32_t ush = bsh - 3;                   // NARROW: a unit of the tile = 1 << ush bytes (one byte per lane)
    const uint32_t lane_unit = lane_byte >> 3;      // NARROW: this lane's byte in a unit
    const prog1_ptr cls = (prog1_ptr)(uintptr_t)cls_words;   // NARROW: one class word per header of the program
    uint32_t opi = 0;                               // NARROW: headers decoded so far (index into cls)
    uint32_t dst = WIDE ? slot_begin : NARROW ? slot_begin << ush : slot_begin << bsh;  // next output: slot index (WIDE) or byte offset in the tile (NARROW: slot_begin = first unit)
    auto ldv = [&](uint32_t slot) -> uint64_t {
        if constexpr (probe::NO_LOADS) return (uint64_t)slot * 0x9E3779B97F4A7C15ull + lane_byte;   // (operand values without the memory access)
        if constexpr (WIDE) return wide_cells[(size_t)slot << tsh];
        if constexpr (NARROW) {   // `slot` is an address word; its class is wave-uniform (a scalar branch around one of two loads)
            if (slot & zkgeom::AW_BYTE) return (uint64_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, lane_unit, (slot & zkgeom::AW_MASK) << ush, 0);
            u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_byte, slot << ush, 0);
            return (uint64_t)v.x | ((uint64_t)v.y << 32);
        }
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_byte, slot << bsh, 0);
        return (uint64_t)v.x | ((uint64_t)v.y << 32);
    };
    auto st = [&](uint64_t v) {
        if constexpr (probe::NO_STORES) {
            asm volatile("" ::"v"(v), "s"(dst));
            dst += WIDE ? 1 : bstep;
        } else if constexpr (WIDE) {
            wide_cells[(size_t)dst << tsh] = v;
            dst += 1;
        } else {
            u32x2 o;
            o.x = (uint32_t)v; o.y = (uint32_t)(v >> 32);
            __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, lane_byte, dst, 0);
            dst += bstep;
        }
    };

    // SELECT flags as bit planes (ZK_OP_FLAG_PLANES; plain kernels, loop scope): [wavefront of the block][0: != 0, 1: > 1][plane id]
    // (strand form: the wavefronts of the workgroup share ONE tile, hence one set of planes; a plane is written in the level after its
    // flag's and read from the level after that — cs.cpp build_strands — with the workgroup barrier of ZK_OP_BARRIER in between)
    // (a strand form of the planes existed in round 5 and was never measured; compiled in, it cost the strand kernels of EVERY circuit registers —
    // k_witness_strands2<false,false> 79 -> 92 VGPRs, 6 -> 5 wavefronts per SIMD, profiles/r5_resource_usage.md — and was deleted in round 6)
    constexpr bool PLANES = !STRANDS;
    __shared__ uint64_t flag_planes[!PLANES ? 1 : STRANDS ? 2 * zkdev::FLAG_PLANES : (BLOCK / 64) * 2 * zkdev::FLAG_PLANES];
    uint64_t* const planes = flag_planes + (STRANDS ? 0 : uni(threadIdx.x >> 6) * 2 * zkdev::FLAG_PLANES);
    // (this lane's index in its wavefront is recomputed where used — two mbcnt — rather than held in a VGPR across the interpreter loop)
    auto wave_lane_now = [] { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };

    constexpr uint32_t D = STRANDS ? 1 : 0;  // destination words per op
    auto out_to = [&](uint32_t slot) { if constexpr (STRANDS) dst = WIDE ? slot : slot << bsh; };
    uint32_t pc = word_begin;
    uint32_t nonbool_seen = 0;   // uniform: some flag copied into a plane held a value > 1 in some lane (never, on a satisfiable witness)
    bool fused_bad = false;   // fused mode: a gate evaluated here (SELECT's exception, a lookup miss) is violated; reported once, below
    uint32_t ocm = 0;         // NARROW: class word of the header being executed (bit k: its k-th output is a byte-class value)
    auto stn = [&](uint64_t v, uint32_t k) {   // output k of the header (ops whose outputs may be byte-class: cs.cpp narrow_capable)
        if constexpr (NARROW) {
            if ((ocm >> k) & 1u) {
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, rsrc, lane_unit, dst, 0);
                fused_bad |= v > 0xffull;   // does not fit: no satisfying witness holds this value (the host repeats the step on the ordinary store)
                dst += 1u << ush;
                return;
            }
        }
        st(v);
    };
    while (pc < word_end) {
        const u32x16_a4 W = *(prog16_ptr)(prog + pc);  // s_load_dwordx16: header + up to 15 operand words (host pads the program)
        if constexpr (NARROW) { ocm = cls[opi]; ++opi; }   // (a second scalar load in flight beside the header's)
        const uint32_t h = W[0];
        const uint32_t op = h & 0xff, pa = (h >> 8) & 0xff, pb = h >> 16;
        switch (op) {
        case ZK_OP_CONST: {  // the only op whose operand carries a kind: a pool constant, or (loop scope) a value of the outer scope
            const uint32_t w = W[1];
            out_to(W[2]);
            pc += 2 + D;
            uint64_t v;
            if ((w & ZK_OPERAND_KIND_MASK) == ZK_OPERAND_OUTER) v = sc.outer_cells[cell_off(sc.outer_n_cells, w & ZK_OPERAND_IDX_MASK, inst)];
            else v = cpool[w & ZK_OPERAND_IDX_MASK];
            stn(v, 0);
        } break;
        // Grouped ops: one straight-line instance per group size (no predication: every operand load of the group is issued
        // before the first wait, and the register allocator sees exactly the live set of that size).
        case ZK_OP_INPUT: {
            auto body = [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) v[g] = sc.inputs[(size_t)W[1 + g] * sc.in_stride + lane];
                pc += 1 + N + D * N;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) { out_to(W[(1 + N + g) & 15]); stn(v[g], g); }
            };
            switch (pb) {
            case 0: body(GroupSize<1>{}); break;
            case 1: body(GroupSize<2>{}); break;
            case 2: body(GroupSize<3>{}); break;
            case 3: body(GroupSize<4>{}); break;
            case 4: body(GroupSize<5>{}); break;
            case 5: body(GroupSize<6>{}); break;
            case 6: body(GroupSize<7>{}); break;
            default: body(GroupSize<8>{}); break;
            }
        } break;
        case ZK_OP_FMA: {  // [q, l: pool indices][a, b, c: slots] per member
            auto body = [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t in[N][3], q[N], l[N];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
#pragma unroll
                    for (uint32_t i = 0; i < 3; ++i) in[g][i] = ldv(W[1 + g * 5 + 2 + i]);
                    q[g] = cpool[W[1 + g * 5]];
                    l[g] = cpool[W[1 + g * 5 + 1]];
                }
                pc += 1 + N * 5 + D * N;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
                    out_to(W[(1 + N * 5 + g) & 15]);
                    if constexpr (probe::NO_FMA) stn(in[g][0] ^ in[g][1] ^ in[g][2] ^ q[g] ^ l[g], g);
                    else {
                        const uint64_t ab = gl::mul(in[g][0], in[g][1]);
                        stn(gl::add(q[g] == 1 ? ab : gl::mul(q[g], ab), l[g] == 1 ? in[g][2] : gl::mul(l[g], in[g][2])), g);
                    }
                }
            };
            switch (pb) {
            case 0: body(GroupSize<1>{}); break;
            case 1: body(GroupSize<2>{}); break;
            default: body(GroupSize<3>{}); break;
            }
        } break;
        case ZK_OP_LC4: {  // [k0..3: pool indices][t0..3: slots]
            uint64_t t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = ldv(W[5 + i]);
            uint64_t r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r = gl::fma(cpool[W[1 + i]], t[i], r);
            out_to(W[9]);
            pc += 9 + D;
            stn(r, 0);
        } break;
        case ZK_OP_FLAG_PLANES: if constexpr (PLANES) {
            const uint32_t n = pb + 1;
            uint64_t v[7];
#pragma unroll
            for (uint32_t k = 0; k < 7; ++k) if (k < n) v[k] = ldv(W[1 + 2 * k]);
            pc += 1 + 2 * n;
#pragma unroll
            for (uint32_t k = 0; k < 7; ++k)
                if (k < n) {
                    const uint64_t m = __ballot(v[k] != 0), nb = __ballot(v[k] > 1);
                    if (wave_lane_now() == 0) { planes[W[2 + 2 * k]] = m; planes[zkdev::FLAG_PLANES + W[2 + 2 * k]] = nb; }
                    if constexpr (!STRANDS) nonbool_seen |= (uint32_t)(nb != 0);   // (strands: another wavefront may have copied the flag — the SELECT reads both planes)
                }
        } else { return; } break;   // (not emitted for this form: cs.cpp)
        case ZK_OP_SELECT:
        if (PLANES && pa == 1) {
            // flags from the bit planes.  A wavefront whose lanes agree on a flag loads the selected operand twice (the second load hits
            // the line the first one brought) instead of both: no branch, no fetch of the branch nobody takes.
            auto body = [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t a[N], b[N];
                const uint32_t wave_lane = wave_lane_now();
                uint32_t fbits = 0;   // this lane's flag of member g in bit g
                uint32_t nbany = nonbool_seen;   // uniform: a flag > 1 somewhere in the wavefront (strands: read per member from the second plane)
                uint64_t mv[N], nv[N];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {   // all plane reads in flight before the first wait
                    mv[g] = planes[W[1 + g * 3]];
                    if constexpr (STRANDS) nv[g] = planes[zkdev::FLAG_PLANES + W[1 + g * 3]];
                }
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
                    const uint32_t mlo = uni((uint32_t)mv[g]), mhi = uni((uint32_t)(mv[g] >> 32));
                    uint32_t nz = nonbool_seen;
                    if constexpr (STRANDS) { nz = uni((uint32_t)nv[g] | (uint32_t)(nv[g] >> 32)); nbany |= nz; }
                    fbits |= (uint32_t)((mv[g] >> wave_lane) & 1) << g;
                    const uint32_t sa = W[1 + g * 3 + 1], sb = W[1 + g * 3 + 2];
                    a[g] = ldv((mlo | mhi) == 0 ? sb : sa);
                    b[g] = ldv(((mlo & mhi) == ~0u && !nz) ? sa : sb);
                }
                pc += 1 + N * 3 + D * N;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) { out_to(W[(1 + N * 3 + g) & 15]); stn(((fbits >> g) & 1) ? a[g] : b[g], g); }
                if (nbany) {   // SelectionGate on the operands held here: violated iff the selector is not 0 / 1 and the branches differ
#pragma unroll
                    for (uint32_t g = 0; g < N; ++g) fused_bad |= ((planes[zkdev::FLAG_PLANES + W[1 + g * 3]] >> wave_lane) & 1) && a[g] != b[g];
                }
            };
            switch (pb) {
            case 0: body(GroupSize<1>{}); break;
            case 1: body(GroupSize<2>{}); break;
            case 2: body(GroupSize<3>{}); break;
            case 3: body(GroupSize<4>{}); break;
            default: body(GroupSize<5>{}); break;
            }
        } else {
            auto body = [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t in[N][3];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
#pragma unroll
                    for (uint32_t i = 0; i < 3; ++i) in[g][i] = ldv(W[1 + g * 3 + i]);
                }
                pc += 1 + N * 3 + D * N;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) { out_to(W[(1 + N * 3 + g) & 15]); stn(in[g][0] ? in[g][1] : in[g][2], g); }
                // SelectionGate on the operands held here: violated iff the selector is not 0 / 1 and the branches differ
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) fused_bad |= in[g][0] > 1 && in[g][1] != in[g][2];
            };
            switch (pb) {
            case 0: body(GroupSize<1>{}); break;
            case 1: body(GroupSize<2>{}); break;
            case 2: body(GroupSize<3>{}); break;
            case 3: body(GroupSize<4>{}); break;
            default: body(GroupSize<5>{}); break;
            }
        } break;
        case ZK_OP_ISZERO: {
            const uint64_t x = ldv(W[1]);
            out_to(W[2]);
            pc += 2 + D;
            stn(x == 0 ? 1ull : 0ull, 0);
            if constexpr (probe::NO_INV) stn(x, 1);
            else stn(p2::inv_wave(x), 1);   // small |x| in every lane (flags, counters, position differences): one gather instead of 72 multiplications
        } break;
        case ZK_OP_UADD: {
            const uint64_t x = ldv(W[1]), y = ldv(W[2]), ci = ldv(W[3]);
            out_to(W[4]);
            pc += 4 + D;
            const uint64_t s = x + y + ci;  // operands < 2^32
            stn(s & ((1ull << pa) - 1), 0);
            stn(s >> pa, 1);
        } break;
        case ZK_OP_USUB: {
            const uint64_t x = ldv(W[1]), y = ldv(W[2]), bi = ldv(W[3]);
            out_to(W[4]);
            pc += 4 + D;
            const uint64_t sub = y + bi;
            const uint64_t borrow = x < sub ? 1 : 0;
            stn((x + (borrow << pa)) - sub, 0);
            stn(borrow, 1);
        } break;
        case ZK_OP_DOT4: {
            uint64_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ldv(W[1 + i]);
            out_to(W[9]);
            pc += 9 + D;
            uint64_t r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r = gl::fma(v[2 * i], v[2 * i + 1], r);
            st(r);
        } break;
        case ZK_OP_MATMUL12: {
            uint64_t s[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) s[i] = ldv(W[1 + i]);
            out_to(W[13]);
            pc += 13 + D;
            if (pa == 0) p2::mds_external(s); else p2::mds_inner(s);
#pragma unroll
            for (int i = 0; i < 12; ++i) st(s[i]);
        } break;
        case ZK_OP_SPLIT: {
            uint64_t x = ldv(W[1]);
            out_to(W[2]);
            pc += 2 + D;
            for (uint32_t i = 0; i < pa; ++i) {
                stn(i + 1 == pa ? x : (x & ((1ull << pb) - 1)), i < 32 ? i : 31);   // (class word: 32 outputs; the host keeps chunk 31 onwards in 8-byte slots)
                x >>= pb;
            }
        } break;
        case ZK_OP_LOOKUP: {  // [table id][keys of every member]; pb = n_vals | (members - 1) << 8
            const uint32_t tid = W[1];
            const zk_table_desc t = load_table_desc(sc.tables, tid);
            const uint32_t nv = pb & 0xff, grp = (pb >> 8) + 1;
            const uint8_t* __restrict__ tb = reinterpret_cast<const uint8_t*>(sc.table_words + (t.dense >> 2));
            const uint32_t w = t.n_keys + t.n_vals;
            if (pa <= 2 && nv <= 2) {
                auto body = [&](auto n_) {
                    constexpr uint32_t N = decltype(n_)::value;
                    uint64_t k0[N], k1[N], val[N][2];
                    uint32_t row[N];
#pragma unroll
                    for (uint32_t g = 0; g < N; ++g) {
                        k0[g] = pa == 2 ? ldv(W[2 + 2 * g]) : ldv(W[2 + g]);
                        k1[g] = pa == 2 ? ldv(W[3 + 2 * g]) : 0;
                    }
                    const uint32_t dpos = 2 + N * pa;  // destination words (strand form)
                    pc += 2 + N * pa + D * N;
#pragma unroll
                    for (uint32_t g = 0; g < N; ++g) {
                        if constexpr (probe::NO_FIND) row[g] = (uint32_t)k0[g] & 7u;
                        else row[g] = table_find2(t, sc.table_words, k0[g], k1[g]);
                    }
                    // the value gathers of the whole group back to back, unconditionally (a missing key reads row 0 and is zeroed
                    // below): no exec-masked branch and no wait between the members' gathers
                    const bool bytes = (t.dense & 2u) != 0;   // scalar: packed one-byte values (xor8 / and8 / andn8 / byte splits ...)
                    const bool two = nv > 1;
                    uint64_t raw[N][2];
#pragma unroll
                    for (uint32_t g = 0; g < N; ++g) {
                        const uint32_t rr = row[g] < t.n_rows ? row[g] : 0u;
                        if (bytes) {
                            const uint8_t* __restrict__ pv = tb + (size_t)rr * t.n_vals;
                            raw[g][0] = nv ? pv[0] : 0;
                            raw[g][1] = two ? pv[1] : 0;
                        } else {
                            const uint64_t* __restrict__ pv = sc.table_words + (size_t)t.word_off + (size_t)rr * w + t.n_keys;
                            raw[g][0] = nv ? pv[0] : 0;
                            raw[g][1] = two ? pv[1] : 0;
                        }
                    }
#pragma unroll
                    for (uint32_t g = 0; g < N; ++g) {
                        const bool found = row[g] < t.n_rows;
                        val[g][0] = found ? raw[g][0] : 0ull;
                        val[g][1] = found ? raw[g][1] : 0ull;
                    }
#pragma unroll
                    for (uint32_t g = 0; g < N; ++g) {
                        if constexpr (STRANDS) out_to(pa == 2 ? W[(2 + 2 * N + g) & 15] : W[(2 + N + g) & 15]);
                        (void)dpos;
#pragma unroll
                        for (uint32_t i = 0; i < 2; ++i)
                            if (i < nv) stn(val[g][i], g * nv + i);
                        mult_add(sc.mult, (size_t)inst * sc.total_table_rows + t.mult_off + row[g], ZKGL_MULT_ON && row[g] < t.n_rows && active && sc.mult);
                    }
                    // fused mode: the tuple (keys, the values stored here) is a table row iff the keys were found
#pragma unroll
                    for (uint32_t g = 0; g < N; ++g) fused_bad |= row[g] >= t.n_rows;
                };
                switch (grp) {
                case 1: body(GroupSize<1>{}); break;
                case 2: body(GroupSize<2>{}); break;
                case 3: body(GroupSize<3>{}); break;
                default: body(GroupSize<4>{}); break;
                }
            } else {  // wide tuples: one lookup per header
                uint64_t key[3] = {0, 0, 0};
                key[0] = ldv(W[2]);
                if (pa > 1) key[1] = ldv(W[3]);
                if (pa > 2) key[2] = ldv(W[4]);
                if constexpr (STRANDS) out_to(pa == 1 ? W[3] : pa == 2 ? W[4] : W[5]);
                pc += 2 + pa + D;
                const uint32_t row = table_find(t, sc.table_words, key);
                const bool found = row < t.n_rows;
                for (uint32_t i = 0; i < nv; ++i)
                    stn(!found ? 0ull
                        : (t.dense & 2u) ? (uint64_t)tb[(size_t)row * t.n_vals + i]
                                         : sc.table_words[(size_t)t.word_off + (size_t)row * w + t.n_keys + i], i);
                mult_add(sc.mult, (size_t)inst * sc.total_table_rows + t.mult_off + row, ZKGL_MULT_ON && found && active && sc.mult);
                fused_bad |= !found;
            }
        } break;
        case ZK_OP_POSEIDON2:      // witness-only permutation: 12 outputs
        case ZK_OP_P2_ROUNDS: {    // in-circuit permutation: every intermediate the gates constrain, in the order of gadgets.cpp
                                   // compute_round_function; state in LDS so that the S-box loop is not unrolled (I-cache)
            // deferred mode (sc.defer_p2): an in-circuit permutation is computed like a witness-only one and stores its 12 final outputs
            // only, 950 slots further on — nothing in the fused step reads the intermediates; k_fill_p2 regenerates them on demand
            const bool deferred = (op == ZK_OP_P2_ROUNDS) && sc.defer_p2 != 0;
            const bool emit = (op == ZK_OP_P2_ROUNDS) && !deferred;
            uint64_t s[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) s[i] = ldv(W[1 + i]);
            // gated form (pa = 1, witness-only): [.., execute] -> zeros where the flag is off (simulate_round_function(cs, state, execute));
            // a wavefront whose 64 cycles all have it off skips the permutation altogether
            const bool gated = (op == ZK_OP_POSEIDON2) && pa != 0;
            bool lane_off = false;
            if (gated) {
                lane_off = ldv(W[13]) == 0;
                out_to(W[14]);
                pc += 14 + D;
                const bool all_off = __builtin_amdgcn_ballot_w64(!lane_off) == 0;
                if (sc.p2_stats && (threadIdx.x & 63) == 0) atomicAdd(sc.p2_stats + (all_off ? 0 : 1), 1ull);
                if (all_off) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) st(0ull);
                    break;
                }
            }
            else {
                out_to(W[13]);
                pc += 13 + D;
            }
            if (deferred) dst += WIDE ? 950u : 950u * bstep;
            p2::mds_external(s);
            {
                // state in registers, the twelve S-boxes of a full round unrolled, ONE copy of the full-round body (12 KB of code: the
                // fully unrolled permutation of round 1 was 190 KB and instruction-cache-bound, the LDS-staged rolled form that replaced
                // it costs 10 % of the loop kernel against this one; no LDS: a 1 024-thread strand block would need 96 KB for it)
                if (emit) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) st(s[i]);
                }
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {
#pragma unroll 1
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = half * 26 + r4;
#pragma unroll
                        for (int i = 0; i < 12; ++i) {
                            const uint64_t t = gl::add(s[i], p2::RC[12 * r + i]);
                            const uint64_t x2 = probe::NO_P2_SBOX ? t ^ 1 : gl::sqr(t), x3 = probe::NO_P2_SBOX ? t ^ 2 : gl::mul(x2, t), x4 = probe::NO_P2_SBOX ? t ^ 3 : gl::sqr(x2),
                                           x7 = probe::NO_P2_SBOX ? t ^ 4 : gl::mul(x3, x4);
                            if (emit) { st(t); st(x2); st(x3); st(x4); st(x7); }
                            s[i] = x7;
                        }
                        if constexpr (!probe::NO_P2_LINEAR) p2::mds_external(s);
                        if (emit) {
#pragma unroll
                            for (int i = 0; i < 12; ++i) st(s[i]);
                        }
                    }
                    if (half == 0) {
#pragma unroll 1
                        for (int r = 4; r < 26; ++r) {
                            const uint64_t t = gl::add(s[0], p2::RC[12 * r]);
                            const uint64_t x2 = probe::NO_P2_SBOX ? t ^ 1 : gl::sqr(t), x3 = probe::NO_P2_SBOX ? t ^ 2 : gl::mul(x2, t), x4 = probe::NO_P2_SBOX ? t ^ 3 : gl::sqr(x2),
                                           x7 = probe::NO_P2_SBOX ? t ^ 4 : gl::mul(x3, x4);
                            if (emit) { st(t); st(x2); st(x3); st(x4); st(x7); }
                            s[0] = x7;
                            if constexpr (!probe::NO_P2_LINEAR) p2::mds_inner(s);
                            if (emit) {
#pragma unroll
                                for (int i = 0; i < 12; ++i) st(s[i]);
                            }
                        }
                    }
                }
                if (!emit) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) st(lane_off ? 0ull : s[i]);
                }
            }
        } break;
        case ZK_OP_LOOP_LAST: {
            const uint32_t c = W[1];
            out_to(W[2]);
            pc += 2 + D;
            st(sc.loop_cells[cell_off(sc.loop_n_cells, c, lane * sc.loop_limit + (sc.loop_limit - 1))]);
        } break;
        case ZK_OP_U32MULADD: {
            auto body = [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t in[N][4];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) in[g][i] = ldv(W[1 + g * 4 + i]);
                }
                pc += 1 + N * 4 + D * N;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
                    out_to(W[(1 + N * 4 + g) & 15]);
                    const uint64_t r = in[g][0] * in[g][1] + in[g][2] + in[g][3];  // < 2^64 for u32 operands
                    st(r & 0xffffffffull);
                    st(r >> 32);
                }
            };
            switch (pb) {
            case 0: body(GroupSize<1>{}); break;
            case 1: body(GroupSize<2>{}); break;
            default: body(GroupSize<3>{}); break;
            }
        } break;
        case ZK_OP_U8X4FMA: {   // [16 byte slots] -> 10 bytes; header + 16 operands = 17 words: the 17th (and the strand destination) from a second fetch
            uint64_t in[16], out[10];
#pragma unroll
            for (int i = 0; i < 15; ++i) in[i] = ldv(W[1 + i]);
            const uint32_t w16 = prog[pc + 16];
            in[15] = ldv(w16);
            if constexpr (STRANDS) out_to(prog[pc + 17]);
            pc += 17 + D;
            gl::u8x4_fma(in, out);
#pragma unroll
            for (int i = 0; i < 10; ++i) stn(out[i], (uint32_t)i);
        } break;
        case ZK_OP_SHA256_ROUNDS: if constexpr (WITH_BIGINT) {
            // K8: a whole SHA-256 compression as ONE op.  [32 state byte slots, 64 block byte slots] -> every intermediate, in the gadget's
            // allocation order (both walk zks::compress, sha256_macro.hpp).  An input that is not a byte is the fused mode's lookup miss.
            uint32_t wd[24];
            bool not_bytes = false;
#pragma unroll
            for (int c = 0; c < 6; ++c) {   // 16 operand words = four u32 words per scalar fetch
                const u32x16_a4 Wc = *(prog16_ptr)(prog + pc + 1 + 16 * c);
                uint64_t b[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) b[i] = ldv(Wc[i]);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    uint32_t v = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { not_bytes |= b[4 * h + k] > 0xff; v |= ((uint32_t)b[4 * h + k] & 0xffu) << (8 * k); }
                    wd[4 * c + h] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (STRANDS) out_to(prog[pc + 97]);
            pc += 97 + D;
            // (pa = 1, the reference's 4-bit table set, runs on the kernels instantiated with X_SHA4: the host launches those — zkdev::launch_witness*)
            if constexpr (!WIDE && !probe::NO_STORES) {
                const uint32_t n_sh = STRANDS ? (uint32_t)(blockDim.x >> 6) : 1u;
                if constexpr ((XMACROS & X_SHA4) != 0) {
                    if (pa == 1) dst = sha256_rounds4_stream(rsrc, lane_byte, dst, bstep, wd, STRANDS ? uni(threadIdx.x >> 6) : 0u, n_sh - 1);
                    else dst = sha256_rounds_stream(rsrc, lane_byte, dst, bstep, wd, STRANDS ? uni(threadIdx.x >> 6) : 0u, n_sh - 1);
                } else dst = sha256_rounds_stream(rsrc, lane_byte, dst, bstep, wd, STRANDS ? uni(threadIdx.x >> 6) : 0u, n_sh - 1);
            } else {
            bool done4 = false;
            if constexpr ((XMACROS & X_SHA4) != 0) if (pa == 1) {   // (wide scopes / probe builds: every output through st)
                done4 = true;
                auto st1 = [&](uint64_t v) { st(v); };
                struct EmitAll4 {
                    decltype(st1)& f;
                    __device__ __forceinline__ void one(uint64_t v) { f(v); }
                } emit{st1};
                uint32_t sst[8], blk[16], w[64];
                typename zks4::ComputeBackend<EmitAll4>::Splits wsp[64];
#pragma unroll
                for (int i = 0; i < 8; ++i) sst[i] = wd[i];
#pragma unroll
                for (int i = 0; i < 16; ++i) blk[i] = wd[8 + i];
                zks4::ComputeBackend<EmitAll4> be(emit);
                zks4::compress(be, sst, blk, w, wsp, zks::K);
            }
            if (!done4) {
                auto st1 = [&](uint64_t v) { st(v); };
                struct EmitAll {
                    decltype(st1)& f;
                    __device__ __forceinline__ void block(const uint64_t* v, int n) {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (k < n) f(v[k]);
                    }
                } emit{st1};
                uint32_t sst[8], blk[16], w[64];
#pragma unroll
                for (int i = 0; i < 8; ++i) sst[i] = wd[i];
#pragma unroll
                for (int i = 0; i < 16; ++i) blk[i] = wd[8 + i];
                zks::ComputeBackend<EmitAll> be(emit);
                zks::compress(be, sst, blk, w, zks::K);
            }
            }
            fused_bad |= not_bytes;
        } else { return; } break;
        case ZK_OP_KECCAK_F: if constexpr (WITH_BIGINT) {
            // K8: a whole Keccak-f[1600] as ONE op.  [200 state byte slots] -> every intermediate of the byte-table decomposition, in the
            // order the gadget allocated them (both walk zkk::keccak_f, keccak_macro.hpp): the state lives in 25 register pairs, every
            // primitive streams its outputs to the next consecutive store slots.  The lookup tuples on these outputs are table rows iff
            // the 200 inputs are bytes (every later key is a byte computed here): an input >= 256 is the fused mode's lookup miss.
            uint64_t sl[25];   // indexed dynamically below: lives in scratch by design (keccak_macro.hpp)
            bool not_bytes = false;
#pragma unroll 1
            for (uint32_t l = 0; l < 25; ++l) {   // one lane = 8 operand words per step, all eight loads in flight together
                uint64_t b[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) b[k] = ldv(prog[pc + 1 + 8 * l + k]);
                uint64_t v = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) { not_bytes |= b[k] > 0xff; v |= (b[k] & 0xff) << (8 * k); }
                sl[l] = v;
            }
            if constexpr (STRANDS) out_to(prog[pc + 201]);
            pc += 201 + D;
            if constexpr (!WIDE && !probe::NO_STORES) {
                // strand form: the op is in every strand's program (cs.cpp build_strands), strand w stores every (blockDim / 64)-th output
                const uint32_t n_sh = STRANDS ? (uint32_t)(blockDim.x >> 6) : 1u;
                dst = keccak_f_stream(rsrc, lane_byte, dst, bstep, sl, STRANDS ? uni(threadIdx.x >> 6) : 0u, n_sh - 1);
            } else
            {
                auto st1 = [&](uint64_t v) { st(v); };
                struct EmitAll {
                    decltype(st1)& f;
                    __device__ __forceinline__ void block8(uint64_t v) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) f((v >> (8 * k)) & 0xff);
                    }
                    __device__ __forceinline__ void one(uint64_t v) { f(v); }
                } emit{st1};
                zkk::ComputeBackend<EmitAll> be(emit);
                zkk::keccak_f(be, sl, zkk::RC);   // 64-bit addressing (linear_hasher's scope): the rolled walk, state in scratch
            }
            fused_bad |= not_bytes;
        } else { return; } break;
        case ZK_OP_NN_MULMOD: if constexpr (WITH_BIGINT) {
            // fixed layout (cs.cpp emit_scope): 16 modulus limbs (pool indices), 17 A slots, 17 B slots (unused ones 0) = 51 words,
            // four scalar fetches with static word positions
            const u32x16_a4 W1 = *(prog16_ptr)(prog + pc + 16), W2 = *(prog16_ptr)(prog + pc + 32), W3 = *(prog16_ptr)(prog + pc + 48);
            auto word = [&](auto k_) -> uint32_t {
                constexpr uint32_t K = decltype(k_)::value;
                if constexpr (K < 16) return W[K];
                else if constexpr (K < 32) return W1[K - 16];
                else if constexpr (K < 48) return W2[K - 32];
                else return W3[K - 48];
            };
            uint32_t mv[16], av[17], bv[17], res[19 + 16];
            uint64_t mraw[16], araw[17], braw[17];
            // static word positions: the loops below are fully unrolled, `word` needs its index as a compile-time constant
#define ZK_NN_M(I) mraw[I] = cpool[word(GroupSize<1 + I>{})];
#define ZK_NN_AB(I) araw[I] = I < pa ? ldv(word(GroupSize<17 + I>{})) : 0; braw[I] = I < pb ? ldv(word(GroupSize<34 + I>{})) : 0;
            ZK_NN_M(0) ZK_NN_M(1) ZK_NN_M(2) ZK_NN_M(3) ZK_NN_M(4) ZK_NN_M(5) ZK_NN_M(6) ZK_NN_M(7)
            ZK_NN_M(8) ZK_NN_M(9) ZK_NN_M(10) ZK_NN_M(11) ZK_NN_M(12) ZK_NN_M(13) ZK_NN_M(14) ZK_NN_M(15)
            ZK_NN_AB(0) ZK_NN_AB(1) ZK_NN_AB(2) ZK_NN_AB(3) ZK_NN_AB(4) ZK_NN_AB(5) ZK_NN_AB(6) ZK_NN_AB(7) ZK_NN_AB(8)
            ZK_NN_AB(9) ZK_NN_AB(10) ZK_NN_AB(11) ZK_NN_AB(12) ZK_NN_AB(13) ZK_NN_AB(14) ZK_NN_AB(15) ZK_NN_AB(16)
#undef ZK_NN_M
#undef ZK_NN_AB
            for (int i = 0; i < 16; ++i) mv[i] = (uint32_t)mraw[i];
            for (int i = 0; i < 17; ++i) { av[i] = (uint32_t)araw[i]; bv[i] = (uint32_t)braw[i]; }
            out_to(W3[3]);
            pc += 51 + D;
            const uint32_t nq = pa + pb - 15;
            nn_mulmod(av, pa, bv, pb, mv, nq, res);
            for (uint32_t i = 0; i < nq + 16; ++i) st(res[i]);
        } else { return; } break;
        case ZK_OP_DIVREM: {
            const uint64_t x = ldv(W[1]);
            out_to(W[2]);
            pc += 2 + D;
            stn(x / pb, 0);
            stn(x % pb, 1);
        } break;
        case ZK_OP_U256_MULWIDE: {
            // column-wise schoolbook product, a 96-bit column accumulator (up to 8 products of 64 bits + carry)
            const uint32_t w16 = prog[pc + 16];
            uint32_t a[8], b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (uint32_t)ldv(W[1 + i]);
#pragma unroll
            for (int i = 0; i < 7; ++i) b[i] = (uint32_t)ldv(W[9 + i]);
            b[7] = (uint32_t)ldv(w16);
            if constexpr (STRANDS) out_to(prog[pc + 17]);
            pc += 17 + D;
            uint64_t lo = 0;
            uint32_t hi = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int j = k - i;
                    if (j >= 0 && j < 8) {
                        const uint64_t p = (uint64_t)a[i] * b[j];
                        lo += p;
                        hi += lo < p;
                    }
                }
                st((uint64_t)(uint32_t)lo);
                lo = (lo >> 32) | ((uint64_t)hi << 32);
                hi = 0;
            }
        } break;
        case ZK_OP_U256_DIVREM: {
            // restoring shift-subtract division, 256 fixed steps in registers (two per VM cycle: Div and the right shifts)
            const uint32_t w16 = prog[pc + 16];
            uint32_t a[8], b[8], r[8];
            uint32_t bnz = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (uint32_t)ldv(W[1 + i]);
#pragma unroll
            for (int i = 0; i < 7; ++i) b[i] = (uint32_t)ldv(W[9 + i]);
            b[7] = (uint32_t)ldv(w16);
#pragma unroll
            for (int i = 0; i < 8; ++i) { bnz |= b[i]; r[i] = 0; }
            if constexpr (STRANDS) out_to(prog[pc + 17]);
            pc += 17 + D;
            // b == 0: q = 0, r = a (mul_div.rs:96-172).  A wavefront whose lanes all divide by zero (the VM's masked-out Div / Shr
            // cycles) skips the loop; a zero divisor next to real ones walks it with b = 0, which shifts a into r bit by bit
            // (r == a at the end) and fills the quotient with ones, cleared below.
            if (__builtin_amdgcn_ballot_w64(bnz != 0) != 0) {
#pragma unroll 1
                for (int step = 0; step < 256; ++step) {
                    // (r, a) <<= 1 : the quotient bits enter a from the bottom as the dividend bits leave at the top
                    const uint32_t top = r[7] >> 31;  // r < b <= 2^256 - 1 before the shift; a 257-bit r is handled by `top`
#pragma unroll
                    for (int i = 7; i > 0; --i) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
                    r[0] = (r[0] << 1) | (a[7] >> 31);
#pragma unroll
                    for (int i = 7; i > 0; --i) a[i] = (a[i] << 1) | (a[i - 1] >> 31);
                    a[0] <<= 1;
                    uint32_t d[8];
                    uint32_t borrow = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint64_t t = (uint64_t)r[i] - b[i] - borrow;
                        d[i] = (uint32_t)t;
                        borrow = (uint32_t)(t >> 63);
                    }
                    if (top | (borrow ^ 1u)) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) r[i] = d[i];
                        a[0] |= 1u;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = bnz ? a[i] : 0u;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { r[i] = a[i]; a[i] = 0; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) st((uint64_t)a[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) st((uint64_t)r[i]);
        } break;
        case ZK_OP_BARRIER: if constexpr (STRANDS) {
            // end of a dependency level: this strand's stores must be visible to the other wavefronts of the tile (same CU, shared
            // L1) before any of them starts the next level
            pc += 1;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
        } else { return; } break;
        default:
            // ZK_OP_BYTEBUF_FILL exists only in the kernels instantiated with X_BYTEBUF (a `case` of its own would change the jump table — and with it the code —
            // of every other instantiation: tools/isa_diff.py keeps the measured kernels instruction-identical)
            if constexpr (WITH_BIGINT && (XMACROS & X_BYTEBUF) != 0) {
                if (op == ZK_OP_BYTEBUF_FILL) {
            // K8: one ByteBuffer fill as ONE op.  [192 buffer bytes, filled, 32 input bytes, offset, meaningful] -> every intermediate, in the
            // gadget's allocation order (both walk zkb::fill_with_bytes).  The op works on small integers: an operand outside its range
            // (byte > 255, filled > 192, offset > 31, meaningful > 32 — each of them is range-checked by the circuit) is reported as the
            // fused mode's failure, and the values stored for it then violate the op's own gates.
            uint32_t packed[zkb::BUF / 4 + zkb::IN / 4];   // buffer bytes, then input bytes, four per word (passed to the out-of-line walk)
            int32_t sc3[3] = {0, 0, 0};
            bool out_of_range = false;
#pragma unroll 1
            for (uint32_t c8 = 0; c8 < (zkb::N_INPUTS + 7) / 8; ++c8) {   // eight operand loads in flight per step
                uint64_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = c8 * 8 + k < (uint32_t)zkb::N_INPUTS ? ldv(prog[pc + 1 + c8 * 8 + k]) : 0;
                // operand i: [0, 192) buffer bytes, 192 filled, [193, 225) input bytes, 225 offset, 226 meaningful; eight per step, so the
                // words of `packed` are whole steps except around `filled` (step 24 holds filled + 7 input bytes, ...): assemble by position
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t i = c8 * 8 + k;
                    if (i < (uint32_t)zkb::BUF) { out_of_range |= v[k] > 0xff; packed[i >> 2] = (k & 3) ? packed[i >> 2] | (((uint32_t)v[k] & 0xffu) << (8 * (i & 3))) : ((uint32_t)v[k] & 0xffu); }
                    else if (i == (uint32_t)zkb::BUF) { out_of_range |= v[k] > (uint64_t)zkb::BUF; sc3[0] = (int32_t)(v[k] & 0xff); }
                    else if (i < (uint32_t)(zkb::BUF + 1 + zkb::IN)) {
                        const uint32_t m = i - zkb::BUF - 1;
                        out_of_range |= v[k] > 0xff;
                        packed[zkb::BUF / 4 + (m >> 2)] = (m & 3) ? packed[zkb::BUF / 4 + (m >> 2)] | (((uint32_t)v[k] & 0xffu) << (8 * (m & 3))) : ((uint32_t)v[k] & 0xffu);
                    }
                    else if (i == (uint32_t)(zkb::BUF + 1 + zkb::IN)) { out_of_range |= v[k] > 31; sc3[1] = (int32_t)(v[k] & 31); }
                    else if (i == (uint32_t)(zkb::BUF + 2 + zkb::IN)) { out_of_range |= v[k] > 32; sc3[2] = (int32_t)(v[k] & 63); }
                }
            }
            if constexpr (STRANDS) out_to(prog[pc + 1 + zkb::N_INPUTS]);
            pc += 1 + zkb::N_INPUTS + D;
            if constexpr (!WIDE) {
                const uint32_t n_sh = STRANDS ? (uint32_t)(blockDim.x >> 6) : 1u;
                dst = bytebuf_fill_stream(rsrc, lane_byte, dst, bstep, packed, sc3[0], sc3[1], sc3[2], STRANDS ? uni(threadIdx.x >> 6) : 0u, n_sh - 1);
            } else {
                auto st1 = [&](uint64_t v) { st(v); };
                struct EmitAll {
                    decltype(st1)& f;
                    __device__ __forceinline__ void one(uint64_t v) { f(v); }
                } emit{st1};
                BytebufInv inv;
                zkb::ComputeBackend<EmitAll, BytebufInv> be(emit, inv);
#pragma unroll
                for (int k = 0; k < zkb::BUF / 4; ++k) be.bytes_[k] = packed[k];
#pragma unroll
                for (int k = 0; k < zkb::IN / 4; ++k) be.in_[k] = be.sh_[k] = packed[zkb::BUF / 4 + k];
#pragma unroll
                for (int k = 0; k < zkb::BUF / 32; ++k) be.pl_[k] = 0;
                int32_t f = sc3[0];
                zkb::fill_with_bytes(be, f, sc3[1], sc3[2]);
            }
            fused_bad |= out_of_range;
                            break;
                }
            }
            return;  // malformed program: host validates before upload
        }
    }
    if (sc.fail && fused_bad && active) report_fused(sc.fail, lane);
}

template <bool WITH_BIGINT, bool WIDE, bool NARROW = false, int XMACROS = 0>
__device__ __forceinline__ void witness_entry2(const ScopeDev& sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin, const uint32_t* cls_words = nullptr) {
    uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if ((blockIdx.x * TPB + (threadIdx.x & ~63u)) >= sc.n_lanes) return;  // whole wave out of range
    const bool active = lane < sc.n_lanes;
    lane = active ? lane : sc.n_lanes - 1;
    // clock probe (loop launch of resolve_and_check): the first wavefront of the grid reads the shader clock counter (s_memtime) and the
    // constant 100 MHz counter (s_memrealtime) around its own run — about half of the launch — so that the host can tell the
    // shader clock the chip's power management gave THIS launch (profiles/r3_loop_probe.md §4)
    const bool probe = sc.clock_probe && blockIdx.x == 0 && threadIdx.x < 64;
    uint64_t t0 = 0, r0 = 0;
    if (probe) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    run_tile2<WITH_BIGINT, WIDE, TPB, false, NARROW, XMACROS>(sc, lane, sc.is_loop ? lane / sc.limit : lane, active, word_begin, word_end, slot_begin, cls_words);
    if (probe && threadIdx.x == 0) {
        sc.clock_probe[0] = __builtin_readcyclecounter() - t0;
        sc.clock_probe[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
// Strand mode: a scope with too few lanes to fill the chip (hash circuits: lanes = instances x cycles; every outer scope: lanes =
// instances) runs one 64-lane tile per BLOCK of 8 wavefronts.  Wavefront w walks strand w of the program: the ops of every
// dependency level of the op graph are dealt out over the strands by the host (cs.cpp build_strands), ZK_OP_BARRIER between levels.
template <bool WITH_BIGINT, bool WIDE, int XMACROS = 0>
__global__ __launch_bounds__(64 * STRANDS_PER_TILE) void k_witness_strands2(ScopeDev sc, StrandTab tab) {
    if (blockIdx.x * 64 >= sc.n_lanes) return;
    const uint32_t w = uni(threadIdx.x >> 6);
    uint32_t lane = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool active = lane < sc.n_lanes;
    lane = active ? lane : sc.n_lanes - 1;
    run_tile2<WITH_BIGINT, WIDE, 64 * STRANDS_PER_TILE, true, false, XMACROS>(sc, lane, sc.is_loop ? lane / sc.limit : lane, active, tab.begin[w], tab.end[w], 0);
}

// Separate symbols so that profiles separate the loop-scope launch (the dominant kernel: B * limit lanes) from the
// outer-scope launches (B lanes, latency-bound).
// Occupancy (measured at B=384 on one box, profiles/r2_summary.md): capped by LDS padding 2 / 3 / 4 wavefronts per SIMD = 58.8 / 50.0 /
// 46.4 ms, the natural 6 (77 VGPRs) 41.4, 7 (72 VGPRs, 12 B of scratch) 40.1, 8 (64 VGPRs, 60 B of scratch) 41.7.
#ifndef ZKGL_LOOP_WAVES2
#define ZKGL_LOOP_WAVES2 7
#endif
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(ZKGL_LOOP_WAVES2, 8))) void k_witness_loop(ScopeDev sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin) {
    witness_entry2<false, false>(sc, word_begin, word_end, slot_begin);
}
// the loop scope over a NARROW store (store_geom.hpp): address-word operands, byte-class outputs in one-byte slots; same program order, same values
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(ZKGL_LOOP_WAVES2, 8))) void k_witness_loop_narrow(ScopeDev sc, const uint32_t* cls_words, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin) {
    witness_entry2<false, false, true>(sc, word_begin, word_end, slot_begin, cls_words);   // cls_words: one class word per header (bit k: the k-th output is byte-class)
}
__global__ __launch_bounds__(TPB) void k_witness_outer(ScopeDev sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin) {
    witness_entry2<false, false>(sc, word_begin, word_end, slot_begin);
}
// scopes with >= 2^23 store slots per lane: 64-bit addressing
__global__ __launch_bounds__(TPB) void k_witness_wide(ScopeDev sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin) {
    witness_entry2<true, true>(sc, word_begin, word_end, slot_begin);
}
__global__ __launch_bounds__(TPB) void k_witness_loop_bigint(ScopeDev sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin) {
    witness_entry2<true, false>(sc, word_begin, word_end, slot_begin);
}
__global__ __launch_bounds__(TPB) void k_witness_outer_bigint(ScopeDev sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin) {
    witness_entry2<true, false>(sc, word_begin, word_end, slot_begin);
}
// plain form (loop or outer scope) of the circuits that record a macro-op beyond the basic set (XMACROS: X_SHA4 / X_BYTEBUF); their strand form is
// k_witness_strands2<true, false, XMACROS>
template <int XMACROS>
__global__ __launch_bounds__(TPB) void k_witness_plain_x(ScopeDev sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin) {
    witness_entry2<true, false, false, XMACROS>(sc, word_begin, word_end, slot_begin);
}


// ------------------------------------------------------------------------------------------------------------------------
// Cone seeding, scalar-decoded (k_seed_cone_strands2).  Same strand program words as kernels_engine.hpp k_seed_cone_strands
// (header, operands, destination slots; values in an LDS slot store), but fetched by the scalar unit — the seeding loop is a
// latency chain of limit x levels steps for ONE wavefront per strand, so what counts is the number of dependent instructions
// per op: 2 384 cycles x 136 levels of main_vm took 2.77 s through the readlane-window interpreter, 0.49 s of it Poseidon2.
// Ops without a handler here (hash macro-ops, NN_MULMOD) keep the v1 kernel (cs.cpp seed_v2_ok_).
// ------------------------------------------------------------------------------------------------------------------------
template <int BLOCK>
__device__ __forceinline__ void run_seed2(const ScopeDev& sc, const uint32_t inst, uint32_t word_begin, uint32_t word_end, const uint32_t* seed_prog,
                                          uint64_t* slots, const uint32_t stride, const uint64_t* in_area, uint64_t* prof = nullptr) {
    const prog1_ptr prog = (prog1_ptr)(uintptr_t)seed_prog;
    const cpool_ptr cpool = (cpool_ptr)(uintptr_t)sc.consts;
    auto ldv = [&](uint32_t slot) -> uint64_t { return slots[slot * stride]; };
    auto stv = [&](uint32_t slot, uint64_t v) { slots[slot * stride] = v; };
    // The op stream is a dependent chain for this one wavefront, so the NEXT op's words are fetched before the current op runs:
    // its length follows from the header alone (fixed per opcode, SPLIT / LOOKUP add their counts).  Lengths, 6 bits per opcode:
    constexpr uint64_t LEN_A = 0ull | (3ull << 6) | (3ull << 12) | (7ull << 18) | (10ull << 24) | (5ull << 30) | (4ull << 36) | (6ull << 42) | (6ull << 48) | (10ull << 54);  // ops 0..9
    constexpr uint64_t LEN_B = 25ull | (2ull << 6) | (2ull << 12) | (25ull << 18) | (25ull << 24) | (0ull << 30) | (7ull << 36) | (0ull << 42) | (4ull << 48) | (0ull << 54);  // ops 10..19
    constexpr uint64_t LEN_C = 0ull | (0ull << 6) | (1ull << 12) | (33ull << 18) | (33ull << 24);  // ops 20..24
    uint32_t pc = word_begin;
    u32x16_a4 Wn = *(prog16_ptr)(prog + pc);
    while (pc < word_end) {
        const u32x16_a4 W = Wn;
        const uint32_t h = W[0];
        const uint32_t op = h & 0xff, pa = (h >> 8) & 0xff, pb = h >> 16;
        {
            const uint64_t tab = op < 10 ? LEN_A : op < 20 ? LEN_B : LEN_C;
            const uint32_t sub = op < 10 ? op : op < 20 ? op - 10 : op - 20;
            uint32_t len = (uint32_t)(tab >> (6 * sub)) & 63u;
            if (op == ZK_OP_SPLIT) len += pa;
            if (op == ZK_OP_LOOKUP) len += pa + (pb & 0xff);
            Wn = *(prog16_ptr)(prog + pc + len);
        }
        switch (op) {
        case ZK_OP_BARRIER:
            pc += 1;
            __syncthreads();  // LDS only: no store drain needed beyond the barrier's own lgkmcnt wait
            break;
        case ZK_OP_CONST: {
            const uint32_t w = W[1];
            stv(W[2], (w & ZK_OPERAND_KIND_MASK) == ZK_OPERAND_OUTER ? sc.outer_cells[cell_off(sc.outer_n_cells, w & ZK_OPERAND_IDX_MASK, inst)]
                                                                      : (uint64_t)cpool[w & ZK_OPERAND_IDX_MASK]);
            pc += 3;
        } break;
        case ZK_OP_INPUT:
            stv(W[2], in_area[W[1] * stride]);  // staged in LDS by the kernel prologue
            pc += 3;
            break;
        case ZK_OP_FMA: {
            const uint64_t a = ldv(W[3]), b = ldv(W[4]), c = ldv(W[5]);
            const uint64_t q = cpool[W[1] & ZK_OPERAND_IDX_MASK], l = cpool[W[2] & ZK_OPERAND_IDX_MASK];
            const uint64_t ab = gl::mul(a, b);
            stv(W[6], gl::add(q == 1 ? ab : gl::mul(q, ab), l == 1 ? c : gl::mul(l, c)));
            pc += 7;
        } break;
        case ZK_OP_LC4: {
            uint64_t t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = ldv(W[5 + i]);
            uint64_t r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r = gl::fma(cpool[W[1 + i] & ZK_OPERAND_IDX_MASK], t[i], r);
            stv(W[9], r);
            pc += 10;
        } break;
        case ZK_OP_SELECT: {
            const uint64_t sl = ldv(W[1]), a = ldv(W[2]), b = ldv(W[3]);
            stv(W[4], sl ? a : b);
            pc += 5;
        } break;
        case ZK_OP_ISZERO: {
            const uint64_t x = ldv(W[1]);
            stv(W[2], x == 0 ? 1ull : 0ull);
            stv(W[3], p2::inv_wave(x));
            pc += 4;
        } break;
        case ZK_OP_UADD: {
            const uint64_t sum = ldv(W[1]) + ldv(W[2]) + ldv(W[3]);
            stv(W[4], sum & ((1ull << pa) - 1));
            stv(W[5], sum >> pa);
            pc += 6;
        } break;
        case ZK_OP_USUB: {
            const uint64_t x = ldv(W[1]), sub = ldv(W[2]) + ldv(W[3]);
            const uint64_t borrow = x < sub ? 1 : 0;
            stv(W[4], (x + (borrow << pa)) - sub);
            stv(W[5], borrow);
            pc += 6;
        } break;
        case ZK_OP_DOT4: {
            uint64_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ldv(W[1 + i]);
            uint64_t r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r = gl::fma(v[2 * i], v[2 * i + 1], r);
            stv(W[9], r);
            pc += 10;
        } break;
        case ZK_OP_MATMUL12: {
            const u32x16_a4 W1 = *(prog16_ptr)(prog + pc + 16);
            uint64_t st[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) st[i] = ldv(W[1 + i]);
            if (pa == 0) p2::mds_external(st); else p2::mds_inner(st);
#pragma unroll
            for (int i = 0; i < 3; ++i) stv(W[13 + i], st[i]);
#pragma unroll
            for (int i = 3; i < 12; ++i) stv(W1[i - 3], st[i]);
            pc += 25;
        } break;
        case ZK_OP_SPLIT: {
            uint64_t x = ldv(W[1]);
            for (uint32_t i = 0; i < pa; ++i) {
                stv(prog[pc + 2 + i], i + 1 == pa ? x : (x & ((1ull << pb) - 1)));
                x >>= pb;
            }
            pc += 2 + pa;
        } break;
        case ZK_OP_LOOKUP: {
            const zk_table_desc t = load_table_desc(sc.tables, W[1]);
            const uint32_t nv = pb & 0xff;
            const uint64_t k0 = ldv(W[2]), k1 = pa > 1 ? ldv(W[3]) : 0, k2 = pa > 2 ? ldv(W[4]) : 0;
            const uint32_t row = table_find3(t, sc.table_words, k0, k1, k2);
            const bool found = row < t.n_rows;
            const uint8_t* __restrict__ tb = reinterpret_cast<const uint8_t*>(sc.table_words + (t.dense >> 2));
            const uint32_t w = t.n_keys + t.n_vals;
            for (uint32_t i = 0; i < nv; ++i)
                stv(prog[pc + 2 + pa + i], !found ? 0ull
                                           : (t.dense & 2u) ? (uint64_t)tb[(size_t)row * t.n_vals + i]
                                                            : sc.table_words[(size_t)t.word_off + (size_t)row * w + t.n_keys + i]);
            pc += 2 + pa + nv;
        } break;
        case ZK_OP_POSEIDON2: {
            // The permutation is a dependent chain for this one wavefront (six of them in a row per main_vm cycle): state in
            // registers, the twelve S-boxes of a full round unrolled — the LDS-staged form of the throughput kernels costs twice
            // the latency here (143 k vs ~70 k clock ticks per permutation, in-kernel profile).  One copy of the full-round body.
            const u32x16_a4 W1 = *(prog16_ptr)(prog + pc + 16);
            uint64_t st[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) st[i] = ldv(W[1 + i]);
            p2::mds_external(st);
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
#pragma unroll 1
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = half * 26 + r4;
#pragma unroll
                    for (int i = 0; i < 12; ++i) st[i] = gl::pow7(gl::add(st[i], p2::RC[12 * r + i]));
                    p2::mds_external(st);
                }
                if (half == 0) {
#pragma unroll 1
                    for (int r = 4; r < 26; ++r) {
                        st[0] = gl::pow7(gl::add(st[0], p2::RC[12 * r]));
                        p2::mds_inner(st);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) stv(W[13 + i], st[i]);
#pragma unroll
            for (int i = 3; i < 12; ++i) stv(W1[i - 3], st[i]);
            pc += 25;
        } break;
        case ZK_OP_U32MULADD: {
            const uint64_t r = ldv(W[1]) * ldv(W[2]) + ldv(W[3]) + ldv(W[4]);
            stv(W[5], r & 0xffffffffull);
            stv(W[6], r >> 32);
            pc += 7;
        } break;
        case ZK_OP_DIVREM: {
            const uint64_t x = ldv(W[1]);
            stv(W[2], x / pb);
            stv(W[3], x % pb);
            pc += 4;
        } break;
        case ZK_OP_U256_MULWIDE: {
            const u32x16_a4 W1 = *(prog16_ptr)(prog + pc + 16), W2 = *(prog16_ptr)(prog + pc + 32);
            uint32_t a[8], b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (uint32_t)ldv(W[1 + i]);
#pragma unroll
            for (int i = 0; i < 7; ++i) b[i] = (uint32_t)ldv(W[9 + i]);
            b[7] = (uint32_t)ldv(W1[0]);
            uint64_t lo = 0;
            uint32_t hi = 0, out[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int j = k - i;
                    if (j >= 0 && j < 8) {
                        const uint64_t p = (uint64_t)a[i] * b[j];
                        lo += p;
                        hi += lo < p;
                    }
                }
                out[k] = (uint32_t)lo;
                lo = (lo >> 32) | ((uint64_t)hi << 32);
                hi = 0;
            }
#pragma unroll
            for (int i = 0; i < 15; ++i) stv(W1[1 + i], (uint64_t)out[i]);
            stv(W2[0], (uint64_t)out[15]);
            pc += 33;
        } break;
        case ZK_OP_U256_DIVREM: {
            const u32x16_a4 W1 = *(prog16_ptr)(prog + pc + 16), W2 = *(prog16_ptr)(prog + pc + 32);
            uint32_t a[8], b[8], r[8];
            uint32_t bnz = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (uint32_t)ldv(W[1 + i]);
#pragma unroll
            for (int i = 0; i < 7; ++i) b[i] = (uint32_t)ldv(W[9 + i]);
            b[7] = (uint32_t)ldv(W1[0]);
#pragma unroll
            for (int i = 0; i < 8; ++i) { bnz |= b[i]; r[i] = 0; }
            if (__builtin_amdgcn_ballot_w64(bnz != 0) != 0) {  // see run_tile2
#pragma unroll 1
                for (int step = 0; step < 256; ++step) {
                    const uint32_t top = r[7] >> 31;
#pragma unroll
                    for (int i = 7; i > 0; --i) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
                    r[0] = (r[0] << 1) | (a[7] >> 31);
#pragma unroll
                    for (int i = 7; i > 0; --i) a[i] = (a[i] << 1) | (a[i - 1] >> 31);
                    a[0] <<= 1;
                    uint32_t d[8];
                    uint32_t borrow = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint64_t t = (uint64_t)r[i] - b[i] - borrow;
                        d[i] = (uint32_t)t;
                        borrow = (uint32_t)(t >> 63);
                    }
                    if (top | (borrow ^ 1u)) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) r[i] = d[i];
                        a[0] |= 1u;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = bnz ? a[i] : 0u;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { r[i] = a[i]; a[i] = 0; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) stv(W1[1 + i], (uint64_t)a[i]);
#pragma unroll
            for (int i = 0; i < 7; ++i) stv(W1[9 + i], (uint64_t)r[i]);
            stv(W2[0], (uint64_t)r[7]);
            pc += 33;
        } break;
        default:
            return;  // not a seed-v2 op: the host never selects this kernel for such cones
        }
    }
}

__global__ __launch_bounds__(64 * SEED_STRANDS_PER_TILE) void k_seed_cone_strands2(ScopeDev sc, const uint32_t* __restrict__ seed_sprog, StrandTab tab,
                                                                             const SeedCarryDev* carries, uint32_t n_carries, uint64_t* inputs_rw,
                                                                             uint32_t n_instances, uint32_t lpb, uint32_t n_slots, uint32_t n_input_words) {
    constexpr uint32_t NT = 64 * SEED_STRANDS_PER_TILE;
    __shared__ uint64_t lds[SEED_LDS_WORDS];
    if (blockIdx.x * lpb >= n_instances) return;
    uint64_t* const slot_store = lds;                 // [n_slots][lpb]
    uint64_t* const in_store = lds + n_slots * lpb;   // [n_input_words][lpb]: this iteration's input stream words
    const uint32_t w = uni(threadIdx.x >> 6);
    const uint32_t l = (threadIdx.x & 63) % lpb;      // lanes >= lpb mirror lane % lpb: identical values, benign duplicate stores
    const uint32_t inst = min(blockIdx.x * lpb + l, n_instances - 1);
    const uint32_t ml = inst - blockIdx.x * lpb;
    const uint32_t wb = tab.begin[w], we = tab.end[w];
    for (uint32_t k = 0; k < sc.limit; ++k) {
        for (uint32_t idx = threadIdx.x; idx < n_input_words * lpb; idx += NT) {
            const uint32_t wd = idx / lpb, ll = idx % lpb;
            const uint32_t li = min(blockIdx.x * lpb + ll, n_instances - 1);
            in_store[wd * lpb + li - blockIdx.x * lpb] = inputs_rw[(size_t)wd * sc.in_stride + (size_t)li * sc.limit + k];
        }
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < n_carries * lpb; idx += NT) {
            const SeedCarryDev cd = carries[idx / lpb];
            const uint32_t ll = idx % lpb;
            const uint32_t li = min(blockIdx.x * lpb + ll, n_instances - 1), col = li - blockIdx.x * lpb;
            if (k == 0 && !cd.has_first) continue;
            const uint64_t v = k == 0 ? sc.outer_cells[cell_off(sc.outer_n_cells, cd.first_outer_cell, li)] : slot_store[cd.out_slot * lpb + col];
            in_store[cd.word * lpb + col] = v;
            inputs_rw[(size_t)cd.word * sc.in_stride + (size_t)li * sc.limit + k] = v;
        }
        __syncthreads();
        run_seed2<(int)NT>(sc, inst, wb, we, seed_sprog, slot_store + ml, lpb, in_store + ml);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Gate / lookup checker over the variable store, scalar-decoded like the witness interpreter ("check program", cs.cpp
// build_check_program).  The row-descriptor checker (kernels_engine.hpp check_gates_body) reads, per gate instance, the row
// descriptor, then one alias word per column, then the value — three dependent round trips, 87 % of its wave time in
// s_waitcnt (profiles/r2_pmc_vm_v2_before_mult_fix.txt).  Here a packet carries everything in one scalar fetch:
//   w0 = kind | count << 8 | first instance << 16, w1 = row (for the failure report), w2 = offset of the row's constants,
//   then the store slots of the columns of `count` consecutive instances of the row (lookup packets: kind 0x40, w2 = table
//   id, count tuples of n_keys + n_vals slots).  All loads of a packet are issued back to back; a packet never exceeds the
//   16-word fetch except the 24-column matrix gates (second fetch).
// The program is cut into chunks of whole packets (chunk_tab); a workgroup checks 256 lanes x chunks_per_block chunks.
// ------------------------------------------------------------------------------------------------------------------------
struct CheckProgDev {
    const uint64_t* cells; uint64_t n_cells; uint32_t n_lanes;
    const uint32_t* prog; const uint32_t* chunk_tab; uint32_t n_chunks, chunks_per_block;
    const uint64_t* rowconsts; const zk_table_desc* tables; const uint64_t* table_words;
    unsigned long long* fail;
};
constexpr uint32_t ZK_CHECK_LOOKUP = 0x40, ZK_CHECK_P2 = 0x41;

template <uint32_t N, class F>
__device__ __forceinline__ void dispatch_count(uint32_t n, F&& f) {
    if constexpr (N == 1) f(GroupSize<1>{});
    else { if (n >= N) f(GroupSize<N>{}); else dispatch_count<N - 1>(n, f); }
}

#ifndef ZKGL_CHECK_WAVES
#define ZKGL_CHECK_WAVES 4
#endif
// NARROW: cd.cells is a narrow store (store_geom.hpp) and the packets carry address words (cs.cpp build_narrow_layout) — same packets, same relations
template <bool NARROW>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(ZKGL_CHECK_WAVES, 8))) void k_check_prog_t(CheckProgDev cd) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if ((blockIdx.x * TPB + (threadIdx.x & ~63u)) >= cd.n_lanes) return;
    const bool active = lane < cd.n_lanes;
    lane = active ? lane : cd.n_lanes - 1;
    const TileAddr ta = tile_addr(const_cast<uint64_t*>(cd.cells), cd.n_cells, lane);
    const uint32_t lane_byte = ta.lane_byte, bsh = uni(ta.shift);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ta.base, 0, -1, 0x00020000);
    const prog1_ptr prog = (prog1_ptr)(uintptr_t)cd.prog;
    const prog1_ptr tab = (prog1_ptr)(uintptr_t)cd.chunk_tab;
    const cpool_ptr consts = (cpool_ptr)(uintptr_t)cd.rowconsts;
    const uint32_t c0 = blockIdx.y * cd.chunks_per_block, c1 = min(c0 + cd.chunks_per_block, cd.n_chunks);
    uint32_t pc = tab[c0];
    const uint32_t end = tab[c1];
    auto ldv = [&](uint32_t slot) -> uint64_t {
        if constexpr (NARROW) {
            if (slot & zkgeom::AW_BYTE) return (uint64_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, lane_byte >> 3, (slot & zkgeom::AW_MASK) << (bsh - 3), 0);
            u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_byte, slot << (bsh - 3), 0);
            return (uint64_t)v.x | ((uint64_t)v.y << 32);
        }
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_byte, slot << bsh, 0);
        return (uint64_t)v.x | ((uint64_t)v.y << 32);
    };
    while (pc < end) {
        const u32x16_a4 W = *(prog16_ptr)(prog + pc);
        const uint32_t kind = W[0] & 0xff, cnt = (W[0] >> 8) & 0xff, j0 = W[0] >> 16, slot = W[1];
        const cpool_ptr k = consts + W[2];
        auto bad = [&](bool cond, uint32_t j, uint32_t rel) { if (cond && active) report(cd.fail, lane, slot, j, rel); };
        switch (kind) {
        case ZK_GATE_CONST: {
            dispatch_count<8>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) v[g] = ldv(W[3 + g]);
                pc += 3 + N;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) bad(v[g] != k[j0 + g], j0 + g, 0);  // instance j is bound to constant j
            });
        } break;
        case ZK_GATE_BOOLEAN: {
            dispatch_count<8>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) v[g] = ldv(W[3 + g]);
                pc += 3 + N;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) bad(v[g] > 1, j0 + g, 0);  // v^2 == v over a field: v in {0, 1}
            });
        } break;
        case ZK_GATE_FMA: {
            dispatch_count<3>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N][4];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) v[g][i] = ldv(W[3 + g * 4 + i]);
                pc += 3 + N * 4;
                const uint64_t q = k[0], l = k[1];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
                    const uint64_t ab = gl::mul(v[g][0], v[g][1]);
                    bad(gl::add(q == 1 ? ab : gl::mul(q, ab), l == 1 ? v[g][2] : gl::mul(l, v[g][2])) != v[g][3], j0 + g, 0);
                }
            });
        } break;
        case ZK_GATE_REDUCTION4: {
            dispatch_count<2>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N][5];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
#pragma unroll
                    for (uint32_t i = 0; i < 5; ++i) v[g][i] = ldv(W[3 + g * 5 + i]);
                pc += 3 + N * 5;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
                    uint64_t r = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) r = gl::fma(k[i], v[g][i], r);
                    bad(r != v[g][4], j0 + g, 0);
                }
            });
        } break;
        case ZK_GATE_SELECT: {
            dispatch_count<3>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N][4];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) v[g][i] = ldv(W[3 + g * 4 + i]);
                pc += 3 + N * 4;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)  // a, b, s, r : s * (a - b) + b == r
                    bad(gl::add(gl::mul(v[g][2], gl::sub(v[g][0], v[g][1])), v[g][1]) != v[g][3], j0 + g, 0);
            });
        } break;
        case ZK_GATE_ZEROCHECK: {
            dispatch_count<4>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N][3];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
#pragma unroll
                    for (uint32_t i = 0; i < 3; ++i) v[g][i] = ldv(W[3 + g * 3 + i]);
                pc += 3 + N * 3;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
                    bad(gl::mul(v[g][0], v[g][1]) != gl::sub(1, v[g][2]), j0 + g, 0);
                    bad(gl::mul(v[g][0], v[g][2]) != 0, j0 + g, 1);
                }
            });
        } break;
        case ZK_GATE_UINTX_ADD: {
            dispatch_count<2>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N][5];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
#pragma unroll
                    for (uint32_t i = 0; i < 5; ++i) v[g][i] = ldv(W[3 + g * 5 + i]);
                pc += 3 + N * 5;
                const uint64_t shift = k[0];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
                    bad(gl::add(gl::add(v[g][0], v[g][1]), v[g][2]) != gl::add(v[g][3], gl::mul(shift, v[g][4])), j0 + g, 0);
            });
        } break;
        case ZK_GATE_DOT4: {
            uint64_t v[9];
#pragma unroll
            for (uint32_t i = 0; i < 9; ++i) v[i] = ldv(W[3 + i]);
            pc += 12;
            uint64_t r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r = gl::fma(v[2 * i], v[2 * i + 1], r);
            bad(r != v[8], j0, 0);
        } break;
        case ZK_GATE_MATMUL12_EXT:
        case ZK_GATE_MATMUL12_INT: {
            const u32x16_a4 W2 = *(prog16_ptr)(prog + pc + 16);
            uint64_t s[12], o[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) s[i] = ldv(W[3 + i]);
            o[0] = ldv(W[15]);
#pragma unroll
            for (int i = 1; i < 12; ++i) o[i] = ldv(W2[i - 1]);
            pc += 27;
            if (kind == ZK_GATE_MATMUL12_EXT) p2::mds_external(s); else p2::mds_inner(s);
#pragma unroll
            for (int i = 0; i < 12; ++i) bad(s[i] != o[i], j0, i);
        } break;
        case ZK_GATE_U32_FMA: {
            dispatch_count<2>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N][6];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
#pragma unroll
                    for (uint32_t i = 0; i < 6; ++i) v[g][i] = ldv(W[3 + g * 6 + i]);
                pc += 3 + N * 6;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
                    bad(gl::add(gl::add(gl::mul(v[g][0], v[g][1]), v[g][2]), v[g][3]) != gl::add(v[g][4], gl::mul(v[g][5], 1ull << 32)), j0 + g, 0);
            });
        } break;
        case ZK_GATE_U8X4_FMA: {   // 3 header words + 26 slots: two 16-word fetches
            const u32x16_a4 W2 = *(prog16_ptr)(prog + pc + 16);
            uint64_t v[26], r0, r1;
#pragma unroll
            for (int i = 0; i < 13; ++i) v[i] = ldv(W[3 + i]);
#pragma unroll
            for (int i = 13; i < 26; ++i) v[i] = ldv(W2[i - 13]);
            pc += 29;
            gl::u8x4_relations(v, r0, r1);
            bad(r0 != 0, j0, 0);
            bad(r1 != 0, j0, 1);
        } break;
        case ZK_GATE_REDUCTION_BY_POWERS4: {
            dispatch_count<2>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t v[N][5];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
#pragma unroll
                    for (uint32_t i = 0; i < 5; ++i) v[g][i] = ldv(W[3 + g * 5 + i]);
                pc += 3 + N * 5;
                const uint64_t c = k[0];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {  // Horner in the row constant c
                    uint64_t r = gl::fma(v[g][3], c, v[g][2]);
                    r = gl::fma(r, c, v[g][1]);
                    r = gl::fma(r, c, v[g][0]);
                    bad(r != v[g][4], j0 + g, 0);
                }
            });
        } break;
        case ZK_CHECK_LOOKUP: {  // w2 = table id; cnt tuples of n_keys + n_vals (<= 4) slots each, (keys.., values..) must be a table row
            const zk_table_desc t = load_table_desc(cd.tables, W[2]);
            const uint32_t nk = t.n_keys, nv = t.n_vals, tw = nk + nv;
            const uint8_t* __restrict__ tb = reinterpret_cast<const uint8_t*>(cd.table_words + (t.dense >> 2));
            dispatch_count<3>(cnt, [&](auto n_) {
                constexpr uint32_t N = decltype(n_)::value;
                uint64_t x[N][4];
#pragma unroll
                for (uint32_t g = 0; g < N; ++g)
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) x[g][i] = i < tw ? ldv(W[3 + g * 4 + i]) : 0;
                pc += 3 + N * 4;
#pragma unroll
                for (uint32_t g = 0; g < N; ++g) {
                    const uint32_t row = table_find3(t, cd.table_words, x[g][0], nk > 1 ? x[g][1] : 0, nk > 2 ? x[g][2] : 0);
                    bool ok = row < t.n_rows;
#pragma unroll
                    for (uint32_t i = 0; i < 3; ++i)
                        if (ok && i < nv) {
                            const uint64_t have = nk == 1 ? x[g][(1 + i) & 3] : nk == 2 ? x[g][(2 + i) & 3] : x[g][3];
                            const uint64_t want = (t.dense & 2u) ? (uint64_t)tb[(size_t)row * nv + i]
                                                                 : cd.table_words[(size_t)t.word_off + (size_t)row * tw + nk + i];
                            ok = have == want;
                        }
                    bad(!ok, 0x80 | (j0 + g), 15);
                }
            });
        } break;
        default:
            return;  // malformed program: built by the host
        }
    }
}
// (one kernel template, not a body function called from two kernels: wrapped, the same source got another register allocation — 125 VGPRs for 123 — which
//  tools/isa_diff.py showed; k_check_prog_t<false> is instruction for instruction the k_check_prog a device has measured)

// narrow store -> ordinary store (store_geom.hpp; CS::ensure_p2_filled): every reader outside the fused step sees 8-byte slots.  aw[slot] = address
// word of the slot's value.  A block = 256 lanes x one chunk of slots: reads 64 B or 512 B per wavefront and value, writes 512 B.
__global__ __launch_bounds__(TPB) void k_widen_store(const uint64_t* __restrict__ narrow, uint64_t narrow_geom, uint64_t* __restrict__ wide, uint64_t wide_geom,
                                                     uint32_t n_lanes, const uint32_t* __restrict__ aw, uint32_t n_slots, uint32_t slots_per_chunk) {
    const uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if (lane >= n_lanes) return;
    const uint32_t s0 = blockIdx.y * slots_per_chunk, s1 = min(s0 + slots_per_chunk, n_slots);
    const prog1_ptr awp = (prog1_ptr)(uintptr_t)aw;
    uint64_t* __restrict__ out = wide + cell_off(wide_geom, 0, lane);
    const uint32_t tsh = zkgeom::tile_log2(wide_geom);
    for (uint32_t slot = s0; slot < s1; ++slot) out[(size_t)slot << tsh] = load_value(narrow, narrow_geom, awp[slot], lane);
}
// the same for a LIST of slots at the last iteration of every instance: what ZK_OP_LOOP_LAST of the outer post phase reads (thread = (instance, list entry))
__global__ void k_widen_last(const uint64_t* __restrict__ narrow, uint64_t narrow_geom, uint64_t* __restrict__ wide, uint64_t wide_geom,
                             uint32_t n_instances, uint32_t limit, const uint32_t* __restrict__ aw, const uint32_t* __restrict__ slots, uint32_t n_list) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_instances * n_list) return;
    const uint32_t inst = (uint32_t)(t % n_instances), slot = slots[t / n_instances];
    const uint32_t lane = inst * limit + (limit - 1);
    wide[cell_off(wide_geom, slot, lane)] = load_value(narrow, narrow_geom, aw[slot], lane);
}

// ------------------------------------------------------------------------------------------------------------------------
// In-circuit Poseidon2 permutations, checked as MACRO packets (k_check_p2; cs.cpp build_check_program).  gadgets.cpp
// compute_round_function constrains the 962 outputs of a ZK_OP_P2_ROUNDS op with 31 matrix gates and 590 FMA gates: 3 104 value
// references to 974 distinct values that the gate-by-gate program meets in three different rows.  Here every stored value is loaded
// ONCE, in order (the outputs are consecutive slots), and the same 621 relations — t = x + rc, x2 = t t, x3 = x2 t, x4 = x2 x2,
// x7 = x3 x4, out = M in — are evaluated on the STORED operands.  A kernel of its own: the unrolled rounds are 40 KB of code and want
// their own register budget.  A failure is reported under row 0xffffe; the host then runs the gate-by-gate program to name the gate.
// descriptor = 14 words: 12 input slots, slot of the first output, row of the first gate.
// ------------------------------------------------------------------------------------------------------------------------
struct CheckP2Dev { const uint64_t* cells; uint64_t n_cells; uint32_t n_lanes; const uint32_t* macros; uint32_t n_macros, per_block; unsigned long long* fail; };
__global__ __launch_bounds__(TPB) void k_check_p2(CheckP2Dev cd) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if ((blockIdx.x * TPB + (threadIdx.x & ~63u)) >= cd.n_lanes) return;
    const bool active = lane < cd.n_lanes;
    lane = active ? lane : cd.n_lanes - 1;
    const TileAddr ta = tile_addr(const_cast<uint64_t*>(cd.cells), cd.n_cells, lane);
    const uint32_t lane_byte = ta.lane_byte, bsh = uni(ta.shift);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ta.base, 0, -1, 0x00020000);
    const prog1_ptr macros = (prog1_ptr)(uintptr_t)cd.macros;
    auto ldv = [&](uint32_t slot) -> uint64_t {
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_byte, slot << bsh, 0);
        return (uint64_t)v.x | ((uint64_t)v.y << 32);
    };
    const uint32_t m0 = blockIdx.y * cd.per_block, m1 = min(m0 + cd.per_block, cd.n_macros);
    bool wrong = false;
    for (uint32_t m = m0; m < m1; ++m) {
        const u32x16_a4 W = *(prog16_ptr)(macros + 14 * m);
        uint64_t cur[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) cur[i] = ldv(W[i]);
        uint32_t o = W[12];
        auto matrix = [&](bool external) {
            uint64_t mm[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) mm[i] = ldv(o + i);
            o += 12;
            if (external) p2::mds_external(cur); else p2::mds_inner(cur);
#pragma unroll
            for (int i = 0; i < 12; ++i) { wrong |= cur[i] != mm[i]; cur[i] = mm[i]; }
        };
        auto sbox_check = [&](uint64_t& x, uint64_t rc, uint64_t t, uint64_t x2, uint64_t x3, uint64_t x4, uint64_t x7) {
            wrong |= t != gl::add(x, rc);
            wrong |= x2 != gl::sqr(t);
            wrong |= x3 != gl::mul(x2, t);
            wrong |= x4 != gl::sqr(x2);
            wrong |= x7 != gl::mul(x3, x4);
            x = x7;
        };
        matrix(true);
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
#pragma unroll 1
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = half * 26 + r4;
#pragma unroll
                for (int q = 0; q < 3; ++q) {   // four S-boxes at a time: 20 loads in flight
                    uint64_t v[20];
#pragma unroll
                    for (int i = 0; i < 20; ++i) v[i] = ldv(o + i);
                    o += 20;
#pragma unroll
                    for (int i = 0; i < 4; ++i) sbox_check(cur[4 * q + i], p2::RC[12 * r + 4 * q + i], v[5 * i], v[5 * i + 1], v[5 * i + 2], v[5 * i + 3], v[5 * i + 4]);
                }
                matrix(true);
            }
            if (half == 0) {
#pragma unroll 1
                for (int r = 4; r < 26; ++r) {
                    uint64_t v[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) v[i] = ldv(o + i);
                    o += 5;
                    sbox_check(cur[0], p2::RC[12 * r], v[0], v[1], v[2], v[3], v[4]);
                    matrix(false);
                }
            }
        }
    }
    if (wrong && active) report(cd.fail, lane, 0xffffeu, 0, 0);
}

// public inputs of every instance of the batch, packed [instance][n_public] (the payload of zk_cs_gather_commitments)
__global__ void k_pack_public(const uint64_t* __restrict__ outer_store, uint64_t n_store, const uint32_t* __restrict__ slots, uint32_t n_public,
                              uint32_t n_instances, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_instances * n_public) return;
    const uint32_t inst = i / n_public, k = i % n_public;
    out[i] = outer_store[cell_off(n_store, slots[k], inst)];
}

// hook_compare_witness (/root/reference/src/fsm_input_output/mod.rs:102-133) on the device: the circuit's values of a list of outer
// variables (the closed-form input: hidden_fsm_output, observable_output ...) against the host's expectation [n_vars][n_instances];
// the first difference (lowest instance, then lowest position) is reported through the packed failure key of the checkers
__global__ void k_hook_compare(const uint64_t* __restrict__ outer_store, uint64_t n_store, const uint32_t* __restrict__ slots, uint32_t n_vars,
                               uint32_t n_instances, const uint64_t* __restrict__ expected, unsigned long long* fail) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_instances * n_vars) return;
    const uint32_t inst = i / n_vars, k = i % n_vars;
    if (outer_store[cell_off(n_store, slots[k], inst)] != expected[(size_t)k * n_instances + inst])
        atomicMin(fail, ((unsigned long long)inst << 32) | k);
}

// every input stream word must be a canonical field element; failure key: lane, slot 0xfffff, j = word index (mod 256)
__global__ void k_check_inputs(const uint64_t* __restrict__ inputs, uint32_t n_words, uint32_t n_lanes, uint64_t stride, unsigned long long* fail) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= n_lanes) return;
    for (uint32_t w = blockIdx.y; w < n_words; w += gridDim.y)
        if (inputs[(size_t)w * stride + lane] >= gl::P) report(fail, lane, 0xfffffu, w & 0xff, 0);
}

// The 950 intermediates of every in-circuit Poseidon2 permutation (all outputs of ZK_OP_P2_ROUNDS but the final 12), recomputed from
// the 12 stored inputs and written to their slots — the other half of the deferred mode (ScopeDev::defer_p2): same values, same order
// as run_tile2's emit path.  Descriptors = the check macros (12 input slots, first output slot, row).
__global__ __launch_bounds__(TPB) void k_fill_p2(CheckP2Dev cd) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    uint32_t lane = blockIdx.x * TPB + threadIdx.x;
    if ((blockIdx.x * TPB + (threadIdx.x & ~63u)) >= cd.n_lanes) return;
    lane = lane < cd.n_lanes ? lane : cd.n_lanes - 1;
    const TileAddr ta = tile_addr(const_cast<uint64_t*>(cd.cells), cd.n_cells, lane);
    const uint32_t lane_byte = ta.lane_byte, bsh = uni(ta.shift);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ta.base, 0, -1, 0x00020000);
    const prog1_ptr macros = (prog1_ptr)(uintptr_t)cd.macros;
    const uint32_t m0 = blockIdx.y * cd.per_block, m1 = min(m0 + cd.per_block, cd.n_macros);
    for (uint32_t m = m0; m < m1; ++m) {
        const u32x16_a4 W = *(prog16_ptr)(macros + 14 * m);
        uint64_t s[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_byte, W[i] << bsh, 0);
            s[i] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        }
        uint32_t o = W[12] << bsh;
        const uint32_t step = 1u << bsh;
        auto st = [&](uint64_t v) {
            u32x2 w;
            w.x = (uint32_t)v; w.y = (uint32_t)(v >> 32);
            __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, lane_byte, o, 0);
            o += step;
        };
        p2::mds_external(s);
#pragma unroll
        for (int i = 0; i < 12; ++i) st(s[i]);
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
#pragma unroll 1
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = half * 26 + r4;
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const uint64_t t = gl::add(s[i], p2::RC[12 * r + i]);
                    const uint64_t x2 = gl::sqr(t), x3 = gl::mul(x2, t), x4 = gl::sqr(x2), x7 = gl::mul(x3, x4);
                    st(t); st(x2); st(x3); st(x4); st(x7);
                    s[i] = x7;
                }
                p2::mds_external(s);
                if (!(half == 1 && r4 == 3)) {   // the last layer's outputs are the 12 the witness kernel wrote
#pragma unroll
                    for (int i = 0; i < 12; ++i) st(s[i]);
                }
            }
            if (half == 0) {
#pragma unroll 1
                for (int r = 4; r < 26; ++r) {
                    const uint64_t t = gl::add(s[0], p2::RC[12 * r]);
                    const uint64_t x2 = gl::sqr(t), x3 = gl::mul(x2, t), x4 = gl::sqr(x2), x7 = gl::mul(x3, x4);
                    st(t); st(x2); st(x3); st(x4); st(x7);
                    s[0] = x7;
                    p2::mds_inner(s);
#pragma unroll
                    for (int i = 0; i < 12; ++i) st(s[i]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Lookup multiplicities WITHOUT global atomics (k_multiplicities).  The witness kernels used to add 1 to
// mult[instance][table row] per lookup: ~1e9 L2 atomic operations per step for the hash circuits (64 lanes, 64 different rows: the
// L2 atomic rate, 31 G/s, was 60 % of keccak's and 48 % of sha256's loop kernel) and serialised same-address storms for the VM.
// Now a pass of its own after the witness kernels: one workgroup owns (instance, table, chunk of <= 32 768 rows), walks every
// lookup site of that table over the instance's lanes — key values read back from the variable store, coalesced over lanes —
// counts in LDS (ds atomics) and adds its counts to the instance's multiplicity vector with plain stores (exclusive ownership).
// ------------------------------------------------------------------------------------------------------------------------
struct MultDev {
    const uint64_t* store; uint64_t n_store; uint32_t lanes_per_instance, n_lanes;
    const uint32_t* sites; uint32_t n_sites;   // 3 key slots per site (0xffffffff: absent)
    zk_table_desc t; const uint64_t* table_words;
    uint32_t* mult; uint32_t total_table_rows; uint32_t chunk_rows;
    uint32_t site_splits;   // >= 1: gridDim.z = lane ranges x site_splits — few instances x few lanes (eip_4844: 8 blobs) still fill the chip
};
constexpr uint32_t MULT_CHUNK_ROWS = 32768;
// PACKED: two 16-bit counters per LDS word, so that ONE workgroup covers a 65 536-row table (Xor8 / And8 / AndN8) and the keys of its
// sites are read once instead of once per 32 768-row chunk (round 3: keccak's pass 7.5 ms, bandwidth-bound on re-read keys).  Bit 15 of
// a counter is a guard: the increment that sets it books 32 768 on the instance's vector and clears it, so a row hit any number of
// times (the all-zero rows of padded blocks) never carries into its neighbour.
template <bool PACKED>
__global__ __launch_bounds__(1024) void k_multiplicities(MultDev a) {
    __shared__ uint32_t cnt[MULT_CHUNK_ROWS];
    const uint32_t base = blockIdx.x * a.chunk_rows, inst = blockIdx.y;
    const uint32_t rows_here = min(a.chunk_rows, a.t.n_rows - base);
    for (uint32_t i = threadIdx.x; i < (PACKED ? (rows_here + 1) / 2 : rows_here); i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    uint32_t* out = a.mult + (size_t)inst * a.total_table_rows + a.t.mult_off + base;
    // few instances: gridDim.z workgroups share an instance, each takes a contiguous range of its lanes (multiples of 64) and one of
    // site_splits interleaved subsets of the sites
    const uint32_t lane_splits = gridDim.z / a.site_splits, lane_split = blockIdx.z / a.site_splits, site_split = blockIdx.z % a.site_splits;
    const uint32_t per = ((a.lanes_per_instance + lane_splits - 1) / lane_splits + 63) & ~63u;
    const uint32_t lo = lane_split * per, hi = min(lo + per, a.lanes_per_instance);
    const uint32_t wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6, l0 = threadIdx.x & 63;
    for (uint32_t site = site_split * n_waves + wave; site < a.n_sites; site += n_waves * a.site_splits) {
        const uint32_t s0 = uni(a.sites[3 * site]), s1 = uni(a.sites[3 * site + 1]), s2 = uni(a.sites[3 * site + 2]);
        // four lane groups per trip: the key loads of all four go out before the first table search (the pass is bandwidth-bound on
        // re-read keys; one group at a time left each wavefront with two or three loads in flight)
        for (uint32_t l = lo + l0; l < hi; l += 256) {
            uint64_t k0[4], k1[4], k2[4];
            bool live[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t ll = l + 64 * j;
                const uint32_t lane = inst * a.lanes_per_instance + ll;
                live[j] = ll < hi && lane < a.n_lanes;
                const uint32_t la = live[j] ? lane : inst * a.lanes_per_instance + lo;   // a lane of this range: loaded, not counted
                k0[j] = a.store[cell_off(a.n_store, s0, la)];
                k1[j] = s1 != 0xffffffffu ? a.store[cell_off(a.n_store, s1, la)] : 0;
                k2[j] = s2 != 0xffffffffu ? a.store[cell_off(a.n_store, s2, la)] : 0;
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t row = table_find3(a.t, a.table_words, k0[j], k1[j], k2[j]);
                if (live[j] && row < a.t.n_rows && row - base < rows_here) {
                    const uint32_t r = row - base;
                    if constexpr (PACKED) {
                        const uint32_t sh = (r & 1u) * 16u;
                        const uint32_t old = atomicAdd(&cnt[r >> 1], 1u << sh);
                        if (((old >> sh) & 0xffffu) == 0x7fffu) {        // this increment set the field's guard bit: book 32 768 and clear it
                            atomicSub(&cnt[r >> 1], 0x8000u << sh);      // (other increments may land in between: the field stays far below 2^16,
                            atomicAdd(out + r, 32768u);                  //  so nothing ever carries into the neighbouring counter)
                        }
                    } else {
                        atomicAdd(&cnt[r], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < rows_here; i += blockDim.x) {
        const uint32_t v = PACKED ? (cnt[i >> 1] >> ((i & 1u) * 16u)) & 0xffffu : cnt[i];
        if (!v) continue;
        // the vector is zeroed at the start of resolve and the launches of the two scopes are ordered on the stream; workgroups sharing
        // an instance (gridDim.z > 1) meet on a counter, and so do the overflow bookings of the packed form
        if (gridDim.z > 1 || PACKED) atomicAdd(out + i, v); else out[i] += v;
    }
}

}  // namespace zke
