// cs.cpp — recorder / placer / program emitter / GPU executor.  See cs.hpp.
#include "cs.hpp"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <array>
#include <cstdio>
#include <functional>
#include <map>
#include <queue>
#include "device_api.hpp"
#include "poseidon_consts.hpp"
#include "../../include/zkgl_vm.h"
#include "keccak_macro.hpp"
#include "sha256_macro.hpp"
#include "sha256_macro4.hpp"
#include "bytebuf_macro.hpp"

static constexpr int ZK_MACRO_FAILURE = 0x7fff0001;  // internal: a macro check packet failed, re-run the gate-by-gate program

namespace zkgl {

int device_cu_count();   // capi.cpp: the CU count zk_init read from the device

namespace {

struct GateInfo { uint32_t width, n_consts, n_relations; };
// widths match zke::GATE_WIDTH (kernels_engine.hpp)
const GateInfo GATES[ZK_GATE__COUNT] = {
    {0, 0, 0},   // NOP
    {1, 1, 1},   // CONST (one constant per instance, see place_scope)
    {1, 0, 1},   // BOOLEAN
    {4, 2, 1},   // FMA
    {5, 4, 1},   // REDUCTION4
    {4, 0, 1},   // SELECT
    {3, 0, 2},   // ZEROCHECK
    {5, 1, 1},   // UINTX_ADD
    {9, 0, 1},   // DOT4
    {24, 0, 12}, // MATMUL12_EXT
    {24, 0, 12}, // MATMUL12_INT
    {1, 0, 0},   // PUBLIC_INPUT
    {6, 0, 1},   // U32_FMA
    {5, 1, 1},   // REDUCTION_BY_POWERS4
    {26, 0, 2},  // U8X4_FMA
};

// element offset of (cell, lane) in the wave-tiled cell storage (see kernels_engine.hpp)
size_t tiled_offset(uint64_t geom, uint64_t cell, uint64_t lane) { return zkgeom::offset(geom, cell, lane); }  // geom: store_geom.hpp (a bare count = 64-lane tiles)

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw ZkError(ZK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
void dev_check(int rc) {
    if (rc != 0) throw ZkError(ZK_ERR_HIP, zkdev::last_hip_error());
}

template <class T>
T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    hip_check(hipMalloc((void**)&d, bytes), "hipMalloc");
    if (!v.empty()) hip_check(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
    return d;
}

}  // namespace

CS::CS(const zk_geometry& g, uint64_t max_trace_len, uint64_t max_variables)
    : geo_(g), max_trace_len_(max_trace_len), max_variables_(max_variables) {
    if (g.num_columns_under_copy_permutation < 24)
        throw ZkError(ZK_ERR_INVALID, "geometry: need >= 24 copy columns (MatrixMultiplicationGate<12>)");
    if (g.num_constant_columns < 4) throw ZkError(ZK_ERR_INVALID, "geometry: need >= 4 constant columns");
    loop_.is_loop = true;
    // NOP and PUBLIC_INPUT are always available
    allowed_gates_ = (1ull << ZK_GATE_NOP) | (1ull << ZK_GATE_PUBLIC_INPUT);
}

CS::~CS() {
    free_scope_device(outer_);
    free_scope_device(loop_);
    if (d_tables_) hipFree(d_tables_);
    if (d_table_words_) hipFree(d_table_words_);
    if (d_mult_) hipFree(d_mult_);
    if (d_links_) hipFree(d_links_);
    if (d_links_store_) hipFree(d_links_store_);
    if (d_links_store_n_) hipFree(d_links_store_n_);
    if (d_loop_last_slots_) hipFree(d_loop_last_slots_);
    for (auto p : d_streams_store_) if (p) hipFree(p);
    for (auto p : d_streams_store_n_) if (p) hipFree(p);
    if (d_public_slots_) hipFree(d_public_slots_);
    if (d_seed_prog_) hipFree(d_seed_prog_);
    if (d_seed_wprog_) hipFree(d_seed_wprog_);
    if (d_seed_wcarries_) hipFree(d_seed_wcarries_);
    if (d_seed_sprog_) hipFree(d_seed_sprog_);
    if (d_seed_scarries_) hipFree(d_seed_scarries_);
    if (d_seed_carries_) hipFree(d_seed_carries_);
    if (d_native_blob_) hipFree(d_native_blob_);
    if (d_state0_slot_) hipFree(d_state0_slot_);
    if (d_native_outer_slots_) hipFree(d_native_outer_slots_);
    if (d_native_scratch_) hipFree(d_native_scratch_);
    if (d_seed_outer_) hipFree(d_seed_outer_);
    for (auto p : d_streams_) if (p) hipFree(p);
    if (d_carries_) hipFree(d_carries_);
    for (int i = 0; i < 2; ++i) {
        if (d_sig_rel_[i]) hipFree(d_sig_rel_[i]);
        if (d_ep_index_[i]) hipFree(d_ep_index_[i]);
        if (d_ovr_[i]) hipFree(d_ovr_[i]);
    }
    if (d_fail_) hipFree(d_fail_);
    if (aux_stream_) hipStreamDestroy((hipStream_t)aux_stream_);
    for (auto& e : ev2_)
        if (e) hipEventDestroy((hipEvent_t)e);
    for (auto& e : ev_)
        if (e) hipEventDestroy((hipEvent_t)e);
}

void CS::free_scope_device(Scope& s) {
    if (s.d_prog) hipFree(s.d_prog);
    if (s.d_prog2) hipFree(s.d_prog2);
    if (s.d_cprog) hipFree(s.d_cprog);
    if (s.d_cchunks) hipFree(s.d_cchunks);
    if (s.d_cprog_full) hipFree(s.d_cprog_full);
    if (s.d_cmacros) hipFree(s.d_cmacros);
    if (s.d_cchunks_full) hipFree(s.d_cchunks_full);
    if (s.d_cprog_fused) hipFree(s.d_cprog_fused);
    if (s.d_cchunks_fused) hipFree(s.d_cchunks_fused);
    s.d_cprog_fused = nullptr; s.d_cchunks_fused = nullptr;
    if (s.d_mult_sites) hipFree(s.d_mult_sites);
    if (s.d_sprog) hipFree(s.d_sprog);
    if (s.d_sprog_n) hipFree(s.d_sprog_n);
    s.d_sprog = nullptr; s.d_sprog_n = nullptr;
    if (s.d_consts) hipFree(s.d_consts);
    if (s.d_rows) hipFree(s.d_rows);
    if (s.d_rowconsts) hipFree(s.d_rowconsts);
    if (s.d_lrows) hipFree(s.d_lrows);
    if (s.d_copies) hipFree(s.d_copies);
    if (s.d_alias) hipFree(s.d_alias);
    s.d_alias = nullptr;
    if (s.d_slot1) hipFree(s.d_slot1);
    s.d_slot1 = nullptr;
    if (s.d_mat_pairs) hipFree(s.d_mat_pairs);
    s.d_mat_pairs = nullptr;
    if (s.d_store) hipFree(s.d_store);
    s.d_store = nullptr;
    if (s.d_store_n) hipFree(s.d_store_n);
    if (s.d_prog2n) hipFree(s.d_prog2n);
    if (s.d_cprog_fused_n) hipFree(s.d_cprog_fused_n);
    if (s.d_slot_aw) hipFree(s.d_slot_aw);
    if (s.d_aw1) hipFree(s.d_aw1);
    s.d_aw1 = nullptr;
    s.d_store_n = nullptr; s.d_prog2n = nullptr; s.d_cprog_fused_n = nullptr; s.d_slot_aw = nullptr;
    if (s.d_cells) hipFree(s.d_cells);
    s.d_prog = nullptr; s.d_prog2 = nullptr; s.d_cprog = nullptr; s.d_cchunks = nullptr; s.d_cprog_full = nullptr; s.d_cchunks_full = nullptr; s.d_cmacros = nullptr; s.d_mult_sites = nullptr; s.d_consts = nullptr; s.d_rows = nullptr; s.d_rowconsts = nullptr; s.d_lrows = nullptr;
    s.d_copies = nullptr; s.d_cells = nullptr;
}

// ------------------------------------------------------------------ configuration
void CS::allow_lookup(uint32_t width, uint32_t reps, bool share) {
    if (finalized_) throw ZkError(ZK_ERR_INVALID, "allow_lookup after finalize");
    if (width < 2 || width > 4 || reps == 0) throw ZkError(ZK_ERR_INVALID, "lookup width must be 2..4, reps > 0");
    lookup_width_ = width; lookup_reps_ = reps; lookup_share_id_ = share;
}
void CS::allow_gate(uint32_t kind) {
    if (kind >= ZK_GATE__COUNT) throw ZkError(ZK_ERR_INVALID, "unknown gate kind");
    allowed_gates_ |= 1ull << kind;
}
bool CS::gate_is_allowed(uint32_t kind) const { return kind < ZK_GATE__COUNT && ((allowed_gates_ >> kind) & 1); }

uint32_t CS::add_table(uint32_t marker, uint32_t n_keys, uint32_t n_vals, const uint64_t* rows, uint32_t n_rows) {
    if (finalized_) throw ZkError(ZK_ERR_INVALID, "add_table after finalize");
    if (lookup_width_ == 0) throw ZkError(ZK_ERR_INVALID, "add_table: lookups not allowed in this CS");
    if (n_keys == 0 || n_keys > 3 || n_keys + n_vals > lookup_width_ || n_rows == 0)
        throw ZkError(ZK_ERR_INVALID, "add_table: bad shape for the configured lookup width");
    for (auto& t : tables_)
        if (t.marker == marker) throw ZkError(ZK_ERR_INVALID, "add_table: marker already registered");
    TableRec t;
    t.marker = marker; t.n_keys = n_keys; t.n_vals = n_vals; t.n_rows = n_rows;
    const uint32_t w = n_keys + n_vals;
    // sort rows by key tuple (the device binary-searches; the dense test below needs sorted rows too)
    std::vector<uint32_t> order(n_rows);
    for (uint32_t i = 0; i < n_rows; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        for (uint32_t k = 0; k < n_keys; ++k) {
            if (rows[(size_t)x * w + k] != rows[(size_t)y * w + k]) return rows[(size_t)x * w + k] < rows[(size_t)y * w + k];
        }
        return x < y;
    });
    t.rows.resize((size_t)n_rows * w);
    for (uint32_t i = 0; i < n_rows; ++i) {
        std::memcpy(&t.rows[(size_t)i * w], &rows[(size_t)order[i] * w], w * sizeof(uint64_t));
        for (uint32_t k = 0; k < w; ++k)
            if (t.rows[(size_t)i * w + k] >= 0xFFFFFFFF00000001ull) throw ZkError(ZK_ERR_INVALID, "add_table: non-canonical element");
        if (i > 0 && std::memcmp(&t.rows[(size_t)i * w], &t.rows[(size_t)(i - 1) * w], n_keys * sizeof(uint64_t)) == 0)
            throw ZkError(ZK_ERR_INVALID, "add_table: duplicate key tuple");
    }
    t.byte_valued = n_vals > 0;
    for (uint32_t i = 0; i < n_rows && t.byte_valued; ++i)
        for (uint32_t k = n_keys; k < w; ++k) t.byte_valued = t.byte_valued && t.rows[(size_t)i * w + k] < 256;
    // dense test: keys are a full product of power-of-two ranges, last key fastest
    t.dense = false;
    t.key_shift[0] = t.key_shift[1] = t.key_shift[2] = 0;
    {
        uint64_t maxk[3] = {0, 0, 0};
        for (uint32_t i = 0; i < n_rows; ++i)
            for (uint32_t k = 0; k < n_keys; ++k) maxk[k] = std::max(maxk[k], t.rows[(size_t)i * w + k]);
        uint32_t bits[3] = {0, 0, 0};
        bool ok = true;
        uint32_t total = 0;
        for (uint32_t k = 0; k < n_keys; ++k) {
            uint64_t m = maxk[k] + 1;
            if (m & (m - 1)) { ok = false; break; }
            while ((1ull << bits[k]) < m) ++bits[k];
            total += bits[k];
        }
        if (ok && total < 31 && (1ull << total) == n_rows) {
            uint32_t sh = 0;
            for (int k = (int)n_keys - 1; k >= 0; --k) { t.key_shift[k] = sh; sh += bits[k]; }
            // sorted + unique + full product => row index == packed key
            t.dense = true;
        }
    }
    tables_.push_back(std::move(t));
    return (uint32_t)tables_.size();
}

uint32_t CS::table_id(uint32_t marker) const {
    for (size_t i = 0; i < tables_.size(); ++i)
        if (tables_[i].marker == marker) return (uint32_t)i + 1;
    throw ZkError(ZK_ERR_INVALID, "table must be added before");  // reference: src/main_vm/utils.rs:95-97
}

bool CS::has_table(uint32_t marker) const {
    for (auto& t : tables_)
        if (t.marker == marker) return true;
    return false;
}

// ------------------------------------------------------------------ recording
void CS::check_var(zk_var v, bool want_loop) const {
    if (v == ZK_VAR_NONE) throw ZkError(ZK_ERR_INVALID, "placeholder variable used");
    const Scope& s = is_loop_var(v) ? loop_ : outer_;
    if (var_index(v) >= s.n_vars) throw ZkError(ZK_ERR_INVALID, "variable index out of range");
    if (is_loop_var(v) != want_loop)
        throw ZkError(ZK_ERR_INVALID, want_loop ? "outer variable used inside the loop without loop_import"
                                                : "loop variable used outside the loop without loop_last");
}

zk_var CS::alloc_vars(uint32_t n) {
    if (finalized_) throw ZkError(ZK_ERR_INVALID, "alloc after finalize");
    Scope& s = cur();
    if ((uint64_t)outer_.n_vars + loop_.n_vars + n > max_variables_) throw ZkError(ZK_ERR_CAPACITY, "max_variables exceeded");
    uint32_t first = s.n_vars;
    s.n_vars += n;
    return first | (in_loop_ ? LOOP_BIT : 0);
}
zk_var CS::alloc_var() { return alloc_vars(1); }

uint32_t CS::pool_const(Scope& s, uint64_t v) {
    auto it = s.const_pool_idx.find(v);
    if (it != s.const_pool_idx.end()) return it->second;
    uint32_t idx = (uint32_t)s.const_pool.size();
    s.const_pool.push_back(v);
    s.const_pool_idx.emplace(v, idx);
    return idx;
}

zk_var CS::alloc_constant(uint64_t value) {
    if (value >= 0xFFFFFFFF00000001ull) throw ZkError(ZK_ERR_INVALID, "allocate_constant: non-canonical value");
    Scope& s = cur();
    auto it = s.const_vars.find(value);
    if (it != s.const_vars.end()) return it->second | (in_loop_ ? LOOP_BIT : 0);
    if (!gate_is_allowed(ZK_GATE_CONST)) throw ZkError(ZK_ERR_GATE_NOT_ALLOWED, "ConstantsAllocatorGate not allowed");
    zk_var v = alloc_var();
    OpRec op{ZK_OP_CONST, 0, 0, {{Operand::CONSTPOOL, pool_const(s, value)}}, {var_index(v)}};
    s.ops.push_back(std::move(op));
    GateRec g{ZK_GATE_CONST, {var_index(v)}, {value}};
    s.gates.push_back(std::move(g));
    s.const_vars.emplace(value, var_index(v));
    return v;
}

zk_var CS::input(uint32_t word) {
    Scope& s = cur();
    zk_var v = alloc_var();
    OpRec op{ZK_OP_INPUT, 0, 0, {{Operand::RAW, word}}, {var_index(v)}};
    s.ops.push_back(std::move(op));
    s.n_input_words = std::max(s.n_input_words, word + 1);
    s.input_word[var_index(v)] = word;
    return v;
}

void CS::place_gate(uint32_t kind, const zk_var* vars, uint32_t n_vars, const uint64_t* consts, uint32_t n_consts) {
    if (finalized_) throw ZkError(ZK_ERR_INVALID, "place_gate after finalize");
    if (kind >= ZK_GATE__COUNT || kind == ZK_GATE_NOP) throw ZkError(ZK_ERR_INVALID, "place_gate: bad kind");
    if (!gate_is_allowed(kind)) throw ZkError(ZK_ERR_GATE_NOT_ALLOWED, "gate kind not configured for this CS");
    const GateInfo& gi = GATES[kind];
    if (n_vars != gi.width || n_consts != gi.n_consts) throw ZkError(ZK_ERR_INVALID, "place_gate: wrong arity");
    if (gi.n_consts > geo_.num_constant_columns) throw ZkError(ZK_ERR_INVALID, "gate needs more constant columns");
    if (gi.width > geo_.num_columns_under_copy_permutation) throw ZkError(ZK_ERR_INVALID, "gate is wider than the copy-permutation columns of this geometry");
    Scope& s = cur();
    GateRec g;
    g.kind = kind;
    for (uint32_t i = 0; i < n_vars; ++i) {
        check_var(vars[i], in_loop_);
        g.vars.push_back(var_index(vars[i]));
    }
    for (uint32_t i = 0; i < n_consts; ++i) {
        if (consts[i] >= 0xFFFFFFFF00000001ull) throw ZkError(ZK_ERR_INVALID, "place_gate: non-canonical constant");
        g.consts.push_back(consts[i]);
    }
    if (kind == ZK_GATE_PUBLIC_INPUT) {
        if (in_loop_) throw ZkError(ZK_ERR_INVALID, "public inputs must be outer-scope variables");
        public_vars_.push_back(g.vars[0]);
    }
    if (macro_window_op_ >= 0 && macro_window_loop_ == in_loop_) g.owner = (int32_t)cur().ops[macro_window_op_].outs.front();
    s.gates.push_back(std::move(g));
}

void CS::emit_op(uint32_t opcode, uint32_t a, uint32_t b, const zk_var* ins, uint32_t n_in, const zk_var* outs,
                 uint32_t n_out, const uint64_t* imm, uint32_t n_imm) {
    if (finalized_) throw ZkError(ZK_ERR_INVALID, "emit_op after finalize");
    Scope& s = cur();
    OpRec op;
    op.opcode = (uint8_t)opcode; op.a = (uint8_t)a; op.b = (uint16_t)b;
    auto need = [&](uint32_t nin, uint32_t nout, uint32_t nimm) {
        if (n_in != nin || n_out != nout || n_imm != nimm) throw ZkError(ZK_ERR_INVALID, "emit_op: wrong arity for opcode");
    };
    switch (opcode) {
    case ZK_OP_FMA: need(3, 1, 2); break;
    case ZK_OP_LC4: need(4, 1, 4); break;
    case ZK_OP_SELECT: need(3, 1, 0); break;
    case ZK_OP_ISZERO: need(1, 2, 0); break;
    case ZK_OP_UADD: case ZK_OP_USUB:
        need(3, 2, 0);
        if (a == 0 || a > 32) throw ZkError(ZK_ERR_INVALID, "UADD/USUB: bits must be 1..32");
        break;
    case ZK_OP_DOT4: need(8, 1, 0); break;
    case ZK_OP_MATMUL12: need(12, 12, 0); if (a > 1) throw ZkError(ZK_ERR_INVALID, "MATMUL12: matrix id 0/1"); break;
    case ZK_OP_SPLIT:
        if (n_in != 1 || n_imm != 0 || n_out != a || a == 0 || b == 0 || b > 32) throw ZkError(ZK_ERR_INVALID, "SPLIT: bad shape");
        break;
    case ZK_OP_POSEIDON2: if (a == 1) need(13, 12, 0); else { need(12, 12, 0); if (a) throw ZkError(ZK_ERR_INVALID, "POSEIDON2: a must be 0 / 1"); } break;
    case ZK_OP_P2_ROUNDS: need(12, 962, 0); break;
    case ZK_OP_U32MULADD: need(4, 2, 0); break;
    case ZK_OP_U8X4FMA: need(16, 10, 0); break;
    case ZK_OP_KECCAK_F: {
        if (!allow_macro_ops_) throw ZkError(ZK_ERR_INVALID, "emit_op: macro-ops are recorded by the engine's gadgets only");
        zkk::CountBackend cb; int dummy[25] = {0}; uint64_t rc[24];
        for (int i = 0; i < 24; ++i) rc[i] = zkk::RC[i];
        zkk::keccak_f(cb, dummy, rc);
        need(200, cb.n, 0);
        s.uses_bigint = true;   // the heavy kernel variants (register budget of the macro-op) carry its handler
        uses_lookup_macros_ = true;
    } break;
    case ZK_OP_SHA256_ROUNDS: {
        if (!allow_macro_ops_) throw ZkError(ZK_ERR_INVALID, "emit_op: macro-ops are recorded by the engine's gadgets only");
        if (a > 1) throw ZkError(ZK_ERR_INVALID, "SHA256_ROUNDS: a must be 0 (8-bit tables) / 1 (the reference's 4-bit-chunk tables)");
        uint32_t n_outs;
        if (a == 1) {   // the reference's table set: csrc/sha256_macro4.hpp
            zks4::CountBackend cb; int cst[8] = {0}, cblk[16] = {0}, cw[64]; zks4::CountBackend::Splits csp[64];
            zks4::compress(cb, cst, cblk, cw, csp, zks::K);
            n_outs = cb.n;
            uses_sha4_macro_ = true;
        } else {
            zks::CountBackend cb; int cst[8] = {0}, cblk[16] = {0}, cw[64];
            zks::compress(cb, cst, cblk, cw, zks::K);
            n_outs = cb.n;
        }
        need(96, n_outs, 0);
        s.uses_bigint = true;
        uses_lookup_macros_ = true;
    } break;
    case ZK_OP_BYTEBUF_FILL:
        if (!allow_macro_ops_) throw ZkError(ZK_ERR_INVALID, "emit_op: macro-ops are recorded by the engine's gadgets only");
        uses_bytebuf_macro_ = true;
        need(zkb::N_INPUTS, zkb::n_outputs(), 0);
        s.uses_bigint = true;
        uses_lookup_macros_ = true;
        break;
    case ZK_OP_NN_MULMOD:
        if (a == 0 || a > 17 || b == 0 || b > 17 || a + b < 16 || n_in != a + b || n_imm != 16 || n_out != a + b - 15 + 16)
            throw ZkError(ZK_ERR_INVALID, "NN_MULMOD: bad shape");
        if (imm[15] == 0 || imm[15] > 0xffff) throw ZkError(ZK_ERR_INVALID, "NN_MULMOD: top modulus limb must be a non-zero u16");
        s.uses_bigint = true;
        break;
    case ZK_OP_DIVREM: need(1, 2, 0); if (b == 0 || b > 65535) throw ZkError(ZK_ERR_INVALID, "DIVREM: divisor must be 1..65535"); break;
    case ZK_OP_U256_MULWIDE: case ZK_OP_U256_DIVREM: need(16, 16, 0); break;
    default: throw ZkError(ZK_ERR_INVALID, "emit_op: opcode not recordable through this entry");
    }
    for (uint32_t i = 0; i < n_imm; ++i) {
        if (imm[i] >= 0xFFFFFFFF00000001ull) throw ZkError(ZK_ERR_INVALID, "emit_op: non-canonical immediate");
        op.ins.push_back({Operand::CONSTPOOL, pool_const(s, imm[i])});
    }
    for (uint32_t i = 0; i < n_in; ++i) {
        check_var(ins[i], in_loop_);
        op.ins.push_back({Operand::VAR, var_index(ins[i])});
    }
    for (uint32_t i = 0; i < n_out; ++i) {
        check_var(outs[i], in_loop_);
        op.outs.push_back(var_index(outs[i]));
    }
    s.ops.push_back(std::move(op));
}

void CS::lookup(uint32_t tid, const zk_var* keys, uint32_t n_keys, zk_var* vals, uint32_t n_vals) {
    if (tid == 0 || tid > tables_.size()) throw ZkError(ZK_ERR_INVALID, "perform_lookup: unknown table id");
    const TableRec& t = tables_[tid - 1];
    if (n_keys != t.n_keys || n_vals != t.n_vals) throw ZkError(ZK_ERR_INVALID, "perform_lookup: K/V mismatch with table");
    Scope& s = cur();
    LookupRec lr;
    lr.table = tid;
    OpRec op;
    op.opcode = ZK_OP_LOOKUP; op.a = (uint8_t)n_keys; op.b = (uint16_t)n_vals;
    op.ins.push_back({Operand::RAW, tid});
    for (uint32_t i = 0; i < n_keys; ++i) {
        check_var(keys[i], in_loop_);
        op.ins.push_back({Operand::VAR, var_index(keys[i])});
        lr.vars.push_back(var_index(keys[i]));
    }
    zk_var first = alloc_vars(n_vals);
    for (uint32_t i = 0; i < n_vals; ++i) {
        vals[i] = first + i;
        op.outs.push_back(var_index(first) + i);
        lr.vars.push_back(var_index(first) + i);
    }
    lr.owner = OWNER_LOOKUP_OP;   // the ZK_OP_LOOKUP below finds the row (or reports the miss) and fills the fresh value variables
    s.ops.push_back(std::move(op));
    s.lookups.push_back(std::move(lr));
}

void CS::lookup_given(uint32_t tid, const zk_var* keys, uint32_t n_keys, const zk_var* vals, uint32_t n_vals) {
    if (tid == 0 || tid > tables_.size()) throw ZkError(ZK_ERR_INVALID, "lookup_given: unknown table id");
    const TableRec& t = tables_[tid - 1];
    if (n_keys != t.n_keys || n_vals != t.n_vals) throw ZkError(ZK_ERR_INVALID, "lookup_given: K/V mismatch with table");
    LookupRec lr;
    lr.table = tid;
    for (uint32_t i = 0; i < n_keys; ++i) { check_var(keys[i], in_loop_); lr.vars.push_back(var_index(keys[i])); }
    for (uint32_t i = 0; i < n_vals; ++i) { check_var(vals[i], in_loop_); lr.vars.push_back(var_index(vals[i])); }
    // a tuple over existing variables is evaluated by the witness kernels only when a macro-op's gadget gives it inside its window AND the
    // op produces every value of it (the op computes the row from the keys and tests that they are table keys); any other tuple is
    // checked from the store in every mode
    if (macro_window_op_ >= 0 && macro_window_loop_ == in_loop_) {
        const OpRec& mop = cur().ops[macro_window_op_];
        bool mine = n_vals > 0 && !mop.outs.empty();
        for (uint32_t i = 0; i < n_vals && mine; ++i) { const uint32_t v = var_index(vals[i]); mine = v >= mop.outs.front() && v <= mop.outs.back(); }
        if (mine) lr.owner = (int32_t)mop.outs.front();
    }
    cur().lookups.push_back(std::move(lr));
}

void CS::emit_macro_op(uint32_t opcode, const zk_var* ins, uint32_t n_in, zk_var first_out, uint32_t n_out, uint32_t a) {
    std::vector<zk_var> outs(n_out);
    for (uint32_t i = 0; i < n_out; ++i) outs[i] = first_out + i;
    allow_macro_ops_ = true;
    if (macro_window_op_ >= 0) throw ZkError(ZK_ERR_INVALID, "emit_macro_op: the previous macro-op's window is still open");
    try { emit_op(opcode, a, 0, ins, n_in, outs.data(), n_out, nullptr, 0); } catch (...) { allow_macro_ops_ = false; throw; }
    allow_macro_ops_ = false;
    macro_window_op_ = (int32_t)cur().ops.size() - 1;
    macro_window_loop_ = in_loop_;
}

void CS::end_macro_op() {
    if (macro_window_op_ < 0) throw ZkError(ZK_ERR_INVALID, "end_macro_op without emit_macro_op");
    macro_window_op_ = -1;
}

void CS::loop_begin(uint32_t limit) {
    if (in_loop_ || loop_done_) throw ZkError(ZK_ERR_INVALID, "only one loop scope per circuit");
    if (limit == 0) throw ZkError(ZK_ERR_INVALID, "loop limit must be > 0");
    in_loop_ = true;
    limit_ = limit;
    if (outer_.pre_ops == SIZE_MAX) outer_.pre_ops = outer_.ops.size();  // no side phase recorded
    outer_.side_ops = outer_.ops.size();
}

// Everything recorded between side_begin() and loop_begin() is outer-scope work that neither the loop needs
// (it must not be loop_import-ed) nor depends on the loop: the fused pipeline runs it CONCURRENTLY with the
// loop kernel (e.g. the commitments of observable_input / hidden_fsm_input).
void CS::side_begin() {
    if (in_loop_ || loop_done_ || outer_.pre_ops != SIZE_MAX) throw ZkError(ZK_ERR_INVALID, "side_begin: once, before loop_begin");
    outer_.pre_ops = outer_.ops.size();
    pre_vars_ = outer_.n_vars;
}
void CS::loop_end() {
    if (!in_loop_) throw ZkError(ZK_ERR_INVALID, "loop_end without loop_begin");
    in_loop_ = false;
    loop_done_ = true;
}

void CS::link(uint32_t kind, zk_var loop_var, zk_var other) {
    if (kind > ZK_LINK_BCAST) throw ZkError(ZK_ERR_INVALID, "link: bad kind");
    check_var(loop_var, true);
    check_var(other, kind == ZK_LINK_CARRY);
    // the loop (and the seeding pass) only sees outer values of the PRE phase
    if ((kind == ZK_LINK_FIRST || kind == ZK_LINK_BCAST) && var_index(other) >= pre_vars_)
        throw ZkError(ZK_ERR_INVALID, "link: the outer variable is produced after side_begin (side/post phase), not visible to the loop");
    links_raw_.push_back({kind, var_index(loop_var), var_index(other), 0});
}

void CS::seed_hint(uint32_t opcode, const zk_var* ins, uint32_t n_in, const zk_var* outs, uint32_t n_out) {
    if (!in_loop_) throw ZkError(ZK_ERR_INVALID, "seed_hint outside the loop");
    if (!((opcode == ZK_OP_KECCAK_ABSORB && n_in == 336 && n_out == 200) || (opcode == ZK_OP_SHA256_COMPRESS && n_in == 96 && n_out == 32)))
        throw ZkError(ZK_ERR_INVALID, "seed_hint: unknown opcode / arity");
    OpRec op;
    op.opcode = (uint8_t)opcode; op.a = 0; op.b = 0; op.seed_only = true;
    for (uint32_t i = 0; i < n_in; ++i) { check_var(ins[i], true); op.ins.push_back({Operand::VAR, var_index(ins[i])}); }
    for (uint32_t i = 0; i < n_out; ++i) { check_var(outs[i], true); op.outs.push_back(var_index(outs[i])); }
    loop_.ops.push_back(std::move(op));
}

void CS::stream_link(const zk_var* a, uint32_t pa, const zk_var* b, uint32_t pb, uint32_t n_total) {
    if (!in_loop_) throw ZkError(ZK_ERR_INVALID, "stream_link outside the loop");
    if (!pa || !pb || !n_total || (uint64_t)limit_ * pa < n_total || (uint64_t)limit_ * pb < n_total)
        throw ZkError(ZK_ERR_INVALID, "stream_link: n_total exceeds limit * period");
    StreamRec r;
    r.n_total = n_total;
    for (uint32_t i = 0; i < pa; ++i) { check_var(a[i], true); r.a.push_back(var_index(a[i])); }
    for (uint32_t i = 0; i < pb; ++i) { check_var(b[i], true); r.b.push_back(var_index(b[i])); }
    streams_raw_.push_back(std::move(r));
}

zk_var CS::loop_last(zk_var loop_var) {
    if (in_loop_ || !loop_done_) throw ZkError(ZK_ERR_INVALID, "loop_last only after loop_end");
    check_var(loop_var, true);
    zk_var v = alloc_var();
    OpRec op{ZK_OP_LOOP_LAST, 0, 0, {{Operand::RAW, var_index(loop_var)}}, {var_index(v)}};
    outer_.ops.push_back(std::move(op));
    links_raw_.push_back({ZK_LINK_LAST, var_index(loop_var), var_index(v), 0});
    return v;
}

zk_var CS::loop_import(zk_var outer_var) {
    if (!in_loop_) throw ZkError(ZK_ERR_INVALID, "loop_import outside the loop");
    check_var(outer_var, false);
    if (var_index(outer_var) >= pre_vars_) throw ZkError(ZK_ERR_INVALID, "loop_import: the outer variable is produced after side_begin");
    zk_var v = alloc_var();
    // value copy: FMA-free "select"-less move expressed as CONST-like op with an OUTER operand
    OpRec op{ZK_OP_CONST, 0, 0, {{Operand::OUTER_VAR, var_index(outer_var)}}, {var_index(v)}};
    loop_.ops.push_back(std::move(op));
    links_raw_.push_back({ZK_LINK_BCAST, var_index(v), var_index(outer_var), 0});
    return v;
}

uint64_t CS::next_available_row() const {
    if (finalized_) return (uint64_t)loop_.n_slots * limit_ + outer_.n_slots;
    return 0;
}

// ------------------------------------------------------------------ placement
void CS::place_scope(Scope& s) {
    const uint32_t C = geo_.num_columns_under_copy_permutation;
    struct Open { uint32_t slot, used, cap; };
    std::map<std::pair<uint32_t, std::vector<uint64_t>>, Open> open;
    std::vector<std::pair<uint32_t, uint32_t>> tmp_cells;  // (col, slot) per placed cell, per var appended below
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> vc(s.n_vars);
    s.rows.clear(); s.rowconsts.clear(); s.row_gates.clear();
    uint32_t n_gate_slots = 0;
    for (auto& g : s.gates) {
        const GateInfo& gi = GATES[g.kind];
        std::pair<uint32_t, std::vector<uint64_t>> key{g.kind, g.kind == ZK_GATE_CONST ? std::vector<uint64_t>{} : g.consts};
        uint32_t cap = C / gi.width;
        if (g.kind == ZK_GATE_CONST) cap = std::min(cap, geo_.num_constant_columns);
        auto it = open.find(key);
        if (it == open.end() || it->second.used == it->second.cap) {
            Open o{n_gate_slots++, 0, cap};
            zk_row_desc rd{g.kind, 0, (uint32_t)s.rowconsts.size(), g.kind == ZK_GATE_CONST ? 0u : gi.n_consts};
            s.rows.push_back(rd);
            if (g.kind == ZK_GATE_CONST) s.rowconsts.resize(s.rowconsts.size() + cap, 0);
            else s.rowconsts.insert(s.rowconsts.end(), g.consts.begin(), g.consts.end());
            if (it == open.end()) it = open.emplace(key, o).first; else it->second = o;
        }
        Open& o = it->second;
        uint32_t j = o.used++;
        if (s.row_gates.size() <= o.slot) s.row_gates.resize(o.slot + 1);
        s.row_gates[o.slot].push_back((uint32_t)(&g - s.gates.data()));  // instance j of the row == this gate
        zk_row_desc& rd = s.rows[o.slot];
        rd.n_instances = o.used;
        if (g.kind == ZK_GATE_CONST) { s.rowconsts[rd.const_off + j] = g.consts[0]; rd.n_consts = o.used; }
        for (uint32_t c = 0; c < gi.width; ++c) vc[g.vars[c]].push_back({j * gi.width + c, o.slot});
        s.gate_counts[g.kind]++;
        s.n_constraints += gi.n_relations;
    }
    // lookups: one table per row, lookup_reps_ tuples per row
    s.lrows.clear(); s.row_lookups.clear();
    uint32_t n_lookup_slots = 0;
    std::map<uint32_t, Open> lopen;
    for (auto& l : s.lookups) {
        auto it = lopen.find(l.table);
        if (it == lopen.end() || it->second.used == it->second.cap) {
            Open o{n_lookup_slots++, 0, lookup_reps_};
            s.lrows.push_back({l.table, 0});
            if (it == lopen.end()) it = lopen.emplace(l.table, o).first; else it->second = o;
        }
        Open& o = it->second;
        uint32_t u = o.used++;
        s.lrows[o.slot].n_tuples = o.used;
        if (s.row_lookups.size() <= o.slot) s.row_lookups.resize(o.slot + 1);
        s.row_lookups[o.slot].push_back((uint32_t)(&l - s.lookups.data()));   // tuple u of the row == this record
        for (uint32_t c = 0; c < l.vars.size(); ++c) vc[l.vars[c]].push_back({C + u * lookup_width_ + c, o.slot});
        s.n_constraints += 1;
    }
    s.n_gate_slots = n_gate_slots;
    s.n_lookup_slots = n_lookup_slots;
    s.n_slots = std::max<uint32_t>(1, std::max(n_gate_slots, n_lookup_slots));
    s.rows.resize(s.n_slots, zk_row_desc{ZK_GATE_NOP, 0, 0, 0});
    s.lrows.resize(s.n_slots, zk_lookup_row_desc{0xffffffffu, 0});
    const uint32_t total_cols = C + lookup_width_ * lookup_reps_;
    uint64_t ntc = (uint64_t)total_cols * s.n_slots;
    if (ntc >= 0x3fffffffull) throw ZkError(ZK_ERR_CAPACITY, "scope too large for 30-bit cell indices");
    s.n_trace_cells = (uint32_t)ntc;
    s.var_cells.assign(s.n_vars, {});
    s.n_scratch = 0;
    for (uint32_t v = 0; v < s.n_vars; ++v) {
        for (auto& cs : vc[v]) s.var_cells[v].push_back(cs.second * total_cols + cs.first);
        if (s.var_cells[v].empty()) s.var_cells[v].push_back(s.n_trace_cells + s.n_scratch++);
    }
    s.n_cells = s.n_trace_cells + s.n_scratch;
    if (s.n_cells >= 0x3fffffffu) throw ZkError(ZK_ERR_CAPACITY, "scope too large for 30-bit cell indices");
    s.copies.clear();
    for (uint32_t v = 0; v < s.n_vars; ++v)
        for (size_t i = 1; i < s.var_cells[v].size(); ++i) s.copies.push_back({s.var_cells[v][i], s.var_cells[v][0]});
}

// Check program of the compact checker (kernels_engine2.hpp k_check_prog): per row, packets of consecutive gate instances
// (kind | count << 8 | first instance << 16, row, offset of the row constants, then the store slots of their columns) and of
// lookup tuples (0x40 | count << 8 | first tuple << 16, row, table id, 4 slot words per tuple).  A packet fits one 16-word
// scalar fetch (the 24-column matrix gates take two).  Chunks of whole packets, balanced by words, for the launch grid.
void CS::build_check_program(Scope& s) {
    s.cprog.clear(); s.cchunks.clear(); s.cprog_full.clear(); s.cchunks_full.clear(); s.cmacros.clear(); s.n_macro_p2 = 0;
    s.cprog_fused.clear(); s.cchunks_fused.clear();
    const uint32_t C = geo_.num_columns_under_copy_permutation, NC = C + lookup_width_ * lookup_reps_;
    std::vector<uint32_t> starts;
    auto cap_of = [](uint32_t kind) -> uint32_t {
        switch (kind) {
        case ZK_GATE_CONST: case ZK_GATE_BOOLEAN: return 8;
        case ZK_GATE_FMA: case ZK_GATE_SELECT: return 3;
        case ZK_GATE_ZEROCHECK: return 4;
        case ZK_GATE_REDUCTION4: case ZK_GATE_UINTX_ADD: case ZK_GATE_U32_FMA: case ZK_GATE_REDUCTION_BY_POWERS4: return 2;
        case ZK_GATE_DOT4: case ZK_GATE_MATMUL12_EXT: case ZK_GATE_MATMUL12_INT: case ZK_GATE_U8X4_FMA: return 1;
        default: return 0;  // NOP, PUBLIC_INPUT: no relation
        }
    };
    // ---- in-circuit Poseidon2 permutations as MACRO packets.  compute_round_function (gadgets.cpp) constrains the 962 outputs of a
    // ZK_OP_P2_ROUNDS op with 31 matrix gates and 590 FMA gates = 3 104 value references for 974 distinct values; gate by gate the
    // checker re-fetches most of them from HBM (every value sits in two or three rows visited at different times).  A macro packet
    // loads the 12 inputs and the 962 stored outputs ONCE, in order, and evaluates the same 621 relations on the stored values in
    // registers.  The ownership of every gate is verified here against the gadget's structure (variables, constants, the `one` and
    // round-constant variables); anything unexpected leaves the op to the ordinary packets.
    std::vector<int32_t> gate_macro(s.gates.size(), -1);
    struct MacroP2 { uint32_t in_slots[12]; uint32_t first_out; uint32_t first_gate; };
    std::vector<MacroP2> macros;
    {
        std::vector<uint64_t> const_full(s.n_vars, 0);
        std::vector<uint8_t> is_const(s.n_vars, 0);
        for (auto& op : s.ops)
            if (!op.seed_only && op.opcode == ZK_OP_CONST && op.ins[0].kind == Operand::CONSTPOOL) { const_full[op.outs[0]] = s.const_pool[op.ins[0].idx]; is_const[op.outs[0]] = 1; }
        std::unordered_map<uint32_t, uint32_t> fma_by_d, mm_by_out0;   // output variable -> gate index
        for (uint32_t gi = 0; gi < s.gates.size(); ++gi) {
            const GateRec& g = s.gates[gi];
            if (g.kind == ZK_GATE_FMA) { if (fma_by_d.count(g.vars[3])) fma_by_d[g.vars[3]] = UINT32_MAX; else fma_by_d[g.vars[3]] = gi; }
            else if (g.kind == ZK_GATE_MATMUL12_EXT || g.kind == ZK_GATE_MATMUL12_INT) { if (mm_by_out0.count(g.vars[12])) mm_by_out0[g.vars[12]] = UINT32_MAX; else mm_by_out0[g.vars[12]] = gi; }
        }
        const uint64_t* RC = poseidon_round_constants();
        for (auto& op : s.ops) {
            if (op.seed_only || op.opcode != ZK_OP_P2_ROUNDS || op.outs.size() != 962 || op.ins.size() != 12) continue;
            std::vector<uint32_t> owned;
            bool ok = true;
            auto want_mm = [&](uint32_t kind, const uint32_t* in12, const uint32_t* out12) {
                auto it = mm_by_out0.find(out12[0]);
                if (it == mm_by_out0.end() || it->second == UINT32_MAX) { ok = false; return; }
                const GateRec& g = s.gates[it->second];
                if (g.kind != kind) { ok = false; return; }
                for (int i = 0; i < 12; ++i) if (g.vars[i] != in12[i] || g.vars[12 + i] != out12[i]) { ok = false; return; }
                owned.push_back(it->second);
            };
            // an FMA gate (q, l; a, b, c -> d) of the gadget: d identifies it; t_gate: b must be the constant 1 and c the constant `rc`
            auto want_fma = [&](uint64_t q, uint64_t l, uint32_t va, uint32_t vb, uint32_t vc, uint32_t vd, bool t_gate, uint64_t rc) {
                auto it = fma_by_d.find(vd);
                if (it == fma_by_d.end() || it->second == UINT32_MAX) { ok = false; return; }
                const GateRec& g = s.gates[it->second];
                if (g.consts.size() != 2 || g.consts[0] != q || g.consts[1] != l || g.vars[0] != va) { ok = false; return; }
                if (t_gate) {
                    if (!is_const[g.vars[1]] || const_full[g.vars[1]] != 1 || !is_const[g.vars[2]] || const_full[g.vars[2]] != rc) { ok = false; return; }
                } else if (g.vars[1] != vb || g.vars[2] != vc) { ok = false; return; }
                owned.push_back(it->second);
            };
            uint32_t in12[12], cur[12];
            for (int i = 0; i < 12; ++i) { if (op.ins[i].kind != Operand::VAR) ok = false; in12[i] = op.ins[i].idx; }
            if (!ok) continue;
            size_t pos = 0;
            want_mm(ZK_GATE_MATMUL12_EXT, in12, &op.outs[0]);
            for (int i = 0; i < 12; ++i) cur[i] = op.outs[i];
            pos = 12;
            for (int r = 0; r < 30 && ok; ++r) {
                const bool full = r < 4 || r >= 26;
                const int n = full ? 12 : 1;
                for (int i = 0; i < n && ok; ++i) {
                    const uint32_t t = op.outs[pos], x2 = op.outs[pos + 1], x3 = op.outs[pos + 2], x4 = op.outs[pos + 3], x7 = op.outs[pos + 4];
                    want_fma(1, 1, cur[i], 0, 0, t, true, RC[12 * r + i]);
                    want_fma(1, 0, t, t, t, x2, false, 0);
                    want_fma(1, 0, x2, t, t, x3, false, 0);
                    want_fma(1, 0, x2, x2, x2, x4, false, 0);
                    want_fma(1, 0, x3, x4, x3, x7, false, 0);
                    cur[i] = x7;
                    pos += 5;
                }
                if (!ok) break;
                want_mm(full ? ZK_GATE_MATMUL12_EXT : ZK_GATE_MATMUL12_INT, cur, &op.outs[pos]);
                for (int i = 0; i < 12; ++i) cur[i] = op.outs[pos + i];
                pos += 12;
            }
            if (!ok || pos != 962 || owned.size() != 621) continue;
            for (size_t q = 0; q < 962; ++q) if (s.var_slot[op.outs[q]] != s.var_slot[op.outs[0]] + q) { ok = false; break; }
            if (!ok) continue;
            for (uint32_t gi : owned) if (gate_macro[gi] >= 0) { ok = false; break; }
            if (!ok) continue;
            MacroP2 m;
            for (int i = 0; i < 12; ++i) m.in_slots[i] = s.var_slot[in12[i]];
            m.first_out = s.var_slot[op.outs[0]];
            m.first_gate = owned[0];
            for (uint32_t gi : owned) gate_macro[gi] = (int32_t)macros.size();
            macros.push_back(m);
        }
    }
    // ---- gates MIRRORED by the witness op that produces their output: the gadget layer emits `op; gate` pairs over the same variables
    // and constants (gadgets.cpp: fma, linear_combination, select, is_zero, dot4, u32 add / fma-with-carry, allocate_constant), so the
    // relation IS the op's field arithmetic on the registers it stores (FMA, linear combinations, dot products, zero checks with their
    // exact inverse, constants) for every input whatsoever; SELECT's relation s (a - b) + b - r equals its op (r = s ? a : b) unless
    // s > 1 and a != b, which the witness kernels test on the operands they hold (kernels_engine2.hpp) and report like a macro packet.
    // So in the fused mode of resolve_and_check these gates are evaluated where their values are produced, and the check program
    // keeps every OTHER gate — enforcements, booleans of inputs, integer add / multiply relations, relations whose output is a given
    // variable.  Lookups: the op that looks the keys up reports a miss (below).  Stored values (after write_cell / poke, or on request) are verified by the full
    // programs as before.
    s.gate_mirrored.assign(s.gates.size(), 0);
    {
        std::vector<int32_t> producer(s.n_vars, -1);
        for (size_t oi = 0; oi < s.ops.size(); ++oi)
            if (!s.ops[oi].seed_only) for (uint32_t ov : s.ops[oi].outs) producer[ov] = (int32_t)oi;
        auto is_var = [](const Operand& o, uint32_t v) { return o.kind == Operand::VAR && o.idx == v; };
        auto pool = [&](const Operand& o, uint64_t& out) { if (o.kind != Operand::CONSTPOOL) return false; out = s.const_pool[o.idx]; return true; };
        // the gate was placed by the gadget of THIS macro-op inside its window (owner = the op's first output variable: stable under the schedulers)
        auto owned_by = [](const GateRec& g, const OpRec* op) { return g.owner >= 0 && !op->outs.empty() && (uint32_t)g.owner == op->outs.front(); };
        for (size_t gi = 0; gi < s.gates.size(); ++gi) {
            const GateRec& g = s.gates[gi];
            bool m = false;
            auto prod = [&](uint32_t v) -> const OpRec* { return producer[v] >= 0 ? &s.ops[producer[v]] : nullptr; };
            switch (g.kind) {
            case ZK_GATE_FMA: {
                const OpRec* op = prod(g.vars[3]);
                // inside the ByteBuffer macro-op every FMA / Selection / ZeroCheck gate is placed by the gadget from the structure the op
                // walks (bytebuf_macro.hpp): the op computes exactly these relations on the values it stores; its selectors are its own
                // is-zero flags and their and / or / not, 0 / 1 for every input.  "Placed by the gadget" is recorded, not assumed: the gate
                // carries the op it was placed for (GateRec::owner, set only inside the emit_macro_op .. end_macro_op window); a gate
                // somebody else places on a macro output (zk_cs_place_gate is public) has no owner and stays in the check program
                if (op && op->opcode == ZK_OP_BYTEBUF_FILL) { m = owned_by(g, op); break; }
                // the 4-bit-chunk SHA-256 macro-op (a = 1, sha256_macro4.hpp): rotated nibbles hi + 2^(4-s) lo, x & 7 = 2 ((x >> 1) & 3) + (x & 1), the
                // byte recompositions and 2^32 carry + low == sum are identities of the op's integer arithmetic on the values it stores
                if (op && op->opcode == ZK_OP_SHA256_ROUNDS && op->a == 1) { m = owned_by(g, op); break; }
                uint64_t q, l;
                m = op && op->opcode == ZK_OP_FMA && op->ins.size() == 5 && pool(op->ins[0], q) && pool(op->ins[1], l) && q == g.consts[0] && l == g.consts[1] &&
                    is_var(op->ins[2], g.vars[0]) && is_var(op->ins[3], g.vars[1]) && is_var(op->ins[4], g.vars[2]);
            } break;
            case ZK_GATE_REDUCTION4: case ZK_GATE_REDUCTION_BY_POWERS4: {
                const OpRec* op = prod(g.vars[4]);
                // a rotated byte inside the Keccak macro-op: lo 2^b + hi computed by the op from the very lo / hi it stores (keccak_macro.hpp
                // rotl); only the engine's gadget can record the op (emit_macro_op), and it places this gate from the same structure
                if (op && (op->opcode == ZK_OP_KECCAK_F || op->opcode == ZK_OP_SHA256_ROUNDS)) { m = g.kind == ZK_GATE_REDUCTION4 && owned_by(g, op); break; }
                if (op && op->opcode == ZK_OP_LC4 && op->ins.size() == 8) {
                    m = true;
                    uint64_t pw = 1;
                    for (int i = 0; i < 4 && m; ++i) {
                        uint64_t k;
                        const uint64_t want = g.kind == ZK_GATE_REDUCTION4 ? g.consts[i] : pw;
                        m = pool(op->ins[i], k) && k == want && is_var(op->ins[4 + i], g.vars[i]);
                        if (g.kind != ZK_GATE_REDUCTION4) pw = (uint64_t)((unsigned __int128)pw * g.consts[0] % 0xFFFFFFFF00000001ull);
                    }
                }
            } break;
            case ZK_GATE_SELECT: {
                const OpRec* op = prod(g.vars[3]);
                if (op && op->opcode == ZK_OP_BYTEBUF_FILL) { m = owned_by(g, op); break; }
                m = op && op->opcode == ZK_OP_SELECT && op->ins.size() == 3 && is_var(op->ins[0], g.vars[2]) && is_var(op->ins[1], g.vars[0]) && is_var(op->ins[2], g.vars[1]);
            } break;
            case ZK_GATE_ZEROCHECK: {
                const OpRec* op = prod(g.vars[2]);
                if (op && op->opcode == ZK_OP_BYTEBUF_FILL) { m = owned_by(g, op) && producer[g.vars[1]] == producer[g.vars[2]]; break; }
                m = op && op->opcode == ZK_OP_ISZERO && is_var(op->ins[0], g.vars[0]) && op->outs[0] == g.vars[2] && op->outs[1] == g.vars[1];
            } break;
            case ZK_GATE_DOT4: {
                const OpRec* op = prod(g.vars[8]);
                m = op && op->opcode == ZK_OP_DOT4 && op->ins.size() == 8;
                for (int i = 0; i < 8 && m; ++i) m = is_var(op->ins[i], g.vars[i]);
            } break;
            // ZK_GATE_UINTX_ADD / ZK_GATE_U32_FMA: their ops work on integers — the field relation equals the op only for operands in
            // range, which other gates establish: they stay in the check program
            case ZK_GATE_CONST: {
                const OpRec* op = prod(g.vars[0]);
                uint64_t c;
                m = op && op->opcode == ZK_OP_CONST && pool(op->ins[0], c) && c == g.consts[0];
            } break;
            case ZK_GATE_BOOLEAN: {   // a flag some op produces as 0 / 1 for EVERY input: the is-zero flag, a masked 1-bit chunk of a SPLIT.
                // NOT the last chunk of a SPLIT: it keeps the residual x >> (n-1) unmasked (kernels_engine2.hpp ZK_OP_SPLIT), which is
                // 0 / 1 only when x < 2^n — exactly what this gate is placed to enforce (spread_into_bits: the range check of x)
                const OpRec* op = prod(g.vars[0]);
                m = op && ((op->opcode == ZK_OP_ISZERO && op->outs[0] == g.vars[0]) ||
                           (op->opcode == ZK_OP_SPLIT && op->b == 1 && !op->outs.empty() && op->outs.back() != g.vars[0]));
            } break;
            default: break;
            }
            if (gate_macro[gi] >= 0) m = true;   // the Poseidon2 gadget's gates: mirrored by ZK_OP_P2_ROUNDS (verified above)
            s.gate_mirrored[gi] = m ? 1 : 0;
        }
        if (getenv("ZKGL_PROG_STATS")) {
            uint64_t tot[ZK_GATE__COUNT] = {0}, mir[ZK_GATE__COUNT] = {0}, refs_tot = 0, refs_left = 0;
            for (size_t gi = 0; gi < s.gates.size(); ++gi) {
                tot[s.gates[gi].kind]++; mir[s.gates[gi].kind] += s.gate_mirrored[gi];
                refs_tot += s.gates[gi].vars.size(); if (!s.gate_mirrored[gi]) refs_left += s.gates[gi].vars.size();
            }
            fprintf(stderr, "[zkgl] %s scope gates mirrored by their producing op (kind: mirrored / total):", s.is_loop ? "loop" : "outer");
            for (int k = 1; k < ZK_GATE__COUNT; ++k) if (tot[k]) fprintf(stderr, " %d: %llu/%llu", k, (unsigned long long)mir[k], (unsigned long long)tot[k]);
            uint64_t lk = 0; for (auto& l : s.lookups) lk += l.vars.size();
            fprintf(stderr, "; gate references %llu -> %llu left, lookup references %llu\n", (unsigned long long)refs_tot, (unsigned long long)refs_left, (unsigned long long)lk);
        }
    }
    bool lookups_ok = true;
    // mode 0: every gate instance on its own; 1: Poseidon2 gadgets as macro packets (k_check_p2); 2: fused — the gates mirrored by their
    // producing op (Poseidon2 gadgets included) are left to the witness kernels, no macro packets
    auto emit = [&](std::vector<uint32_t>& prog, std::vector<uint32_t>& chunks, int mode) {
        const bool use_macros = mode == 1;
        auto skipped = [&](uint32_t slot, uint32_t j) -> bool {
            if (mode == 0 || slot >= s.row_gates.size() || j >= s.row_gates[slot].size()) return false;
            const uint32_t gi = s.row_gates[slot][j];
            return mode == 1 ? gate_macro[gi] >= 0 : s.gate_mirrored[gi] != 0;
        };
        starts.clear();
        std::vector<uint8_t> macro_done(macros.size(), 0);
        for (uint32_t slot = 0; slot < s.n_slots; ++slot) {
            const zk_row_desc& rd = s.rows[slot];
            const uint32_t cap = rd.kind < ZK_GATE__COUNT ? cap_of(rd.kind) : 0, w = rd.kind < ZK_GATE__COUNT ? GATES[rd.kind].width : 0;
            uint32_t j = 0;
            while (cap && j < rd.n_instances) {
                if (mode == 2 && skipped(slot, j)) { ++j; continue; }
                const int32_t mi = use_macros && slot < s.row_gates.size() && j < s.row_gates[slot].size() ? gate_macro[s.row_gates[slot][j]] : -1;
                if (mi >= 0) {   // owned by a macro packet: emitted once, where its first gate sits
                    if (!macro_done[mi]) {   // the permutation goes to k_check_p2: descriptor = inputs, first output, row of its first gate
                        macro_done[mi] = 1;
                        for (int i = 0; i < 12; ++i) s.cmacros.push_back(macros[mi].in_slots[i]);
                        s.cmacros.push_back(macros[mi].first_out);
                        s.cmacros.push_back(slot);
                    }
                    ++j;
                    continue;
                }
                uint32_t cnt = 0;
                while (cnt < cap && j + cnt < rd.n_instances && !skipped(slot, j + cnt)) ++cnt;
                starts.push_back((uint32_t)prog.size());
                prog.push_back(rd.kind | (cnt << 8) | (j << 16));
                prog.push_back(slot);
                prog.push_back(rd.const_off);
                for (uint32_t g = 0; g < cnt; ++g)
                    for (uint32_t c = 0; c < w; ++c) prog.push_back(s.alias[(size_t)slot * NC + (j + g) * w + c]);
                j += cnt;
            }
            const zk_lookup_row_desc& lr = s.lrows[slot];
            if (lr.table == 0xffffffffu || lr.n_tuples == 0) continue;
            const TableRec& t = tables_[lr.table - 1];
            const uint32_t tw = t.n_keys + t.n_vals;
            if (tw > 4 || t.n_keys > 3) { lookups_ok = false; return; }  // the row-descriptor checker handles such scopes
            // fused: a tuple recorded by CS::lookup is (the keys of a ZK_OP_LOOKUP, the fresh variables that op fills from the row it found), a
            // tuple given inside a macro-op's window is (keys, values the op computes from them) — either is a table row iff the op found /
            // accepted its keys: the witness kernel reports a miss.  LookupRec::owner says so; a tuple nobody owns (lookup_given over
            // variables of other producers) is read from the store like in the other modes.
            auto owned = [&](uint32_t u) { return mode == 2 && slot < s.row_lookups.size() && u < s.row_lookups[slot].size() && s.lookups[s.row_lookups[slot][u]].owner >= 0; };
            for (uint32_t u0 = 0; u0 < lr.n_tuples;) {
                if (owned(u0)) { ++u0; continue; }
                uint32_t cnt = 0;
                while (cnt < 3 && u0 + cnt < lr.n_tuples && !owned(u0 + cnt)) ++cnt;
                starts.push_back((uint32_t)prog.size());
                prog.push_back(0x40u | (cnt << 8) | (u0 << 16));
                prog.push_back(slot);
                prog.push_back(lr.table);
                for (uint32_t g = 0; g < cnt; ++g)
                    for (uint32_t i = 0; i < 4; ++i) prog.push_back(i < tw ? s.alias[(size_t)slot * NC + C + (u0 + g) * lookup_width_ + i] : 0u);
                u0 += cnt;
            }
        }
        if (starts.empty()) { prog.clear(); return; }
        // chunks of whole packets, balanced by words
        std::vector<uint64_t> weight(starts.size() + 1, 0);
        for (size_t p = 0; p < starts.size(); ++p) {
            const uint32_t end = p + 1 < starts.size() ? starts[p + 1] : (uint32_t)prog.size();
            weight[p + 1] = weight[p] + (end - starts[p]);
        }
        const uint64_t total_w = weight.back();
        const uint32_t n_chunks = (uint32_t)std::min<size_t>(256, starts.size());
        chunks.push_back(0);
        size_t p = 0;
        for (uint32_t cidx = 1; cidx < n_chunks; ++cidx) {
            const uint64_t target = total_w * cidx / n_chunks;
            while (p + 1 < starts.size() && weight[p] < target) ++p;
            if (starts[p] > chunks.back()) chunks.push_back(starts[p]);
        }
        chunks.push_back((uint32_t)prog.size());
    };
    emit(s.cprog, s.cchunks, 1);
    if (lookups_ok && !macros.empty()) emit(s.cprog_full, s.cchunks_full, 0);
    s.cprog_fused.clear(); s.cchunks_fused.clear();
    if (lookups_ok) {
        emit(s.cprog_fused, s.cchunks_fused, 2);
        if (s.cprog_fused.empty()) { s.cprog_fused.assign(1, 0); s.cchunks_fused = {0, 0}; }   // nothing left to read: an empty program, not "no program"
    }
    if (!lookups_ok) { s.cprog.clear(); s.cchunks.clear(); s.cprog_full.clear(); s.cchunks_full.clear(); s.cmacros.clear(); s.cprog_fused.clear(); s.cchunks_fused.clear(); return; }
    s.n_macro_p2 = (uint32_t)macros.size();
    s.n_p2_rounds_ops = 0;
    for (auto& op : s.ops) s.n_p2_rounds_ops += (!op.seed_only && op.opcode == ZK_OP_P2_ROUNDS);
    if (getenv("ZKGL_PROG_STATS"))
        fprintf(stderr, "[zkgl] %s scope check program: %zu words, %u Poseidon2 macro packets (gate by gate: %zu words)\n", s.is_loop ? "loop" : "outer", s.cprog.size(),
                s.n_macro_p2, s.cprog_full.size());
}

// Census for the 4-byte-slot store: an upper bound per variable that holds in EVERY satisfying witness, derived from the
// constraints alone (never from witness ops): constant gates, boolean gates, the is-zero flag, membership in a lookup table column,
// reductions / FMA gates over bounded terms that cannot wrap, selections by a boolean-constrained selector.  A variable with bound
// <= 2^32 could live in a 4-byte slot: a witness that puts a larger value there is unsatisfiable anyway (the store would report it).
void CS::bound_values(Scope& s) {
    typedef unsigned __int128 u128;
    const u128 INF = ~(u128)0, PP = (u128)0xFFFFFFFF00000001ull;
    std::vector<u128> ub(s.n_vars, INF);   // exclusive upper bound
    auto lower = [&](uint32_t v, u128 b, bool& changed) { if (b < ub[v]) { ub[v] = b; changed = true; } };
    std::vector<std::vector<uint64_t>> colmax(tables_.size());
    for (size_t t = 0; t < tables_.size(); ++t) {
        const TableRec& tr = tables_[t];
        const uint32_t w = tr.n_keys + tr.n_vals;
        colmax[t].assign(w, 0);
        for (uint32_t r = 0; r < tr.n_rows; ++r)
            for (uint32_t c = 0; c < w; ++c) colmax[t][c] = std::max(colmax[t][c], tr.rows[(size_t)r * w + c]);
    }
    bool changed = true;
    for (int round = 0; round < 64 && changed; ++round) {
        changed = false;
        for (auto& l : s.lookups)
            for (size_t c = 0; c < l.vars.size() && c < colmax[l.table - 1].size(); ++c) lower(l.vars[c], (u128)colmax[l.table - 1][c] + 1, changed);
        for (auto& g : s.gates) {
            switch (g.kind) {
            case ZK_GATE_CONST: lower(g.vars[0], (u128)g.consts[0] + 1, changed); break;
            case ZK_GATE_BOOLEAN: lower(g.vars[0], 2, changed); break;
            case ZK_GATE_ZEROCHECK: lower(g.vars[2], 2, changed); break;   // x aux = 1 - flag, flag x = 0: flag is 0 / 1
            case ZK_GATE_SELECT:   // (a, b, s, r): r in {a, b} once s is 0 / 1
                if (ub[g.vars[2]] <= 2 && ub[g.vars[0]] != INF && ub[g.vars[1]] != INF) lower(g.vars[3], std::max(ub[g.vars[0]], ub[g.vars[1]]), changed);
                break;
            case ZK_GATE_REDUCTION4: case ZK_GATE_REDUCTION_BY_POWERS4: {
                u128 sum = 0, pw = 1;
                bool ok = true;
                for (int i = 0; i < 4 && ok; ++i) {
                    const u128 k = g.kind == ZK_GATE_REDUCTION4 ? (u128)g.consts[i] : pw;
                    if (g.kind != ZK_GATE_REDUCTION4) pw = pw * g.consts[0] % PP;
                    if (k == 0) continue;
                    if (ub[g.vars[i]] == INF) { ok = false; break; }
                    sum += k * (ub[g.vars[i]] - 1);
                    if (sum >= PP) ok = false;
                }
                if (ok) lower(g.vars[4], sum + 1, changed);
            } break;
            case ZK_GATE_FMA: {   // q a b + l c = d
                const u128 q = g.consts[0], l = g.consts[1];
                u128 sum = 0;
                bool ok = true;
                if (q) { if (ub[g.vars[0]] == INF || ub[g.vars[1]] == INF) ok = false; else { const u128 pr = (ub[g.vars[0]] - 1) * (ub[g.vars[1]] - 1); if (pr >= PP || q * pr >= PP) ok = false; else sum += q * pr; } }
                if (ok && l) { if (ub[g.vars[2]] == INF || l * (ub[g.vars[2]] - 1) >= PP) ok = false; else sum += l * (ub[g.vars[2]] - 1); }
                if (ok && sum < PP) lower(g.vars[3], sum + 1, changed);
            } break;
            default: break;
            }
        }
    }
    s.values_below_2_32 = s.values_below_2_8 = 0;
    s.value_class.assign(s.n_vars, 0);
    for (uint32_t v = 0; v < s.n_vars; ++v) {
        s.values_below_2_32 += ub[v] <= ((u128)1 << 32); s.values_below_2_8 += ub[v] <= 256;
        s.value_class[v] = ub[v] <= 256 ? 2 : ub[v] <= ((u128)1 << 32) ? 1 : 0;
    }
    if (getenv("ZKGL_PROG_STATS"))
        fprintf(stderr, "[zkgl] %s scope: %u of %u variables are < 2^32 in every satisfying witness (%u of them < 2^8): candidates for 4-byte store slots\n",
                s.is_loop ? "loop" : "outer", s.values_below_2_32, s.n_vars, s.values_below_2_8);
}

// NARROW STORE of the loop scope (store_geom.hpp; a batch uses it when ZKGL_NARROW_STORE=1 is set at zk_cs_set_batch).  bound_values says which variables are bytes in
// every satisfying witness; a byte-class value takes ONE unit of its wavefront's tile (one byte per lane) instead of eight.  The class is a
// storage decision of the host: a variable is byte-class only when (a) its bound is <= 2^8, (b) the op producing it is one whose handler in the
// narrow kernel stores through the class word (narrow_capable), (c) it is among the first 31 outputs of its op.  A witness that puts a larger
// value there is unsatisfiable; the kernel reports the overflow as the fused mode's failure and the step is repeated over the ordinary store
// (CS::resolve_and_check), so verdicts and reported gates are those of the ordinary store.  Units are assigned in slot (= production) order,
// no padding: a wavefront still streams its results out front to back.
static bool narrow_capable(uint32_t opcode) {
    switch (opcode) {
    case ZK_OP_CONST: case ZK_OP_INPUT: case ZK_OP_FMA: case ZK_OP_LC4: case ZK_OP_SELECT: case ZK_OP_ISZERO: case ZK_OP_UADD: case ZK_OP_USUB:
    case ZK_OP_SPLIT: case ZK_OP_LOOKUP: case ZK_OP_DIVREM: case ZK_OP_U8X4FMA: return true;
    default: return false;
    }
}
void CS::build_narrow_layout(Scope& s) {
    s.narrow_ok = false; s.slot_aw.clear(); s.narrow_units = 0; s.narrow_byte_values = 0;
    narrow_enabled_ = false;
    // the layout and its programs are built for every loop scope that can use them (host memory only: ~4 B per program word); a BATCH takes
    // the narrow store when ZKGL_NARROW_STORE=1 is set at zk_cs_set_batch.  ZKGL_NARROW_STORE=0 at finalize: no layout at all.
    const char* e = getenv("ZKGL_NARROW_STORE");
    if ((e && e[0] == '0') || !s.is_loop || !limit_ || s.uses_bigint) return;
    for (auto& op : s.ops)   // macro-ops stream their outputs through their own store paths (and their circuits run in strand form)
        if (!op.seed_only && (op.opcode == ZK_OP_KECCAK_F || op.opcode == ZK_OP_SHA256_ROUNDS || op.opcode == ZK_OP_BYTEBUF_FILL || op.opcode == ZK_OP_NN_MULMOD)) return;
    std::vector<uint8_t> is_byte(s.n_store, 0);
    for (auto& op : s.ops) {
        if (op.seed_only || !narrow_capable(op.opcode)) continue;
        for (size_t q = 0; q < op.outs.size() && q < 31; ++q)
            if (s.value_class[op.outs[q]] == 2) is_byte[s.var_slot[op.outs[q]]] = 1;
    }
    s.slot_aw.assign(s.n_store, 0);
    uint64_t unit = 0;
    for (uint32_t slot = 0; slot < s.n_store; ++slot) {
        if (unit >= zkgeom::AW_MASK - 8) { s.slot_aw.clear(); return; }   // (a tile beyond 2^28 units: 64-bit addressing anyway)
        s.slot_aw[slot] = (uint32_t)unit | (is_byte[slot] ? zkgeom::AW_BYTE : 0u);
        unit += is_byte[slot] ? 1 : 8;
        s.narrow_byte_values += is_byte[slot];
    }
    s.narrow_units = (uint32_t)unit;
    if (getenv("ZKGL_PROG_STATS"))
        fprintf(stderr, "[zkgl] loop scope narrow store: %u of %u values in one-byte slots, %u B per lane instead of %u (%.3f)\n", s.narrow_byte_values, s.n_store, s.narrow_units,
                s.n_store * 8, (double)s.narrow_units / (8.0 * s.n_store));
}
// the fused check program over the narrow store: the same packets (kinds, counts, rows, constants, chunk table) with address words for the slots
void CS::build_narrow_check_program(Scope& s) {
    s.cprog_fused_n.clear();
    if (s.slot_aw.empty() || s.prog2n.empty() || s.cprog_fused.empty()) { s.slot_aw.clear(); s.prog2n.clear(); return; }
    const std::vector<uint32_t>& src = s.cprog_fused;
    std::vector<uint32_t> out(src);
    size_t pc = 0;
    if (!(src.size() == 1 && src[0] == 0))   // (an empty program is the single word 0)
        while (pc < src.size()) {
            const uint32_t kind = src[pc] & 0xff, cnt = (src[pc] >> 8) & 0xff;
            const uint32_t w = kind == 0x40u ? 4u : (kind < ZK_GATE__COUNT ? GATES[kind].width : 0u);
            if (!w || !cnt || pc + 3 + (size_t)cnt * w > src.size()) throw ZkError(ZK_ERR_INVALID, "internal: malformed fused check program");
            for (size_t q = pc + 3; q < pc + 3 + (size_t)cnt * w; ++q) {
                if (src[q] >= s.slot_aw.size()) throw ZkError(ZK_ERR_INVALID, "internal: fused check program names a slot outside the store");
                out[q] = s.slot_aw[src[q]];
            }
            pc += 3 + (size_t)cnt * w;
        }
    s.cprog_fused_n = std::move(out);
    s.narrow_ok = true;
    narrow_enabled_ = true;
}

// Lookup sites of a scope grouped by table (k_multiplicities): the key slots of every recorded lookup
void CS::build_mult_sites(Scope& s) {
    const size_t nt = tables_.size() + 1;  // table ids are 1-based
    std::vector<std::vector<uint32_t>> by_table(nt);
    for (auto& l : s.lookups) {   // every recorded tuple: those of ZK_OP_LOOKUP and those placed on a macro-op's outputs (lookup_given)
        const uint32_t tid = l.table;
        if (tid == 0 || tid >= nt) throw ZkError(ZK_ERR_INVALID, "internal: lookup into an unknown table");
        const uint32_t nk = tables_[tid - 1].n_keys;
        for (size_t q = 0; q < 3; ++q) by_table[tid].push_back(q < nk ? s.var_slot[l.vars[q]] : 0xffffffffu);
    }
    s.mult_sites.clear(); s.mult_site_off.assign(nt + 1, 0);
    for (size_t t = 0; t < nt; ++t) {
        s.mult_site_off[t] = (uint32_t)(s.mult_sites.size() / 3);
        s.mult_sites.insert(s.mult_sites.end(), by_table[t].begin(), by_table[t].end());
    }
    s.mult_site_off[nt] = (uint32_t)(s.mult_sites.size() / 3);
}

// Two ways to the lookup multiplicities.  INLINE: the witness kernels add as they go, one atomic per distinct row among a
// wavefront's 64 lanes — cheap when the lanes (consecutive cycles of one instance) mostly look up the same rows (main_vm: a
// cleared flag, a zero limb; < 1 ms of the loop kernel), ruinous when they do not (hash circuits: byte tables over pseudo-random
// state, 64 lanes = 64 rows, ~1e9 L2 atomics per step = 60 % of keccak's loop kernel).  PASS: no atomics in the witness kernels,
// k_multiplicities recounts from the stored keys afterwards (keccak 57.7 -> 22.9 ms loop kernel + 6 ms pass; but +5 ms for
// main_vm at B=384 against 1.5 ms of inline atomics).  ZKGL_MULT_MODE=inline|pass forces one; default: by the number of lookups a
// loop lane makes (> 2 048: a hash-style circuit -> PASS).
bool CS::inline_multiplicities() const {
    const char* e = getenv("ZKGL_MULT_MODE");
    if (e && e[0] == 'i') return true;
    if (e && e[0] == 'p') return false;
    if (uses_lookup_macros_) return false;   // tuples evaluated inside a macro-op: no per-tuple atomics there
    return (limit_ ? loop_ : outer_).lookups.size() <= 2048;
}

// multiplicities of the batch from the resolved variable stores (PASS mode): per scope, per table
void CS::count_multiplicities(void* stream, int scopes) {
    if (inline_multiplicities() || !total_table_rows_) return;
    for (int sc = 0; sc < 2; ++sc) {
        if (!(scopes & (1 << sc))) continue;   // bit 0: outer scope, bit 1: loop scope
        const Scope& s = sc ? loop_ : outer_;
        if ((sc && !limit_) || !s.d_mult_sites) continue;
        for (size_t t = 1; t <= tables_.size(); ++t) {
            const uint32_t n = s.mult_site_off[t + 1] - s.mult_site_off[t];
            if (!n) continue;
            dev_check(zkdev::launch_multiplicities(s.d_store, s.store_geom(), sc ? limit_ : 1, s.n_lanes, batch_, s.d_mult_sites + 3 * (size_t)s.mult_site_off[t], n,
                                                   tdesc_host_[t], d_table_words_, d_mult_, total_table_rows_, stream));
        }
    }
}

// Store slots in production order of the FINAL op order (after scheduling); variables no op produces keep slot 0 and are
// reported by emit_scope.  Also the trace-cell -> slot alias of the compact gate checker and the materialisation list.
void CS::assign_store_slots(Scope& s) {
    s.var_slot.assign(s.n_vars, 0);
    std::vector<uint8_t> seen(s.n_vars, 0);
    uint32_t next = 0;
    for (auto& op : s.ops) {
        if (op.seed_only) continue;
        for (uint32_t ov : op.outs)
            if (!seen[ov]) { seen[ov] = 1; s.var_slot[ov] = next++; }
    }
    for (uint32_t v = 0; v < s.n_vars; ++v)
        if (!seen[v]) s.var_slot[v] = next++;
    s.n_store = std::max<uint32_t>(next, 1);
    s.alias.assign(s.n_trace_cells, 0);
    s.mat_pairs.clear();
    for (uint32_t v = 0; v < s.n_vars; ++v)
        for (uint32_t c : s.var_cells[v]) {
            if (c < s.n_trace_cells) s.alias[c] = s.var_slot[v];
            s.mat_pairs.push_back({c, s.var_slot[v]});
        }
}

// ------------------------------------------------------------------ op scheduling (loop scope)
// The interpreter kernel is bound by two resources used in long separate phases when ops run in recording order:
// VALU (Poseidon2 permutations: ~15k instructions each, the witness-only ones store 12 words) and the HBM write path
// (everything else: ~1 instruction per stored word).  All waves run the same program nearly in lockstep, so the phases
// of different waves coincide and the two resources idle in turn (measured: VALU busy 49 %, stores at 65 % of the
// achievable write rate).  This pass reorders the ops — any topological order of the value dependencies fills the same
// cells — so that every prefix of the program has used both resources in proportion: a greedy list scheduler that
// always emits the ready op bringing |ALU_done/ALU_total - MEM_done/MEM_total| closest to zero, ties to recording order.
static uint32_t group_cap(const OpRec& op, bool v2);
void CS::schedule_loop_ops() {
    Scope& s = loop_;
    if (!limit_ || s.ops.size() < 3) return;
    const size_t n = s.ops.size();
    auto alu_cost = [](const OpRec& op) -> double {
        switch (op.opcode) {
        case ZK_OP_POSEIDON2: case ZK_OP_P2_ROUNDS: return 14600;
        case ZK_OP_ISZERO: return 2300;
        case ZK_OP_NN_MULMOD: return 3000;
        case ZK_OP_U256_DIVREM: return 8000;
        case ZK_OP_U256_MULWIDE: return 300;
        case ZK_OP_MATMUL12: return 250;
        case ZK_OP_LOOKUP: return 60;
        default: return 15;
        }
    };
    std::vector<double> a(n), m(n);
    double a_tot = 0, m_tot = 0;
    for (size_t i = 0; i < n; ++i) {
        const OpRec& op = s.ops[i];
        double stores = 0;
        stores += (double)op.outs.size();
        a[i] = alu_cost(op);
        m[i] = stores + 0.25 * (double)op.ins.size();
        a_tot += a[i]; m_tot += m[i];
    }
    bool any_heavy = false;
    for (size_t i = 0; i < n; ++i) any_heavy |= a[i] > 1000;
    if (!any_heavy || a_tot <= 0 || m_tot <= 0) return;
    // dependencies through loop variables
    std::vector<uint32_t> producer(s.n_vars, UINT32_MAX), hint_producer(s.n_vars, UINT32_MAX);
    for (size_t i = 0; i < n; ++i)
        for (auto ov : s.ops[i].outs) (s.ops[i].seed_only ? hint_producer : producer)[ov] = (uint32_t)i;
    std::vector<std::vector<uint32_t>> succ(n);
    std::vector<uint32_t> n_pred(n, 0);
    for (size_t i = 0; i < n; ++i) {
        std::vector<uint32_t> ps;
        for (auto& in : s.ops[i].ins)
            if (in.kind == Operand::VAR) {
                if (producer[in.idx] != UINT32_MAX) ps.push_back(producer[in.idx]);
                if (hint_producer[in.idx] != UINT32_MAX && hint_producer[in.idx] != i) ps.push_back(hint_producer[in.idx]);
            }
        if (s.ops[i].seed_only)  // a hint runs after the decomposed producers of its outputs (keeps it next to them)
            for (auto ov : s.ops[i].outs)
                if (producer[ov] != UINT32_MAX) ps.push_back(producer[ov]);
        std::sort(ps.begin(), ps.end());
        ps.erase(std::unique(ps.begin(), ps.end()), ps.end());
        for (auto p : ps) {
            if (p >= i) return;  // not in dependency order: leave the program alone (emit_scope reports it)
            succ[p].push_back((uint32_t)i);
            ++n_pred[i];
        }
    }
    schedule_by_locality(a, m, a_tot, m_tot, succ, n_pred);   // (rounds 1-2 kept the light ops in recording order; measured and retired, profiles/r2_summary.md)
}

// The default schedule: the same resource balance for the heavy ops, but among the ready light ops the one whose operands were touched
// most recently goes first (consumers of a value cluster behind its producer and behind each other), continuing the open group of
// same-kind ops when it can; ties to recording order.  Why: a wavefront of k_witness_loop can count on ~16 values (8 KB) of L2, and in
// recording order 10.3 k of a VM cycle's 18.8 k operand reads come back later than that (LRU model, ZKGL_PROG_STATS=1; measured
// FETCH_SIZE agrees); this order leaves 6.6 k.  A window of 16 touches is the flat optimum of the model (12..20: 6.6-6.8 k; 8: 7.4 k,
// 48: 8.2 k) and of the kernel (B=384, one box: recording order 43.5 ms, windows 8 / 16 / 48: 42.2 / 40.0 / 42.9 ms).  Weighting misses,
// a last-consumer bonus or a larger group bonus change the model by < 2 %.
// (Round 5 carried two more passes here — mux chains of SELECTs as one op, and the gated witness-only permutations of a dependency level under
// one header executed in rounds — as compile-time variants that no device ever ran; round 6 deleted both with their kernel halves.  What they were
// after is in profiles/r5_iszero_stats.json; by the elimination runs of profiles/r3_loop_probe.md the kernel's memory traffic binds first.)

// SELECT flags the plain loop kernel keeps as bit planes (ZK_OP_FLAG_PLANES): the FLAG_PLANES most used flag variables of a loop scope
// that at least two SELECTs read.  One rule for the scheduler (such a read costs no operand fetch) and for emit_scope (the plane ids).
std::vector<uint32_t> CS::select_plane_vars(const Scope& s) const {
    std::vector<uint32_t> plane_of(s.n_vars, UINT32_MAX);
    const char* fp_env = getenv("ZKGL_FLAG_PLANES");
    if (!s.is_loop || (fp_env && fp_env[0] == '0')) return plane_of;
    std::vector<uint32_t> uses(s.n_vars, 0);
    for (auto& op : s.ops)
        if (!op.seed_only && op.opcode == ZK_OP_SELECT && op.ins[0].kind == Operand::VAR) ++uses[op.ins[0].idx];
    std::vector<uint32_t> order;
    for (uint32_t v = 0; v < s.n_vars; ++v) if (uses[v] >= 2) order.push_back(v);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return uses[x] > uses[y]; });
    if (order.size() > zkdev::FLAG_PLANES) order.resize(zkdev::FLAG_PLANES);
    for (uint32_t k = 0; k < order.size(); ++k) plane_of[order[k]] = k;
    return plane_of;
}

void CS::schedule_by_locality(const std::vector<double>& a, const std::vector<double>& m, double a_tot, double m_tot,
                              const std::vector<std::vector<uint32_t>>& succ, std::vector<uint32_t>& n_pred) {
    Scope& s = loop_;
    const size_t n = s.ops.size();
    const int64_t W = 16;   // touches (reads + writes) a value stays "near" for (flat optimum 12..20 of the model and of the kernel, see above)
    std::vector<int64_t> last(s.n_vars, INT64_MIN / 2);
    int64_t stamp = 0;
    std::vector<uint32_t> ready_light, ready_heavy, order;
    std::vector<uint8_t> in_ready(n, 0);
    const bool balance = true;   // heavy ops (permutations, inversions) spread by resource balance
    auto make_ready = [&](uint32_t i) { (balance && a[i] > 1000 ? ready_heavy : ready_light).push_back(i); };
    for (size_t i = 0; i < n; ++i) if (n_pred[i] == 0) make_ready((uint32_t)i);
    double a_done = 0, m_done = 0;
    auto imbalance_after = [&](uint32_t i) { return std::fabs((a_done + a[i]) / a_tot - (m_done + m[i]) / m_tot); };
    uint32_t prev_opcode = UINT32_MAX, run = 0;
    auto score = [&](uint32_t i) {
        const OpRec& op = s.ops[i];
        double sc = 0; int vars = 0;
        for (auto& in : op.ins) if (in.kind == Operand::VAR) { ++vars; sc += (stamp - last[in.idx] <= W) ? 1.0 : -1.0; }
        if (!vars) sc = 0.25;   // constants / inputs: neutral, slightly ahead of a miss
        if ((uint32_t)op.opcode == prev_opcode && run < group_cap(op, true)) sc += 0.6;
        return sc;
    };
    while (order.size() < n) {
        // light candidate: best score, ties to the lowest recording index
        size_t bl = SIZE_MAX; double bs = -1e300;
        const size_t scan0 = ready_light.size() > 4096 ? ready_light.size() - 4096 : 0;   // bound the scan: the newest ready ops are the local ones
        for (size_t k = scan0; k < ready_light.size(); ++k) {
            const double sc = score(ready_light[k]);
            if (sc > bs + 1e-12 || (std::fabs(sc - bs) <= 1e-12 && ready_light[k] < ready_light[bl])) { bs = sc; bl = k; }
        }
        uint32_t best = bl == SIZE_MAX ? UINT32_MAX : ready_light[bl];
        double best_v = best == UINT32_MAX ? 1e300 : imbalance_after(best);
        size_t best_h = SIZE_MAX;
        for (size_t hi = 0; hi < ready_heavy.size(); ++hi) {
            const uint32_t h = ready_heavy[hi];
            const double v = imbalance_after(h);
            if (v < best_v - 1e-12 || (std::fabs(v - best_v) <= 1e-12 && h < best)) { best = h; best_v = v; best_h = hi; }
        }
        if (best == UINT32_MAX) return;
        if (best_h != SIZE_MAX) { ready_heavy[best_h] = ready_heavy.back(); ready_heavy.pop_back(); }
        else { ready_light[bl] = ready_light.back(); ready_light.pop_back(); }
        order.push_back(best);
        const OpRec& op = s.ops[best];
        if ((uint32_t)op.opcode == prev_opcode) ++run; else { prev_opcode = op.opcode; run = 1; }
        for (auto& in : op.ins) if (in.kind == Operand::VAR) last[in.idx] = ++stamp;
        if (op.opcode == ZK_OP_P2_ROUNDS) { stamp += (int64_t)op.outs.size(); for (size_t q = op.outs.size() - 12; q < op.outs.size(); ++q) last[op.outs[q]] = stamp; }
        else for (auto ov : op.outs) last[ov] = ++stamp;
        a_done += a[best]; m_done += m[best];
        for (auto nx : succ[best]) if (--n_pred[nx] == 0) make_ready(nx);
    }
    std::vector<OpRec> reordered;
    reordered.reserve(n);
    for (auto i : order) reordered.push_back(std::move(s.ops[i]));
    s.ops = std::move(reordered);
}

// ------------------------------------------------------------------ program emission
// header, operand words, destination lists of one op
void CS::emit_op(const Scope& s, const OpRec& op, std::vector<uint32_t>& out) const {
    out.push_back((uint32_t)op.opcode | ((uint32_t)op.a << 8) | ((uint32_t)op.b << 16));
    for (auto& in : op.ins) {
        switch (in.kind) {
        case Operand::VAR: out.push_back(home(s, in.idx)); break;
        case Operand::CONSTPOOL: out.push_back(ZK_OPERAND_CONST | in.idx); break;
        case Operand::OUTER_VAR: out.push_back(ZK_OPERAND_OUTER | home(outer_, in.idx)); break;
        case Operand::RAW:
            if (op.opcode == ZK_OP_LOOP_LAST) out.push_back(home(loop_, in.idx));
            else out.push_back(in.idx);
            break;
        }
    }
    emit_dests(s, op, out);
}
void CS::emit_dests(const Scope& s, const OpRec& op, std::vector<uint32_t>& out) const {
    for (uint32_t ov : op.outs) {
        const auto& cells = s.var_cells[ov];
        if (!emit_full_) { out.push_back(s.var_slot[ov]); continue; }  // device program: the variable's store slot
        for (size_t i = 0; i < cells.size(); ++i) out.push_back(cells[i] | (i + 1 < cells.size() ? ZK_DEST_MORE : 0));
    }
}

// Strand form of a scope's program (kernels_engine.hpp k_witness_strands).  Per phase: level(op) = 1 + the deepest producer of
// its operands inside the phase; the ops of a level are independent of each other and are dealt out over the 8 strands
// (heaviest first, always to the lightest strand); every strand ends the level with ZK_OP_BARRIER.  Values cross strands only
// through the cells, between levels.
void CS::build_strands(Scope& s, uint32_t NS, bool narrow) {
    // narrow: the second strand form of a loop scope (NARROW_STRANDS per tile) kept beside the full one; launch_phase picks by tile count
    constexpr uint32_t NS_MAX = zkdev::STRANDS_PER_TILE;
    if (NS == 0 || NS > NS_MAX) throw ZkError(ZK_ERR_INVALID, "internal: strand count");
    std::vector<uint32_t>& sprog = narrow ? s.sprog_n : s.sprog;
    auto& s_begin = narrow ? s.sn_begin : s.s_begin;
    auto& s_end = narrow ? s.sn_end : s.s_end;
    auto& s_levels = narrow ? s.sn_levels : s.s_levels;
    auto& s_gain = narrow ? s.sn_gain : s.s_gain;
    sprog.clear();
    for (int ph0 = 0; ph0 < 3; ++ph0) for (uint32_t k = 0; k < NS_MAX; ++k) s_begin[ph0][k] = s_end[ph0][k] = 0;
    const size_t n_ops = s.ops.size();
    size_t bounds[4] = {0, n_ops, n_ops, n_ops};
    if (!s.is_loop) { bounds[1] = std::min(s.pre_ops, n_ops); bounds[2] = std::min(std::max(s.side_ops, bounds[1]), n_ops); }
    std::vector<int64_t> producer(s.n_vars, -1);
    std::vector<uint32_t> level(n_ops, 0);
    std::vector<std::vector<uint32_t>> strand(NS);
    for (int ph = 0; ph < 3; ++ph) {
        const size_t o0 = bounds[ph], o1 = bounds[ph + 1];
        // two-tier levels (see build_seed_program): tier = heavy ops (Poseidon2 permutations) on the longest producer
        // path; light ops levelled as soon as possible inside the tier, the tier's heavy ops share its last level
        uint32_t n_levels = 0;
        {
            auto heavy = [&](const OpRec& op) { return op.opcode == ZK_OP_P2_ROUNDS || op.opcode == ZK_OP_POSEIDON2; };
            std::vector<uint32_t> tier(o1 - o0, 0), local(o1 - o0, 0);
            uint32_t n_tiers = 0;
            for (size_t oi = o0; oi < o1; ++oi) {
                const OpRec& op = s.ops[oi];
                if (op.seed_only) continue;
                uint32_t t = 0, l = 0;
                for (auto& in : op.ins)
                    if (in.kind == Operand::VAR && producer[in.idx] >= (int64_t)o0) {
                        const size_t p = (size_t)producer[in.idx];
                        t = std::max(t, tier[p - o0] + (heavy(s.ops[p]) ? 1u : 0u));
                    }
                if (!heavy(op))
                    for (auto& in : op.ins)
                        if (in.kind == Operand::VAR && producer[in.idx] >= (int64_t)o0) {
                            const size_t p = (size_t)producer[in.idx];
                            if (tier[p - o0] == t && !heavy(s.ops[p])) l = std::max(l, local[p - o0] + 1);
                        }
                tier[oi - o0] = t; local[oi - o0] = l;
                n_tiers = std::max(n_tiers, t + 1);
                for (uint32_t ov : op.outs) producer[ov] = (int64_t)oi;
            }
            std::vector<uint32_t> light_levels(n_tiers, 0), has_heavy(n_tiers, 0), base(n_tiers + 1, 0);
            for (size_t oi = o0; oi < o1; ++oi) {
                if (s.ops[oi].seed_only) continue;
                if (heavy(s.ops[oi])) has_heavy[tier[oi - o0]] = 1;
                else light_levels[tier[oi - o0]] = std::max(light_levels[tier[oi - o0]], local[oi - o0] + 1);
            }
            for (uint32_t t = 0; t < n_tiers; ++t) base[t + 1] = base[t] + light_levels[t] + has_heavy[t];
            n_levels = base[n_tiers];
            for (size_t oi = o0; oi < o1; ++oi)
                if (!s.ops[oi].seed_only)
                    level[oi] = base[tier[oi - o0]] + (heavy(s.ops[oi]) ? light_levels[tier[oi - o0]] : local[oi - o0]);
        }
        std::vector<std::vector<uint32_t>> by_level(n_levels);
        for (size_t oi = o0; oi < o1; ++oi)
            if (!s.ops[oi].seed_only) by_level[level[oi]].push_back((uint32_t)oi);
        auto cost = [&](uint32_t oi) -> uint64_t {
            const OpRec& op = s.ops[oi];
            uint64_t c = 8 + op.ins.size();
            c += 2 * op.outs.size();
            if (op.opcode == ZK_OP_P2_ROUNDS || op.opcode == ZK_OP_POSEIDON2) c += 4000;
            if (op.opcode == ZK_OP_NN_MULMOD) c += 2000;
            if (op.opcode == ZK_OP_U256_DIVREM) c += 2500;
            if (op.opcode == ZK_OP_KECCAK_F || op.opcode == ZK_OP_SHA256_ROUNDS || op.opcode == ZK_OP_BYTEBUF_FILL) c += 8000;   // + 2 per output above
            return c;
        };
        for (auto& st : strand) st.clear();
        uint64_t total = 0, critical = 0;
        // (strand programs keep SELECT flags in store slots: a strand form of the bit planes existed in round 5, unmeasured, and was deleted in round 6)
        for (uint32_t lv = 0; lv < n_levels; ++lv) {
            auto& ops = by_level[lv];
            std::stable_sort(ops.begin(), ops.end(), [&](uint32_t a, uint32_t b) { return cost(a) > cost(b); });
            uint64_t load[NS_MAX] = {0};
            std::vector<uint32_t> mine[NS_MAX];
            for (uint32_t oi : ops) {
                if ((s.ops[oi].opcode == ZK_OP_KECCAK_F || s.ops[oi].opcode == ZK_OP_SHA256_ROUNDS || s.ops[oi].opcode == ZK_OP_BYTEBUF_FILL) && (NS & (NS - 1)) == 0) {
                    // cooperative macro-op: every strand runs it and stores its share of the outputs (kernels_engine2.hpp keccak_f_stream)
                    for (uint32_t k = 0; k < NS; ++k) { load[k] += cost(oi) / NS + 3000; mine[k].push_back(oi); }
                    continue;
                }
                uint32_t best = 0;
                for (uint32_t k = 1; k < NS; ++k) if (load[k] < load[best]) best = k;
                load[best] += cost(oi);
                mine[best].push_back(oi);
            }
            // The ops of a level are independent, so a strand's ops of one kind go out in groups under one header (scalar-decoded
            // strand form, kernels_engine2.hpp k_witness_strands2): header, operand words of every member as in the plain v2
            // form, then ONE destination word per member — the store slot of its first output (an op's outputs are consecutive
            // slots).  Caps: a group fits the 16-word fetch (SELECT 3, FMA 2, INPUT 7, LOOKUP of <= 2 keys 3, U32MULADD 2).
            const bool grouping = true;
            for (uint32_t k = 0; k < NS; ++k) {
                std::map<uint64_t, std::vector<size_t>> groups;  // kind key -> ops of this strand and level
                std::vector<uint64_t> order;
                for (uint32_t oi : mine[k]) {
                    const OpRec& op = s.ops[oi];
                    uint64_t key = ((uint64_t)1 << 63) | oi;  // ungrouped: its own key
                    if (grouping) {
                        if (op.opcode == ZK_OP_LOOKUP && op.a <= 2 && op.b <= 2) key = ((uint64_t)ZK_OP_LOOKUP << 56) | ((uint64_t)op.ins[0].idx << 24) | ((uint64_t)op.a << 8) | op.b;
                        else if (op.opcode == ZK_OP_SELECT || op.opcode == ZK_OP_FMA || op.opcode == ZK_OP_INPUT || op.opcode == ZK_OP_U32MULADD) key = (uint64_t)op.opcode << 56;
                    }
                    if (!groups.count(key)) order.push_back(key);
                    groups[key].push_back(oi);
                }
                for (uint64_t key : order) {
                    const auto& g = groups[key];
                    const OpRec& f0 = s.ops[g[0]];
                    size_t cap = 1;
                    bool counted = false;
                    if (!(key >> 63)) {
                        counted = true;
                        cap = f0.opcode == ZK_OP_SELECT ? 3 : f0.opcode == ZK_OP_FMA ? 2 : f0.opcode == ZK_OP_INPUT ? 7 : f0.opcode == ZK_OP_U32MULADD ? 2 : 3;
                    } else counted = f0.opcode == ZK_OP_SELECT || f0.opcode == ZK_OP_FMA || f0.opcode == ZK_OP_INPUT || f0.opcode == ZK_OP_U32MULADD || f0.opcode == ZK_OP_LC4;
                    for (size_t i0 = 0; i0 < g.size(); i0 += cap) {
                        std::vector<size_t> part(g.begin() + i0, g.begin() + std::min(g.size(), i0 + cap));
                        emit_group_v2(s, part, counted, strand[k]);
                        for (size_t oi : part) {
                            const OpRec& op = s.ops[oi];
                            for (size_t q = 0; q < op.outs.size(); ++q)
                                if (s.var_slot[op.outs[q]] != s.var_slot[op.outs[0]] + q) throw ZkError(ZK_ERR_INVALID, "internal: an op's outputs are not consecutive store slots");
                            strand[k].push_back(op.outs.empty() ? 0u : s.var_slot[op.outs[0]]);
                        }
                    }
                }
            }
            for (uint32_t k = 0; k < NS; ++k) total += load[k];
            critical += *std::max_element(load, load + NS) + 200;  // + the barrier: every strand drains its stores
            if (lv + 1 < n_levels) for (auto& st : strand) st.push_back(ZK_OP_BARRIER);
        }
        plane_of_ = nullptr;
        s_gain[ph] = critical ? (float)total / (float)critical : 0.f;
        if (getenv("ZKGL_PROG_STATS") && o1 > o0) {
            uint32_t narrow = 0, wide = 0;
            for (auto& l : by_level) { narrow += l.size() < 8; wide += l.size() >= 64; }
            fprintf(stderr, "[zkgl] strands %s phase %d: %zu ops, %u levels (%u with fewer than 8 ops, %u with 64 or more), estimated gain %.2f\n",
                    s.is_loop ? "loop" : "outer", ph, o1 - o0, n_levels, narrow, wide, s_gain[ph]);
        }
        for (uint32_t k = 0; k < NS; ++k) {
            s_begin[ph][k] = (uint32_t)sprog.size();
            sprog.insert(sprog.end(), strand[k].begin(), strand[k].end());
            s_end[ph][k] = (uint32_t)sprog.size();
        }
        s_levels[ph] = n_levels;
    }
}

// ZKGL_VERIFY_DEVICE_PROGRAMS=1 (CPU tests): an independent walk over the scalar-decoded device programs of a scope — the plain form
// (prog2) and the strand forms — with the op layouts as the KERNELS decode them (kernels_engine2.hpp run_tile2: words per op, members per
// header, destination words of the strand form, the plane / chain forms of SELECT, ZK_OP_FLAG_PLANES), checked against the recorded
// ops: every op once (a cooperative macro-op once per strand), its operand words, its output slots, consecutive slots in the plain
// form, operands produced in an earlier level in the strand form, every plane written before it is read (plain: earlier in the
// program; strands: in an earlier level, and after the level that produces the flag).  Throws on the first mismatch.
void CS::verify_device_programs(const Scope& s) const {
    auto fail = [&](const std::string& what, size_t pc) { throw ZkError(ZK_ERR_INVALID, "device program check (" + std::string(s.is_loop ? "loop" : "outer") + " scope, word " + std::to_string(pc) + "): " + what); };
    auto operands_per_member = [&](const OpRec& op) -> size_t {   // what the kernel reads per member, from ITS case (not from op.ins)
        switch (op.opcode) {
        case ZK_OP_CONST: case ZK_OP_INPUT: case ZK_OP_ISZERO: case ZK_OP_SPLIT: case ZK_OP_LOOP_LAST: case ZK_OP_DIVREM: return 1;
        case ZK_OP_FMA: return 5; case ZK_OP_LC4: case ZK_OP_DOT4: return 8; case ZK_OP_SELECT: case ZK_OP_UADD: case ZK_OP_USUB: return 3;
        case ZK_OP_MATMUL12: case ZK_OP_P2_ROUNDS: return 12; case ZK_OP_POSEIDON2: return op.a ? 13 : 12;
        case ZK_OP_LOOKUP: return op.a; case ZK_OP_U32MULADD: return 4; case ZK_OP_U8X4FMA: case ZK_OP_U256_MULWIDE: case ZK_OP_U256_DIVREM: return 16;
        case ZK_OP_SHA256_ROUNDS: return 96; case ZK_OP_KECCAK_F: return 200; case ZK_OP_BYTEBUF_FILL: return zkb::N_INPUTS; case ZK_OP_NN_MULMOD: return 50;
        default: return SIZE_MAX;
        }
    };
    uint32_t max_slot = 0;
    for (uint32_t v = 0; v < s.n_vars; ++v) if (s.var_slot[v] != UINT32_MAX) max_slot = std::max(max_slot, s.var_slot[v]);
    std::vector<int64_t> producer(s.n_vars, -1), op_by_slot((size_t)max_slot + 2, -1), var_by_slot((size_t)max_slot + 2, -1);
    for (size_t oi = 0; oi < s.ops.size(); ++oi) {
        if (s.ops[oi].seed_only) continue;
        for (uint32_t ov : s.ops[oi].outs) { producer[ov] = (int64_t)oi; if (s.var_slot[ov] < var_by_slot.size()) var_by_slot[s.var_slot[ov]] = ov; }
        if (!s.ops[oi].outs.empty() && s.var_slot[s.ops[oi].outs[0]] < op_by_slot.size()) op_by_slot[s.var_slot[s.ops[oi].outs[0]]] = (int64_t)oi;
    }
    const std::vector<uint32_t> plane_all = select_plane_vars(s);
    // one member of one header against its op; returns nothing, throws
    auto check_member = [&](const std::vector<uint32_t>& prog, size_t at, const OpRec& op, size_t pc) {
        std::vector<uint32_t> want;
        if (op.opcode == ZK_OP_NN_MULMOD) {
            for (size_t q = 0; q < 16; ++q) operand_v2(s, op, q, want);
            for (size_t q = 0; q < 17; ++q) { if (q < op.a) operand_v2(s, op, 16 + q, want); else want.push_back(0); }
            for (size_t q = 0; q < 17; ++q) { if (q < op.b) operand_v2(s, op, 16 + op.a + q, want); else want.push_back(0); }
        } else
            for (size_t q = (op.opcode == ZK_OP_LOOKUP ? 1 : 0); q < op.ins.size(); ++q) operand_v2(s, op, q, want);
        if (want.size() != operands_per_member(op)) fail("opcode " + std::to_string(op.opcode) + ": the kernel reads " + std::to_string(operands_per_member(op)) + " operand words, the op has " + std::to_string(want.size()), pc);
        for (size_t k = 0; k < want.size(); ++k)
            if (at + k >= prog.size() || prog[at + k] != want[k]) fail("opcode " + std::to_string(op.opcode) + ": operand word " + std::to_string(k) + " differs", pc);
    };
    // decode one header at pc of `prog`; D = destination words per member (strand form).  Calls on_op(op index, first output slot) per member,
    // on_plane_write(slot, id), on_plane_read(id, op index).  Returns the next pc, or SIZE_MAX at a barrier.
    struct Hooks { std::function<void(size_t, uint32_t, size_t)> on_op; std::function<void(uint32_t, uint32_t, size_t)> plane_write; std::function<void(uint32_t, size_t, size_t)> plane_read;
                   std::function<int64_t(size_t, uint32_t)> expect; };
    auto step = [&](const std::vector<uint32_t>& prog, size_t pc, uint32_t D, const Hooks& hk, bool& barrier) -> size_t {
        barrier = false;
        if (pc >= prog.size()) fail("ran past the end", pc);
        const uint32_t h = prog[pc], opc = h & 0xff, pa = (h >> 8) & 0xff, pb = h >> 16;
        if (opc == ZK_OP_BARRIER) { barrier = true; return pc + 1; }
        if (opc == ZK_OP_FLAG_PLANES) {
            const uint32_t n = pb + 1;
            if (n > 7) fail("FLAG_PLANES with more than 7 flags", pc);
            for (uint32_t k = 0; k < n; ++k) hk.plane_write(prog[pc + 1 + 2 * k], prog[pc + 2 + 2 * k], pc);
            return pc + 1 + 2 * n;
        }
        if (opc == ZK_OP_SELECT && pa == 2) fail("SELECT header with a = 2 (the mux-chain form was deleted in round 6: no kernel decodes it)", pc);
        // members of the header, as the kernel derives them
        uint32_t N = 1;
        bool grouped_lookup = false;
        if (opc == ZK_OP_INPUT || opc == ZK_OP_SELECT || opc == ZK_OP_FMA || opc == ZK_OP_U32MULADD) N = pb + 1;
        if (opc == ZK_OP_POSEIDON2 && pa == 2) fail("POSEIDON2 header with a = 2 (the merged gated form was deleted in round 6: no kernel decodes it)", pc);
        if (opc == ZK_OP_LOOKUP) { const uint32_t nv = pb & 0xff; grouped_lookup = pa <= 2 && nv <= 2; N = grouped_lookup ? (pb >> 8) + 1 : 1; if (!grouped_lookup && (pb >> 8)) fail("wide lookup with members", pc); }
        const size_t first_operand = pc + 1 + (opc == ZK_OP_LOOKUP ? 1 : 0);
        size_t K = SIZE_MAX, at = first_operand;
        std::vector<size_t> members;
        for (uint32_t g = 0; g < N; ++g) {
            // which op is this member?  plain form: the next op in order; strand form: the op whose first output slot is the destination word
            int64_t oi;
            if (D) {
                // the member's operand count is needed to find its destination word: take it from the first member's op kind (same kind in a group)
                if (K == SIZE_MAX) {
                    // find K by trying the op at the destination position for each plausible K is circular: use the table by opcode / header
                    OpRec probe; probe.opcode = (uint8_t)opc; probe.a = (uint8_t)pa; probe.b = (uint8_t)pb;
                    if (opc == ZK_OP_NN_MULMOD) K = 50; else K = operands_per_member(probe);
                    if (opc == ZK_OP_SELECT && pa == 1) K = 3;
                }
                const size_t dpos = first_operand + (size_t)N * K + g;
                if (dpos >= prog.size()) fail("destination word past the end", pc);
                oi = hk.expect(pc, prog[dpos]);
            } else oi = hk.expect(pc, 0);
            if (oi < 0) fail("no op for a member of opcode " + std::to_string(opc), pc);
            const OpRec& op = s.ops[(size_t)oi];
            if (op.opcode != opc) fail("opcode " + std::to_string(opc) + " in the program, op " + std::to_string(op.opcode) + " recorded", pc);
            if (K == SIZE_MAX) K = operands_per_member(op);
            if (K == SIZE_MAX) fail("opcode without a kernel layout", pc);
            if (opc == ZK_OP_LOOKUP && prog[pc + 1] != op.ins[0].idx) fail("lookup table id", pc);
            const bool counted = opc == ZK_OP_INPUT || opc == ZK_OP_SELECT || opc == ZK_OP_FMA || opc == ZK_OP_U32MULADD || opc == ZK_OP_LC4 || opc == ZK_OP_LOOKUP;
            if (!(opc == ZK_OP_SELECT && pa == 1) && pa != op.a) fail("header a", pc);
            if (!counted && pb != op.b) fail("header b", pc);
            if (opc == ZK_OP_LOOKUP && (pb & 0xff) != op.b) fail("lookup n_vals", pc);
            if (opc == ZK_OP_SELECT && pa == 1) {   // plane form: [plane id, a, b]
                if (plane_all[op.ins[0].idx] == UINT32_MAX || prog[at] != plane_all[op.ins[0].idx]) fail("plane SELECT: plane id", pc);
                if (prog[at + 1] != s.var_slot[op.ins[1].idx] || prog[at + 2] != s.var_slot[op.ins[2].idx]) fail("plane SELECT: a / b slot", pc);
                hk.plane_read(prog[at], (size_t)oi, pc);
            } else check_member(prog, at, op, pc);
            at += K;
            members.push_back((size_t)oi);
        }
        for (uint32_t g = 0; g < N; ++g) {
            const OpRec& op = s.ops[members[g]];
            const uint32_t first_slot = op.outs.empty() ? 0u : s.var_slot[op.outs[0]];
            if (D && prog[at + g] != first_slot) fail("destination word", pc);
            for (size_t q = 0; q < op.outs.size(); ++q) if (s.var_slot[op.outs[q]] != first_slot + q) fail("an op's outputs are not consecutive slots", pc);
            hk.on_op(members[g], first_slot, pc);
        }
        return at + (size_t)D * N;
    };
    // ---- plain form: every op in order, consecutive output slots, planes written before they are read
    {
        std::vector<uint32_t> prog(s.prog2);
        if (const char* sab = getenv("ZKGL_VERIFY_SABOTAGE")) {   // tests: the check must notice a changed word (loop scope's plain form)
            const size_t at = (size_t)atoll(sab);
            if (s.is_loop && at < prog.size()) prog[at] ^= 1u;
        }
        prog.resize(prog.size() + 16, 0);
        size_t cursor = 0, pc = 0;
        uint32_t next_slot = 0;
        std::vector<uint8_t> written(zkdev::FLAG_PLANES, 0);
        auto next_op = [&]() -> int64_t { while (cursor < s.ops.size() && s.ops[cursor].seed_only) ++cursor; return cursor < s.ops.size() ? (int64_t)cursor++ : -1; };
        Hooks hk;
        hk.expect = [&](size_t, uint32_t) { return next_op(); };
        hk.on_op = [&](size_t oi, uint32_t first_slot, size_t at) {
            if (!s.ops[oi].outs.empty() && first_slot != next_slot) fail("plain form: outputs are not the next consecutive slots", at);
            next_slot += (uint32_t)s.ops[oi].outs.size();
        };
        hk.plane_write = [&](uint32_t slot, uint32_t id, size_t at) {
            if (id >= zkdev::FLAG_PLANES || slot >= next_slot || var_by_slot[slot] < 0 || plane_all[(size_t)var_by_slot[slot]] != id) fail("FLAG_PLANES: slot / id do not name a produced flag variable", at);
            written[id] = 1;
        };
        hk.plane_read = [&](uint32_t id, size_t, size_t at) { if (id >= zkdev::FLAG_PLANES || !written[id]) fail("a SELECT reads a plane nothing has written", at); };
        while (pc < s.prog2.size()) { bool bar; pc = step(prog, pc, 0, hk, bar); if (bar) fail("barrier in the plain form", pc); }
        if (next_op() != -1) fail("plain form ends before the last op", pc);
    }
    // ---- strand forms
    for (int form = 0; form < 2; ++form) {
        const std::vector<uint32_t>& sp = form ? s.sprog_n : s.sprog;
        if (sp.empty()) continue;
        const auto& sb = form ? s.sn_begin : s.s_begin;
        const auto& se = form ? s.sn_end : s.s_end;
        const uint32_t NS = form ? NARROW_STRANDS : zkdev::STRANDS_PER_TILE;
        std::vector<uint32_t> prog(sp);
        prog.resize(prog.size() + 16, 0);
        std::vector<int64_t> level_of(s.ops.size(), -1);       // global level (phases in order)
        std::vector<uint32_t> seen(s.ops.size(), 0);
        std::vector<int64_t> plane_level(zkdev::FLAG_PLANES, -1);
        int64_t level_base = 0;
        for (int ph = 0; ph < 3; ++ph) {
            int64_t levels_here = 0;
            std::vector<std::pair<uint32_t, std::pair<size_t, int64_t>>> reads;   // plane id, (op, level)
            for (uint32_t k = 0; k < NS; ++k) {
                size_t pc = sb[ph][k];
                int64_t lv = 0;
                Hooks hk;
                hk.expect = [&](size_t at, uint32_t dest) -> int64_t { if (dest >= op_by_slot.size()) fail("destination slot out of range", at); return op_by_slot[dest]; };
                hk.on_op = [&](size_t oi, uint32_t, size_t at) {
                    const bool coop = s.ops[oi].opcode == ZK_OP_KECCAK_F || s.ops[oi].opcode == ZK_OP_SHA256_ROUNDS || s.ops[oi].opcode == ZK_OP_BYTEBUF_FILL;
                    if (level_of[oi] >= 0 && !(coop && level_of[oi] == level_base + lv)) fail("an op appears twice in the strand programs", at);
                    level_of[oi] = level_base + lv; ++seen[oi];
                };
                hk.plane_write = [&](uint32_t slot, uint32_t id, size_t at) {
                    if (id >= zkdev::FLAG_PLANES || slot >= var_by_slot.size() || var_by_slot[slot] < 0 || plane_all[(size_t)var_by_slot[slot]] != id) fail("FLAG_PLANES: slot / id do not name a flag variable", at);
                    if (plane_level[id] >= 0) fail("a plane is written twice", at);
                    plane_level[id] = level_base + lv;
                };
                hk.plane_read = [&](uint32_t id, size_t oi, size_t) { reads.push_back({id, {oi, level_base + lv}}); };
                while (pc < se[ph][k]) { bool bar; pc = step(prog, pc, 1, hk, bar); if (bar) ++lv; }
                if (pc != se[ph][k]) fail("a strand's last op runs past its end", pc);
                levels_here = std::max(levels_here, lv + 1);
            }
            for (auto& r : reads)
                if (r.first >= zkdev::FLAG_PLANES || plane_level[r.first] < 0 || !(plane_level[r.first] < r.second.second)) fail("strand form: a SELECT reads a plane that no EARLIER level has written", 0);
            level_base += levels_here;
        }
        for (size_t oi = 0; oi < s.ops.size(); ++oi) {
            const OpRec& op = s.ops[oi];
            if (op.seed_only) continue;
            const bool coop = (op.opcode == ZK_OP_KECCAK_F || op.opcode == ZK_OP_SHA256_ROUNDS || op.opcode == ZK_OP_BYTEBUF_FILL) && (NS & (NS - 1)) == 0;
            if (seen[oi] != (coop ? NS : 1u)) fail("op " + std::to_string(oi) + " (opcode " + std::to_string(op.opcode) + ") appears " + std::to_string(seen[oi]) + " times in the strand programs", 0);
            for (auto& in : op.ins)
                if (in.kind == Operand::VAR && producer[in.idx] >= 0 && !(level_of[(size_t)producer[in.idx]] < level_of[oi])) fail("strand form: an operand is produced in the same or a later level", 0);
        }
        for (uint32_t id = 0; id < zkdev::FLAG_PLANES; ++id) {
            if (plane_level[id] < 0) continue;
            for (uint32_t v = 0; v < s.n_vars; ++v)
                if (plane_all[v] == id && producer[v] >= 0 && !(level_of[(size_t)producer[v]] < plane_level[id])) fail("strand form: a plane is copied in the level that produces its flag", 0);
        }
    }
}

// phase 0 = loop body / outer pre, 1 = outer side, 2 = outer post
bool CS::loop_runs_strands(const Scope& s, int phase, uint32_t n_lanes) const {
    const char* e = getenv("ZKGL_STRANDS");
    const int mode = e ? atoi(e) : -1;
    const uint32_t waves = (n_lanes + 63) / 64;
    const bool have = uploaded_ ? s.d_sprog != nullptr : !s.sprog.empty();
    return have && mode != 0 && (mode == 1 || (waves <= 4 * (uint32_t)device_cu_count() && s.s_gain[phase] >= (waves <= (uint32_t)device_cu_count() / 4 ? 1.5f : 3.2f)));
}

void CS::launch_phase(const Scope& s, zkdev::ScopeArgs a, int phase, void* stream, uint32_t n_lanes) const {
    a.xmacros = (uses_sha4_macro_ ? 1u : 0u) | (uses_bytebuf_macro_ ? 2u : 0u);   // kernels_engine2.hpp X_SHA4 / X_BYTEBUF: the kernels that carry those macro-op backends
    const char* e = getenv("ZKGL_STRANDS");  // 0 off, 1 always, unset: by size and estimated gain
    const int mode = e ? atoi(e) : -1;
    const uint32_t waves = ((n_lanes ? n_lanes : s.n_lanes) + 63) / 64;
    // worth it when the scope is short of wavefronts AND its op graph is wide (hash circuits); chains of Poseidon2
    // permutations (queue circuits, the commitments of every outer scope) only pay for the barriers.  A scope of a few
    // wavefronts (outer scopes) has the chip to itself and takes any gain; one of hundreds needs a clear one (measured:
    // log_sorter's loop body at an estimated 2.8 runs 1.4x slower in strand form, keccak's at 3.7 runs 1.9x faster).
    const bool strands = s.d_sprog && mode != 0 && (mode == 1 || (waves <= 4 * (uint32_t)device_cu_count() && s.s_gain[phase] >= (waves <= (uint32_t)device_cu_count() / 4 ? 1.5f : 3.2f)));   // short of wavefronts: fewer than one per SIMD
    if (!strands) {
        const uint32_t end = (uint32_t)s.prog2.size();
        uint32_t w0 = 0, w1 = end, slot0 = 0;
        if (!s.is_loop) {
            if (phase == 0) w1 = s.pre_words2;
            else if (phase == 1) { w0 = s.pre_words2; w1 = s.side_words2; slot0 = s.pre_slots; }
            else { w0 = s.side_words2; slot0 = s.side_slots; }
        }
        a.prog = s.d_prog2; a.n_words = end;
        if (a.cls) { a.prog = s.d_prog2n; a.cls = s.d_prog2n + s.cls_off; }   // narrow store (resolve_and_check): same words, address-word operands, class words behind
        dev_check(zkdev::launch_witness(a, w0, w1, slot0, stream));
        return;
    }
    if (a.cls) throw ZkError(ZK_ERR_INVALID, "internal: the narrow store was bound for a batch whose loop launch takes the strand form (ZKGL_STRANDS changed after zk_cs_set_batch?)");
    // Strands per tile: 16 wavefronts of a tile share its levels' work, but only two such workgroups fit a CU; a loop scope with more
    // tiles than that (keccak FSM at 128 instances: 672 tiles on 256 CUs) runs them in rounds.  The narrow form (8 strands: four
    // workgroups per CU) keeps every tile resident: keccak FSM 23.7 -> 20.8 ms; eip_4844 (120 tiles) is 1.24 x slower with it and
    // keeps 16 (profiles/r3_strands_ab.txt).  ZKGL_STRANDS_NARROW=0 / 1 forces the choice.
    // (with cooperative macro-ops in the program the 16-strand form wins again: sixteen wavefronts share a macro-op's stores — keccak FSM
    // 20.6 ms narrow, 19.0 ms wide, round 4)
    bool narrow = s.is_loop && s.d_sprog_n && phase == 0 && waves > 2 * (uint32_t)device_cu_count() && !uses_lookup_macros_;   // two 16-strand workgroups fit a CU
    if (const char* ne = getenv("ZKGL_STRANDS_NARROW")) narrow = s.is_loop && s.d_sprog_n && phase == 0 && ne[0] == '1';
    if (narrow) {
        a.prog = s.d_sprog_n; a.n_words = (uint32_t)s.sprog_n.size();
        dev_check(zkdev::launch_witness_strands(a, s.sn_begin[phase], s.sn_end[phase], stream, NARROW_STRANDS));
        return;
    }
    a.prog = s.d_sprog; a.n_words = (uint32_t)s.sprog.size();
    dev_check(zkdev::launch_witness_strands(a, s.s_begin[phase], s.s_end[phase], stream, zkdev::STRANDS_PER_TILE));
}

// one operand word of the scalar-decoded device forms (kernels_engine2.hpp): data operands are bare store slots, FMA / LC4 /
// NN_MULMOD immediates bare pool indices, ZK_OP_CONST keeps a kind (pool constant or outer value), raw words as recorded
void CS::operand_v2(const Scope& s, const OpRec& op, size_t pos, std::vector<uint32_t>& out) const {
    const Operand& in = op.ins[pos];
    const bool coeff = (op.opcode == ZK_OP_FMA && pos < 2) || (op.opcode == ZK_OP_LC4 && pos < 4) || (op.opcode == ZK_OP_NN_MULMOD && pos < 16);
    if (coeff) {
        if (in.kind != Operand::CONSTPOOL) throw ZkError(ZK_ERR_INVALID, "internal: FMA / LC4 / NN_MULMOD immediate is not a pool constant");
        out.push_back(in.idx);
    } else if (op.opcode == ZK_OP_CONST) {
        if (in.kind == Operand::CONSTPOOL) out.push_back(ZK_OPERAND_CONST | in.idx);
        else if (in.kind == Operand::OUTER_VAR) out.push_back(ZK_OPERAND_OUTER | outer_.var_slot[in.idx]);
        else throw ZkError(ZK_ERR_INVALID, "internal: ZK_OP_CONST of a variable");
    } else if (in.kind == Operand::VAR) out.push_back(narrow_emit_ ? (*narrow_emit_)[s.var_slot[in.idx]] : s.var_slot[in.idx]);   // narrow form: the value's address word
    else if (in.kind == Operand::RAW) out.push_back(op.opcode == ZK_OP_LOOP_LAST ? loop_.var_slot[in.idx] : in.idx);
    else throw ZkError(ZK_ERR_INVALID, "internal: pool constant / outer value in a data operand position");
}
// header + operand words of a group of same-kind ops in the scalar-decoded forms (no destinations)
void CS::emit_group_v2(const Scope& s, const std::vector<size_t>& group, bool counted, std::vector<uint32_t>& out) const {
    const OpRec& first = s.ops[group[0]];
    const size_t n = group.size();
    if (first.opcode == ZK_OP_LOOKUP) {
        out.push_back((uint32_t)ZK_OP_LOOKUP | ((uint32_t)first.a << 8) | (((uint32_t)first.b | ((uint32_t)(n - 1) << 8)) << 16));
        out.push_back(first.ins[0].idx);  // table id
        for (size_t oi : group)
            for (size_t q = 1; q < s.ops[oi].ins.size(); ++q) operand_v2(s, s.ops[oi], q, out);
        return;
    }
    if (first.opcode == ZK_OP_SELECT && plane_of_ && (*plane_of_)[first.ins[0].idx] != UINT32_MAX) {   // flags from the bit planes
        out.push_back((uint32_t)ZK_OP_SELECT | (1u << 8) | ((uint32_t)(n - 1) << 16));
        for (size_t oi : group) {
            out.push_back((*plane_of_)[s.ops[oi].ins[0].idx]);
            operand_v2(s, s.ops[oi], 1, out);
            operand_v2(s, s.ops[oi], 2, out);
        }
        return;
    }
    out.push_back((uint32_t)first.opcode | ((uint32_t)first.a << 8) | ((counted ? (uint32_t)(n - 1) : (uint32_t)first.b) << 16));
    if (first.opcode == ZK_OP_NN_MULMOD) {
        // fixed layout: 16 modulus limbs, 17 A slots, 17 B slots (unused ones 0): static word positions for the kernel's scalar fetches
        for (size_t q = 0; q < 16; ++q) operand_v2(s, first, q, out);
        for (size_t q = 0; q < 17; ++q) { if (q < first.a) operand_v2(s, first, 16 + q, out); else out.push_back(0); }
        for (size_t q = 0; q < 17; ++q) { if (q < first.b) operand_v2(s, first, 16 + first.a + q, out); else out.push_back(0); }
        return;
    }
    for (size_t oi : group)
        for (size_t q = 0; q < s.ops[oi].ins.size(); ++q) operand_v2(s, s.ops[oi], q, out);
}

// Device programs group runs of consecutive, mutually independent ops of one kind under ONE header: header b carries
// (members - 1), the operand words of every member follow.  The interpreter issues all operand loads of a group before the
// first use — the one-op-at-a-time form is bound by the latency of each op's dependent loads, not by bandwidth (main_vm:
// 7 073 ops per cycle, 55 % of them SELECTs, most of them recorded as parallel_select over 8 / 12 elements).  Two device forms:
//   v1 (`prog`: strand builder input, k_witness_seq): operands carry a kind, explicit destination words (store slots);
//       caps INPUT 8, SELECT / LOOKUP 4, FMA / LC4 2 (kernels_engine.hpp GS / GF).
//   v2 (`prog2`: the plain kernels, kernels_engine2.hpp): scalar-decoded, data operands are bare store slots, FMA / LC4
//       coefficients bare pool indices, NO destination words (an op's outputs are the next consecutive store slots);
//       caps INPUT 8, SELECT 5, FMA 3, LOOKUP 4, U32MULADD 3 — header + operands of a group fit one 16-word scalar fetch.
// The exported program (oracle) stays ungrouped with every destination cell spelled out.
static uint32_t group_cap(const OpRec& op, bool v2) {
    switch (op.opcode) {
    case ZK_OP_INPUT: return 8;
    case ZK_OP_SELECT: return v2 ? 5 : 4;
    case ZK_OP_FMA: return v2 ? 3 : 2;
    case ZK_OP_LC4: return v2 ? 1 : 2;
    case ZK_OP_U32MULADD: return v2 ? 3 : 1;
    case ZK_OP_LOOKUP: return (op.a <= 2 && op.b <= 2) ? 4 : 1;
    default: return 1;
    }
}

void CS::emit_scope(Scope& s) {
    std::vector<uint8_t> defined(s.n_vars, 0);
    s.prog.clear(); s.prog_full.clear(); s.prog2.clear();
    s.pre_words = 0; s.pre_words_full = 0; s.pre_words2 = 0; s.side_words2 = 0; s.pre_slots = 0; s.side_slots = 0;
    s.cells_written = 0; s.cells_populated = 0;
    const bool grouping = true;
    // ---- the exported program + validation
    for (size_t oi = 0; oi < s.ops.size(); ++oi) {
        if (!s.is_loop && oi == s.pre_ops) s.pre_words_full = (uint32_t)s.prog_full.size();
        const OpRec& op = s.ops[oi];
        if (op.seed_only) continue;
        for (auto& in : op.ins)
            if (in.kind == Operand::VAR && !defined[in.idx]) throw ZkError(ZK_ERR_UNRESOLVED, "witness op reads a variable no earlier op produced");
        emit_full_ = true;
        emit_op(s, op, s.prog_full);
        emit_full_ = false;
        for (uint32_t ov : op.outs) {
            if (defined[ov]) throw ZkError(ZK_ERR_INVALID, "variable produced twice");
            if (s.var_slot[ov] != s.cells_written) throw ZkError(ZK_ERR_INVALID, "internal: store slots are not in production order");
            defined[ov] = 1;
            s.cells_written += 1;
            s.cells_populated += s.var_cells[ov].size();
        }
    }
    if (!s.is_loop && s.pre_ops >= s.ops.size()) s.pre_words_full = (uint32_t)s.prog_full.size();
    // ---- the two device forms
    // form 3 (loop scopes with a narrow layout, build_narrow_layout): the v2 form again with address words in the data operand positions,
    // and one CLASS WORD per header behind the program (bit k: the header's k-th output is a byte-class value) — same ops, same groups
    s.prog2n.clear(); s.cls_off = 0;
    std::vector<uint32_t> cls_words;
    std::vector<uint8_t> slot_is_byte;
    if (s.is_loop && !s.slot_aw.empty()) { slot_is_byte.resize(s.slot_aw.size()); for (size_t q = 0; q < s.slot_aw.size(); ++q) slot_is_byte[q] = (s.slot_aw[q] & zkgeom::AW_BYTE) != 0; }
    auto emit_form = [&](int form) {
        const bool v2 = form >= 2, nform = form == 3;
        std::vector<uint32_t>& out = nform ? s.prog2n : v2 ? s.prog2 : s.prog;
        narrow_emit_ = nform ? &s.slot_aw : nullptr;
        std::vector<uint32_t> produced_in_group(s.n_vars, UINT32_MAX);  // var -> id of the open group that produces it
        uint32_t group_id = 0, slots_done = 0;
        std::vector<size_t> group;  // op indices of the open group
        // SELECT flags as bit planes (plain loop kernels: ZK_OP_FLAG_PLANES, kernels_engine2.hpp): the FLAG_PLANES most used flag
        // variables of a loop scope get a plane id; a flag is copied into its plane by a ZK_OP_FLAG_PLANES op emitted lazily, in
        // front of the first SELECT that needs it, together with every other flag produced by then (up to 7 per op)
        const char* fp_env = getenv("ZKGL_FLAG_PLANES");
        const bool planes_on = v2 && s.is_loop && !(fp_env && fp_env[0] == '0');
        std::vector<uint32_t> plane_of = planes_on ? select_plane_vars(s) : std::vector<uint32_t>(s.n_vars, UINT32_MAX);
        std::vector<uint8_t> plane_saved(s.n_vars, 0);
        std::vector<uint32_t> plane_pending;   // produced, not yet copied
        if (planes_on) { s.flag_planes = 0; for (uint32_t pv : plane_of) s.flag_planes += pv != UINT32_MAX; }
        plane_of_ = planes_on ? &plane_of : nullptr;
        auto operand = [&](const OpRec& op, size_t pos) {
            const Operand& in = op.ins[pos];
            if (v2) { operand_v2(s, op, pos, out); return; }
            if (in.kind == Operand::VAR) out.push_back(s.var_slot[in.idx]);
            else if (in.kind == Operand::CONSTPOOL) out.push_back(ZK_OPERAND_CONST | in.idx);
            else if (in.kind == Operand::OUTER_VAR) out.push_back(ZK_OPERAND_OUTER | outer_.var_slot[in.idx]);
            else out.push_back(op.opcode == ZK_OP_LOOP_LAST ? loop_.var_slot[in.idx] : in.idx);
        };
        auto flush = [&]() {
            if (group.empty()) return;
            const OpRec& first = s.ops[group[0]];
            const size_t n = group.size();
            const bool counted = group_cap(first, v2) > 1 || first.opcode == ZK_OP_INPUT || first.opcode == ZK_OP_SELECT || first.opcode == ZK_OP_FMA ||
                                 first.opcode == ZK_OP_LC4 || (v2 && first.opcode == ZK_OP_U32MULADD);
            if (nform) {   // class word of this header: outputs in the order the kernel stores them (members in order, each member's outputs in order)
                uint32_t mask = 0, k = 0;
                for (size_t oi : group)
                    for (uint32_t ov : s.ops[oi].outs) { if (slot_is_byte[s.var_slot[ov]]) { if (k >= 32) throw ZkError(ZK_ERR_INVALID, "internal: byte-class output beyond the class word"); mask |= 1u << k; } ++k; }
                cls_words.push_back(mask);
            }
            if (v2) emit_group_v2(s, group, counted, out);
            else if (first.opcode == ZK_OP_LOOKUP) {
                out.push_back((uint32_t)ZK_OP_LOOKUP | ((uint32_t)first.a << 8) | (((uint32_t)first.b | ((uint32_t)(n - 1) << 8)) << 16));
                out.push_back(first.ins[0].idx);  // table id
                for (size_t oi : group)
                    for (size_t q = 1; q < s.ops[oi].ins.size(); ++q) operand(s.ops[oi], q);
            } else {
                out.push_back((uint32_t)first.opcode | ((uint32_t)first.a << 8) | ((counted ? (uint32_t)(n - 1) : (uint32_t)first.b) << 16));
                for (size_t oi : group)
                    for (size_t q = 0; q < s.ops[oi].ins.size(); ++q) operand(s.ops[oi], q);
            }
            for (size_t oi : group) {
                if (!v2) emit_dests(s, s.ops[oi], out);
                slots_done += (uint32_t)s.ops[oi].outs.size();
            }
            group.clear();
            ++group_id;
        };
        for (size_t oi = 0; oi < s.ops.size(); ++oi) {
            if (!s.is_loop && (oi == s.pre_ops || oi == s.side_ops)) flush();  // phases are launched separately
            if (!s.is_loop && oi == s.pre_ops) { (v2 ? s.pre_words2 : s.pre_words) = (uint32_t)out.size(); if (v2) s.pre_slots = slots_done; }
            if (!s.is_loop && oi == s.side_ops) { (v2 ? s.side_words2 : s.side_words) = (uint32_t)out.size(); if (v2) s.side_slots = slots_done; }
            const OpRec& op = s.ops[oi];
            if (op.seed_only) continue;
            // join the open group when the kind matches and no operand comes from a member of that group
            bool joins = grouping && !group.empty() && group.size() < group_cap(op, v2) && s.ops[group[0]].opcode == op.opcode;
            if (joins && op.opcode == ZK_OP_LOOKUP) {
                const OpRec& f = s.ops[group[0]];
                joins = f.a == op.a && f.b == op.b && f.ins[0].idx == op.ins[0].idx;
            }
            if (joins)
                for (auto& in : op.ins)
                    if (in.kind == Operand::VAR && produced_in_group[in.idx] == group_id) { joins = false; break; }
            if (planes_on && op.opcode == ZK_OP_SELECT) {
                const uint32_t fv = op.ins[0].idx;
                const bool mine = plane_of[fv] != UINT32_MAX;
                // a group is homogeneous: plane flags or slot flags
                if (joins && ((plane_of[s.ops[group[0]].ins[0].idx] != UINT32_MAX) != mine)) joins = false;
                if (mine && !plane_saved[fv]) {
                    flush();   // the flag's producer may sit in the open group
                    joins = false;
                    for (size_t at = 0; at < plane_pending.size(); at += 7) {
                        const size_t nn = std::min<size_t>(7, plane_pending.size() - at);
                        out.push_back((uint32_t)ZK_OP_FLAG_PLANES | ((uint32_t)(nn - 1) << 16));
                        if (nform) cls_words.push_back(0);   // a header without outputs
                        for (size_t k = 0; k < nn; ++k) { out.push_back(nform ? s.slot_aw[s.var_slot[plane_pending[at + k]]] : s.var_slot[plane_pending[at + k]]); out.push_back(plane_of[plane_pending[at + k]]); plane_saved[plane_pending[at + k]] = 1; }
                    }
                    plane_pending.clear();
                    if (!plane_saved[fv]) throw ZkError(ZK_ERR_INVALID, "internal: SELECT flag not produced before its use");
                }
            }
            if (!joins) flush();
            group.push_back(oi);
            for (uint32_t ov : op.outs) {
                produced_in_group[ov] = group_id;
                if (planes_on && plane_of[ov] != UINT32_MAX) plane_pending.push_back(ov);
            }
        }
        flush();
        plane_of_ = nullptr;
        narrow_emit_ = nullptr;
        if (nform) {   // [program, padded for the kernel's 16-word fetches][class words]
            if (out.size() != s.prog2.size()) throw ZkError(ZK_ERR_INVALID, "internal: the narrow form of the loop program differs in length from the plain form");
            out.resize(((out.size() + 63) / 64) * 64 + 64, 0);
            s.cls_off = (uint32_t)out.size();
            out.insert(out.end(), cls_words.begin(), cls_words.end());
            out.resize(out.size() + 16, 0);
        }
        if (!s.is_loop && s.pre_ops >= s.ops.size()) { (v2 ? s.pre_words2 : s.pre_words) = (uint32_t)out.size(); if (v2) s.pre_slots = slots_done; }
        if (!s.is_loop && s.side_ops >= s.ops.size()) { (v2 ? s.side_words2 : s.side_words) = (uint32_t)out.size(); if (v2) s.side_slots = slots_done; }
    };
    emit_form(1);
    emit_form(2);
    if (!slot_is_byte.empty()) {   // the narrow form is an extra: anything unexpected in it leaves the circuit without a narrow layout, never without its programs
        try { emit_form(3); }
        catch (const ZkError& e) {
            if (getenv("ZKGL_PROG_STATS")) fprintf(stderr, "[zkgl] no narrow layout: %s\n", e.what());
            s.prog2n.clear(); s.slot_aw.clear(); s.cls_off = 0; narrow_emit_ = nullptr; plane_of_ = nullptr;
        }
    }
    if (!s.is_loop && s.side_words < s.pre_words) { s.side_words = s.pre_words; s.side_words2 = s.pre_words2; s.side_slots = s.pre_slots; }
    if (getenv("ZKGL_PROG_STATS")) {
        // op mix of the scope: ops, operands by kind (data positions), outputs; distance (in produced values) from an operand's producer
        std::map<uint32_t, std::array<uint64_t, 6>> mix;  // opcode -> ops, var operands, const operands, outer operands, outs, raw
        std::vector<uint32_t> born(s.n_vars, 0);
        uint64_t produced = 0, hist[8] = {0};
        for (auto& op : s.ops) {
            if (op.seed_only) continue;
            auto& m = mix[op.opcode];
            m[0]++;
            for (auto& in : op.ins) {
                if (in.kind == Operand::VAR) {
                    m[1]++;
                    const uint64_t d = produced - born[in.idx];
                    hist[d <= 8 ? 0 : d <= 32 ? 1 : d <= 128 ? 2 : d <= 512 ? 3 : d <= 2048 ? 4 : d <= 8192 ? 5 : 6]++;
                } else if (in.kind == Operand::CONSTPOOL) m[2]++;
                else if (in.kind == Operand::OUTER_VAR) m[3]++;
                else m[5]++;
            }
            m[4] += op.outs.size();
            for (uint32_t ov : op.outs) born[ov] = (uint32_t)produced++;
        }
        fprintf(stderr, "[zkgl] %s scope: %zu ops, %u vars, device programs %zu (v1) / %zu (v2) words\n", s.is_loop ? "loop" : "outer", s.ops.size(), s.n_vars, s.prog.size(), s.prog2.size());
        for (auto& kv : mix)
            fprintf(stderr, "   op %2u: n=%6llu var=%7llu const=%6llu outer=%6llu raw=%6llu outs=%7llu\n", kv.first, (unsigned long long)kv.second[0],
                    (unsigned long long)kv.second[1], (unsigned long long)kv.second[2], (unsigned long long)kv.second[3], (unsigned long long)kv.second[5],
                    (unsigned long long)kv.second[4]);
        {
            std::vector<uint8_t> is_const(s.n_vars, 0);
            for (auto& op : s.ops) if (!op.seed_only && op.opcode == ZK_OP_CONST) for (uint32_t ov : op.outs) is_const[ov] = 1;
            uint64_t op_refs = 0, op_const = 0, gate_refs = 0, gate_const = 0;
            for (auto& op : s.ops) if (!op.seed_only) for (auto& in : op.ins) if (in.kind == Operand::VAR) { op_refs++; op_const += is_const[in.idx]; }
            for (auto& g : s.gates) for (uint32_t v : g.vars) { gate_refs++; gate_const += is_const[v]; }
            for (auto& l : s.lookups) for (uint32_t v : l.vars) { gate_refs++; gate_const += is_const[v]; }
            fprintf(stderr, "   operand references to constant variables: ops %llu of %llu, gates/lookups %llu of %llu\n", (unsigned long long)op_const,
                    (unsigned long long)op_refs, (unsigned long long)gate_const, (unsigned long long)gate_refs);
        }
        {   // LRU model of a wavefront's share of L2 (K values of 512 B; reads and writes touch): operand fetches left to HBM, with the
            // Poseidon2 intermediates (never read again by the witness kernel) allocating like any store, or streamed past the cache
            auto lru_misses = [&](size_t K, bool p2_bypass) {
                std::vector<int64_t> last(s.n_vars, -1);
                int64_t stamp = 0; uint64_t miss = 0;
                std::vector<int64_t> ring;   // stamps are distinct per touch; a value is resident iff fewer than K distinct values were touched since
                // distinct-touch counting with a Fenwick tree over stamps
                const size_t cap = 4 * (size_t)s.n_vars + 8 * s.ops.size() + 64;
                std::vector<int32_t> bit(cap + 1, 0);
                auto upd = [&](size_t i, int d) { for (++i; i <= cap; i += i & (~i + 1)) bit[i] += d; };
                auto sum = [&](size_t i) { int64_t r = 0; for (++i; i > 0; i -= i & (~i + 1)) r += bit[i]; return r; };
                auto touch = [&](uint32_t v, bool read) {
                    if (read) {
                        if (last[v] < 0) ++miss;
                        else { const int64_t distinct_since = sum((size_t)stamp) - sum((size_t)last[v]); if ((size_t)distinct_since >= K) ++miss; }
                    }
                    if (last[v] >= 0) upd((size_t)last[v], -1);
                    ++stamp; last[v] = stamp; upd((size_t)stamp, +1);
                };
                for (auto& op : s.ops) {
                    if (op.seed_only) continue;
                    for (auto& in : op.ins) if (in.kind == Operand::VAR) touch(in.idx, true);
                    for (size_t q = 0; q < op.outs.size(); ++q) {
                        if (p2_bypass && op.opcode == ZK_OP_P2_ROUNDS && q + 12 < op.outs.size()) continue;
                        touch(op.outs[q], false);
                    }
                }
                return miss;
            };
            {   // how many of the modelled fetches (16 values) are SELECT flags, and how concentrated they are
                const size_t K = 16;
                std::vector<int64_t> last(s.n_vars, -1);
                int64_t stamp = 0;
                const size_t cap = 4 * (size_t)s.n_vars + 8 * s.ops.size() + 64;
                std::vector<int32_t> bit(cap + 1, 0);
                auto upd = [&](size_t i, int d) { for (++i; i <= cap; i += i & (~i + 1)) bit[i] += d; };
                auto sum = [&](size_t i) { int64_t r = 0; for (++i; i > 0; i -= i & (~i + 1)) r += bit[i]; return r; };
                std::vector<uint32_t> flag_miss(s.n_vars, 0), flag_uses(s.n_vars, 0);
                uint64_t miss_all = 0, miss_flags = 0, sel = 0;
                std::map<uint32_t, uint64_t> miss_by_op;   // opcode * 4 + min(operand position, 3)
                auto touch = [&](uint32_t v, bool read) -> bool {
                    bool miss = false;
                    if (read) miss = last[v] < 0 || (size_t)(sum((size_t)stamp) - sum((size_t)last[v])) >= K;
                    if (last[v] >= 0) upd((size_t)last[v], -1);
                    ++stamp; last[v] = stamp; upd((size_t)stamp, +1);
                    return miss;
                };
                const std::vector<uint32_t> plane_of = select_plane_vars(s);
                uint64_t plane_reads = 0;
                for (auto& op : s.ops) {
                    if (op.seed_only) continue;
                    for (size_t q = 0; q < op.ins.size(); ++q) {
                        if (op.ins[q].kind != Operand::VAR) continue;
                        if (op.opcode == ZK_OP_SELECT && q == 0 && plane_of[op.ins[0].idx] != UINT32_MAX) { ++plane_reads; continue; }   // from LDS
                        const bool m = touch(op.ins[q].idx, true);
                        miss_all += m;
                        if (m) ++miss_by_op[op.opcode * 4 + std::min<size_t>(q, 3)];
                        if (op.opcode == ZK_OP_SELECT && q == 0) { ++sel; ++flag_uses[op.ins[q].idx]; if (m) { ++miss_flags; ++flag_miss[op.ins[q].idx]; } }
                    }
                    for (auto ov : op.outs) touch(ov, false);
                }
                std::vector<std::pair<uint32_t, uint32_t>> fl;
                for (uint32_t v = 0; v < s.n_vars; ++v) if (flag_uses[v]) fl.push_back({flag_miss[v], flag_uses[v]});
                std::sort(fl.rbegin(), fl.rend());
                fprintf(stderr, "   %llu SELECT flag reads come from bit planes (not modelled as fetches); modelled fetches (16 values) %llu\n", (unsigned long long)plane_reads, (unsigned long long)miss_all);
                {   // SELECT data operands: how many are the output of another SELECT (mux chains), and how recent that output is
                    std::vector<int64_t> born_at(s.n_vars, -1); std::vector<uint8_t> by_select(s.n_vars, 0);
                    int64_t produced2 = 0; uint64_t n_sel = 0, chain[2] = {0, 0}, chain_near[2] = {0, 0}, both_old = 0;
                    for (auto& op : s.ops) {
                        if (op.seed_only) continue;
                        if (op.opcode == ZK_OP_SELECT) {
                            ++n_sel;
                            bool any_near = false;
                            for (int q = 1; q <= 2; ++q) {
                                if (op.ins[q].kind != Operand::VAR) continue;
                                const uint32_t v = op.ins[q].idx;
                                if (by_select[v]) { ++chain[q - 1]; if (produced2 - born_at[v] <= 16) { ++chain_near[q - 1]; any_near = true; } }
                                else if (produced2 - born_at[v] <= 16) any_near = true;
                            }
                            both_old += !any_near;
                        }
                        for (uint32_t ov : op.outs) { born_at[ov] = produced2++; by_select[ov] = op.opcode == ZK_OP_SELECT; }
                    }
                    fprintf(stderr, "   SELECT data operands: %llu selects; a is a SELECT output in %llu (%llu of them within 16 values), b in %llu (%llu); %llu selects with both operands older than 16 values\n",
                            (unsigned long long)n_sel, (unsigned long long)chain[0], (unsigned long long)chain_near[0], (unsigned long long)chain[1], (unsigned long long)chain_near[1], (unsigned long long)both_old);
                }
                fprintf(stderr, "   modelled fetches by op / operand position:");
                for (auto& kv : miss_by_op) fprintf(stderr, " %u.%u=%llu", kv.first / 4, kv.first % 4, (unsigned long long)kv.second);
                fprintf(stderr, "\n");
                fprintf(stderr, "   SELECT flags: %llu selects, %zu distinct flags, %llu of %llu modelled fetches are flags;", (unsigned long long)sel, fl.size(),
                        (unsigned long long)miss_flags, (unsigned long long)miss_all);
                uint64_t acc = 0, accu = 0; size_t k = 0;
                for (size_t N : {16, 32, 64, 128, 256}) { for (; k < fl.size() && k < N; ++k) { acc += fl[k].first; accu += fl[k].second; } fprintf(stderr, " top %zu: %llu fetches / %llu uses;", N, (unsigned long long)acc, (unsigned long long)accu); }
                fprintf(stderr, "\n");
            }
            for (size_t K : {16, 32, 64, 128})
                fprintf(stderr, "   LRU model, %zu values per wavefront: %llu operand fetches; with the Poseidon2 intermediates streamed past the cache %llu\n", K,
                        (unsigned long long)lru_misses(K, false), (unsigned long long)lru_misses(K, true));
            if (s.is_loop && !s.slot_aw.empty()) {
                // the same model in BYTES, for the narrow store (store_geom.hpp): a wavefront's share of L2 is C bytes; a value occupies 512 B of it, or 64 B when it
                // is byte-class in the narrow layout; a fetch costs the value's bytes.  SELECT flags read from the LDS planes are not fetches.  What the ratio of the
                // two layouts predicts is how the kernel's FETCH_SIZE scales (profiles/r6_predictions.md).
                const std::vector<uint32_t> planes = select_plane_vars(s);
                auto lru_bytes = [&](uint64_t C, bool narrow) {
                    std::vector<int64_t> last(s.n_vars, -1);
                    int64_t stamp = 0; uint64_t fetched = 0;
                    const size_t cap = 4 * (size_t)s.n_vars + 8 * s.ops.size() + 64;
                    std::vector<int64_t> bit(cap + 1, 0);
                    auto upd = [&](size_t i, int64_t d) { for (++i; i <= cap; i += i & (~i + 1)) bit[i] += d; };
                    auto sum = [&](size_t i) { int64_t r = 0; for (++i; i > 0; i -= i & (~i + 1)) r += bit[i]; return r; };
                    auto width = [&](uint32_t v) -> int64_t { return (narrow && (s.slot_aw[s.var_slot[v]] & zkgeom::AW_BYTE)) ? 64 : 512; };
                    auto touch = [&](uint32_t v, bool read) {
                        const int64_t w = width(v);
                        if (read && (last[v] < 0 || (uint64_t)(sum((size_t)stamp) - sum((size_t)last[v])) >= C)) fetched += (uint64_t)w;
                        if (last[v] >= 0) upd((size_t)last[v], -w);
                        ++stamp; last[v] = stamp; upd((size_t)stamp, w);
                    };
                    for (auto& op : s.ops) {
                        if (op.seed_only) continue;
                        for (size_t q = 0; q < op.ins.size(); ++q) {
                            if (op.ins[q].kind != Operand::VAR) continue;
                            if (op.opcode == ZK_OP_SELECT && q == 0 && planes[op.ins[0].idx] != UINT32_MAX) continue;
                            touch(op.ins[q].idx, true);
                        }
                        for (uint32_t ov : op.outs) touch(ov, false);
                    }
                    return fetched;
                };
                for (uint64_t C : {8192ull, 16384ull, 32768ull}) {
                    const uint64_t a = lru_bytes(C, false), b = lru_bytes(C, true);
                    fprintf(stderr, "   LRU model in bytes, %llu B of L2 per wavefront: operand bytes fetched per wavefront-cycle %llu (ordinary store) / %llu (narrow store) = %.3f\n",
                            (unsigned long long)C, (unsigned long long)a, (unsigned long long)b, a ? (double)b / (double)a : 0.0);
                }
            }
        }
        fprintf(stderr, "   operand age (values produced since): <=8 %llu, <=32 %llu, <=128 %llu, <=512 %llu, <=2048 %llu, <=8192 %llu, more %llu\n",
                (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4],
                (unsigned long long)hist[5], (unsigned long long)hist[6]);
    }
    for (auto& g : s.gates)
        for (uint32_t v : g.vars)
            if (!defined[v]) throw ZkError(ZK_ERR_UNRESOLVED, "gate references a variable without a witness producer");
    for (auto& l : s.lookups)
        for (uint32_t v : l.vars)
            if (!defined[v]) throw ZkError(ZK_ERR_UNRESOLVED, "lookup references a variable without a witness producer");
}

// Cone seeding program.  The carried words of iteration k+1 depend on a (usually small) part of iteration k: queue
// heads, accumulators, FSM flags — not on the range-check decompositions, gate intermediates or the 962 constrained
// Poseidon2 intermediates.  Keep only the backward slice of the carried outputs, collapse P2_ROUNDS to the 12-output
// permutation and assign every surviving value an LDS slot by linear scan over its live range.
void CS::build_seed_program() {
    seed_prog_.clear(); seed_carries_.clear(); seed_slots_ = 0; seed_ops_ = 0; seed_cone_unsupported_ = false;
    if (!limit_ || carries_store_.empty()) return;
    const Scope& s = loop_;
    // the cone is built from the ops in RECORDING order: the locality schedule of the trace program (schedule_by_locality) stretches
    // live ranges inside the cone, and the seed kernels hold every live value of an instance in LDS
    // the cone runs the gated witness-only permutations (ZK_OP_POSEIDON2 a = 1) ungated: their outputs only ever reach a carried word
    // through a select on the same flag, and the seed kernels keep one POSEIDON2 form
    std::vector<OpRec> sops_store = loop_ops_recorded_.empty() ? s.ops : loop_ops_recorded_;
    std::vector<uint32_t> out_vars;  // same order as carries_
    for (auto& l : links_raw_)
        if (l.kind == ZK_LINK_CARRY && s.input_word.count(l.loop_cell)) out_vars.push_back(l.other_cell);
    // ... which is VERIFIED, not assumed (ZK_OP_POSEIDON2 a = 1 is recordable through the public zk_cs_emit_op): the ungated outputs
    // differ from the trace's (zeros) exactly where the flag is off.  Taint every value computed from them; a SELECT on the op's own flag
    // that takes the tainted value on its flag-on side only is clean again (flag off: the other branch; flag on: the permutation really
    // ran).  A carried output that stays tainted would be seeded with words the trace does not hold: the cone is then not offered
    // (seed_cone_unsupported_; carry links would reject it anyway — a spurious UNSATISFIED, never unsoundness).
    {
        std::vector<uint32_t> taint(s.n_vars, UINT32_MAX);   // variable -> flag variable of the gated permutation it depends on (UINT32_MAX: clean; UINT32_MAX - 1: several)
        const uint32_t SEVERAL = UINT32_MAX - 1;
        bool any_gated = false;
        for (auto& op : sops_store) {
            // (seed_only hint ops are part of the cone: they propagate taint like any other op — ADVICE r5)
            uint32_t t = UINT32_MAX;
            auto join = [&](uint32_t x) { if (x == UINT32_MAX) return; t = (t == UINT32_MAX || t == x) ? x : SEVERAL; };
            if (op.opcode == ZK_OP_POSEIDON2 && op.a == 1) {
                any_gated = true;
                // a flag that is not a variable (pool constant), or a malformed operand list: nothing to reason about -> not offered
                if (op.ins.size() != 13 || op.ins[12].kind != Operand::VAR) { t = SEVERAL; }
                else {
                    // where the flag f is off the ungated cone computes P(state) while the trace holds zeros: the outputs are tainted by f.
                    // A state input (or the flag) already tainted by ANOTHER flag g makes the ungated value wrong also where f is on and g
                    // is off (chained gated permutations under different flags): 'several'.  Tainted by f itself: still exactly "wrong where f is off".
                    const uint32_t f = op.ins[12].idx;
                    for (auto& in : op.ins) if (in.kind == Operand::VAR) join(taint[in.idx]);
                    t = (t == UINT32_MAX || t == f) ? f : SEVERAL;
                }
            } else {
                if (!any_gated) continue;
                if (op.opcode == ZK_OP_SELECT && op.ins.size() == 3 && op.ins[0].kind == Operand::VAR) {
                    const uint32_t f = op.ins[0].idx;
                    join(taint[f]);                                                               // a tainted selector taints the result
                    if (op.ins[1].kind == Operand::VAR && taint[op.ins[1].idx] != f) join(taint[op.ins[1].idx]);   // flag-on side: clean when tainted by THIS flag
                    if (op.ins[2].kind == Operand::VAR) join(taint[op.ins[2].idx]);                // flag-off side: always propagates
                } else {
                    for (auto& in : op.ins) if (in.kind == Operand::VAR) join(taint[in.idx]);
                }
            }
            // a variable written by a hint AND by its gate-by-gate producers keeps the join of both (the cone may use either)
            for (uint32_t ov : op.outs) {
                const uint32_t prev = taint[ov];
                if (t == UINT32_MAX) continue;
                taint[ov] = (prev == UINT32_MAX || prev == t) ? t : SEVERAL;
            }
        }
        uint32_t bad = 0;
        for (uint32_t v : out_vars) bad += taint[v] != UINT32_MAX;
        if (bad) seed_cone_unsupported_ = true;
        if (getenv("ZKGL_PROG_STATS") && any_gated) fprintf(stderr, "[zkgl] seeding cone: %u of %zu carried outputs depend on a gated permutation outside a select on its flag\n", bad, out_vars.size());
    }
    for (auto& op : sops_store)
        if (op.opcode == ZK_OP_POSEIDON2 && op.a == 1) { op.ins.pop_back(); op.a = 0; }
    const std::vector<OpRec>& sops = sops_store;
    std::vector<uint8_t> need(s.n_vars, 0), keep(sops.size(), 0);
    for (auto v : out_vars) need[v] = 1;
    for (size_t oi = sops.size(); oi-- > 0;) {
        const OpRec& op = sops[oi];
        bool any = false;
        for (auto o : op.outs) any |= need[o] != 0;
        if (!any) continue;
        if (op.opcode == ZK_OP_P2_ROUNDS)
            for (size_t i = 0; i + 12 < op.outs.size(); ++i)
                if (need[op.outs[i]]) return;  // an intermediate feeds the state: keep the generic mode
        keep[oi] = 1;
        if (op.seed_only)  // the hint produces these values for the cone: their gate-by-gate producers are not needed
            for (auto o : op.outs) need[o] = 0;
        for (auto& in : op.ins)
            if (in.kind == Operand::VAR) need[in.idx] = 1;
    }
    seed_v2_ok_ = true;  // every op of the cone has a handler in the scalar-decoded seed kernel (kernels_engine2.hpp run_seed2)
    for (size_t oi = 0; oi < sops.size(); ++oi) {
        if (!keep[oi]) continue;
        switch (sops[oi].opcode) {
        case ZK_OP_CONST: case ZK_OP_INPUT: case ZK_OP_FMA: case ZK_OP_LC4: case ZK_OP_SELECT: case ZK_OP_ISZERO: case ZK_OP_UADD: case ZK_OP_USUB:
        case ZK_OP_DOT4: case ZK_OP_MATMUL12: case ZK_OP_SPLIT: case ZK_OP_LOOKUP: case ZK_OP_POSEIDON2: case ZK_OP_P2_ROUNDS: case ZK_OP_U32MULADD:
        case ZK_OP_DIVREM: case ZK_OP_U256_MULWIDE: case ZK_OP_U256_DIVREM: break;
        default: seed_v2_ok_ = false;
        }
        if (sops[oi].opcode == ZK_OP_BYTEBUF_FILL) seed_cone_unsupported_ = true;   // no seed kernel carries this macro-op (and it has no hint)
    }
    const int64_t INF = INT64_MAX;
    std::vector<int64_t> last_use(s.n_vars, -1);
    for (size_t oi = 0; oi < sops.size(); ++oi)
        if (keep[oi])
            for (auto& in : sops[oi].ins)
                if (in.kind == Operand::VAR) last_use[in.idx] = (int64_t)oi;
    for (auto v : out_vars) last_use[v] = INF;
    std::vector<uint32_t> slot_of(s.n_vars, UINT32_MAX), free_slots;
    uint32_t n_slots = 0;
    const uint32_t DISCARD_MARK = 0x3fffffffu;
    std::vector<uint32_t> prog;
    for (size_t oi = 0; oi < sops.size(); ++oi) {
        if (!keep[oi]) continue;
        const OpRec& op = sops[oi];
        const bool collapse = op.opcode == ZK_OP_P2_ROUNDS;
        prog.push_back((collapse ? (uint32_t)ZK_OP_POSEIDON2 : (uint32_t)op.opcode) | ((uint32_t)op.a << 8) | ((uint32_t)op.b << 16));
        for (auto& in : op.ins) {
            switch (in.kind) {
            case Operand::VAR: prog.push_back(slot_of[in.idx]); break;
            case Operand::CONSTPOOL: prog.push_back(ZK_OPERAND_CONST | in.idx); break;
            case Operand::OUTER_VAR: prog.push_back(ZK_OPERAND_OUTER | outer_.var_slot[in.idx]); break;
            case Operand::RAW: prog.push_back(in.idx); break;
            }
        }
        for (size_t i = collapse ? op.outs.size() - 12 : 0; i < op.outs.size(); ++i) {
            const uint32_t ov = op.outs[i];
            if (last_use[ov] < 0) { prog.push_back(DISCARD_MARK); continue; }
            uint32_t sl;
            if (!free_slots.empty()) { sl = free_slots.back(); free_slots.pop_back(); }
            else sl = n_slots++;
            slot_of[ov] = sl;
            prog.push_back(sl);
        }
        for (auto& in : op.ins)  // release after the outputs are placed: an op never writes over its own operands
            if (in.kind == Operand::VAR && last_use[in.idx] == (int64_t)oi && slot_of[in.idx] != UINT32_MAX) {
                free_slots.push_back(slot_of[in.idx]);
                last_use[in.idx] = -2;  // the same variable may appear twice among the operands
            }
        ++seed_ops_;
    }
    const uint32_t discard = n_slots++;
    for (auto& w : prog)
        if (w == DISCARD_MARK) w = discard;
    if (n_slots + s.n_input_words > zkdev::seed_cone_max_slots()) { seed_ops_ = 0; return; }
    for (size_t i = 0; i < carries_store_.size(); ++i) {
        Carry c = carries_store_[i];
        if (slot_of[out_vars[i]] == UINT32_MAX) { seed_ops_ = 0; return; }  // carried output produced by no op
        c.out_cell = slot_of[out_vars[i]];
        seed_carries_.push_back(c);
    }
    seed_prog_ = std::move(prog);
    seed_slots_ = n_slots;

    // ---- strand form of the same cone (k_seed_cone_strands): one iteration is a latency chain for a single wavefront, but
    // its op graph is wide (independent sponges, byte decompositions).  Ops are ordered by dependency level, dealt over 8
    // strands per level, and an LDS slot is recycled only at the level after its last reader (readers of a level run
    // concurrently with that level's writers).
    seed_sprog_.clear(); seed_scarries_.clear(); seed_sslots_ = 0; seed_sgain_ = 0;
    seed_wprog_.clear(); seed_wcarries_.clear();
    constexpr uint32_t NS = zkdev::SEED_STRANDS_PER_TILE;
    // Levels in two tiers so that the heavy ops of independent chains line up: tier(op) = the number of heavy ops (permutations,
    // hash macro-ops) on the longest path of producers before it; inside a tier the light ops are levelled as soon as
    // possible and ALL heavy ops of the tier share one final level (their consumers sit in the next tier).  With plain
    // as-soon-as-possible levels the k-th permutations of different sponges land on different levels, one strand busy each.
    std::vector<int64_t> producer(s.n_vars, -1);
    std::vector<uint32_t> level(sops.size(), 0);
    uint32_t n_levels = 0;
    {
        auto heavy = [&](const OpRec& op) {
            // U256_DIVREM (256 dependent shift-subtract steps) is a heavy op too: the VM cycle's two divisions then share a level (one
            // segment of the op-parallel kernel, two strands of the strand kernel) instead of running one after the other
            return op.opcode == ZK_OP_P2_ROUNDS || op.opcode == ZK_OP_POSEIDON2 || op.opcode == ZK_OP_KECCAK_ABSORB || op.opcode == ZK_OP_SHA256_COMPRESS ||
                   op.opcode == ZK_OP_NN_MULMOD || op.opcode == ZK_OP_U256_DIVREM;
        };
        std::vector<uint32_t> tier(sops.size(), 0), local(sops.size(), 0);
        uint32_t n_tiers = 0;
        for (size_t oi = 0; oi < sops.size(); ++oi) {
            if (!keep[oi]) continue;
            const OpRec& op = sops[oi];
            uint32_t t = 0;
            for (auto& in : op.ins)
                if (in.kind == Operand::VAR && producer[in.idx] >= 0) {
                    const size_t p = (size_t)producer[in.idx];
                    t = std::max(t, tier[p] + (heavy(sops[p]) ? 1u : 0u));
                }
            uint32_t l = 0;
            if (!heavy(op))
                for (auto& in : op.ins)
                    if (in.kind == Operand::VAR && producer[in.idx] >= 0) {
                        const size_t p = (size_t)producer[in.idx];
                        if (tier[p] == t && !heavy(sops[p])) l = std::max(l, local[p] + 1);
                    }
            tier[oi] = t; local[oi] = l;
            n_tiers = std::max(n_tiers, t + 1);
            for (size_t i = op.opcode == ZK_OP_P2_ROUNDS ? op.outs.size() - 12 : 0; i < op.outs.size(); ++i) producer[op.outs[i]] = (int64_t)oi;
        }
        std::vector<uint32_t> light_levels(n_tiers, 0), has_heavy(n_tiers, 0), base(n_tiers + 1, 0);
        for (size_t oi = 0; oi < sops.size(); ++oi) {
            if (!keep[oi]) continue;
            if (heavy(sops[oi])) has_heavy[tier[oi]] = 1;
            else light_levels[tier[oi]] = std::max(light_levels[tier[oi]], local[oi] + 1);
        }
        for (uint32_t t = 0; t < n_tiers; ++t) base[t + 1] = base[t] + light_levels[t] + has_heavy[t];
        n_levels = base[n_tiers];
        for (size_t oi = 0; oi < sops.size(); ++oi)
            if (keep[oi]) level[oi] = base[tier[oi]] + (heavy(sops[oi]) ? light_levels[tier[oi]] : local[oi]);
    }
    std::vector<std::vector<uint32_t>> by_level(n_levels);
    std::vector<int64_t> last_level(s.n_vars, -1);
    for (size_t oi = 0; oi < sops.size(); ++oi) {
        if (!keep[oi]) continue;
        by_level[level[oi]].push_back((uint32_t)oi);
        for (auto& in : sops[oi].ins)
            if (in.kind == Operand::VAR) last_level[in.idx] = std::max<int64_t>(last_level[in.idx], level[oi]);
    }
    for (auto v : out_vars) last_level[v] = INF;
    // ---- op-parallel form of the same cone (kernels_seed_wave.hpp k_seed_wave): ONE wavefront per instance, its 64 lanes run up to
    // 64 independent ops of a dependency level at once (segments of one op kind, 16-bit records resident in LDS), Poseidon2
    // permutations cooperatively on 12 lanes each.  No workgroup barrier per level, no wasted lanes: the latency chain of a cycle is
    // levels x (a few dozen instructions) + permutation tiers.  Loop-invariant ZK_OP_CONST ops (pool constants, outer imports) run once
    // in a prologue and keep their slots.  Cones with ops the kernel has no lane form for keep the strand kernel.
    {
        seed_wprog_.clear(); seed_wcarries_.clear(); seed_wslots_ = 0; seed_wpro_words_ = 0;
        bool ok = true;
        int fail_reason = 0;
        auto fail_ = [&](int why) { if (ok) fail_reason = why; ok = false; };
        enum : uint16_t { WK_CONST = 1, WK_INPUT, WK_SELECT, WK_FMA, WK_LC4, WK_ISZERO, WK_UADD, WK_USUB, WK_DOT4, WK_SPLIT_S, WK_SPLIT_L, WK_LOOKUP, WK_P2,
                          WK_U32MULADD, WK_DIVREM, WK_U256MUL, WK_U256DIV, WK_END = 0xffff };
        auto kind_of = [&](const OpRec& op) -> uint16_t {
            switch (op.opcode) {
            case ZK_OP_CONST: return WK_CONST; case ZK_OP_INPUT: return WK_INPUT; case ZK_OP_SELECT: return WK_SELECT; case ZK_OP_FMA: return WK_FMA;
            case ZK_OP_LC4: return WK_LC4; case ZK_OP_ISZERO: return WK_ISZERO; case ZK_OP_UADD: return WK_UADD; case ZK_OP_USUB: return WK_USUB;
            case ZK_OP_DOT4: return WK_DOT4; case ZK_OP_SPLIT: return op.a <= 9 ? WK_SPLIT_S : (op.a <= 65 ? WK_SPLIT_L : 0);
            case ZK_OP_LOOKUP: return WK_LOOKUP; case ZK_OP_POSEIDON2: case ZK_OP_P2_ROUNDS: return WK_P2; case ZK_OP_U32MULADD: return WK_U32MULADD;
            case ZK_OP_DIVREM: return WK_DIVREM; case ZK_OP_U256_MULWIDE: return WK_U256MUL; case ZK_OP_U256_DIVREM: return WK_U256DIV;
            default: return 0;
            }
        };
        auto rec_words = [](uint16_t k) -> uint32_t {
            switch (k) {
            case WK_CONST: case WK_INPUT: case WK_SELECT: case WK_ISZERO: case WK_DIVREM: return 4;
            case WK_FMA: case WK_UADD: case WK_USUB: case WK_U32MULADD: case WK_LOOKUP: return 8;
            case WK_LC4: case WK_DOT4: case WK_SPLIT_S: return 12;
            case WK_P2: return 24; case WK_U256MUL: case WK_U256DIV: return 32; case WK_SPLIT_L: return 68;
            default: return 0;
            }
        };
        std::vector<uint32_t> wslot(s.n_vars, UINT32_MAX), wfree;
        uint32_t nws = 0;
        auto new_slot = [&]() { if (!wfree.empty()) { uint32_t v = wfree.back(); wfree.pop_back(); return v; } return nws++; };
        const uint32_t NOSLOT = 0xffff;
        std::vector<std::vector<uint32_t>> wdying(n_levels);
        std::vector<uint16_t> pro, cyc;
        uint32_t wdiscard = UINT32_MAX;
        auto dst_slot = [&](uint32_t ov, uint32_t lv, bool pinned) -> uint16_t {
            if (last_level[ov] < 0) { if (wdiscard == UINT32_MAX) wdiscard = nws++; return (uint16_t)wdiscard; }
            const uint32_t sl = pinned ? nws++ : new_slot();
            wslot[ov] = sl;
            if (!pinned && last_level[ov] != INF) wdying[(size_t)std::max<int64_t>(last_level[ov], lv)].push_back(sl);
            return (uint16_t)sl;
        };
        auto src = [&](const Operand& in) -> uint16_t {
            if (in.kind != Operand::VAR || wslot[in.idx] == UINT32_MAX) { fail_(1); return 0; }
            return (uint16_t)wslot[in.idx];
        };
        auto emit_segment = [&](std::vector<uint16_t>& out, uint16_t kind, const std::vector<uint32_t>& ops, uint32_t lv) {
            const uint32_t rw = rec_words(kind);
            out.push_back(kind); out.push_back((uint16_t)ops.size()); out.push_back(0); out.push_back(0);
            for (uint32_t oi : ops) {
                const OpRec& op = sops[oi];
                std::vector<uint16_t> r;
                auto outs = [&](size_t first = 0) { for (size_t i = first; i < op.outs.size(); ++i) r.push_back(dst_slot(op.outs[i], lv, kind == WK_CONST)); };
                switch (kind) {
                case WK_CONST: {
                    outs();
                    const Operand& in = op.ins[0];
                    const uint32_t idx = in.kind == Operand::OUTER_VAR ? outer_.var_slot[in.idx] : in.idx;
                    if (in.kind != Operand::OUTER_VAR && in.kind != Operand::CONSTPOOL) fail_(2);
                    r.push_back((uint16_t)idx); r.push_back((uint16_t)(idx >> 16)); r.push_back(in.kind == Operand::OUTER_VAR ? 1 : 0);
                } break;
                case WK_INPUT: outs(); r.push_back((uint16_t)op.ins[0].idx); if (op.ins[0].idx > 0xfffe) fail_(3); break;
                case WK_SELECT: for (auto& in : op.ins) r.push_back(src(in)); outs(); break;
                case WK_FMA: case WK_LC4: {
                    const size_t nc = kind == WK_FMA ? 2 : 4;
                    for (size_t i = 0; i < nc; ++i) { if (op.ins[i].kind != Operand::CONSTPOOL || op.ins[i].idx > 0xfffe) fail_(4); r.push_back((uint16_t)op.ins[i].idx); }
                    for (size_t i = nc; i < op.ins.size(); ++i) r.push_back(src(op.ins[i]));
                    outs();
                } break;
                case WK_ISZERO:
                    r.push_back(src(op.ins[0]));
                    outs();
                    r.push_back(last_level[op.outs[1]] < 0 ? 0 : 1);  // the inverse (a gate witness) only if the cone reads it
                    break;
                case WK_DOT4: case WK_U32MULADD: case WK_U256MUL: case WK_U256DIV:
                    for (auto& in : op.ins) r.push_back(src(in));
                    outs();
                    break;
                case WK_UADD: case WK_USUB: r.push_back(op.a); for (auto& in : op.ins) r.push_back(src(in)); outs(); break;
                case WK_DIVREM: r.push_back(op.b); r.push_back(src(op.ins[0])); outs(); break;
                case WK_SPLIT_S: case WK_SPLIT_L: r.push_back(op.a); r.push_back(op.b); r.push_back(src(op.ins[0])); outs(); break;
                case WK_LOOKUP: {
                    if (op.ins.size() > 4 || op.outs.size() > 3 || op.ins[0].idx > 0xfffe) { fail_(5); break; }
                    r.push_back((uint16_t)op.ins[0].idx);
                    r.push_back((uint16_t)((op.ins.size() - 1) | (op.outs.size() << 8)));
                    for (size_t i = 1; i < 4; ++i) r.push_back(i < op.ins.size() ? src(op.ins[i]) : NOSLOT);
                    for (size_t i = 0; i < 3; ++i) r.push_back(i < op.outs.size() ? dst_slot(op.outs[i], lv, false) : NOSLOT);
                } break;
                case WK_P2:
                    for (auto& in : op.ins) r.push_back(src(in));
                    outs(op.outs.size() - 12);  // P2_ROUNDS collapses to its 12 outputs
                    break;
                default: fail_(6);
                }
                if (r.size() > rw) fail_(7);
                r.resize(rw, 0);
                out.insert(out.end(), r.begin(), r.end());
            }
        };
        for (uint32_t lv = 0; lv < n_levels && ok; ++lv) {
            std::map<uint16_t, std::vector<uint32_t>> by_kind;
            for (uint32_t oi : by_level[lv]) {
                const uint16_t k = kind_of(sops[oi]);
                if (!k) { fail_(8); break; }
                by_kind[k].push_back(oi);
            }
            for (auto& kv : by_kind) {
                if (kv.first == WK_CONST) emit_segment(pro, kv.first, kv.second, lv);   // loop-invariant: once, pinned slots
                else
                    for (size_t i0 = 0; i0 < kv.second.size(); i0 += 0xfff0) {
                        std::vector<uint32_t> part(kv.second.begin() + i0, kv.second.begin() + std::min(kv.second.size(), i0 + 0xfff0));
                        emit_segment(cyc, kv.first, part, lv);
                    }
            }
            for (uint32_t sl : wdying[lv]) wfree.push_back(sl);
        }
        for (size_t i = 0; i < carries_store_.size() && ok; ++i) {
            Carry cw = carries_store_[i];
            if (wslot[out_vars[i]] == UINT32_MAX) { fail_(9); break; }
            cw.out_cell = wslot[out_vars[i]];
            seed_wcarries_.push_back(cw);
        }
        if (ok && nws < 0xfff0) {
            for (uint16_t w : {(uint16_t)WK_END, (uint16_t)0, (uint16_t)0, (uint16_t)0}) pro.push_back(w);
            for (uint16_t w : {(uint16_t)WK_END, (uint16_t)0, (uint16_t)0, (uint16_t)0}) cyc.push_back(w);
            seed_wpro_words_ = (uint32_t)pro.size();
            seed_wprog_ = pro;
            seed_wprog_.insert(seed_wprog_.end(), cyc.begin(), cyc.end());
            seed_wslots_ = nws;
            if (!zkdev::seed_wave_fits((uint32_t)seed_wprog_.size(), nws, s.n_input_words)) { seed_wprog_.clear(); seed_wcarries_.clear(); }
        } else { seed_wcarries_.clear(); }
        if (getenv("ZKGL_PROG_STATS"))
            fprintf(stderr, "[zkgl] seed wave program: %zu u16 words (%u prologue; built %zu + %zu, lane forms %s (%d)), %u slots%s\n", seed_wprog_.size(), seed_wpro_words_,
                    pro.size(), cyc.size(), ok ? "ok" : "MISSING", fail_reason, nws, seed_wprog_.empty() ? " - not usable, strand kernel stays" : "");
    }
    auto cost = [&](uint32_t oi) -> uint64_t {
        const OpRec& op = sops[oi];
        uint64_t c = 8 + op.ins.size() + 2 * op.outs.size();
        if (op.opcode == ZK_OP_P2_ROUNDS || op.opcode == ZK_OP_POSEIDON2) c = 4000;
        if (op.opcode == ZK_OP_NN_MULMOD) c += 2000;
        if (op.opcode == ZK_OP_U256_DIVREM) c += 2500;
        if (op.opcode == ZK_OP_KECCAK_ABSORB) c = 20000;
        if (op.opcode == ZK_OP_SHA256_COMPRESS) c = 6000;
        return c;
    };
    std::fill(slot_of.begin(), slot_of.end(), UINT32_MAX);
    free_slots.clear();
    uint32_t ns = 0;
    std::vector<std::vector<uint32_t>> strand(NS);
    std::vector<std::vector<uint32_t>> dying(n_levels);  // slots whose last reader sits at this level
    uint64_t total = 0, critical = 0;
    for (uint32_t lv = 0; lv < n_levels; ++lv) {
        auto& ops = by_level[lv];
        std::stable_sort(ops.begin(), ops.end(), [&](uint32_t a, uint32_t b) { return cost(a) > cost(b); });
        uint64_t load[NS] = {0};
        for (uint32_t oi : ops) {
            uint32_t best = 0;
            for (uint32_t k = 1; k < NS; ++k) if (load[k] < load[best]) best = k;
            load[best] += cost(oi);
            const OpRec& op = sops[oi];
            const bool collapse = op.opcode == ZK_OP_P2_ROUNDS;
            auto& out = strand[best];
            out.push_back((collapse ? (uint32_t)ZK_OP_POSEIDON2 : (uint32_t)op.opcode) | ((uint32_t)op.a << 8) | ((uint32_t)op.b << 16));
            for (auto& in : op.ins) {
                switch (in.kind) {
                case Operand::VAR: out.push_back(slot_of[in.idx]); break;
                case Operand::CONSTPOOL: out.push_back(ZK_OPERAND_CONST | in.idx); break;
                case Operand::OUTER_VAR: out.push_back(ZK_OPERAND_OUTER | outer_.var_slot[in.idx]); break;
                case Operand::RAW: out.push_back(in.idx); break;
                }
            }
            for (size_t i = collapse ? op.outs.size() - 12 : 0; i < op.outs.size(); ++i) {
                const uint32_t ov = op.outs[i];
                if (last_level[ov] < 0) { out.push_back(DISCARD_MARK); continue; }
                uint32_t sl;
                if (!free_slots.empty()) { sl = free_slots.back(); free_slots.pop_back(); }
                else sl = ns++;
                slot_of[ov] = sl;
                out.push_back(sl);
                if (last_level[ov] != INF) dying[(size_t)std::max<int64_t>(last_level[ov], lv)].push_back(sl);
            }
        }
        for (uint32_t k = 0; k < NS; ++k) total += load[k];
        critical += *std::max_element(load, load + NS) + 40;  // LDS barrier: no store drain
        for (uint32_t sl : dying[lv]) free_slots.push_back(sl);  // free from the next level on
        if (lv + 1 < n_levels) for (auto& st : strand) st.push_back(ZK_OP_BARRIER);
    }
    const uint32_t sdiscard = ns++;
    if (ns + s.n_input_words > zkdev::seed_cone_max_slots()) return;
    for (uint32_t k = 0; k < NS; ++k) {
        seed_sbegin_[k] = (uint32_t)seed_sprog_.size();
        for (uint32_t w : strand[k]) seed_sprog_.push_back(w == DISCARD_MARK ? sdiscard : w);
        seed_send_[k] = (uint32_t)seed_sprog_.size();
    }
    for (size_t i = 0; i < carries_store_.size(); ++i) {
        Carry c = carries_store_[i];
        if (slot_of[out_vars[i]] == UINT32_MAX) { seed_sprog_.clear(); return; }
        c.out_cell = slot_of[out_vars[i]];
        seed_scarries_.push_back(c);
    }
    seed_sslots_ = ns;
    seed_sgain_ = critical ? (float)total / (float)critical : 0.f;
    if (getenv("ZKGL_PROG_STATS"))
    {
        fprintf(stderr, "[zkgl] seed cone: %u ops, %u levels, %u slots (plain %u), estimated gain %.2f\n", seed_ops_, n_levels, ns, n_slots, seed_sgain_);
        std::map<uint32_t, uint32_t> hist;
        for (size_t oi = 0; oi < sops.size(); ++oi) if (keep[oi]) hist[sops[oi].opcode]++;
        for (auto& kv : hist) fprintf(stderr, "   cone op %2u: %u\n", kv.first, kv.second);
        uint32_t widest = 0; for (auto& l : by_level) widest = std::max<uint32_t>(widest, (uint32_t)l.size());
        fprintf(stderr, "   widest level %u ops\n", widest);
    }
}

void CS::upload_scope(Scope& s) {
    {
        std::vector<uint32_t> padded(s.prog);  // the device keeps a 128-word prefetch window (ProgWindow)
        padded.resize(((padded.size() + 63) / 64) * 64 + 192, 0);
        s.d_prog = upload(padded);
    }
    {
        std::vector<uint32_t> padded(s.prog2);  // one 16-word scalar fetch per op, the last op's fetch runs past the end
        padded.resize(((padded.size() + 63) / 64) * 64 + 64, 0);
        s.d_prog2 = upload(padded);
    }
    if (!s.cprog.empty()) {
        std::vector<uint32_t> padded(s.cprog);  // a packet is read with one or two 16-word scalar fetches
        padded.resize(((padded.size() + 63) / 64) * 64 + 64, 0);
        s.d_cprog = upload(padded);
        s.d_cchunks = upload(s.cchunks);
    }
    if (!s.cmacros.empty()) {
        std::vector<uint32_t> padded(s.cmacros);
        padded.resize(padded.size() + 16, 0);   // 16-word scalar fetch of a 14-word descriptor
        s.d_cmacros = upload(padded);
    }
    if (!s.cprog_full.empty()) {
        std::vector<uint32_t> padded(s.cprog_full);
        padded.resize(((padded.size() + 63) / 64) * 64 + 64, 0);
        s.d_cprog_full = upload(padded);
        s.d_cchunks_full = upload(s.cchunks_full);
    }
    if (!s.cprog_fused.empty()) {
        std::vector<uint32_t> padded(s.cprog_fused);
        padded.resize(((padded.size() + 63) / 64) * 64 + 64, 0);
        s.d_cprog_fused = upload(padded);
        s.d_cchunks_fused = upload(s.cchunks_fused);
    }
    if (!s.sprog.empty()) {
        std::vector<uint32_t> padded(s.sprog);
        padded.resize(((padded.size() + 63) / 64) * 64 + 192, 0);
        s.d_sprog = upload(padded);
    }
    if (!s.sprog_n.empty()) {
        std::vector<uint32_t> padded(s.sprog_n);
        padded.resize(((padded.size() + 63) / 64) * 64 + 192, 0);
        s.d_sprog_n = upload(padded);
    }
    if (s.narrow_ok) {   // (already padded: emit_scope, build_narrow_check_program)
        s.d_prog2n = upload(s.prog2n);
        std::vector<uint32_t> padded(s.cprog_fused_n);
        padded.resize(((padded.size() + 63) / 64) * 64 + 64, 0);
        s.d_cprog_fused_n = upload(padded);
        s.d_slot_aw = upload(s.slot_aw);
    }
    if (!s.mult_sites.empty()) s.d_mult_sites = upload(s.mult_sites);
    s.d_consts = upload(s.const_pool);
    s.d_rows = upload(s.rows);
    s.d_rowconsts = upload(s.rowconsts);
    s.d_lrows = upload(s.lrows);
    s.d_copies = upload(s.copies);
    s.d_alias = upload(s.alias);
    s.d_mat_pairs = upload(s.mat_pairs);
}

void CS::finalize() {
    if (finalized_) throw ZkError(ZK_ERR_INVALID, "already finalized");
    if (in_loop_) throw ZkError(ZK_ERR_INVALID, "finalize inside loop scope");
    if (macro_window_op_ >= 0) throw ZkError(ZK_ERR_INVALID, "finalize inside a macro-op's gadget window");
    if (!loop_done_) { limit_ = 0; outer_.pre_ops = outer_.side_ops = outer_.ops.size(); }
    place_scope(outer_);
    place_scope(loop_);
    // pre-phase outer vars imported by the loop must be produced before the loop: verified by op order
    if (loop_done_) {
        std::vector<uint8_t> pre_defined(outer_.n_vars, 0);
        for (size_t oi = 0; oi < std::min(outer_.pre_ops, outer_.ops.size()); ++oi)
            for (uint32_t ov : outer_.ops[oi].outs) pre_defined[ov] = 1;
        for (auto& op : loop_.ops)
            for (auto& in : op.ins)
                if (in.kind == Operand::OUTER_VAR && !pre_defined[in.idx])
                    throw ZkError(ZK_ERR_UNRESOLVED, "loop imports an outer variable that is produced after the loop");
    }
    loop_ops_recorded_ = loop_.ops;
    schedule_loop_ops();
    assign_store_slots(outer_);
    assign_store_slots(loop_);
    bound_values(outer_);     // (from the gates, lookups and tables alone: before the programs, the narrow layout wants the classes)
    bound_values(loop_);
    build_narrow_layout(loop_);   // (slot -> address word; its programs come out of emit_scope / build_narrow_check_program below)
    emit_scope(outer_);
    emit_scope(loop_);
    build_check_program(outer_);
    build_check_program(loop_);
    build_narrow_check_program(loop_);
    build_mult_sites(outer_);
    build_mult_sites(loop_);
    build_strands(outer_);
    if (limit_) {
        build_strands(loop_);
        // the narrow form only where the full form would be used at all (wide op graphs: the hash circuits)
        if (loop_.s_gain[0] >= 3.2f && zkdev::STRANDS_PER_TILE > NARROW_STRANDS) build_strands(loop_, NARROW_STRANDS, true);
    }
    if (getenv("ZKGL_VERIFY_DEVICE_PROGRAMS")) {
        verify_device_programs(outer_);
        if (limit_) verify_device_programs(loop_);
    }
    uint64_t rows = (uint64_t)loop_.n_slots * limit_ + outer_.n_slots;
    if (!loop_done_) rows = outer_.n_slots;
    if (rows > max_trace_len_) throw ZkError(ZK_ERR_CAPACITY, "trace rows exceed max_trace_len");
    // carried input words (for the sequential seeding mode): CARRY link whose `in` side is an INPUT
    carries_.clear(); carries_store_.clear();
    for (auto& l : links_raw_) {
        if (l.kind != ZK_LINK_CARRY) continue;
        auto it = loop_.input_word.find(l.loop_cell);
        if (it == loop_.input_word.end()) continue;  // not stream-fed: nothing to seed
        Carry c{it->second, loop_.var_cells[l.other_cell][0], 0, 0}, cs{it->second, loop_.var_slot[l.other_cell], 0, 0};
        for (auto& f : links_raw_)
            if (f.kind == ZK_LINK_FIRST && f.loop_cell == l.loop_cell) {
                c.first_outer_cell = outer_.var_cells[f.other_cell][0];
                cs.first_outer_cell = outer_.var_slot[f.other_cell];
                c.has_first = cs.has_first = 1;
            }
        carries_.push_back(c);
        carries_store_.push_back(cs);
    }
    build_seed_program();
    // links / stream links: variables -> trace cells (export, materialised-trace check) and -> store slots (compact check)
    links_.clear(); links_store_.clear();
    for (auto& l : links_raw_) {
        zk_link r;
        r.kind = l.kind; r.pad = 0;
        const Scope& other = l.kind == ZK_LINK_CARRY ? loop_ : outer_;
        r.loop_cell = loop_.var_cells[l.loop_cell][0];
        r.other_cell = other.var_cells[l.other_cell][0];
        links_.push_back(r);
        r.loop_cell = loop_.var_slot[l.loop_cell];
        r.other_cell = other.var_slot[l.other_cell];
        links_store_.push_back(r);
    }
    streams_.clear(); streams_store_.clear();
    for (auto& sr : streams_raw_) {
        StreamRec r, rs;
        r.n_total = rs.n_total = sr.n_total;
        for (auto v : sr.a) { r.a.push_back(loop_.var_cells[v][0]); rs.a.push_back(loop_.var_slot[v]); }
        for (auto v : sr.b) { r.b.push_back(loop_.var_cells[v][0]); rs.b.push_back(loop_.var_slot[v]); }
        streams_.push_back(std::move(r));
        streams_store_.push_back(std::move(rs));
    }
    // ZK_CHECK_FUSED_DEFER_P2 leaves the 950 intermediates of every in-circuit permutation unwritten during the step: allowed only when
    // nothing of the step reads one — no witness op, lookup tuple, link, stream link, or gate the fused check program keeps (ADVICE r4:
    // ZK_OP_P2_ROUNDS is recordable through zk_cs_emit_op, so "every op has a verified descriptor" does not imply "nobody looks")
    loop_.p2_intermediates_private = true;
    if (limit_ && loop_.n_p2_rounds_ops) {
        std::vector<uint8_t> inter(loop_.n_vars, 0);
        for (auto& op : loop_.ops)
            if (!op.seed_only && op.opcode == ZK_OP_P2_ROUNDS && op.outs.size() > 12)
                for (size_t q = 0; q + 12 < op.outs.size(); ++q) inter[op.outs[q]] = 1;
        bool read = false;
        for (auto& op : loop_.ops) if (!op.seed_only) for (auto& in : op.ins) if (in.kind == Operand::VAR && inter[in.idx]) read = true;   // (seed hints run before the step, on their own values)
        for (auto& l : loop_.lookups) for (uint32_t v : l.vars) if (inter[v]) read = true;
        for (auto& l : links_raw_) { if (inter[l.loop_cell]) read = true; if (l.kind == ZK_LINK_CARRY && inter[l.other_cell]) read = true; }
        for (auto& sr : streams_raw_) { for (auto v : sr.a) if (inter[v]) read = true; for (auto v : sr.b) if (inter[v]) read = true; }
        for (size_t gi = 0; gi < loop_.gates.size(); ++gi)
            if (!loop_.gate_mirrored[gi]) for (uint32_t v : loop_.gates[gi].vars) if (inter[v]) read = true;
        loop_.p2_intermediates_private = !read;
    }
    // tables
    std::vector<zk_table_desc> tdesc(tables_.size() + 1);
    std::memset(tdesc.data(), 0, tdesc.size() * sizeof(zk_table_desc));
    std::vector<uint64_t> words;
    total_table_rows_ = 0;
    for (size_t i = 0; i < tables_.size(); ++i) {
        TableRec& t = tables_[i];
        t.word_off = (uint32_t)words.size();
        t.mult_off = total_table_rows_;
        words.insert(words.end(), t.rows.begin(), t.rows.end());
        total_table_rows_ += t.n_rows;
        zk_table_desc& d = tdesc[i + 1];
        d.word_off = t.word_off; d.mult_off = t.mult_off; d.n_rows = t.n_rows; d.n_keys = t.n_keys; d.n_vals = t.n_vals;
        d.dense = t.dense ? 1 : 0;
        for (int k = 0; k < 3; ++k) d.key_shift[k] = t.key_shift[k];
    }
    // Device-only packed copies of the dense byte-valued tables (xor8 / and8 / andn8 / byte splits ...): one byte per value,
    // appended behind the rows; zk_table_desc.dense = 1 | 2 | (first word of the packed copy << 2).  A lookup then gathers
    // from a 64 KB array instead of a 1.5 MB one, and dense tables need no key words at all (row index == packed key).
    for (size_t i = 0; i < tables_.size(); ++i) {
        const TableRec& t = tables_[i];
        if (!t.dense || !t.byte_valued) continue;
        const uint32_t w = t.n_keys + t.n_vals;
        const size_t first = words.size();
        if (first >= (1u << 29)) break;
        std::vector<uint8_t> bytes((size_t)t.n_rows * t.n_vals);
        for (uint32_t r = 0; r < t.n_rows; ++r)
            for (uint32_t k = 0; k < t.n_vals; ++k) bytes[(size_t)r * t.n_vals + k] = (uint8_t)t.rows[(size_t)r * w + t.n_keys + k];
        words.resize(first + (bytes.size() + 7) / 8, 0);
        std::memcpy(&words[first], bytes.data(), bytes.size());
        tdesc[i + 1].dense = 1u | 2u | ((uint32_t)first << 2);
    }
    tdesc_host_ = tdesc;
    table_words_host_ = words;
    finalized_ = true;
}

// ------------------------------------------------------------------ execution
// device upload (fails loudly without a GPU: the product path has no CPU fallback)
void CS::ensure_uploaded() {
    if (uploaded_) return;
    upload_scope(outer_);
    upload_scope(loop_);
    d_tables_ = upload(tdesc_host_);
    d_table_words_ = upload(table_words_host_);
    d_links_ = upload(links_);
    d_links_store_ = upload(links_store_);
    if (loop_.narrow_ok) {
        loop_last_slots_.clear();
        for (auto& op : outer_.ops)
            if (!op.seed_only && op.opcode == ZK_OP_LOOP_LAST && !op.ins.empty()) loop_last_slots_.push_back(loop_.var_slot[op.ins[0].idx]);
        std::sort(loop_last_slots_.begin(), loop_last_slots_.end());   // (two ZK_OP_LOOP_LAST ops may read one variable: one thread per cell in k_widen_last —
        loop_last_slots_.erase(std::unique(loop_last_slots_.begin(), loop_last_slots_.end()), loop_last_slots_.end());   //  the emulated race detector found the duplicate store)
        if (!loop_last_slots_.empty()) d_loop_last_slots_ = upload(loop_last_slots_);
    }
    if (loop_.narrow_ok) {   // the loop-scope endpoints as address words of the narrow store (outer-scope endpoints stay slots)
        std::vector<zk_link> ln(links_store_);
        for (auto& l : ln) { l.loop_cell = loop_.slot_aw[l.loop_cell]; if (l.kind == ZK_LINK_CARRY) l.other_cell = loop_.slot_aw[l.other_cell]; }
        d_links_store_n_ = upload(ln);
        for (auto& sr : streams_store_) {
            std::vector<uint32_t> cells;
            for (uint32_t c : sr.a) cells.push_back(loop_.slot_aw[c]);
            for (uint32_t c : sr.b) cells.push_back(loop_.slot_aw[c]);
            d_streams_store_n_.push_back(upload(cells));
        }
    }
    for (int k = 0; k < 2; ++k)
        for (auto& sr : (k ? streams_store_ : streams_)) {
            std::vector<uint32_t> cells(sr.a);
            cells.insert(cells.end(), sr.b.begin(), sr.b.end());
            (k ? d_streams_store_ : d_streams_).push_back(upload(cells));
        }
    d_carries_ = (void*)upload(carries_store_);
    if (!seed_prog_.empty()) {
        std::vector<uint32_t> padded(seed_prog_);
        padded.resize(((padded.size() + 63) / 64) * 64 + 192, 0);
        d_seed_prog_ = upload(padded);
        d_seed_carries_ = (void*)upload(seed_carries_);
    }
    if (!seed_wprog_.empty()) {
        std::vector<uint16_t> padded(seed_wprog_);
        padded.resize((padded.size() + 63) / 64 * 64 + 64, 0);
        d_seed_wprog_ = upload(padded);
        d_seed_wcarries_ = (void*)upload(seed_wcarries_);
    }
    if (!seed_sprog_.empty()) {
        std::vector<uint32_t> padded(seed_sprog_);
        padded.resize(((padded.size() + 63) / 64) * 64 + 192, 0);
        d_seed_sprog_ = upload(padded);
        d_seed_scarries_ = (void*)upload(seed_scarries_);
    }
    hip_check(hipMalloc((void**)&d_fail_, 16 * sizeof(unsigned long long)), "hipMalloc fail words");   // 0..5 failure keys, 6..7 clock probe, 8..9 permutation skip counters
    for (auto& e : ev_) {
        hipEvent_t he;
        hip_check(hipEventCreate(&he), "hipEventCreate");
        e = (void*)he;
    }
    uploaded_ = true;
}

void CS::set_batch(uint32_t n) {
    if (!finalized_) throw ZkError(ZK_ERR_INVALID, "set_batch before finalize");
    if (n == 0) throw ZkError(ZK_ERR_INVALID, "batch must be > 0");
    if (uses_bytebuf_macro_ && uses_sha4_macro_) throw ZkError(ZK_ERR_INVALID, "a circuit that records both ZK_OP_BYTEBUF_FILL and the 4-bit-chunk SHA-256 macro-op: no kernel carries both backends");
    ensure_uploaded();
    auto alloc_cells = [&](Scope& s, uint64_t lanes) {
        if (s.d_store) { hipFree(s.d_store); s.d_store = nullptr; }
        if (s.d_cells) { hipFree(s.d_cells); s.d_cells = nullptr; }  // the materialised trace is re-allocated on demand
        s.n_lanes = (uint32_t)lanes;
        s.stride = (lanes + 63) / 64 * 64;  // the materialised trace: whole 64-lane tiles
        // Lane tiling of the variable store (store_geom.hpp): 64-lane tiles (one per wavefront) unless ZKGL_STORE_TILE_LOG2=7..12 asks
        // for wider ones for the loop scope.  Wide tiles (a value = up to 32 KB contiguous, shared by 64 wavefronts) stream 5-8 % faster
        // in the bare store pattern (profiles/r3_layout_probe.jsonl) but NOT in the real kernel: 40.4-40.8 ms against 39.3-40.3 ms for
        // k_witness_loop at B = 384, same box, four fresh processes each (profiles/r3_loop_probe.md §1).  The switch stays for such A/B runs; the buffer-addressed kernels
        // need slot << (T + 3) < 2^32.
        s.store_tile_log2 = zkgeom::WAVE_TILE_LOG2;
        if (s.is_loop) {
            uint32_t t = zkgeom::WAVE_TILE_LOG2;
            if (const char* e = std::getenv("ZKGL_STORE_TILE_LOG2")) t = std::min<uint32_t>(zkgeom::WIDE_TILE_LOG2, std::max<uint32_t>(zkgeom::WAVE_TILE_LOG2, (uint32_t)std::atoi(e)));
            while (t > zkgeom::WAVE_TILE_LOG2 && (uint64_t)s.n_store >= (1ull << (29 - t))) --t;
            s.store_tile_log2 = t;
        }
        const uint64_t store_lanes = zkgeom::padded_lanes(s.store_geom(), lanes);
        size_t bytes = std::max<size_t>((size_t)s.n_store * store_lanes * 8, 8);
        hip_check(hipMalloc((void**)&s.d_store, bytes), "hipMalloc variable store");
        hip_check(hipMemset(s.d_store, 0, bytes), "hipMemset variable store");
        if (std::getenv("ZKGL_PROG_STATS")) fprintf(stderr, "[zkgl] %s store at %p (%zu bytes, tiles of %u lanes)\n", s.is_loop ? "loop" : "outer", (void*)s.d_store, bytes, 1u << s.store_tile_log2);
    };
    uint64_t loop_lanes = (uint64_t)n * limit_;
    if (loop_lanes >= 0xffffffffull) throw ZkError(ZK_ERR_CAPACITY, "batch*limit exceeds 32-bit lane index");
    alloc_cells(outer_, n);
    if (loop_.d_store_n) { hipFree(loop_.d_store_n); loop_.d_store_n = nullptr; }
    narrow_active_ = false; narrow_pending_ = false;
    // Narrow store: this batch's fused steps write the loop scope's values into the narrow store when the loop launch is the plain kernel (not the
    // strand form, not 64-bit addressing) and the multiplicities come from its inline atomics (the k_multiplicities pass reads the ordinary store).
    // The ordinary store is allocated beside it when it fits (every reader outside the fused step sees the widened copy there), on first use otherwise.
    const char* ne = std::getenv("ZKGL_NARROW_STORE");
    const bool want_narrow = ne && ne[0] == '1' && narrow_enabled_ && loop_.narrow_ok && limit_ && inline_multiplicities() && !loop_runs_strands(loop_, 0, (uint32_t)loop_lanes);
    if (want_narrow) {
        loop_.n_lanes = (uint32_t)loop_lanes;
        loop_.store_tile_log2 = zkgeom::WAVE_TILE_LOG2;
        if (const char* e = std::getenv("ZKGL_STORE_TILE_LOG2")) {
            uint32_t t = std::min<uint32_t>(zkgeom::WIDE_TILE_LOG2, std::max<uint32_t>(zkgeom::WAVE_TILE_LOG2, (uint32_t)std::atoi(e)));
            while (t > zkgeom::WAVE_TILE_LOG2 && (uint64_t)loop_.n_store >= (1ull << (29 - t))) --t;
            loop_.store_tile_log2 = t;
        }
        const uint64_t n8 = zkgeom::slots(loop_.narrow_geom());
        if (n8 < (1ull << (29 - loop_.store_tile_log2))) {
            const size_t nbytes = std::max<size_t>((size_t)n8 * zkgeom::padded_lanes(loop_.narrow_geom(), loop_lanes) * 8, 8);
            if (hipMalloc((void**)&loop_.d_store_n, nbytes) == hipSuccess) {
                hip_check(hipMemset(loop_.d_store_n, 0, nbytes), "hipMemset narrow store");
                narrow_active_ = true;
            } else { loop_.d_store_n = nullptr; (void)hipGetLastError(); }
        }
    }
    if (narrow_active_) {
        // the ordinary store: beside the narrow one when it fits, else on first use (ensure_wide_store)
        try { alloc_cells(loop_, loop_lanes); } catch (const ZkError&) { loop_.d_store = nullptr; (void)hipGetLastError(); }
        if (!loop_.d_store && !loop_last_slots_.empty()) {   // the outer post phase needs the ordinary store in every step: no room for both, no narrow store
            hipFree(loop_.d_store_n); loop_.d_store_n = nullptr; narrow_active_ = false;
            alloc_cells(loop_, loop_lanes);
        }
    } else alloc_cells(loop_, loop_lanes);
    if (d_mult_) { hipFree(d_mult_); d_mult_ = nullptr; }
    size_t mbytes = std::max<size_t>((size_t)n * total_table_rows_ * 4, 4);
    hip_check(hipMalloc((void**)&d_mult_, mbytes), "hipMalloc multiplicities");
    batch_ = n;
    compact_ = true;
    outer_.d_inputs = nullptr; loop_.d_inputs = nullptr;
    outer_.bound_input_words = loop_.bound_input_words = 0;
    outer_.input_stride = loop_.input_stride = 0;
}

void CS::bind_inputs(bool loop_scope, const uint64_t* dev_words, uint32_t n_words, uint64_t lane_stride) {
    if (batch_ == 0) throw ZkError(ZK_ERR_INVALID, "bind_inputs before set_batch");
    Scope& s = loop_scope ? loop_ : outer_;
    if (n_words < s.n_input_words) throw ZkError(ZK_ERR_INVALID, "bind_inputs: fewer words than the circuit reads");
    if (lane_stride && lane_stride < s.n_lanes) throw ZkError(ZK_ERR_INVALID, "bind_inputs: lane stride shorter than the batch");
    s.d_inputs = dev_words;
    s.bound_input_words = n_words;
    s.input_stride = lane_stride;
}

static zkdev::ScopeArgs scope_args(const Scope& s, const Scope& outer, const Scope& loop, uint32_t limit,
                                   const zk_table_desc* tables, const uint64_t* words, uint32_t* mult, uint32_t total_rows) {
    zkdev::ScopeArgs a;
    a.prog = s.d_prog; a.n_words = (uint32_t)s.prog.size(); a.n_lanes = s.n_lanes; a.consts = s.d_consts;
    a.cells = s.d_store; a.n_cells = s.store_geom(); a.inputs = s.d_inputs;  // the witness kernels work on the variable store
    a.in_stride = s.input_stride ? s.input_stride : s.n_lanes;
    a.outer_cells = outer.d_store; a.outer_n_cells = outer.store_geom();
    a.limit = s.is_loop ? limit : 1; a.is_loop = s.is_loop ? 1 : 0;
    a.tables = tables; a.table_words = words; a.mult = mult; a.total_table_rows = total_rows;
    a.loop_cells = loop.d_store; a.loop_n_cells = loop.store_geom(); a.loop_limit = limit;
    a.uses_bigint = s.uses_bigint ? 1 : 0;
    return a;   // (a.xmacros: CS::scope_args_x below)
}

// the cone seeding launch over `n` instances: la = loop-scope arguments whose outer_cells hold the pre phase of those instances
uint32_t CS::layout_word(const char* scope, const char* name) const {
    const std::string key = std::string(scope) + " " + name + " ";
    size_t pos = 0;
    while (pos < input_layout.size()) {
        size_t eol = input_layout.find('\n', pos);
        if (eol == std::string::npos) eol = input_layout.size();
        if (input_layout.compare(pos, key.size(), key) == 0) return (uint32_t)std::strtoul(input_layout.c_str() + pos + key.size(), nullptr, 10);
        pos = eol + 1;
    }
    return UINT32_MAX;
}

void CS::set_seed_given(const uint32_t* loop_words, uint32_t n) {
    if (!finalized_) throw ZkError(ZK_ERR_INVALID, "set_seed_given before finalize");
    std::vector<uint32_t> w(loop_words, loop_words + n);
    for (uint32_t x : w) {
        bool carried = false;
        for (auto& c : carries_store_) carried |= c.word == x;
        if (!carried) throw ZkError(ZK_ERR_INVALID, "set_seed_given: not a loop-carried word");
    }
    seed_given_words_ = std::move(w);
}
bool CS::seed_words_given(const uint32_t* words, uint32_t n) const {
    for (uint32_t i = 0; i < n; ++i)
        if (std::find(seed_given_words_.begin(), seed_given_words_.end(), words[i]) == seed_given_words_.end()) return false;
    return true;
}

// a circuit's native seeder instead of the cone of the carried outputs: main_vm (kind 1): walker + Poseidon2 chains + fill
// (kernels_vm_seed.hpp); ram_permutation (kind 2) once the host has declared the queue heads given: scans (kernels_queue_seed.hpp)
bool CS::launch_seed_native(const zkdev::ScopeArgs& la, const zkdev::ScopeArgs& oa, uint64_t* dev_loop_inputs_rw, uint32_t n, void* stream) {
    const char* e = std::getenv("ZKGL_SEED_NATIVE");
    if (e && e[0] == '0') return false;
    {   // every loop-carried word declared given (a packer that had all the queue states in the witness: zk_pack_*_witness_tails):
        // nothing to derive — the links and the queue constraints of the circuit judge the words the host wrote
        bool all = !carries_store_.empty();
        for (auto& c : carries_store_)
            if (std::find(seed_given_words_.begin(), seed_given_words_.end(), c.word) == seed_given_words_.end()) { all = false; break; }
        if (all) return true;
    }
    if (native_seed_kind == 2) {
        uint32_t heads[24];
        for (uint32_t i = 0; i < 12; ++i) { heads[i] = 1 + i; heads[12 + i] = 14 + i; }
        if (!seed_words_given(heads, 24) || native_seed_outer_vars.size() != 16 || carries_store_.size() != 46) return false;
        if (!d_state0_slot_) {
            std::vector<uint32_t> slots(46, UINT32_MAX), ch;
            for (auto& c : carries_store_)
                if (c.word < 46 && c.has_first) slots[c.word] = c.first_outer_cell;
            for (uint32_t sl : slots)
                if (sl == UINT32_MAX) return false;
            for (zk_var v : native_seed_outer_vars) ch.push_back(outer_.var_slot[var_index(v)]);
            d_state0_slot_ = upload(slots);
            d_native_outer_slots_ = upload(ch);
        }
        zkdev::RamSeedArgs a;
        a.loop = dev_loop_inputs_rw; a.in_stride = la.in_stride; a.limit = limit_; a.n_instances = n;
        a.outer_store = la.outer_cells; a.outer_n_store = la.outer_n_cells; a.state0_slot = d_state0_slot_; a.ch_slot = d_native_outer_slots_;
        a.bootloader_heap_page = native_seed_param;
        dev_check(zkdev::launch_ram_seed(a, stream));
        return true;
    }
    if (native_seed_kind == 7) {   // eip_4844: native_seed_param = n_chunks; every carried word starts at zero (circuits/eip4844.cpp)
        const uint32_t n_chunks = native_seed_param;
        if (carries_store_.size() != 217 || loop_.n_input_words < 217 + 136 + 31 || (loop_.n_input_words - 217 - 136) % 31 || outer_.n_input_words != 64) return false;
        const uint32_t cpi = (loop_.n_input_words - 217 - 136) / 31;
        if (cpi > 8 || (uint64_t)cpi * limit_ < n_chunks) return false;
        zkdev::EipSeedArgs a;
        a.loop = dev_loop_inputs_rw; a.in_stride = la.in_stride; a.limit = limit_; a.n_instances = n; a.n_chunks = n_chunks; a.cpi = cpi;
        a.outer_inputs = oa.inputs; a.outer_in_stride = oa.in_stride;
        dev_check(zkdev::launch_eip4844_seed(a, stream));
        return true;
    }
    if (native_seed_kind == 5 || native_seed_kind == 6) {   // storage_validity / log_sorter: the host packer walked the integer state
        const bool storage = native_seed_kind == 5;
        const uint32_t n_carried = storage ? 67 : 57, acc = storage ? 2 : 1, tail = storage ? 17 : 15;
        if (carries_store_.size() != n_carried || native_seed_outer_vars.size() != (storage ? 41u : 40u)) return false;
        std::vector<uint32_t> need;
        for (uint32_t w = 0; w < n_carried; ++w)
            if (!(w >= acc && w < acc + 4) && !(w >= tail && w < tail + 4)) need.push_back(w);
        if (!seed_words_given(need.data(), (uint32_t)need.size())) return false;
        uint32_t tw[4] = {tail, tail + 1, tail + 2, tail + 3};
        const bool tail_given = seed_words_given(tw, 4);
        if (!d_state0_slot_) {
            std::vector<uint32_t> slots(n_carried, UINT32_MAX), ch;
            for (auto& c : carries_store_)
                if (c.word < n_carried && c.has_first) slots[c.word] = c.first_outer_cell;
            for (uint32_t sl : slots)
                if (sl == UINT32_MAX) return false;
            for (zk_var v : native_seed_outer_vars) ch.push_back(outer_.var_slot[var_index(v)]);
            d_state0_slot_ = upload(slots);
            d_native_outer_slots_ = upload(ch);
        }
        zkdev::LogqSeedArgs a;
        a.kind = storage ? 0 : 1; a.with_chain = tail_given ? 0 : 1;
        a.loop = dev_loop_inputs_rw; a.in_stride = la.in_stride; a.limit = limit_; a.n_instances = n;
        a.outer_store = la.outer_cells; a.outer_n_store = la.outer_n_cells; a.state0_slot = d_state0_slot_; a.ch_slot = d_native_outer_slots_;
        dev_check(zkdev::launch_logq_seed(a, stream));
        return true;
    }
    if (native_seed_kind == 8) {   // sort_decommittment_requests: all but the four accumulator words (1..4) from the host packer with tails
        if (carries_store_.size() != 65 || native_seed_outer_vars.size() != 16) return false;
        std::vector<uint32_t> need;
        for (uint32_t w = 0; w < 65; ++w)
            if (!(w >= 1 && w <= 4)) need.push_back(w);
        if (!seed_words_given(need.data(), (uint32_t)need.size())) return false;
        if (!d_state0_slot_) {
            std::vector<uint32_t> slots(65, UINT32_MAX), ch;
            for (auto& c : carries_store_)
                if (c.word < 65 && c.has_first) slots[c.word] = c.first_outer_cell;
            for (uint32_t sl : slots)
                if (sl == UINT32_MAX) return false;
            for (zk_var v : native_seed_outer_vars) ch.push_back(outer_.var_slot[var_index(v)]);
            d_state0_slot_ = upload(slots);
            d_native_outer_slots_ = upload(ch);
        }
        zkdev::LogqSeedArgs a;
        a.kind = 2; a.with_chain = 0;
        a.loop = dev_loop_inputs_rw; a.in_stride = la.in_stride; a.limit = limit_; a.n_instances = n;
        a.outer_store = la.outer_cells; a.outer_n_store = la.outer_n_cells; a.state0_slot = d_state0_slot_; a.ch_slot = d_native_outer_slots_;
        dev_check(zkdev::launch_logq_seed(a, stream));
        return true;
    }
    if (native_seed_kind == 3 || native_seed_kind == 4) {   // keccak256 / sha256 round function FSMs (kernels_fsm_seed.hpp)
        const uint32_t n_carried = native_seed_kind == 3 ? 423 : 60, n_words = native_seed_kind == 3 ? 507 : 112;
        if (carries_store_.size() != n_carried || loop_.n_input_words != n_words) return false;
        if (!d_state0_slot_) {
            std::vector<uint32_t> slots(n_carried, UINT32_MAX);
            for (auto& c : carries_store_)
                if (c.word < n_carried && c.has_first) slots[c.word] = c.first_outer_cell;
            for (uint32_t sl : slots)
                if (sl == UINT32_MAX) return false;
            d_state0_slot_ = upload(slots);
        }
        zkdev::FsmSeedArgs a;
        a.kind = native_seed_kind - 3;
        a.loop = dev_loop_inputs_rw; a.in_stride = la.in_stride; a.limit = limit_; a.n_instances = n;
        a.outer_store = la.outer_cells; a.outer_n_store = la.outer_n_cells; a.state0_slot = d_state0_slot_;
        dev_check(zkdev::launch_fsm_seed(a, stream));
        return true;
    }
    if (native_seed_kind != 1) return false;
    if (circuit_blob.size() != sizeof(zk_opcode_defs)) return false;
    zkdev::VmSeedArgs a;
    const char* names[14] = {"code_word", "src0_read_value", "src0_read_is_ptr", "log_pubdata_refund", "log_storage_read_value", "log_rollback_queue_prev_head",
                             "near_call_rollback_queue_tail", "far_call_code_hash_read_value", "far_call_decommit_suggested_page", "far_call_rollback_queue_tail",
                             "ret_popped_context", "ret_previous_callstack_state", "uma_read_a", "uma_read_b"};
    uint32_t* raw = &a.raw.code_word;
    for (int i = 0; i < 14; ++i) {
        raw[i] = layout_word("loop", names[i]);
        if (raw[i] == UINT32_MAX) return false;
    }
    a.w_zkporter = layout_word("outer", "zkporter_is_available");
    a.w_default_aa = layout_word("outer", "default_aa_code_hash");
    const uint32_t w_state = layout_word("loop", "state");
    if (a.w_zkporter == UINT32_MAX || a.w_default_aa == UINT32_MAX || w_state != 0) return false;
    hipStream_t st = (hipStream_t)stream;
    if (!d_native_blob_) {
        std::vector<uint32_t> slots(243, UINT32_MAX);
        for (auto& c : carries_store_)
            if (c.word < 243 && c.has_first) slots[c.word] = c.first_outer_cell;
        for (uint32_t sl : slots)
            if (sl == UINT32_MAX) return false;
        hip_check(hipMalloc(&d_native_blob_, circuit_blob.size()), "hipMalloc vm defs");
        hip_check(hipMemcpy(d_native_blob_, circuit_blob.data(), circuit_blob.size(), hipMemcpyHostToDevice), "copy vm defs");
        hip_check(hipMalloc((void**)&d_state0_slot_, slots.size() * 4), "hipMalloc state0 slots");
        hip_check(hipMemcpy(d_state0_slot_, slots.data(), slots.size() * 4, hipMemcpyHostToDevice), "copy state0 slots");
    }
    const size_t need = zkdev::vm_seed_scratch_bytes(limit_, n);
    if (need > native_scratch_bytes_) {
        if (d_native_scratch_) hipFree(d_native_scratch_);
        d_native_scratch_ = nullptr; native_scratch_bytes_ = 0;
        hip_check(hipMalloc((void**)&d_native_scratch_, need), "hipMalloc vm seed scratch");
        native_scratch_bytes_ = need;
    }
    a.defs_dev = d_native_blob_; a.defs_host = circuit_blob.data();
    a.loop = dev_loop_inputs_rw; a.in_stride = la.in_stride; a.limit = limit_; a.n_instances = n; a.n_loop_words = loop_.n_input_words;
    a.outer_store = la.outer_cells; a.outer_n_store = la.outer_n_cells; a.state0_slot = d_state0_slot_;
    a.outer_inputs = oa.inputs; a.outer_in_stride = oa.in_stride;
    a.scratch = d_native_scratch_;
    const bool timed = std::getenv("ZKGL_SEED_PHASE_MS") != nullptr;
    dev_check(zkdev::launch_vm_seed(a, st, timed ? last_seed_phase_ms : nullptr));
    return true;
}

void CS::launch_seed(const zkdev::ScopeArgs& la, const zkdev::ScopeArgs& oa, uint64_t* dev_loop_inputs_rw, uint32_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (launch_seed_native(la, oa, dev_loop_inputs_rw, n, stream)) return;
    const char* force_generic = std::getenv("ZKGL_SEED_GENERIC");
    const char* seed_strands = std::getenv("ZKGL_SEED_STRANDS");  // 0: plain cone, 1: strand form whenever it exists
    const bool generic = force_generic && force_generic[0] == '1';
    if (seed_cone_unsupported_ && !generic) throw ZkError(ZK_ERR_INVALID, "the seeding cone of this circuit cannot be run by the seed kernels (it contains ZK_OP_BYTEBUF_FILL, or a carried output depends on a gated ZK_OP_POSEIDON2 outside a select on its flag): use the native seeder or ZKGL_SEED_GENERIC=1");
    const bool use_strands = d_seed_sprog_ && !(seed_strands && seed_strands[0] == '0') && ((seed_strands && seed_strands[0] == '1') || seed_sgain_ >= 1.5f);
    if (d_seed_wprog_ && !generic)
        dev_check(zkdev::launch_seed_wave(la, d_seed_wprog_, (uint32_t)seed_wprog_.size(), seed_wpro_words_, seed_wslots_, loop_.n_input_words,
                                          (const zkdev::CarryArgs*)d_seed_wcarries_, (uint32_t)seed_wcarries_.size(), dev_loop_inputs_rw, n, st));
    else if (use_strands && !generic)
        dev_check(zkdev::launch_seed_cone_strands(la, d_seed_sprog_, seed_sbegin_, seed_send_, seed_sslots_, loop_.n_input_words,
                                                  (const zkdev::CarryArgs*)d_seed_scarries_, (uint32_t)seed_scarries_.size(), dev_loop_inputs_rw, n, seed_v2_ok_, st));
    else if (d_seed_prog_ && !generic)
        dev_check(zkdev::launch_seed_cone(la, d_seed_prog_, (uint32_t)seed_prog_.size(), seed_slots_, loop_.n_input_words, (const zkdev::CarryArgs*)d_seed_carries_,
                                          (uint32_t)seed_carries_.size(), dev_loop_inputs_rw, n, st));
    else {
        // the generic sequential interpreter is the round-1 (v1) form, which has no handler for the macro-ops: their circuits seed
        // through the native seeders / the cone (whose hints replace the macro-ops)
        if (uses_lookup_macros_) throw ZkError(ZK_ERR_INVALID, "generic sequential seeding (ZKGL_SEED_GENERIC) does not run circuits recorded with hash macro-ops; record with ZKGL_NO_HASH_MACROS=1 or use the cone / native seeder");
        if (!la.cells) throw ZkError(ZK_ERR_CAPACITY, "generic sequential seeding works in the ordinary loop store, which does not fit beside the narrow one at this batch size");   // (narrow batches: set_batch allocates both when they fit)
        dev_check(zkdev::launch_witness_seq(la, (const zkdev::CarryArgs*)d_carries_, (uint32_t)carries_store_.size(), dev_loop_inputs_rw,
                                            n, st));
    }
}

void CS::seed_carried_inputs(uint64_t* dev_loop_inputs_rw, void* stream) {
    if (batch_ == 0) throw ZkError(ZK_ERR_INVALID, "seed_carried_inputs before set_batch");
    if (!limit_) return;
    if (outer_.n_input_words && !outer_.d_inputs) throw ZkError(ZK_ERR_INVALID, "outer input stream not bound");
    if (!dev_loop_inputs_rw || loop_.d_inputs != dev_loop_inputs_rw)
        throw ZkError(ZK_ERR_INVALID, "seed_carried_inputs: pass the (writable) buffer bound as the loop input stream");
    hipStream_t st = (hipStream_t)stream;
    hip_check(hipMemsetAsync(d_mult_, 0, std::max<size_t>((size_t)batch_ * total_table_rows_ * 4, 4), st), "memset mult");
    auto oa = scope_args(outer_, outer_, loop_, limit_, d_tables_, d_table_words_, nullptr, total_table_rows_);
    auto la = scope_args(loop_, outer_, loop_, limit_, d_tables_, d_table_words_, nullptr, total_table_rows_);
    launch_phase(outer_, oa, 0, st);
    launch_seed(la, oa, dev_loop_inputs_rw, batch_, st);
    hip_check(hipStreamSynchronize(st), "seed sync");
}

// Seeds a STREAM of n instances (any n, independent of set_batch): the outer pre phase runs on a temporary outer store of n
// lanes, then the cone.  The chain of `limit` iterations is a latency bound per instance and the kernel keeps one small block
// per few instances, so a pass over ~1000 instances costs what a pass over 8 does; a host seeds a long stream once and then
// resolves it in windows (bind_inputs with lane_stride = the stream length).  Layouts: outer [word][n], loop [word][n * limit].
void CS::seed_stream(uint32_t n, const uint64_t* dev_outer_inputs, uint64_t* dev_loop_inputs_rw, void* stream, bool synchronize, uint64_t outer_stride,
                     uint64_t loop_stride) {
    if (!finalized_) throw ZkError(ZK_ERR_INVALID, "seed_stream before finalize");
    if (!limit_ || n == 0) return;
    ensure_uploaded();
    if (outer_.n_input_words && !dev_outer_inputs) throw ZkError(ZK_ERR_INVALID, "seed_stream: outer input stream missing");
    if (!dev_loop_inputs_rw) throw ZkError(ZK_ERR_INVALID, "seed_stream: loop input stream missing");
    if ((uint64_t)n * limit_ >= 0xffffffffull) throw ZkError(ZK_ERR_CAPACITY, "n*limit exceeds 32-bit lane index");
    hipStream_t st = (hipStream_t)stream;
    const uint64_t tiles = ((uint64_t)n + 63) / 64;
    // temporary outer store of the stream's lane count: kept across calls (a host seeds stream after stream of the same size)
    const size_t need = std::max<size_t>((size_t)outer_.n_store * tiles * 64 * 8, 8);
    if (need > seed_outer_bytes_) {
        if (d_seed_outer_) hipFree(d_seed_outer_);
        d_seed_outer_ = nullptr; seed_outer_bytes_ = 0;
        hip_check(hipMalloc((void**)&d_seed_outer_, need), "hipMalloc seed outer store");
        seed_outer_bytes_ = need;
    }
    // the scopes' own arguments with the stream's lane counts and buffers patched in (no copy of the Scope objects: their program
    // vectors are tens of MB for main_vm)
    auto oa = scope_args(outer_, outer_, loop_, limit_, d_tables_, d_table_words_, nullptr, total_table_rows_);
    auto la = scope_args(loop_, outer_, loop_, limit_, d_tables_, d_table_words_, nullptr, total_table_rows_);
    if ((outer_stride && outer_stride < n) || (loop_stride && loop_stride < (uint64_t)n * limit_)) throw ZkError(ZK_ERR_INVALID, "seed_stream: lane stride shorter than the window");
    oa.cells = d_seed_outer_; oa.n_lanes = n; oa.inputs = dev_outer_inputs; oa.in_stride = outer_stride ? outer_stride : n; oa.outer_cells = d_seed_outer_; oa.loop_cells = nullptr;
    la.cells = nullptr; la.n_lanes = (uint32_t)((uint64_t)n * limit_); la.inputs = dev_loop_inputs_rw; la.in_stride = loop_stride ? loop_stride : (uint64_t)n * limit_;
    la.outer_cells = d_seed_outer_; la.loop_cells = nullptr;
    launch_phase(outer_, oa, 0, st, n);
    launch_seed(la, oa, dev_loop_inputs_rw, n, st);
    if (synchronize) hip_check(hipStreamSynchronize(st), "seed sync");
}

void CS::resolve(void* stream) {
    if (batch_ == 0) throw ZkError(ZK_ERR_INVALID, "resolve before set_batch");
    p2_pending_ = false;   // a plain resolve writes every value
    if (narrow_active_) ensure_wide_store();
    narrow_pending_ = false;   // ... into the ordinary store
    if (outer_.n_input_words && !outer_.d_inputs) throw ZkError(ZK_ERR_INVALID, "outer input stream not bound");
    if (loop_.n_input_words && !loop_.d_inputs) throw ZkError(ZK_ERR_INVALID, "loop input stream not bound");
    hipStream_t st = (hipStream_t)stream;
    hip_check(hipMemsetAsync(d_mult_, 0, std::max<size_t>((size_t)batch_ * total_table_rows_ * 4, 4), st), "memset mult");
    uint32_t* const mult_inline = inline_multiplicities() ? d_mult_ : nullptr;
    auto oa = scope_args(outer_, outer_, loop_, limit_, d_tables_, d_table_words_, mult_inline, total_table_rows_);
    auto la = scope_args(loop_, outer_, loop_, limit_, d_tables_, d_table_words_, mult_inline, total_table_rows_);
    hip_check(hipEventRecord((hipEvent_t)ev_[0], st), "event");
    launch_phase(outer_, oa, 0, st);
    hip_check(hipEventRecord((hipEvent_t)ev_[1], st), "event");
    if (limit_) launch_phase(loop_, la, 0, st);
    hip_check(hipEventRecord((hipEvent_t)ev_[2], st), "event");
    launch_phase(outer_, oa, 1, st);
    launch_phase(outer_, oa, 2, st);
    count_multiplicities(st);
    hip_check(hipEventRecord((hipEvent_t)ev_[3], st), "event");
    hip_check(hipStreamSynchronize(st), "resolve sync");
    float a = 0, b = 0, c = 0;
    hipEventElapsedTime(&a, (hipEvent_t)ev_[0], (hipEvent_t)ev_[1]);
    hipEventElapsedTime(&b, (hipEvent_t)ev_[1], (hipEvent_t)ev_[2]);
    hipEventElapsedTime(&c, (hipEvent_t)ev_[2], (hipEvent_t)ev_[3]);
    ms_[0] = a + b + c; ms_[1] = b; ms_[4] = a + c;
    compact_ = true;  // home cells only: see check_satisfied / ensure_materialized
}

zkdev::CheckArgs CS::check_args(const Scope& s, unsigned long long* fail, bool compact, bool macro, bool fused, bool narrow) const {
    zkdev::CheckArgs a;
    a.alias = compact ? s.d_alias : nullptr;
    const bool full = !macro && s.d_cprog_full;   // the program without macro packets: locates a failure a macro packet reported
    a.cprog = compact ? (full ? s.d_cprog_full : s.d_cprog) : nullptr; a.chunk_tab = full ? s.d_cchunks_full : s.d_cchunks;
    a.n_chunks = full ? (uint32_t)s.cchunks_full.size() - 1 : (s.cchunks.empty() ? 0 : (uint32_t)s.cchunks.size() - 1);
    a.macros = (compact && !full) ? s.d_cmacros : nullptr; a.n_macros = (compact && !full) ? s.n_macro_p2 : 0;
    if (fused && compact && s.d_cprog_fused) {   // the gates the witness kernels did not evaluate themselves
        a.cprog = s.d_cprog_fused; a.chunk_tab = s.d_cchunks_fused; a.n_chunks = (uint32_t)s.cchunks_fused.size() - 1;
        a.macros = nullptr; a.n_macros = 0;
    }
    a.cells = compact ? s.d_store : s.d_cells; a.n_cells = compact ? s.store_geom() : s.n_cells; a.n_cols = geo_.num_columns_under_copy_permutation + lookup_width_ * lookup_reps_;
    if (narrow) {   // the fused step over the narrow store: the same packets with address words
        if (!(fused && compact && s.d_cprog_fused_n && s.d_store_n)) throw ZkError(ZK_ERR_INVALID, "internal: narrow check arguments outside the fused step");
        a.cprog = s.d_cprog_fused_n; a.cells = s.d_store_n; a.n_cells = s.narrow_geom();
    }
    a.n_lanes = s.n_lanes; a.n_slots = s.n_slots; a.rows = s.d_rows;
    a.rowconsts = s.d_rowconsts; a.lrows = s.d_lrows; a.n_copy_cols = geo_.num_columns_under_copy_permutation;
    a.lookup_width = lookup_width_; a.tables = d_tables_; a.table_words = d_table_words_; a.fail = fail;
    // >= ~2048 workgroups: lane tiles x slot chunks
    uint32_t lane_tiles = (s.n_lanes + 255) / 256;
    // at least two slot chunks per lane tile: with one, a thread walks every row of its lane and the kernel runs ~25 % slower
    // (measured at 152 instances of the VM shape, where the lane tiles alone exceed 2048)
    uint32_t chunks = std::max<uint32_t>(2, (2048 + lane_tiles - 1) / std::max<uint32_t>(lane_tiles, 1));
    chunks = std::min(chunks, s.n_slots);
    a.slots_per_chunk = (s.n_slots + chunks - 1) / chunks;
    return a;
}

int CS::check_satisfied(void* stream, zk_failure* first) {
    const int rc = check_satisfied_impl(stream, first, true);
    // a macro packet reported: the gate-by-gate program names the gate (same verdict, rare path)
    return rc == ZK_MACRO_FAILURE ? check_satisfied_impl(stream, first, false) : rc;
}

int CS::check_satisfied_impl(void* stream, zk_failure* first, bool macro) {
    if (batch_ == 0) throw ZkError(ZK_ERR_INVALID, "check_satisfied before set_batch");
    ensure_p2_filled(stream);
    hipStream_t st = (hipStream_t)stream;
    // compact trace (straight from the witness kernels): a variable has ONE stored value, the gate checker reads every cell
    // through the alias map and copy constraints inside a scope hold by construction (boojum's check_if_satisfied has no such
    // pass either: its values live per variable).  Materialised trace (zk_cs_trace_ptr / write_cell / prover-stage kernels were
    // used): every cell is checked as stored, and every copy pair explicitly.
    const bool compact = compact_;
    hip_check(hipMemsetAsync(d_fail_, 0xff, 8 * sizeof(unsigned long long), st), "memset fail");
    hip_check(hipEventRecord((hipEvent_t)ev_[4], st), "event");
    dev_check(zkdev::launch_check_gates(check_args(outer_, d_fail_, compact, macro), st));
    check_inputs_canonical(st, st);
    if (!compact)
        dev_check(zkdev::launch_check_copies(outer_.d_cells, outer_.n_cells, outer_.n_lanes, outer_.d_copies,
                                             (uint32_t)outer_.copies.size(), d_fail_, st));
    hip_check(hipEventRecord((hipEvent_t)ev_[5], st), "event");
    if (limit_) {
        dev_check(zkdev::launch_check_gates(check_args(loop_, d_fail_ + 3, compact, macro), st));
        hip_check(hipEventRecord((hipEvent_t)ev_[6], st), "event");
        if (!compact)
            dev_check(zkdev::launch_check_copies(loop_.d_cells, loop_.n_cells, loop_.n_lanes, loop_.d_copies,
                                                 (uint32_t)loop_.copies.size(), d_fail_ + 3, st));
        if (compact)
            dev_check(zkdev::launch_check_links(loop_.d_store, loop_.store_geom(), loop_.n_lanes, limit_, outer_.d_store,
                                                outer_.store_geom(), d_links_store_, (uint32_t)links_store_.size(), d_fail_ + 3, st));
        else
            dev_check(zkdev::launch_check_links(loop_.d_cells, loop_.n_cells, loop_.n_lanes, limit_, outer_.d_cells,
                                                outer_.n_cells, d_links_, (uint32_t)links_.size(), d_fail_ + 3, st));
        check_streams(st, compact);
    } else {
        hip_check(hipEventRecord((hipEvent_t)ev_[6], st), "event");
    }
    hip_check(hipEventRecord((hipEvent_t)ev_[7], st), "event");
    unsigned long long f[8];
    hip_check(hipMemcpyAsync(f, d_fail_, sizeof f, hipMemcpyDeviceToHost, st), "memcpy fail");
    hip_check(hipStreamSynchronize(st), "check sync");
    float tot = 0, g = 0;
    hipEventElapsedTime(&tot, (hipEvent_t)ev_[4], (hipEvent_t)ev_[7]);
    hipEventElapsedTime(&g, (hipEvent_t)ev_[5], (hipEvent_t)ev_[6]);
    ms_[2] = tot; ms_[3] = g;
    return decode_failure(f, first);
}

void CS::ensure_materialized(void* stream) {
    if (!compact_ || batch_ == 0) return;
    ensure_p2_filled(stream);
    hipStream_t st = (hipStream_t)stream;
    for (Scope* s : {&outer_, &loop_}) {
        if (s->is_loop && !limit_) {
            if (!s->d_cells) hip_check(hipMalloc((void**)&s->d_cells, 8), "hipMalloc trace");
            continue;
        }
        if (!s->d_cells) {  // the full trace exists only for its consumers: prover-stage kernels, trace readers, write_cell
            size_t bytes = std::max<size_t>((size_t)s->n_cells * s->stride * 8, 8);
            if (hipMalloc((void**)&s->d_cells, bytes) != hipSuccess) {
                s->d_cells = nullptr;
                (void)hipGetLastError();  // the refusal is reported here: it must not surface again at the next launch check
                throw ZkError(ZK_ERR_CAPACITY, "materialised trace does not fit in device memory at this batch size (the variable store is 4x smaller)");
            }
            hip_check(hipMemsetAsync(s->d_cells, 0, bytes, st), "hipMemset trace");
        }
        dev_check(zkdev::launch_materialize(s->d_cells, s->n_cells, s->d_store, s->store_geom(), s->n_lanes, s->d_mat_pairs, (uint32_t)s->mat_pairs.size(), st));
    }
    hip_check(hipStreamSynchronize(st), "materialize sync");
    compact_ = false;
}

// K10 (kernels_lookup_arg.hpp): the witness side sums 1/f over every lookup tuple of the trace, the table side sums
// multiplicity/f over the table rows; equality per instance is the log-derivative lookup argument.
uint32_t CS::lookup_argument(const uint64_t beta[2], const uint64_t gamma[2], void* stream, std::vector<uint64_t>& out) {
    if (batch_ == 0 || !uploaded_) throw ZkError(ZK_ERR_INVALID, "lookup_argument before set_batch / resolve");
    ensure_materialized(stream);
    if (lookup_reps_ > 32) throw ZkError(ZK_ERR_INVALID, "lookup_argument: more than 32 repetitions per row");
    hipStream_t st = (hipStream_t)stream;
    auto emul = [](const uint64_t x[2], const uint64_t y[2], uint64_t r[2]) {  // host GF(p^2), X^2 = 7
        auto mulm = [](uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % 0xFFFFFFFF00000001ull); };
        auto addm = [](uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % 0xFFFFFFFF00000001ull); };
        uint64_t a = addm(mulm(x[0], y[0]), mulm(7, mulm(x[1], y[1]))), b = addm(mulm(x[0], y[1]), mulm(x[1], y[0]));
        r[0] = a; r[1] = b;
    };
    if (lookup_width_ > 4) throw ZkError(ZK_ERR_INVALID, "lookup_argument: lookup width above 4");
    uint64_t ch[10] = {beta[0], beta[1], gamma[0], gamma[1], 0, 0, 0, 0, 0, 0};
    emul(gamma, gamma, ch + 4);
    emul(ch + 4, gamma, ch + 6);
    emul(ch + 6, gamma, ch + 8);
    for (int i = 0; i < 10; ++i)
        if (ch[i] >= 0xFFFFFFFF00000001ull) throw ZkError(ZK_ERR_INVALID, "lookup_argument: non-canonical challenge");
    uint64_t *d_acc_o = nullptr, *d_acc_l = nullptr, *d_inv = nullptr, *d_ab = nullptr;
    struct Temps {  // device temporaries of this call: released on every exit path, exceptions included
        std::vector<void*> ptrs;
        ~Temps() { for (void* p : ptrs) hipFree(p); }
    } temps;
    auto alloc = [&](uint64_t** p, size_t words) {
        hip_check(hipMalloc((void**)p, std::max<size_t>(words, 1) * 8), "hipMalloc lookup_argument");
        temps.ptrs.push_back(*p);
    };
    alloc(&d_acc_o, 2 * (size_t)outer_.n_lanes);
    alloc(&d_acc_l, 2 * (size_t)loop_.n_lanes);
    alloc(&d_inv, 2 * (size_t)total_table_rows_);
    alloc(&d_ab, 4 * (size_t)batch_);
    const uint32_t n_cols = geo_.num_columns_under_copy_permutation + lookup_width_ * lookup_reps_;
    auto side = [&](const Scope& s, uint64_t* acc) {
        zkdev::LookupArgArgs a{s.d_cells, s.n_cells, n_cols, s.n_lanes, s.n_slots, geo_.num_columns_under_copy_permutation, lookup_width_, s.d_lrows, acc};
        dev_check(zkdev::launch_lookup_arg_witness(a, ch, st));
    };
    side(outer_, d_acc_o);
    if (limit_) side(loop_, d_acc_l);
    dev_check(zkdev::launch_lookup_arg_witness_sum(d_acc_o, d_acc_l, limit_, batch_, d_ab, st));
    dev_check(zkdev::launch_lookup_arg_tables(d_tables_, (uint32_t)tables_.size() + 1, d_table_words_, total_table_rows_, lookup_width_, ch, d_inv, d_mult_, batch_,
                                              d_ab + 2 * (size_t)batch_, st));
    std::vector<uint64_t> h(4 * (size_t)batch_);
    hip_check(hipMemcpyAsync(h.data(), d_ab, h.size() * 8, hipMemcpyDeviceToHost, st), "memcpy lookup_argument");
    hip_check(hipStreamSynchronize(st), "lookup_argument sync");
    out.resize(4 * (size_t)batch_);
    uint32_t bad = 0;
    for (uint32_t i = 0; i < batch_; ++i) {
        out[4 * i] = h[2 * i]; out[4 * i + 1] = h[2 * i + 1];
        out[4 * i + 2] = h[2 * ((size_t)batch_ + i)]; out[4 * i + 3] = h[2 * ((size_t)batch_ + i) + 1];
        bad += (out[4 * i] != out[4 * i + 2] || out[4 * i + 1] != out[4 * i + 3]);
    }
    return bad;
}

void CS::check_streams(void* stream, bool compact, bool narrow) {
    const auto& streams = compact ? streams_store_ : streams_;
    const auto& dev = narrow ? d_streams_store_n_ : compact ? d_streams_store_ : d_streams_;
    for (size_t i = 0; i < streams.size(); ++i) {
        const auto& sr = streams[i];
        dev_check(zkdev::launch_check_stream(narrow ? loop_.d_store_n : compact ? loop_.d_store : loop_.d_cells, narrow ? loop_.narrow_geom() : compact ? loop_.store_geom() : loop_.n_cells, batch_, limit_, dev[i],
                                             (uint32_t)sr.a.size(), dev[i] + sr.a.size(), (uint32_t)sr.b.size(), sr.n_total, (uint32_t)i,
                                             d_fail_ + 3, stream));
    }
}

// every word of the bound input streams is a canonical field element (< p): the kernels compare values as plain u64
void CS::check_inputs_canonical(void* outer_stream, void* loop_stream) {
    for (int sc = 0; sc < 2; ++sc) {
        const Scope& s = sc ? loop_ : outer_;
        if (!s.d_inputs || !s.n_input_words || (sc && !limit_)) continue;
        dev_check(zkdev::launch_check_inputs(s.d_inputs, s.n_input_words, s.n_lanes, s.input_stride ? s.input_stride : s.n_lanes, d_fail_ + 3 * sc,
                                             sc ? loop_stream : outer_stream));
    }
}

int CS::decode_failure(const unsigned long long* f, zk_failure* first) const {
    const unsigned long long NONE = ~0ull;
    for (int sc = 0; sc < 2; ++sc) {
        const Scope& s = sc ? loop_ : outer_;
        const unsigned long long* ff = f + 3 * sc;
        uint32_t lim = sc ? limit_ : 1;
        auto fill = [&](unsigned long long key, uint32_t slot, uint32_t kind, uint32_t rel) {
            if (first) {
                uint32_t lane = (uint32_t)(key >> 32);
                first->scope = sc; first->instance = lane / lim; first->iteration = lane % lim;
                first->slot = slot; first->kind = kind; first->relation = rel;
            }
        };
        if (ff[0] != NONE) {
            uint32_t slot = (uint32_t)((ff[0] >> 12) & 0xfffff), j = (uint32_t)((ff[0] >> 4) & 0xff), rel = (uint32_t)(ff[0] & 0xf);
            bool is_lookup = (j & 0x80) && rel == 15;
            if (slot == 0xfffffu) { fill(ff[0], j, ZK_FAILURE_NONCANONICAL_INPUT, 0); return ZK_ERR_UNSATISFIED; }  // input word j (mod 256) >= p
            if (slot == 0xffffeu) return ZK_MACRO_FAILURE;  // a macro packet: the caller re-runs the gate-by-gate program to name the gate
            fill(ff[0], slot, is_lookup ? 0x100u : s.rows[slot].kind, is_lookup ? (j & 0x7f) : rel);
            return ZK_ERR_UNSATISFIED;
        }
        if (ff[1] != NONE) {  // copy constraint: report the cell index in `slot`
            uint32_t pi = (uint32_t)(ff[1] & 0xffffffffu);
            fill(ff[1], s.copies[pi].cell, 0x200u, pi);
            return ZK_ERR_UNSATISFIED;
        }
        if (ff[2] != NONE) {
            uint32_t li = (uint32_t)(ff[2] & 0xffffffffu);
            if (li & 0x80000000u) fill(ff[2], 0, ZK_FAILURE_STREAM_LINK, li & 0x7fffffffu);  // stream link: relation = stream index
            else fill(ff[2], links_[li].loop_cell, 0x300u | links_[li].kind, li);
            return ZK_ERR_UNSATISFIED;
        }
    }
    return ZK_OK;
}


// Fused witness generation + satisfiability check with the latency-bound outer scope (lane ==
// instance, one wave) overlapped with the bandwidth-bound loop-scope kernels on a second stream:
//   aux : outer PRE --ev--> ........................ outer POST -> gates(outer) -> copies(outer) --ev-->
//   main:            wait -> LOOP witness --ev--> gates(loop) -> copies(loop) ............ wait -> links
int CS::resolve_and_check(void* stream, zk_failure* first) {
    if (batch_ == 0) throw ZkError(ZK_ERR_INVALID, "resolve_and_check before set_batch");
    if (outer_.n_input_words && !outer_.d_inputs) throw ZkError(ZK_ERR_INVALID, "outer input stream not bound");
    if (loop_.n_input_words && !loop_.d_inputs) throw ZkError(ZK_ERR_INVALID, "loop input stream not bound");
    hipStream_t st = (hipStream_t)stream;
    if (!aux_stream_) {
        hipStream_t a;
        hip_check(hipStreamCreateWithFlags(&a, hipStreamNonBlocking), "hipStreamCreate aux");
        aux_stream_ = (void*)a;
        for (auto& e : ev2_) { hipEvent_t he; hip_check(hipEventCreate(&he), "hipEventCreate"); e = (void*)he; }
    }
    hipStream_t ax = (hipStream_t)aux_stream_;
    auto E = [&](int i) { return (hipEvent_t)ev2_[i]; };
    hip_check(hipMemsetAsync(d_mult_, 0, std::max<size_t>((size_t)batch_ * total_table_rows_ * 4, 4), st), "memset mult");
    hip_check(hipMemsetAsync(d_fail_, 0xff, 8 * sizeof(unsigned long long), st), "memset fail");
    uint32_t* const mult_inline = inline_multiplicities() ? d_mult_ : nullptr;
    auto oa = scope_args(outer_, outer_, loop_, limit_, d_tables_, d_table_words_, mult_inline, total_table_rows_);
    auto la = scope_args(loop_, outer_, loop_, limit_, d_tables_, d_table_words_, mult_inline, total_table_rows_);
    // FUSED mode (default): the witness kernels evaluate the gates mirrored by their producing ops on the values they hold, the
    // checkers read what is left.  ZKGL_VERIFY_STORED=1: every gate re-evaluated from the stored values (what check_if_satisfied does).
    const char* vs = std::getenv("ZKGL_VERIFY_STORED");
    const bool fused = !check_stored_ && !(vs && vs[0] == '1') && outer_.d_cprog_fused && (!limit_ || loop_.d_cprog_fused);
    last_check_fused_ = fused;
    if (fused) { oa.fail = d_fail_; la.fail = d_fail_ + 3; }
    // NARROW store (store_geom.hpp): the fused step writes and reads the loop scope's values there — the loop kernel, the fused check program,
    // the links, ZK_OP_LOOP_LAST of the outer post phase; the ordinary store is stale until somebody asks for it (ensure_p2_filled widens)
    const bool narrow = fused && narrow_active_ && !narrow_suspended_ && limit_;
    if (narrow) {
        la.cells = loop_.d_store_n; la.n_cells = loop_.narrow_geom(); la.cls = loop_.d_prog2n + loop_.cls_off;
        if (!loop_last_slots_.empty()) ensure_wide_store();   // ZK_OP_LOOP_LAST of the outer post phase reads the ordinary store: k_widen_last below
        ++narrow_steps_;
    } else if (narrow_active_) ensure_wide_store();
    // deferred mode: only when the step is fused (nothing in it reads the intermediates) and every in-circuit permutation of the loop scope
    // has a verified descriptor for the fill kernel
    const bool defer = fused && defer_p2_ && limit_ && loop_.d_cmacros && loop_.n_macro_p2 == loop_.n_p2_rounds_ops && loop_.p2_intermediates_private;
    la.defer_p2 = defer ? 1u : 0u;
    la.clock_probe = d_fail_ + 6;   // words 6, 7 of the block travel back with the verdict
    hip_check(hipMemsetAsync(d_fail_ + 8, 0, 2 * sizeof(unsigned long long), st), "memset p2 stats");
    la.p2_stats = d_fail_ + 8;      // words 8, 9: gated witness-only permutations skipped / run by the loop kernel's wavefronts
    hip_check(hipEventRecord(E(0), st), "event");                     // t0 (+ memsets done)
    hip_check(hipStreamWaitEvent(ax, E(0), 0), "wait");
    launch_phase(outer_, oa, 0, ax);    // outer PRE
    hip_check(hipEventRecord(E(1), ax), "event");
    hip_check(hipStreamWaitEvent(st, E(1), 0), "wait");
    hip_check(hipEventRecord(E(2), st), "event");
    if (limit_) launch_phase(loop_, la, 0, st);   // LOOP
    hip_check(hipEventRecord(E(3), st), "event");
    launch_phase(outer_, oa, 1, ax);             // outer SIDE (|| LOOP)
    if (narrow && !loop_last_slots_.empty()) {   // the values the outer post phase takes from the last iteration, into the ordinary store it reads
        dev_check(zkdev::launch_widen_last(loop_.d_store_n, loop_.narrow_geom(), loop_.d_store, loop_.store_geom(), batch_, limit_, loop_.d_slot_aw, d_loop_last_slots_,
                                           (uint32_t)loop_last_slots_.size(), st));
        hip_check(hipEventRecord(E(8), st), "event");
        hip_check(hipStreamWaitEvent(ax, E(8), 0), "wait");
    }
    hip_check(hipStreamWaitEvent(ax, E(3), 0), "wait");
    launch_phase(outer_, oa, 2, ax);  // outer POST
    // compact traces: the gate checkers read every cell through the alias map; no copy pass (see check_satisfied)
    dev_check(zkdev::launch_check_gates(check_args(outer_, d_fail_, true, true, fused), ax));
    check_inputs_canonical(ax, ax);   // both scopes on the auxiliary stream: it has slack behind the loop-scope kernels
    hip_check(hipEventRecord(E(4), ax), "event");
    if (limit_) {
        dev_check(zkdev::launch_check_gates(check_args(loop_, d_fail_ + 3, true, true, fused, narrow), st));
        hip_check(hipEventRecord(E(5), st), "event");
    } else {
        hip_check(hipEventRecord(E(5), st), "event");
    }
    hip_check(hipEventRecord(E(6), st), "event");
    // the lookup multiplicities of the batch (k_multiplicities, no atomics in the witness kernels): the loop scope's pass needs only
    // the loop kernels and runs underneath the outer POST phase of the auxiliary stream (keccak FSM: a 5.7 ms pass under a 6.6 ms
    // phase of sequential commitments); the outer scope's after it
    count_multiplicities(st, 2);
    hip_check(hipStreamWaitEvent(st, E(4), 0), "wait");
    count_multiplicities(st, 1);
    if (limit_) {
        dev_check(zkdev::launch_check_links(narrow ? loop_.d_store_n : loop_.d_store, narrow ? loop_.narrow_geom() : loop_.store_geom(), loop_.n_lanes, limit_, outer_.d_store,
                                            outer_.store_geom(), narrow ? d_links_store_n_ : d_links_store_, (uint32_t)links_store_.size(), d_fail_ + 3, st));
        check_streams(st, true, narrow);
    }
    hip_check(hipEventRecord(E(7), st), "event");
    unsigned long long f[10];
    hip_check(hipMemcpyAsync(f, d_fail_, sizeof f, hipMemcpyDeviceToHost, st), "memcpy fail");
    hip_check(hipStreamSynchronize(st), "pipeline sync");
    p2_skipped_ = f[8]; p2_run_ = f[9];
    float loop_ms = 0, gates_ms = 0, copies_ms = 0, total = 0, outer_post = 0;
    hipEventElapsedTime(&loop_ms, E(2), E(3));
    hipEventElapsedTime(&gates_ms, E(3), E(5));
    hipEventElapsedTime(&copies_ms, E(5), E(6));
    hipEventElapsedTime(&total, E(0), E(7));
    hipEventElapsedTime(&outer_post, E(3), E(4));
    ms_[0] = total; ms_[1] = loop_ms; ms_[2] = gates_ms + copies_ms; ms_[3] = gates_ms; ms_[4] = outer_post;
    // shader clock of the loop launch: s_memtime ticks per s_memrealtime tick (100 MHz) over the grid's first wavefront
    loop_shader_mhz_ = (limit_ && f[7] != ~0ull && f[7] != 0) ? (float)((double)f[6] / (double)f[7] * 100.0) : 0.0f;
    compact_ = true;
    p2_pending_ = defer;
    narrow_pending_ = narrow;
    const int rc = decode_failure(f, first);
    if (narrow && rc != ZK_OK) {
        // anything reported over the narrow store — a violated relation, or a value that does not fit its byte slot (stored truncated) — is
        // decided by repeating the step over the ordinary store: the verdict and the reported gate are exactly those of a build without narrow slots
        ++narrow_repeats_;
        narrow_suspended_ = true;
        int rc2;
        try { rc2 = resolve_and_check(stream, first); } catch (...) { narrow_suspended_ = false; throw; }
        narrow_suspended_ = false;
        return rc2;
    }
    return rc == ZK_MACRO_FAILURE ? check_satisfied_impl(stream, first, false) : rc;
}

void CS::set_check_mode(uint32_t mode) {
    check_stored_ = mode == ZK_CHECK_STORED;
    defer_p2_ = mode == ZK_CHECK_FUSED_DEFER_P2;
}

// deferred mode (ZK_CHECK_FUSED_DEFER_P2): the loop kernel left the 950 intermediates of every in-circuit Poseidon2 permutation
// unwritten; everything that reads the store beyond the fused step (the full check, trace readers, the prover-stage kernels, read_var,
// the fault-injection hooks) comes through here first
std::vector<uint32_t> CS::narrow_byte_input_words() const {
    if (!finalized_) throw ZkError(ZK_ERR_INVALID, "narrow_byte_input_words before finalize");
    std::vector<uint32_t> out;
    if (!loop_.narrow_ok) return out;
    for (auto& kv : loop_.input_word)
        if (loop_.slot_aw[loop_.var_slot[kv.first]] & zkgeom::AW_BYTE) out.push_back(kv.second);
    std::sort(out.begin(), out.end());
    return out;
}

void CS::ensure_wide_store() {
    if (loop_.d_store || !limit_) return;
    const uint64_t store_lanes = zkgeom::padded_lanes(loop_.store_geom(), loop_.n_lanes);
    const size_t bytes = std::max<size_t>((size_t)loop_.n_store * store_lanes * 8, 8);
    if (hipMalloc((void**)&loop_.d_store, bytes) != hipSuccess) {
        loop_.d_store = nullptr;
        (void)hipGetLastError();
        throw ZkError(ZK_ERR_CAPACITY, "the ordinary loop store does not fit beside the narrow one at this batch size (a reader outside the fused step asked for it)");
    }
    hip_check(hipMemset(loop_.d_store, 0, bytes), "hipMemset variable store");
}

void CS::ensure_p2_filled(void* stream) {
    if (narrow_active_) ensure_wide_store();   // every reader that comes through here addresses the ordinary store: it exists from here on (ZK_ERR_CAPACITY if it cannot)
    if (narrow_pending_) {   // the last step wrote the narrow store: expand it into the ordinary store every other reader addresses
        ensure_wide_store();
        dev_check(zkdev::launch_widen_store(loop_.d_store_n, loop_.narrow_geom(), loop_.d_store, loop_.store_geom(), loop_.n_lanes, loop_.d_slot_aw, loop_.n_store, stream));
        hip_check(hipStreamSynchronize((hipStream_t)stream), "widen sync");
        narrow_pending_ = false;
    }
    if (!p2_pending_) return;
    dev_check(zkdev::launch_fill_p2(loop_.d_store, loop_.store_geom(), loop_.n_lanes, loop_.d_cmacros, loop_.n_macro_p2, stream));
    hip_check(hipStreamSynchronize((hipStream_t)stream), "fill_p2 sync");
    p2_pending_ = false;
}

uint64_t CS::read_var(zk_var v, uint32_t instance, uint32_t iteration) {
    if (batch_ == 0) throw ZkError(ZK_ERR_INVALID, "read_var before set_batch");
    Scope& s = scope_of(v);
    if (s.is_loop || !compact_) ensure_p2_filled(nullptr);   // (only the loop scope's store can be incomplete: deferred intermediates, a pending narrow store)
    if (var_index(v) >= s.n_vars || instance >= batch_) throw ZkError(ZK_ERR_INVALID, "read_var: out of range");
    uint64_t lane = instance;
    if (s.is_loop) {
        if (iteration >= limit_) throw ZkError(ZK_ERR_INVALID, "read_var: iteration out of range");
        lane = (uint64_t)instance * limit_ + iteration;
    }
    uint64_t out = 0;
    if (compact_)
        hip_check(hipMemcpy(&out, s.d_store + tiled_offset(s.store_geom(), s.var_slot[var_index(v)], lane), 8, hipMemcpyDeviceToHost), "read_var memcpy");
    else
        hip_check(hipMemcpy(&out, s.d_cells + tiled_offset(s.n_cells, s.var_cells[var_index(v)][0], lane), 8, hipMemcpyDeviceToHost), "read_var memcpy");
    return out;
}

void CS::write_cell(bool loop_scope, uint32_t cell, uint32_t lane, uint64_t value) {
    Scope& s = loop_scope ? loop_ : outer_;
    if (batch_ == 0 || cell >= s.n_cells || lane >= s.n_lanes) throw ZkError(ZK_ERR_INVALID, "write_cell: out of range");
    ensure_materialized(nullptr);  // an externally modified trace is checked cell by cell, copies and links included
    hip_check(hipMemcpy(s.d_cells + tiled_offset(s.n_cells, cell, lane), &value, 8, hipMemcpyHostToDevice), "write_cell memcpy");
}

int CS::hook_compare_witness(const zk_var* vars, uint32_t n_vars, const uint64_t* dev_expected, void* stream, zk_failure* first) {
    if (batch_ == 0) throw ZkError(ZK_ERR_INVALID, "hook_compare_witness before set_batch");
    if (!vars || !n_vars || !dev_expected) throw ZkError(ZK_ERR_INVALID, "hook_compare_witness: null argument");
    std::vector<uint32_t> slots;
    for (uint32_t i = 0; i < n_vars; ++i) {
        if (is_loop_var(vars[i])) throw ZkError(ZK_ERR_INVALID, "hook_compare_witness: the closed-form input lives in the outer scope");
        if (var_index(vars[i]) >= outer_.n_vars) throw ZkError(ZK_ERR_INVALID, "hook_compare_witness: variable out of range");
        slots.push_back(outer_.var_slot[var_index(vars[i])]);
    }
    hipStream_t st = (hipStream_t)stream;
    uint32_t* d_slots = upload(slots);
    hip_check(hipMemsetAsync(d_fail_, 0xff, sizeof(unsigned long long), st), "memset fail");
    dev_check(zkdev::launch_hook_compare(outer_.d_store, outer_.store_geom(), d_slots, n_vars, batch_, dev_expected, d_fail_, st));
    unsigned long long f = 0;
    hip_check(hipMemcpyAsync(&f, d_fail_, sizeof f, hipMemcpyDeviceToHost, st), "memcpy fail");
    hip_check(hipStreamSynchronize(st), "hook sync");
    hipFree(d_slots);
    if (f == ~0ull) return ZK_OK;
    if (first) { std::memset(first, 0, sizeof *first); first->instance = (uint32_t)(f >> 32); first->slot = (uint32_t)f; first->kind = ZK_FAILURE_HOOK_DIFF; }
    return ZK_ERR_UNSATISFIED;
}

// test hook: overwrite one value of the variable store (the compact trace stays compact) — fault injection below the level of
// write_cell, which materialises the trace; lets tests corrupt a Poseidon2 intermediate that only a macro check packet reads
void CS::debug_poke_store(bool loop_scope, uint32_t slot, uint32_t lane, uint64_t value) {
    Scope& s = loop_scope ? loop_ : outer_;
    if (batch_ == 0 || !compact_ || slot >= s.n_store || lane >= s.n_lanes) throw ZkError(ZK_ERR_INVALID, "debug_poke_store: out of range or trace materialised");
    ensure_p2_filled(nullptr);
    hip_check(hipMemcpy(s.d_store + tiled_offset(s.store_geom(), slot, lane), &value, 8, hipMemcpyHostToDevice), "poke memcpy");
}

uint32_t CS::pack_public_inputs(uint64_t* dev_out, void* stream) {
    if (batch_ == 0) throw ZkError(ZK_ERR_INVALID, "pack_public_inputs before set_batch");
    const uint32_t n = (uint32_t)public_vars_.size();
    if (!n) return 0;
    if (!d_public_slots_) {
        std::vector<uint32_t> slots;
        for (uint32_t v : public_vars_) slots.push_back(outer_.var_slot[v]);
        d_public_slots_ = upload(slots);
    }
    dev_check(zkdev::launch_pack_public(outer_.d_store, outer_.store_geom(), d_public_slots_, n, batch_, dev_out, stream));
    return n;
}

std::vector<uint64_t> CS::public_inputs(uint32_t instance) {
    std::vector<uint64_t> out;
    for (uint32_t v : public_vars_) out.push_back(read_var(v, instance, 0));
    return out;
}

uint32_t CS::var_cell(zk_var v) const {
    if (!finalized_) throw ZkError(ZK_ERR_INVALID, "var_cell before finalize");
    const Scope& s = is_loop_var(v) ? loop_ : outer_;
    if (var_index(v) >= s.n_vars) throw ZkError(ZK_ERR_INVALID, "var_cell: out of range");
    return s.var_cells[var_index(v)][0];
}
std::vector<uint32_t> CS::public_cells() const {
    if (!finalized_) throw ZkError(ZK_ERR_INVALID, "public_cells before finalize");
    std::vector<uint32_t> out;
    for (uint32_t v : public_vars_) out.push_back(outer_.var_cells[v][0]);
    return out;
}

std::vector<uint32_t> CS::multiplicities(uint32_t instance) {
    if (instance >= batch_) throw ZkError(ZK_ERR_INVALID, "multiplicities: instance out of range");
    std::vector<uint32_t> out(total_table_rows_);
    if (total_table_rows_)
        hip_check(hipMemcpy(out.data(), d_mult_ + (size_t)instance * total_table_rows_, (size_t)total_table_rows_ * 4,
                            hipMemcpyDeviceToHost), "multiplicities memcpy");
    return out;
}

void CS::stats(zk_stats* o) const {
    std::memset(o, 0, sizeof *o);
    o->loop_slots = loop_.n_slots * (limit_ ? 1 : 0); o->outer_slots = outer_.n_slots; o->limit = limit_;
    o->rows_per_instance = (uint64_t)loop_.n_slots * limit_ + outer_.n_slots;
    o->copy_columns = geo_.num_columns_under_copy_permutation;
    o->lookup_columns = lookup_width_ * lookup_reps_;
    o->variables_outer = outer_.n_vars; o->variables_loop = loop_.n_vars;
    o->constraints_per_instance = outer_.n_constraints + loop_.n_constraints * limit_;
    o->var_cells_per_instance = o->rows_per_instance * (o->copy_columns + o->lookup_columns);
    static_assert(ZK_GATE__COUNT <= 16, "zk_stats.gate_instances holds 16 kinds");
    for (int k = 0; k < ZK_GATE__COUNT; ++k) o->gate_instances[k] = outer_.gate_counts[k] + loop_.gate_counts[k] * limit_;
    o->lookups_per_instance = outer_.lookups.size() + loop_.lookups.size() * (uint64_t)limit_;
    o->program_words_outer = outer_.prog2.size(); o->program_words_loop = loop_.prog2.size();
    o->scratch_cells_outer = outer_.n_scratch; o->scratch_cells_loop = loop_.n_scratch;
    o->cells_written_outer = outer_.cells_written; o->cells_written_loop = loop_.cells_written;
    o->cells_populated_outer = outer_.cells_populated; o->cells_populated_loop = loop_.cells_populated;
    o->loop_store_tile_lanes = batch_ ? (1ull << loop_.store_tile_log2) : 0;
    o->copy_pairs_outer = outer_.copies.size(); o->copy_pairs_loop = loop_.copies.size();
    o->seed_ops = seed_ops_; o->seed_words = seed_prog_.size(); o->seed_slots = seed_slots_; o->loop_ops = loop_.ops.size();
    // fused mode: relations left to the check program (gates not mirrored by their producing op) / evaluated where they are produced
    auto from_store = [](const Scope& s) {
        uint64_t n = 0;
        for (size_t gi = 0; gi < s.gates.size(); ++gi)
            if (gi >= s.gate_mirrored.size() || !s.gate_mirrored[gi]) n += GATES[s.gates[gi].kind].n_relations;
        for (auto& l : s.lookups) n += l.owner < 0;   // a tuple nobody's witness op evaluates stays in the fused check program (lookup_given outside a macro window)
        return n;
    };
    const bool fused_ok = !outer_.cprog_fused.empty() && (!limit_ || !loop_.cprog_fused.empty());
    o->constraints_from_store_fused = fused_ok ? from_store(outer_) + from_store(loop_) * limit_ : o->constraints_per_instance;
    o->constraints_in_witness_fused = o->constraints_per_instance - o->constraints_from_store_fused;
    o->values_below_2_32_outer = outer_.values_below_2_32; o->values_below_2_32_loop = loop_.values_below_2_32;
    o->seed_cone_unsupported = seed_cone_unsupported_;
    o->store_bytes_per_lane_loop = (uint64_t)loop_.cells_written * 8;
    o->narrow_store_bytes_per_lane_loop = loop_.narrow_ok ? loop_.narrow_units : 0;
    o->narrow_byte_values_loop = loop_.narrow_ok ? loop_.narrow_byte_values : 0;
    o->narrow_store_active = narrow_active_ ? 1 : 0;
    o->narrow_steps = narrow_steps_; o->narrow_repeats = narrow_repeats_;
    o->narrow_store_pending = narrow_pending_ ? 1 : 0;
}

float CS::last_ms(int which) const {
    if (which >= 5 && which < 8) return last_seed_phase_ms[which - 5];  // native seeding phases (ZKGL_SEED_PHASE_MS=1): walker, chains, fill
    if (which == 8) return loop_shader_mhz_;
    if (which == 9) return (p2_skipped_ + p2_run_) ? (float)((double)p2_skipped_ / (double)(p2_skipped_ + p2_run_)) : 0.0f;  // not a time either  // not a time: the shader clock (MHz) the last resolve_and_check's loop launch ran at
    return (which >= 0 && which < 5) ? ms_[which] : -1.0f;
}

// Serialised scope for the CPU oracle:
// [magic, is_loop, n_cells, n_trace_cells, n_slots, n_copy_cols, lookup_width, n_input_words, limit,
//  pre_words, n_prog, n_consts, n_rows, n_rowconsts, n_lrows, n_copies, n_tables, n_table_words, n_links, n_carries]
// followed by the sections in that order (u64 sections as lo,hi u32 pairs).
std::vector<uint32_t> CS::export_scope(bool loop_scope) const {
    const Scope& s = loop_scope ? loop_ : outer_;
    std::vector<uint32_t> o;
    auto p64 = [&](uint64_t v) { o.push_back((uint32_t)v); o.push_back((uint32_t)(v >> 32)); };
    std::vector<uint64_t> words;
    for (auto& t : tables_) words.insert(words.end(), t.rows.begin(), t.rows.end());
    std::vector<uint32_t> stream_words;
    if (loop_scope)
        for (auto& sr : streams_) {
            stream_words.push_back((uint32_t)sr.a.size()); stream_words.push_back((uint32_t)sr.b.size()); stream_words.push_back(sr.n_total);
            stream_words.insert(stream_words.end(), sr.a.begin(), sr.a.end());
            stream_words.insert(stream_words.end(), sr.b.begin(), sr.b.end());
        }
    uint32_t hdr[21] = {0x5a4b4733u, s.is_loop ? 1u : 0u, s.n_cells, s.n_trace_cells, s.n_slots,
                        geo_.num_columns_under_copy_permutation, lookup_width_, s.n_input_words, limit_, s.pre_words_full,
                        (uint32_t)s.prog_full.size(), (uint32_t)s.const_pool.size(), (uint32_t)s.rows.size(),
                        (uint32_t)s.rowconsts.size(), (uint32_t)s.lrows.size(), (uint32_t)s.copies.size(),
                        (uint32_t)tables_.size() + 1, (uint32_t)words.size(), (uint32_t)(loop_scope ? links_.size() : 0),
                        (uint32_t)(loop_scope ? carries_.size() : 0), (uint32_t)stream_words.size()};
    o.insert(o.end(), hdr, hdr + 21);
    o.insert(o.end(), s.prog_full.begin(), s.prog_full.end());  // the oracle materialises every cell
    for (uint64_t c : s.const_pool) p64(c);
    for (auto& r : s.rows) { o.push_back(r.kind); o.push_back(r.n_instances); o.push_back(r.const_off); o.push_back(r.n_consts); }
    for (uint64_t c : s.rowconsts) p64(c);
    for (auto& r : s.lrows) { o.push_back(r.table); o.push_back(r.n_tuples); }
    for (auto& c : s.copies) { o.push_back(c.cell); o.push_back(c.home); }
    for (uint32_t i = 0; i < 9; ++i) o.push_back(0);  // table 0 = none
    for (auto& t : tables_) {
        o.push_back(t.word_off); o.push_back(t.mult_off); o.push_back(t.n_rows); o.push_back(t.n_keys); o.push_back(t.n_vals);
        o.push_back(t.dense ? 1 : 0); o.push_back(t.key_shift[0]); o.push_back(t.key_shift[1]); o.push_back(t.key_shift[2]);
    }
    for (uint64_t w : words) p64(w);
    if (loop_scope)
        for (auto& l : links_) { o.push_back(l.kind); o.push_back(l.loop_cell); o.push_back(l.other_cell); o.push_back(0); }
    if (loop_scope)
        for (auto& c : carries_) { o.push_back(c.word); o.push_back(c.out_cell); o.push_back(c.first_outer_cell); o.push_back(c.has_first); }
    o.insert(o.end(), stream_words.begin(), stream_words.end());
    return o;
}

// the trace as a VIEW of the compact store: d_slot1[trace cell] = store slot + 1 of the variable placed there, 0 = unpopulated (readers
// that walk trace cells without materialising the batch: zk_cs_trace_columns*, K12)
void CS::ensure_trace_view() {
    for (Scope* s : {&outer_, &loop_})
        if (!s->d_slot1 && s->n_trace_cells) {
            std::vector<uint32_t> t(s->n_trace_cells, 0);
            for (auto& pr : s->mat_pairs)
                if (pr.cell < s->n_trace_cells) t[pr.cell] = pr.home + 1;
            s->d_slot1 = upload(t);
            if (s->narrow_ok && !s->d_aw1) {   // the same view over the narrow store: address word + 1
                std::vector<uint32_t> ta(s->n_trace_cells, 0);
                for (auto& pr : s->mat_pairs)
                    if (pr.cell < s->n_trace_cells) ta[pr.cell] = s->slot_aw[pr.home] + 1;
                s->d_aw1 = upload(ta);
            }
        }
}

void CS::trace_columns(uint32_t instance, uint64_t* d_out, uint32_t log_n, uint64_t stride, void* stream, uint32_t n_instances, uint64_t instance_stride) {
    if (!finalized_ || batch_ == 0) throw ZkError(ZK_ERR_INVALID, "trace_columns before set_batch");
    // a compact batch stays compact: one instance's columns are read through the trace view (cell -> slot), the whole batch's trace
    // (4x the store) is never allocated for this
    if (compact_) ensure_trace_view();
    // the last fused step wrote the narrow store and left nothing out: the columns are read from it directly (k_trace_columns_batch decodes address
    // words), the ordinary store is not expanded for them.  With deferred Poseidon2 intermediates the fill needs the ordinary store: widen, fill, read that.
    const bool from_narrow = compact_ && narrow_pending_ && !p2_pending_ && limit_ && loop_.d_aw1;
    if (!from_narrow) ensure_p2_filled(stream);
    if (n_instances == 0) return;
    if (instance >= batch_ || n_instances > batch_ - instance) throw ZkError(ZK_ERR_INVALID, "trace_columns: instance out of range");
    const uint64_t rows = (uint64_t)loop_.n_slots * limit_ + outer_.n_slots;
    if (log_n > 32 || ((uint64_t)1 << log_n) < rows) throw ZkError(ZK_ERR_INVALID, "trace_columns: 2^log_n smaller than the trace");
    if (stride < ((uint64_t)1 << log_n)) throw ZkError(ZK_ERR_INVALID, "trace_columns: stride smaller than the column");
    zkdev::ColumnsArgs a;
    if (compact_) {
        a.loop_cells = loop_.d_store; a.loop_n_cells = loop_.store_geom(); a.outer_cells = outer_.d_store; a.outer_n_cells = outer_.store_geom();
        a.loop_slot1 = loop_.d_slot1; a.outer_slot1 = outer_.d_slot1;
        if (from_narrow) { a.loop_cells = loop_.d_store_n; a.loop_n_cells = loop_.narrow_geom(); a.loop_slot1 = loop_.d_aw1; }
    } else {
        a.loop_cells = loop_.d_cells; a.loop_n_cells = loop_.n_cells; a.outer_cells = outer_.d_cells; a.outer_n_cells = outer_.n_cells;
    }
    a.n_cols = geo_.num_columns_under_copy_permutation + lookup_width_ * lookup_reps_;
    a.loop_slots = limit_ ? loop_.n_slots : 0; a.outer_slots = outer_.n_slots; a.limit = limit_; a.instance = instance;
    a.out = d_out; a.stride = stride; a.n_rows_padded = (uint64_t)1 << log_n;
    const uint64_t n_cols = a.n_cols;
    if (n_instances > 1 && instance_stride < (n_cols - 1) * stride + ((uint64_t)1 << log_n)) throw ZkError(ZK_ERR_INVALID, "trace_columns: instance stride smaller than an instance's columns");
    a.n_instances = n_instances; a.instance_stride = instance_stride;
    dev_check(zkdev::launch_trace_columns(a, stream));
}

void CS::trace_ptr(bool loop_scope, uint64_t** cells, uint64_t* n_cells, uint64_t* stride) {
    ensure_materialized(nullptr);
    const Scope& s = loop_scope ? loop_ : outer_;
    *cells = s.d_cells; *n_cells = s.n_cells; *stride = s.stride;
}

}  // namespace zkgl
