// zkgl_device.hip — the ONE device translation unit of libzkgl.so (gfx950 only).
// Kernels live in kernels_primitives.hpp / kernels_engine.hpp; this file holds the launchers.
#include <hip/hip_runtime.h>
// Several kernels size their static LDS for gfx950's 160 KB per CU (k_seed_wave 156 KB, k_multiplicities 128 KB): refuse other targets
// at compile time instead of failing at launch.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libzkgl's kernels are written for gfx950 (MI355X): LDS sizes, DPP rows and buffer addressing assume it"
#endif
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include "device_api.hpp"
#include "kernels_primitives.hpp"
#include "kernels_engine.hpp"
#include "kernels_engine2.hpp"
#include "kernels_seed_wave.hpp"
#include "kernels_vm_seed.hpp"
#include "kernels_queue_seed.hpp"
#include "kernels_fsm_seed.hpp"
#include "kernels_lookup_arg.hpp"
#include "kernels_ntt.hpp"
#include "kernels_perm.hpp"

namespace zkdev {

static thread_local std::string g_hip_err;
const char* last_hip_error() { return g_hip_err.c_str(); }

static int chk(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    g_hip_err = std::string(what) + ": " + hipGetErrorString(e);
    return -2;
}
#define LAUNCH_CHECK(what) chk(hipGetLastError(), what)

static inline unsigned grid_for(size_t n, int per_block, unsigned cap = 0x7fffffffu) {
    size_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

int upload_round_constants(const uint64_t rc[360]) {
    int e = chk(hipMemcpyToSymbol(HIP_SYMBOL(p2::RC), rc, 360 * sizeof(uint64_t)), "hipMemcpyToSymbol(RC)");
    if (e) return e;
    // inverses of 1 .. INV_SMALL_N - 1 by Montgomery's trick: prefix products, one exponentiation, back-substitution (host, once)
    static uint64_t inv[p2::INV_SMALL_N];
    auto mulm = [](uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % 0xFFFFFFFF00000001ull); };
    auto powm = [&](uint64_t a, uint64_t e2) { uint64_t r = 1; while (e2) { if (e2 & 1) r = mulm(r, a); a = mulm(a, a); e2 >>= 1; } return r; };
    static uint64_t pre[p2::INV_SMALL_N];
    pre[0] = 1;
    for (uint32_t k = 1; k < p2::INV_SMALL_N; ++k) pre[k] = mulm(pre[k - 1], k);
    uint64_t run = powm(pre[p2::INV_SMALL_N - 1], 0xFFFFFFFF00000001ull - 2);
    inv[0] = 0;
    for (uint32_t k = p2::INV_SMALL_N - 1; k >= 1; --k) { inv[k] = mulm(run, pre[k - 1]); run = mulm(run, k); }
    return chk(hipMemcpyToSymbol(HIP_SYMBOL(p2::INV_SMALL), inv, sizeof inv), "hipMemcpyToSymbol(INV_SMALL)");
}

int launch_col(int op, uint64_t* dst, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t q, uint64_t l,
               size_t n, void* stream) {
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // memory-bound: cap at 256 CUs x 8 workgroups and grid-stride the rest
    unsigned g = grid_for(n / 2 + 1, zkk::TPB, 2048);
    switch (op) {
    case zkk::COL_FMA: zkk::k_col<zkk::COL_FMA><<<g, zkk::TPB, 0, s>>>(dst, a, b, c, q, l, n); break;
    case zkk::COL_ADD: zkk::k_col<zkk::COL_ADD><<<g, zkk::TPB, 0, s>>>(dst, a, b, c, q, l, n); break;
    case zkk::COL_SUB: zkk::k_col<zkk::COL_SUB><<<g, zkk::TPB, 0, s>>>(dst, a, b, c, q, l, n); break;
    case zkk::COL_MUL: zkk::k_col<zkk::COL_MUL><<<g, zkk::TPB, 0, s>>>(dst, a, b, c, q, l, n); break;
    case zkk::COL_SELECT: zkk::k_col<zkk::COL_SELECT><<<g, zkk::TPB, 0, s>>>(dst, a, b, c, q, l, n); break;
    case zkk::COL_INV: zkk::k_col<zkk::COL_INV><<<g, zkk::TPB, 0, s>>>(dst, a, b, c, q, l, n); break;
    default: g_hip_err = "bad column op"; return -1;
    }
    return LAUNCH_CHECK("k_col");
}

int launch_poseidon2_soa(uint64_t* st, size_t n, size_t stride, void* stream) {
    if (n == 0) return 0;
    zkk::k_poseidon2_soa<<<grid_for(n, zkk::TPB), zkk::TPB, 0, (hipStream_t)stream>>>(st, n, stride);
    return LAUNCH_CHECK("k_poseidon2_soa");
}
int launch_poseidon2_aos(uint64_t* st, size_t n, void* stream) {
    if (n == 0) return 0;
    zkk::k_poseidon2_aos<<<grid_for(n, zkk::TPB), zkk::TPB, 0, (hipStream_t)stream>>>(st, n);
    return LAUNCH_CHECK("k_poseidon2_aos");
}
int launch_commit_encoding(const uint64_t* in, size_t len, size_t n, uint64_t* out, void* stream) {
    if (n == 0) return 0;
    zkk::k_commit_encoding<<<grid_for(n, zkk::TPB), zkk::TPB, 0, (hipStream_t)stream>>>(in, len, n, out);
    return LAUNCH_CHECK("k_commit_encoding");
}
int launch_queue_full_chain(const uint64_t* enc, size_t nq, size_t items, uint64_t* tail_io, uint64_t* states_out,
                            void* stream) {
    if (nq == 0) return 0;
    zkk::k_queue_full_chain<<<grid_for(nq, 64), 64, 0, (hipStream_t)stream>>>(enc, nq, items, tail_io, states_out);
    return LAUNCH_CHECK("k_queue_full_chain");
}
int launch_memory_query_encode(const uint64_t* q, size_t n, uint64_t* enc, void* stream) {
    if (n == 0) return 0;
    zkk::k_memory_query_encode<<<grid_for(n, zkk::TPB), zkk::TPB, 0, (hipStream_t)stream>>>(q, n, enc);
    return LAUNCH_CHECK("k_memory_query_encode");
}
int launch_execution_context_encode(const uint64_t* rec, size_t n, uint64_t* enc, void* stream) {
    if (n == 0) return 0;
    zkk::k_execution_context_encode<<<grid_for(n, zkk::TPB), zkk::TPB, 0, (hipStream_t)stream>>>(rec, n, enc);
    return LAUNCH_CHECK("k_execution_context_encode");
}
int launch_grand_product(const uint64_t* enc, const uint64_t* flags, const uint64_t* ch, size_t enc_len, size_t n,
                         uint64_t init, uint64_t* acc, uint64_t* scratch, void* stream) {
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    size_t ntiles = (n + zkk::GP_TILE - 1) / zkk::GP_TILE;
    zkk::k_gp_local<<<(unsigned)ntiles, zkk::TPB, 0, s>>>(enc, flags, ch, enc_len, n, acc, scratch);
    zkk::k_gp_tiles<<<1, zkk::TPB, 0, s>>>(scratch, ntiles, init);
    zkk::k_gp_apply<<<(unsigned)ntiles, zkk::TPB, 0, s>>>(acc, scratch, n);
    return LAUNCH_CHECK("k_gp");
}

// the buffer-addressed kernels put slot << (T + 3) in a 32-bit soffset (store_geom.hpp); larger stores take 64-bit addresses
static bool needs_wide_addressing(uint64_t geom) { return zkgeom::slots(geom) >= (1ull << (29 - zkgeom::tile_log2(geom))); }

static zke::ScopeDev to_dev(const ScopeArgs& a) {
    zke::ScopeDev d;
    d.prog = a.prog; d.n_words = a.n_words; d.n_lanes = a.n_lanes; d.consts = a.consts; d.cells = a.cells;
    d.n_cells = a.n_cells; d.inputs = a.inputs; d.in_stride = a.in_stride ? a.in_stride : a.n_lanes; d.outer_cells = a.outer_cells; d.outer_n_cells = a.outer_n_cells;
    d.limit = a.limit; d.is_loop = a.is_loop; d.tables = a.tables; d.table_words = a.table_words; d.mult = a.mult;
    d.total_table_rows = a.total_table_rows; d.loop_cells = a.loop_cells; d.loop_n_cells = a.loop_n_cells;
    d.loop_limit = a.loop_limit;
    d.fail = a.fail;
    d.clock_probe = a.clock_probe;
    d.p2_stats = a.p2_stats;
    d.defer_p2 = a.defer_p2;
    return d;
}

// Occupancy experiments: unused dynamic LDS per workgroup caps the workgroups resident on a CU (fewer wavefronts = a larger share of
// L2 per wavefront against less latency hiding).  Bytes from the environment, 0 by default.
static unsigned lds_pad(const char* name) {
    const char* v = std::getenv(name);
    return v ? (unsigned)std::strtoul(v, nullptr, 10) : 0u;
}

int launch_witness(const ScopeArgs& sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin, void* stream) {
    // sc.prog = the scope's v2 program (kernels_engine2.hpp), slot_begin = the store slot of the first output of word_begin
    if (sc.n_lanes == 0 || word_begin >= word_end) return 0;
    const dim3 grid = grid_for(sc.n_lanes, zke::TPB);
    hipStream_t s = (hipStream_t)stream;
    if (sc.cls) {   // narrow store (store_geom.hpp): the host offers it to the plain loop kernel only (cs.cpp narrow_usable)
        if (!sc.is_loop || sc.uses_bigint || needs_wide_addressing(sc.n_cells) || !zkgeom::narrow(sc.n_cells)) { g_hip_err = "launch_witness: a narrow store outside the plain loop kernel"; return -1; }
        zke::k_witness_loop_narrow<<<grid, zke::TPB, lds_pad("ZKGL_LOOP_LDS_PAD"), s>>>(to_dev(sc), sc.cls, word_begin, word_end, slot_begin);
    }
    else if (sc.xmacros) {   // a circuit that records ZK_OP_SHA256_ROUNDS a = 1 / ZK_OP_BYTEBUF_FILL: the kernels that carry that backend
        if (needs_wide_addressing(sc.n_cells)) { g_hip_err = "launch_witness: the 4-bit SHA / ByteBuffer macro-ops are not offered to scopes with 64-bit store addressing"; return -1; }
        if (sc.xmacros == zke::X_SHA4) zke::k_witness_plain_x<zke::X_SHA4><<<grid, zke::TPB, 0, s>>>(to_dev(sc), word_begin, word_end, slot_begin);
        else if (sc.xmacros == zke::X_BYTEBUF) zke::k_witness_plain_x<zke::X_BYTEBUF><<<grid, zke::TPB, 0, s>>>(to_dev(sc), word_begin, word_end, slot_begin);
        else { g_hip_err = "launch_witness: no kernel carries this combination of macro-op backends"; return -1; }
    }
    else if (needs_wide_addressing(sc.n_cells)) zke::k_witness_wide<<<grid, zke::TPB, 0, s>>>(to_dev(sc), word_begin, word_end, slot_begin);
    else if (sc.is_loop && sc.uses_bigint) zke::k_witness_loop_bigint<<<grid, zke::TPB, 0, s>>>(to_dev(sc), word_begin, word_end, slot_begin);
    else if (sc.is_loop) zke::k_witness_loop<<<grid, zke::TPB, lds_pad("ZKGL_LOOP_LDS_PAD"), s>>>(to_dev(sc), word_begin, word_end, slot_begin);
    else if (sc.uses_bigint) zke::k_witness_outer_bigint<<<grid, zke::TPB, 0, s>>>(to_dev(sc), word_begin, word_end, slot_begin);
    else zke::k_witness_outer<<<grid, zke::TPB, 0, s>>>(to_dev(sc), word_begin, word_end, slot_begin);
    return LAUNCH_CHECK("k_witness");
}

int launch_witness_strands(const ScopeArgs& sc, const uint32_t begin[STRANDS_PER_TILE], const uint32_t end[STRANDS_PER_TILE], void* stream, uint32_t n_strands) {
    if (sc.n_lanes == 0) return 0;
    if (n_strands == 0 || n_strands > STRANDS_PER_TILE) { g_hip_err = "launch_witness_strands: strand count"; return -1; }
    const unsigned block = 64 * n_strands;   // one wavefront per strand; the barrier of a level spans exactly these
    static_assert(zke::STRANDS_PER_TILE == (int)STRANDS_PER_TILE, "strand count");
    zke::StrandTab tab;
    bool any = false;
    for (int i = 0; i < zke::STRANDS_PER_TILE; ++i) { tab.begin[i] = begin[i]; tab.end[i] = end[i]; any |= end[i] > begin[i]; }
    if (!any) return 0;
    const unsigned grid = grid_for(sc.n_lanes, 64);
    if (sc.xmacros) {
        if (needs_wide_addressing(sc.n_cells)) { g_hip_err = "launch_witness_strands: the 4-bit SHA / ByteBuffer macro-ops are not offered to scopes with 64-bit store addressing"; return -1; }
        if (sc.xmacros == zke::X_SHA4) zke::k_witness_strands2<true, false, zke::X_SHA4><<<grid, block, 0, (hipStream_t)stream>>>(to_dev(sc), tab);
        else if (sc.xmacros == zke::X_BYTEBUF) zke::k_witness_strands2<true, false, zke::X_BYTEBUF><<<grid, block, 0, (hipStream_t)stream>>>(to_dev(sc), tab);
        else { g_hip_err = "launch_witness_strands: no kernel carries this combination of macro-op backends"; return -1; }
    }
    else if (needs_wide_addressing(sc.n_cells)) zke::k_witness_strands2<true, true><<<grid, block, 0, (hipStream_t)stream>>>(to_dev(sc), tab);  // 64-bit addressing
    else if (sc.uses_bigint) zke::k_witness_strands2<true, false><<<grid, block, 0, (hipStream_t)stream>>>(to_dev(sc), tab);
    else zke::k_witness_strands2<false, false><<<grid, block, 0, (hipStream_t)stream>>>(to_dev(sc), tab);
    return LAUNCH_CHECK("k_witness_strands");
}

int launch_witness_seq(const ScopeArgs& sc, const CarryArgs* d_carries, uint32_t n_carries, uint64_t* inputs_rw,
                       uint32_t n_instances, void* stream) {
    if (n_instances == 0 || sc.limit == 0) return 0;
    static_assert(sizeof(CarryArgs) == sizeof(zke::CarryDev), "CarryArgs layout");
    if (sc.uses_bigint)
        zke::k_witness_seq<true><<<grid_for(n_instances, 64), 64, 0, (hipStream_t)stream>>>(
            to_dev(sc), reinterpret_cast<const zke::CarryDev*>(d_carries), n_carries, inputs_rw, n_instances);
    else
        zke::k_witness_seq<false><<<grid_for(n_instances, 64), 64, 0, (hipStream_t)stream>>>(
            to_dev(sc), reinterpret_cast<const zke::CarryDev*>(d_carries), n_carries, inputs_rw, n_instances);
    return LAUNCH_CHECK("k_witness_seq");
}

int launch_check_gates(const CheckArgs& a, void* stream) {
    if (a.n_lanes == 0 || a.n_slots == 0) return 0;
    zke::CheckDev d;
    d.cells = a.cells; d.n_cells = a.n_cells; d.n_cols = a.n_cols; d.n_lanes = a.n_lanes; d.n_slots = a.n_slots; d.rows = a.rows;
    d.rowconsts = a.rowconsts; d.lrows = a.lrows; d.n_copy_cols = a.n_copy_cols; d.lookup_width = a.lookup_width;
    d.tables = a.tables; d.table_words = a.table_words; d.fail = a.fail; d.slots_per_chunk = a.slots_per_chunk; d.alias = a.alias;
    if (a.alias && a.cprog && a.n_chunks && !needs_wide_addressing(a.n_cells)) {
        zke::CheckProgDev p;
        p.cells = a.cells; p.n_cells = a.n_cells; p.n_lanes = a.n_lanes; p.prog = a.cprog; p.chunk_tab = a.chunk_tab; p.n_chunks = a.n_chunks;
        p.rowconsts = a.rowconsts; p.tables = a.tables; p.table_words = a.table_words; p.fail = a.fail;
        const unsigned lane_tiles = grid_for(a.n_lanes, zke::TPB);
        const uint32_t wanted = std::max<uint32_t>(2, (2048 + lane_tiles - 1) / lane_tiles);  // >= ~2048 workgroups
        p.chunks_per_block = std::max<uint32_t>(1, a.n_chunks / wanted);
        dim3 grid(lane_tiles, (a.n_chunks + p.chunks_per_block - 1) / p.chunks_per_block);
        if (zkgeom::narrow(a.n_cells)) zke::k_check_prog_t<true><<<grid, zke::TPB, lds_pad("ZKGL_CHECK_LDS_PAD"), (hipStream_t)stream>>>(p);
        else zke::k_check_prog_t<false><<<grid, zke::TPB, lds_pad("ZKGL_CHECK_LDS_PAD"), (hipStream_t)stream>>>(p);
        if (a.macros && a.n_macros) {
            if (zkgeom::narrow(a.n_cells)) { g_hip_err = "launch_check_gates: Poseidon2 macro packets over a narrow store"; return -1; }
            zke::CheckP2Dev m;
            m.cells = a.cells; m.n_cells = a.n_cells; m.n_lanes = a.n_lanes; m.macros = a.macros; m.n_macros = a.n_macros; m.fail = a.fail;
            const uint32_t want_y = std::max<uint32_t>(1, std::min<uint32_t>(a.n_macros, (2048 + lane_tiles - 1) / lane_tiles));
            m.per_block = (a.n_macros + want_y - 1) / want_y;
            dim3 g2(lane_tiles, (a.n_macros + m.per_block - 1) / m.per_block);
            zke::k_check_p2<<<g2, zke::TPB, lds_pad("ZKGL_CHECK_P2_LDS_PAD"), (hipStream_t)stream>>>(m);
        }
        return LAUNCH_CHECK("k_check_prog");
    }
    if (zkgeom::narrow(a.n_cells)) { g_hip_err = "launch_check_gates: only the check program reads a narrow store"; return -1; }
    dim3 grid(grid_for(a.n_lanes, zke::TPB), (a.n_slots + a.slots_per_chunk - 1) / a.slots_per_chunk);
    if (a.alias) zke::k_check_gates_compact<<<grid, zke::TPB, 0, (hipStream_t)stream>>>(d);
    else zke::k_check_gates<<<grid, zke::TPB, 0, (hipStream_t)stream>>>(d);
    return LAUNCH_CHECK("k_check_gates");
}

int launch_pack_public(const uint64_t* outer_store, uint64_t n_store, const uint32_t* slots, uint32_t n_public, uint32_t n_instances, uint64_t* out, void* stream) {
    if (!n_public || !n_instances) return 0;
    zke::k_pack_public<<<grid_for((size_t)n_public * n_instances, 256), 256, 0, (hipStream_t)stream>>>(outer_store, n_store, slots, n_public, n_instances, out);
    return LAUNCH_CHECK("k_pack_public");
}

int launch_hook_compare(const uint64_t* outer_store, uint64_t n_store, const uint32_t* slots, uint32_t n_vars, uint32_t n_instances, const uint64_t* expected,
                        unsigned long long* fail, void* stream) {
    if (!n_vars || !n_instances) return 0;
    zke::k_hook_compare<<<grid_for((size_t)n_vars * n_instances, 256), 256, 0, (hipStream_t)stream>>>(outer_store, n_store, slots, n_vars, n_instances, expected, fail);
    return LAUNCH_CHECK("k_hook_compare");
}

int launch_check_inputs(const uint64_t* inputs, uint32_t n_words, uint32_t n_lanes, uint64_t stride, unsigned long long* fail, void* stream) {
    if (!n_words || !n_lanes) return 0;
    dim3 grid(grid_for(n_lanes, 256), std::min<uint32_t>(n_words, 64));
    zke::k_check_inputs<<<grid, 256, 0, (hipStream_t)stream>>>(inputs, n_words, n_lanes, stride, fail);
    return LAUNCH_CHECK("k_check_inputs");
}

int launch_multiplicities(const uint64_t* store, uint64_t n_store, uint32_t lanes_per_instance, uint32_t n_lanes, uint32_t n_instances, const uint32_t* sites,
                          uint32_t n_sites, const zk_table_desc& t, const uint64_t* table_words, uint32_t* mult, uint32_t total_table_rows, void* stream) {
    if (!n_sites || !n_instances || !t.n_rows) return 0;
    zke::MultDev a;
    a.store = store; a.n_store = n_store; a.lanes_per_instance = lanes_per_instance; a.n_lanes = n_lanes; a.sites = sites; a.n_sites = n_sites;
    a.t = t; a.table_words = table_words; a.mult = mult; a.total_table_rows = total_table_rows;
    const bool packed = t.n_rows > zke::MULT_CHUNK_ROWS && t.n_rows <= 2 * zke::MULT_CHUNK_ROWS;   // 16-bit LDS counters: one workgroup per 65 536-row table
    a.chunk_rows = packed ? 2 * zke::MULT_CHUNK_ROWS : zke::MULT_CHUNK_ROWS;
    const uint32_t chunks = (t.n_rows + a.chunk_rows - 1) / a.chunk_rows;
    // >= ~512 workgroups: few instances share each one among several workgroups (lane ranges)
    uint32_t splits = std::max<uint32_t>(1, 512 / std::max<uint32_t>(1, chunks * n_instances));
    splits = std::min<uint32_t>(splits, std::max<uint32_t>(1, lanes_per_instance / 256));
    // one wavefront per site at a time: small site lists (outer scopes) do not need 16 wavefronts
    const unsigned threads = n_sites >= 16 ? 1024 : 256;
    // still short of ~512 workgroups (eip_4844: 8 blobs x 2 chunks x 3 lane ranges = 48 on 256 CUs): split the sites as well, each
    // workgroup keeping at least two rounds of sites for its wavefronts
    uint32_t site_splits = std::max<uint32_t>(1, 512 / std::max<uint32_t>(1, chunks * n_instances * splits));
    site_splits = std::min<uint32_t>(site_splits, std::max<uint32_t>(1, n_sites / (2 * (threads / 64))));
    site_splits = std::min<uint32_t>(site_splits, 65535u / std::max<uint32_t>(1, splits));
    a.site_splits = site_splits;
    dim3 grid(chunks, n_instances, splits * site_splits);
    if (packed) zke::k_multiplicities<true><<<grid, threads, 0, (hipStream_t)stream>>>(a);
    else zke::k_multiplicities<false><<<grid, threads, 0, (hipStream_t)stream>>>(a);
    return LAUNCH_CHECK("k_multiplicities");
}
int launch_materialize(uint64_t* trace, uint64_t n_cells, const uint64_t* store, uint64_t n_store, uint32_t n_lanes, const zk_copy_pair* pairs,
                       uint32_t n_pairs, void* stream) {
    if (n_lanes == 0 || n_pairs == 0) return 0;
    unsigned lane_tiles = grid_for(n_lanes, zke::TPB);
    uint32_t chunks = std::max<uint32_t>(2, (2048 + lane_tiles - 1) / lane_tiles);
    if (chunks > n_pairs) chunks = n_pairs;
    uint32_t per = (n_pairs + chunks - 1) / chunks;
    dim3 grid(lane_tiles, (n_pairs + per - 1) / per);
    zke::k_materialize<<<grid, zke::TPB, 0, (hipStream_t)stream>>>(trace, n_cells, store, n_store, n_lanes, pairs, n_pairs, per);
    return LAUNCH_CHECK("k_materialize");
}

int launch_check_copies(const uint64_t* cells, uint64_t n_cells, uint32_t n_lanes, const zk_copy_pair* pairs,
                        uint32_t n_pairs, unsigned long long* fail, void* stream) {
    if (n_lanes == 0 || n_pairs == 0) return 0;
    unsigned lane_tiles = grid_for(n_lanes, zke::TPB);
    // aim for >= ~2048 workgroups
    uint32_t chunks = std::max<uint32_t>(2, (2048 + lane_tiles - 1) / lane_tiles);  // never one chunk: see CS::check_args
    if (chunks > n_pairs) chunks = n_pairs;
    if (chunks < 1) chunks = 1;
    uint32_t per = (n_pairs + chunks - 1) / chunks;
    dim3 grid(lane_tiles, (n_pairs + per - 1) / per);
    zke::k_check_copies<<<grid, zke::TPB, 0, (hipStream_t)stream>>>(cells, n_cells, n_lanes, pairs, n_pairs, per, fail);
    return LAUNCH_CHECK("k_check_copies");
}

int launch_ntt_pass(const NttPassArgs& a, uint32_t n_polys, void* stream) {
    if (n_polys == 0) return 0;
    zkn::PassDev d;
    d.src = a.src; d.dst = a.dst; d.src_stride = a.src_stride; d.dst_stride = a.dst_stride;
    d.log_n = a.log_n; d.seg = a.seg; d.r = a.r; d.t = a.t; d.dit = a.dit; d.coset_store = a.coset_store; d.coset_brev = a.coset_brev;
    d.root1024 = a.root1024; d.tw_lo = a.tw_lo; d.tw_hi = a.tw_hi; d.c_lo = a.c_lo; d.c_hi = a.c_hi;
    const uint64_t blocks = (uint64_t)n_polys << (a.log_n - a.r - a.t);
    if (blocks > 0x7fffffffull) { g_hip_err = "k_ntt_pass: grid too large"; return -2; }
    const size_t lds = sizeof(uint64_t) * (512 + zkn::padded_elems(1u << (a.r + a.t)));
    static bool lds_raised = false;  // > 64 KB of dynamic LDS needs the opt-in (gfx950: 160 KB per CU)
    if (!lds_raised) {
        if (int rc = chk(hipFuncSetAttribute((const void*)zkn::k_ntt_pass, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "k_ntt_pass LDS"))
            return rc;
        lds_raised = true;
    }
    zkn::k_ntt_pass<<<(unsigned)blocks, zkn::TPB, lds, (hipStream_t)stream>>>(d);
    return LAUNCH_CHECK("k_ntt_pass");
}

int launch_widen_store(const uint64_t* narrow, uint64_t narrow_geom, uint64_t* wide, uint64_t wide_geom, uint32_t n_lanes, const uint32_t* aw, uint32_t n_slots, void* stream) {
    if (!n_lanes || !n_slots) return 0;
    if (!zkgeom::narrow(narrow_geom) || zkgeom::narrow(wide_geom)) { g_hip_err = "launch_widen_store: geometry words"; return -1; }
    const unsigned lane_tiles = grid_for(n_lanes, zke::TPB);
    uint32_t chunks = std::max<uint32_t>(1, std::min<uint32_t>(n_slots, (4096 + lane_tiles - 1) / lane_tiles));
    const uint32_t per = (n_slots + chunks - 1) / chunks;
    chunks = (n_slots + per - 1) / per;
    zke::k_widen_store<<<dim3(lane_tiles, chunks), zke::TPB, 0, (hipStream_t)stream>>>(narrow, narrow_geom, wide, wide_geom, n_lanes, aw, n_slots, per);
    return LAUNCH_CHECK("k_widen_store");
}

int launch_widen_last(const uint64_t* narrow, uint64_t narrow_geom, uint64_t* wide, uint64_t wide_geom, uint32_t n_instances, uint32_t limit, const uint32_t* aw,
                      const uint32_t* slots, uint32_t n_list, void* stream) {
    if (!n_instances || !n_list || !limit) return 0;
    if (!zkgeom::narrow(narrow_geom) || zkgeom::narrow(wide_geom)) { g_hip_err = "launch_widen_last: geometry words"; return -1; }
    zke::k_widen_last<<<grid_for((size_t)n_instances * n_list, 256), 256, 0, (hipStream_t)stream>>>(narrow, narrow_geom, wide, wide_geom, n_instances, limit, aw, slots, n_list);
    return LAUNCH_CHECK("k_widen_last");
}

int launch_fill_p2(uint64_t* store, uint64_t n_store, uint32_t n_lanes, const uint32_t* macros, uint32_t n_macros, void* stream) {
    if (!n_lanes || !n_macros) return 0;
    zke::CheckP2Dev d;
    d.cells = store; d.n_cells = n_store; d.n_lanes = n_lanes; d.macros = macros; d.n_macros = n_macros; d.fail = nullptr;
    const unsigned lane_tiles = grid_for(n_lanes, zke::TPB);
    uint32_t chunks = std::max<uint32_t>(1, std::min<uint32_t>(n_macros, (4096 + lane_tiles - 1) / lane_tiles));
    d.per_block = (n_macros + chunks - 1) / chunks;
    chunks = (n_macros + d.per_block - 1) / d.per_block;
    zke::k_fill_p2<<<dim3(lane_tiles, chunks), zke::TPB, 0, (hipStream_t)stream>>>(d);
    return LAUNCH_CHECK("k_fill_p2");
}
int launch_perm_lane(const PermArgs& a, void* stream) {
    if (a.n_lanes == 0) return 0;
    zkp::PermDev d;
    d.cells = a.cells; d.n_cells = a.n_cells; d.n_cols = a.n_cols; d.n_lanes = a.n_lanes; d.n_slots = a.n_slots;
    d.n_copy_cols = a.n_copy_cols; d.lookup_width = a.lookup_width; d.rows = a.rows; d.lrows = a.lrows;
    d.sigma_rel = a.sigma_rel; d.ep_index = a.ep_index; d.ovr = a.ovr; d.lanes_per_instance = a.lanes_per_instance;
    d.label_base = a.label_base; d.label_step = a.label_step; d.tb = a.tb;
    d.beta = {a.beta[0], a.beta[1]}; d.gamma = {a.gamma[0], a.gamma[1]};
    d.slots_per_chunk = a.slots_per_chunk; d.n_chunks = a.n_chunks; d.lane_out = a.lane_out; d.prefix = a.prefix; d.slot1 = a.slot1;
    zkp::k_perm_lane<<<dim3(grid_for(a.n_lanes, zkp::TPB), a.n_chunks), zkp::TPB, 0, (hipStream_t)stream>>>(d);
    return LAUNCH_CHECK("k_perm_lane");
}
int launch_perm_tb(const uint64_t beta[2], const uint32_t* sigma_rel, uint64_t* tb, uint32_t n, void* stream) {
    if (n == 0) return 0;
    zkp::k_perm_tb<<<grid_for(n, zkp::TPB), zkp::TPB, 0, (hipStream_t)stream>>>(zkl::E{beta[0], beta[1]}, sigma_rel, tb, n);
    return LAUNCH_CHECK("k_perm_tb");
}
int launch_perm_scan(const uint64_t* part, uint32_t per, const uint64_t* seed, uint64_t* excl, uint64_t* total, uint32_t n_instances, void* stream) {
    if (n_instances == 0) return 0;
    zkp::k_perm_scan<<<n_instances, zkp::TPB, 0, (hipStream_t)stream>>>(part, per, seed, excl, total);
    return LAUNCH_CHECK("k_perm_scan");
}
int launch_perm_z(const uint64_t* excl, const uint64_t* prefix, uint32_t n_lanes, uint32_t n_slots, uint32_t slots_per_chunk, uint32_t n_chunks,
                  uint32_t lanes_per_instance, uint64_t row_base, uint64_t rows_per_instance, uint64_t* z, void* stream) {
    if (n_lanes == 0) return 0;
    zkp::k_perm_z<<<dim3(grid_for(n_lanes, zkp::TPB), n_chunks), zkp::TPB, 0, (hipStream_t)stream>>>(excl, prefix, n_lanes, n_slots, slots_per_chunk,
                                                                                                  n_chunks, lanes_per_instance, row_base, rows_per_instance, z);
    return LAUNCH_CHECK("k_perm_z");
}

int launch_trace_columns(const ColumnsArgs& a, void* stream) {
    zkn::ColumnsDev d;
    d.loop_cells = a.loop_cells; d.loop_n_cells = a.loop_n_cells; d.outer_cells = a.outer_cells; d.outer_n_cells = a.outer_n_cells;
    d.n_cols = a.n_cols; d.loop_slots = a.loop_slots; d.outer_slots = a.outer_slots; d.limit = a.limit; d.instance = a.instance;
    d.out = a.out; d.stride = a.stride; d.n_rows_padded = a.n_rows_padded;
    d.loop_slot1 = a.loop_slot1; d.outer_slot1 = a.outer_slot1;
    if (a.limit && a.loop_slots) {
        if (zkgeom::tile_log2(a.loop_n_cells) < 6) { g_hip_err = "k_trace_columns_batch: store tiles narrower than a wavefront"; return -2; }
        const uint64_t lane0 = (uint64_t)a.instance * a.limit, lane1 = (uint64_t)(a.instance + a.n_instances) * a.limit;
        const uint32_t first_tile = (uint32_t)(lane0 >> 6), n_tiles = (uint32_t)(((lane1 + 63) >> 6) - first_tile);
        const uint32_t slot_groups = (a.loop_slots + 63) / 64;
        const uint64_t work = (uint64_t)n_tiles * slot_groups * a.n_cols;
        // launches of at most 2^30 blocks (grid.x is a 31-bit count): whole tiles per launch
        const uint32_t tiles_per_launch = (uint32_t)std::max<uint64_t>(1, ((uint64_t)1 << 30) / ((uint64_t)slot_groups * a.n_cols));
        (void)work;
        for (uint32_t t0 = 0; t0 < n_tiles; t0 += tiles_per_launch) {
            const uint32_t nt = std::min(tiles_per_launch, n_tiles - t0);
            const uint64_t w = (uint64_t)nt * slot_groups * a.n_cols;
            const uint32_t blocks = (uint32_t)((w + 7) / 8 * 8);
            if (zkgeom::narrow(a.loop_n_cells)) zkn::k_trace_columns_batch_t<true><<<blocks, 256, 0, (hipStream_t)stream>>>(d, first_tile + t0, nt, slot_groups, a.instance_stride, a.instance, a.n_instances);
            else zkn::k_trace_columns_batch_t<false><<<blocks, 256, 0, (hipStream_t)stream>>>(d, first_tile + t0, nt, slot_groups, a.instance_stride, a.instance, a.n_instances);
            if (int rc = LAUNCH_CHECK("k_trace_columns_batch")) return rc;
        }
    }
    const uint64_t tail = a.n_rows_padded - (uint64_t)a.limit * a.loop_slots;
    if (tail) {
        for (uint32_t i0 = 0; i0 < a.n_instances; i0 += 65535) {
            zkn::ColumnsDev dd = d;
            dd.instance = a.instance + i0; dd.out = a.out + (size_t)i0 * a.instance_stride;
            dim3 grid((unsigned)((tail + 255) / 256), a.n_cols, std::min<uint32_t>(65535, a.n_instances - i0));
            zkn::k_trace_columns_tail<<<grid, 256, 0, (hipStream_t)stream>>>(dd, a.instance_stride);
            if (int rc = LAUNCH_CHECK("k_trace_columns_tail")) return rc;
        }
    }
    return 0;
}

int launch_coset_tables(uint64_t base, uint64_t scale, uint64_t* c_lo, uint64_t* c_hi, uint32_t n_hi, void* stream) {
    zkn::k_coset_tables<<<grid_for(1024 + n_hi, 256), 256, 0, (hipStream_t)stream>>>(base, scale, c_lo, c_hi, n_hi);
    return LAUNCH_CHECK("k_coset_tables");
}

int launch_lookup_arg_witness(const LookupArgArgs& a, const uint64_t ch[10], void* stream) {
    if (a.n_lanes == 0) return 0;
    zkl::LookupArgDev d;
    d.cells = a.cells; d.n_cells = a.n_cells; d.n_cols = a.n_cols; d.n_lanes = a.n_lanes; d.n_slots = a.n_slots;
    d.n_copy_cols = a.n_copy_cols; d.lookup_width = a.lookup_width; d.lrows = a.lrows; d.acc = a.acc;
    d.beta = {ch[0], ch[1]}; d.g1 = {ch[2], ch[3]}; d.g2 = {ch[4], ch[5]}; d.g3 = {ch[6], ch[7]}; d.g4 = {ch[8], ch[9]};
    zkl::k_lookup_arg_witness<<<grid_for(a.n_lanes, zkl::TPB), zkl::TPB, 0, (hipStream_t)stream>>>(d);
    return LAUNCH_CHECK("k_lookup_arg_witness");
}

int launch_lookup_arg_tables(const zk_table_desc* tables, uint32_t n_tables, const uint64_t* table_words, uint32_t total_rows, uint32_t lookup_width,
                             const uint64_t ch[10], uint64_t* inv_f, const uint32_t* mult, uint32_t n_instances, uint64_t* out_b, void* stream) {
    if (n_instances == 0) return 0;
    if (total_rows) {
        zkl::TableArgDev d;
        d.tables = tables; d.n_tables = n_tables; d.table_words = table_words; d.total_rows = total_rows; d.inv_f = inv_f;
        d.beta = {ch[0], ch[1]}; d.g1 = {ch[2], ch[3]}; d.g2 = {ch[4], ch[5]}; d.g3 = {ch[6], ch[7]}; d.g4 = {ch[8], ch[9]};
        d.lookup_width = lookup_width;
        zkl::k_lookup_arg_table_rows<<<grid_for(total_rows, zkl::TPB), zkl::TPB, 0, (hipStream_t)stream>>>(d);
    }
    zkl::k_lookup_arg_table_sum<<<n_instances, zkl::TPB, 0, (hipStream_t)stream>>>(mult, inv_f, total_rows, out_b);
    return LAUNCH_CHECK("k_lookup_arg_table");
}

int launch_lookup_arg_witness_sum(const uint64_t* acc_outer, const uint64_t* acc_loop, uint32_t limit, uint32_t n_instances,
                                  uint64_t* out_a, void* stream) {
    if (n_instances == 0) return 0;
    zkl::k_lookup_arg_witness_sum<<<n_instances, zkl::TPB, 0, (hipStream_t)stream>>>(acc_outer, acc_loop, limit, out_a);
    return LAUNCH_CHECK("k_lookup_arg_witness_sum");
}

uint32_t seed_cone_max_slots() { return zke::SEED_LDS_WORDS; }

int launch_seed_cone(const ScopeArgs& sc, const uint32_t* seed_prog, uint32_t n_words, uint32_t n_slots, uint32_t n_input_words,
                     const CarryArgs* d_carries, uint32_t n_carries, uint64_t* inputs_rw, uint32_t n_instances, void* stream) {
    if (n_instances == 0 || sc.limit == 0) return 0;
    static_assert(sizeof(CarryArgs) == sizeof(zke::SeedCarryDev), "CarryArgs layout");
    const uint32_t per_lane = n_slots + n_input_words;  // LDS words one instance needs
    if (n_slots == 0 || per_lane > zke::SEED_LDS_WORDS) return -1;
    uint32_t lpb = 1;
    while (lpb < 16 && per_lane * (lpb * 2) <= zke::SEED_LDS_WORDS) lpb *= 2;  // instances per 64-thread block
    const unsigned grid = (n_instances + lpb - 1) / lpb;
    auto c = reinterpret_cast<const zke::SeedCarryDev*>(d_carries);
    if (sc.uses_bigint)
        zke::k_seed_cone<true><<<grid, 64, 0, (hipStream_t)stream>>>(to_dev(sc), seed_prog, n_words, c, n_carries, inputs_rw, n_instances, lpb, n_slots, n_input_words);
    else
        zke::k_seed_cone<false><<<grid, 64, 0, (hipStream_t)stream>>>(to_dev(sc), seed_prog, n_words, c, n_carries, inputs_rw, n_instances, lpb, n_slots, n_input_words);
    return LAUNCH_CHECK("k_seed_cone");
}

int launch_seed_cone_strands(const ScopeArgs& sc, const uint32_t* seed_sprog, const uint32_t begin[STRANDS_PER_TILE], const uint32_t end[STRANDS_PER_TILE], uint32_t n_slots,
                             uint32_t n_input_words, const CarryArgs* d_carries, uint32_t n_carries, uint64_t* inputs_rw, uint32_t n_instances, bool v2, void* stream) {
    if (n_instances == 0 || sc.limit == 0) return 0;
    const uint32_t per_lane = n_slots + n_input_words;
    if (n_slots == 0 || per_lane > zke::SEED_LDS_WORDS) return -1;
    uint32_t lpb = 1;
    while (lpb < 16 && per_lane * (lpb * 2) <= zke::SEED_LDS_WORDS) lpb *= 2;
    // few instances: spread them over more blocks (each block is a latency chain of sc.limit iterations)
    while (lpb > 1 && (n_instances + lpb - 1) / lpb < 128 && (n_instances + lpb / 2 - 1) / (lpb / 2) <= 256) lpb /= 2;
    const unsigned grid = (n_instances + lpb - 1) / lpb;
    zke::StrandTab tab;
    for (int i = 0; i < zke::STRANDS_PER_TILE; ++i) { tab.begin[i] = i < zke::SEED_STRANDS_PER_TILE ? begin[i] : 0; tab.end[i] = i < zke::SEED_STRANDS_PER_TILE ? end[i] : 0; }
    auto c = reinterpret_cast<const zke::SeedCarryDev*>(d_carries);
    if (v2)  // scalar-decoded (kernels_engine2.hpp); the host selects it when every op of the cone has a handler there
        zke::k_seed_cone_strands2<<<grid, 64 * zke::SEED_STRANDS_PER_TILE, 0, (hipStream_t)stream>>>(to_dev(sc), seed_sprog, tab, c, n_carries, inputs_rw, n_instances, lpb, n_slots, n_input_words);
    else if (sc.uses_bigint)
        zke::k_seed_cone_strands<true><<<grid, 64 * zke::SEED_STRANDS_PER_TILE, 0, (hipStream_t)stream>>>(to_dev(sc), seed_sprog, tab, c, n_carries, inputs_rw, n_instances, lpb, n_slots, n_input_words);
    else
        zke::k_seed_cone_strands<false><<<grid, 64 * zke::SEED_STRANDS_PER_TILE, 0, (hipStream_t)stream>>>(to_dev(sc), seed_sprog, tab, c, n_carries, inputs_rw, n_instances, lpb, n_slots, n_input_words);
    return LAUNCH_CHECK("k_seed_cone_strands");
}

bool seed_wave_fits(uint32_t prog_u16, uint32_t n_slots, uint32_t n_input_words) {
    return prog_u16 + 64 <= zke::SW_PROG_U16 && n_slots + n_input_words + 128 <= zke::SW_AREA && n_input_words <= 512;
}
int launch_seed_wave(const ScopeArgs& sc, const uint16_t* prog, uint32_t prog_u16, uint32_t pro_words, uint32_t n_slots, uint32_t n_input_words,
                     const CarryArgs* d_carries, uint32_t n_carries, uint64_t* inputs_rw, uint32_t n_instances, void* stream) {
    if (n_instances == 0 || sc.limit == 0) return 0;
    if (!seed_wave_fits(prog_u16, n_slots, n_input_words)) return -1;
    zke::SeedWaveDev a;
    a.sc = to_dev(sc); a.prog = prog; a.prog_u16 = prog_u16; a.pro_words = pro_words;
    a.carries = reinterpret_cast<const zke::SeedCarryDev*>(d_carries); a.n_carries = n_carries;
    a.inputs_rw = inputs_rw; a.n_instances = n_instances; a.n_slots = n_slots; a.n_input_words = n_input_words;
    // one wavefront per instance; up to 8 instances share a workgroup (and its LDS copy of the program); few instances spread over the CUs
    uint32_t waves = std::min<uint32_t>(zke::SW_WAVES, std::max<uint32_t>(1, (n_instances + 255) / 256));
    const unsigned grid = (n_instances + waves - 1) / waves;
    zke::k_seed_wave<<<grid, 64 * waves, 0, (hipStream_t)stream>>>(a);
    return LAUNCH_CHECK("k_seed_wave");
}

// ---- chain-specialised main_vm seeding
namespace {
constexpr uint32_t VM_SEED_MAX_CHUNKS = 16;
struct VmScratch { size_t mem_ev, dec_ev, fwd_ev, sp_ev, mem_snap, dec_snap, fwd_snap, sp_snap, counts, totals, chunk_totals, saved, end; };
VmScratch vm_scratch_layout(uint32_t limit, uint32_t n) {
    VmScratch L;
    const size_t cap_mem = (size_t)limit * zkvm::MEM_EVENTS_PER_CYCLE, cap_one = limit;
    size_t off = 0;
    auto take = [&](size_t words) { const size_t o = off; off += (words * 8 + 255) & ~(size_t)255; return o; };
    L.mem_ev = take((size_t)n * cap_mem * zkvm::EV_MEM);
    L.dec_ev = take((size_t)n * cap_one * zkvm::EV_DEC);
    L.fwd_ev = take((size_t)n * cap_one * zkvm::EV_FWD);
    L.sp_ev = take((size_t)n * cap_one * zkvm::EV_SP);
    L.mem_snap = take((size_t)n * cap_mem * 12);
    L.dec_snap = take((size_t)n * cap_one * 12);
    L.fwd_snap = take((size_t)n * cap_one * 4);
    L.sp_snap = take((size_t)n * cap_one * 12);
    L.counts = take((size_t)n * limit * 2);
    L.totals = take((size_t)n * 2);
    L.chunk_totals = take((size_t)n * 2 * VM_SEED_MAX_CHUNKS);
    L.saved = take(((size_t)n * zkvm::SAVE_WORDS + 1) / 2);
    L.end = off;
    return L;
}
// two helper streams + events of the chunk pipeline (per process, created on first use)
struct VmSeedStreams {
    hipStream_t walk = nullptr, chains = nullptr;
    hipEvent_t begin = nullptr, done = nullptr, walked[VM_SEED_MAX_CHUNKS] = {};
    int device = -1;
};
VmSeedStreams g_vm_seed;
int vm_seed_streams() {
    int dev = 0;
    if (int r = chk(hipGetDevice(&dev), "hipGetDevice")) return r;
    if (g_vm_seed.device == dev) return 0;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (int r = chk(hipStreamCreateWithPriority(&g_vm_seed.walk, hipStreamNonBlocking, hi), "hipStreamCreate")) return r;
    if (int r = chk(hipStreamCreateWithPriority(&g_vm_seed.chains, hipStreamNonBlocking, hi), "hipStreamCreate")) return r;
    if (int r = chk(hipEventCreateWithFlags(&g_vm_seed.begin, hipEventDisableTiming), "hipEventCreate")) return r;
    if (int r = chk(hipEventCreateWithFlags(&g_vm_seed.done, hipEventDisableTiming), "hipEventCreate")) return r;
    for (auto& e : g_vm_seed.walked) if (int r = chk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return r;
    g_vm_seed.device = dev;
    return 0;
}
}  // namespace
size_t vm_seed_scratch_bytes(uint32_t limit, uint32_t n_instances) { return vm_scratch_layout(limit, n_instances).end; }

int launch_vm_seed(const VmSeedArgs& v, void* stream, float* phase_ms) {
    if (v.n_instances == 0 || v.limit == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    zkvm::SeedDev a;
    vmn::defs_prepare(a.D, (const zk_opcode_defs*)v.defs_host, (const zk_opcode_defs*)v.defs_dev);  // tables on the device, the rest by value
    a.loop = v.loop; a.in_stride = v.in_stride; a.limit = v.limit; a.n_instances = v.n_instances;
    static_assert(sizeof(zkvm::RawLayout) == sizeof(VmRawLayout), "layout mirrors");
    std::memcpy(&a.raw, &v.raw, sizeof a.raw);
    a.outer_store = v.outer_store; a.outer_n_store = v.outer_n_store; a.state0_slot = v.state0_slot;
    a.outer_inputs = v.outer_inputs; a.outer_in_stride = v.outer_in_stride; a.w_zkporter = v.w_zkporter; a.w_default_aa = v.w_default_aa;
    const VmScratch L = vm_scratch_layout(v.limit, v.n_instances);
    char* const base = (char*)v.scratch;
    a.mem_ev = (uint64_t*)(base + L.mem_ev); a.dec_ev = (uint64_t*)(base + L.dec_ev); a.fwd_ev = (uint64_t*)(base + L.fwd_ev); a.sp_ev = (uint64_t*)(base + L.sp_ev);
    a.mem_snap = (uint64_t*)(base + L.mem_snap); a.dec_snap = (uint64_t*)(base + L.dec_snap); a.fwd_snap = (uint64_t*)(base + L.fwd_snap);
    a.sp_snap = (uint64_t*)(base + L.sp_snap);
    a.counts = (uint4*)(base + L.counts); a.totals = (uint4*)(base + L.totals);
    a.chunk_totals = (uint4*)(base + L.chunk_totals); a.saved_state = (uint32_t*)(base + L.saved);
    a.cap_mem = v.limit * zkvm::MEM_EVENTS_PER_CYCLE; a.cap_one = v.limit;
    a.n_loop_words = v.n_loop_words;
    if (v.n_loop_words < (uint32_t)vmn::STATE_WORDS || v.n_loop_words - vmn::STATE_WORDS > zkvm::RAW_MAX) { g_hip_err = "vm seed: oracle words per cycle exceed the staging buffer"; return -1; }
    const uint64_t groups = (uint64_t)v.n_instances * 4, per_block = 4 * zkvm::CH_GROUPS;
    const unsigned chain_grid = (unsigned)((groups + per_block - 1) / per_block);
    // chunks of cycles: the chains of chunk j (their own stream) run underneath the walker's chunk j + 1.  Timed passes run one chunk,
    // phase after phase.
    uint32_t chunks = 6;
    if (const char* e = std::getenv("ZKGL_VM_SEED_CHUNKS")) chunks = (uint32_t)std::max(1, atoi(e));
    chunks = std::min<uint32_t>(std::min<uint32_t>(chunks, VM_SEED_MAX_CHUNKS), std::max<uint32_t>(1, v.limit / 64));
    if (phase_ms) chunks = 1;
    // instances per walking wavefront: one walker lane per SIMD while the chip has room, then two or four share a wavefront
    // (profiles/r3_summary.md §2: the pass is flat up to one wavefront per SIMD and linear beyond)
    int ni = v.n_instances > 2048 ? 4 : v.n_instances > 1024 ? 2 : 1;
    if (const char* e = std::getenv("ZKGL_VM_WALK_NI")) { const int x = atoi(e); if (x == 1 || x == 2 || x == 4) ni = x; }
    auto walk = [&](hipStream_t s_) {
        const unsigned grid = (v.n_instances + ni - 1) / ni;
        if (ni == 1) zkvm::k_vm_walk<1><<<grid, 64, 0, s_>>>(a);
        else if (ni == 2) zkvm::k_vm_walk<2><<<grid, 64, 0, s_>>>(a);
        else zkvm::k_vm_walk<4><<<grid, 64, 0, s_>>>(a);
    };
    if (chunks == 1) {
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        if (phase_ms) for (auto& e : ev) if (int r = chk(hipEventCreate(&e), "hipEventCreate")) return r;
        if (phase_ms) hipEventRecord(ev[0], st);
        a.c0 = 0; a.c1 = v.limit; a.chunk = 0;
        walk(st);
        if (int r = LAUNCH_CHECK("k_vm_walk")) return r;
        if (phase_ms) hipEventRecord(ev[1], st);
        zkvm::k_vm_chains<<<chain_grid, 256, 0, st>>>(a);
        if (int r = LAUNCH_CHECK("k_vm_chains")) return r;
        if (phase_ms) hipEventRecord(ev[2], st);
        zkvm::k_vm_fill<<<grid_for((size_t)v.n_instances * v.limit, 256), 256, 0, st>>>(a);
        if (int r = LAUNCH_CHECK("k_vm_fill")) return r;
        if (phase_ms) {
            hipEventRecord(ev[3], st);
            if (int r = chk(hipEventSynchronize(ev[3]), "vm seed sync")) return r;
            for (int i = 0; i < 3; ++i) hipEventElapsedTime(&phase_ms[i], ev[i], ev[i + 1]);
            for (auto& e : ev) hipEventDestroy(e);
        }
        return 0;
    }
    if (int r = vm_seed_streams()) return r;
    VmSeedStreams& S = g_vm_seed;
    if (int r = chk(hipEventRecord(S.begin, st), "hipEventRecord")) return r;
    if (int r = chk(hipStreamWaitEvent(S.walk, S.begin, 0), "hipStreamWaitEvent")) return r;
    for (uint32_t j = 0; j < chunks; ++j) {
        a.chunk = j;
        a.c0 = (uint32_t)((uint64_t)v.limit * j / chunks);
        a.c1 = (uint32_t)((uint64_t)v.limit * (j + 1) / chunks);
        walk(S.walk);
        if (int r = LAUNCH_CHECK("k_vm_walk")) return r;
        if (int r = chk(hipEventRecord(S.walked[j], S.walk), "hipEventRecord")) return r;
        if (int r = chk(hipStreamWaitEvent(S.chains, S.walked[j], 0), "hipStreamWaitEvent")) return r;
        zkvm::k_vm_chains<<<chain_grid, 256, 0, S.chains>>>(a);
        if (int r = LAUNCH_CHECK("k_vm_chains")) return r;
    }
    if (int r = chk(hipEventRecord(S.done, S.chains), "hipEventRecord")) return r;
    if (int r = chk(hipStreamWaitEvent(st, S.done, 0), "hipStreamWaitEvent")) return r;
    zkvm::k_vm_fill<<<grid_for((size_t)v.n_instances * v.limit, 256), 256, 0, st>>>(a);
    return LAUNCH_CHECK("k_vm_fill");
}

int launch_ram_seed(const RamSeedArgs& v, void* stream) {
    if (v.n_instances == 0 || v.limit == 0) return 0;
    zkq::RamSeedDev a;
    a.loop = v.loop; a.in_stride = v.in_stride; a.limit = v.limit; a.n_instances = v.n_instances;
    a.outer_store = v.outer_store; a.outer_n_store = v.outer_n_store; a.state0_slot = v.state0_slot; a.ch_slot = v.ch_slot;
    a.bootloader_heap_page = v.bootloader_heap_page;
    zkq::k_ram_seed<<<v.n_instances, 256, 0, (hipStream_t)stream>>>(a);
    return LAUNCH_CHECK("k_ram_seed");
}

int launch_logq_seed(const LogqSeedArgs& v, void* stream) {
    if (v.n_instances == 0 || v.limit == 0) return 0;
    zkq::LogqSeedDev a;
    a.loop = v.loop; a.in_stride = v.in_stride; a.limit = v.limit; a.n_instances = v.n_instances;
    a.outer_store = v.outer_store; a.outer_n_store = v.outer_n_store; a.state0_slot = v.state0_slot; a.ch_slot = v.ch_slot;
    hipStream_t st = (hipStream_t)stream;
    if (v.kind == 0) {
        zkq::k_logq_seed<0><<<v.n_instances, zkq::LOGQ_TPB, 0, st>>>(a);
        if (v.with_chain) zkq::k_tail4_chain<0><<<v.n_instances, 64, 0, st>>>(a);
    } else if (v.kind == 1) {
        zkq::k_logq_seed<1><<<v.n_instances, zkq::LOGQ_TPB, 0, st>>>(a);
        if (v.with_chain) zkq::k_tail4_chain<1><<<v.n_instances, 64, 0, st>>>(a);
    } else if (v.kind == 2) {   // sort_decommittment_requests: every queue state is the host's, the scans are here
        zkq::k_decommit_seed<<<v.n_instances, zkq::LOGQ_TPB, 0, st>>>(a);
    } else return -1;
    return LAUNCH_CHECK("k_logq_seed");
}

int launch_eip4844_seed(const EipSeedArgs& v, void* stream) {
    if (v.n_instances == 0 || v.limit == 0) return 0;
    if (v.cpi == 0 || v.cpi > zkf::EIP_MAX_CPI) return -1;
    zkf::EipSeedDev a;
    a.loop = v.loop; a.in_stride = v.in_stride; a.limit = v.limit; a.n_instances = v.n_instances; a.n_chunks = v.n_chunks; a.cpi = v.cpi;
    a.outer_inputs = v.outer_inputs; a.outer_in_stride = v.outer_in_stride;
    zkf::k_eip4844_seed<<<v.n_instances, 128, 0, (hipStream_t)stream>>>(a);
    return LAUNCH_CHECK("k_eip4844_seed");
}

int launch_fsm_seed(const FsmSeedArgs& v, void* stream) {
    if (v.n_instances == 0 || v.limit == 0) return 0;
    zkf::FsmSeedDev a;
    a.loop = v.loop; a.in_stride = v.in_stride; a.limit = v.limit; a.n_instances = v.n_instances;
    a.outer_store = v.outer_store; a.outer_n_store = v.outer_n_store; a.state0_slot = v.state0_slot;
    const char* dbg = std::getenv("ZKGL_FSM_SEED_DEBUG");
    a.debug = dbg ? (uint32_t)std::atoi(dbg) : 0;
    if (v.kind == 0) zkf::k_fsm_seed<zkf::Keccak><<<v.n_instances, 128, 0, (hipStream_t)stream>>>(a);
    else if (v.kind == 1) zkf::k_fsm_seed<zkf::Sha256><<<v.n_instances, 128, 0, (hipStream_t)stream>>>(a);
    else return -1;
    return LAUNCH_CHECK("k_fsm_seed");
}

int launch_check_stream(const uint64_t* loop_cells, uint64_t loop_n_cells, uint32_t n_instances, uint32_t limit,
                        const uint32_t* a_cells, uint32_t pa, const uint32_t* b_cells, uint32_t pb, uint32_t n_total,
                        uint32_t stream_index, unsigned long long* fail, void* stream) {
    if (!n_instances || !n_total) return 0;
    if (zkgeom::narrow(loop_n_cells))
        zke::k_check_stream_t<true><<<grid_for((size_t)n_instances * n_total, zke::TPB), zke::TPB, 0, (hipStream_t)stream>>>(
            loop_cells, loop_n_cells, n_instances, limit, a_cells, pa, b_cells, pb, n_total, stream_index, fail);
    else
        zke::k_check_stream_t<false><<<grid_for((size_t)n_instances * n_total, zke::TPB), zke::TPB, 0, (hipStream_t)stream>>>(
            loop_cells, loop_n_cells, n_instances, limit, a_cells, pa, b_cells, pb, n_total, stream_index, fail);
    return LAUNCH_CHECK("k_check_stream");
}

int launch_check_links(const uint64_t* loop_cells, uint64_t loop_n_cells, uint32_t n_lanes, uint32_t limit,
                       const uint64_t* outer_cells, uint64_t outer_n_cells, const zk_link* links, uint32_t n_links,
                       unsigned long long* fail, void* stream) {
    if (n_lanes == 0 || n_links == 0) return 0;
    if (zkgeom::narrow(loop_n_cells))
        zke::k_check_links_t<true><<<grid_for(n_lanes, zke::TPB), zke::TPB, 0, (hipStream_t)stream>>>(
            loop_cells, loop_n_cells, n_lanes, limit, outer_cells, outer_n_cells, links, n_links, fail);
    else
        zke::k_check_links_t<false><<<grid_for(n_lanes, zke::TPB), zke::TPB, 0, (hipStream_t)stream>>>(
            loop_cells, loop_n_cells, n_lanes, limit, outer_cells, outer_n_cells, links, n_links, fail);
    return LAUNCH_CHECK("k_check_links");
}

}  // namespace zkdev
