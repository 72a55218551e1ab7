// device_api.hpp — internal launch interface between the host C++ (recorder, C ABI) and the
// single device translation unit zkgl_device.hip.  Not part of the public ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/zkgl_ir.h"
#include "store_geom.hpp"

namespace zkdev {

// Every (pointer, n_cells / n_store) pair below is a store and its GEOMETRY WORD (store_geom.hpp): the slot count, with the lane tiling in
// the top byte when it is not the 64-lane default.
struct ScopeArgs {  // mirrors zke::ScopeDev (plain data)
    const uint32_t* prog; uint32_t n_words; uint32_t n_lanes;
    const uint64_t* consts; uint64_t* cells; uint64_t n_cells; const uint64_t* inputs;
    const uint64_t* outer_cells; uint64_t outer_n_cells; uint32_t limit; uint32_t is_loop;
    const zk_table_desc* tables; const uint64_t* table_words; uint32_t* mult; uint32_t total_table_rows;
    const uint64_t* loop_cells; uint64_t loop_n_cells; uint32_t loop_limit;
    uint64_t in_stride;    // lanes between consecutive words of the input stream (>= n_lanes: a batch may be a window of a longer stream)
    uint32_t uses_bigint;  // host only: the program contains ZK_OP_NN_MULMOD -> launch the *_bigint kernel variants
    uint32_t xmacros = 0;  // host only: macro-op backends beyond the basic set the circuit records (kernels_engine2.hpp X_SHA4 = 1, X_BYTEBUF = 2): their own kernels
    unsigned long long* fail = nullptr;  // fused mode: where the witness kernels report a gate they evaluate themselves (SELECT with a non-boolean selector)
    uint32_t defer_p2 = 0;                       // 1: ZK_OP_P2_ROUNDS stores only its 12 final outputs (ZK_CHECK_FUSED_DEFER_P2), the 950 intermediates come from launch_fill_p2
    unsigned long long* p2_stats = nullptr;     // two counters: gated witness-only permutations a wavefront skipped / ran (ZK_OP_POSEIDON2 a = 1)
    unsigned long long* clock_probe = nullptr;  // two words: shader-clock and 100 MHz ticks of the grid's first wavefront (kernels_engine2.hpp witness_entry2)
    // host only — narrow store (store_geom.hpp): the class words of `prog` (one per header); cells / n_cells are then the narrow store and its
    // geometry word, and launch_witness takes k_witness_loop_narrow
    const uint32_t* cls = nullptr;
};
struct CheckArgs {  // mirrors zke::CheckDev
    const uint64_t* cells; uint64_t n_cells; uint32_t n_cols; uint32_t n_lanes; uint32_t n_slots;
    const zk_row_desc* rows; const uint64_t* rowconsts; const zk_lookup_row_desc* lrows;
    uint32_t n_copy_cols; uint32_t lookup_width; const zk_table_desc* tables; const uint64_t* table_words;
    unsigned long long* fail; uint32_t slots_per_chunk;
    const uint32_t* alias;  // compact trace: trace cell -> home cell (nullptr: materialised trace)
    const uint32_t* cprog = nullptr; const uint32_t* chunk_tab = nullptr; uint32_t n_chunks = 0;
    const uint32_t* macros = nullptr; uint32_t n_macros = 0;  // Poseidon2 macro descriptors whose gates the check program leaves out (k_check_p2)  // compact trace: the check program (k_check_prog)
};

int upload_round_constants(const uint64_t rc[360]);
int launch_col(int op, uint64_t* dst, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t q, uint64_t l,
               size_t n, void* stream);
int launch_poseidon2_soa(uint64_t* st, size_t n, size_t stride, void* stream);
int launch_poseidon2_aos(uint64_t* st, size_t n, void* stream);
int launch_commit_encoding(const uint64_t* in, size_t len, size_t n, uint64_t* out, void* stream);
int launch_queue_full_chain(const uint64_t* enc, size_t nq, size_t items, uint64_t* tail_io, uint64_t* states_out,
                            void* stream);
int launch_memory_query_encode(const uint64_t* q, size_t n, uint64_t* enc, void* stream);
int launch_execution_context_encode(const uint64_t* rec, size_t n, uint64_t* enc, void* stream);
int launch_grand_product(const uint64_t* enc, const uint64_t* flags, const uint64_t* ch, size_t enc_len, size_t n,
                         uint64_t init, uint64_t* acc, uint64_t* scratch, void* stream);
int launch_pack_public(const uint64_t* outer_store, uint64_t n_store, const uint32_t* slots, uint32_t n_public, uint32_t n_instances, uint64_t* out, void* stream);
int launch_hook_compare(const uint64_t* outer_store, uint64_t n_store, const uint32_t* slots, uint32_t n_vars, uint32_t n_instances, const uint64_t* expected,
                        unsigned long long* fail, void* stream);
int launch_check_inputs(const uint64_t* inputs, uint32_t n_words, uint32_t n_lanes, uint64_t stride, unsigned long long* fail, void* stream);
int launch_multiplicities(const uint64_t* store, uint64_t n_store, uint32_t lanes_per_instance, uint32_t n_lanes, uint32_t n_instances, const uint32_t* sites,
                          uint32_t n_sites, const zk_table_desc& t, const uint64_t* table_words, uint32_t* mult, uint32_t total_table_rows, void* stream);
int launch_witness(const ScopeArgs& sc, uint32_t word_begin, uint32_t word_end, uint32_t slot_begin, void* stream);
// strand mode (kernels_engine.hpp k_witness_strands): sc.prog = the strand program, begin/end = 8 word ranges
#ifndef ZKGL_STRANDS_PER_TILE
#define ZKGL_STRANDS_PER_TILE 16
#endif
constexpr uint32_t STRANDS_PER_TILE = ZKGL_STRANDS_PER_TILE;
// SELECT flags kept as per-wavefront bit planes in LDS by the plain loop kernels (ZK_OP_FLAG_PLANES): plane ids per scope
constexpr uint32_t FLAG_PLANES = 256;
// the seeding kernels keep 8: a seeding pass is a latency chain per instance and its throughput is the number of resident blocks
// (2048 instances: 2.5 s with 8-wave blocks, 3.6 s with 16-wave blocks)
constexpr uint32_t SEED_STRANDS_PER_TILE = 8;
// n_strands <= STRANDS_PER_TILE wavefronts per tile walk strands 0 .. n_strands - 1 (the program was dealt for that many)
int launch_witness_strands(const ScopeArgs& sc, const uint32_t begin[STRANDS_PER_TILE], const uint32_t end[STRANDS_PER_TILE], void* stream,
                           uint32_t n_strands = STRANDS_PER_TILE);
struct CarryArgs { uint32_t word, out_cell, first_outer_cell, has_first; };  // mirrors zke::CarryDev
// op-parallel seeding (kernels_seed_wave.hpp): one wavefront per instance, the program resident in LDS
bool seed_wave_fits(uint32_t prog_u16, uint32_t n_slots, uint32_t n_input_words);
int launch_seed_wave(const ScopeArgs& sc, const uint16_t* prog, uint32_t prog_u16, uint32_t pro_words, uint32_t n_slots, uint32_t n_input_words,
                     const CarryArgs* d_carries, uint32_t n_carries, uint64_t* inputs_rw, uint32_t n_instances, void* stream);
// chain-specialised main_vm seeding (kernels_vm_seed.hpp): native walker + Poseidon2 chains + fill.  Plain mirror of zkvm::SeedDev.
struct VmRawLayout { uint32_t code_word, src0_value, src0_is_ptr, refund, log_read, log_prev_head, near_tail, far_code_hash, far_page, far_tail, ret_ctx, ret_state, uma_a, uma_b; };
struct VmSeedArgs {
    const void* defs_dev;            // zk_opcode_defs, device copy
    const void* defs_host;           // the same blob on the host (derived fields are computed from it)
    uint64_t* loop; uint64_t in_stride; uint32_t limit, n_instances, n_loop_words;
    VmRawLayout raw;
    const uint64_t* outer_store; uint64_t outer_n_store; const uint32_t* state0_slot;
    const uint64_t* outer_inputs; uint64_t outer_in_stride; uint32_t w_zkporter, w_default_aa;
    uint64_t* scratch;               // vm_seed_scratch_bytes(limit, n_instances) bytes
};
size_t vm_seed_scratch_bytes(uint32_t limit, uint32_t n_instances);
// phase_ms (optional, 3 floats): walker / chains / fill, measured with events on `stream` (synchronises)
int launch_vm_seed(const VmSeedArgs& a, void* stream, float* phase_ms);
// ram_permutation seeding with the queue heads given (kernels_queue_seed.hpp): scans, no chain.  Mirrors zkq::RamSeedDev.
struct RamSeedArgs {
    uint64_t* loop; uint64_t in_stride; uint32_t limit, n_instances;
    const uint64_t* outer_store; uint64_t outer_n_store; const uint32_t* state0_slot; const uint32_t* ch_slot; uint32_t bootloader_heap_page;
};
int launch_ram_seed(const RamSeedArgs& a, void* stream);
// precompile FSM seeding (kernels_fsm_seed.hpp): kind 0 keccak256_round_function, 1 sha256_round_function.  Mirrors zkf::FsmSeedDev.
struct FsmSeedArgs {
    int kind; uint64_t* loop; uint64_t in_stride; uint32_t limit, n_instances;
    const uint64_t* outer_store; uint64_t outer_n_store; const uint32_t* state0_slot;
};
int launch_fsm_seed(const FsmSeedArgs& a, void* stream);
// eip_4844 seeding (kernels_fsm_seed.hpp: sponge lane + Horner lane per instance).  Mirrors zkf::EipSeedDev.
struct EipSeedArgs {
    uint64_t* loop; uint64_t in_stride; uint32_t limit, n_instances, n_chunks, cpi; const uint64_t* outer_inputs; uint64_t outer_in_stride;
};
int launch_eip4844_seed(const EipSeedArgs& a, void* stream);
// the LogQuery sorters with the integer state given by the host packer (kernels_queue_seed.hpp): kind 0 storage_validity, 1 log_sorter;
// with_chain: the output queue's tail is not given and is hashed here.  Mirrors zkq::LogqSeedDev.
struct LogqSeedArgs {
    int kind, with_chain; uint64_t* loop; uint64_t in_stride; uint32_t limit, n_instances;
    const uint64_t* outer_store; uint64_t outer_n_store; const uint32_t* state0_slot; const uint32_t* ch_slot;
};
int launch_logq_seed(const LogqSeedArgs& a, void* stream);
int launch_witness_seq(const ScopeArgs& loop_sc, const CarryArgs* d_carries, uint32_t n_carries, uint64_t* inputs_rw,
                       uint32_t n_instances, void* stream);
int launch_check_gates(const CheckArgs& cd, void* stream);
int launch_materialize(uint64_t* trace, uint64_t n_cells, const uint64_t* store, uint64_t n_store, uint32_t n_lanes, const zk_copy_pair* pairs,
                       uint32_t n_pairs, void* stream);  // pairs: {trace cell, store slot}
int launch_check_copies(const uint64_t* cells, uint64_t n_cells, uint32_t n_lanes, const zk_copy_pair* pairs,
                        uint32_t n_pairs, unsigned long long* fail, void* stream);
// K10 lookup-argument accumulators (kernels_lookup_arg.hpp).  ch = beta, gamma, gamma^2, gamma^3, gamma^4 (2 words each)
struct LookupArgArgs {
    const uint64_t* cells; uint64_t n_cells; uint32_t n_cols, n_lanes, n_slots, n_copy_cols, lookup_width;
    const zk_lookup_row_desc* lrows; uint64_t* acc;
};
int launch_lookup_arg_witness(const LookupArgArgs& a, const uint64_t ch[10], void* stream);
int launch_lookup_arg_tables(const zk_table_desc* tables, uint32_t n_tables, const uint64_t* table_words, uint32_t total_rows, uint32_t lookup_width,
                             const uint64_t ch[10], uint64_t* inv_f, const uint32_t* mult, uint32_t n_instances, uint64_t* out_b, void* stream);
int launch_lookup_arg_witness_sum(const uint64_t* acc_outer, const uint64_t* acc_loop, uint32_t limit, uint32_t n_instances,
                                  uint64_t* out_a, void* stream);
// K11 NTT (kernels_ntt.hpp); mirrors zkn::PassDev
struct NttPassArgs {
    const uint64_t* src; uint64_t* dst; uint64_t src_stride, dst_stride;
    uint32_t log_n, seg, r, t, dit, coset_store, coset_brev;
    const uint64_t* root1024; const uint64_t* tw_lo; const uint64_t* tw_hi; const uint64_t* c_lo; const uint64_t* c_hi;
};
int launch_ntt_pass(const NttPassArgs& a, uint32_t n_polys, void* stream);
// K12 copy-permutation grand product (kernels_perm.hpp); mirrors zkp::PermDev
struct PermArgs {
    const uint64_t* cells; uint64_t n_cells; uint32_t n_cols, n_lanes, n_slots, n_copy_cols, lookup_width;
    const zk_row_desc* rows; const zk_lookup_row_desc* lrows;
    const uint32_t* sigma_rel; const uint32_t* ep_index; const uint64_t* ovr; uint32_t lanes_per_instance;
    uint64_t label_base, label_step; const uint64_t* tb; uint64_t beta[2], gamma[2];
    uint32_t slots_per_chunk, n_chunks; uint64_t* lane_out; uint64_t* prefix;
    const uint32_t* slot1 = nullptr;   // compact batch: `cells` is the variable store, slot1[trace cell] = store slot + 1
};
// the 950 intermediates of every in-circuit Poseidon2 permutation of a scope, recomputed from its 12 stored inputs (descriptors = the check macros)
int launch_fill_p2(uint64_t* store, uint64_t n_store, uint32_t n_lanes, const uint32_t* macros, uint32_t n_macros, void* stream);
// narrow store -> ordinary store (store_geom.hpp): aw[slot] = address word of the slot's value; the geometry words say which is which
int launch_widen_store(const uint64_t* narrow, uint64_t narrow_geom, uint64_t* wide, uint64_t wide_geom, uint32_t n_lanes, const uint32_t* aw, uint32_t n_slots, void* stream);
// the same for a list of slots at the last iteration of every instance (what ZK_OP_LOOP_LAST of the outer post phase reads from the ordinary store)
int launch_widen_last(const uint64_t* narrow, uint64_t narrow_geom, uint64_t* wide, uint64_t wide_geom, uint32_t n_instances, uint32_t limit, const uint32_t* aw,
                      const uint32_t* slots, uint32_t n_list, void* stream);
int launch_perm_lane(const PermArgs& a, void* stream);
int launch_perm_tb(const uint64_t beta[2], const uint32_t* sigma_rel, uint64_t* tb, uint32_t n, void* stream);
int launch_perm_scan(const uint64_t* part, uint32_t per, const uint64_t* seed, uint64_t* excl, uint64_t* total, uint32_t n_instances, void* stream);
int launch_perm_z(const uint64_t* excl, const uint64_t* prefix, uint32_t n_lanes, uint32_t n_slots, uint32_t slots_per_chunk, uint32_t n_chunks,
                  uint32_t lanes_per_instance, uint64_t row_base, uint64_t rows_per_instance, uint64_t* z, void* stream);
struct ColumnsArgs {  // mirrors zkn::ColumnsDev
    const uint64_t* loop_cells; uint64_t loop_n_cells; const uint64_t* outer_cells; uint64_t outer_n_cells;
    uint32_t n_cols, loop_slots, outer_slots, limit, instance; uint64_t* out; uint64_t stride; uint64_t n_rows_padded;
    const uint32_t* loop_slot1 = nullptr; const uint32_t* outer_slot1 = nullptr;  // compact mode: trace cell -> store slot + 1 (0: unpopulated)
    uint32_t n_instances = 1; uint64_t instance_stride = 0;   // batch form: instances [instance, instance + n_instances), out + i * instance_stride
};
int launch_trace_columns(const ColumnsArgs& a, void* stream);
int launch_coset_tables(uint64_t base, uint64_t scale, uint64_t* c_lo, uint64_t* c_hi, uint32_t n_hi, void* stream);
// cone seeding: seed_prog in device memory (padded like every program), carries = {input word, out slot, first outer cell, has_first}
int launch_seed_cone(const ScopeArgs& sc, const uint32_t* seed_prog, uint32_t n_words, uint32_t n_slots, uint32_t n_input_words,
                     const CarryArgs* d_carries, uint32_t n_carries, uint64_t* inputs_rw, uint32_t n_instances, void* stream);
int launch_seed_cone_strands(const ScopeArgs& sc, const uint32_t* seed_sprog, const uint32_t begin[STRANDS_PER_TILE], const uint32_t end[STRANDS_PER_TILE], uint32_t n_slots,
                             uint32_t n_input_words, const CarryArgs* d_carries, uint32_t n_carries, uint64_t* inputs_rw, uint32_t n_instances, bool v2, void* stream);
uint32_t seed_cone_max_slots();
int launch_check_stream(const uint64_t* loop_cells, uint64_t loop_n_cells, uint32_t n_instances, uint32_t limit,
                        const uint32_t* a_cells, uint32_t pa, const uint32_t* b_cells, uint32_t pb, uint32_t n_total,
                        uint32_t stream_index, unsigned long long* fail, void* stream);
int launch_check_links(const uint64_t* loop_cells, uint64_t loop_n_cells, uint32_t n_lanes, uint32_t limit,
                       const uint64_t* outer_cells, uint64_t outer_n_cells, const zk_link* links, uint32_t n_links,
                       unsigned long long* fail, void* stream);
const char* last_hip_error();

}  // namespace zkdev
