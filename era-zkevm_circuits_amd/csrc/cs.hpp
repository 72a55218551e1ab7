// cs.hpp — host-side constraint-system recorder, placer, program emitter and GPU executor.
//
// Mirrors the surface of boojum's `ConstraintSystem<F>` + `CsBuilder` as the reference uses it
// (/root/reference/src/ram_permutation/mod.rs:419-556; call-site inventory in SURVEY.md §8b).
// Differences that make it MI355X-native:
//   * witness closures are replaced by the closed op set of include/zkgl_ir.h;
//   * the `for _cycle in 0..limit` body is recorded once (loop scope) and executed with
//     lane == (instance, iteration); the per-instance prologue/epilogue is the outer scope;
//   * the trace is slot-major/lane-minor (cells[(col*slots+slot)*stride + lane]) so every wave
//     access is one coalesced 512 B transaction; gate selectors/constants are per-slot scalars.
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/zkgl.h"
#include "device_api.hpp"

namespace zkgl {

struct ZkError : std::runtime_error {
    int code;
    ZkError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

constexpr uint32_t LOOP_BIT = 0x80000000u;
inline bool is_loop_var(zk_var v) { return (v & LOOP_BIT) != 0; }
inline uint32_t var_index(zk_var v) { return v & ~LOOP_BIT; }

struct Operand {
    enum Kind : uint8_t { VAR, CONSTPOOL, OUTER_VAR, RAW } kind;
    uint32_t idx;
};

struct OpRec {
    uint8_t opcode, a;
    uint16_t b;
    std::vector<Operand> ins;
    std::vector<uint32_t> outs;  // var indices (same scope)
    bool seed_only = false;      // seed hint: not part of the trace program, a second producer of `outs` for the seeding cone
};

struct GateRec {
    uint32_t kind;
    std::vector<uint32_t> vars;  // var indices
    std::vector<uint64_t> consts;
    int32_t owner = -1;          // first output variable of the macro-op whose gadget placed this gate inside its window (CS::emit_macro_op .. end_macro_op); -1: anybody
};

constexpr int32_t OWNER_LOOKUP_OP = 0x7fffffff;
struct LookupRec {
    uint32_t table;
    std::vector<uint32_t> vars;  // keys then values, var indices
    int32_t owner = -1;          // who evaluates this tuple in the witness kernels: OWNER_LOOKUP_OP = the ZK_OP_LOOKUP CS::lookup records with it; >= 0 = first output
                                 // variable of the macro-op whose gadget gave it inside its window (CS::lookup_given); -1: nobody — the fused check program keeps it
};

struct TableRec {
    uint32_t marker, n_keys, n_vals, n_rows;
    std::vector<uint64_t> rows;  // sorted by key tuple, row-major
    bool byte_valued = false;    // every value column < 256: the device keeps a packed copy, one byte per value
    bool dense;
    uint32_t key_shift[3];
    uint32_t word_off, mult_off;
};

struct Scope {
    bool is_loop = false;
    uint32_t n_vars = 0;
    std::vector<OpRec> ops;
    std::vector<GateRec> gates;
    std::vector<LookupRec> lookups;
    std::unordered_map<uint64_t, uint32_t> const_vars;  // allocate_constant cache
    std::vector<uint64_t> const_pool;
    std::unordered_map<uint64_t, uint32_t> const_pool_idx;
    uint32_t n_input_words = 0;
    std::unordered_map<uint32_t, uint32_t> input_word;  // var index -> input stream word (ZK_OP_INPUT)
    bool uses_bigint = false;    // an op of the big-integer family (ZK_OP_NN_MULMOD) was recorded
    size_t pre_ops = SIZE_MAX;   // outer scope: ops recorded before side_begin/loop_begin (the loop may import them)
    size_t side_ops = SIZE_MAX;  // outer scope: end of the side phase (== loop_begin position)

    // ---- filled by finalize ----
    uint32_t n_slots = 0, n_gate_slots = 0, n_lookup_slots = 0;
    uint32_t n_trace_cells = 0, n_cells = 0, n_scratch = 0;
    std::vector<std::vector<uint32_t>> var_cells;  // per var: cells, [0] = home
    // device program: COMPACT — every value is stored to the home cell of its variable only (the gate checker reads through
    // `alias`, k_materialize fills the other cells on demand).  prog_full: every cell of every variable (export to the oracle).
    std::vector<uint32_t> prog, prog_full;
    uint32_t pre_words = 0, side_words = 0, pre_words_full = 0;
    // prog2: the scalar-decoded form of the plain kernels (kernels_engine2.hpp) — bare slot operands, no destination words;
    // pre/side_slots = store slot of the first output of the side / post phase
    std::vector<uint32_t> prog2;
    // check program of the compact gate / lookup checker (kernels_engine2.hpp k_check_prog) + its chunk table (word offsets of
    // whole-packet chunks, n_chunks + 1 entries); empty when the scope cannot use it (a lookup tuple wider than 4 columns)
    std::vector<uint32_t> cprog, cchunks;
    // the same without macro packets (every gate instance on its own): run when a macro packet reports, to locate the failing gate
    std::vector<uint32_t> cprog_full, cchunks_full;
    // fused mode: without the gates mirrored by their producing op (gate_mirrored) — those are evaluated by the witness kernels
    std::vector<uint32_t> cprog_fused, cchunks_fused;
    std::vector<uint32_t> cmacros;   // Poseidon2 macro descriptors (k_check_p2), 14 words each
    std::vector<std::vector<uint32_t>> row_gates;  // [row][instance] -> index into `gates`
    std::vector<std::vector<uint32_t>> row_lookups;  // [row][tuple] -> index into `lookups`
    uint32_t n_macro_p2 = 0;
    std::vector<uint8_t> value_class;   // per variable, CS::bound_values: 2 = < 2^8, 1 = < 2^32, 0 = a field element (in every satisfying witness)
    uint32_t values_below_2_32 = 0, values_below_2_8 = 0;   // census (CS::bound_values): variables bounded by the constraints in every satisfying witness
    bool p2_intermediates_private = true;   // no op / lookup / link / kept gate of the step reads an intermediate of a ZK_OP_P2_ROUNDS (deferred mode)
    uint32_t n_p2_rounds_ops = 0;    // ZK_OP_P2_ROUNDS ops of the scope (deferred mode needs a verified descriptor for each)
    std::vector<uint8_t> gate_mirrored;   // per gate: its relation is the semantics of the op producing its output (same variables, constants)
    // lookup sites by table for k_multiplicities: 3 key slots per site; site_off[table id] .. site_off[table id + 1]
    std::vector<uint32_t> mult_sites, mult_site_off;
    uint32_t pre_words2 = 0, side_words2 = 0, pre_slots = 0, side_slots = 0;
    uint32_t flag_planes = 0;   // SELECT flags the plain loop kernel keeps as bit planes in LDS (emit_scope)
    // VARIABLE STORE: the witness kernels keep ONE value per variable, in a dense store indexed by production order
    // (store[((lane >> 6) * n_store + slot) * 64 + (lane & 63)]): a wave streams its results out sequentially and reads its
    // operands from recently written, nearby slots (TLB / L2 locality), and a VM instance needs 4x less memory than its trace.
    // The trace (cell = slot * n_columns + column, scratch behind) exists on demand only: k_materialize copies store -> trace.
    std::vector<uint32_t> var_slot;   // variable -> store slot
    uint32_t n_store = 0;
    std::vector<uint32_t> alias;      // trace cell -> store slot of the variable placed there (0 for unpopulated cells)
    std::vector<zk_copy_pair> mat_pairs;  // {trace or scratch cell, store slot} of every populated cell
    uint64_t cells_populated = 0; // trace cells + scratch cells holding a value == destination words of prog_full
    // strand form of the program (build_strands): phase 0 = loop body / outer pre, 1 = outer side, 2 = outer post
    std::vector<uint32_t> sprog;
    uint32_t s_begin[3][zkdev::STRANDS_PER_TILE] = {}, s_end[3][zkdev::STRANDS_PER_TILE] = {};
    uint32_t s_levels[3] = {0, 0, 0};
    float s_gain[3] = {0, 0, 0};  // estimated work / critical path over 8 strands: the strand form is used from 3 upwards
    // narrow strand form (loop scopes with a wide op graph): NARROW_STRANDS per tile, used when the tiles outnumber what the chip
    // keeps resident at STRANDS_PER_TILE wavefronts each (launch_phase)
    std::vector<uint32_t> sprog_n;
    uint32_t sn_begin[3][zkdev::STRANDS_PER_TILE] = {}, sn_end[3][zkdev::STRANDS_PER_TILE] = {};
    uint32_t sn_levels[3] = {0, 0, 0};
    float sn_gain[3] = {0, 0, 0};
    std::vector<zk_row_desc> rows;
    std::vector<uint64_t> rowconsts;
    std::vector<zk_lookup_row_desc> lrows;
    std::vector<zk_copy_pair> copies;
    uint64_t gate_counts[ZK_GATE__COUNT] = {0};
    uint64_t n_constraints = 0;
    uint64_t cells_written = 0;  // destination words of the program == cells one lane stores

    // ---- device ----
    uint32_t* d_prog = nullptr;
    uint32_t* d_prog2 = nullptr;
    uint32_t* d_cprog = nullptr;
    uint32_t* d_cchunks = nullptr;
    uint32_t* d_cmacros = nullptr;
    uint32_t* d_cprog_full = nullptr;
    uint32_t* d_cchunks_full = nullptr;
    uint32_t* d_cprog_fused = nullptr;
    uint32_t* d_cchunks_fused = nullptr;
    uint32_t* d_mult_sites = nullptr;
    uint32_t* d_sprog = nullptr;
    uint32_t* d_sprog_n = nullptr;
    uint64_t* d_consts = nullptr;
    zk_row_desc* d_rows = nullptr;
    uint64_t* d_rowconsts = nullptr;
    zk_lookup_row_desc* d_lrows = nullptr;
    zk_copy_pair* d_copies = nullptr;
    uint32_t* d_alias = nullptr;
    uint32_t* d_slot1 = nullptr;   // trace cell -> store slot + 1 of the variable placed there, 0 = unpopulated (trace_columns on the compact store)
    zk_copy_pair* d_mat_pairs = nullptr;
    uint64_t* d_store = nullptr;   // variable store, allocated by set_batch
    uint32_t store_tile_log2 = zkgeom::WAVE_TILE_LOG2;  // lane tiling of d_store (store_geom.hpp), chosen by set_batch
    // what the launch interface carries beside d_store: the slot count with the tiling in its top byte (a bare count = 64-lane tiles)
    uint64_t store_geom() const { return store_tile_log2 == zkgeom::WAVE_TILE_LOG2 ? (uint64_t)n_store : zkgeom::pack(n_store, store_tile_log2); }
    // NARROW STORE (store_geom.hpp; loop scopes; cs.cpp build_narrow_layout; used by a batch when ZKGL_NARROW_STORE=1 at set_batch): the layout and the device programs of the
    // fused step over it.  The ordinary store and its programs stay: everything outside the fused step reads the widened copy.
    bool narrow_ok = false;              // a layout + programs exist
    std::vector<uint32_t> slot_aw;       // store slot -> address word (first unit | class << 28)
    uint32_t narrow_units = 0;           // units of a tile = bytes one lane writes
    uint32_t narrow_byte_values = 0;     // values held in one-byte slots
    std::vector<uint32_t> prog2n;        // prog2 with address-word operands; its class words (one per header) start at cls_off
    uint32_t cls_off = 0;
    std::vector<uint32_t> cprog_fused_n; // cprog_fused with address words (same packets: cchunks_fused)
    uint32_t* d_prog2n = nullptr;
    uint32_t* d_cprog_fused_n = nullptr;
    uint32_t* d_slot_aw = nullptr;
    uint32_t* d_aw1 = nullptr;           // trace cell -> address word + 1 of the variable placed there (trace_columns straight from the narrow store)
    uint64_t* d_store_n = nullptr;       // the narrow store of the bound batch (set_batch), nullptr when the batch does not use it
    uint64_t narrow_geom() const {
        const uint64_t n8 = ((uint64_t)narrow_units + 7) / 8;   // the tile in 8-byte slots: allocation and tile addressing as for an ordinary store
        return zkgeom::NARROW | (store_tile_log2 == zkgeom::WAVE_TILE_LOG2 ? n8 : zkgeom::pack(n8, store_tile_log2));
    }
    uint64_t* d_cells = nullptr;   // materialised trace, allocated by the first ensure_materialized
    uint64_t stride = 0;
    uint32_t n_lanes = 0;
    const uint64_t* d_inputs = nullptr;
    uint64_t input_stride = 0;     // lanes between consecutive stream words (0: n_lanes)
    uint32_t bound_input_words = 0;
};

class CS {
  public:
    CS(const zk_geometry& g, uint64_t max_trace_len, uint64_t max_variables);
    ~CS();

    // configuration
    void allow_lookup(uint32_t width, uint32_t reps, bool share_table_id);
    void allow_gate(uint32_t kind);
    bool gate_is_allowed(uint32_t kind) const;
    uint32_t add_table(uint32_t marker, uint32_t n_keys, uint32_t n_vals, const uint64_t* rows, uint32_t n_rows);
    uint32_t table_id(uint32_t marker) const;
    bool has_table(uint32_t marker) const;

    // recording
    zk_var alloc_var();
    zk_var alloc_vars(uint32_t n);
    zk_var alloc_constant(uint64_t value);
    zk_var input(uint32_t word);
    void place_gate(uint32_t kind, const zk_var* vars, uint32_t n_vars, const uint64_t* consts, uint32_t n_consts);
    void emit_op(uint32_t opcode, uint32_t a, uint32_t b, const zk_var* ins, uint32_t n_in, const zk_var* outs,
                 uint32_t n_out, const uint64_t* imm, uint32_t n_imm);
    void lookup(uint32_t table_id, const zk_var* keys, uint32_t n_keys, zk_var* vals, uint32_t n_vals);
    // gadget layer only (not in the C ABI): a lookup TUPLE over variables a macro-op produces (no ZK_OP_LOOKUP is recorded), and the
    // macro-ops themselves (ZK_OP_KECCAK_F): the fused check trusts such an op to evaluate the tuples and reduction gates placed on
    // its outputs, which holds because gadget and op walk one structure (csrc/keccak_macro.hpp)
    void lookup_given(uint32_t table_id, const zk_var* keys, uint32_t n_keys, const zk_var* vals, uint32_t n_vals);
    // A macro-op and the WINDOW of its gadget: between emit_macro_op and end_macro_op every gate placed / tuple given on the op's outputs is tagged
    // with the op (GateRec::owner, LookupRec::owner).  Only tagged gates count as "evaluated by the macro-op" in the fused check; the window
    // cannot be opened through the C ABI (zk_cs_place_gate on a macro output from outside stays in the check program).
    void emit_macro_op(uint32_t opcode, const zk_var* ins, uint32_t n_in, zk_var first_out, uint32_t n_out, uint32_t a = 0);
    void end_macro_op();
    bool uses_lookup_macros() const { return uses_lookup_macros_; }
    void side_begin();
    void loop_begin(uint32_t limit);
    void loop_end();
    void link(uint32_t kind, zk_var loop_var, zk_var other);
    void stream_link(const zk_var* a, uint32_t pa, const zk_var* b, uint32_t pb, uint32_t n_total);
    // seed-only macro-op: `outs` are variables ALREADY produced by recorded ops; in the seeding program this op produces
    // them directly from `ins` and the decomposition behind them drops out of the cone
    void seed_hint(uint32_t opcode, const zk_var* ins, uint32_t n_in, const zk_var* outs, uint32_t n_out);
    zk_var loop_last(zk_var loop_var);
    zk_var loop_import(zk_var outer_var);
    uint64_t next_available_row() const;
    void finalize();

    // execution
    void set_batch(uint32_t n_instances);
    void bind_inputs(bool loop_scope, const uint64_t* dev_words, uint32_t n_words, uint64_t lane_stride = 0);
    // public inputs of the whole batch packed on the device: out[instance * n_public + k]; returns n_public
    void debug_poke_store(bool loop_scope, uint32_t slot, uint32_t lane, uint64_t value);
    uint32_t store_slots(bool loop_scope) const { return (loop_scope ? loop_ : outer_).n_store; }
    uint32_t pack_public_inputs(uint64_t* dev_out, void* stream);
    // 0 equal; ZK_ERR_UNSATISFIED + first difference (scope 0, instance, slot = position in `vars`, kind = ZK_FAILURE_HOOK_DIFF)
    int check_satisfied_impl(void* stream, zk_failure* first, bool macro);
    int hook_compare_witness(const zk_var* vars, uint32_t n_vars, const uint64_t* dev_expected, void* stream, zk_failure* first);
    uint32_t batch() const { return batch_; }
    void seed_stream(uint32_t n_instances, const uint64_t* dev_outer_inputs, uint64_t* dev_loop_inputs_rw, void* stream, bool synchronize = true,
                     uint64_t outer_stride = 0, uint64_t loop_stride = 0);  // strides: lanes between words when the n instances are a window of a longer stream
    void resolve(void* stream);
    // sequential seeding of the carried input words (generic, slow): see kernels_engine.hpp k_witness_seq
    void seed_carried_inputs(uint64_t* dev_loop_inputs_rw, void* stream);
    void launch_seed(const zkdev::ScopeArgs& la, const zkdev::ScopeArgs& oa, uint64_t* dev_loop_inputs_rw, uint32_t n, void* stream);
    // chain-specialised seeding: a circuit whose carried state has a native walker registers it here (main_vm: kind 1).  Used by
    // launch_seed instead of the cone kernels unless ZKGL_SEED_NATIVE=0; last_seed_phase_ms: walker / chains / fill of the last pass.
    int native_seed_kind = 0;                       // 1: main_vm (walker + chains + fill); 2: ram_permutation (scans, needs the queue heads given); 3 / 4: keccak256 / sha256 round function FSM
    std::vector<zk_var> native_seed_outer_vars;     // outer-scope variables the native seeder reads (ram: challenges[r][1..8]); slots uploaded on first use
    uint32_t native_seed_param = 0;                 // ram: BOOTLOADER_HEAP_PAGE
    void set_seed_given(const uint32_t* loop_words, uint32_t n);
    bool seed_words_given(const uint32_t* words, uint32_t n) const;
    float last_seed_phase_ms[3] = {0, 0, 0};
    bool launch_seed_native(const zkdev::ScopeArgs& la, const zkdev::ScopeArgs& oa, uint64_t* dev_loop_inputs_rw, uint32_t n, void* stream);
    // first word of a field of the recorded input layout ("outer" / "loop"), UINT32_MAX when absent
    uint32_t layout_word(const char* scope, const char* name) const;
    int check_satisfied(void* stream, zk_failure* first);
    int resolve_and_check(void* stream, zk_failure* first);
    void set_check_mode(uint32_t mode);   // ZK_CHECK_FUSED / ZK_CHECK_STORED / ZK_CHECK_FUSED_DEFER_P2
    std::vector<uint32_t> narrow_byte_input_words() const;   // loop input words held in one-byte slots of the narrow store (empty: no layout)
    void ensure_p2_filled(void* stream);   // deferred mode: regenerate the Poseidon2 intermediates the last resolve_and_check left out
    uint64_t read_var(zk_var v, uint32_t instance, uint32_t iteration);
    void write_cell(bool loop_scope, uint32_t cell, uint32_t lane, uint64_t value);
    std::vector<uint64_t> public_inputs(uint32_t instance);
    uint32_t var_cell(zk_var v) const;
    std::vector<uint32_t> public_cells() const;
    std::vector<uint32_t> multiplicities(uint32_t instance);
    // K10: log-derivative lookup-argument accumulators over the resolved trace; out[instance] = {A.a, A.b, B.a, B.b};
    // returns the number of instances with A != B
    uint32_t lookup_argument(const uint64_t beta[2], const uint64_t gamma[2], void* stream, std::vector<uint64_t>& out);
    // K12 (cs_perm.cpp): copy-permutation grand product over the resolved trace.  out[instance] = {num.a, num.b, den.a, den.b} of
    // z[rows] = num / den; returns the number of instances with num != den.  d_z (optional, device): [batch][rows + 1][2]
    uint32_t copy_permutation(const uint64_t beta[2], const uint64_t gamma[2], void* stream, uint64_t* d_z, std::vector<uint64_t>& out);
    std::vector<uint64_t> sigma_labels(bool loop_scope, uint32_t iteration);  // sigma(label) of every trace cell of the scope / iteration
    void stats(zk_stats* out) const;
    float last_ms(int which) const;
    std::vector<uint32_t> export_scope(bool loop_scope) const;
    // columns of one instance's trace as polynomials of 2^log_n values (kernels_ntt.hpp k_trace_columns_*)
    void ensure_trace_view();
    void trace_columns(uint32_t instance, uint64_t* d_out, uint32_t log_n, uint64_t stride, void* stream, uint32_t n_instances = 1, uint64_t instance_stride = 0);
    void trace_ptr(bool loop_scope, uint64_t** cells, uint64_t* n_cells, uint64_t* stride);

    // circuit-layer attachments: the main_vm opcode-defs blob (include/zkgl_vm.h) handed to configure, and the text
    // description of the input streams the recorded circuit reads (zk_circuit_main_vm_layout)
    std::vector<uint8_t> circuit_blob;
    std::string input_layout;
    // closed-form-input variable groups a circuit publishes for zk_cs_hook_compare_witness ("hidden_fsm_output", ...)
    std::map<std::string, std::vector<zk_var>> hooks;

    const zk_geometry& geometry() const { return geo_; }
    bool in_loop() const { return in_loop_; }
    uint32_t limit() const { return limit_; }
    // the loop input words that are loop-carried (tied to the previous iteration's outputs): what seeding fills
    std::vector<uint32_t> carried_words() const { std::vector<uint32_t> w; for (auto& c : carries_store_) w.push_back(c.word); return w; }
    uint32_t lookup_width() const { return lookup_width_; }
    bool finalized() const { return finalized_; }
    // words of input the circuits layer registered (for zk_circuit_input_words)
    uint32_t outer_input_words() const { return outer_.n_input_words; }
    uint32_t loop_input_words() const { return loop_.n_input_words; }

  private:
    Scope& cur() { return in_loop_ ? loop_ : outer_; }
    Scope& scope_of(zk_var v) { return is_loop_var(v) ? loop_ : outer_; }
    uint32_t pool_const(Scope& s, uint64_t v);
    void place_scope(Scope& s);
    std::vector<OpRec> loop_ops_recorded_;   // the loop body as recorded (build_seed_program)
    void schedule_loop_ops();
    void schedule_by_locality(const std::vector<double>& a, const std::vector<double>& m, double a_tot, double m_tot,
                              const std::vector<std::vector<uint32_t>>& succ, std::vector<uint32_t>& n_pred);
    void emit_scope(Scope& s);
    void emit_op(const Scope& s, const OpRec& op, std::vector<uint32_t>& out) const;
    void emit_dests(const Scope& s, const OpRec& op, std::vector<uint32_t>& out) const;
    mutable bool emit_full_ = false;  // emit_dests: every cell of the variable (export) instead of the home cell only
    // compact trace -> full trace (kernels_engine.hpp k_materialize); no-op when the trace is already materialised
    void ensure_materialized(void* stream);
    bool compact_ = true;
    static constexpr uint32_t NARROW_STRANDS = 8;
    void build_strands(Scope& s, uint32_t n_strands = zkdev::STRANDS_PER_TILE, bool narrow = false);
    void assign_store_slots(Scope& s);
    uint32_t home(const Scope& s, uint32_t var) const { return emit_full_ ? s.var_cells[var][0] : s.var_slot[var]; }
    void check_streams(void* stream, bool compact, bool narrow = false);
    void check_inputs_canonical(void* outer_stream, void* loop_stream);
    // one witness launch: the plain program, or its strand form when the scope has too few wavefronts to fill the chip
    void launch_phase(const Scope& s, zkdev::ScopeArgs a, int phase, void* stream, uint32_t n_lanes = 0) const;  // n_lanes: lane count when the arguments were patched for a stream
    void build_check_program(Scope& s);
    // narrow store of the loop scope (store_geom.hpp): classes + address words (before emit_scope), the fused check program over them (after build_check_program)
    void build_narrow_layout(Scope& s);
    void build_narrow_check_program(Scope& s);
    bool loop_runs_strands(const Scope& s, int phase, uint32_t n_lanes) const;   // launch_phase's choice of the strand form
    void ensure_wide_store();            // narrow batches: the ordinary loop store exists (allocated with the batch when it fits, here otherwise)
    void build_mult_sites(Scope& s);
    void bound_values(Scope& s);
    void count_multiplicities(void* stream, int scopes = 3);
    // true: wave-aggregated atomics inside the witness kernels; false: the k_multiplicities pass after them (cs.cpp)
    bool inline_multiplicities() const;
    void operand_v2(const Scope& s, const OpRec& op, size_t pos, std::vector<uint32_t>& out) const;
    std::vector<uint32_t> select_plane_vars(const Scope& s) const;
    void verify_device_programs(const Scope& s) const;   // ZKGL_VERIFY_DEVICE_PROGRAMS=1: independent walk over prog2 / the strand programs
    void emit_group_v2(const Scope& s, const std::vector<size_t>& group, bool counted, std::vector<uint32_t>& out) const;
    const std::vector<uint32_t>* narrow_emit_ = nullptr;   // emit_scope, narrow form of a loop scope: store slot -> address word (operand_v2)
    const std::vector<uint32_t>* plane_of_ = nullptr;   // emit_scope, v2 form of a loop scope: variable -> SELECT flag plane id (UINT32_MAX: none)
    void upload_scope(Scope& s);
    void ensure_uploaded();
    void free_scope_device(Scope& s);
    void check_var(zk_var v, bool want_loop) const;
    int decode_failure(const unsigned long long* f, zk_failure* first) const;
    zkdev::CheckArgs check_args(const Scope& s, unsigned long long* fail, bool compact, bool macro = true, bool fused = false, bool narrow = false) const;

    zk_geometry geo_;
    uint64_t max_trace_len_, max_variables_;
    uint32_t lookup_width_ = 0, lookup_reps_ = 0;
    bool lookup_share_id_ = true;
    uint64_t allowed_gates_ = 0;
    std::vector<TableRec> tables_;  // id = index + 1
    Scope outer_, loop_;
    bool in_loop_ = false, loop_done_ = false, finalized_ = false;
    uint32_t limit_ = 0;
    uint32_t pre_vars_ = UINT32_MAX;  // outer variables allocated before side_begin (all of them when there is no side phase)
    std::vector<zk_link> links_raw_;  // vars, resolved to cells at finalize
    std::vector<zk_link> links_;       // endpoints as trace cells (export, materialised-trace check)
    std::vector<zk_link> links_store_; // endpoints as store slots (compact check)
    zk_link* d_links_store_ = nullptr;
    zk_link* d_links_store_n_ = nullptr;          // the same with the loop-scope endpoints as address words of the narrow store
    std::vector<uint32_t*> d_streams_store_n_;
    std::vector<uint32_t> loop_last_slots_;       // loop store slots ZK_OP_LOOP_LAST ops of the outer scope read (k_widen_last after a narrow loop launch)
    uint32_t* d_loop_last_slots_ = nullptr;
    std::vector<uint32_t> public_vars_;
    struct StreamRec { std::vector<uint32_t> a, b; uint32_t n_total; };  // loop var indices -> home cells at finalize
    std::vector<StreamRec> streams_raw_, streams_, streams_store_;
    std::vector<uint32_t*> d_streams_, d_streams_store_;  // per stream: a cells then b cells (trace cells / store slots)

    // device-wide
    uint32_t batch_ = 0;
    zk_table_desc* d_tables_ = nullptr;
    uint64_t* d_table_words_ = nullptr;
    uint32_t total_table_rows_ = 0;
    std::vector<zk_table_desc> tdesc_host_;
    std::vector<uint64_t> table_words_host_;
    bool uploaded_ = false;
    uint32_t* d_mult_ = nullptr;
    zk_link* d_links_ = nullptr;
    struct Carry { uint32_t word, out_cell, first_outer_cell, has_first; };
    std::vector<Carry> carries_;        // out / first cells as trace cells (export)
    std::vector<Carry> carries_store_;  // as store slots (device)
    void* d_carries_ = nullptr;
    // cone seeding: backward slice of the carried outputs over LDS slots (empty => generic sequential mode)
    void build_seed_program();
    std::vector<uint32_t> seed_prog_;
    std::vector<Carry> seed_carries_;  // out_cell = slot of the carried output
    uint32_t seed_slots_ = 0, seed_ops_ = 0;
    uint32_t* d_seed_prog_ = nullptr;
    void* d_seed_carries_ = nullptr;
    // strand form of the cone (8 wavefronts per block, level barriers; slots recycled per level)
    uint32_t* d_public_slots_ = nullptr;
    bool seed_v2_ok_ = false;
    bool seed_cone_unsupported_ = false;   // the cone holds an op no seed kernel runs (ZK_OP_BYTEBUF_FILL): seeding needs the native seeder
    // op-parallel seed program (k_seed_wave): 16-bit records, prologue segments then cycle segments
    std::vector<uint16_t> seed_wprog_;
    std::vector<Carry> seed_wcarries_;
    uint32_t seed_wslots_ = 0, seed_wpro_words_ = 0;
    uint16_t* d_seed_wprog_ = nullptr;
    void* d_seed_wcarries_ = nullptr;
    std::vector<uint32_t> seed_sprog_;
    std::vector<Carry> seed_scarries_;
    uint32_t seed_sslots_ = 0, seed_sbegin_[zkdev::STRANDS_PER_TILE] = {}, seed_send_[zkdev::STRANDS_PER_TILE] = {};
    float seed_sgain_ = 0;
    uint32_t* d_seed_sprog_ = nullptr;
    void* d_seed_scarries_ = nullptr;
    // K12: sigma = per-cell image inside the scope / iteration + absolute labels of the link endpoints [endpoint][iteration]
    void build_sigma();
    bool sigma_built_ = false;
    std::vector<uint32_t> sig_rel_[2], ep_index_[2];
    std::vector<uint64_t> ovr_[2];
    uint32_t n_ep_[2] = {0, 0};
    uint32_t* d_sig_rel_[2] = {nullptr, nullptr};
    uint32_t* d_ep_index_[2] = {nullptr, nullptr};
    uint64_t* d_ovr_[2] = {nullptr, nullptr};
    unsigned long long* d_fail_ = nullptr;
    // native seeding: device copies of the circuit blob and of the state-word -> outer slot table, scratch grown on demand
    void* d_native_blob_ = nullptr;
    uint32_t* d_state0_slot_ = nullptr;
    uint32_t* d_native_outer_slots_ = nullptr;
    std::vector<uint32_t> seed_given_words_;
    uint64_t* d_native_scratch_ = nullptr;
    size_t native_scratch_bytes_ = 0;
    uint64_t* d_seed_outer_ = nullptr;   // seed_stream's outer store
    size_t seed_outer_bytes_ = 0;
    void* ev_[8] = {nullptr};
    void* ev2_[9] = {nullptr};
    void* aux_stream_ = nullptr;
    float ms_[5] = {0, 0, 0, 0, 0};
    float loop_shader_mhz_ = 0;   // clock probe of the last resolve_and_check's loop launch (last_ms(8))
    bool last_check_fused_ = false;
    uint64_t p2_skipped_ = 0, p2_run_ = 0;   // gated witness-only permutations of the last resolve_and_check's loop launch, per wavefront
    bool check_stored_ = false;
    bool defer_p2_ = false;          // ZK_CHECK_FUSED_DEFER_P2
    bool p2_pending_ = false;        // the loop store lacks the intermediates of its in-circuit permutations (k_fill_p2 not run yet)
    bool narrow_enabled_ = false;    // the loop scope has a narrow layout with its programs
    bool narrow_active_ = false;     // the bound batch runs its fused steps over the narrow store (set_batch: plain loop kernel, inline multiplicities)
    bool narrow_pending_ = false;    // the last step wrote the narrow store only: the ordinary store is stale until k_widen_store (ensure_p2_filled)
    bool narrow_suspended_ = false;  // a step that reported a failure over the narrow store is being repeated over the ordinary one
    uint64_t narrow_steps_ = 0, narrow_repeats_ = 0;
    bool uses_lookup_macros_ = false;   // a macro-op whose outputs carry lookup tuples was recorded: multiplicities by the k_multiplicities pass
    bool uses_bytebuf_macro_ = false, uses_sha4_macro_ = false;   // macro-ops with their own kernel instantiations (launch_phase: ScopeArgs::xmacros)
    int32_t macro_window_op_ = -1;   // index (current scope) of the macro-op whose gadget window is open
    bool macro_window_loop_ = false;
    bool allow_macro_ops_ = false;   // zk_cs_set_check_mode(ZK_CHECK_STORED)
};

// K11 (ntt.cpp): batched Goldilocks NTT / coset LDE over device-resident polynomials, see include/zkgl.h
uint64_t two_adic_root(uint32_t log_n);
void ntt(uint64_t* d_data, uint32_t log_n, uint32_t n_polys, uint64_t stride, uint32_t mode, uint64_t coset_shift, void* stream);
void lde(const uint64_t* d_coeffs, uint64_t src_stride, uint64_t* d_out, uint32_t log_n, uint32_t log_blowup, uint32_t n_polys,
         uint32_t mode, uint64_t coset_shift, void* stream);

}  // namespace zkgl
