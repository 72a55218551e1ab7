/*
 * zkgl.h — C ABI of libzkgl.so, the MI355X-native witness-generation + constraint-evaluation
 * engine for the per-circuit hot path of matter-labs/era-zkevm_circuits.
 *
 * This is the drop-in boundary: what a Rust `impl ConstraintSystem<F>` shim
 * would bind through `extern "C"` (INTEGRATION.md shows the stub).  Plain pointers and
 * sizes only.  Every entry returns 0 on success, a negative zk_status otherwise, and never
 * aborts; `zk_last_error()` returns the message of the last failure on the calling thread.
 *
 * Reference interfaces replaced (all in the external crate `boojum`, as *used* by the
 * reference at the cited lines):
 *   field / hash primitives  <- boojum::field::goldilocks, boojum::implementations::poseidon2
 *                               (src/ram_permutation/mod.rs:405,411)
 *   zk_cs_*                  <- boojum::cs::traits::cs::ConstraintSystem + cs_builder
 *                               (call sites: src/ram_permutation/mod.rs:419-556,
 *                                src/main_vm/utils.rs:51-99)
 *   zk_circuit_*             <- the `*_entry_point` functions (src/ram_permutation/mod.rs:31)
 *
 * Device pointers are raw HIP device addresses (e.g. `torch.Tensor.data_ptr()`); `stream` is a
 * `hipStream_t` passed as void* (NULL = the null stream).
 */
#ifndef ZKGL_H
#define ZKGL_H
#include <stddef.h>
#include <stdint.h>
#include "zkgl_ir.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum zk_status {
    ZK_OK = 0,
    ZK_ERR_INVALID = -1,   /* bad argument / API misuse (reference: panic / expect) */
    ZK_ERR_HIP = -2,       /* HIP runtime failure, no usable GPU */
    ZK_ERR_UNRESOLVED = -3,/* a variable has no witness producer (reference: resolver hang/panic) */
    ZK_ERR_CAPACITY = -4,  /* trace exceeds max_trace_len / max_variables */
    ZK_ERR_UNSATISFIED = -5,
    ZK_ERR_GATE_NOT_ALLOWED = -6 /* reference: `unimplemented!()` src/main_vm/utils.rs:87-89 */
} zk_status;

const char *zk_last_error(void);
/* Macro-op device backends this library carries.  Since round 6 there is ONE build and it carries every device path of the tree (rounds 4-5 kept
 * unmeasured paths in variant libraries: they were promoted into kernels of their own, or deleted): the value is the constant
 * ZK_BUILD_BYTEBUF_KERNEL | ZK_BUILD_SHA4_KERNEL.  Bits 2, 4, 8, 32 named variants that no longer exist. */
#define ZK_BUILD_BYTEBUF_KERNEL 1u        /* ZK_OP_BYTEBUF_FILL on the device (recordings made with ZKGL_BYTEBUF_MACRO=1): kernels k_witness_*_x<X_BYTEBUF> */
#define ZK_BUILD_SHA4_KERNEL 16u          /* ZK_OP_SHA256_ROUNDS a = 1: the reference's 4-bit-chunk SHA tables as a macro-op (the default recording of
                                           * configure_sha256(reference_tables); ZKGL_SHA4_MACRO=0 records op by op): kernels k_witness_*_x<X_SHA4> */
uint32_t zk_build_features(void);
/* Select the device, upload Poseidon2 constants.  Fails loudly (ZK_ERR_HIP) without a GPU.  One device per process: a second
 * call with another device index is ZK_ERR_INVALID (process-wide device tables are bound to the first). */
int zk_init(int device);
int zk_device_count(void);
/* host-side derivation of the 360 round constants (no GPU needed) */
int zk_poseidon_round_constants(uint64_t out[360]);

/* ---------------- device memory helpers (thin hipMalloc/hipMemcpy wrappers) ---------------- */
int zk_malloc(void **dptr, size_t bytes);
int zk_free(void *dptr);
int zk_memset(void *dptr, int value, size_t bytes, void *stream);
int zk_h2d(void *dptr, const void *hptr, size_t bytes, void *stream);
int zk_d2h(void *hptr, const void *dptr, size_t bytes, void *stream);
int zk_sync(void *stream);

/* ---------------- K1: Goldilocks column arithmetic ----------------------------------------- */
/* dst[i] = q*a[i]*b[i] + l*c[i]   (FmaGateInBaseFieldWithoutConstant witness, column form) */
int zk_gl_fma_cols(uint64_t *dst, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                   uint64_t q, uint64_t l, size_t n, void *stream);
int zk_gl_add_cols(uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t n, void *stream);
int zk_gl_sub_cols(uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t n, void *stream);
int zk_gl_mul_cols(uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t n, void *stream);
/* dst[i] = s[i] ? a[i] : b[i] */
int zk_gl_select_cols(uint64_t *dst, const uint64_t *s, const uint64_t *a, const uint64_t *b,
                      size_t n, void *stream);
/* dst[i] = a[i]^-1, 0 -> 0 */
int zk_gl_inv_cols(uint64_t *dst, const uint64_t *a, size_t n, void *stream);

/* ---------------- K2: batched Poseidon2 permutation ---------------------------------------- */
/* column-major: element j of state i at states[j*stride + i]; in place */
int zk_poseidon2_permute_soa(uint64_t *states, size_t n, size_t stride, void *stream);
/* row-major: state i at states[12*i .. 12*i+11]; in place; staged through LDS */
int zk_poseidon2_permute_aos(uint64_t *states, size_t n, void *stream);

/* ---------------- K3: sponge chains --------------------------------------------------------- */
/* commit_encoding (src/fsm_input_output/mod.rs:281-326) for n independent encodings of equal
 * length `len`; input[j*n + i] = element j of encoding i; out[j*n + i], j < 4 */
int zk_commit_encoding_batch(const uint64_t *input, size_t len, size_t n, uint64_t *out, void *stream);
/* full-state queue hash chains (src/main_vm/utils.rs:194-213): nq independent queues, each
 * pushing `items` encodings; enc[(q*items + t)*8 + j]; tail_io[q*12 + j] in/out;
 * if states_out != NULL, states_out[(q*items + t)*12 + j] = tail BEFORE push t */
int zk_queue_full_push_chain(const uint64_t *enc, size_t nq, size_t items, uint64_t *tail_io,
                             uint64_t *states_out, void *stream);

/* ---------------- a9: MemoryQuery::encode, column form ------------------------------------- */
/* q[f*n + i], f < 13 (ts, page, index, rw, is_ptr, value limbs 0..7); enc[j*n + i], j < 8 */
int zk_memory_query_encode(const uint64_t *q, size_t n, uint64_t *enc, void *stream);

/* ---------------- a11: ExecutionContextRecord::encode, column form ------------------------- */
/* rec[f*n + i], f < 42 in declaration order (src/base_structures/vm_state/saved_context.rs:36-66);
 * enc[j*n + i], j < 32 (saved_context.rs:111-266) */
int zk_execution_context_encode(const uint64_t *rec, size_t n, uint64_t *enc, void *stream);

/* ---------------- K4: permutation grand product -------------------------------------------- */
/* enc[j*n + i] (j < enc_len), flags[i] in {0,1}, challenges[enc_len+1];
 * acc_out[i] = init * prod_{t<=i, flags[t]} (ch[enc_len] + sum_j enc[j][t]*ch[j])
 * (src/utils.rs:81-137, one repetition).  scratch: >= n u64 device words. */
int zk_grand_product(const uint64_t *enc, const uint64_t *flags, const uint64_t *challenges,
                     size_t enc_len, size_t n, uint64_t init, uint64_t *acc_out,
                     uint64_t *scratch, void *stream);

/* ---------------- constraint-system recorder + GPU executor -------------------------------- */
typedef struct zk_cs zk_cs; /* opaque handle; NOT thread-safe, one recording thread, one device */
typedef uint32_t zk_var;
#define ZK_VAR_NONE 0xffffffffu

typedef struct zk_geometry { /* boojum::cs::CSGeometry, src/main_vm/cycle.rs:959-966 */
    uint32_t num_columns_under_copy_permutation;
    uint32_t num_witness_columns;
    uint32_t num_constant_columns;
    uint32_t max_allowed_constraint_degree;
} zk_geometry;

int zk_cs_create(const zk_geometry *geometry, uint64_t max_trace_len, uint64_t max_variables,
                 zk_cs **out);
int zk_cs_destroy(zk_cs *cs);
/* LookupParameters::UseSpecializedColumnsWithTableIdAsConstant (src/ram_permutation/mod.rs:435-441) */
int zk_cs_allow_lookup(zk_cs *cs, uint32_t width, uint32_t num_repetitions, int share_table_id);
/* G::configure_builder(..) (src/ram_permutation/mod.rs:442-483) */
int zk_cs_allow_gate(zk_cs *cs, uint32_t gate_kind);
int zk_cs_gate_is_allowed(zk_cs *cs, uint32_t gate_kind); /* 1 / 0 */
/* add_lookup_table::<T, W> (src/ram_permutation/mod.rs:500-501). rows[r*(n_keys+n_vals) + c].
 * marker = caller-chosen table identity (the Rust type in the reference). Returns table id >= 1 in *id. */
int zk_cs_add_table(zk_cs *cs, uint32_t marker, uint32_t n_keys, uint32_t n_vals,
                    const uint64_t *rows, uint32_t n_rows, uint32_t *id);
int zk_cs_table_id(zk_cs *cs, uint32_t marker, uint32_t *id); /* get_table_id_for_marker */

/* --- recording (single thread) --- */
int zk_cs_alloc_vars(zk_cs *cs, uint32_t n, zk_var *first);          /* alloc_multiple_variables_without_values */
int zk_cs_alloc_constant(zk_cs *cs, uint64_t value, zk_var *out);    /* allocate_constant */
int zk_cs_input(zk_cs *cs, uint32_t word, zk_var *out);              /* witness value from the bound input stream */
/* Gate::add_to_cs: vars in the kind's column order, consts = the row-shared parameters */
int zk_cs_place_gate(zk_cs *cs, uint32_t gate_kind, const zk_var *vars, uint32_t n_vars,
                     const uint64_t *consts, uint32_t n_consts);
/* set_values_with_dependencies with a closure from the closed op set (zk_opcode) */
int zk_cs_emit_op(zk_cs *cs, uint32_t opcode, uint32_t a, uint32_t b, const zk_var *ins,
                  uint32_t n_in, const zk_var *outs, uint32_t n_out, const uint64_t *imm,
                  uint32_t n_imm);
/* perform_lookup::<K,V> (src/main_vm/decoded_opcode.rs:492): allocates V outputs */
int zk_cs_lookup(zk_cs *cs, uint32_t table_id, const zk_var *keys, uint32_t n_keys, zk_var *vals,
                 uint32_t n_vals);
/* loop scope: the `for _cycle in 0..limit` body (src/ram_permutation/mod.rs:246) is recorded ONCE */
/* optional: outer-scope work recorded between zk_cs_side_begin and zk_cs_loop_begin neither feeds the loop
 * nor depends on it; the fused pipeline runs it concurrently with the loop kernel */
int zk_cs_side_begin(zk_cs *cs);
int zk_cs_loop_begin(zk_cs *cs, uint32_t limit);
int zk_cs_loop_end(zk_cs *cs);
int zk_cs_link(zk_cs *cs, uint32_t link_kind, zk_var loop_var, zk_var other_var);
/* stream link (include/zkgl_ir.h): loop variables a_vars[k % period_a] of iteration k / period_a and
 * b_vars[k % period_b] of iteration k / period_b are the same value for every k < n_total */
int zk_cs_stream_link(zk_cs *cs, const zk_var *a_vars, uint32_t period_a, const zk_var *b_vars, uint32_t period_b, uint32_t n_total);
/* seed hint (loop scope): `outs` are variables already produced by recorded ops; the seed-only macro-op `opcode`
 * (ZK_OP_KECCAK_ABSORB: 336 ins / 200 outs, ZK_OP_SHA256_COMPRESS: 96 ins / 32 outs, byte variables) produces them directly in
 * the cone seeding program, so the gate-by-gate decomposition behind them is not replayed by zk_cs_seed_carried_inputs.
 * The trace program, the gates and the checks are unaffected. */
int zk_cs_seed_hint(zk_cs *cs, uint32_t opcode, const zk_var *ins, uint32_t n_in, const zk_var *outs, uint32_t n_out);
/* value of a loop variable at the last iteration, as an outer variable (post phase) */
int zk_cs_loop_last(zk_cs *cs, zk_var loop_var, zk_var *outer_out);
/* use an outer variable inside the loop (broadcast; pre phase must define it) */
int zk_cs_loop_import(zk_cs *cs, zk_var outer_var, zk_var *loop_out);
int zk_cs_next_available_row(zk_cs *cs, uint64_t *row);
/* pad_and_shrink + into_assembly: placement, program emission, upload */
int zk_cs_finalize(zk_cs *cs);

/* --- execution --- */
int zk_cs_set_batch(zk_cs *cs, uint32_t n_instances); /* allocates device trace for the batch */
/* input streams: outer scope words[w*B + inst]; loop scope words[w*(B*limit) + inst*limit + k]; device ptrs */
int zk_cs_bind_inputs(zk_cs *cs, int loop_scope, const uint64_t *dev_words, uint32_t n_words);
/* a batch that is a WINDOW of a longer stream: dev_words points at the window's first lane, consecutive words are lane_stride
 * lanes apart (outer scope: the stream's instance count; loop scope: instances * limit) */
int zk_cs_bind_inputs_window(zk_cs *cs, int loop_scope, const uint64_t *dev_words, uint32_t n_words, uint64_t lane_stride);
int zk_cs_resolve(zk_cs *cs, void *stream);          /* witness generation */
/* Generic sequential seeding: fills the loop-carried words of the bound (writable) loop input stream
 * from the circuit's own recurrence, one iteration after another, lane == instance.  Needed only when
 * the host has the raw witness but not the per-iteration state (the reference's closures get exactly
 * that); hosts that already know the per-cycle state (e.g. VmLocalState per cycle) skip it. */
int zk_cs_seed_carried_inputs(zk_cs *cs, uint64_t *dev_loop_inputs_rw, void *stream);
/* The loop-scope input words that are loop-carried (CARRY links onto stream words): the words seeding fills and a host may leave
 * blank.  words = NULL returns the count. */
int zk_cs_carried_words(zk_cs *cs, uint32_t *words, uint32_t max_words, uint32_t *n_words);
/* Declares loop-carried words the host fills itself in the streams it hands to the seeding entry points from now on (e.g. queue heads
 * taken from the previous tails the reference's queue witnesses carry).  A circuit's native seeder that needs no chain once they are
 * given takes the parallel path (ram_permutation); the generic cone kernels recompute and overwrite them with the same values.
 * n_words = 0 clears the declaration.  ZK_ERR_INVALID: a word that is not loop-carried. */
int zk_cs_set_seed_given(zk_cs *cs, const uint32_t *loop_words, uint32_t n_words);
/* The same over a stream of n_instances, independent of zk_cs_set_batch (layouts: outer words[w*n + inst], loop
 * words[w*(n*limit) + inst*limit + k]).  Seeding is a latency chain of `limit` iterations per instance: one pass over ~1000
 * instances costs what a pass over 8 does, so a host seeds a long stream once and resolves it in windows
 * (zk_cs_bind_inputs_window).  Replaces the sequential part of the reference's witness resolution
 * (/root/reference/src/main_vm/mod.rs: the `for _cycle_idx in 0..limit` loop over vm_cycle). */
int zk_cs_seed_stream(zk_cs *cs, uint32_t n_instances, const uint64_t *dev_outer_inputs, uint64_t *dev_loop_inputs_rw, void *stream);
/* A WINDOW of a longer stream, without the final synchronisation: n_instances consecutive instances whose first lane the two pointers
 * address (outer word w of instance i at dev_outer_window[w * outer_lane_stride + i], loop word w of cycle c at
 * dev_loop_window_rw[w * loop_lane_stride + i * limit + c]; a stride of 0 = dense).  The kernels are queued on `stream` and the call
 * returns: a host seeds the next window of raw witness on one HIP stream while zk_cs_resolve_and_check works through the previous one
 * on another (seeding is a latency chain on a few hundred wavefronts, the step kernels are bandwidth-bound: they overlap).  One seeding
 * pass at a time per zk_cs (its scratch buffers are reused); the window must not be read until the stream has been synchronised. */
int zk_cs_seed_window_async(zk_cs *cs, uint32_t n_instances, const uint64_t *dev_outer_window, uint64_t outer_lane_stride,
                            uint64_t *dev_loop_window_rw, uint64_t loop_lane_stride, void *stream);
typedef struct zk_failure { uint32_t scope, instance, iteration, slot, kind, relation; } zk_failure;
/* kind: a zk_gate_kind; 0x100 lookup tuple (relation = tuple); 0x200 copy constraint; 0x300 | zk_link_kind: a link (slot = the loop
 * cell, relation = link index); ZK_FAILURE_STREAM_LINK: a stream link (relation = stream index);
 * ZK_FAILURE_NONCANONICAL_INPUT: input stream word `slot` (mod 256) of that lane is not a canonical field element (>= p) */
/* numbering: 0x400 has meant NONCANONICAL_INPUT since the kind was introduced (a stream-link failure was reported under it too for a
 * while); stream links got their own value 0x500 — hosts built against either earlier header keep reading 0x400 correctly */
#define ZK_FAILURE_NONCANONICAL_INPUT 0x400u
#define ZK_FAILURE_STREAM_LINK 0x500u
/* check_if_satisfied: 0 satisfied; ZK_ERR_UNSATISFIED + first failure otherwise */
int zk_cs_check_satisfied(zk_cs *cs, void *stream, zk_failure *first);
/* fused resolve + check_if_satisfied; the latency-bound outer scope runs on an internal second stream
 * concurrently with the loop-scope kernels.  Same result contract as zk_cs_check_satisfied. */
int zk_cs_resolve_and_check(zk_cs *cs, void *stream, zk_failure *first);
/* How zk_cs_resolve_and_check evaluates the relations (zk_stats.constraints_from_store_fused / constraints_in_witness_fused):
 *   ZK_CHECK_FUSED (default)  gates mirrored by the witness op that produces their output and lookup tuples are evaluated by the
 *                             witness kernels on the values they hold; the check kernels read the rest from the store;
 *   ZK_CHECK_STORED           every relation is re-evaluated from the stored values, as zk_cs_check_satisfied does
 *                             (check_if_satisfied, /root/reference/src/ram_permutation/mod.rs:556).
 * Both give the same verdict for every input stream (tests/test_fused_differential.py).  The environment variable
 * ZKGL_VERIFY_STORED=1 forces ZK_CHECK_STORED for every zk_cs of the process. */
#define ZK_CHECK_FUSED 0u
#define ZK_CHECK_STORED 1u
/* ZK_CHECK_FUSED_DEFER_P2: the fused mode, and the loop kernel does not write the 950 intermediates of an in-circuit Poseidon2
 * permutation (main_vm: 8 550 of a cycle's 17 700 values) — nothing in the fused step reads them.  They are regenerated from the 12
 * stored inputs, bit for bit, the first time anything reads the store beyond the step: zk_cs_check_satisfied, zk_cs_trace_columns*,
 * zk_cs_trace_ptr, the prover-stage entry points, zk_cs_read_var, the fault-injection hooks (k_fill_p2).  Same verdicts, same
 * values; a different split of the work between the step and its readers, reported separately by bench.py. */
#define ZK_CHECK_FUSED_DEFER_P2 2u
int zk_cs_set_check_mode(zk_cs *cs, uint32_t mode);
/* ZK_CHECK_FUSED_DEFER_P2 only: write the values the last zk_cs_resolve_and_check left out now (every reader does it implicitly; a host
 * that wants the cost on its own clock calls this).  No-op otherwise. */
int zk_cs_complete_store(zk_cs *cs, void *stream);
/* NARROW STORE (zk_stats, csrc/store_geom.hpp): the loop input stream words whose variables the narrow layout holds in one-byte slots, i.e. the
 * words the circuit's own constraints bound below 2^8 in every satisfying witness (a larger word there is reported by the step like any other
 * violated range check).  buf = NULL for the count; 0 words when the circuit has no narrow layout.  zk_cs_complete_store also expands the
 * narrow store into the ordinary one when the last step left it pending. */
int zk_cs_narrow_byte_input_words(zk_cs *cs, uint32_t *buf, size_t max_words, size_t *n_words);
int zk_cs_read_var(zk_cs *cs, zk_var var, uint32_t instance, uint32_t iteration, uint64_t *out); /* witness_hook */
/* hook_compare_witness (/root/reference/src/fsm_input_output/mod.rs:102-133) as a device-side diff: the circuit's values of the outer
 * variables `vars` (the closed-form input the host cares about: hidden_fsm_output, observable_output ...; recorded handles) against
 * dev_expected[k * batch + instance].  0 when equal; ZK_ERR_UNSATISFIED and *first = {instance, slot = position k in `vars`,
 * kind = ZK_FAILURE_HOOK_DIFF} of the first difference (the reference panics with the pretty-printed diff). */
/* the variable groups a recorded circuit publishes for the comparison, by the reference's field name ("hidden_fsm_output"); *n = group size */
int zk_circuit_hook_vars(zk_cs *cs, const char *name, zk_var *vars, uint32_t max, uint32_t *n);
#define ZK_FAILURE_HOOK_DIFF 0xfeu
int zk_cs_hook_compare_witness(zk_cs *cs, const zk_var *vars, uint32_t n_vars, const uint64_t *dev_expected, void *stream, zk_failure *first);
int zk_cs_write_cell(zk_cs *cs, int loop_scope, uint32_t cell, uint32_t lane, uint64_t value); /* fault injection for tests */
/* fault injection below write_cell: overwrite one value of the VARIABLE STORE after zk_cs_resolve (the trace stays compact), slot <
 * zk_cs_store_slots; a following zk_cs_check_satisfied sees the corrupted witness value */
int zk_cs_debug_poke_store(zk_cs *cs, int loop_scope, uint32_t slot, uint32_t lane, uint64_t value);
int zk_cs_store_slots(zk_cs *cs, int loop_scope, uint32_t *n);
int zk_cs_public_inputs(zk_cs *cs, uint32_t instance, uint64_t *out, uint32_t max, uint32_t *n);
/* ---- the path's only collective: instances are sharded across GPUs with no data-path exchange; the 4-element
 * input commitments of every instance are all-gathered once per step over RCCL / xGMI (32 B per instance: latency-bound).
 * One process per GPU.  Rank 0 draws the id, the host's launcher distributes it, every rank creates its communicator after
 * zk_init (the communicator binds to that device).  Replaces nothing in the reference (its circuits are synthesised one at a
 * time on CPU threads, /root/reference/src/main_vm/mod.rs); it is what a multi-GPU host of this library needs instead. */
#define ZK_COMM_ID_BYTES 128
typedef struct zk_comm zk_comm;
int zk_comm_unique_id(uint8_t id[ZK_COMM_ID_BYTES]);
int zk_comm_create(zk_comm **out, const uint8_t id[ZK_COMM_ID_BYTES], int rank, int world);
int zk_comm_destroy(zk_comm *comm);
/* dev_out[rank][instance][k], k < *n_public (the circuit's public inputs: 4 for every circuit of the path), u64 device buffer of
 * world * batch * n_public words; every rank holds the same batch size.  Enqueued on `stream`. */
int zk_cs_gather_commitments(zk_cs *cs, zk_comm *comm, uint64_t *dev_out, uint32_t *n_public, void *stream);
/* placement query (after finalize, no GPU needed): home cell of a variable / of the public inputs */
int zk_cs_var_cell(zk_cs *cs, zk_var var, uint32_t *cell);
int zk_cs_public_cells(zk_cs *cs, uint32_t *cells, uint32_t max, uint32_t *n);
/* lookup multiplicities of one instance: mult[table_row_global] (u32), n = total rows of all tables */
int zk_cs_multiplicities(zk_cs *cs, uint32_t instance, uint32_t *out, uint32_t max, uint32_t *n);

typedef struct zk_stats {
    uint64_t rows_per_instance;      /* trace rows used (loop slots*limit + outer slots) */
    uint64_t loop_slots, outer_slots, limit;
    uint64_t copy_columns, lookup_columns;
    uint64_t variables_outer, variables_loop;
    uint64_t constraints_per_instance; /* relations of placed gate instances + lookup tuples */
    uint64_t var_cells_per_instance;   /* trace cells (rows * variable columns) */
    uint64_t gate_instances[16];     /* per instance, indexed by zk_gate_kind (16 slots: the struct does not change size with the gate set) */
    uint64_t lookups_per_instance;
    uint64_t program_words_outer, program_words_loop;
    uint64_t scratch_cells_outer, scratch_cells_loop;
    uint64_t cells_written_outer, cells_written_loop; /* words one lane's witness kernels store: one per variable (home cells) */
    uint64_t copy_pairs_outer, copy_pairs_loop;
    /* cone seeding program (backward slice of the carried outputs; 0 when the generic sequential mode is used) */
    uint64_t seed_ops, seed_words, seed_slots, loop_ops;
    uint64_t cells_populated_outer, cells_populated_loop; /* trace + scratch cells holding a value == cells the gate checker reads */
    /* lane tiling of the loop scope's variable store for the bound batch (0 before zk_cs_set_batch): 64 = one tile per wavefront,
     * 4096 = the wide tiling large batches get (csrc/store_geom.hpp).  A layout property only: every reader goes through the C ABI. */
    uint64_t loop_store_tile_lanes;
    /* Where zk_cs_resolve_and_check's default (fused) mode evaluates the relations counted in constraints_per_instance:
     *   constraints_from_store_fused   — relations the CHECK kernels evaluate on stored values: enforcements, booleans / range checks of
     *                                    inputs, integer add / multiply relations, relations whose output is a given variable;
     *   constraints_in_witness_fused   — relations of gates mirrored by the witness op that produces their output (proved per gate at
     *                                    finalize) + lookup tuples: evaluated by the witness kernel on the values it holds (SELECT's
     *                                    exception and lookup misses are tested there), never re-read from memory.
     * The two add up to constraints_per_instance.  With ZKGL_VERIFY_STORED=1 / zk_cs_check_satisfied every relation is evaluated from
     * the stored values. */
    uint64_t constraints_from_store_fused, constraints_in_witness_fused;
    /* census: variables (of cells_written_*) that are < 2^32 in EVERY satisfying witness, by the constraints alone (lookup-table membership,
     * boolean / constant gates, non-wrapping reductions and products of bounded terms, selections by a boolean selector) — what a store
     * with 4-byte slots could hold in half the bytes */
    uint64_t values_below_2_32_outer, values_below_2_32_loop;
    /* 1: the seed kernels are not offered this circuit's cone (zk_cs_seed_* answers ZK_ERR_INVALID unless a native seeder is registered):
     * it holds ZK_OP_BYTEBUF_FILL, or a carried output depends on a gated ZK_OP_POSEIDON2 other than through a select on that op's flag */
    uint64_t seed_cone_unsupported;
    /* NARROW STORE of the loop scope (csrc/store_geom.hpp; a batch takes it when ZKGL_NARROW_STORE=1 is set at zk_cs_set_batch — opt-in).  Values that are bytes in
     * every satisfying witness (the census above) live in one-byte slots of the variable store the fused step writes and reads:
     *   store_bytes_per_lane_loop         — bytes one loop lane's witness kernel writes into the ordinary store (8 per value)
     *   narrow_store_bytes_per_lane_loop  — the same with the narrow layout (0: the circuit has none — macro-op / big-integer loop scope, or ZKGL_NARROW_STORE=0 at finalize)
     *   narrow_byte_values_loop           — values of a lane held in one-byte slots
     *   narrow_store_active               — 1: the bound batch runs its fused steps over the narrow store (zk_cs_set_batch decides: plain loop
     *                                       kernel, inline multiplicities); every other reader sees the ordinary store, expanded on demand
     *   narrow_steps / narrow_repeats     — fused steps run over the narrow store / of those, steps repeated over the ordinary store because
     *                                       something was reported (a violated relation, or a value that does not fit its slot): verdicts and
     *                                       reported gates are always those of the ordinary store */
    uint64_t store_bytes_per_lane_loop, narrow_store_bytes_per_lane_loop, narrow_byte_values_loop, narrow_store_active, narrow_steps, narrow_repeats;
    uint64_t narrow_store_pending;   /* 1: the last step's values are in the narrow store only (zk_cs_trace_columns* read them there; any other reader expands them first) */
} zk_stats;
/* K12 — copy-permutation grand product over the resolved trace (SURVEY 8f-3 "copy-permutation grand product z(X)"; boojum's
 * column chunking and cell identifiers are [EXT], the argument is defined in csrc/kernels_perm.hpp).  Labels: outer-scope
 * trace cell c -> c; loop-scope cell c of iteration k -> NT_outer + k * NT_loop + c.  sigma permutes the labels with one cycle
 * per copy class (cells of a variable, joined across iterations and scopes by the links).  Rows of an instance: iteration *
 * loop_slots + slot, then the outer scope's slots (as in zk_cs_trace_columns).  z[0] = 1, z[r + 1] = z[r] * prod over the
 * populated cells of row r of (w + beta * label + gamma) / (w + beta * sigma(label) + gamma) in GF(p^2), X^2 = 7.
 *   out (4 words per instance, may be NULL): numerator (a, b) and denominator (a, b) of z[rows]; *n_mismatch = instances with
 *   z[rows] != 1;  dev_z (device, may be NULL): [instances][rows + 1][2] words, the column z itself.
 * zk_cs_sigma: sigma(label) for every trace cell of the scope (loop scope: of the given iteration); buf = NULL for the size. */
int zk_cs_copy_permutation(zk_cs *cs, const uint64_t beta[2], const uint64_t gamma[2], void *stream, uint64_t *dev_z, uint64_t *out,
                           uint32_t max_instances, uint32_t *n_mismatch);
int zk_cs_sigma(zk_cs *cs, int loop_scope, uint32_t iteration, uint64_t *buf, size_t max_words, size_t *n_words);
/* K11 — batched Goldilocks NTT and coset low-degree extension over device-resident polynomials (prover stage after
 * satisfiability, SURVEY 8f-3 "LDE/NTT over Goldilocks"; boojum's transforms are [EXT], entry implied by `into_assembly`,
 * /root/reference/src/ram_permutation/mod.rs:554).  Defined here: omega_N = 7^((p-1)/N), N = 2^log_n, g = coset_shift,
 *   A[k] = sum_i a[i] (g omega_N^k)^i.
 * zk_ntt works in place over n_polys polynomials, polynomial q at dev_data + q * stride (stride >= N); log_n <= 30; g must
 * be a nonzero canonical field element (1 = the subgroup itself).  mode bits:
 *   0                          forward: natural-order coefficients -> values, A[k] stored at bitrev(k)
 *   ZK_NTT_INVERSE             the inverse map of the forward mode with the same other bits
 *   ZK_NTT_NATURAL_VALUES      values in natural order (A[k] at k: trace rows), coefficients in bit-reversed order (a[i] at bitrev(i))
 * zk_lde: out[q][j][.] (q < n_polys, j < 2^log_blowup, stride N * 2^log_blowup per polynomial) = forward transform of
 * polynomial q on the coset g * eta^bitrev(j) * <omega_N>, eta = omega_{N * 2^log_blowup}; with mode 0 the blocks of one
 * polynomial read together are the size-(N * 2^log_blowup) forward transform of its zero-padded coefficients; mode may be
 * 0 or ZK_NTT_NATURAL_VALUES (bit-reversed coefficients in, natural-order values per coset out).  Source untouched. */
#define ZK_NTT_INVERSE 1u
#define ZK_NTT_NATURAL_VALUES 2u
int zk_two_adic_root(uint32_t log_n, uint64_t *out);
int zk_ntt(uint64_t *dev_data, uint32_t log_n, uint32_t n_polys, uint64_t stride, uint32_t mode, uint64_t coset_shift, void *stream);
int zk_lde(const uint64_t *dev_coeffs, uint64_t src_stride, uint64_t *dev_out, uint32_t log_n, uint32_t log_blowup, uint32_t n_polys,
           uint32_t mode, uint64_t coset_shift, void *stream);
/* K10 — log-derivative lookup-argument accumulators over the resolved trace (prover stage after satisfiability, SURVEY 8f-3;
 * boojum's polynomial form is [EXT], the sums are defined in csrc/kernels_lookup_arg.hpp).  beta, gamma: canonical GF(p^2)
 * elements (a + bX, X^2 = 7).  out (4 words per instance, may be NULL): witness-side sum A (a, b) then table-side sum B (a, b);
 * *n_mismatch = number of instances with A != B.  Call after zk_cs_resolve / zk_cs_resolve_and_check. */
int zk_cs_lookup_argument(zk_cs *cs, const uint64_t beta[2], const uint64_t gamma[2], void *stream, uint64_t *out, uint32_t max_instances,
                          uint32_t *n_mismatch);
int zk_cs_stats(zk_cs *cs, zk_stats *out);           /* print_gate_stats counterpart */
/* last execution times in ms measured with HIP events on the execution stream:
 * which: 0 resolve total (fused: whole pipeline), 1 loop witness kernel, 2 check total (fused: loop gates+copies),
 * 3 gate-check loop kernel, 4 outer kernels (fused: outer post + outer checks);
 * 5 / 6 / 7: the last seeding pass of a circuit with a native seeder (main_vm) when ZKGL_SEED_PHASE_MS is set: state walker,
 * Poseidon2 chains, fill;
 * 8: not a time — the shader clock in MHz that the loop witness kernel of the last zk_cs_resolve_and_check ran at (s_memtime against
 * the constant 100 MHz counter over the grid's first wavefront): the chip's power management picks it per launch;
 * 9: not a time — the share of the gated witness-only permutations (simulate_round_function(cs, state, execute), ZK_OP_POSEIDON2 a = 1)
 * that the wavefronts of the last loop launch skipped because all their 64 cycles had the flag off */
int zk_cs_last_ms(zk_cs *cs, int which, float *ms);
/* serialised scope (program + descriptors) for the CPU oracle / offline tooling.
 * Call with buf = NULL to get the size in words. */
int zk_cs_export(zk_cs *cs, int loop_scope, uint32_t *buf, size_t max_words, size_t *n_words);
/* The resolved trace of one instance as column polynomials for K11: dev_out[col * stride + row] for every copy and lookup
 * column, row = iteration * loop_slots + slot for the loop scope's rows, then the outer scope's slots, zero padded to
 * 2^log_n (>= the instance's row count, zk_stats.rows_per_instance); stride >= 2^log_n.  The layout `into_assembly`
 * hands to the prover (/root/reference/src/ram_permutation/mod.rs:554) is boojum's ([EXT]); this one is the engine's. */
int zk_cs_trace_columns(zk_cs *cs, uint32_t instance, uint64_t *dev_out, uint32_t log_n, uint64_t stride, void *stream);
/* The same for instances [first_instance, first_instance + n_instances) of the bound batch in ONE pass: instance i at
 * dev_out + (i - first_instance) * instance_stride (instance_stride >= (columns - 1) * stride + 2^log_n words).  A batch's columns are
 * 4x its variable store (main_vm: 1.375 GB per instance), so a host materialises them in chunks it has room for. */
int zk_cs_trace_columns_batch(zk_cs *cs, uint32_t first_instance, uint32_t n_instances, uint64_t *dev_out, uint32_t log_n, uint64_t stride,
                              uint64_t instance_stride, void *stream);
int zk_cs_trace_ptr(zk_cs *cs, int loop_scope, uint64_t **dev_cells, uint64_t *n_cells, uint64_t *stride);

/* ---------------- circuits (host side mirrors of the reference entry points) ---------------- */
/* ram_permutation_entry_point (src/ram_permutation/mod.rs:31-210) recorded with `limit` cycles.
 * Input stream layouts are documented in DESIGN.md §ram_permutation. */
int zk_circuit_ram_permutation(zk_cs *cs, uint32_t limit);
/* configure a CS the way the reference test does (geometry 100/0/8/4, xor8 table, gate set):
 * src/ram_permutation/mod.rs:419-501 */
int zk_circuit_ram_permutation_configure(zk_cs *cs);
/* number of input words per lane for (outer, loop) scopes of the recorded circuit */
int zk_circuit_input_words(zk_cs *cs, uint32_t *outer_words, uint32_t *loop_words);
/* sort_and_deduplicate_storage_access_entry_point (src/storage_validity_by_grand_product/mod.rs:166-506).
 * enforce_permutation = 0 drops the final lhs == rhs enforcement, i.e. checks what the reference's own test
 * checks (the `_inner` function on the fixture of test_input.rs, mod.rs:1034-1135). */
int zk_circuit_storage_validity_configure(zk_cs *cs);
int zk_circuit_storage_validity(zk_cs *cs, uint32_t limit, int enforce_permutation);
/* sort_and_deduplicate_events_entry_point (src/log_sorter/mod.rs:34-232) */
int zk_circuit_log_sorter_configure(zk_cs *cs);
int zk_circuit_log_sorter(zk_cs *cs, uint32_t limit);
/* Keccak-256 over n_blocks pre-padded 136-byte blocks, Keccak-f[1600] through 8-bit lookup tables
 * (keccak256_absorb_and_run_permutation, src/keccak256_round_function/mod.rs:796-838; eip_4844's keccak256
 * gadget, src/eip_4844/mod.rs:156-163).  Public inputs = the 32 digest bytes. */
int zk_circuit_keccak_configure(zk_cs *cs);
int zk_circuit_keccak256_blocks(zk_cs *cs, uint32_t n_blocks);
/* Hash permutations as gadget calls of a circuit recorded through this ABI (what the crate's circuits reach through boojum's
 * keccak256 / sha256 round functions: src/keccak256_round_function/mod.rs:796-838, src/sha256_round_function/mod.rs:271-285).
 * Variables of the CURRENT scope, bytes, range-checked by the caller (the xor tables of the first round do it for Keccak); the CS must
 * carry the table set of zk_circuit_keccak_configure / zk_circuit_sha256_configure{,_reference_tables}.  Records the permutation (one
 * macro witness op + the lookup tuples and reduction gates its gadget places) and replaces state_io by the output byte variables.
 * Keccak: state_io[8 (x + 5 y) + k] = byte k (little-endian) of lane (x, y).  SHA-256: state_io[4 w + k] / block[4 w + k] = byte k
 * (little-endian) of working word w / message word w. */
int zk_gadget_keccak_f1600(zk_cs *cs, zk_var state_io[200]);
int zk_gadget_sha256_compress(zk_cs *cs, zk_var state_io[32], const zk_var block[64]);
/* keccak256_round_function_entry_point (src/keccak256_round_function/mod.rs:672-794): the precompile FSM — request
 * queue pop, 6 conditional unaligned memory reads into the 192-byte ByteBuffer, padding, one Keccak-f per cycle,
 * conditional digest write; `limit` cycles.  Uses zk_circuit_keccak_configure.  Outer stream 474 words, loop 507. */
int zk_circuit_keccak256_round_function(zk_cs *cs, uint32_t limit);
/* demultiplex_storage_logs_enty_point (src/demux_log_queue/mod.rs:38-232): one LogQuery popped per cycle and pushed to
 * one of six queues by aux byte / shard / address.  Outer stream 73 words, loop stream 71 (see circuits/demux_log_queue.cpp). */
int zk_circuit_demux_log_queue_configure(zk_cs *cs);
int zk_circuit_demux_log_queue(zk_cs *cs, uint32_t limit);
/* sort_and_deduplicate_code_decommittments_entry_point (src/sort_decommittment_requests/mod.rs:40-222): two full-state
 * DecommitQuery queues under the grand-product argument, equal code hashes collapsed into the result queue.
 * Outer stream 151 words, loop stream 87 (see circuits/sort_decommits.cpp). */
int zk_circuit_sort_decommits_configure(zk_cs *cs);
int zk_circuit_sort_decommits(zk_cs *cs, uint32_t limit);
/* unpack_code_into_memory_entry_point (src/code_unpacker_sha256/mod.rs:33-142): decommit requests popped from a full-state
 * queue, bytecode words written to the memory queue two per cycle, SHA-256 over them compared with the versioned code hash.
 * Outer stream 125 words, loop stream 101 (see circuits/code_unpacker.cpp). */
int zk_circuit_code_unpacker_configure(zk_cs *cs);
int zk_circuit_code_unpacker(zk_cs *cs, uint32_t limit);
/* linear_hasher_entry_point (src/linear_hasher/mod.rs:35-212): Keccak-256 of the 88-byte serialisations of a queue of L2->L1
 * message logs; `limit` cycles, a multiple of 17 (the loop body is one 17-cycle / 11-block period of the reference's static buffer).
 * Outer stream 10 words, loop stream 818 words per period (see circuits/linear_hasher.cpp). */
int zk_circuit_linear_hasher_configure(zk_cs *cs);
int zk_circuit_linear_hasher(zk_cs *cs, uint32_t limit);
/* eip_4844_entry_point (src/eip_4844/mod.rs:107-260): Horner evaluation of the blob polynomial at the Fiat-Shamir point
 * over the non-native BLS12-381 scalar field + linear keccak256 of the blob + output hash; `n_chunks` 31-byte chunks
 * (the reference fixes 4096).  Outer stream 64 words (versioned_hash | linear_hash_output); loop stream
 * 217 + 136 + 31*ceil(n_chunks / n_blocks) words with n_blocks = 31*n_chunks / 136 + 1 (see circuits/eip4844.cpp). */
int zk_circuit_eip_4844_configure(zk_cs *cs);
int zk_circuit_eip_4844(zk_cs *cs, uint32_t n_chunks);
/* SHA-256 over n_blocks pre-padded 64-byte blocks (compression step of sha256_precompile_inner,
 * src/sha256_round_function/mod.rs:271-285) through 8-bit lookup tables.  Public inputs = the 32 digest bytes. */
int zk_circuit_sha256_configure(zk_cs *cs);
/* The reference's OWN configuration of the SHA circuits (src/code_unpacker_sha256/mod.rs:484-566): lookup width 4 x 8 repetitions and
 * exactly its five width-4 tables — Maj4Table, TriXor4Table, Ch4Table, Split4BitChunkTable<1>, Split4BitChunkTable<2> — and no 8-bit
 * table.  zk_circuit_sha256_blocks / zk_circuit_sha256_round_function / zk_circuit_code_unpacker recorded into a CS configured this
 * way use the 4-bit-chunk compression (csrc/circuits/sha256_gadget4.hpp) and range-check bytes through TriXor4; same input streams,
 * same public inputs as with zk_circuit_sha256_configure (the 8-bit engine tables), a different row count (zk_stats). */
int zk_circuit_sha256_configure_reference_tables(zk_cs *cs);
int zk_circuit_sha256_blocks(zk_cs *cs, uint32_t n_blocks);
/* sha256_round_function_entry_point (src/sha256_round_function/mod.rs:347-468): the precompile FSM — request
 * queue pop, 2 memory reads + 1 conditional write per cycle on the full-state memory queue, one compression
 * per cycle; `limit` cycles.  Uses zk_circuit_sha256_configure.  Outer stream 87 words, loop stream 112. */
int zk_circuit_sha256_round_function(zk_cs *cs, uint32_t limit);

#ifdef __cplusplus
}
#endif
#endif
