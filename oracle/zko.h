/*
 * oracle/zko.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic that era-zkevm_circuits drives through
 * boojum (Goldilocks field, Poseidon2 width-12 sponge, queue/commitment rules,
 * gate relations, witness-IR interpreter).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (era-zkevm_circuits_amd/) never includes, links or calls it.
 *
 * PARITY STATUS ("what pins this oracle"):
 *   - boojum (git dep, branch main, Cargo.toml:19 of the reference) is NOT in
 *     /root/reference, so the field/hash layer is restated from its published
 *     algorithm.  Pinned pieces:
 *       * Poseidon round constants: re-derived from first principles
 *         (ChaCha8Rng::seed_from_u64(0) -> gen_range(0..p), the procedure
 *         plonky2 used for the Poseidon-Goldilocks constants boojum re-uses);
 *         the derivation reproduces the 14 independently known published
 *         values (tests/golden/poseidon_rc_known.json).
 *       * absorb-with-replacement / capacity carry-over: pinned by the
 *         reference itself (src/utils.rs:41-44, src/main_vm/utils.rs:197-210).
 *       * Keccak-f / SHA-256: pinned against hashlib KATs.
 *   - NOT pinned ("parity unpinned"): Poseidon2 matrix layout
 *     (M4 block / inner-diagonal shifts are recollection of boojum), the
 *     capacity slot of apply_length_specialization, boojum's gate->column
 *     placement.  No reference test holds a Poseidon2 output value
 *     (last bullet).
 */
#ifndef ZKO_H
#define ZKO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKO_P 0xFFFFFFFF00000001ULL
#define ZKO_EPS 0xFFFFFFFFULL

/* ---- Goldilocks (boojum::field::goldilocks [EXT]) ---- */
uint64_t zko_gl_add(uint64_t a, uint64_t b);
uint64_t zko_gl_sub(uint64_t a, uint64_t b);
uint64_t zko_gl_mul(uint64_t a, uint64_t b);
uint64_t zko_gl_mul_slow(uint64_t a, uint64_t b); /* (a * b) % p by 128-bit division: cross-check only */
uint64_t zko_gl_pow(uint64_t a, uint64_t e);
uint64_t zko_gl_inv(uint64_t a); /* 0 -> 0 */
uint64_t zko_gl_reduce(uint64_t a); /* canonical representative */

/* column ops over n elements: dst[i] = q*a[i]*b[i] + l*c[i] */
void zko_gl_fma_cols(uint64_t *dst, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                     uint64_t q, uint64_t l, size_t n);

/* ---- Poseidon2 width 12 (boojum::implementations::poseidon2 [EXT]) ---- */
const uint64_t *zko_poseidon_round_constants(void); /* 360 values, derived at first call */
void zko_poseidon2_permute(uint64_t state[12]);
/* batch over n states, AoS layout states[i*12 + j] */
void zko_poseidon2_permute_batch(uint64_t *states, size_t n);
/* external (M_E) and inner (M_I) linear layers, exposed for gate tests */
void zko_poseidon2_mds_external(uint64_t state[12]);
void zko_poseidon2_mds_inner(uint64_t state[12]);

/* ---- sponge rules (reference src/fsm_input_output/mod.rs:296-326, src/utils.rs:12-78) ---- */
/* commit_encoding: empty state, length specialisation, zero pad to x8, absorb with
 * replacement keeping capacity, permute; output first 4 elements. */
void zko_commit_encoding(const uint64_t *input, size_t len, uint64_t out[4]);
/* produce_fs_challenges: input = tailA||lenA||tailB||lenB (len elements), writes
 * reps*nchal challenges, slot 0 of each repetition is the constant 1. */
void zko_fs_challenges(const uint64_t *fs_input, size_t len, uint64_t *out, size_t reps, size_t nchal);
/* full-state queue push (src/main_vm/utils.rs:194-213): tail' = P([enc0..7, tail8..11]) */
void zko_queue_full_push(uint64_t tail[12], const uint64_t enc[8]);
/* 4-wide-tail queue push of a 20-element encoding (src/main_vm/opcodes/log.rs:508-585) */
void zko_queue_tail4_push20(uint64_t tail[4], const uint64_t enc[20]);

/* ---- encodings ---- */
/* MemoryQuery::encode (src/base_structures/memory_query/mod.rs:103-221).
 * q = {timestamp, memory_page, index, rw_flag, is_ptr, value limbs 0..7 (u32 LE)} */
void zko_memory_query_encode(const uint64_t q[13], uint64_t enc[8]);

/* ExecutionContextRecord::encode (src/base_structures/vm_state/saved_context.rs:111-266).
 * rec = flattened record in declaration order (saved_context.rs:36-66): this[5], caller[5], code_address[5],
 * code_page, base_page, heap_upper_bound, aux_heap_upper_bound, reverted_queue_head[4], reverted_queue_tail[4],
 * reverted_queue_segment_len, pc, sp, exception_handler_loc, ergs_remaining, is_static_execution,
 * is_kernel_mode, this_shard_id, caller_shard_id, code_shard_id, context_u128_value_composite[4], is_local_call */
void zko_execution_context_encode(const uint64_t rec[42], uint64_t enc[32]);

/* ---- grand product (src/utils.rs:81-137) ---- */
/* contributions: contrib[i] = ch[enc_len] + sum_j enc[i*enc_len+j]*ch[j];
 * running accumulators: acc_out[i] = value of the accumulator AFTER item i,
 * acc_{i} = flag[i] ? acc_{i-1}*contrib[i] : acc_{i-1}, acc_{-1} = init. */
void zko_grand_product(const uint64_t *enc, const uint8_t *flags, const uint64_t *challenges,
                       size_t enc_len, size_t n, uint64_t init, uint64_t *acc_out);

/* ---- Keccak-f[1600] / SHA-256 compression (pinned by hashlib) ---- */
void zko_keccak_f1600(uint64_t st[25]);
void zko_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]);
void zko_sha256_compress(uint32_t state[8], const uint8_t block[64]);

/* ---- K11: NTT / coset LDE (zko_ntt.c; transform defined in include/zkgl.h, boojum's is [EXT]) ---- */
uint64_t zko_two_adic_root(unsigned log_n);
void zko_ntt_naive(const uint64_t *a, uint64_t *out, unsigned log_n, uint64_t shift);
void zko_ntt(uint64_t *a, unsigned log_n, int inverse, uint64_t shift);
void zko_ntt_batch(uint64_t *a, unsigned log_n, size_t n_polys, size_t stride, int inverse, uint64_t shift);
void zko_lde(const uint64_t *coeffs, uint64_t *out, unsigned log_n, unsigned log_blowup, uint64_t shift);

#ifdef __cplusplus
}
#endif
#endif
