/*
 * zkgl_witness.h — product-side witness packers: the reference's per-circuit witness structs as plain C structs, turned into the
 * input streams the recorded circuits consume.  The host fills the structs (or decodes them from the
 * bincode bytes the reference's witness generator emits), the packer writes one instance's words into the host staging arrays of a
 * batch, the arrays are copied to the device and bound with zk_cs_bind_inputs; the loop-carried words are left zero and filled on
 * the device by zk_cs_seed_carried_inputs / zk_cs_seed_stream.
 *
 * Stream layouts (words are u64, canonical Goldilocks values): outer scope words[w * batch + instance];
 * loop scope words[w * (batch * limit) + instance * limit + cycle].
 */
#ifndef ZKGL_WITNESS_H
#define ZKGL_WITNESS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- the host pool.  Every zk_pack_* entry writes ONE instance's words of the batch staging arrays and shares nothing with the other
 * instances: the reference resolves its witness closures on a worker pool (/root/reference/src/ram_permutation/mod.rs:553-556; closures
 * `Send + Sync`, src/base_structures/memory_query/mod.rs:236).  zk_parallel_for runs fn(ctx, job) for job = 0 .. n_jobs-1 on n_threads
 * host threads (0: every hardware thread; the calling thread works too) — fn packs instance `job` with whichever packer applies.  Returns
 * the code of the LOWEST failing job (its message, prefixed "job <j>: ", in zk_last_error; *first_failed_job = j, UINT32_MAX when none;
 * may be NULL); the other jobs still run.  (zk_pack_main_vm_witness_batch, include/zkgl_vm.h: the array form for main_vm.) */
typedef int (*zk_job_fn)(void *ctx, uint32_t job);
int zk_parallel_for(uint32_t n_jobs, uint32_t n_threads, zk_job_fn fn, void *ctx, uint32_t *first_failed_job);
int zk_host_threads(void); /* hardware threads of the host (what n_threads = 0 uses) */

/* MemoryQueryWitness, /root/reference/src/base_structures/memory_query/mod.rs:30-37 (value: UInt256 as 8 little-endian u32 limbs) */
typedef struct zk_memory_query_witness {
    uint32_t timestamp, memory_page, index;
    uint8_t rw_flag, is_ptr;
    uint32_t value[8];
} zk_memory_query_witness;

/* QueueStateWitness<F, FULL_SPONGE_QUEUE_STATE_WIDTH = 12>: head, tail.tail, tail.length ([EXT] boojum gadgets::queue) */
typedef struct zk_full_queue_state_witness {
    uint64_t head[12];
    uint64_t tail[12];
    uint32_t length;
} zk_full_queue_state_witness;

/* RamPermutationFSMInputOutput witness, /root/reference/src/ram_permutation/input.rs:53-63 */
typedef struct zk_ram_fsm_witness {
    uint64_t lhs_accumulator[2], rhs_accumulator[2];
    zk_full_queue_state_witness current_unsorted_queue_state, current_sorted_queue_state;
    uint32_t previous_sorting_key[3];
    uint32_t previous_full_key[2];
    uint32_t previous_value[8];
    uint8_t previous_is_ptr;
    uint32_t num_nondeterministic_writes;
} zk_ram_fsm_witness;

/* RamPermutationCircuitInstanceWitness, /root/reference/src/ram_permutation/input.rs:99-117: closed_form_input (start_flag,
 * completion_flag, observable_input = RamPermutationInputData :28-32, observable_output = (), hidden_fsm_input, hidden_fsm_output,
 * /root/reference/src/fsm_input_output/mod.rs:42-47), then the two queue witnesses: the elements in pop order (the previous-tail
 * half of every (element, tail) pair is not consumed by the circuit: the pops re-derive the heads) */
typedef struct zk_ram_permutation_witness {
    uint8_t start_flag, completion_flag;
    zk_full_queue_state_witness unsorted_queue_initial_state, sorted_queue_initial_state;
    uint32_t non_deterministic_bootloader_memory_snapshot_length;
    zk_ram_fsm_witness hidden_fsm_input, hidden_fsm_output;
    const zk_memory_query_witness *unsorted_queue_witness; uint32_t n_unsorted;
    const zk_memory_query_witness *sorted_queue_witness; uint32_t n_sorted;
    /* the second member of every (element, tail) pair of the two queue witnesses: the queue state BEFORE that element was pushed
     * (src/ram_permutation/input.rs:103-116) = the head the circuit holds before it pops it.  Optional (NULL: absent).  When both are
     * given, zk_pack_ram_witness writes the queue heads of every cycle (24 of the 46 loop-carried words) from them — no hashing —
     * and the host declares those words given (zk_cs_set_seed_given, zk_ram_head_words): seeding then has no chain left and runs as
     * scans, lane = (instance, cycle). */
    const uint64_t (*unsorted_previous_tails)[12];
    const uint64_t (*sorted_previous_tails)[12];
} zk_ram_permutation_witness;

#define ZK_RAM_OUTER_WORDS 121
#define ZK_RAM_LOOP_WORDS 72
/* One instance of ram_permutation_entry_point (/root/reference/src/ram_permutation/mod.rs:31-382) into the batch's host staging
 * arrays: outer_words[ZK_RAM_OUTER_WORDS][batch], loop_words[ZK_RAM_LOOP_WORDS][batch * limit].  The 46 loop-carried words of every
 * cycle are zeroed (device seeding fills them); cycles past the queue length get the zero item, as the reference's
 * `pop_front().unwrap_or_default()` does.  ZK_ERR_INVALID: more elements than `limit`, lengths of the two queues differ. */
int zk_pack_ram_witness(const zk_ram_permutation_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                        uint64_t *outer_words, uint64_t *loop_words);

/* bincode 1.x (little-endian, fixed-width integers, u64 sequence lengths) of RamPermutationCircuitInstanceWitness in the field
 * order above.  [EXT] parts — not visible in /root/reference, restated from the crates' public behaviour and marked "unpinned" in
 * a field element is its canonical u64; QueueStateWitness = head[12], tail[12], length u32; a queue witness is
 * `elements: VecDeque<(MemoryQueryWitness, [F; 12])>`; U256 (ethereum-types with impl-serde) is a string: u64 length, then "0x" and
 * the hex digits without leading zeros ("0x0" for zero).
 * The element arrays are caller-owned (capacity in elements); *consumed = bytes read.  ZK_ERR_INVALID on truncated / malformed
 * input, ZK_ERR_CAPACITY when a queue holds more elements than its buffer. */
int zk_decode_ram_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_ram_permutation_witness *out,
                                  zk_memory_query_witness *unsorted_buf, uint32_t unsorted_cap,
                                  zk_memory_query_witness *sorted_buf, uint32_t sorted_cap, size_t *consumed);
/* The same, keeping the previous tails: unsorted_tails / sorted_tails are caller-owned [cap][12] arrays (NULL: dropped as above). */
int zk_decode_ram_witness_bincode_tails(const uint8_t *bytes, size_t n_bytes, zk_ram_permutation_witness *out,
                                        zk_memory_query_witness *unsorted_buf, uint32_t unsorted_cap,
                                        zk_memory_query_witness *sorted_buf, uint32_t sorted_cap,
                                        uint64_t (*unsorted_tails)[12], uint64_t (*sorted_tails)[12], size_t *consumed);
/* the 24 loop-stream words of ram_permutation that hold the two queue heads (what the packer fills from the previous tails) */
#define ZK_RAM_HEAD_WORDS 24
void zk_ram_head_words(uint32_t words[ZK_RAM_HEAD_WORDS]);


/* ---- LogQuery witness, /root/reference/src/base_structures/log_query/mod.rs:23-35 (UInt160 address: 5 little-endian u32 limbs) */
typedef struct zk_log_query_witness {
    uint32_t address[5];
    uint32_t key[8], read_value[8], written_value[8];
    uint8_t aux_byte, rw_flag, rollback, is_service, shard_id;
    uint32_t tx_number_in_block, timestamp;
} zk_log_query_witness;
/* QueueStateWitness<F, QUEUE_STATE_WIDTH = 4>: head, tail.tail, tail.length */
typedef struct zk_queue_state_witness { uint64_t head[4]; uint64_t tail[4]; uint32_t length; } zk_queue_state_witness;

/* StorageDeduplicatorInstanceWitness, /root/reference/src/storage_validity_by_grand_product/input.rs:131-136 (FSM :37-52, input data :84-88) */
typedef struct zk_storage_fsm_witness {
    uint64_t lhs_accumulator[2], rhs_accumulator[2];
    zk_queue_state_witness current_unsorted_queue_state, current_intermediate_sorted_queue_state, current_final_sorted_queue_state;
    uint32_t cycle_idx;
    uint32_t previous_packed_key[13];
    uint32_t previous_key[8];
    uint32_t previous_address[5];
    uint32_t previous_timestamp;
    uint8_t this_cell_has_explicit_read_and_rollback_depth_zero;
    uint32_t this_cell_base_value[8], this_cell_current_value[8];
    uint32_t this_cell_current_depth;
} zk_storage_fsm_witness;
typedef struct zk_timestamped_log_record_witness { zk_log_query_witness record; uint32_t timestamp; } zk_timestamped_log_record_witness;
typedef struct zk_storage_validity_witness {
    uint8_t start_flag, completion_flag;
    uint8_t shard_id_to_process;
    zk_queue_state_witness unsorted_log_queue_state, intermediate_sorted_queue_state;
    zk_storage_fsm_witness hidden_fsm_input, hidden_fsm_output;
    const zk_log_query_witness *unsorted_queue_witness; uint32_t n_unsorted;
    const zk_timestamped_log_record_witness *intermediate_sorted_queue_witness; uint32_t n_sorted;
    /* Optional (NULL / 0: absent).  The previous tail of every queue element, as the CircuitQueueRawWitness elements carry them
     * (item, previous_tail): with BOTH present zk_pack_storage_witness also writes the integer carried state of every cycle
     * (everything but the four grand-product accumulators and the output queue's tail) and the device seeds without a chain
     * over the input queues.  output_tails[j] = tail of the final sorted queue after its (j + 1)-th push in this instance (known
     * to a host that simulates the queue, e.g. from the next circuit's input witness): with it the output chain is skipped too. */
    const uint64_t (*unsorted_previous_tails)[4];
    const uint64_t (*sorted_previous_tails)[4];
    const uint64_t (*output_tails)[4]; uint32_t n_output_tails;
} zk_storage_validity_witness;
#define ZK_STORAGE_OUTER_WORDS 97
#define ZK_STORAGE_LOOP_WORDS 140
/* sort_and_deduplicate_storage_access_entry_point (/root/reference/src/storage_validity_by_grand_product/mod.rs:166-506): outer_words
 * [97][batch], loop_words[140][batch * limit]; the 67 carried words of every cycle zeroed (device seeding), popped LogQuery (36 words)
 * and TimestampedStorageLogRecord (37 words) per cycle, zero items past the queue length */
int zk_pack_storage_witness(const zk_storage_validity_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                            uint64_t *outer_words, uint64_t *loop_words);
/* the loop-carried words (of the 67) a witness with previous tails makes the packer fill — pass them to zk_cs_set_seed_given;
 * returns their count (0 without previous tails: the device seeds everything through the recorded cone) */
uint32_t zk_storage_given_words(const zk_storage_validity_witness *w, uint32_t words[67]);

/* EventsDeduplicatorInstanceWitness, /root/reference/src/log_sorter/input.rs:101-106 (FSM :28-36, input data :57-60) */
typedef struct zk_log_sorter_fsm_witness {
    uint64_t lhs_accumulator[2], rhs_accumulator[2];
    zk_queue_state_witness initial_unsorted_queue_state, intermediate_sorted_queue_state, final_result_queue_state;
    uint32_t previous_key;
    zk_log_query_witness previous_item;
} zk_log_sorter_fsm_witness;
typedef struct zk_log_sorter_witness {
    uint8_t start_flag, completion_flag;
    zk_queue_state_witness initial_log_queue_state, intermediate_sorted_queue_state;
    zk_log_sorter_fsm_witness hidden_fsm_input, hidden_fsm_output;
    const zk_log_query_witness *initial_queue_witness; uint32_t n_initial;
    const zk_log_query_witness *intermediate_sorted_queue_witness; uint32_t n_sorted;
    /* optional, as in zk_storage_validity_witness; output_tails = the result queue's tail after each of its pushes */
    const uint64_t (*initial_previous_tails)[4];
    const uint64_t (*sorted_previous_tails)[4];
    const uint64_t (*output_tails)[4]; uint32_t n_output_tails;
} zk_log_sorter_witness;
#define ZK_LOG_SORTER_OUTER_WORDS 87
#define ZK_LOG_SORTER_LOOP_WORDS 129
/* sort_and_deduplicate_events_entry_point (/root/reference/src/log_sorter/mod.rs:34-441): outer_words[87][batch],
 * loop_words[129][batch * limit]; 57 carried words zeroed, two LogQuery items (36 words each) per cycle */
int zk_pack_log_sorter_witness(const zk_log_sorter_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                               uint64_t *outer_words, uint64_t *loop_words);
uint32_t zk_log_sorter_given_words(const zk_log_sorter_witness *w, uint32_t words[57]);

/* EIP4844CircuitInstanceWitness, /root/reference/src/eip_4844/input.rs:61-66: versioned_hash, linear_hash_output, data_chunks
 * (BlobChunkWitness = 31 bytes, :31-33); the closed-form input has no observable input and no FSM state to pack */
typedef struct zk_eip4844_witness {
    uint8_t versioned_hash[32];
    uint8_t linear_hash_output[32];
    const uint8_t *data_chunks;   /* n_chunks x 31 bytes, chunk after chunk */
    uint32_t n_chunks;
} zk_eip4844_witness;
#define ZK_EIP4844_OUTER_WORDS 64
/* loop-scope shape of zk_circuit_eip_4844(cs, n_chunks): iterations (= Keccak blocks of the blob) and words per iteration */
int zk_eip4844_stream_shape(uint32_t n_chunks, uint32_t *n_iterations, uint32_t *loop_words);
/* eip_4844_entry_point (/root/reference/src/eip_4844/mod.rs:107-260) recorded with zk_circuit_eip_4844(cs, n_chunks): outer_words
 * [64][batch] = versioned_hash | linear_hash_output bytes; loop_words[loop_words][batch * n_iterations]: 217 carried words zeroed (Keccak
 * state, opening limbs, iteration counter: device seeding), then the 136 blob bytes of the iteration's Keccak block and the 31-byte
 * chunks of its Horner steps (zero bytes past the blob: the circuit pads the last block itself) */
int zk_pack_eip4844_witness(const zk_eip4844_witness *w, uint32_t instance, uint32_t batch, uint64_t *outer_words, uint64_t *loop_words);
/* The same with the 217 carried words of every iteration written by the host (zk_eip4844_given_words -> zk_cs_set_seed_given: no device
 * seeding): the sponge state before each Keccak block of the blob, the 16 opening limbs before each iteration's Horner steps (BLS12-381
 * scalar field, z from keccak256(linear_hash_output | versioned_hash)), the iteration counter — the values the reference's own test
 * computes out of circuit (mod.rs:595-683). */
int zk_pack_eip4844_witness_full(const zk_eip4844_witness *w, uint32_t instance, uint32_t batch, uint64_t *outer_words, uint64_t *loop_words);
uint32_t zk_eip4844_given_words(uint32_t words[217]);

/* Sha256RoundFunctionCircuitInstanceWitness, /root/reference/src/sha256_round_function/input.rs:85-89 (FSM :24-32, :53-57; call params:
 * input_page, input_offset, output_page, output_offset, num_rounds) */
typedef struct zk_sha256_fsm_witness {
    uint8_t read_precompile_call, read_words_for_round, completed;
    uint32_t sha256_inner_state[8];
    uint32_t timestamp_to_use_for_read, timestamp_to_use_for_write;
    uint32_t input_page, input_offset, output_page, output_offset, num_rounds;
    zk_queue_state_witness log_queue_state;
    zk_full_queue_state_witness memory_queue_state;
} zk_sha256_fsm_witness;
typedef struct zk_sha256_round_function_witness {
    uint8_t start_flag, completion_flag;
    zk_queue_state_witness initial_log_queue_state;
    zk_full_queue_state_witness initial_memory_queue_state;
    zk_sha256_fsm_witness hidden_fsm_input, hidden_fsm_output;
    const zk_log_query_witness *requests_queue_witness; uint32_t n_requests;   /* in pop order */
    const uint32_t (*memory_reads_witness)[8]; uint32_t n_reads;               /* VecDeque<U256> in pop order, 8 LE u32 limbs each */
} zk_sha256_round_function_witness;
#define ZK_SHA256_OUTER_WORDS 87
#define ZK_SHA256_LOOP_WORDS 112
/* sha256_round_function_entry_point (/root/reference/src/sha256_round_function/mod.rs:88-468).  The reference pops requests and read
 * values lazily inside the cycle loop; the streams want them at the cycle that consumes them, so the packer walks the FSM's
 * SCHEDULE (flags, rounds left, queue length — no hashing): a request is placed at the cycle whose read_precompile_call is set,
 * two read values at every cycle with rounds left.  60 carried words per cycle zeroed (device seeding).  ZK_ERR_INVALID when the
 * witness runs out of requests / read values before the schedule does. */
int zk_pack_sha256_witness(const zk_sha256_round_function_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                           uint64_t *outer_words, uint64_t *loop_words);

/* The same with every carried word written by the host (nothing to seed: zk_sha256_given_words -> zk_cs_set_seed_given):
 * `request_previous_tails[n_requests][4]` = the head of the request queue before each pop — the previous_tail bincode carries beside
 * every element of the witness (input.rs:85-89); `memory_tails[k][12]` = the memory queue's tail after the k-th push of this instance
 * (two reads per round with rounds left, one digest write per finished call, in that order) — the previous tails of the RAM
 * permutation's unsorted queue witness.  Flags, call parameters, timestamps and the SHA-256 inner state (one native compression per
 * cycle) are walked on the host.  ZK_ERR_INVALID when there are fewer memory tails than pushes. */
int zk_pack_sha256_witness_tails(const zk_sha256_round_function_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                 uint64_t *outer_words, uint64_t *loop_words, const uint64_t *request_previous_tails,
                                 const uint64_t *memory_tails, uint32_t n_memory_tails);
uint32_t zk_sha256_given_words(uint32_t words[60]);

/* Keccak256RoundFunctionCircuitInstanceWitness, /root/reference/src/keccak256_round_function/input.rs (FSM :29-39, call params
 * mod.rs:48-55, ByteBuffer buffer/mod.rs:42-48: bytes[192] + filled) */
typedef struct zk_keccak_fsm_witness {
    uint8_t read_precompile_call, read_unaligned_words_for_round, padding_round, completed;
    uint8_t keccak_internal_state[5][5][8];   /* [i][j][k]: byte k of lane x = i, y = j */
    uint32_t timestamp_to_use_for_read, timestamp_to_use_for_write;
    uint32_t input_page, input_memory_byte_offset, input_memory_byte_length, output_page, output_word_offset;
    uint8_t needs_full_padding_round;
    uint8_t buffer_bytes[192];
    uint32_t buffer_filled;
    zk_queue_state_witness log_queue_state;
    zk_full_queue_state_witness memory_queue_state;
} zk_keccak_fsm_witness;
typedef struct zk_keccak_round_function_witness {
    uint8_t start_flag, completion_flag;
    zk_queue_state_witness initial_log_queue_state;
    zk_full_queue_state_witness initial_memory_queue_state;
    zk_keccak_fsm_witness hidden_fsm_input, hidden_fsm_output;
    const zk_log_query_witness *requests_queue_witness; uint32_t n_requests;
    const uint32_t (*memory_reads_witness)[8]; uint32_t n_reads;
} zk_keccak_round_function_witness;
#define ZK_KECCAK_OUTER_WORDS 474
#define ZK_KECCAK_LOOP_WORDS 507
/* keccak256_round_function_entry_point (/root/reference/src/keccak256_round_function/mod.rs:155-794): like the sha256 packer, the
 * FSM's schedule is walked — flags, bytes left, byte offset, ByteBuffer fill level (six conditional unaligned reads per cycle,
 * mod.rs:300-420; buffer/mod.rs:90-163) — to place every request and read value at its cycle; 423 carried words zeroed */
int zk_pack_keccak_witness(const zk_keccak_round_function_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                           uint64_t *outer_words, uint64_t *loop_words);
/* The same with every carried word written by the host (zk_keccak_given_words -> zk_cs_set_seed_given), as zk_pack_sha256_witness_tails:
 * `request_previous_tails[n_requests][4]`, `memory_tails[k][12]` = the memory queue's tail after the k-th push of this instance (up to six
 * reads, then the digest write of a finished call, per cycle).  The ByteBuffer and the sponge state (one native Keccak-f per cycle)
 * are walked on the host. */
int zk_pack_keccak_witness_tails(const zk_keccak_round_function_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                 uint64_t *outer_words, uint64_t *loop_words, const uint64_t *request_previous_tails,
                                 const uint64_t *memory_tails, uint32_t n_memory_tails);
uint32_t zk_keccak_given_words(uint32_t words[423]);

/* ---- LogDemuxerCircuitInstanceWitness, /root/reference/src/demux_log_queue/input.rs:118-121 (FSM :26-34, input data :57-59; output
 * queues in the FSM's field order: storage, events, l1 messages, keccak256, sha256, ecrecover) */
typedef struct zk_demux_fsm_witness {
    zk_queue_state_witness initial_log_queue_state;
    zk_queue_state_witness output_queue_states[6];
} zk_demux_fsm_witness;
typedef struct zk_demux_log_queue_witness {
    uint8_t start_flag, completion_flag;
    zk_queue_state_witness initial_log_queue_state;
    zk_demux_fsm_witness hidden_fsm_input, hidden_fsm_output;
    const zk_log_query_witness *initial_queue_witness; uint32_t n_initial;   /* in pop order */
} zk_demux_log_queue_witness;
#define ZK_DEMUX_OUTER_WORDS 73
#define ZK_DEMUX_LOOP_WORDS 71
/* demultiplex_storage_logs_enty_point (/root/reference/src/demux_log_queue/mod.rs:38-396): outer_words[73][batch],
 * loop_words[71][batch * limit]; 35 carried words zeroed (device seeding), one popped LogQuery (36 words) per cycle */
int zk_pack_demux_witness(const zk_demux_log_queue_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                          uint64_t *outer_words, uint64_t *loop_words);
/* The same with the queue states the reference's witnesses hold (round 4): input_previous_tails[n_initial][4] = the second member of
 * every (LogQuery, previous tail) element of initial_queue_witness (input.rs:118-121) — the head before that element is popped;
 * output_tails[n_initial][4] = the tail of the element's TARGET queue after its push (the previous tails of the next circuits' input
 * witnesses; ignored for an element no queue takes).  All 35 carried words of every cycle are then written by the packer
 * (zk_demux_given_words -> zk_cs_set_seed_given): seeding has nothing left to compute (round 3: a cone of 7 dependent permutations per
 * cycle, 86 ms for 1 187 cycles against a 1.7 ms step). */
int zk_pack_demux_witness_tails(const zk_demux_log_queue_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                uint64_t *outer_words, uint64_t *loop_words, const uint64_t *input_previous_tails, const uint64_t *output_tails);
uint32_t zk_demux_given_words(uint32_t words[35]);

/* ---- DecommitQuery witness, /root/reference/src/base_structures/decommit_query/mod.rs:22-29 */
typedef struct zk_decommit_query_witness {
    uint32_t code_hash[8];   /* U256, little-endian u32 limbs */
    uint32_t page;
    uint8_t is_first;
    uint32_t timestamp;
} zk_decommit_query_witness;
/* CodeDecommittmentsDeduplicatorInstanceWitness, /root/reference/src/sort_decommittment_requests/input.rs:110-124 (FSM :26-37, input
 * data :62-65) */
typedef struct zk_sort_decommits_fsm_witness {
    zk_full_queue_state_witness initial_queue_state, sorted_queue_state, final_queue_state;
    uint64_t lhs_accumulator[2], rhs_accumulator[2];
    uint32_t previous_packed_key[9];
    uint32_t first_encountered_timestamp;
    zk_decommit_query_witness previous_record;
} zk_sort_decommits_fsm_witness;
typedef struct zk_sort_decommits_witness {
    uint8_t start_flag, completion_flag;
    zk_full_queue_state_witness initial_queue_state, sorted_queue_initial_state;
    zk_sort_decommits_fsm_witness hidden_fsm_input, hidden_fsm_output;
    const zk_decommit_query_witness *initial_queue_witness; uint32_t n_initial;
    const zk_decommit_query_witness *sorted_queue_witness; uint32_t n_sorted;
} zk_sort_decommits_witness;
#define ZK_SORT_DECOMMITS_OUTER_WORDS 151
#define ZK_SORT_DECOMMITS_LOOP_WORDS 87
/* sort_and_deduplicate_code_decommittments_entry_point (/root/reference/src/sort_decommittment_requests/mod.rs:40-372):
 * outer_words[151][batch], loop_words[87][batch * limit]; 65 carried words zeroed, one DecommitQuery (11 words) of each queue per cycle */
int zk_pack_sort_decommits_witness(const zk_sort_decommits_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                   uint64_t *outer_words, uint64_t *loop_words);

/* The same with the integer state and the queue states of every cycle written by the host (61 of the 65 carried words:
 * zk_sort_decommits_given_words -> zk_cs_set_seed_given; the four grand-product words are a device scan, k_decommit_seed):
 * `initial_previous_tails` / `sorted_previous_tails` [n][12] = the 12-word head of each input queue before each pop — the
 * previous_tail the reference's FullStateCircuitQueueRawWitness holds beside every element (input.rs:110-124); `result_tails[k][12]` =
 * the result queue's tail after the k-th push inside this instance's loop — the previous tails of the decommitter's requests queue
 * witness.  ZK_ERR_INVALID when there are fewer result tails than pushes. */
int zk_pack_sort_decommits_witness_tails(const zk_sort_decommits_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                         uint64_t *outer_words, uint64_t *loop_words, const uint64_t *initial_previous_tails,
                                         const uint64_t *sorted_previous_tails, const uint64_t *result_tails, uint32_t n_result_tails);
uint32_t zk_sort_decommits_given_words(uint32_t words[65]);

/* CodeDecommitterCircuitInstanceWitness, /root/reference/src/code_unpacker_sha256/input.rs:134-140 (internal FSM :23-34, FSM :61-65,
 * input data :80-83) */
typedef struct zk_code_unpacker_fsm_witness {
    uint32_t sha256_inner_state[8];
    uint32_t hash_to_compare_against[8];
    uint32_t current_index, current_page, timestamp;
    uint16_t num_rounds_left;
    uint32_t length_in_bits;
    uint8_t state_get_from_queue, state_decommit, finished;
    zk_full_queue_state_witness decommittment_requests_queue_state, memory_queue_state;
} zk_code_unpacker_fsm_witness;
typedef struct zk_code_unpacker_witness {
    uint8_t start_flag, completion_flag;
    zk_full_queue_state_witness memory_queue_initial_state, sorted_requests_queue_initial_state;
    zk_code_unpacker_fsm_witness hidden_fsm_input, hidden_fsm_output;
    const zk_decommit_query_witness *sorted_requests_queue_witness; uint32_t n_requests;   /* in pop order */
    const uint32_t (*code_words)[8]; uint32_t n_code_words;   /* Vec<Vec<U256>> flattened in consumption order, 8 LE u32 limbs each */
} zk_code_unpacker_witness;
#define ZK_CODE_UNPACKER_OUTER_WORDS 125
#define ZK_CODE_UNPACKER_LOOP_WORDS 101
/* unpack_code_into_memory_entry_point (/root/reference/src/code_unpacker_sha256/mod.rs:33-442).  Requests and code words are placed at
 * the cycles that consume them by walking the FSM's schedule (state_get_from_queue / state_decommit / rounds left: mod.rs:167-255,
 * :402-430; no hashing): a request at every cycle that pops, one code word at every decommit cycle and a second one unless it is the
 * last round.  Outer stream order: start_flag, requests queue state, memory queue state (the circuit's allocation order), FSM;
 * 74 carried words zeroed.  ZK_ERR_INVALID when the witness runs out of requests before the schedule does. */
int zk_pack_code_unpacker_witness(const zk_code_unpacker_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                  uint64_t *outer_words, uint64_t *loop_words);

/* The same with every carried word written by the host (nothing to seed: zk_code_unpacker_given_words -> zk_cs_set_seed_given):
 * `request_previous_tails[n_requests][12]` = the 12-word head of the requests queue before each pop — the previous_tail the reference's
 * FullStateCircuitQueueRawWitness holds beside every element (input.rs:134-140); `memory_tails[n_code_words][12]` = the memory queue's
 * tail after each code word of this instance was pushed — the previous tails of the RAM permutation's unsorted queue witness.  The
 * FSM scalars and the SHA-256 inner state are walked natively (one compression per cycle).  ZK_ERR_INVALID when the code witness
 * runs out before the schedule does (a pushed word without a tail). */
int zk_pack_code_unpacker_witness_tails(const zk_code_unpacker_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                        uint64_t *outer_words, uint64_t *loop_words, const uint64_t *request_previous_tails,
                                        const uint64_t *memory_tails);
uint32_t zk_code_unpacker_given_words(uint32_t words[74]);

/* LinearHasherCircuitInstanceWitness, /root/reference/src/linear_hasher/input.rs:71-80 (input data :27-29; no FSM state) */
typedef struct zk_linear_hasher_witness {
    uint8_t start_flag, completion_flag;
    zk_queue_state_witness queue_state;
    const zk_log_query_witness *queue_witness; uint32_t n_queue;
} zk_linear_hasher_witness;
#define ZK_LINEAR_HASHER_OUTER_WORDS 10
#define ZK_LINEAR_HASHER_LOOP_WORDS 818
#define ZK_LINEAR_HASHER_PERIOD 17
/* linear_hasher_entry_point (/root/reference/src/linear_hasher/mod.rs:35-212): one loop iteration = one period of 17 pops (the
 * reference's statically unrolled byte buffer repeats every 17 cycles = 11 blocks); `limit` (cycles) must be a multiple of 17.
 * outer_words[10][batch], loop_words[818][batch * limit / 17]: 206 carried words zeroed, then 17 LogQuery items (36 words each) */
int zk_pack_linear_hasher_witness(const zk_linear_hasher_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                  uint64_t *outer_words, uint64_t *loop_words);
/* The same with the 206 carried words of every period written by the host (zk_linear_hasher_given_words): `queue_previous_tails[n_queue][4]`
 * = the previous tail bincode carries beside every queue element (the head before its pop); the sponge is walked on the host. */
int zk_pack_linear_hasher_witness_tails(const zk_linear_hasher_witness *w, uint32_t limit, uint32_t instance, uint32_t batch,
                                        uint64_t *outer_words, uint64_t *loop_words, const uint64_t *queue_previous_tails);
uint32_t zk_linear_hasher_given_words(uint32_t words[206]);

/* ---- bincode decoders for the witnesses built from LogQuery queues (same conventions and the same [EXT] caveats as
 * zk_decode_ram_witness_bincode: "parity unpinned" until reference-produced bytes exist).  ClosedFormInputWitness = start_flag,
 * completion_flag, observable_input, observable_output, hidden_fsm_input, hidden_fsm_output (src/fsm_input_output/mod.rs:32-47; the
 * observable output is read and dropped: the packers do not need it); a QueueStateWitness<F, 4> = head[4], tail[4], length u32; a queue
 * witness = `elements: VecDeque<(ItemWitness, [F; 4])>`; LogQueryWitness in its field order with the address (ethereum-types H160 with
 * impl-serde) as the string "0x" + 40 hex digits and U256 as "0x" + hex digits without leading zeros.  Element arrays are caller-owned. */
int zk_decode_storage_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_storage_validity_witness *out,
                                      zk_log_query_witness *unsorted_buf, uint32_t unsorted_cap,
                                      zk_timestamped_log_record_witness *sorted_buf, uint32_t sorted_cap, size_t *consumed);
int zk_decode_log_sorter_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_log_sorter_witness *out,
                                         zk_log_query_witness *initial_buf, uint32_t initial_cap,
                                         zk_log_query_witness *sorted_buf, uint32_t sorted_cap, size_t *consumed);
/* the same two decoders keeping the previous tails (caller-owned [cap][4] arrays) */
int zk_decode_storage_witness_bincode_tails(const uint8_t *bytes, size_t n_bytes, zk_storage_validity_witness *out,
                                            zk_log_query_witness *unsorted_buf, uint32_t unsorted_cap,
                                            zk_timestamped_log_record_witness *sorted_buf, uint32_t sorted_cap,
                                            uint64_t (*unsorted_tails)[4], uint64_t (*sorted_tails)[4], size_t *consumed);
int zk_decode_log_sorter_witness_bincode_tails(const uint8_t *bytes, size_t n_bytes, zk_log_sorter_witness *out,
                                               zk_log_query_witness *initial_buf, uint32_t initial_cap,
                                               zk_log_query_witness *sorted_buf, uint32_t sorted_cap,
                                               uint64_t (*initial_tails)[4], uint64_t (*sorted_tails)[4], size_t *consumed);
int zk_decode_demux_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_demux_log_queue_witness *out,
                                    zk_log_query_witness *initial_buf, uint32_t initial_cap, size_t *consumed);
/* the same, keeping the previous tail bincode carries beside every queue element ([initial_cap][4], caller's buffer): the input of
 * zk_pack_demux_witness_tails */
int zk_decode_demux_witness_bincode_tails(const uint8_t *bytes, size_t n_bytes, zk_demux_log_queue_witness *out,
                                          zk_log_query_witness *initial_buf, uint32_t initial_cap, uint64_t (*initial_tails)[4],
                                          size_t *consumed);
int zk_decode_linear_hasher_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_linear_hasher_witness *out,
                                            zk_log_query_witness *queue_buf, uint32_t queue_cap, size_t *consumed);

/* the precompile and decommitment witnesses: PrecompileFunctionInputData = log queue state (4-wide), memory queue state (12-wide);
 * the observable output (final memory queue state) is read and dropped; `memory_reads_witness: VecDeque<U256>` = u64 count + U256
 * strings; ByteBuffer = bytes[192] + filled (u8); a full-state queue witness = `elements: VecDeque<(DecommitQueryWitness, [F; 12])>`;
 * `code_words: Vec<Vec<U256>>` is flattened into the caller's word buffer in order */
int zk_decode_sha256_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_sha256_round_function_witness *out,
                                     zk_log_query_witness *requests_buf, uint32_t requests_cap,
                                     uint32_t (*reads_buf)[8], uint32_t reads_cap, size_t *consumed);
int zk_decode_keccak_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_keccak_round_function_witness *out,
                                     zk_log_query_witness *requests_buf, uint32_t requests_cap,
                                     uint32_t (*reads_buf)[8], uint32_t reads_cap, size_t *consumed);
/* the same two, keeping the 4-word previous tail bincode carries beside every request ([requests_cap][4], caller's buffer): an input of
 * zk_pack_sha256_witness_tails / zk_pack_keccak_witness_tails */
int zk_decode_sha256_witness_bincode_tails(const uint8_t *bytes, size_t n_bytes, zk_sha256_round_function_witness *out,
                                           zk_log_query_witness *requests_buf, uint32_t requests_cap, uint32_t (*reads_buf)[8],
                                           uint32_t reads_cap, uint64_t (*request_tails)[4], size_t *consumed);
int zk_decode_keccak_witness_bincode_tails(const uint8_t *bytes, size_t n_bytes, zk_keccak_round_function_witness *out,
                                           zk_log_query_witness *requests_buf, uint32_t requests_cap, uint32_t (*reads_buf)[8],
                                           uint32_t reads_cap, uint64_t (*request_tails)[4], size_t *consumed);
int zk_decode_sort_decommits_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_sort_decommits_witness *out,
                                             zk_decommit_query_witness *initial_buf, uint32_t initial_cap,
                                             zk_decommit_query_witness *sorted_buf, uint32_t sorted_cap, size_t *consumed);
int zk_decode_code_unpacker_witness_bincode(const uint8_t *bytes, size_t n_bytes, zk_code_unpacker_witness *out,
                                            zk_decommit_query_witness *requests_buf, uint32_t requests_cap,
                                            uint32_t (*words_buf)[8], uint32_t words_cap, size_t *consumed);
/* the same two, keeping the 12-word previous tail bincode carries beside every FullStateCircuitQueueRawWitness element (caller's
 * buffers, one row per element): the inputs of zk_pack_sort_decommits_witness_tails / zk_pack_code_unpacker_witness_tails */
int zk_decode_sort_decommits_witness_bincode_tails(const uint8_t *bytes, size_t n_bytes, zk_sort_decommits_witness *out,
                                                   zk_decommit_query_witness *initial_buf, uint32_t initial_cap,
                                                   zk_decommit_query_witness *sorted_buf, uint32_t sorted_cap,
                                                   uint64_t (*initial_tails)[12], uint64_t (*sorted_tails)[12], size_t *consumed);
int zk_decode_code_unpacker_witness_bincode_tails(const uint8_t *bytes, size_t n_bytes, zk_code_unpacker_witness *out,
                                                  zk_decommit_query_witness *requests_buf, uint32_t requests_cap,
                                                  uint32_t (*words_buf)[8], uint32_t words_cap, uint64_t (*request_tails)[12],
                                                  size_t *consumed);

#ifdef __cplusplus
}
#endif
#endif
