/*
 * zkgl_vm.h — everything the main_vm circuit takes from the crate `zkevm_opcode_defs` (absent from /root/reference: a git
 * dependency, Cargo.toml:18, branch v1.4.1), gathered into ONE data blob that the host passes to
 * zk_circuit_main_vm_configure().  The circuit STRUCTURE is restated from /root/reference/src/main_vm; the blob holds only
 * data: the 2^11-row opcode table (OPCODES_PROPS_INTEGER_BITMASKS / OPCODES_PRICES, src/tables/opcodes_decoding.rs:14-38), the
 * bit positions `OpcodeBitmask::from_full_mask` slices (src/main_vm/opcode_bitmask.rs:60-128), the sub-variant / flag indices the
 * opcode closures ask for by name, and `system_params::*`.
 *
 * [EXT] zk_opcode_defs_default() fills a blob of the reference's SHAPE with this build's own enumeration of the variants and
 * recollected constants; a host that links the real crate fills the same struct from it.
 */
#ifndef ZKGL_VM_H
#define ZKGL_VM_H
#include <stdint.h>
#include "zkgl.h"
#include "zkgl_witness.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_VM_OPCODE_TABLE_ROWS 2048 /* 1 << OPCODES_TABLE_WIDTH */
#define ZK_VM_REGISTERS 15           /* REGISTERS_COUNT, src/base_structures/vm_state/mod.rs:29 */

/* Opcode::variant_idx(): index of the family's one-hot boolean in opcode_type_booleans (16 = OPCODE_TYPE_BITS) */
enum zk_vm_family {
    ZK_VMF_INVALID = 0, ZK_VMF_NOP, ZK_VMF_ADD, ZK_VMF_SUB, ZK_VMF_MUL, ZK_VMF_DIV, ZK_VMF_JUMP, ZK_VMF_CONTEXT, ZK_VMF_SHIFT,
    ZK_VMF_BINOP, ZK_VMF_PTR, ZK_VMF_NEAR_CALL, ZK_VMF_LOG, ZK_VMF_FAR_CALL, ZK_VMF_RET, ZK_VMF_UMA, ZK_VMF__COUNT
};
/* the sub-variants the circuit asks for through boolean_for_variant (materialize_subvariant_idx) */
enum zk_vm_variant {
    ZK_VMV_SHIFT_SHL = 0, ZK_VMV_SHIFT_SHR, ZK_VMV_SHIFT_ROL, ZK_VMV_SHIFT_ROR,
    ZK_VMV_BINOP_XOR, ZK_VMV_BINOP_AND, ZK_VMV_BINOP_OR,
    ZK_VMV_PTR_ADD, ZK_VMV_PTR_SUB, ZK_VMV_PTR_PACK, ZK_VMV_PTR_SHRINK,
    ZK_VMV_CTX_THIS, ZK_VMV_CTX_CALLER, ZK_VMV_CTX_CODE_ADDRESS, ZK_VMV_CTX_META, ZK_VMV_CTX_ERGS_LEFT, ZK_VMV_CTX_SP,
    ZK_VMV_CTX_GET_CONTEXT_U128, ZK_VMV_CTX_SET_CONTEXT_U128, ZK_VMV_CTX_SET_ERGS_PER_PUBDATA, ZK_VMV_CTX_INC_TX_NUMBER,
    ZK_VMV_LOG_STORAGE_READ, ZK_VMV_LOG_STORAGE_WRITE, ZK_VMV_LOG_TO_L1, ZK_VMV_LOG_EVENT, ZK_VMV_LOG_PRECOMPILE_CALL,
    ZK_VMV_FAR_NORMAL, ZK_VMV_FAR_DELEGATE, ZK_VMV_FAR_MIMIC,
    ZK_VMV_RET_OK, ZK_VMV_RET_REVERT, ZK_VMV_RET_PANIC,
    ZK_VMV_UMA_HEAP_READ, ZK_VMV_UMA_HEAP_WRITE, ZK_VMV_UMA_AUX_HEAP_READ, ZK_VMV_UMA_AUX_HEAP_WRITE, ZK_VMV_UMA_FAT_PTR_READ,
    ZK_VMV__COUNT
};
/* flag_booleans indices by their reference names */
enum zk_vm_flag {
    ZK_VMFL_SET_FLAGS = 0,   /* SET_FLAGS_FLAG_IDX */
    ZK_VMFL_SWAP_ARITH,      /* SWAP_OPERANDS_FLAG_IDX_FOR_ARITH_OPCODES */
    ZK_VMFL_SWAP_PTR,        /* SWAP_OPERANDS_FLAG_IDX_FOR_PTR_OPCODE */
    ZK_VMFL_FIRST_MESSAGE,   /* FIRST_MESSAGE_FLAG_IDX */
    ZK_VMFL_UMA_INCREMENT,   /* UMA_INCREMENT_FLAG_IDX */
    ZK_VMFL_FAR_CALL_STATIC, /* FAR_CALL_STATIC_FLAG_IDX */
    ZK_VMFL_FAR_CALL_SHARD,  /* FAR_CALL_SHARD_FLAG_IDX */
    ZK_VMFL_RET_TO_LABEL,    /* ret::RET_TO_LABEL_BIT_IDX */
    ZK_VMFL__COUNT
};
/* ImmMemHandlerFlags::variant_index() */
enum zk_vm_operand_mode {
    ZK_VMM_REG_ONLY = 0, ZK_VMM_STACK_PUSH_POP, ZK_VMM_STACK_OFFSET, ZK_VMM_ABSOLUTE_STACK, ZK_VMM_IMM16, ZK_VMM_CODE_PAGE, ZK_VMM__COUNT
};
/* Condition, in the order of the match in src/tables/conditional.rs:35-44 */
enum zk_vm_condition { ZK_VMC_ALWAYS = 0, ZK_VMC_LT, ZK_VMC_EQ, ZK_VMC_GT, ZK_VMC_GE, ZK_VMC_LE, ZK_VMC_NE, ZK_VMC_GT_OR_LT, ZK_VMC__COUNT };
/* system_params / constants read by the circuit */
enum zk_vm_param {
    ZK_VMP_VM_MAX_STACK_DEPTH = 0, ZK_VMP_NEW_FRAME_MEMORY_STIPEND, ZK_VMP_NEW_MEMORY_PAGES_PER_FAR_CALL, ZK_VMP_UNMAPPED_PAGE,
    ZK_VMP_BOOTLOADER_BASE_PAGE, ZK_VMP_BOOTLOADER_CODE_PAGE, ZK_VMP_BOOTLOADER_CALLDATA_PAGE, ZK_VMP_STARTING_BASE_PAGE,
    ZK_VMP_STARTING_TIMESTAMP, ZK_VMP_INITIAL_FRAME_FORMAL_EH_LOCATION, ZK_VMP_VM_INITIAL_FRAME_ERGS,
    ZK_VMP_BOOTLOADER_FORMAL_ADDRESS_LOW, ZK_VMP_BOOTLOADER_MAX_MEMORY, ZK_VMP_DEPLOYER_SYSTEM_CONTRACT_ADDRESS_LOW,
    ZK_VMP_ERGS_PER_CODE_WORD_DECOMMITTMENT, ZK_VMP_INITIAL_STORAGE_WRITE_PUBDATA_BYTES, ZK_VMP_L1_MESSAGE_PUBDATA_BYTES,
    ZK_VMP_STORAGE_AUX_BYTE, ZK_VMP_EVENT_AUX_BYTE, ZK_VMP_L1_MESSAGE_AUX_BYTE, ZK_VMP_PRECOMPILE_AUX_BYTE,
    ZK_VMP_CODE_HASH_VERSION_BYTE, ZK_VMP_CODE_YET_CONSTRUCTED_MARKER, ZK_VMP_CODE_AT_REST_MARKER,
    ZK_VMP_FAR_CALL_FORWARDING_MODE_BYTE_IDX, ZK_VMP_FAR_CALL_SHARD_ID_BYTE_IDX, ZK_VMP_FAR_CALL_CONSTRUCTOR_CALL_BYTE_IDX,
    ZK_VMP_FAR_CALL_SYSTEM_CALL_BYTE_IDX, ZK_VMP_FORWARD_USE_HEAP, ZK_VMP_FORWARD_FAT_POINTER, ZK_VMP_FORWARD_USE_AUX_HEAP,
    ZK_VMP_CALL_IMPLICIT_PARAMETER_REG_IDX, ZK_VMP_CALL_SYSTEM_ABI_REGISTERS_BEGIN, ZK_VMP_CALL_SYSTEM_ABI_REGISTERS_END,
    ZK_VMP_CALL_RESERVED_RANGE_BEGIN, ZK_VMP_CALL_RESERVED_RANGE_END,
    ZK_VMP__COUNT
};

typedef struct zk_opcode_defs {
    uint32_t version;                              /* SUPPORTED_ISA_VERSION, src/main_vm/opcode_bitmask.rs:16 */
    uint32_t n_valid;                              /* rows [0, n_valid) hold defined variants, the rest decode as Invalid */
    uint64_t props[ZK_VM_OPCODE_TABLE_ROWS];       /* OPCODES_PROPS_INTEGER_BITMASKS: 48 description bits, then the aux bits */
    uint32_t prices[ZK_VM_OPCODE_TABLE_ROWS];      /* OPCODES_PRICES */
    /* description bits: [type 16 | variant 10 | flags 2 | src mode 6 | dst mode 4] = 38 meaningful of 48, aux from bit 48 */
    uint32_t type_bits, variant_bits, flag_bits, src_mode_bits, dst_mode_bits, description_bits_flattened, aux_bits;
    uint32_t aux_kernel_mode, aux_static_ok, aux_explicit_panic; /* KERNER_MODE_FLAG_IDX, CAN_BE_USED_IN_STATIC_CONTEXT_FLAG_IDX, EXPLICIT_PANIC_FLAG_IDX */
    uint32_t variant_idx[ZK_VMV__COUNT];           /* index into opcode_variant_booleans */
    uint32_t flag_idx[ZK_VMFL__COUNT];             /* index into flag_booleans */
    uint32_t condition_idx[ZK_VMC__COUNT];         /* Condition::variant_index(): key of the conditional resolution table */
    uint32_t can_write_dst0_into_memory[ZK_VMF__COUNT];
    uint64_t nop_encoding, panic_encoding;         /* EncodingModeProduction::nop_encoding() / exception_revert_encoding() */
    uint64_t nop_bitspread, panic_bitspread;       /* NOP_BITSPREAD_U64 / PANIC_BITSPREAD_U64 */
    uint32_t params[ZK_VMP__COUNT];
} zk_opcode_defs;

/* [EXT] this build's blob (csrc/circuits/opcode_defs.cpp).  Opcode word layout (EncodingModeProduction, visible at
 * src/main_vm/decoded_opcode.rs:408-514): bits 0..11 variant, 11..13 unused, 13..16 condition, 16..20 src0 register, 20..24 src1,
 * 24..28 dst0, 28..32 dst1, 32..48 imm0, 48..64 imm1. */
int zk_opcode_defs_default(zk_opcode_defs *out);
/* index of the table row that encodes (family, variant index inside the family, src mode, dst mode, flag bits), or -1 */
int zk_opcode_defs_find(const zk_opcode_defs *defs, uint32_t family, uint32_t variant, uint32_t src_mode, uint32_t dst_mode, uint32_t flags);

/* main_vm_entry_point (src/main_vm/mod.rs:47-232) recorded with `limit` cycles: vm_cycle (src/main_vm/cycle.rs:28-795) is the
 * loop body.  configure: geometry check (src/main_vm/cycle.rs:959-966), gate set, the VM tables of the src/tables sources built from
 * the blob, BinopTable, Xor8 (range checks).  Input streams: see csrc/circuits/main_vm.cpp and zk_circuit_main_vm_layout. */
int zk_circuit_main_vm_configure(zk_cs *cs, const zk_opcode_defs *defs);
/* The same with options.  ZK_VM_CFG_U32_FMA_ROLE: do NOT allow U8x4FMAGate (ZK_GATE_U8X4_FMA); the mul / div relation
 * (enforce_mul_relation, src/main_vm/opcodes/mod.rs:130-180) is then recorded with the engine's one-relation u32 gate
 * ZK_GATE_U32_FMA, the decomposition of rounds 1-3 — kept to report both row counts; the reference has only the U8x4FMAGate
 * branch (`else { unimplemented!() }`), which is what zk_circuit_main_vm_configure records. */
#define ZK_VM_CFG_U32_FMA_ROLE 1u
int zk_circuit_main_vm_configure_flags(zk_cs *cs, const zk_opcode_defs *defs, uint32_t flags);
int zk_circuit_main_vm(zk_cs *cs, uint32_t limit);
/* text description of the input streams of the recorded circuit, one field per line: "<scope> <name> <first word> <n words>\n"
 * (scope = outer | loop).  buf = NULL returns the size. */
int zk_circuit_main_vm_layout(zk_cs *cs, char *buf, size_t max_bytes, size_t *n_bytes);

/* ---- product-side input path of main_vm: VmCircuitWitness { closed_form_input, witness_oracle }
 * (/root/reference/src/fsm_input_output/circuit_inputs/main_vm.rs:64-71) -> the circuit's two input streams.
 *
 * The reference's WitnessOracle (src/main_vm/witness_oracle.rs:45-91) is a trait whose getters answer only under `execute`; the
 * production implementation keeps one FIFO per getter.  zk_vm_witness_oracle is exactly that: the answers of every getter in call
 * order, consumed front to back by the calls made with execute == true.  The streams want every answer AT ITS CYCLE, and which
 * cycle pops which queue depends on the VM state (decode, `execute` flags) — so the packer walks the cycles natively
 * (csrc/vm_native.hpp: the same walker the device seeding runs) and never hashes. */
typedef struct zk_vm_memory_witness { uint32_t value[8]; uint32_t is_ptr; } zk_vm_memory_witness;        /* MemoryWitness :9-13 */
typedef struct zk_vm_callstack_witness { uint64_t context[42]; uint64_t state[12]; } zk_vm_callstack_witness;
                                            /* (ExecutionContextRecordWitness flattened as saved_context.rs:279-323, [F; 12]) */
typedef struct zk_vm_witness_oracle {
    const zk_vm_memory_witness *memory_reads; size_t n_memory_reads;             /* get_memory_witness_for_read: code word, src0, UMA a, UMA b */
    const uint32_t (*storage_reads)[8]; size_t n_storage_reads;                  /* get_storage_read_witness (needs_witness && execute): LOG, far-call code hash */
    const uint32_t *refunds; size_t n_refunds;                                   /* get_refunds */
    const uint64_t (*rollback_queue_witness)[4]; size_t n_rollback_queue_witness;/* get_rollback_queue_witness: the claimed previous head */
    const uint64_t (*rollback_tails_for_call)[4]; size_t n_rollback_tails_for_call; /* get_rollback_queue_tail_witness_for_call: near / far call */
    const zk_vm_callstack_witness *callstack; size_t n_callstack;                /* get_callstack_witness: ret */
    const uint32_t *decommit_pages; size_t n_decommit_pages;                     /* get_decommittment_request_suggested_page */
} zk_vm_witness_oracle;
/* VmCircuitInputOutputWitness: start_flag, VmInputData (circuit_inputs/main_vm.rs:9-17), hidden_fsm_input = VmLocalState flattened in
 * declaration order (src/base_structures/vm_state/mod.rs:92-109; 243 words, ignored under start_flag) */
typedef struct zk_vm_closed_form_input {
    uint32_t start_flag;
    uint64_t rollback_queue_tail_for_block[4];
    uint64_t memory_queue_initial_tail[12]; uint32_t memory_queue_initial_length;
    uint64_t decommitment_queue_initial_tail[12]; uint32_t decommitment_queue_initial_length;
    uint32_t zkporter_is_available; uint32_t default_aa_code_hash[8];
    uint64_t hidden_fsm_input[243];
} zk_vm_closed_form_input;
typedef struct zk_vm_pack_report {
    size_t used_memory_reads, used_storage_reads, used_refunds, used_rollback_queue_witness, used_rollback_tails_for_call, used_callstack,
        used_decommit_pages;      /* how far every FIFO was consumed: the next chunk of the same execution continues there */
    uint32_t underflow;           /* a getter was called under `execute` with its FIFO empty (answered with zeros) */
    uint64_t final_state[243];    /* hidden_fsm_output of this chunk (hash-chain words valid only with ZK_VM_PACK_FILL_STATE) */
} zk_vm_pack_report;
/* what a VmCircuitInputOutputWitness holds beyond the circuit's inputs: the prover-side expectation of the outputs (compared with
 * zk_cs_hook_compare_witness against the hook groups "hidden_fsm_output" / "observable_output") */
typedef struct zk_vm_closed_form_rest {
    uint32_t completion_flag;
    zk_queue_state_witness log_queue_final_state;                       /* VmOutputData, circuit_inputs/main_vm.rs:32-38 */
    zk_full_queue_state_witness memory_queue_final_state, decommitment_queue_final_state;
    uint64_t hidden_fsm_output[243];
} zk_vm_closed_form_rest;
/* bincode 1.x bytes of VmCircuitInputOutputWitness<F> (ClosedFormInputWitness: start_flag, completion_flag, observable_input,
 * observable_output, hidden_fsm_input, hidden_fsm_output — src/fsm_input_output/mod.rs:42-47) in serde's derive order; the [EXT] leaf
 * forms are those of the other decoders (include/zkgl_witness.h: field element = u64, U256 / Address = hex strings, UInt16 = u16,
 * UInt8 = u8, Boolean = 1 byte).  `rest` may be NULL.  VmCircuitWitness's second member, the witness oracle, is a host-defined type
 * (any `W: WitnessOracle`): its per-getter FIFOs are handed over as zk_vm_witness_oracle. */
int zk_decode_vm_closed_form_input_bincode(const uint8_t *bytes, size_t n_bytes, zk_vm_closed_form_input *out, zk_vm_closed_form_rest *rest,
                                           size_t *consumed);
#define ZK_VM_PACK_FILL_STATE 1u  /* also write the 243 VmLocalState words of every cycle (host-side chains: one core, for hosts
                                     that want a finished stream; the default leaves them to zk_cs_seed_stream on the device) */
/* The queue states a witness generator holds beside the oracle: the tail of each queue the VM pushes to, after every push of this chunk
 * — the memory queue (= previous tails of the RAM permutation's unsorted-queue witness), the code decommitment queue (= previous tails of
 * sort_decommittment_requests' witness), the forward log queue (the log-queue simulator's states).  With them the packer writes every
 * VmLocalState word of every cycle WITHOUT hashing these chains (only a call's callstack-sponge push is hashed on the host: four
 * permutations per call).  `used_*` are outputs: how far the chunk consumed each array (the next chunk continues there).
 * ZK_VM_PACK_RECORD_STATES (with ZK_VM_PACK_FILL_STATE): the packer hashes the chains and WRITES the tails into the arrays (capacity
 * n_*) — the witness generator's role, for fixtures; ZK_VM_PACK_STATES_FROM_WITNESS: it READS them instead of hashing. */
typedef struct zk_vm_queue_states {
    uint64_t (*memory_tails)[12]; size_t n_memory_tails, used_memory_tails;
    uint64_t (*decommit_tails)[12]; size_t n_decommit_tails, used_decommit_tails;
    uint64_t (*log_forward_tails)[4]; size_t n_log_forward_tails, used_log_forward_tails;
    size_t host_permutations;   /* out: Poseidon2 permutations the packer ran on the host for this chunk */
} zk_vm_queue_states;
#define ZK_VM_PACK_RECORD_STATES 2u
#define ZK_VM_PACK_STATES_FROM_WITNESS 4u
/* The host side of device seeding: `loop_words` holds ONLY the oracle rows of the loop stream — loop_words[(w - 243) * (batch * limit) +
 * instance * limit + cycle] for the words w >= 243 of the recorded layout — and is copied to row 243 of the device stream in one piece
 * (the stream is word-major: rows 243.. are contiguous).  The 243 VmLocalState rows (words 0..242 of every cycle) are written on the
 * device by zk_cs_seed_stream / zk_cs_seed_window_async, every one of them for every cycle, so a host that seeds there neither fills nor
 * copies them: 117 of 360 rows cross PCIe.  Excludes ZK_VM_PACK_FILL_STATE / _STATES_FROM_WITNESS. */
#define ZK_VM_PACK_ORACLE_WORDS_ONLY 8u
/* zk_pack_main_vm_witness with the queue states (flags as above; `states` may be NULL when neither flag is set).  A chunk that needs
 * more states than the arrays hold reports `underflow` (ZK_VM_PACK_STATES_FROM_WITNESS) or fails with ZK_ERR_CAPACITY (RECORD). */
int zk_pack_main_vm_witness_states(zk_cs *cs, const zk_vm_closed_form_input *input, const zk_vm_witness_oracle *oracle, zk_vm_queue_states *states,
                                   uint32_t instance, uint32_t batch, uint64_t *outer_words, uint64_t *loop_words, uint32_t flags,
                                   zk_vm_pack_report *report);
/* One instance (chunk of `limit` cycles of the recorded circuit) into the batch's host staging arrays, in the layout of every other
 * packer (include/zkgl_witness.h): outer_words[w * batch + instance], loop_words[w * (batch * limit) + instance * limit + cycle].
 * ZK_ERR_INVALID: cs is not a recorded main_vm circuit.  FIFO underflow is reported, not fatal (the circuit will reject the trace). */
int zk_pack_main_vm_witness(zk_cs *cs, const zk_vm_closed_form_input *input, const zk_vm_witness_oracle *oracle, uint32_t instance, uint32_t batch,
                            uint64_t *outer_words, uint64_t *loop_words, uint32_t flags, zk_vm_pack_report *report);

/* n_instances chunks (inputs[j], oracles[j], states[j] — `states` may be NULL —, reports[j]) into instances first_instance + j of the batch,
 * on n_threads host threads (0: all; include/zkgl_witness.h zk_parallel_for).  The chunks are independent: chunks of ONE execution depend
 * on each other only through hidden_fsm_input / the FIFO positions, which a witness generator holds per chunk
 * (/root/reference/src/fsm_input_output/circuit_inputs/main_vm.rs:64-71).  Returns the code of the lowest failing chunk. */
int zk_pack_main_vm_witness_batch(zk_cs *cs, uint32_t n_instances, const zk_vm_closed_form_input *inputs, const zk_vm_witness_oracle *oracles,
                                  zk_vm_queue_states *states, uint32_t first_instance, uint32_t batch, uint64_t *outer_words, uint64_t *loop_words,
                                  uint32_t flags, zk_vm_pack_report *reports, uint32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
