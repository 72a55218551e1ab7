"""zkgl — Python binding (ctypes) of libzkgl.so, the MI355X-native witness-generation and
constraint-evaluation engine for the era-zkevm_circuits hot path.

Everything here goes through the C ABI declared in include/zkgl.h; there is no CPU fallback:
loading fails loudly when the HIP library has not been built, and every compute entry fails with
ZkError(ZK_ERR_HIP) when no GPU is visible.

Reference surface mirrored (names follow boojum as used by /root/reference):
  ConstraintSystem.alloc_multiple_variables_without_values / allocate_constant / perform_lookup /
  pad_and_shrink(=finalize) / check_if_satisfied / print_gate_stats(=stats)
  (/root/reference/src/ram_permutation/mod.rs:419-556).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ZKGL_LIB") or os.path.join(os.path.dirname(_HERE), "libzkgl.so")  # ZKGL_LIB: kernel-variant experiments

P = 0xFFFFFFFF00000001

ZK_OK = 0
ZK_ERR_INVALID = -1
ZK_ERR_HIP = -2
ZK_ERR_UNRESOLVED = -3
ZK_ERR_CAPACITY = -4
ZK_ERR_UNSATISFIED = -5
ZK_ERR_GATE_NOT_ALLOWED = -6

# zk_opcode / zk_gate_kind / zk_link_kind (include/zkgl_ir.h)
OP = dict(END=0, CONST=1, INPUT=2, FMA=3, LC4=4, SELECT=5, ISZERO=6, UADD=7, USUB=8, DOT4=9, MATMUL12=10,
          SPLIT=11, LOOKUP=12, POSEIDON2=13, P2_ROUNDS=14, LOOP_LAST=15, U32MULADD=16, DIVREM=18, NN_MULMOD=19, KECCAK_ABSORB=20, SHA256_COMPRESS=21, BARRIER=22, U256_MULWIDE=23, U256_DIVREM=24, U8X4FMA=25)
GATE = dict(NOP=0, CONST=1, BOOLEAN=2, FMA=3, REDUCTION4=4, SELECT=5, ZEROCHECK=6, UINTX_ADD=7, DOT4=8,
            MATMUL12_EXT=9, MATMUL12_INT=10, PUBLIC_INPUT=11, U32_FMA=12, REDUCTION_BY_POWERS4=13, U8X4_FMA=14)
GATE_NAMES = {v: k for k, v in GATE.items()}
LINK = dict(CARRY=0, FIRST=1, LAST=2, BCAST=3)


class ZkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"zkgl error {code}: {msg}")
        self.code = code
        self.msg = msg


class _Geometry(C.Structure):
    _fields_ = [("num_columns_under_copy_permutation", C.c_uint32), ("num_witness_columns", C.c_uint32),
                ("num_constant_columns", C.c_uint32), ("max_allowed_constraint_degree", C.c_uint32)]


class _Failure(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("scope", "instance", "iteration", "slot", "kind", "relation")]


class _Stats(C.Structure):
    _fields_ = [("rows_per_instance", C.c_uint64), ("loop_slots", C.c_uint64), ("outer_slots", C.c_uint64),
                ("limit", C.c_uint64), ("copy_columns", C.c_uint64), ("lookup_columns", C.c_uint64),
                ("variables_outer", C.c_uint64), ("variables_loop", C.c_uint64),
                ("constraints_per_instance", C.c_uint64), ("var_cells_per_instance", C.c_uint64),
                ("gate_instances", C.c_uint64 * 16), ("lookups_per_instance", C.c_uint64),
                ("program_words_outer", C.c_uint64), ("program_words_loop", C.c_uint64),
                ("scratch_cells_outer", C.c_uint64), ("scratch_cells_loop", C.c_uint64),
                ("cells_written_outer", C.c_uint64), ("cells_written_loop", C.c_uint64),
                ("copy_pairs_outer", C.c_uint64), ("copy_pairs_loop", C.c_uint64),
                ("seed_ops", C.c_uint64), ("seed_words", C.c_uint64), ("seed_slots", C.c_uint64), ("loop_ops", C.c_uint64),
                ("cells_populated_outer", C.c_uint64), ("cells_populated_loop", C.c_uint64),
                ("loop_store_tile_lanes", C.c_uint64),
                ("constraints_from_store_fused", C.c_uint64), ("constraints_in_witness_fused", C.c_uint64),
                ("values_below_2_32_outer", C.c_uint64), ("values_below_2_32_loop", C.c_uint64),
                ("seed_cone_unsupported", C.c_uint64),
                ("store_bytes_per_lane_loop", C.c_uint64), ("narrow_store_bytes_per_lane_loop", C.c_uint64), ("narrow_byte_values_loop", C.c_uint64),
                ("narrow_store_active", C.c_uint64), ("narrow_steps", C.c_uint64), ("narrow_repeats", C.c_uint64), ("narrow_store_pending", C.c_uint64)]


_lib = None


def lib() -> C.CDLL:
    """Load libzkgl.so (built in-tree by `era-zkevm_circuits_amd/build.sh` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). zkgl has no CPU fallback.")
        _lib = C.CDLL(_LIB_PATH)
        _lib.zk_last_error.restype = C.c_char_p
    return _lib


_testlib = None


def testlib() -> C.CDLL:
    """libzkgl_testcircuits.so: circuits that exist for the tests only (csrc/testing/), built beside libzkgl.so and linked against it"""
    global _testlib
    if _testlib is None:
        lib()
        _testlib = C.CDLL(os.path.join(os.path.dirname(_LIB_PATH), "libzkgl_testcircuits.so"))
    return _testlib


def _check(rc: int):
    if rc != 0:
        raise ZkError(rc, lib().zk_last_error().decode())


_inited_device = None


def init(device: int = 0):
    """zk_init: select the GPU and upload the Poseidon2 constants. Raises without a GPU."""
    global _inited_device
    _check(lib().zk_init(C.c_int(device)))
    _inited_device = device


def device_count() -> int:
    return int(lib().zk_device_count())


def poseidon_round_constants() -> np.ndarray:
    out = np.zeros(360, dtype=np.uint64)
    _check(lib().zk_poseidon_round_constants(out.ctypes.data_as(C.c_void_p)))
    return out


BUILD_BYTEBUF_KERNEL, BUILD_SHA4_KERNEL = 1, 16   # (round 6: one build, every device path in it; bits 2, 4, 8, 32 named variants that no longer exist)


def build_features() -> int:
    """zk_build_features: macro-op device backends the loaded library carries (round 6: one build, always BYTEBUF | SHA4)"""
    f = lib().zk_build_features
    f.restype = C.c_uint32
    return int(f())


def host_threads() -> int:
    """zk_host_threads: hardware threads of the host (what n_threads = 0 means to the host pool)"""
    return int(lib().zk_host_threads())


def _ptr(x):
    if isinstance(x, DeviceBuffer):
        return C.c_void_p(x.ptr)
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):  # torch tensor on the GPU
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"not a device pointer: {type(x)}")


# ---- product-side witness packers (include/zkgl_witness.h)
class MemoryQueryWitness(C.Structure):
    _fields_ = [("timestamp", C.c_uint32), ("memory_page", C.c_uint32), ("index", C.c_uint32), ("rw_flag", C.c_uint8), ("is_ptr", C.c_uint8),
                ("value", C.c_uint32 * 8)]


class FullQueueStateWitness(C.Structure):
    _fields_ = [("head", C.c_uint64 * 12), ("tail", C.c_uint64 * 12), ("length", C.c_uint32)]


class RamFsmWitness(C.Structure):
    _fields_ = [("lhs_accumulator", C.c_uint64 * 2), ("rhs_accumulator", C.c_uint64 * 2),
                ("current_unsorted_queue_state", FullQueueStateWitness), ("current_sorted_queue_state", FullQueueStateWitness),
                ("previous_sorting_key", C.c_uint32 * 3), ("previous_full_key", C.c_uint32 * 2), ("previous_value", C.c_uint32 * 8),
                ("previous_is_ptr", C.c_uint8), ("num_nondeterministic_writes", C.c_uint32)]


class RamPermutationWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8),
                ("unsorted_queue_initial_state", FullQueueStateWitness), ("sorted_queue_initial_state", FullQueueStateWitness),
                ("non_deterministic_bootloader_memory_snapshot_length", C.c_uint32),
                ("hidden_fsm_input", RamFsmWitness), ("hidden_fsm_output", RamFsmWitness),
                ("unsorted_queue_witness", C.POINTER(MemoryQueryWitness)), ("n_unsorted", C.c_uint32),
                ("sorted_queue_witness", C.POINTER(MemoryQueryWitness)), ("n_sorted", C.c_uint32),
                ("unsorted_previous_tails", C.POINTER(C.c_uint64 * 12)), ("sorted_previous_tails", C.POINTER(C.c_uint64 * 12))]


RAM_OUTER_WORDS, RAM_LOOP_WORDS = 121, 72


def pack_ram_witness(w: RamPermutationWitness, limit: int, instance: int, outer: np.ndarray, loop: np.ndarray):
    """zk_pack_ram_witness: one instance into the host staging arrays outer [121, B], loop [72, B * limit] (C-contiguous u64)"""
    batch = outer.shape[1]
    assert outer.shape == (RAM_OUTER_WORDS, batch) and loop.shape == (RAM_LOOP_WORDS, batch * limit)
    assert outer.dtype == np.uint64 and loop.dtype == np.uint64 and outer.flags.c_contiguous and loop.flags.c_contiguous
    _check(lib().zk_pack_ram_witness(C.byref(w), limit, instance, batch, outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p)))


def decode_ram_witness_bincode(data: bytes, max_elements: int, keep_tails: bool = False):
    """zk_decode_ram_witness_bincode[_tails] -> (RamPermutationWitness, bytes consumed); the element arrays stay referenced by the result.
    keep_tails: also keep the previous tail of every queue element (the packer then writes the queue heads of every cycle)"""
    w = RamPermutationWitness()
    ub, sb = (MemoryQueryWitness * max(max_elements, 1))(), (MemoryQueryWitness * max(max_elements, 1))()
    used = C.c_size_t(0)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    if keep_tails:
        ut, stl = ((C.c_uint64 * 12) * max(max_elements, 1))(), ((C.c_uint64 * 12) * max(max_elements, 1))()
        _check(lib().zk_decode_ram_witness_bincode_tails(buf, C.c_size_t(len(data)), C.byref(w), ub, max_elements, sb, max_elements, ut, stl, C.byref(used)))
        w._keep = (ub, sb, ut, stl)
    else:
        _check(lib().zk_decode_ram_witness_bincode(buf, C.c_size_t(len(data)), C.byref(w), ub, max_elements, sb, max_elements, C.byref(used)))
        w._keep = (ub, sb)
    return w, used.value


def ram_head_words():
    arr = (C.c_uint32 * 24)()
    lib().zk_ram_head_words(arr)
    return [int(x) for x in arr]


class LogQueryWitness(C.Structure):
    _fields_ = [("address", C.c_uint32 * 5), ("key", C.c_uint32 * 8), ("read_value", C.c_uint32 * 8), ("written_value", C.c_uint32 * 8),
                ("aux_byte", C.c_uint8), ("rw_flag", C.c_uint8), ("rollback", C.c_uint8), ("is_service", C.c_uint8), ("shard_id", C.c_uint8),
                ("tx_number_in_block", C.c_uint32), ("timestamp", C.c_uint32)]


class QueueStateWitness(C.Structure):
    _fields_ = [("head", C.c_uint64 * 4), ("tail", C.c_uint64 * 4), ("length", C.c_uint32)]


class StorageFsmWitness(C.Structure):
    _fields_ = [("lhs_accumulator", C.c_uint64 * 2), ("rhs_accumulator", C.c_uint64 * 2),
                ("current_unsorted_queue_state", QueueStateWitness), ("current_intermediate_sorted_queue_state", QueueStateWitness),
                ("current_final_sorted_queue_state", QueueStateWitness), ("cycle_idx", C.c_uint32), ("previous_packed_key", C.c_uint32 * 13),
                ("previous_key", C.c_uint32 * 8), ("previous_address", C.c_uint32 * 5), ("previous_timestamp", C.c_uint32),
                ("this_cell_has_explicit_read_and_rollback_depth_zero", C.c_uint8), ("this_cell_base_value", C.c_uint32 * 8),
                ("this_cell_current_value", C.c_uint32 * 8), ("this_cell_current_depth", C.c_uint32)]


class TimestampedLogRecordWitness(C.Structure):
    _fields_ = [("record", LogQueryWitness), ("timestamp", C.c_uint32)]


class StorageValidityWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8), ("shard_id_to_process", C.c_uint8),
                ("unsorted_log_queue_state", QueueStateWitness), ("intermediate_sorted_queue_state", QueueStateWitness),
                ("hidden_fsm_input", StorageFsmWitness), ("hidden_fsm_output", StorageFsmWitness),
                ("unsorted_queue_witness", C.POINTER(LogQueryWitness)), ("n_unsorted", C.c_uint32),
                ("intermediate_sorted_queue_witness", C.POINTER(TimestampedLogRecordWitness)), ("n_sorted", C.c_uint32),
                ("unsorted_previous_tails", C.POINTER(C.c_uint64 * 4)), ("sorted_previous_tails", C.POINTER(C.c_uint64 * 4)),
                ("output_tails", C.POINTER(C.c_uint64 * 4)), ("n_output_tails", C.c_uint32)]


class LogSorterFsmWitness(C.Structure):
    _fields_ = [("lhs_accumulator", C.c_uint64 * 2), ("rhs_accumulator", C.c_uint64 * 2), ("initial_unsorted_queue_state", QueueStateWitness),
                ("intermediate_sorted_queue_state", QueueStateWitness), ("final_result_queue_state", QueueStateWitness),
                ("previous_key", C.c_uint32), ("previous_item", LogQueryWitness)]


class LogSorterWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8), ("initial_log_queue_state", QueueStateWitness),
                ("intermediate_sorted_queue_state", QueueStateWitness), ("hidden_fsm_input", LogSorterFsmWitness),
                ("hidden_fsm_output", LogSorterFsmWitness), ("initial_queue_witness", C.POINTER(LogQueryWitness)), ("n_initial", C.c_uint32),
                ("intermediate_sorted_queue_witness", C.POINTER(LogQueryWitness)), ("n_sorted", C.c_uint32),
                ("initial_previous_tails", C.POINTER(C.c_uint64 * 4)), ("sorted_previous_tails", C.POINTER(C.c_uint64 * 4)),
                ("output_tails", C.POINTER(C.c_uint64 * 4)), ("n_output_tails", C.c_uint32)]


def _pack(fn, w, limit, instance, outer, loop, n_outer, n_loop):
    batch = outer.shape[1]
    assert outer.shape == (n_outer, batch) and loop.shape == (n_loop, batch * limit)
    assert outer.dtype == np.uint64 and loop.dtype == np.uint64 and outer.flags.c_contiguous and loop.flags.c_contiguous
    _check(fn(C.byref(w), limit, instance, batch, outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p)))


def pack_storage_witness(w, limit, instance, outer, loop):
    """zk_pack_storage_witness: outer [97, B], loop [140, B * limit]"""
    _pack(lib().zk_pack_storage_witness, w, limit, instance, outer, loop, 97, 140)


def pack_log_sorter_witness(w, limit, instance, outer, loop):
    """zk_pack_log_sorter_witness: outer [87, B], loop [129, B * limit]"""
    _pack(lib().zk_pack_log_sorter_witness, w, limit, instance, outer, loop, 87, 129)


class Eip4844Witness(C.Structure):
    _fields_ = [("versioned_hash", C.c_uint8 * 32), ("linear_hash_output", C.c_uint8 * 32), ("data_chunks", C.POINTER(C.c_uint8)), ("n_chunks", C.c_uint32)]


def eip4844_stream_shape(n_chunks: int):
    it, lw = C.c_uint32(0), C.c_uint32(0)
    _check(lib().zk_eip4844_stream_shape(n_chunks, C.byref(it), C.byref(lw)))
    return it.value, lw.value


def pack_eip4844_witness(blob: bytes, versioned_hash: bytes, linear_hash: bytes, instance, outer, loop, full: bool = False):
    """zk_pack_eip4844_witness: outer [64, B], loop [loop_words, B * iterations]; full: zk_pack_eip4844_witness_full (the 217 carried
    words too — nothing to seed; declare zk_eip4844_given_words = all of them)"""
    n_chunks = len(blob) // 31
    assert len(blob) == 31 * n_chunks and len(versioned_hash) == 32 and len(linear_hash) == 32
    w = Eip4844Witness()
    w.versioned_hash[:] = versioned_hash; w.linear_hash_output[:] = linear_hash
    buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
    w.data_chunks, w.n_chunks = buf, n_chunks
    batch = outer.shape[1]
    it, lw = eip4844_stream_shape(n_chunks)
    assert outer.shape == (64, batch) and loop.shape == (lw, batch * it) and outer.flags.c_contiguous and loop.flags.c_contiguous
    fn = lib().zk_pack_eip4844_witness_full if full else lib().zk_pack_eip4844_witness
    _check(fn(C.byref(w), instance, batch, outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p)))


class Sha256FsmWitness(C.Structure):
    _fields_ = [("read_precompile_call", C.c_uint8), ("read_words_for_round", C.c_uint8), ("completed", C.c_uint8),
                ("sha256_inner_state", C.c_uint32 * 8), ("timestamp_to_use_for_read", C.c_uint32), ("timestamp_to_use_for_write", C.c_uint32),
                ("input_page", C.c_uint32), ("input_offset", C.c_uint32), ("output_page", C.c_uint32), ("output_offset", C.c_uint32),
                ("num_rounds", C.c_uint32), ("log_queue_state", QueueStateWitness), ("memory_queue_state", FullQueueStateWitness)]


class Sha256RoundFunctionWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8), ("initial_log_queue_state", QueueStateWitness),
                ("initial_memory_queue_state", FullQueueStateWitness), ("hidden_fsm_input", Sha256FsmWitness), ("hidden_fsm_output", Sha256FsmWitness),
                ("requests_queue_witness", C.POINTER(LogQueryWitness)), ("n_requests", C.c_uint32),
                ("memory_reads_witness", C.POINTER(C.c_uint32 * 8)), ("n_reads", C.c_uint32)]


def pack_sha256_witness(w, limit, instance, outer, loop):
    """zk_pack_sha256_witness: outer [87, B], loop [112, B * limit]"""
    _pack(lib().zk_pack_sha256_witness, w, limit, instance, outer, loop, 87, 112)


class KeccakFsmWitness(C.Structure):
    _fields_ = [("read_precompile_call", C.c_uint8), ("read_unaligned_words_for_round", C.c_uint8), ("padding_round", C.c_uint8), ("completed", C.c_uint8),
                ("keccak_internal_state", ((C.c_uint8 * 8) * 5) * 5), ("timestamp_to_use_for_read", C.c_uint32), ("timestamp_to_use_for_write", C.c_uint32),
                ("input_page", C.c_uint32), ("input_memory_byte_offset", C.c_uint32), ("input_memory_byte_length", C.c_uint32), ("output_page", C.c_uint32),
                ("output_word_offset", C.c_uint32), ("needs_full_padding_round", C.c_uint8), ("buffer_bytes", C.c_uint8 * 192), ("buffer_filled", C.c_uint32),
                ("log_queue_state", QueueStateWitness), ("memory_queue_state", FullQueueStateWitness)]


class KeccakRoundFunctionWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8), ("initial_log_queue_state", QueueStateWitness),
                ("initial_memory_queue_state", FullQueueStateWitness), ("hidden_fsm_input", KeccakFsmWitness), ("hidden_fsm_output", KeccakFsmWitness),
                ("requests_queue_witness", C.POINTER(LogQueryWitness)), ("n_requests", C.c_uint32),
                ("memory_reads_witness", C.POINTER(C.c_uint32 * 8)), ("n_reads", C.c_uint32)]


def pack_keccak_witness(w, limit, instance, outer, loop):
    """zk_pack_keccak_witness: outer [474, B], loop [507, B * limit]"""
    _pack(lib().zk_pack_keccak_witness, w, limit, instance, outer, loop, 474, 507)


class DemuxFsmWitness(C.Structure):
    _fields_ = [("initial_log_queue_state", QueueStateWitness), ("output_queue_states", QueueStateWitness * 6)]


class DemuxLogQueueWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8), ("initial_log_queue_state", QueueStateWitness),
                ("hidden_fsm_input", DemuxFsmWitness), ("hidden_fsm_output", DemuxFsmWitness),
                ("initial_queue_witness", C.POINTER(LogQueryWitness)), ("n_initial", C.c_uint32)]


def pack_demux_witness(w, limit, instance, outer, loop):
    """zk_pack_demux_witness: outer [73, B], loop [71, B * limit]"""
    _pack(lib().zk_pack_demux_witness, w, limit, instance, outer, loop, 73, 71)


def pack_demux_witness_tails(w, limit, instance, outer, loop, input_previous_tails, output_tails):
    """zk_pack_demux_witness_tails: every carried word from the witness's own queue states ([n, 4] u64 each); returns the given words"""
    assert outer.shape[0] == 73 and loop.shape[0] == 71 and loop.shape[1] == outer.shape[1] * limit
    ipt = np.ascontiguousarray(input_previous_tails, dtype=np.uint64); ot = np.ascontiguousarray(output_tails, dtype=np.uint64)
    _check(lib().zk_pack_demux_witness_tails(C.byref(w), limit, instance, outer.shape[1], outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p),
                                             ipt.ctypes.data_as(C.c_void_p), ot.ctypes.data_as(C.c_void_p)))
    words = (C.c_uint32 * 35)()
    n = lib().zk_demux_given_words(words)
    return list(words[:n])


class DecommitQueryWitness(C.Structure):
    _fields_ = [("code_hash", C.c_uint32 * 8), ("page", C.c_uint32), ("is_first", C.c_uint8), ("timestamp", C.c_uint32)]


class SortDecommitsFsmWitness(C.Structure):
    _fields_ = [("initial_queue_state", FullQueueStateWitness), ("sorted_queue_state", FullQueueStateWitness), ("final_queue_state", FullQueueStateWitness),
                ("lhs_accumulator", C.c_uint64 * 2), ("rhs_accumulator", C.c_uint64 * 2), ("previous_packed_key", C.c_uint32 * 9),
                ("first_encountered_timestamp", C.c_uint32), ("previous_record", DecommitQueryWitness)]


class SortDecommitsWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8), ("initial_queue_state", FullQueueStateWitness),
                ("sorted_queue_initial_state", FullQueueStateWitness), ("hidden_fsm_input", SortDecommitsFsmWitness),
                ("hidden_fsm_output", SortDecommitsFsmWitness), ("initial_queue_witness", C.POINTER(DecommitQueryWitness)), ("n_initial", C.c_uint32),
                ("sorted_queue_witness", C.POINTER(DecommitQueryWitness)), ("n_sorted", C.c_uint32)]


def pack_sort_decommits_witness(w, limit, instance, outer, loop):
    """zk_pack_sort_decommits_witness: outer [151, B], loop [87, B * limit]"""
    _pack(lib().zk_pack_sort_decommits_witness, w, limit, instance, outer, loop, 151, 87)


class CodeUnpackerFsmWitness(C.Structure):
    _fields_ = [("sha256_inner_state", C.c_uint32 * 8), ("hash_to_compare_against", C.c_uint32 * 8), ("current_index", C.c_uint32),
                ("current_page", C.c_uint32), ("timestamp", C.c_uint32), ("num_rounds_left", C.c_uint16), ("length_in_bits", C.c_uint32),
                ("state_get_from_queue", C.c_uint8), ("state_decommit", C.c_uint8), ("finished", C.c_uint8),
                ("decommittment_requests_queue_state", FullQueueStateWitness), ("memory_queue_state", FullQueueStateWitness)]


class CodeUnpackerWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8), ("memory_queue_initial_state", FullQueueStateWitness),
                ("sorted_requests_queue_initial_state", FullQueueStateWitness), ("hidden_fsm_input", CodeUnpackerFsmWitness),
                ("hidden_fsm_output", CodeUnpackerFsmWitness), ("sorted_requests_queue_witness", C.POINTER(DecommitQueryWitness)),
                ("n_requests", C.c_uint32), ("code_words", C.POINTER(C.c_uint32 * 8)), ("n_code_words", C.c_uint32)]


def pack_code_unpacker_witness(w, limit, instance, outer, loop):
    """zk_pack_code_unpacker_witness: outer [125, B], loop [101, B * limit]"""
    _pack(lib().zk_pack_code_unpacker_witness, w, limit, instance, outer, loop, 125, 101)


def pack_linear_hasher_witness_tails(w, limit, instance, outer, loop, queue_previous_tails):
    """zk_pack_linear_hasher_witness_tails: every carried word from the witness; returns the given words (all 206)"""
    assert outer.shape[0] == 10 and loop.shape[0] == 818 and loop.shape[1] == outer.shape[1] * (limit // 17)
    qp = np.ascontiguousarray(queue_previous_tails, dtype=np.uint64)
    _check(lib().zk_pack_linear_hasher_witness_tails(C.byref(w), limit, instance, outer.shape[1], outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p),
                                                     qp.ctypes.data_as(C.c_void_p)))
    words = (C.c_uint32 * 206)()
    n = lib().zk_linear_hasher_given_words(words)
    return list(words[:n])


def pack_keccak_witness_tails(w, limit, instance, outer, loop, request_previous_tails, memory_tails):
    """zk_pack_keccak_witness_tails: every carried word from the witness; returns the given words (all 423)"""
    assert outer.shape[0] == 474 and loop.shape[0] == 507 and loop.shape[1] == outer.shape[1] * limit
    rp = np.ascontiguousarray(request_previous_tails, dtype=np.uint64); mt = np.ascontiguousarray(memory_tails, dtype=np.uint64).reshape(-1, 12)
    _check(lib().zk_pack_keccak_witness_tails(C.byref(w), limit, instance, outer.shape[1], outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p),
                                              rp.ctypes.data_as(C.c_void_p), mt.ctypes.data_as(C.c_void_p), mt.shape[0]))
    words = (C.c_uint32 * 423)()
    n = lib().zk_keccak_given_words(words)
    return list(words[:n])


def pack_sha256_witness_tails(w, limit, instance, outer, loop, request_previous_tails, memory_tails):
    """zk_pack_sha256_witness_tails: every carried word from the witness ([n, 4] request previous tails, [pushes, 12] memory tails);
    returns the given words (all 60)"""
    assert outer.shape[0] == 87 and loop.shape[0] == 112 and loop.shape[1] == outer.shape[1] * limit
    rp = np.ascontiguousarray(request_previous_tails, dtype=np.uint64); mt = np.ascontiguousarray(memory_tails, dtype=np.uint64).reshape(-1, 12)
    _check(lib().zk_pack_sha256_witness_tails(C.byref(w), limit, instance, outer.shape[1], outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p),
                                              rp.ctypes.data_as(C.c_void_p), mt.ctypes.data_as(C.c_void_p), mt.shape[0]))
    words = (C.c_uint32 * 60)()
    n = lib().zk_sha256_given_words(words)
    return list(words[:n])


def pack_sort_decommits_witness_tails(w, limit, instance, outer, loop, initial_previous_tails, sorted_previous_tails, result_tails):
    """zk_pack_sort_decommits_witness_tails: integer state and queue states from the witness ([n, 12] u64 each); returns the given words
    (all but the four grand-product words, which the device scans)"""
    assert outer.shape[0] == 151 and loop.shape[0] == 87 and loop.shape[1] == outer.shape[1] * limit
    ip = np.ascontiguousarray(initial_previous_tails, dtype=np.uint64); sp = np.ascontiguousarray(sorted_previous_tails, dtype=np.uint64)
    rt = np.ascontiguousarray(result_tails, dtype=np.uint64).reshape(-1, 12)
    _check(lib().zk_pack_sort_decommits_witness_tails(C.byref(w), limit, instance, outer.shape[1], outer.ctypes.data_as(C.c_void_p),
                                                      loop.ctypes.data_as(C.c_void_p), ip.ctypes.data_as(C.c_void_p), sp.ctypes.data_as(C.c_void_p),
                                                      rt.ctypes.data_as(C.c_void_p), rt.shape[0]))
    words = (C.c_uint32 * 65)()
    n = lib().zk_sort_decommits_given_words(words)
    return list(words[:n])


def pack_code_unpacker_witness_tails(w, limit, instance, outer, loop, request_previous_tails, memory_tails):
    """zk_pack_code_unpacker_witness_tails: every carried word from the witness ([n, 12] u64 queue states); returns the given words"""
    assert outer.shape[0] == 125 and loop.shape[0] == 101 and loop.shape[1] == outer.shape[1] * limit
    rp = np.ascontiguousarray(request_previous_tails, dtype=np.uint64); mt = np.ascontiguousarray(memory_tails, dtype=np.uint64)
    _check(lib().zk_pack_code_unpacker_witness_tails(C.byref(w), limit, instance, outer.shape[1], outer.ctypes.data_as(C.c_void_p),
                                                     loop.ctypes.data_as(C.c_void_p), rp.ctypes.data_as(C.c_void_p), mt.ctypes.data_as(C.c_void_p)))
    words = (C.c_uint32 * 74)()
    n = lib().zk_code_unpacker_given_words(words)
    return list(words[:n])


class LinearHasherWitness(C.Structure):
    _fields_ = [("start_flag", C.c_uint8), ("completion_flag", C.c_uint8), ("queue_state", QueueStateWitness),
                ("queue_witness", C.POINTER(LogQueryWitness)), ("n_queue", C.c_uint32)]


def pack_linear_hasher_witness(w, limit, instance, outer, loop):
    """zk_pack_linear_hasher_witness: outer [10, B], loop [818, B * limit / 17] (one loop iteration = a period of 17 pops)"""
    batch = outer.shape[1]
    assert limit % 17 == 0 and outer.shape == (10, batch) and loop.shape == (818, batch * (limit // 17))
    assert outer.dtype == np.uint64 and loop.dtype == np.uint64 and outer.flags.c_contiguous and loop.flags.c_contiguous
    _check(lib().zk_pack_linear_hasher_witness(C.byref(w), limit, instance, batch, outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p)))


def _decode_bincode(fn, w, data: bytes, bufs):
    used = C.c_size_t(0)
    raw = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    args = []
    for arr in bufs:
        args += [arr, len(arr)]
    _check(fn(raw, C.c_size_t(len(data)), C.byref(w), *args, C.byref(used)))
    w._keep = bufs
    return w, used.value


def _decode_bincode_tails(fn, w, data: bytes, bufs, tails):
    used = C.c_size_t(0)
    raw = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    args = []
    for arr in bufs:
        args += [arr, len(arr)]
    _check(fn(raw, C.c_size_t(len(data)), C.byref(w), *args, *tails, C.byref(used)))
    w._keep = list(bufs) + list(tails)
    return w, used.value


def decode_storage_witness_bincode(data: bytes, max_elements: int, keep_tails: bool = False):
    """zk_decode_storage_witness_bincode[_tails] -> (StorageValidityWitness, bytes consumed); keep_tails: the previous tail of every queue
    element is kept and zk_pack_storage_witness then walks the integer carried state (see storage_given_words)"""
    n = max(max_elements, 1)
    bufs = [(LogQueryWitness * n)(), (TimestampedLogRecordWitness * n)()]
    if keep_tails:
        return _decode_bincode_tails(lib().zk_decode_storage_witness_bincode_tails, StorageValidityWitness(), data, bufs, [((C.c_uint64 * 4) * n)(), ((C.c_uint64 * 4) * n)()])
    return _decode_bincode(lib().zk_decode_storage_witness_bincode, StorageValidityWitness(), data, bufs)


def decode_log_sorter_witness_bincode(data: bytes, max_elements: int, keep_tails: bool = False):
    n = max(max_elements, 1)
    bufs = [(LogQueryWitness * n)(), (LogQueryWitness * n)()]
    if keep_tails:
        return _decode_bincode_tails(lib().zk_decode_log_sorter_witness_bincode_tails, LogSorterWitness(), data, bufs, [((C.c_uint64 * 4) * n)(), ((C.c_uint64 * 4) * n)()])
    return _decode_bincode(lib().zk_decode_log_sorter_witness_bincode, LogSorterWitness(), data, bufs)


def set_output_tails(w, tails):
    """attach the output queue's tail after each push (host-simulated) to a Storage / LogSorter witness"""
    arr = ((C.c_uint64 * 4) * max(len(tails), 1))()
    for a, t in zip(arr, tails):
        a[:] = [int(x) for x in t]
    w.output_tails = C.cast(arr, C.POINTER(C.c_uint64 * 4))
    w.n_output_tails = len(tails)
    w._keep_out = arr


def storage_given_words(w):
    arr = (C.c_uint32 * 67)()
    n = lib().zk_storage_given_words(C.byref(w), arr)
    return [int(arr[i]) for i in range(n)]


def log_sorter_given_words(w):
    arr = (C.c_uint32 * 57)()
    n = lib().zk_log_sorter_given_words(C.byref(w), arr)
    return [int(arr[i]) for i in range(n)]


def decode_demux_witness_bincode(data: bytes, max_elements: int, keep_tails: bool = False):
    """keep_tails: w._keep[-1] = the previous tail of every element ([n][4]), the input of pack_demux_witness_tails"""
    n = max(max_elements, 1)
    if keep_tails:
        return _decode_bincode_tails(lib().zk_decode_demux_witness_bincode_tails, DemuxLogQueueWitness(), data, [(LogQueryWitness * n)()], [((C.c_uint64 * 4) * n)()])
    return _decode_bincode(lib().zk_decode_demux_witness_bincode, DemuxLogQueueWitness(), data, [(LogQueryWitness * n)()])


def decode_linear_hasher_witness_bincode(data: bytes, max_elements: int):
    return _decode_bincode(lib().zk_decode_linear_hasher_witness_bincode, LinearHasherWitness(), data, [(LogQueryWitness * max(max_elements, 1))()])


def decode_sha256_witness_bincode(data: bytes, max_requests: int, max_reads: int, keep_tails: bool = False):
    """keep_tails: w._keep[-1] = the previous tail of every request ([n][4]), an input of pack_sha256_witness_tails"""
    bufs = [(LogQueryWitness * max(max_requests, 1))(), ((C.c_uint32 * 8) * max(max_reads, 1))()]
    if keep_tails:
        return _decode_bincode_tails(lib().zk_decode_sha256_witness_bincode_tails, Sha256RoundFunctionWitness(), data, bufs, [((C.c_uint64 * 4) * max(max_requests, 1))()])
    return _decode_bincode(lib().zk_decode_sha256_witness_bincode, Sha256RoundFunctionWitness(), data, bufs)


def decode_keccak_witness_bincode(data: bytes, max_requests: int, max_reads: int, keep_tails: bool = False):
    bufs = [(LogQueryWitness * max(max_requests, 1))(), ((C.c_uint32 * 8) * max(max_reads, 1))()]
    if keep_tails:
        return _decode_bincode_tails(lib().zk_decode_keccak_witness_bincode_tails, KeccakRoundFunctionWitness(), data, bufs, [((C.c_uint64 * 4) * max(max_requests, 1))()])
    return _decode_bincode(lib().zk_decode_keccak_witness_bincode, KeccakRoundFunctionWitness(), data, bufs)


def decode_sort_decommits_witness_bincode(data: bytes, max_elements: int, keep_tails: bool = False):
    """keep_tails: w._keep[-2:] = the previous tails of the two queues' elements ([n][12] each), inputs of pack_sort_decommits_witness_tails"""
    n = max(max_elements, 1)
    if keep_tails:
        return _decode_bincode_tails(lib().zk_decode_sort_decommits_witness_bincode_tails, SortDecommitsWitness(), data,
                                     [(DecommitQueryWitness * n)(), (DecommitQueryWitness * n)()], [((C.c_uint64 * 12) * n)(), ((C.c_uint64 * 12) * n)()])
    return _decode_bincode(lib().zk_decode_sort_decommits_witness_bincode, SortDecommitsWitness(), data, [(DecommitQueryWitness * n)(), (DecommitQueryWitness * n)()])


def decode_code_unpacker_witness_bincode(data: bytes, max_requests: int, max_words: int, keep_tails: bool = False):
    """keep_tails: w._keep[-1] = the previous tail of every request ([n][12]), an input of pack_code_unpacker_witness_tails"""
    if keep_tails:
        return _decode_bincode_tails(lib().zk_decode_code_unpacker_witness_bincode_tails, CodeUnpackerWitness(), data,
                                     [(DecommitQueryWitness * max(max_requests, 1))(), ((C.c_uint32 * 8) * max(max_words, 1))()],
                                     [((C.c_uint64 * 12) * max(max_requests, 1))()])
    return _decode_bincode(lib().zk_decode_code_unpacker_witness_bincode, CodeUnpackerWitness(), data,
                           [(DecommitQueryWitness * max(max_requests, 1))(), ((C.c_uint32 * 8) * max(max_words, 1))()])


class Comm:
    """RCCL communicator behind the C ABI (zk_comm_*): one process per GPU; rank 0's `unique_id()` bytes reach the other ranks through
    the host's launcher (bench.py: a torch.distributed broadcast)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _check(lib().zk_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid: bytes, rank: int, world: int):
        self._h = C.c_void_p()
        self.rank, self.world, self._out = rank, world, None
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _check(lib().zk_comm_create(C.byref(self._h), buf, rank, world))

    def gathered(self) -> np.ndarray:
        """the words of the last gather_commitments_async as u64 [world, batch, n_public] (the caller synchronised the stream)"""
        w, b, n = self._shape
        return self._out.to_numpy()[: w * b * n].reshape(w, b, n)

    def close(self):
        if self._h:
            lib().zk_comm_destroy(self._h)
            self._h = C.c_void_p()


class DeviceBuffer:
    """A hipMalloc'd buffer of u64/u32 words owned by Python (zk_malloc / zk_free)."""

    def __init__(self, n_words: int, dtype=np.uint64):
        self.dtype = np.dtype(dtype)
        self.n = int(n_words)
        p = C.c_void_p()
        _check(lib().zk_malloc(C.byref(p), C.c_size_t(max(self.n, 1) * self.dtype.itemsize)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        b = cls(arr.size, arr.dtype)
        if arr.size:
            _check(lib().zk_h2d(C.c_void_p(b.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), None))
        return b

    def to_numpy(self) -> np.ndarray:
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            _check(lib().zk_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), None))
        return out

    def zero(self):
        _check(lib().zk_memset(C.c_void_p(self.ptr), 0, C.c_size_t(self.n * self.dtype.itemsize), None))
        sync()

    def free(self):
        if self.ptr:
            lib().zk_free(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync(stream=None):
    _check(lib().zk_sync(_ptr(stream)))


# ------------------------------------------------------------------------------------------------
# primitives (K1..K4, a9).  Arguments are DeviceBuffer / torch CUDA tensors / raw device addresses.
# ------------------------------------------------------------------------------------------------
def gl_fma_cols(dst, a, b, c, q: int, l: int, n: int, stream=None):
    _check(lib().zk_gl_fma_cols(_ptr(dst), _ptr(a), _ptr(b), _ptr(c), C.c_uint64(q), C.c_uint64(l), C.c_size_t(n), _ptr(stream)))


def gl_add_cols(dst, a, b, n, stream=None):
    _check(lib().zk_gl_add_cols(_ptr(dst), _ptr(a), _ptr(b), C.c_size_t(n), _ptr(stream)))


def gl_sub_cols(dst, a, b, n, stream=None):
    _check(lib().zk_gl_sub_cols(_ptr(dst), _ptr(a), _ptr(b), C.c_size_t(n), _ptr(stream)))


def gl_mul_cols(dst, a, b, n, stream=None):
    _check(lib().zk_gl_mul_cols(_ptr(dst), _ptr(a), _ptr(b), C.c_size_t(n), _ptr(stream)))


def gl_select_cols(dst, s, a, b, n, stream=None):
    _check(lib().zk_gl_select_cols(_ptr(dst), _ptr(s), _ptr(a), _ptr(b), C.c_size_t(n), _ptr(stream)))


def gl_inv_cols(dst, a, n, stream=None):
    _check(lib().zk_gl_inv_cols(_ptr(dst), _ptr(a), C.c_size_t(n), _ptr(stream)))


def poseidon2_permute_soa(states, n, stride=None, stream=None):
    _check(lib().zk_poseidon2_permute_soa(_ptr(states), C.c_size_t(n), C.c_size_t(n if stride is None else stride), _ptr(stream)))


def poseidon2_permute_aos(states, n, stream=None):
    _check(lib().zk_poseidon2_permute_aos(_ptr(states), C.c_size_t(n), _ptr(stream)))


def commit_encoding_batch(inp, length, n, out, stream=None):
    _check(lib().zk_commit_encoding_batch(_ptr(inp), C.c_size_t(length), C.c_size_t(n), _ptr(out), _ptr(stream)))


def queue_full_push_chain(enc, nq, items, tail_io, states_out=None, stream=None):
    _check(lib().zk_queue_full_push_chain(_ptr(enc), C.c_size_t(nq), C.c_size_t(items), _ptr(tail_io), _ptr(states_out), _ptr(stream)))


def memory_query_encode(q, n, enc, stream=None):
    _check(lib().zk_memory_query_encode(_ptr(q), C.c_size_t(n), _ptr(enc), _ptr(stream)))


def execution_context_encode(rec, n, enc, stream=None):
    _check(lib().zk_execution_context_encode(_ptr(rec), C.c_size_t(n), _ptr(enc), _ptr(stream)))


def grand_product(enc, flags, challenges, enc_len, n, init, acc_out, scratch, stream=None):
    _check(lib().zk_grand_product(_ptr(enc), _ptr(flags), _ptr(challenges), C.c_size_t(enc_len), C.c_size_t(n),
                                  C.c_uint64(init), _ptr(acc_out), _ptr(scratch), _ptr(stream)))


# ------------------------------------------------------------------------------------------------
# constraint system
# ------------------------------------------------------------------------------------------------
# K11: Goldilocks NTT / coset LDE over device-resident polynomials (include/zkgl.h)
def two_adic_root(log_n: int) -> int:
    out = C.c_uint64()
    _check(lib().zk_two_adic_root(C.c_uint32(log_n), C.byref(out)))
    return out.value


NTT_INVERSE, NTT_NATURAL_VALUES = 1, 2


def ntt(data, log_n: int, n_polys: int = 1, stride=None, inverse: bool = False, coset_shift: int = 1, stream=None, natural_values: bool = False):
    """in place.  forward: coefficients -> values A[k] = sum_i a[i] (coset_shift * omega^k)^i, inverse: the inverse map.
    natural_values=False: coefficients natural, values bit-reversed; True: values natural (trace rows), coefficients bit-reversed."""
    stride = (1 << log_n) if stride is None else stride
    mode = (NTT_INVERSE if inverse else 0) | (NTT_NATURAL_VALUES if natural_values else 0)
    _check(lib().zk_ntt(_ptr(data), C.c_uint32(log_n), C.c_uint32(n_polys), C.c_uint64(stride), C.c_uint32(mode),
                        C.c_uint64(coset_shift), _ptr(stream)))


def lde(coeffs, out, log_n: int, log_blowup: int, n_polys: int = 1, src_stride=None, coset_shift: int = 1, stream=None,
        natural_values: bool = False):
    """out[q][j][:] = forward transform of polynomial q on the coset coset_shift * eta^bitrev(j) * <omega_N>"""
    src_stride = (1 << log_n) if src_stride is None else src_stride
    _check(lib().zk_lde(_ptr(coeffs), C.c_uint64(src_stride), _ptr(out), C.c_uint32(log_n), C.c_uint32(log_blowup), C.c_uint32(n_polys),
                        C.c_uint32(NTT_NATURAL_VALUES if natural_values else 0), C.c_uint64(coset_shift), _ptr(stream)))


# ---- include/zkgl_vm.h
VM_FAMILY = dict(INVALID=0, NOP=1, ADD=2, SUB=3, MUL=4, DIV=5, JUMP=6, CONTEXT=7, SHIFT=8, BINOP=9, PTR=10, NEAR_CALL=11, LOG=12,
                 FAR_CALL=13, RET=14, UMA=15)
VM_VARIANT = {n: i for i, n in enumerate(
    "SHIFT_SHL SHIFT_SHR SHIFT_ROL SHIFT_ROR BINOP_XOR BINOP_AND BINOP_OR PTR_ADD PTR_SUB PTR_PACK PTR_SHRINK CTX_THIS CTX_CALLER "
    "CTX_CODE_ADDRESS CTX_META CTX_ERGS_LEFT CTX_SP CTX_GET_CONTEXT_U128 CTX_SET_CONTEXT_U128 CTX_SET_ERGS_PER_PUBDATA CTX_INC_TX_NUMBER "
    "LOG_STORAGE_READ LOG_STORAGE_WRITE LOG_TO_L1 LOG_EVENT LOG_PRECOMPILE_CALL FAR_NORMAL FAR_DELEGATE FAR_MIMIC RET_OK RET_REVERT RET_PANIC "
    "UMA_HEAP_READ UMA_HEAP_WRITE UMA_AUX_HEAP_READ UMA_AUX_HEAP_WRITE UMA_FAT_PTR_READ".split())}
VM_FLAG = {n: i for i, n in enumerate("SET_FLAGS SWAP_ARITH SWAP_PTR FIRST_MESSAGE UMA_INCREMENT FAR_CALL_STATIC FAR_CALL_SHARD RET_TO_LABEL".split())}
VM_MODE = dict(REG_ONLY=0, STACK_PUSH_POP=1, STACK_OFFSET=2, ABSOLUTE_STACK=3, IMM16=4, CODE_PAGE=5)
VM_CONDITION = dict(ALWAYS=0, LT=1, EQ=2, GT=3, GE=4, LE=5, NE=6, GT_OR_LT=7)
VM_PARAM = {n: i for i, n in enumerate(
    "VM_MAX_STACK_DEPTH NEW_FRAME_MEMORY_STIPEND NEW_MEMORY_PAGES_PER_FAR_CALL UNMAPPED_PAGE BOOTLOADER_BASE_PAGE BOOTLOADER_CODE_PAGE "
    "BOOTLOADER_CALLDATA_PAGE STARTING_BASE_PAGE STARTING_TIMESTAMP INITIAL_FRAME_FORMAL_EH_LOCATION VM_INITIAL_FRAME_ERGS "
    "BOOTLOADER_FORMAL_ADDRESS_LOW BOOTLOADER_MAX_MEMORY DEPLOYER_SYSTEM_CONTRACT_ADDRESS_LOW ERGS_PER_CODE_WORD_DECOMMITTMENT "
    "INITIAL_STORAGE_WRITE_PUBDATA_BYTES L1_MESSAGE_PUBDATA_BYTES STORAGE_AUX_BYTE EVENT_AUX_BYTE L1_MESSAGE_AUX_BYTE PRECOMPILE_AUX_BYTE "
    "CODE_HASH_VERSION_BYTE CODE_YET_CONSTRUCTED_MARKER CODE_AT_REST_MARKER FAR_CALL_FORWARDING_MODE_BYTE_IDX FAR_CALL_SHARD_ID_BYTE_IDX "
    "FAR_CALL_CONSTRUCTOR_CALL_BYTE_IDX FAR_CALL_SYSTEM_CALL_BYTE_IDX FORWARD_USE_HEAP FORWARD_FAT_POINTER FORWARD_USE_AUX_HEAP "
    "CALL_IMPLICIT_PARAMETER_REG_IDX CALL_SYSTEM_ABI_REGISTERS_BEGIN CALL_SYSTEM_ABI_REGISTERS_END CALL_RESERVED_RANGE_BEGIN "
    "CALL_RESERVED_RANGE_END".split())}


class OpcodeDefs(C.Structure):
    """zk_opcode_defs: everything main_vm takes from zkevm_opcode_defs, as one data blob"""
    _fields_ = [("version", C.c_uint32), ("n_valid", C.c_uint32), ("props", C.c_uint64 * 2048), ("prices", C.c_uint32 * 2048),
                ("type_bits", C.c_uint32), ("variant_bits", C.c_uint32), ("flag_bits", C.c_uint32), ("src_mode_bits", C.c_uint32),
                ("dst_mode_bits", C.c_uint32), ("description_bits_flattened", C.c_uint32), ("aux_bits", C.c_uint32),
                ("aux_kernel_mode", C.c_uint32), ("aux_static_ok", C.c_uint32), ("aux_explicit_panic", C.c_uint32),
                ("variant_idx", C.c_uint32 * len(VM_VARIANT)), ("flag_idx", C.c_uint32 * len(VM_FLAG)), ("condition_idx", C.c_uint32 * 8),
                ("can_write_dst0_into_memory", C.c_uint32 * 16), ("nop_encoding", C.c_uint64), ("panic_encoding", C.c_uint64),
                ("nop_bitspread", C.c_uint64), ("panic_bitspread", C.c_uint64), ("params", C.c_uint32 * len(VM_PARAM))]

    def find(self, family: int, variant: int = 0, src_mode: int = 0, dst_mode: int = 0, flags: int = 0) -> int:
        r = int(lib().zk_opcode_defs_find(C.byref(self), C.c_uint32(family), C.c_uint32(variant), C.c_uint32(src_mode), C.c_uint32(dst_mode),
                                          C.c_uint32(flags)))
        if r < 0:
            raise KeyError((family, variant, src_mode, dst_mode, flags))
        return r


class VmMemoryWitness(C.Structure):          # zk_vm_memory_witness
    _fields_ = [("value", C.c_uint32 * 8), ("is_ptr", C.c_uint32)]


class VmCallstackWitness(C.Structure):       # zk_vm_callstack_witness
    _fields_ = [("context", C.c_uint64 * 42), ("state", C.c_uint64 * 12)]


class VmWitnessOracle(C.Structure):          # zk_vm_witness_oracle: one FIFO per WitnessOracle getter (witness_oracle.rs:45-91)
    _fields_ = [("memory_reads", C.POINTER(VmMemoryWitness)), ("n_memory_reads", C.c_size_t),
                ("storage_reads", C.POINTER(C.c_uint32 * 8)), ("n_storage_reads", C.c_size_t),
                ("refunds", C.POINTER(C.c_uint32)), ("n_refunds", C.c_size_t),
                ("rollback_queue_witness", C.POINTER(C.c_uint64 * 4)), ("n_rollback_queue_witness", C.c_size_t),
                ("rollback_tails_for_call", C.POINTER(C.c_uint64 * 4)), ("n_rollback_tails_for_call", C.c_size_t),
                ("callstack", C.POINTER(VmCallstackWitness)), ("n_callstack", C.c_size_t),
                ("decommit_pages", C.POINTER(C.c_uint32)), ("n_decommit_pages", C.c_size_t)]


class VmClosedFormInput(C.Structure):        # zk_vm_closed_form_input
    _fields_ = [("start_flag", C.c_uint32), ("rollback_queue_tail_for_block", C.c_uint64 * 4),
                ("memory_queue_initial_tail", C.c_uint64 * 12), ("memory_queue_initial_length", C.c_uint32),
                ("decommitment_queue_initial_tail", C.c_uint64 * 12), ("decommitment_queue_initial_length", C.c_uint32),
                ("zkporter_is_available", C.c_uint32), ("default_aa_code_hash", C.c_uint32 * 8), ("hidden_fsm_input", C.c_uint64 * 243)]


class VmPackReport(C.Structure):             # zk_vm_pack_report
    _fields_ = [("used_memory_reads", C.c_size_t), ("used_storage_reads", C.c_size_t), ("used_refunds", C.c_size_t),
                ("used_rollback_queue_witness", C.c_size_t), ("used_rollback_tails_for_call", C.c_size_t), ("used_callstack", C.c_size_t),
                ("used_decommit_pages", C.c_size_t), ("underflow", C.c_uint32), ("final_state", C.c_uint64 * 243)]


class VmQueueStates(C.Structure):            # zk_vm_queue_states
    _fields_ = [("memory_tails", C.c_void_p), ("n_memory_tails", C.c_size_t), ("used_memory_tails", C.c_size_t),
                ("decommit_tails", C.c_void_p), ("n_decommit_tails", C.c_size_t), ("used_decommit_tails", C.c_size_t),
                ("log_forward_tails", C.c_void_p), ("n_log_forward_tails", C.c_size_t), ("used_log_forward_tails", C.c_size_t),
                ("host_permutations", C.c_size_t)]

    @staticmethod
    def over(memory: np.ndarray, decommit: np.ndarray, log_forward: np.ndarray) -> "VmQueueStates":
        """arrays [n, 12], [n, 12], [n, 4] (u64, C-contiguous): written under VM_PACK_RECORD_STATES, read under VM_PACK_STATES_FROM_WITNESS"""
        q = VmQueueStates()
        for a, w in ((memory, 12), (decommit, 12), (log_forward, 4)):
            assert a.dtype == np.uint64 and a.flags.c_contiguous and a.ndim == 2 and a.shape[1] == w
        q.memory_tails, q.n_memory_tails = memory.ctypes.data, memory.shape[0]
        q.decommit_tails, q.n_decommit_tails = decommit.ctypes.data, decommit.shape[0]
        q.log_forward_tails, q.n_log_forward_tails = log_forward.ctypes.data, log_forward.shape[0]
        q._keep = (memory, decommit, log_forward)
        return q


VM_PACK_RECORD_STATES, VM_PACK_STATES_FROM_WITNESS, VM_PACK_ORACLE_WORDS_ONLY = 2, 4, 8


class VmClosedFormRest(C.Structure):        # zk_vm_closed_form_rest
    _fields_ = [("completion_flag", C.c_uint32), ("log_queue_final_state", QueueStateWitness), ("memory_queue_final_state", FullQueueStateWitness),
                ("decommitment_queue_final_state", FullQueueStateWitness), ("hidden_fsm_output", C.c_uint64 * 243)]


def decode_vm_closed_form_input_bincode(data: bytes):
    """zk_decode_vm_closed_form_input_bincode -> (VmClosedFormInput, VmClosedFormRest, bytes consumed)"""
    cf, rest, used = VmClosedFormInput(), VmClosedFormRest(), C.c_size_t(0)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    _check(lib().zk_decode_vm_closed_form_input_bincode(buf, C.c_size_t(len(data)), C.byref(cf), C.byref(rest), C.byref(used)))
    return cf, rest, used.value


VM_PACK_FILL_STATE = 1
FAILURE_NONCANONICAL_INPUT, FAILURE_STREAM_LINK = 0x400, 0x500   # zk_failure.kind beyond the gate kinds (include/zkgl.h)


class VmOracleQueues:
    """the WitnessOracle's per-getter FIFOs as Python lists; view(used) -> the zk_vm_witness_oracle of what is left"""

    def __init__(self):
        self.memory_reads, self.storage_reads, self.refunds = [], [], []
        self.rollback_queue_witness, self.rollback_tails_for_call, self.callstack, self.decommit_pages = [], [], [], []

    def freeze(self):
        self._mem = (VmMemoryWitness * max(1, len(self.memory_reads)))()
        for i, (v, p) in enumerate(self.memory_reads):
            self._mem[i].value[:] = [int(x) for x in v]
            self._mem[i].is_ptr = int(p)
        self._sto = ((C.c_uint32 * 8) * max(1, len(self.storage_reads)))()
        for i, v in enumerate(self.storage_reads):
            self._sto[i][:] = [int(x) for x in v]
        self._ref = (C.c_uint32 * max(1, len(self.refunds)))(*[int(x) for x in self.refunds])
        self._rq = ((C.c_uint64 * 4) * max(1, len(self.rollback_queue_witness)))()
        for i, v in enumerate(self.rollback_queue_witness):
            self._rq[i][:] = [int(x) for x in v]
        self._ct = ((C.c_uint64 * 4) * max(1, len(self.rollback_tails_for_call)))()
        for i, v in enumerate(self.rollback_tails_for_call):
            self._ct[i][:] = [int(x) for x in v]
        self._cs = (VmCallstackWitness * max(1, len(self.callstack)))()
        for i, (ctx, st) in enumerate(self.callstack):
            self._cs[i].context[:] = [int(x) for x in ctx]
            self._cs[i].state[:] = [int(x) for x in st]
        self._dp = (C.c_uint32 * max(1, len(self.decommit_pages)))(*[int(x) for x in self.decommit_pages])
        return self

    def view(self, used=(0,) * 7) -> VmWitnessOracle:
        o = VmWitnessOracle()
        def sub(arr, n, k, typ):
            return C.cast(C.byref(arr, k * C.sizeof(arr._type_)), C.POINTER(typ)), n - k
        o.memory_reads, o.n_memory_reads = sub(self._mem, len(self.memory_reads), used[0], VmMemoryWitness)
        o.storage_reads, o.n_storage_reads = sub(self._sto, len(self.storage_reads), used[1], C.c_uint32 * 8)
        o.refunds, o.n_refunds = sub(self._ref, len(self.refunds), used[2], C.c_uint32)
        o.rollback_queue_witness, o.n_rollback_queue_witness = sub(self._rq, len(self.rollback_queue_witness), used[3], C.c_uint64 * 4)
        o.rollback_tails_for_call, o.n_rollback_tails_for_call = sub(self._ct, len(self.rollback_tails_for_call), used[4], C.c_uint64 * 4)
        o.callstack, o.n_callstack = sub(self._cs, len(self.callstack), used[5], VmCallstackWitness)
        o.decommit_pages, o.n_decommit_pages = sub(self._dp, len(self.decommit_pages), used[6], C.c_uint32)
        return o


def opcode_defs_default() -> OpcodeDefs:
    d = OpcodeDefs()
    _check(lib().zk_opcode_defs_default(C.byref(d)))
    return d


@dataclass
class CSGeometry:  # boojum::cs::CSGeometry (src/main_vm/cycle.rs:959-966)
    num_columns_under_copy_permutation: int
    num_witness_columns: int
    num_constant_columns: int
    max_allowed_constraint_degree: int


@dataclass
class Failure:
    scope: int
    instance: int
    iteration: int
    slot: int
    kind: int
    relation: int


class ConstraintSystem:
    """Recorder + GPU executor handle (zk_cs). Recording works without a GPU; set_batch/resolve need one."""

    def __init__(self, geometry: CSGeometry, max_trace_len: int = 1 << 20, max_variables: int = 1 << 26):
        g = _Geometry(geometry.num_columns_under_copy_permutation, geometry.num_witness_columns,
                      geometry.num_constant_columns, geometry.max_allowed_constraint_degree)
        h = C.c_void_p()
        _check(lib().zk_cs_create(C.byref(g), C.c_uint64(max_trace_len), C.c_uint64(max_variables), C.byref(h)))
        self._h = h
        self.geometry = geometry
        self._keep = []
        self._keep_inputs = {}

    def close(self):
        if self._h:
            lib().zk_cs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- configuration ---
    def allow_lookup(self, width=3, num_repetitions=8, share_table_id=True):
        _check(lib().zk_cs_allow_lookup(self._h, width, num_repetitions, int(share_table_id)))

    def allow_gate(self, kind: int):
        _check(lib().zk_cs_allow_gate(self._h, kind))

    def gate_is_allowed(self, kind: int) -> bool:
        return bool(lib().zk_cs_gate_is_allowed(self._h, kind))

    def add_lookup_table(self, marker: int, n_keys: int, n_vals: int, rows: np.ndarray) -> int:
        rows = np.ascontiguousarray(rows, dtype=np.uint64).reshape(-1, n_keys + n_vals)
        tid = C.c_uint32()
        _check(lib().zk_cs_add_table(self._h, marker, n_keys, n_vals, rows.ctypes.data_as(C.c_void_p), rows.shape[0], C.byref(tid)))
        return tid.value

    def get_table_id_for_marker(self, marker: int) -> int:
        tid = C.c_uint32()
        _check(lib().zk_cs_table_id(self._h, marker, C.byref(tid)))
        return tid.value

    # --- recording ---
    def alloc_multiple_variables_without_values(self, n: int):
        first = C.c_uint32()
        _check(lib().zk_cs_alloc_vars(self._h, n, C.byref(first)))
        return [first.value + i for i in range(n)]

    def alloc_variable_without_value(self) -> int:
        return self.alloc_multiple_variables_without_values(1)[0]

    def allocate_constant(self, value: int) -> int:
        v = C.c_uint32()
        _check(lib().zk_cs_alloc_constant(self._h, C.c_uint64(value), C.byref(v)))
        return v.value

    def input(self, word: int) -> int:
        v = C.c_uint32()
        _check(lib().zk_cs_input(self._h, word, C.byref(v)))
        return v.value

    def place_gate(self, kind: int, variables, consts=()):
        va = (C.c_uint32 * len(variables))(*variables)
        ca = (C.c_uint64 * max(len(consts), 1))(*consts)
        _check(lib().zk_cs_place_gate(self._h, kind, va, len(variables), ca, len(consts)))

    def emit_op(self, opcode: int, ins, outs, imm=(), a=0, b=0):
        ia = (C.c_uint32 * max(len(ins), 1))(*ins)
        oa = (C.c_uint32 * max(len(outs), 1))(*outs)
        ma = (C.c_uint64 * max(len(imm), 1))(*imm)
        _check(lib().zk_cs_emit_op(self._h, opcode, a, b, ia, len(ins), oa, len(outs), ma, len(imm)))

    def perform_lookup(self, table_id: int, keys, n_vals: int):
        ka = (C.c_uint32 * len(keys))(*keys)
        va = (C.c_uint32 * max(n_vals, 1))()
        _check(lib().zk_cs_lookup(self._h, table_id, ka, len(keys), va, n_vals))
        return [va[i] for i in range(n_vals)]

    def side_begin(self):
        _check(lib().zk_cs_side_begin(self._h))

    def loop_begin(self, limit: int):
        _check(lib().zk_cs_loop_begin(self._h, limit))

    def loop_end(self):
        _check(lib().zk_cs_loop_end(self._h))

    def link(self, kind: int, loop_var: int, other: int):
        _check(lib().zk_cs_link(self._h, kind, loop_var, other))

    def lookup_argument(self, beta, gamma, stream=None):
        """K10: -> (n_mismatch, array [batch, 4] = witness-side sum (a, b), table-side sum (a, b) in GF(p^2))"""
        b = (C.c_uint64 * 2)(*beta)
        g = (C.c_uint64 * 2)(*gamma)
        out = np.zeros((self.batch, 4), dtype=np.uint64)
        bad = C.c_uint32()
        _check(lib().zk_cs_lookup_argument(self._h, b, g, C.c_void_p(stream or 0), out.ctypes.data_as(C.c_void_p), self.batch, C.byref(bad)))
        return bad.value, out

    def seed_hint(self, opcode: int, ins, outs):
        a = (C.c_uint32 * len(ins))(*ins)
        b = (C.c_uint32 * len(outs))(*outs)
        _check(lib().zk_cs_seed_hint(self._h, opcode, a, len(ins), b, len(outs)))

    def stream_link(self, a_vars, b_vars, n_total: int):
        a = (C.c_uint32 * len(a_vars))(*a_vars)
        b = (C.c_uint32 * len(b_vars))(*b_vars)
        _check(lib().zk_cs_stream_link(self._h, a, len(a_vars), b, len(b_vars), n_total))

    def loop_last(self, loop_var: int) -> int:
        v = C.c_uint32()
        _check(lib().zk_cs_loop_last(self._h, loop_var, C.byref(v)))
        return v.value

    def loop_import(self, outer_var: int) -> int:
        v = C.c_uint32()
        _check(lib().zk_cs_loop_import(self._h, outer_var, C.byref(v)))
        return v.value

    def next_available_row(self) -> int:
        r = C.c_uint64()
        _check(lib().zk_cs_next_available_row(self._h, C.byref(r)))
        return r.value

    def pad_and_shrink(self):
        """pad_and_shrink + into_assembly: placement and program emission (no GPU needed)."""
        _check(lib().zk_cs_finalize(self._h))

    finalize = pad_and_shrink

    # --- circuits ---
    def configure_ram_permutation(self):
        _check(lib().zk_circuit_ram_permutation_configure(self._h))

    def ram_permutation_entry_point(self, limit: int):
        _check(lib().zk_circuit_ram_permutation(self._h, limit))

    def configure_storage_validity(self):
        _check(lib().zk_circuit_storage_validity_configure(self._h))

    def sort_and_deduplicate_storage_access_entry_point(self, limit: int, enforce_permutation: bool = True):
        _check(lib().zk_circuit_storage_validity(self._h, limit, int(enforce_permutation)))

    def configure_log_sorter(self):
        _check(lib().zk_circuit_log_sorter_configure(self._h))

    def sort_and_deduplicate_events_entry_point(self, limit: int):
        _check(lib().zk_circuit_log_sorter(self._h, limit))

    def configure_keccak(self):
        _check(lib().zk_circuit_keccak_configure(self._h))

    def keccak256_blocks_entry_point(self, n_blocks: int):
        _check(lib().zk_circuit_keccak256_blocks(self._h, n_blocks))

    def keccak_f1600(self, state):
        """zk_gadget_keccak_f1600: 200 byte variables (current scope) -> the 200 output byte variables"""
        assert len(state) == 200
        va = (C.c_uint32 * 200)(*state)
        _check(lib().zk_gadget_keccak_f1600(self._h, va))
        return list(va)

    def sha256_compress(self, state, block):
        """zk_gadget_sha256_compress: 32 state byte variables + 64 block byte variables -> the 32 output byte variables"""
        assert len(state) == 32 and len(block) == 64
        va = (C.c_uint32 * 32)(*state)
        ba = (C.c_uint32 * 64)(*block)
        _check(lib().zk_gadget_sha256_compress(self._h, va, ba))
        return list(va)

    def keccak256_round_function_entry_point(self, limit: int):
        _check(lib().zk_circuit_keccak256_round_function(self._h, limit))

    def configure_eip_4844(self):
        _check(lib().zk_circuit_eip_4844_configure(self._h))

    def eip_4844_entry_point(self, n_chunks: int):
        _check(lib().zk_circuit_eip_4844(self._h, n_chunks))

    def configure_demux_log_queue(self):
        _check(lib().zk_circuit_demux_log_queue_configure(self._h))

    def demultiplex_storage_logs_entry_point(self, limit: int):
        _check(lib().zk_circuit_demux_log_queue(self._h, limit))

    def configure_sort_decommits(self):
        _check(lib().zk_circuit_sort_decommits_configure(self._h))

    def sort_and_deduplicate_code_decommittments_entry_point(self, limit: int):
        _check(lib().zk_circuit_sort_decommits(self._h, limit))

    def configure_code_unpacker(self):
        _check(lib().zk_circuit_code_unpacker_configure(self._h))

    def unpack_code_into_memory_entry_point(self, limit: int):
        _check(lib().zk_circuit_code_unpacker(self._h, limit))

    def configure_linear_hasher(self):
        _check(lib().zk_circuit_linear_hasher_configure(self._h))

    def linear_hasher_entry_point(self, limit: int):
        _check(lib().zk_circuit_linear_hasher(self._h, limit))

    def configure_sha256(self, reference_tables: bool = False):
        """reference_tables: the reference's own width-4 table set (Maj4 / TriXor4 / Ch4 / Split4BitChunk<1,2>, lookup width 4) instead of
        the engine's 8-bit tables; the circuits recorded afterwards pick the matching SHA-256 decomposition"""
        if reference_tables:
            _check(lib().zk_circuit_sha256_configure_reference_tables(self._h))
        else:
            _check(lib().zk_circuit_sha256_configure(self._h))

    def sha256_blocks_entry_point(self, n_blocks: int):
        _check(lib().zk_circuit_sha256_blocks(self._h, n_blocks))

    def sha256_round_function_entry_point(self, limit: int):
        _check(lib().zk_circuit_sha256_round_function(self._h, limit))

    def configure_vm_shaped(self):
        _check(testlib().zk_test_circuit_vm_shaped_configure(self._h))

    def vm_shaped_entry_point(self, limit: int):
        _check(testlib().zk_test_circuit_vm_shaped(self._h, limit))

    def configure_main_vm(self, defs: "OpcodeDefs | None" = None, u8x4_fma_gate: bool = True):
        """zk_circuit_main_vm_configure: tables / gate set of the VM CS from the opcode-defs blob (include/zkgl_vm.h).
        u8x4_fma_gate=False: ZK_VM_CFG_U32_FMA_ROLE (the mul / div relation through the one-relation u32 gate of rounds 1-3)"""
        self._defs = defs if defs is not None else opcode_defs_default()
        if u8x4_fma_gate:
            _check(lib().zk_circuit_main_vm_configure(self._h, C.byref(self._defs)))
        else:
            _check(lib().zk_circuit_main_vm_configure_flags(self._h, C.byref(self._defs), 1))

    def main_vm_entry_point(self, limit: int):
        _check(lib().zk_circuit_main_vm(self._h, C.c_uint32(limit)))

    def main_vm_layout(self) -> dict:
        """{'outer': {name: (first word, n words)}, 'loop': {...}} of the recorded main_vm circuit"""
        n = C.c_size_t(0)
        _check(lib().zk_circuit_main_vm_layout(self._h, None, C.c_size_t(0), C.byref(n)))
        buf = C.create_string_buffer(n.value + 1)
        _check(lib().zk_circuit_main_vm_layout(self._h, buf, C.c_size_t(n.value), C.byref(n)))
        out = {"outer": {}, "loop": {}}
        for line in buf.raw[:n.value].decode().splitlines():
            scope, name, first, cnt = line.split()
            out[scope][name] = (int(first), int(cnt))
        return out

    def pack_main_vm_witness(self, closed_form: "VmClosedFormInput", oracle: "VmWitnessOracle", instance: int, batch: int,
                             outer_words: np.ndarray, loop_words: np.ndarray, flags: int = 0) -> "VmPackReport":
        """zk_pack_main_vm_witness: one chunk into outer_words [words, batch] / loop_words [words, batch * limit] (C-contiguous u64)"""
        assert outer_words.dtype == np.uint64 and loop_words.dtype == np.uint64 and outer_words.flags.c_contiguous and loop_words.flags.c_contiguous
        rep = VmPackReport()
        _check(lib().zk_pack_main_vm_witness(self._h, C.byref(closed_form), C.byref(oracle), C.c_uint32(instance), C.c_uint32(batch),
                                             outer_words.ctypes.data_as(C.c_void_p), loop_words.ctypes.data_as(C.c_void_p), C.c_uint32(flags),
                                             C.byref(rep)))
        return rep

    def pack_main_vm_witness_states(self, closed_form, oracle, states: "VmQueueStates", instance: int, batch: int, outer_words: np.ndarray,
                                    loop_words: np.ndarray, flags: int) -> "VmPackReport":
        """zk_pack_main_vm_witness_states: as pack_main_vm_witness, with the queue states (recorded into / read from `states`)"""
        assert outer_words.dtype == np.uint64 and loop_words.dtype == np.uint64 and outer_words.flags.c_contiguous and loop_words.flags.c_contiguous
        rep = VmPackReport()
        _check(lib().zk_pack_main_vm_witness_states(self._h, C.byref(closed_form), C.byref(oracle), C.byref(states), C.c_uint32(instance), C.c_uint32(batch),
                                                    outer_words.ctypes.data_as(C.c_void_p), loop_words.ctypes.data_as(C.c_void_p), C.c_uint32(flags),
                                                    C.byref(rep)))
        return rep

    def pack_main_vm_witness_batch(self, closed_forms, oracles, first_instance: int, batch: int, outer_words: np.ndarray, loop_words: np.ndarray,
                                   flags: int = 0, states=None, n_threads: int = 0):
        """zk_pack_main_vm_witness_batch: len(closed_forms) chunks into instances first_instance.. of the batch, on n_threads host threads
        (0: every hardware thread).  Returns the list of VmPackReport."""
        assert outer_words.dtype == np.uint64 and loop_words.dtype == np.uint64 and outer_words.flags.c_contiguous and loop_words.flags.c_contiguous
        n = len(closed_forms)
        assert len(oracles) == n and (states is None or len(states) == n)
        # ctypes arrays are taken as they are (a caller that packs the same chunks again and again builds them once)
        cfa = closed_forms if isinstance(closed_forms, C.Array) else (VmClosedFormInput * n)(*closed_forms)
        oa = oracles if isinstance(oracles, C.Array) else (VmWitnessOracle * n)(*oracles)
        sa = None if states is None else states if isinstance(states, C.Array) else (VmQueueStates * n)(*states)
        reps = (VmPackReport * n)()
        _check(lib().zk_pack_main_vm_witness_batch(self._h, C.c_uint32(n), cfa, oa, sa, C.c_uint32(first_instance), C.c_uint32(batch),
                                                   outer_words.ctypes.data_as(C.c_void_p), loop_words.ctypes.data_as(C.c_void_p), C.c_uint32(flags), reps,
                                                   C.c_uint32(n_threads)))
        if states is not None and not isinstance(states, C.Array):   # the `used_*` / host_permutations outputs travel back into the caller's objects
            for dst, src in zip(states, sa):
                C.memmove(C.byref(dst), C.byref(src), C.sizeof(VmQueueStates))
        return list(reps)

    def input_words(self):
        a, b = C.c_uint32(), C.c_uint32()
        _check(lib().zk_circuit_input_words(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # --- execution ---
    def set_batch(self, n_instances: int):
        _check(lib().zk_cs_set_batch(self._h, n_instances))
        self._batch = int(n_instances)
        self.batch = n_instances

    def bind_inputs(self, loop_scope: bool, dev_words, n_words: int, lane_stride: int = 0, lane_offset: int = 0):
        """lane_stride / lane_offset: the batch is a window of a longer stream ([word][lane_stride] u64, first lane lane_offset)"""
        self._keep_inputs[bool(loop_scope)] = dev_words   # one reference per scope: the bound buffer must outlive the binding
        # a DeviceBuffer knows its length (the C ABI takes a bare pointer): the window of the bound batch must lie inside it
        if isinstance(dev_words, DeviceBuffer) and getattr(self, "batch", 0):
            if getattr(self, "_limit_cache", None) is None:
                self._limit_cache = int(self.stats()["limit"])
            lanes = self.batch * (self._limit_cache if loop_scope else 1)
            stride = lane_stride or lanes
            if n_words and (n_words - 1) * stride + lane_offset + lanes > dev_words.n:
                raise ZkError(-1, f"bind_inputs: {n_words} words x {lanes} lanes (stride {stride}, offset {lane_offset}) exceed the buffer's {dev_words.n} words")
        if lane_stride or lane_offset:
            base = (_ptr(dev_words).value or 0) + 8 * lane_offset
            _check(lib().zk_cs_bind_inputs_window(self._h, int(loop_scope), C.c_void_p(base), n_words, C.c_uint64(lane_stride)))
        else:
            _check(lib().zk_cs_bind_inputs(self._h, int(loop_scope), _ptr(dev_words), n_words))

    def resolve(self, stream=None):
        _check(lib().zk_cs_resolve(self._h, _ptr(stream)))

    def carried_words(self):
        """input words of the loop stream that are loop-carried (what seeding fills)"""
        n = C.c_uint32(0)
        _check(lib().zk_cs_carried_words(self._h, None, 0, C.byref(n)))
        arr = (C.c_uint32 * max(1, n.value))()
        _check(lib().zk_cs_carried_words(self._h, arr, n.value, C.byref(n)))
        return [int(arr[i]) for i in range(n.value)]

    def set_seed_given(self, words):
        """zk_cs_set_seed_given: loop-carried words the host fills itself in the streams it seeds from now on ([] clears)"""
        arr = (C.c_uint32 * max(1, len(words)))(*words)
        _check(lib().zk_cs_set_seed_given(self._h, arr, len(words)))

    def seed_carried_inputs(self, dev_loop_inputs, stream=None):
        """fill the loop-carried words of the bound loop input stream sequentially on the GPU"""
        _check(lib().zk_cs_seed_carried_inputs(self._h, _ptr(dev_loop_inputs), _ptr(stream)))

    def gather_commitments(self, comm: "Comm", stream=None):
        """all ranks' public inputs -> u64 [world, batch, n_public] (zk_cs_gather_commitments: one RCCL all-gather)"""
        n = C.c_uint32(0)
        if comm._out is None or comm._out.n < comm.world * self._batch * 8:
            comm._out = DeviceBuffer(comm.world * self._batch * 8)
        _check(lib().zk_cs_gather_commitments(self._h, comm._h, _ptr(comm._out), C.byref(n), _ptr(stream)))
        sync(stream)   # the pack kernel and the all-gather were queued on `stream`; the copy below is not ordered after a non-blocking stream
        flat = comm._out.to_numpy()[: comm.world * self._batch * n.value]
        return flat.reshape(comm.world, self._batch, n.value)

    def gather_commitments_async(self, comm: "Comm", stream=None) -> int:
        """zk_cs_gather_commitments queued on `stream` without waiting (one per step inside a timed region); the gathered words stay
        in the communicator's device buffer, read them with Comm.gathered() after a synchronisation.  Returns n_public."""
        n = C.c_uint32(0)
        if comm._out is None or comm._out.n < comm.world * self._batch * 8:
            comm._out = DeviceBuffer(comm.world * self._batch * 8)
        _check(lib().zk_cs_gather_commitments(self._h, comm._h, _ptr(comm._out), C.byref(n), _ptr(stream)))
        comm._shape = (comm.world, self._batch, n.value)
        return n.value

    def debug_poke_store(self, loop_scope: bool, slot: int, lane: int, value: int):
        _check(lib().zk_cs_debug_poke_store(self._h, int(loop_scope), slot, lane, C.c_uint64(value)))

    def store_slots(self, loop_scope: bool) -> int:
        n = C.c_uint32(0)
        _check(lib().zk_cs_store_slots(self._h, int(loop_scope), C.byref(n)))
        return n.value

    def hook_vars(self, name: str):
        n = C.c_uint32(0)
        _check(lib().zk_circuit_hook_vars(self._h, name.encode(), None, 0, C.byref(n)))
        arr = (C.c_uint32 * n.value)()
        _check(lib().zk_circuit_hook_vars(self._h, name.encode(), arr, n.value, C.byref(n)))
        return list(arr)

    def hook_compare_witness(self, vars_, dev_expected, stream=None):
        """(True, None) when the circuit's values of `vars_` equal dev_expected [len(vars_), batch]; else (False, (instance, position))"""
        f = _Failure()
        arr = (C.c_uint32 * len(vars_))(*vars_)
        rc = lib().zk_cs_hook_compare_witness(self._h, arr, len(vars_), _ptr(dev_expected), _ptr(stream), C.byref(f))
        if rc == 0:
            return True, None
        if rc == -5:
            return False, (f.instance, f.slot)
        _check(rc)

    def seed_stream(self, n_instances: int, dev_outer_inputs, dev_loop_inputs, stream=None):
        """seed a stream of n_instances (any n, independent of set_batch): outer [word][n], loop [word][n * limit] (zk_cs_seed_stream)"""
        _check(lib().zk_cs_seed_stream(self._h, n_instances, _ptr(dev_outer_inputs), _ptr(dev_loop_inputs), _ptr(stream)))

    def seed_window_async(self, n_instances: int, dev_outer, n_outer_lanes: int, dev_loop, n_loop_lanes: int, first_instance: int, stream=None):
        """zk_cs_seed_window_async: seed instances [first, first + n) of a stream (outer [word][n_outer_lanes], loop [word][n_loop_lanes]);
        the kernels are queued on `stream`, the caller synchronises it before the window is read"""
        limit = self.stats()["limit"]
        po = C.c_void_p(_ptr(dev_outer).value + 8 * first_instance)
        pl = C.c_void_p(_ptr(dev_loop).value + 8 * first_instance * limit)
        _check(lib().zk_cs_seed_window_async(self._h, n_instances, po, C.c_uint64(n_outer_lanes), pl, C.c_uint64(n_loop_lanes), _ptr(stream)))

    def check_if_satisfied(self, stream=None):
        """Returns (True, None) or (False, Failure)."""
        f = _Failure()
        rc = lib().zk_cs_check_satisfied(self._h, _ptr(stream), C.byref(f))
        if rc == 0:
            return True, None
        if rc == ZK_ERR_UNSATISFIED:
            return False, Failure(f.scope, f.instance, f.iteration, f.slot, f.kind, f.relation)
        _check(rc)

    def set_check_mode(self, stored, defer_p2: bool = False):
        """zk_cs_set_check_mode: False = fused (default), True = every relation re-evaluated from the stored values;
        defer_p2=True: fused, and the Poseidon2 intermediates are written on demand only (ZK_CHECK_FUSED_DEFER_P2)"""
        _check(lib().zk_cs_set_check_mode(self._h, 2 if defer_p2 else (1 if stored else 0)))

    def narrow_byte_input_words(self):
        """zk_cs_narrow_byte_input_words: loop input words the narrow layout (ZKGL_NARROW_STORE=1) holds in one-byte slots"""
        n = C.c_size_t(0)
        _check(lib().zk_cs_narrow_byte_input_words(self._h, None, C.c_size_t(0), C.byref(n)))
        buf = (C.c_uint32 * max(n.value, 1))()
        _check(lib().zk_cs_narrow_byte_input_words(self._h, buf, C.c_size_t(n.value), C.byref(n)))
        return [int(buf[i]) for i in range(n.value)]

    def complete_store(self, stream=None):
        """zk_cs_complete_store: the deferred mode's fill of the Poseidon2 intermediates, on the caller's clock"""
        _check(lib().zk_cs_complete_store(self._h, _ptr(stream)))

    def resolve_and_check(self, stream=None):
        """fused witness generation + check_if_satisfied (outer scope overlapped on a second stream)"""
        f = _Failure()
        rc = lib().zk_cs_resolve_and_check(self._h, _ptr(stream), C.byref(f))
        if rc == 0:
            return True, None
        if rc == ZK_ERR_UNSATISFIED:
            return False, Failure(f.scope, f.instance, f.iteration, f.slot, f.kind, f.relation)
        _check(rc)

    def read_var(self, var: int, instance: int = 0, iteration: int = 0) -> int:
        out = C.c_uint64()
        _check(lib().zk_cs_read_var(self._h, var, instance, iteration, C.byref(out)))
        return out.value

    def write_cell(self, loop_scope: bool, cell: int, lane: int, value: int):
        _check(lib().zk_cs_write_cell(self._h, int(loop_scope), cell, lane, C.c_uint64(value)))

    def public_inputs(self, instance: int = 0):
        n = C.c_uint32()
        buf = (C.c_uint64 * 64)()
        _check(lib().zk_cs_public_inputs(self._h, instance, buf, 64, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def var_cell(self, var: int) -> int:
        c = C.c_uint32()
        _check(lib().zk_cs_var_cell(self._h, var, C.byref(c)))
        return c.value

    def public_cells(self):
        n = C.c_uint32()
        buf = (C.c_uint32 * 64)()
        _check(lib().zk_cs_public_cells(self._h, buf, 64, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def multiplicities(self, instance: int = 0) -> np.ndarray:
        n = C.c_uint32()
        _check(lib().zk_cs_multiplicities(self._h, instance, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        _check(lib().zk_cs_multiplicities(self._h, instance, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def stats(self) -> dict:
        s = _Stats()
        _check(lib().zk_cs_stats(self._h, C.byref(s)))
        d = {f[0]: getattr(s, f[0]) for f in _Stats._fields_ if f[0] != "gate_instances"}
        d["gate_instances"] = {GATE_NAMES[i]: s.gate_instances[i] for i in range(len(GATE))}
        return d

    print_gate_stats = stats

    def last_ms(self, which: int) -> float:
        ms = C.c_float()
        _check(lib().zk_cs_last_ms(self._h, which, C.byref(ms)))
        return ms.value

    def export(self, loop_scope: bool) -> np.ndarray:
        n = C.c_size_t()
        _check(lib().zk_cs_export(self._h, int(loop_scope), None, 0, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint32)
        _check(lib().zk_cs_export(self._h, int(loop_scope), buf.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return buf

    def copy_permutation(self, beta, gamma, z_out=None, stream=None):
        """K12: -> (n_mismatch, array [batch, 4] = numerator (a, b), denominator (a, b) of z[rows]); z_out: device buffer
        [batch][rows + 1][2] receiving the column z"""
        b = (C.c_uint64 * 2)(*beta)
        g = (C.c_uint64 * 2)(*gamma)
        out = np.zeros((self.batch, 4), dtype=np.uint64)
        bad = C.c_uint32()
        _check(lib().zk_cs_copy_permutation(self._h, b, g, _ptr(stream), _ptr(z_out), out.ctypes.data_as(C.c_void_p), C.c_uint32(self.batch),
                                            C.byref(bad)))
        return bad.value, out

    def sigma(self, loop_scope: bool, iteration: int = 0) -> np.ndarray:
        n = C.c_size_t()
        _check(lib().zk_cs_sigma(self._h, int(loop_scope), C.c_uint32(iteration), None, 0, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint64)
        _check(lib().zk_cs_sigma(self._h, int(loop_scope), C.c_uint32(iteration), buf.ctypes.data_as(C.c_void_p), C.c_size_t(n.value), C.byref(n)))
        return buf

    def trace_columns(self, instance: int, out, log_n: int, stride=None, stream=None):
        """K11 input: out[col * stride + row] = the instance's trace columns (loop rows, then outer rows, zero padded to 2^log_n)"""
        stride = (1 << log_n) if stride is None else stride
        _check(lib().zk_cs_trace_columns(self._h, C.c_uint32(instance), _ptr(out), C.c_uint32(log_n), C.c_uint64(stride), _ptr(stream)))

    def trace_columns_batch(self, first_instance: int, n_instances: int, out, log_n: int, n_cols: int, stride=None, instance_stride=None, stream=None):
        """zk_cs_trace_columns_batch: out[(i - first) * instance_stride + col * stride + row] for n_instances instances in one pass"""
        stride = (1 << log_n) if stride is None else stride
        instance_stride = n_cols * stride if instance_stride is None else instance_stride
        _check(lib().zk_cs_trace_columns_batch(self._h, C.c_uint32(first_instance), C.c_uint32(n_instances), _ptr(out), C.c_uint32(log_n), C.c_uint64(stride),
                                               C.c_uint64(instance_stride), _ptr(stream)))

    def trace(self, loop_scope: bool) -> np.ndarray:
        """Copy the scope's cells back as the logical array [n_cells, n_tiles*64] (cell-major, lane-minor).
        Device storage is wave-tiled [tile][cell][64]; cell = slot * n_columns + column."""
        p, n, s = C.c_void_p(), C.c_uint64(), C.c_uint64()
        _check(lib().zk_cs_trace_ptr(self._h, int(loop_scope), C.byref(p), C.byref(n), C.byref(s)))
        out = np.empty(n.value * s.value, dtype=np.uint64)
        if out.size:
            _check(lib().zk_d2h(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes), None))
        tiles = s.value // 64
        return np.ascontiguousarray(out.reshape(tiles, n.value, 64).transpose(1, 0, 2)).reshape(n.value, s.value)
