"""Hand derivation of the bincode 1.x bytes of ONE `RamPermutationCircuitInstanceWitness` — the reference's own test fixture
(/root/reference/src/ram_permutation/mod.rs:559-634: three MemoryQuery) — written field by field in the order serde's derive walks
the structs, every group of bytes annotated with the rule and the reference line that produces it.  Output:

    tests/golden/ram_witness_bincode_vector.hex    the bytes (hex)
    tests/golden/ram_witness_bincode_vector.md     offset | bytes | field | rule

There is no rustc in this image, so these bytes are NOT produced by the reference; the table is what makes the assumptions
reviewable: a maintainer with the crate can `bincode::serialize` the same witness and diff.  Rules used (serde + bincode 1.x defaults:
little-endian, fixed-width integers, u64 lengths):
  R1 struct            = its fields in declaration order, nothing in between
  R2 bool              = 1 byte 0 / 1;  u8 = 1 byte;  u32 = 4 bytes LE;  u64 = 8 bytes LE
  R3 [T; N] (N <= 32)  = the N elements, no length (serde's array impl = tuple); boojum's BigArraySerde is used only for N > 32
  R4 F (GoldilocksField) = its canonical u64, 8 bytes LE                                   [EXT boojum field serde]
  R5 U256 (ethereum-types + impl-serde) = a string: u64 length, then "0x" + lowercase hex without leading zeros ("0x0" for 0) [EXT]
  R6 VecDeque<T> / Vec<T> = u64 element count, then the elements
  R7 (A, B) tuple      = A then B
  R8 ()                = nothing (observable_output of ram_permutation)
  R9 QueueStateWitness<F, 12> = head [F; 12], tail: QueueTailStateWitness { tail [F; 12], length u32 }   [EXT boojum gadgets::queue]
  R10 FullStateCircuitQueueRawWitness { elements: VecDeque<(MemoryQueryWitness, [F; 12])> }   (src/ram_permutation/input.rs:103-116;
      element shape visible at src/keccak256_round_function/mod.rs:1072-1082) — the [F; 12] is the queue state BEFORE that push [EXT]
The previous-tail states are computed with the oracle's Poseidon2 (full-state queue push, src/main_vm/utils.rs:194-213).

    python tests/golden/make_ram_bincode_vector.py
"""
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ram_native as rn  # noqa: E402
from oracle import zko  # noqa: E402

rows, out = [], bytearray()


def emit(b, field, rule):
    rows.append((len(out), bytes(b), field, rule))
    out.extend(b)


def u8(v, f, r="R2 u8 / bool"): emit(struct.pack("<B", v), f, r)
def u32(v, f): emit(struct.pack("<I", v), f, "R2 u32 LE")
def u64(v, f, r="R2 u64 LE"): emit(struct.pack("<Q", v), f, r)
def felt(v, f): emit(struct.pack("<Q", v), f, "R4 field element = canonical u64 LE")


def u256(v, f):
    s = ("0x%x" % v).encode()
    emit(struct.pack("<Q", len(s)), f + " (string length)", "R5 U256 as a string: u64 length")
    emit(s, f + " (hex digits)", "R5 \"0x\" + hex without leading zeros")


def queue_state(head, tail, length, f):
    for i, x in enumerate(head): felt(x, f"{f}.head[{i}]")
    for i, x in enumerate(tail): felt(x, f"{f}.tail.tail[{i}]")
    u32(length, f"{f}.tail.length")


def fsm(f):
    for k in ("lhs_accumulator", "rhs_accumulator"):
        for i in range(2): felt(0, f"{f}.{k}[{i}]")
    queue_state([0] * 12, [0] * 12, 0, f"{f}.current_unsorted_queue_state")
    queue_state([0] * 12, [0] * 12, 0, f"{f}.current_sorted_queue_state")
    for i in range(3): u32(0, f"{f}.previous_sorting_key[{i}]")
    for i in range(2): u32(0, f"{f}.previous_full_key[{i}]")
    u256(0, f"{f}.previous_value")
    u8(0, f"{f}.previous_is_ptr")
    u32(0, f"{f}.num_nondeterministic_writes")


def main():
    fx = json.load(open(os.path.join(HERE, "ram_fixture.json")))

    def rec(row):
        d = dict(zip(fx["fields"], row))
        if d["memory_page"] == "BOOTLOADER_HEAP_PAGE":
            d["memory_page"] = rn.BOOTLOADER_HEAP_PAGE      # zkevm_opcode_defs constant [EXT]
        return d
    unsorted, sorted_ = [rec(r) for r in fx["unsorted"]], [rec(r) for r in fx["sorted"]]

    def chain(items):   # previous tail of every push + the final tail
        tail, prev = [0] * 12, []
        for q in items:
            prev.append(list(tail))
            enc = zko.memory_query_encode(rn.mq(q["timestamp"], q["memory_page"], q["index"], q["rw_flag"], q["is_ptr"], q["value"]))
            tail = zko.poseidon2_permute(list(enc) + tail[8:12])
        return prev, tail

    pu, tu = chain(unsorted)
    ps, ts = chain(sorted_)
    # ---- closed_form_input: ClosedFormInputWitness (src/fsm_input_output/mod.rs:42-47)
    u8(1, "closed_form_input.start_flag")
    u8(0, "closed_form_input.completion_flag")
    queue_state([0] * 12, tu, len(unsorted), "closed_form_input.observable_input.unsorted_queue_initial_state")      # input.rs:28-32
    queue_state([0] * 12, ts, len(sorted_), "closed_form_input.observable_input.sorted_queue_initial_state")
    u32(0, "closed_form_input.observable_input.non_deterministic_bootloader_memory_snapshot_length")
    rows.append((len(out), b"", "closed_form_input.observable_output", "R8 (): no bytes"))
    fsm("closed_form_input.hidden_fsm_input")                                                                            # input.rs:49-62
    fsm("closed_form_input.hidden_fsm_output")
    # ---- the two queue witnesses (input.rs:103-116)
    for name, items, prev in (("unsorted_queue_witness", unsorted, pu), ("sorted_queue_witness", sorted_, ps)):
        u64(len(items), f"{name}.elements (count)", "R6 VecDeque: u64 element count")
        for k, q in enumerate(items):
            f = f"{name}.elements[{k}]"
            u32(q["timestamp"], f + ".0.timestamp")           # MemoryQueryWitness, src/base_structures/memory_query/mod.rs:30-37
            u32(q["memory_page"], f + ".0.memory_page")
            u32(q["index"], f + ".0.index")
            u8(q["rw_flag"], f + ".0.rw_flag")
            u8(q["is_ptr"], f + ".0.is_ptr")
            u256(q["value"], f + ".0.value")
            for i, x in enumerate(prev[k]): felt(x, f"{f}.1[{i}] (queue tail before this push)")
    open(os.path.join(HERE, "ram_witness_bincode_vector.hex"), "w").write(bytes(out).hex() + "\n")
    with open(os.path.join(HERE, "ram_witness_bincode_vector.md"), "w") as md:
        md.write("# bincode bytes of the reference's ram_permutation fixture as `RamPermutationCircuitInstanceWitness`\n\n")
        md.write("Derived by hand (tests/golden/make_ram_bincode_vector.py, rules R1-R10 in its header); NOT produced by the reference crate.\n\n")
        md.write(f"{len(out)} bytes.\n\n| offset | bytes (hex) | field | rule |\n|---|---|---|---|\n")
        for off, b, field, rule in rows:
            md.write(f"| {off} | `{b.hex() or '-'}` | {field} | {rule} |\n")
    print("wrote", len(out), "bytes,", len(rows), "fields")


if __name__ == "__main__":
    main()
