"""Host logic: the scalar-decoded device programs (plain form `prog2`, strand forms) of every circuit, walked at finalize by an
independent decoder that knows the op layouts as the KERNELS read them (csrc/cs.cpp verify_device_programs, ZKGL_VERIFY_DEVICE_PROGRAMS=1):
every op once, operand words, output slots, strand levels, flag planes written before they are read — in the default emission and in
every opt-in form (strand-form planes, the ByteBuffer macro-op, hash macro-ops off, planes off, mux-chain order).  No GPU."""
import os

import pytest

import zkgl

G100 = zkgl.CSGeometry(100, 0, 8, 4)


def _rec(configure, entry, geometry=G100):
    cs = zkgl.ConstraintSystem(geometry, 1 << 24, 1 << 28)
    configure(cs)
    entry(cs)
    cs.pad_and_shrink()
    cs.close()


CIRCUITS = {
    "ram_permutation": lambda: _rec(lambda c: c.configure_ram_permutation(), lambda c: c.ram_permutation_entry_point(6)),
    "storage_validity": lambda: _rec(lambda c: c.configure_storage_validity(), lambda c: c.sort_and_deduplicate_storage_access_entry_point(5, True)),
    "log_sorter": lambda: _rec(lambda c: c.configure_log_sorter(), lambda c: c.sort_and_deduplicate_events_entry_point(5)),
    "keccak_fsm": lambda: _rec(lambda c: c.configure_keccak(), lambda c: c.keccak256_round_function_entry_point(2)),
    "sha256_fsm": lambda: _rec(lambda c: c.configure_sha256(), lambda c: c.sha256_round_function_entry_point(3)),
    "eip_4844": lambda: _rec(lambda c: c.configure_eip_4844(), lambda c: c.eip_4844_entry_point(27)),
    "demux": lambda: _rec(lambda c: c.configure_demux_log_queue(), lambda c: c.demultiplex_storage_logs_entry_point(4)),
    "sort_decommits": lambda: _rec(lambda c: c.configure_sort_decommits(), lambda c: c.sort_and_deduplicate_code_decommittments_entry_point(4)),
    "code_unpacker": lambda: _rec(lambda c: c.configure_code_unpacker(), lambda c: c.unpack_code_into_memory_entry_point(3)),
    "linear_hasher": lambda: _rec(lambda c: c.configure_linear_hasher(), lambda c: c.linear_hasher_entry_point(17)),
    "sha256_fsm_reference_tables": lambda: _rec(lambda c: c.configure_sha256(True), lambda c: c.sha256_round_function_entry_point(3)),
    "code_unpacker_reference_tables": lambda: _rec(lambda c: c.configure_sha256(True), lambda c: c.unpack_code_into_memory_entry_point(3)),
    "main_vm": lambda: _rec(lambda c: c.configure_main_vm(), lambda c: c.main_vm_entry_point(3), zkgl.CSGeometry(140, 0, 8, 8)),
    "vm_shaped": lambda: _rec(lambda c: c.configure_vm_shaped(), lambda c: c.vm_shaped_entry_point(4), zkgl.CSGeometry(140, 0, 8, 8)),
}
FORMS = {
    "default": {},
    "planes_off": {"ZKGL_FLAG_PLANES": "0"},
    "no_hash_macros": {"ZKGL_NO_HASH_MACROS": "1"},
    "bytebuf_macro": {"ZKGL_BYTEBUF_MACRO": "1"},
    "sha4_op_by_op": {"ZKGL_SHA4_MACRO": "0"},          # (the macro-op is the default recording of the reference's table set)
}


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("circuit", sorted(CIRCUITS))
def test_device_programs_decode_to_the_recorded_ops(monkeypatch, circuit, form):
    if form == "bytebuf_macro" and circuit != "keccak_fsm":
        pytest.skip("the ByteBuffer is the keccak precompile's")
    if form == "sha4_op_by_op" and not circuit.endswith("reference_tables"):
        pytest.skip("the 4-bit-chunk decomposition belongs to the reference's table set")
    if form == "no_hash_macros" and circuit not in ("keccak_fsm", "sha256_fsm", "eip_4844", "code_unpacker", "linear_hasher"):
        pytest.skip("no hash gadget")
    for k in ("ZKGL_FLAG_PLANES", "ZKGL_NO_HASH_MACROS", "ZKGL_BYTEBUF_MACRO", "ZKGL_VERIFY_SABOTAGE", "ZKGL_SHA4_MACRO"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("ZKGL_VERIFY_DEVICE_PROGRAMS", "1")
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    CIRCUITS[circuit]()


def test_the_check_notices_a_changed_program_word(monkeypatch):
    """the decoder is not vacuous: one flipped bit anywhere in the first words of main_vm's loop program is a mismatch"""
    monkeypatch.setenv("ZKGL_VERIFY_DEVICE_PROGRAMS", "1")
    caught = 0
    for at in (0, 1, 5, 40, 200, 1001, 1400):
        monkeypatch.setenv("ZKGL_VERIFY_SABOTAGE", str(at))
        try:
            CIRCUITS["ram_permutation"]()
        except zkgl.ZkError:
            caught += 1
    assert caught == 7


@pytest.mark.parametrize("form", ["default", "planes_off"])
def test_random_programs_decode_to_their_ops(monkeypatch, form):
    """the fuzz circuits of tests/test_fuzz_programs.py (random mixes of every light op kind over an outer and a loop scope)"""
    from test_fuzz_programs import random_circuit
    for k in ("ZKGL_FLAG_PLANES", "ZKGL_VERIFY_SABOTAGE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("ZKGL_VERIFY_DEVICE_PROGRAMS", "1")
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    for seed in range(30, 60):
        cs = random_circuit(seed)[0]
        cs.close()
