"""a16 on the fast path: the SHA-256 compression over the REFERENCE's table set (Maj4 / TriXor4 / Ch4 / Split4BitChunk<1,2>,
/root/reference/src/code_unpacker_sha256/mod.rs:554-566) as a macro-op — ZK_OP_SHA256_ROUNDS with a = 1, the default recording of
configure_sha256(reference_tables=True) since round 6 (ZKGL_SHA4_MACRO=0 records op by op).  One structure (csrc/sha256_macro4.hpp) is walked by the host gadget, the device op and the counting backend; the
oracle restates it in C.  The macro recording must be THE SAME circuit as the op-by-op recording — same variables, gates, cells — and
the oracle's restatement must write the same value into every cell; digests equal hashlib.  The device backend is host-compilable and
is walked on the CPU against the recorded gates' arithmetic.  Device parity under -m gpu: tests/test_sha256_reference_tables.py, tests/test_zz_round5_gpu.py
(its kernels: k_witness_strands2<.., X_SHA4> / k_witness_plain_x<X_SHA4>, in the one library)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import zkgl
from oracle import zko
from test_sha256_host import loop_stream

REF_TABLE_ROWS = 3 * 4096 + 2 * 16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(monkeypatch, macro, entry="blocks", n=2):
    if macro:
        monkeypatch.delenv("ZKGL_SHA4_MACRO", raising=False)   # the default recording since round 6
    else:
        monkeypatch.setenv("ZKGL_SHA4_MACRO", "0")             # op by op
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_sha256(reference_tables=True)
    {"blocks": cs.sha256_blocks_entry_point, "fsm": cs.sha256_round_function_entry_point, "unpacker": cs.unpack_code_into_memory_entry_point}[entry](n)
    cs.pad_and_shrink()
    return cs


def test_macro_recording_is_the_same_circuit_and_the_oracle_fills_the_same_cells(monkeypatch):
    plain, macro = record(monkeypatch, False), record(monkeypatch, True)
    sp, sm = plain.stats(), macro.stats()
    for k in ("rows_per_instance", "constraints_per_instance", "loop_slots", "outer_slots", "gate_instances", "lookups_per_instance", "variables_loop",
              "cells_populated_loop"):
        assert sp[k] == sm[k], k
    assert sm["loop_ops"] < 300 < 15000 < sp["loop_ops"]          # ~15.7 k ops of a compression became one
    # the macro-op evaluates its gadget's gates where it produces their values (the 2^32 carry + low == sum enforcements included: both sides
    # are its own integers); what is left to the check program: the nibble recompositions of the INPUT bytes and the circuit's own gates
    assert sm["constraints_from_store_fused"] <= sp["constraints_from_store_fused"]
    assert sm["constraints_from_store_fused"] + sm["constraints_in_witness_fused"] == sm["constraints_per_instance"]
    rng = np.random.default_rng(44)
    msgs = [b"abc" * 20, bytes(rng.integers(0, 256, size=64, dtype=np.uint8)), bytes(rng.integers(0, 256, size=119, dtype=np.uint8)), bytes(56)]
    n_blocks = 2          # 56 .. 119 bytes pad to exactly two blocks
    outer = np.zeros((0, len(msgs)), dtype=np.uint64)
    raw = loop_stream(msgs, n_blocks)
    runs = []
    for cs in (plain, macro):
        seeded = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS).seed(outer, raw)
        r = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS)
        r.resolve(outer, seeded)
        bad, nrel = r.check()
        assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * len(msgs)
        runs.append(r)
    assert np.array_equal(runs[0].oc, runs[1].oc) and np.array_equal(runs[0].lc, runs[1].lc)
    for i, m in enumerate(msgs):
        assert bytes(int(runs[1].oc[c, i]) for c in macro.public_cells()) == hashlib.sha256(m).digest()
    # a wrong stored value is a violated relation for the oracle checker in the macro recording too (the relations are the same gates)
    cell = int(np.flatnonzero(runs[1].lc[:, 1] > 1)[100])          # a populated cell of lane 1
    runs[1].lc[cell, 1] ^= 1
    bad, _ = runs[1].check()
    assert bad > 0


@pytest.mark.parametrize("entry,n", [("fsm", 3), ("unpacker", 3)])
def test_the_fsm_circuits_record_the_macro_op_too(monkeypatch, entry, n):
    plain, macro = record(monkeypatch, False, entry, n), record(monkeypatch, True, entry, n)
    sp, sm = plain.stats(), macro.stats()
    for k in ("rows_per_instance", "constraints_per_instance", "gate_instances", "lookups_per_instance", "variables_loop"):
        assert sp[k] == sm[k], k
    assert sm["loop_ops"] < sp["loop_ops"] - 15000
    assert sm["seed_ops"] == sp["seed_ops"]                      # the seed hint keeps the decomposition out of the cone in both recordings


def test_device_backend_walk_equals_the_oracle_restatement(tmp_path):
    """zks4::ComputeBackend — the device's uint32 form of the walk — is host-compilable: walked on the CPU, its output stream is compared value
    by value with the oracle's plain-C restatement (sh4_compress, itself pinned to the recorded gates by the test above), random states and blocks"""
    exe = str(tmp_path / "sha4check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "sha4_compute_check.cpp")], check=True)
    out = subprocess.run([exe, "200"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok 200 trials"), out            # output count == the counting backend's, final state == a software compression
    rng = np.random.default_rng(46)
    for trial in range(6):
        st = [int(x) for x in rng.integers(0, 1 << 32, size=8)] if trial else [0xffffffff] * 8
        blk = [int(x) for x in rng.integers(0, 1 << 32, size=16)] if trial else [0] * 16
        lines = subprocess.run([exe, "stream"] + [f"{x:x}" for x in st + blk], check=True, capture_output=True, text=True).stdout.splitlines()
        got = np.array([int(x, 16) for x in lines[1].split()], dtype=np.uint64)
        want, final = zko.sha256_rounds_stream(1, st, blk)
        assert int(lines[0]) == got.size == want.size
        assert np.array_equal(got, want), int(np.flatnonzero(got != want)[0])
        assert [int(x, 16) for x in lines[2].split()] == final


# (the device half of this file: tests/test_zz_round5_gpu.py)
