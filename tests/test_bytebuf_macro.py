"""K8, third macro-op: ByteBuffer::fill_with_bytes as ZK_OP_BYTEBUF_FILL (opt-in at record time: ZKGL_BYTEBUF_MACRO=1).  The keccak256
precompile FSM recorded with the macro-op must be THE SAME circuit as the op-by-op recording — same variables, same gates, same cells —
and the oracle's restatement of the macro-op must write the same value into every cell (reference cases of
/root/reference/src/keccak256_round_function/mod.rs:1096-1144, all ten in one batch).  Device parity under -m gpu."""
import os

import numpy as np
import pytest

import zkgl
from oracle import keccak_native as N
from oracle import zko
from test_keccak_fsm_host import REFERENCE_CASES, TABLE_ROWS, reference_case, streams


def record(monkeypatch, macro, limit=2):
    if macro:
        monkeypatch.setenv("ZKGL_BYTEBUF_MACRO", "1")
    else:
        monkeypatch.delenv("ZKGL_BYTEBUF_MACRO", raising=False)
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_keccak()
    cs.keccak256_round_function_entry_point(limit)
    cs.pad_and_shrink()
    return cs


def test_macro_recording_is_the_same_circuit_and_the_oracle_fills_the_same_cells(monkeypatch):
    plain, macro = record(monkeypatch, False), record(monkeypatch, True)
    sp, sm = plain.stats(), macro.stats()
    for k in ("rows_per_instance", "constraints_per_instance", "loop_slots", "outer_slots", "gate_instances"):
        assert sp[k] == sm[k], k
    assert sm["loop_ops"] < sp["loop_ops"] - 6 * 7000          # six fills of ~7.7 k ops each became six ops
    # the fused mode evaluates the same relations where their values are produced in both recordings: the macro-op mirrors exactly the
    # FMA / Selection / ZeroCheck gates its op-by-op form mirrors, nothing moves into or out of the check program
    for k in ("constraints_from_store_fused", "constraints_in_witness_fused"):
        assert sp[k] == sm[k], (k, sp[k], sm[k])
    insts = [reference_case(l, u)[1] for l, u in REFERENCE_CASES]
    outer, loop = streams(insts, 2)
    runs = []
    for cs in (plain, macro):
        r = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS)
        r.resolve(outer, loop)
        bad, nrel = r.check()
        assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * len(insts)
        runs.append(r)
    assert np.array_equal(runs[0].oc, runs[1].oc) and np.array_equal(runs[0].lc, runs[1].lc)
    for i, inst in enumerate(insts):
        assert [int(runs[1].oc[c, i]) for c in macro.public_cells()] == inst["public_input"]


def test_device_backend_walk_equals_the_gate_arithmetic_on_the_cpu(tmp_path):
    """zkb::ComputeBackend — the device's register-packed, sliding form of the four byte arrays — is host-compilable: walked on the CPU
    against a plain field-element backend (the gates' own arithmetic), every emitted value and the final buffer, random in-range operands"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "bbcheck")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "native", "bytebuf_compute_check.cpp")], check=True)
    out = subprocess.run([exe, "300"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok 300 trials"), out


# (the device half of this file: tests/test_zz_round5_gpu.py — device paths that have not run on a GPU yet sort last)
