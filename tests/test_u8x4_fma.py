"""a2: U8x4FMAGate as its own gate kind (ZK_GATE_U8X4_FMA) + its witness op (ZK_OP_U8X4FMA) — UInt32::fma_with_carry in the form
enforce_mul_relation requires (/root/reference/src/main_vm/opcodes/mod.rs:146-158): a*b + c + d = lo + 2^32 hi over little-endian
bytes, two relations that cannot wrap the field, carry bytes range-checked.  Oracle on CPU; device == oracle and fault injection
under -m gpu.  The one-relation u32 gate it replaces accepts lo + 2^32 hi = a*b + c + d - p for large products — shown below."""
import numpy as np
import pytest

import zkgl
from helpers import Rec
from oracle import zko
from zkgl import GATE as G, OP

P = zko.P


def build():
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4))
    cs.allow_lookup(3, 8, True)
    for k in ("CONST", "FMA", "REDUCTION4", "U8X4_FMA", "PUBLIC_INPUT"):
        cs.allow_gate(G[k])
    xor8 = np.array([[a, b, a ^ b] for a in range(256) for b in range(256)], dtype=np.uint64)
    t = cs.add_lookup_table(1, 2, 1, xor8)
    r = Rec(cs)
    words = [r.inp() for _ in range(4)]                    # a, b, c, d as u32 inputs
    byts = []
    for w in words:
        b = r.split(w, 4, 8, [1, 1 << 8, 1 << 16, 1 << 24])
        cs.perform_lookup(t, [b[0], b[1]], 1); cs.perform_lookup(t, [b[2], b[3]], 1)
        byts += b
    outs = cs.alloc_multiple_variables_without_values(10)
    cs.emit_op(OP["U8X4FMA"], byts, outs)
    cs.place_gate(G["U8X4_FMA"], byts + outs)
    for i in range(0, 10, 2):
        cs.perform_lookup(t, [outs[i], outs[i + 1]], 1)
    lo = r.lc4(outs[0:4], [1, 1 << 8, 1 << 16, 1 << 24])
    hi = r.lc4(outs[4:8], [1, 1 << 8, 1 << 16, 1 << 24])
    for v in (lo, hi):
        cs.place_gate(G["PUBLIC_INPUT"], [v])
    cs.pad_and_shrink()
    return cs, outs


def inputs(B):
    rng = np.random.default_rng(11)
    w = rng.integers(0, 1 << 32, size=(4, B), dtype=np.uint64)
    w[:, 0] = (1 << 32) - 1                     # the largest case: (2^32-1)^2 + 2 (2^32-1) = 2^64 - 1 >= p
    w[:, 1] = 0
    return w


def test_u8x4_fma_on_the_oracle():
    cs, outs = build()
    assert cs.stats()["gate_instances"]["U8X4_FMA"] == 1
    B = 40
    inp = inputs(B)
    run = zko.CircuitRun(cs.export(False), cs.export(True), B, 65536)
    run.resolve(inp, np.zeros((0, 0), dtype=np.uint64))
    bad, nrel = run.check()
    assert bad == 0 and nrel == B * cs.stats()["constraints_per_instance"]
    pub = cs.public_cells()
    for i in range(B):
        a, b, c, d = (int(x) for x in inp[:, i])
        v = a * b + c + d
        assert (int(run.oc[pub[0], i]), int(run.oc[pub[1], i])) == (v & 0xffffffff, v >> 32)
    # a wrong result that the ONE-relation gate of rounds 1-3 would accept: lo + 2^32 hi = a*b + c + d - p (mod p the same value)
    a, b, c, d = (int(x) for x in inp[:, 0])
    v = a * b + c + d
    assert v >= P
    forged = v - P
    lo_cell0 = cs.var_cell(outs[0])
    for k in range(8):
        run.oc[cs.var_cell(outs[k]), 0] = (forged >> (8 * k)) & 0xff
    assert (forged & 0xffffffff) + (forged >> 32 << 32) == forged and (a * b + c + d - forged) % P == 0
    assert run.check()[0] > 0                    # rejected: the two byte relations hold over the integers
    assert lo_cell0 >= 0


@pytest.mark.gpu
def test_u8x4_fma_on_the_gpu(zk):
    cs, outs = build()
    B = 200
    inp = inputs(B)
    cs.set_batch(B)
    d = zk.DeviceBuffer.from_numpy(inp)
    cs.bind_inputs(False, d, 4)
    for stored in (False, True):
        cs.set_check_mode(stored)
        ok, f = cs.resolve_and_check()
        assert ok, f
    cs.set_check_mode(False)
    run = zko.CircuitRun(cs.export(False), cs.export(True), B, 65536)
    run.resolve(inp, np.zeros((0, 0), dtype=np.uint64))
    assert np.array_equal(cs.trace(False), run.oc)
    for i in (0, 1, 2, B - 1):
        a, b, c, dd = (int(x) for x in inp[:, i])
        v = a * b + c + dd
        assert cs.public_inputs(i) == [v & 0xffffffff, v >> 32]
    # forged bytes (value - p) in instance 0 (every cell of the trace is then checked as stored): rejected
    a, b, c, dd = (int(x) for x in inp[:, 0])
    forged = a * b + c + dd - P
    for k in range(8):
        cs.write_cell(False, cs.var_cell(outs[k]), 0, (forged >> (8 * k)) & 0xff)
    ok, f = cs.check_if_satisfied()
    assert not ok and f.instance == 0, f
    # the strand form of the same program
    import os
    os.environ["ZKGL_STRANDS"] = "1"
    try:
        ok, f = cs.resolve_and_check()
        assert ok, f
        assert np.array_equal(cs.trace(False), run.oc)
    finally:
        del os.environ["ZKGL_STRANDS"]
