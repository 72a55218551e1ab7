"""a19: eip_4844_entry_point (/root/reference/src/eip_4844/mod.rs:107-260) recorded through the C-ABI and executed on the
CPU oracle interpreter, plus the big-integer witness op (ZK_OP_NN_MULMOD) and stream links it relies on.  The property is
the one the reference's test checks (mod.rs:595-683): the in-circuit linear hash / opening value / output hash equal
keccak256 and a native BLS12-381-scalar evaluation (Python big integers here)."""
import numpy as np
import pytest

import zkgl
from helpers import Rec
from oracle import eip4844_native as N
from oracle import zko
from zkgl import GATE as G, OP

TABLE_ROWS = 65536 * 2 + 7 * 256
_CS = {}


def blob_cs(n_chunks):
    if n_chunks not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4))
        cs.configure_eip_4844()
        cs.eip_4844_entry_point(n_chunks)
        cs.pad_and_shrink()
        _CS[n_chunks] = cs
    return _CS[n_chunks]


def make_instances(n_chunks, seeds):
    insts = []
    for s in seeds:
        rng = np.random.default_rng(0xC5 + s)
        blob = bytes(rng.integers(0, 256, size=31 * n_chunks, dtype=np.uint8))
        vh = b"\x01" + bytes(rng.integers(0, 256, size=31, dtype=np.uint8))
        insts.append(N.instance(blob, vh, n_chunks))
    return insts


def streams(insts):
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    return outer, loop


def run(cs, outer, loop, batch):
    r = zko.CircuitRun(cs.export(False), cs.export(True), batch, TABLE_ROWS)
    r.resolve(outer, loop)
    return r


@pytest.mark.parametrize("n_chunks", [4, 16, 27])   # 1, 4 and 7 Keccak blocks; 27 chunks: last iteration has inactive Horner steps
def test_eip4844_matches_native_bigint_and_keccak(n_chunks):
    cs = blob_cs(n_chunks)
    n_bytes, n_blocks, cpi = N.shape(n_chunks)
    assert cs.input_words() == (64, 217 + 136 + 31 * cpi) and cs.stats()["limit"] == n_blocks
    insts = make_instances(n_chunks, range(3))
    outer, loop = streams(insts)
    blank = loop.copy()
    blank[:217, :] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS).seed(outer, blank)
    assert np.array_equal(seeded, loop), "generic seeding differs from the native Horner / sponge trajectory"
    r = run(cs, outer, loop, len(insts))
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * len(insts)
    for i, inst in enumerate(insts):
        assert [int(r.oc[c, i]) for c in cs.public_cells()] == inst["public_input"]


def test_eip4844_negative():
    n_chunks = 16
    cs = blob_cs(n_chunks)
    inst = make_instances(n_chunks, [9])[0]
    outer, loop = streams([inst])
    assert run(cs, outer, loop, 1).check()[0] == 0
    bad = outer.copy(); bad[40, 0] ^= 1                       # linear_hash_output differs from keccak256(blob)
    assert run(cs, bad, loop, 1).check()[0] > 0
    bad = loop.copy(); bad[217 + 5, 0] ^= 1                   # block view of blob byte 5 differs from its chunk view (stream link)
    assert run(cs, outer, bad, 1).check()[0] > 0
    bad = loop.copy(); bad[200 + 3, 2] = (int(bad[200 + 3, 2]) + 1) % 65536   # carried opening limb differs from the previous output
    assert run(cs, outer, bad, 1).check()[0] > 0


def bigint_ops_cs():
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(30, 0, 4, 4))
    for k in ("CONST", "FMA"):
        cs.allow_gate(G[k])
    r = Rec(cs)
    a = [r.inp() for _ in range(16)]
    b = [r.inp() for _ in range(8)]
    x = r.inp()
    outs = cs.alloc_multiple_variables_without_values(9 + 16)
    cs.emit_op(OP["NN_MULMOD"], a + b, outs, N.limbs16(N.BLS_FR), a=16, b=8)
    qr = cs.alloc_multiple_variables_without_values(2)
    cs.emit_op(OP["DIVREM"], [x], qr, b=136)
    cs.pad_and_shrink()
    return cs, outs, qr


def bigint_ops_inputs(B=64):
    rng = np.random.default_rng(19)
    inp = np.zeros((25, B), dtype=np.uint64)
    inp[:16] = rng.integers(0, 1 << 17, size=(16, B))
    inp[16:24] = rng.integers(0, 1 << 16, size=(8, B))
    inp[:24, 0] = 0
    inp[:16, 1] = (1 << 17) - 1; inp[16:24, 1] = 65535        # extremes
    inp[:16, 2] = N.limbs16(N.BLS_FR); inp[16:24, 2] = 0; inp[16, 2] = 1   # A = M, B = 1 -> q = 1, r = 0
    inp[24] = rng.integers(0, 1 << 32, size=B)
    return inp


def bigint_ops_expected(inp, i):
    A = sum(int(inp[k, i]) << (16 * k) for k in range(16))
    Bv = sum(int(inp[16 + k, i]) << (16 * k) for k in range(8))
    return N.limbs16(A * Bv // N.BLS_FR, 9) + N.limbs16(A * Bv % N.BLS_FR), [int(inp[24, i]) // 136, int(inp[24, i]) % 136]


def test_nn_mulmod_and_divrem_ops_against_python_bigints():
    """the two witness ops added for a17/a19, on the oracle interpreter: random operands incl. lazy (17-bit) limbs"""
    cs, outs, qr = bigint_ops_cs()
    inp = bigint_ops_inputs()
    B = inp.shape[1]
    run_ = zko.CircuitRun(cs.export(False), cs.export(True), B, 0)
    run_.resolve(inp, np.zeros((0, 0), dtype=np.uint64))
    for i in range(B):
        e_nn, e_dr = bigint_ops_expected(inp, i)
        assert [int(run_.oc[cs.var_cell(v), i]) for v in outs] == e_nn
        assert [int(run_.oc[cs.var_cell(v), i]) for v in qr] == e_dr


def test_eip4844_full_size_blob():
    """the reference's size: 4096 chunks = 126 976 bytes = 934 Keccak blocks (src/eip_4844/input.rs:25-26), one blob"""
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4), 1 << 21, 1 << 28)
    cs.configure_eip_4844()
    cs.eip_4844_entry_point(4096)
    cs.pad_and_shrink()
    assert cs.stats()["limit"] == 934 and cs.input_words() == (64, 508)
    inst = make_instances(4096, [1])[0]
    outer, loop = streams([inst])
    r = run(cs, outer, loop, 1)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]
