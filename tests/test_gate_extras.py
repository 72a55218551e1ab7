"""a2 / a16 completeness: ReductionByPowersGate<F,4> (src/main_vm/decoded_opcode.rs:275, opcodes/binop.rs:203-217) and lookup
sub-arguments of width 4 (the shape of boojum's 4-bit SHA tables Maj4 / Ch4 / TriXor4, src/code_unpacker_sha256/mod.rs:490-494,
554-566): oracle on CPU, device vs oracle under -m gpu, K10 lookup argument at width 4."""
import numpy as np
import pytest

import zkgl
from helpers import Rec
from oracle import zko
from zkgl import GATE as G, OP

P = zko.P


def build():
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(40, 0, 8, 4))
    cs.allow_lookup(4, 8, True)
    for k in ("CONST", "FMA", "REDUCTION4", "REDUCTION_BY_POWERS4", "PUBLIC_INPUT"):
        cs.allow_gate(G[k])
    maj = np.array([[a, b, c, (a & b) ^ (a & c) ^ (b & c)] for a in range(16) for b in range(16) for c in range(16)], dtype=np.uint64)
    ch = np.array([[a, b, c, (a & b) ^ (~a & 15 & c)] for a in range(16) for b in range(16) for c in range(16)], dtype=np.uint64)
    t_maj = cs.add_lookup_table(401, 3, 1, maj)
    t_ch = cs.add_lookup_table(402, 3, 1, ch)
    r = Rec(cs)
    a, b, c, d = r.inp(), r.inp(), r.inp(), r.inp()
    (m,) = cs.perform_lookup(t_maj, [a, b, c], 1)
    (h,) = cs.perform_lookup(t_ch, [a, b, c], 1)
    (m2,) = cs.perform_lookup(t_maj, [m, h, d], 1)
    # word = a + 16 b + 256 c + 4096 d by powers of 16 (ReductionByPowersGate), and the same through a ReductionGate
    w1 = cs.alloc_variable_without_value()
    cs.emit_op(OP["LC4"], [a, b, c, d], [w1], [1, 16, 256, 4096])
    cs.place_gate(G["REDUCTION_BY_POWERS4"], [a, b, c, d, w1], [16])
    w2 = r.lc4([a, b, c, d], [1, 16, 256, 4096])
    for v in (m2, w1, w2):
        cs.place_gate(G["PUBLIC_INPUT"], [v])
    cs.pad_and_shrink()
    return cs


def inputs(B):
    return np.random.default_rng(4).integers(0, 16, size=(4, B)).astype(np.uint64)


def expected(col):
    a, b, c, d = (int(x) for x in col)
    m, h = (a & b) ^ (a & c) ^ (b & c), (a & b) ^ (~a & 15 & c)
    return [(m & h) ^ (m & d) ^ (h & d), a + 16 * b + 256 * c + 4096 * d, a + 16 * b + 256 * c + 4096 * d]


def test_width4_lookups_and_reduction_by_powers_on_the_oracle():
    cs = build()
    assert cs.stats()["gate_instances"]["REDUCTION_BY_POWERS4"] == 1 and cs.stats()["lookup_columns"] == 32
    inp = inputs(9)
    run = zko.CircuitRun(cs.export(False), cs.export(True), 9, 8192)
    run.resolve(inp, np.zeros((0, 0), dtype=np.uint64))
    bad, nrel = run.check()
    assert bad == 0 and nrel == 9 * cs.stats()["constraints_per_instance"]
    for i in range(9):
        assert [int(run.oc[c, i]) for c in cs.public_cells()] == expected(inp[:, i])
    res = zko.lookup_argument(run, cs.export(False), cs.export(True), (5, 6), (7, 8), 40 + 32)     # K10 at width 4
    assert all(r[0:2] == r[2:4] != (0, 0) for r in res)
    run.oc[cs.public_cells()[1], 3] += 1      # break the powers relation
    assert run.check()[0] > 0


@pytest.mark.gpu
def test_width4_lookups_and_reduction_by_powers_on_the_gpu(zk):
    cs = build()
    B = 130
    inp = inputs(B)
    cs.set_batch(B)
    d = zk.DeviceBuffer.from_numpy(inp)
    cs.bind_inputs(False, d, 4)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i in range(B):
        assert cs.public_inputs(i) == expected(inp[:, i])
    run = zko.CircuitRun(cs.export(False), cs.export(True), B, 8192)
    run.resolve(inp, np.zeros((0, 0), dtype=np.uint64))
    assert np.array_equal(cs.trace(False), run.oc)
    bad, sums = cs.lookup_argument((11, 12), (13, 14))
    want = zko.lookup_argument(run, cs.export(False), cs.export(True), (11, 12), (13, 14), 40 + 32)
    assert bad == 0 and [tuple(int(x) for x in row) for row in sums] == want
    # K12 on a circuit without a loop scope (width-4 lookup columns take part in the permutation): z closes and equals the oracle
    st = cs.stats()
    rows, n_cols = st["rows_per_instance"], st["copy_columns"] + st["lookup_columns"]
    zbuf = zk.DeviceBuffer(B * (rows + 1) * 2)
    bad, out = cs.copy_permutation((21, 22), (23, 24), zbuf)
    assert bad == 0 and np.array_equal(out[:, :2], out[:, 2:])
    ho, hl = zko.parse_export(cs.export(False)), zko.parse_export(cs.export(True))
    sigma = cs.sigma(False)
    assert zko.sigma_matches_classes(sigma, ho, hl, 0)
    z = zbuf.to_numpy().reshape(B, rows + 1, 2)
    want_z = zko.copy_permutation_z(ho, hl, 0, cs.trace(False)[:, 77], [], sigma, (21, 22), (23, 24), n_cols)
    assert [tuple(int(x) for x in r) for r in z[77]] == want_z and want_z[-1] == (1, 0)
    # trace columns of a loop-free circuit: the outer scope's rows, zero padded
    log_n = int(rows - 1).bit_length()
    cols = zk.DeviceBuffer(n_cols << log_n)
    cs.trace_columns(77, cols, log_n)
    zk.sync()
    got = cols.to_numpy().reshape(n_cols, 1 << log_n)
    to = cs.trace(False)
    for c in (0, 7, n_cols - 1):
        assert np.array_equal(got[c, :rows], to[c::n_cols][:rows, 77]) and not got[c, rows:].any()
    cs.write_cell(False, cs.public_cells()[1], 5, 12345)        # w1 of instance 5 no longer equals the powers sum
    ok, f = cs.check_if_satisfied()
    assert not ok and f.instance == 5 and f.kind == G["REDUCTION_BY_POWERS4"]
