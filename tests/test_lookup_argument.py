"""K10: log-derivative lookup-argument accumulators (csrc/kernels_lookup_arg.hpp).  CPU part: the pure-Python
restatement balances (A == B) on the oracle trace of a circuit with lookups and stops balancing when a multiplicity or a
looked-up value is tampered with.  GPU part (-m gpu): the device sums equal the restatement bit for bit."""
import numpy as np
import pytest

import zkgl
from helpers import load_fixture, oracle_run, ram_cs
from oracle import ram_native as rn
from oracle import zko

BETA, GAMMA = (0x123456789ABCDEF, 0x0FEDCBA987654321), (0x1111111122222222, 0x3333333344444444)
N_COLS = 100 + 24


def ram_case(batch=3, limit=4):
    cs = ram_cs(limit)
    rng = np.random.default_rng(77)
    insts = []
    for _ in range(batch):
        u, s, nd = rn.random_ram_witness(rng, limit - 1, n_cells=3)
        insts.append(rn.instance(u, s, limit, nd))
    outer, loop = rn.pack_streams(insts, limit)
    return cs, outer, loop, batch


def test_restatement_balances_and_detects_tampering():
    cs, outer, loop, batch = ram_case()
    run = oracle_run(cs, outer, loop, batch)
    assert run.check()[0] == 0
    res = zko.lookup_argument(run, cs.export(False), cs.export(True), BETA, GAMMA, N_COLS)
    assert all(r[0:2] == r[2:4] and r[0:2] != (0, 0) for r in res)
    run.mult[int(np.flatnonzero(run.mult)[0])] += 1                      # one multiplicity too many
    res2 = zko.lookup_argument(run, cs.export(False), cs.export(True), BETA, GAMMA, N_COLS)
    assert res2[0][0:2] != res2[0][2:4] and res2[1:] == res[1:]


@pytest.mark.gpu
def test_device_sums_equal_restatement(zk):
    cs, outer, loop, batch = ram_case(batch=5, limit=6)
    cs.set_batch(batch)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    ok, f = cs.resolve_and_check()
    assert ok, f
    bad, sums = cs.lookup_argument(BETA, GAMMA)
    assert bad == 0
    run = oracle_run(cs, outer, loop, batch)
    want = zko.lookup_argument(run, cs.export(False), cs.export(True), BETA, GAMMA, N_COLS)
    assert [tuple(int(x) for x in row) for row in sums] == want
    # a looked-up value changed in the trace: the witness side moves, the table side does not
    cell = 100  # first lookup column of slot 0 ... find a populated lookup cell of the loop scope
    h = zko.parse_export(cs.export(True))
    slot = next(i for i, (t, n) in enumerate(h["lrows"]) if t != 0xFFFFFFFF and n)
    cs.write_cell(True, slot * N_COLS + 100, 0, 999)
    bad2, sums2 = cs.lookup_argument(BETA, GAMMA)
    assert bad2 == 1 and tuple(sums2[0][2:4]) == tuple(sums[0][2:4]) and tuple(sums2[0][0:2]) != tuple(sums[0][0:2])


@pytest.mark.gpu
def test_lookup_argument_on_the_bench_circuit(zk):
    """C2 shape (140 + 24 columns, 6 tables, 102 lookups per cycle) at a small limit: balances on the device"""
    import vm_shaped_fixture as bench
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), max_trace_len=1 << 20)
    cs.configure_vm_shaped()
    cs.vm_shaped_entry_point(40)
    cs.pad_and_shrink()
    n_outer, n_loop = cs.input_words()
    B = 70
    outer, loop = bench.vm_inputs(np.random.default_rng(1), n_outer, n_loop, B, 40)
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, n_outer)
    cs.bind_inputs(True, d_l, n_loop)
    cs.seed_carried_inputs(d_l)
    ok, f = cs.resolve_and_check()
    assert ok, f
    bad, sums = cs.lookup_argument(BETA, GAMMA)
    assert bad == 0 and np.all(sums[:, 0:2] == sums[:, 2:4]) and np.all(sums[:, 0] != 0)
