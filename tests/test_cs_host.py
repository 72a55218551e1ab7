"""Host logic (recorder / placement / program emission) checked WITHOUT a GPU by executing the
exported scopes on the CPU oracle interpreter.  Pattern A of the reference's tests: "the circuit is
satisfiable on the fixture" (/root/reference/src/ram_permutation/mod.rs:417-557) — plus the
permutation-positive / permutation-negative entry-point tests SURVEY.md Appendix D asks for."""
import json
import os

import numpy as np
import pytest

import zkgl
from helpers import (GOLD, LINK, OP, G, P, Rec, load_fixture, new_cs, oracle_run, ram_cs, rand_fe, random_instances)
from oracle import ram_native as rn
from oracle import zko


def all_ops_circuit(cs, limit=4):
    """A circuit touching every recordable op / gate kind, both scopes, every link kind.
    Returns (n_outer_inputs, n_loop_inputs)."""
    xor_rows = np.array([[a, b, a ^ b] for a in range(16) for b in range(16)], dtype=np.uint64)         # dense 4+4 bits
    sparse_rows = np.array([[k * 7 + 3, (k * k) % 1000, 0] for k in range(50)], dtype=np.uint64)[:, :2]  # 1 key -> 1 val, non-dense
    t_xor = cs.add_lookup_table(1001, 2, 1, xor_rows)
    t_sparse = cs.add_lookup_table(1002, 1, 1, sparse_rows)
    r = Rec(cs)
    one = r.const(1)
    a, b, c = r.inp(), r.inp(), r.inp()           # field elements
    x32, y32 = r.inp(), r.inp()                   # u32 values
    nib_a, nib_b = r.inp(), r.inp()               # 4-bit values
    skey = r.inp()                                # key of the sparse table
    sel = r.inp()                                 # boolean
    cs.place_gate(G["BOOLEAN"], [sel])
    d = r.fma(3, a, b, P - 2, c)
    e = r.lc4([a, b, c, d], [1, 1 << 40, P - 1, 7])
    s = r.select(sel, d, e)
    f0, _ = r.iszero(s)
    z = r.fma(1, a, one, P - 1, a)                # a - a == 0
    f1, _ = r.iszero(z)
    cs.place_gate(G["FMA"], [f1, one, f1, one], [1, 0])  # enforce f1 == 1
    sm, co = r.uadd(32, x32, y32, sel)
    df, bo = r.usub(32, x32, y32, sel)
    by = r.split(x32, 4, 8, [1, 1 << 8, 1 << 16, 1 << 24])
    lo, hi = r.u32muladd(x32, y32, sm, df)
    (xv,) = cs.perform_lookup(t_xor, [nib_a, nib_b], 1)
    (sv,) = cs.perform_lookup(t_sparse, [skey], 1)
    dp = r.dot4([a, b, c, d], [e, s, xv, sv])
    st = [a, b, c, d, e, s, dp, lo, hi, sm, df, by[0]]
    m_e = r.matmul(0, st)
    m_i = r.matmul(1, m_e)
    p_w = r.poseidon2_witness_only(m_i)
    acc0 = r.fma(1, p_w[0], one, 1, p_w[5])
    # ---- loop: acc_{k+1} = acc_k * t + k_in ; imports `a` ; exports last ----
    n_outer = r.n_in
    cs.loop_begin(limit)
    r.n_in = 0
    acc_in = r.inp()
    cs.link(LINK["FIRST"], acc_in, acc0)
    t = r.inp()
    a_l = cs.loop_import(a)
    one_l = r.const(1)
    acc_out = r.fma(1, acc_in, t, 1, a_l)
    fl, _ = r.iszero(t)
    tt = r.select(fl, one_l, t)
    bys = r.split(tt, 4, 8, [1, 1 << 8, 1 << 16, 1 << 24])
    (xl,) = cs.perform_lookup(t_xor, [r.inp(), r.inp()], 1)
    r.fma(1, xl, bys[0], 0, xl)
    cs.link(LINK["CARRY"], acc_in, acc_out)
    n_loop = r.n_in
    cs.loop_end()
    fin = cs.loop_last(acc_out)
    pub = r.fma(1, fin, one, 1, acc0)
    cs.place_gate(G["PUBLIC_INPUT"], [pub])
    return n_outer, n_loop


def all_ops_inputs(rng, batch, limit, n_outer, n_loop):
    """raw (non-carried) witness words; loop word 0 (the carried accumulator) is left for seeding"""
    outer = np.zeros((n_outer, batch), dtype=np.uint64)
    loop = np.zeros((n_loop, batch * limit), dtype=np.uint64)
    for i in range(batch):
        a, b, c = rand_fe(rng, 3)
        x32, y32 = int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32))
        if i == 0:
            x32, y32 = 0xFFFFFFFF, 0xFFFFFFFF
        if i == 1:
            x32, y32 = 0, 0xFFFFFFFF
        if i == 2:
            a, b, c = 0, P - 1, 1
        outer[:, i] = [a, b, c, x32, y32, int(rng.integers(0, 16)), int(rng.integers(0, 16)), int(rng.integers(0, 50)) * 7 + 3, i & 1]
        for k in range(limit):
            t = 0 if k == 1 else rand_fe(rng, 1)[0]
            loop[1:, i * limit + k] = [t, int(rng.integers(0, 16)), int(rng.integers(0, 16))]
    return outer, loop


def test_all_ops_circuit_on_oracle():
    limit, batch = 4, 3
    rng = np.random.default_rng(5)
    cs = new_cs()
    n_outer, n_loop = all_ops_circuit(cs, limit)
    cs.pad_and_shrink()
    outer, loop_raw = all_ops_inputs(rng, batch, limit, n_outer, n_loop)
    run = zko.CircuitRun(cs.export(False), cs.export(True), batch, 256 + 50)
    # generic sequential seeding fills the carried accumulator: acc[k+1] = acc[k]*t[k] + a
    loop = run.seed(outer, loop_raw)
    for i in range(batch):
        a = int(outer[0, i])
        for k in range(limit - 1):
            acc, t = int(loop[0, i * limit + k]), int(loop[1, i * limit + k])
            assert int(loop[0, i * limit + k + 1]) == (acc * t + a) % P
    assert np.array_equal(loop[1:], loop_raw[1:])
    run2 = oracle_run(cs, outer, loop, batch, 256 + 50)
    bad, nrel = run2.check()
    assert bad == 0
    assert nrel == cs.stats()["constraints_per_instance"] * batch
    assert np.array_equal(run2.lc, run.lc)  # parallel mode reproduces the sequential trace
    # multiplicities: every lookup counted once
    assert int(run2.mult.sum()) == cs.stats()["lookups_per_instance"] * batch
    # a wrong carried value breaks the link check; the unseeded stream does as well
    loop_bad = loop.copy(); loop_bad[0, 2] ^= 1
    assert oracle_run(cs, outer, loop_bad, batch, 256 + 50).check()[0] > 0
    assert oracle_run(cs, outer, loop_raw, batch, 256 + 50).check()[0] > 0


def test_lookup_of_absent_key_is_unsatisfied():
    cs = new_cs()
    rows = np.array([[k * 7 + 3, k] for k in range(50)], dtype=np.uint64)
    t = cs.add_lookup_table(5, 1, 1, rows)
    key = cs.input(0)
    cs.perform_lookup(t, [key], 1)
    cs.pad_and_shrink()
    for k, good in ((10, True), (11, False), (3 + 49 * 7, True), (3 + 50 * 7, False)):
        run = oracle_run(cs, np.array([[k]], dtype=np.uint64), np.zeros((0, 0), dtype=np.uint64), 1, 50)
        assert (run.check()[0] == 0) == good


def test_poseidon_macro_op_equals_primitive_ops_and_witness_only_op():
    """The in-circuit permutation (962 constrained intermediates) ends in the same 12 values as the
    witness-only op and as the oracle's permutation."""
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_ram_permutation()
    cs.ram_permutation_entry_point(1)
    cs.pad_and_shrink()
    st = cs.stats()
    assert st["gate_instances"]["MATMUL12_EXT"] % 9 == 0 and st["gate_instances"]["MATMUL12_INT"] % 22 == 0


# ------------------------------------------------------------------ ram_permutation, reference fixture
def test_ram_fixture_satisfiable_and_commitment():
    u, s, limit = load_fixture()
    cs = ram_cs(limit)
    assert cs.input_words() == (121, 72)
    inst = rn.instance(u, s, limit, 1)
    outer, loop = rn.pack_streams([inst], limit)
    run = oracle_run(cs, outer, loop, 1)
    bad, nrel = run.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    got = [int(run.oc[c, 0]) for c in cs.public_cells()]
    assert got == inst["commitment"]
    gold = json.load(open(os.path.join(GOLD, "ram_commitments.json")))
    assert got == [int(x, 16) for x in gold["fixture_limit16"]]


def test_ram_geometry_and_stats():
    cs = ram_cs(16)
    st = cs.stats()
    assert st["copy_columns"] == 100 and st["lookup_columns"] == 24
    assert st["rows_per_instance"] == st["loop_slots"] * 16 + st["outer_slots"]
    # per cycle: 2 in-circuit permutations (2 pops), 33 once per instance (5 FS + 28 commitments)
    assert st["gate_instances"]["MATMUL12_EXT"] == 9 * (2 * 16 + 33)
    assert st["gate_instances"]["MATMUL12_INT"] == 22 * (2 * 16 + 33)
    assert st["gate_instances"]["PUBLIC_INPUT"] == 4


@pytest.mark.parametrize("seed,n_items,limit", [(1, 5, 8), (2, 8, 8), (3, 12, 16)])
def test_ram_random_witness_positive(seed, n_items, limit):
    rng = np.random.default_rng(seed)
    u, s, nd = rn.random_ram_witness(rng, n_items)
    inst = rn.instance(u, s, limit, nd)
    assert inst["satisfiable"] and inst["completed"]
    outer, loop = rn.pack_streams([inst], limit)
    run = oracle_run(ram_cs(limit), outer, loop, 1)
    assert run.check()[0] == 0
    got = [int(run.oc[c, 0]) for c in ram_cs(limit).public_cells()]
    assert got == inst["commitment"]
    gold = json.load(open(os.path.join(GOLD, "ram_commitments.json")))
    g = [x for x in gold["random"] if x["seed"] == seed][0]
    assert got == [int(x, 16) for x in g["commitment"]]


def test_ram_empty_queue_and_full_queue():
    limit = 8
    cs = ram_cs(limit)
    empty = rn.instance([], [], limit, 0)           # nothing to pop: every cycle is padding
    assert empty["satisfiable"] and empty["completed"]
    rng = np.random.default_rng(9)
    u, s, nd = rn.random_ram_witness(rng, limit)      # queue exactly fills the instance
    full = rn.instance(u, s, limit, nd)
    outer, loop = rn.pack_streams([empty, full], limit)
    run = oracle_run(cs, outer, loop, 2)
    assert run.check()[0] == 0
    for i, inst in enumerate((empty, full)):
        assert [int(run.oc[c, i]) for c in cs.public_cells()] == inst["commitment"]


def test_ram_negative_not_a_permutation_and_not_sorted():
    u, s, limit = load_fixture()
    cs = ram_cs(limit)
    # (a) sorted side is not a permutation of the unsorted side -> lhs != rhs at completion
    s_bad = [list(x) for x in s]; s_bad[1][5] ^= 4; s_bad[2][5] ^= 4   # consistent read/write pair, different value
    inst = rn.instance(u, s_bad, limit, 1)
    assert not inst["satisfiable"]
    outer, loop = rn.pack_streams([inst], limit)
    assert oracle_run(cs, outer, loop, 1).check()[0] > 0
    # (b) order violated
    inst = rn.instance(u, [s[1], s[0], s[2]], limit, 1)
    outer, loop = rn.pack_streams([inst], limit)
    assert oracle_run(cs, outer, loop, 1).check()[0] > 0
    # (c) read of an uninitialised cell returning non-zero
    lone = [rn.mq(5, 40, 1, 0, 0, 77)]
    inst = rn.instance(lone, lone, limit, 0)
    assert not inst["satisfiable"]
    outer, loop = rn.pack_streams([inst], limit)
    assert oracle_run(cs, outer, loop, 1).check()[0] > 0


def test_ram_two_chunk_continuation():
    """hidden_fsm_output of chunk k is hidden_fsm_input of chunk k+1 (SURVEY §5 checkpoint/resume)."""
    limit = 4
    cs = ram_cs(limit)
    rng = np.random.default_rng(21)
    u, s, nd = rn.random_ram_witness(rng, 7)
    ub, utail = rn.queue_simulate(u); sb, stail = rn.queue_simulate(s)
    ub.append(utail); sb.append(stail)
    obs_u, obs_s = [0] * 12 + utail + [7], [0] * 12 + stail + [7]
    first = rn.instance(u[:4], s[:4], limit, nd, start_flag=True)
    # the first chunk sees the GLOBAL queue (7 items) but only pops 4: rebuild it with global observable state
    fsm0 = rn.empty_fsm()
    c1 = rn.instance(u[:4], s[:4], limit, nd, start_flag=False, fsm_in=dict(fsm0, lhs=[1, 1], rhs=[1, 1], unsorted=obs_u, sorted=obs_s),
                     obs_unsorted=obs_u, obs_sorted=obs_s, heads=(ub, sb))
    assert not c1["completed"]
    c2 = rn.instance(u[4:], s[4:], limit, nd, start_flag=False, fsm_in=c1["fsm_out"], obs_unsorted=obs_u, obs_sorted=obs_s,
                     heads=(ub, sb))
    assert c2["completed"] and c2["satisfiable"]
    outer, loop = rn.pack_streams([c1, c2], limit)
    run = oracle_run(cs, outer, loop, 2)
    assert run.check()[0] == 0
    assert [int(run.oc[c, 1]) for c in cs.public_cells()] == c2["commitment"]
    del first


# ------------------------------------------------------------------ main_vm-shaped cycle (config C2)
def vm_cs(limit):
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8))
    cs.configure_vm_shaped()
    cs.vm_shaped_entry_point(limit)
    cs.pad_and_shrink()
    return cs


VM_TABLE_ROWS = 65536 * 2 + 2048 + 64 + 16 + 1024


def test_vm_shaped_cycle_budget_and_satisfiability():
    from vm_shaped_fixture import vm_inputs
    limit, batch = 3, 2
    cs = vm_cs(limit)
    st1, st3 = vm_cs(1).stats(), cs.stats()
    per_cycle = {k: (st3["gate_instances"][k] - st1["gate_instances"][k]) // 2 for k in st3["gate_instances"]}
    # SURVEY §8 a15 budget: 9 in-circuit permutations, 2x8 UIntXAddGate<32> (+1 ergs), 3x64 fma_with_carry
    assert per_cycle["MATMUL12_EXT"] == 9 * 9 and per_cycle["MATMUL12_INT"] == 9 * 22
    assert per_cycle["UINTX_ADD"] == 17 and per_cycle["U32_FMA"] == 192
    assert st3["copy_columns"] == 140 and st3["lookup_columns"] == 24
    n_outer, n_loop = cs.input_words()
    assert (n_outer, n_loop) == (183, 225)
    rng = np.random.default_rng(0xC2)
    outer, loop_raw = vm_inputs(rng, n_outer, n_loop, batch, limit)
    run = zko.CircuitRun(cs.export(False), cs.export(True), batch, VM_TABLE_ROWS)
    loop = run.seed(outer, loop_raw)
    run2 = oracle_run(cs, outer, loop, batch, VM_TABLE_ROWS)
    bad, nrel = run2.check()
    assert bad == 0 and nrel == st3["constraints_per_instance"] * batch
    assert np.array_equal(run2.lc, run.lc)
    # tampering with one raw oracle word (a code word limb) must break satisfiability
    loop_bad = loop.copy(); loop_bad[183, 1] ^= 1
    assert oracle_run(cs, outer, loop_bad, batch, VM_TABLE_ROWS).check()[0] > 0
