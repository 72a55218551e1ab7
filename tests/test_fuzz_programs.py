"""Randomised differential test of the recorder + interpreter: random straight-line programs over every recordable witness op
(random operand choices, variables reused many times so that destination lists get long, program windows crossed at arbitrary
offsets, both scopes, carried state), executed by the CPU oracle interpreter and — under -m gpu — by the device; traces must be
identical cell for cell, checks must pass, and the device seeding must equal the oracle seeding."""
import numpy as np
import pytest

import zkgl
from helpers import LINK, Rec, new_cs
from oracle import zko

P = zko.P
G = zkgl.GATE


def random_circuit(seed, n_ops=400, limit=5):
    rng = np.random.default_rng(seed)
    fe = lambda: int(rng.integers(0, P, dtype=np.uint64))      # random canonical field element
    cs = new_cs(cols=int(rng.choice([40, 64, 100])), max_trace_len=1 << 22)
    xor_rows = np.array([[a, b, a ^ b] for a in range(16) for b in range(16)], dtype=np.uint64)
    t_xor = cs.add_lookup_table(900, 2, 1, xor_rows)
    r = Rec(cs)

    def body(pool, small, n):
        """pool: field-valued vars; small: vars known to be < 16"""
        for _ in range(n):
            k = int(rng.integers(0, 11))
            pick = lambda: pool[int(rng.integers(0, len(pool)))]
            if k == 0:
                pool.append(r.fma(fe(), pick(), pick(), fe(), pick()))
            elif k == 1:
                pool.append(r.lc4([pick() for _ in range(4)], [fe() for _ in range(4)]))
            elif k == 2:
                f, _ = r.iszero(pick())
                pool.append(r.select(f, pick(), pick()))
            elif k == 3:
                pool.append(r.dot4([pick() for _ in range(4)], [pick() for _ in range(4)]))
            elif k == 4:
                outs = r.matmul(int(rng.integers(0, 2)), [pick() for _ in range(12)])
                pool.extend(outs[:3])
            elif k == 5:
                outs = r.poseidon2_witness_only([pick() for _ in range(12)])
                pool.extend(outs[:2])
            elif k == 6 and len(small) >= 2:
                a, b = small[int(rng.integers(0, len(small)))], small[int(rng.integers(0, len(small)))]
                (x,) = cs.perform_lookup(t_xor, [a, b], 1)
                small.append(x); pool.append(x)
            elif k == 7 and len(small) >= 2:
                a, b = small[int(rng.integers(0, len(small)))], small[int(rng.integers(0, len(small)))]
                s, c = r.uadd(4, a, b, r.const(0))
                small.append(s)
                d, bo = r.usub(4, a, b, r.const(0))
                small.append(d)
            elif k == 8 and small:
                lo, hi = r.u32muladd(small[int(rng.integers(0, len(small)))], small[int(rng.integers(0, len(small)))], small[0], small[-1])
                pool.append(lo)
            elif k == 9 and small:
                x = small[int(rng.integers(0, len(small)))]
                q = cs.alloc_multiple_variables_without_values(2)
                cs.emit_op(zkgl.OP["DIVREM"], [x], q, b=int(rng.integers(1, 9)))
                pool.extend(q)
            else:
                pool.append(r.const(fe()))

    pool, small = [r.const(1)], []
    n_outer_in = 6
    for i in range(n_outer_in):
        v = r.inp()
        (pool if i < 3 else small).append(v)
        if i >= 3:
            pool.append(v)
    body(pool, small, n_ops // 4)
    first = pool[-1]
    n_outer = r.n_in
    cs.loop_begin(limit)
    r.n_in = 0
    acc_in = r.inp()
    cs.link(LINK["FIRST"], acc_in, first)
    lpool = [acc_in, r.const(1), cs.loop_import(pool[1]), cs.loop_import(first)]
    lsmall = []
    for i in range(5):
        v = r.inp()
        (lpool if i < 2 else lsmall).append(v)
        if i >= 2:
            lpool.append(v)
    body(lpool, lsmall, n_ops)
    acc_out = r.fma(1, lpool[-1], lpool[-2], 1, acc_in)
    cs.link(LINK["CARRY"], acc_in, acc_out)
    n_loop = r.n_in
    cs.loop_end()
    fin = cs.loop_last(acc_out)
    body(pool, small, n_ops // 8)
    cs.place_gate(G["PUBLIC_INPUT"], [r.fma(1, fin, pool[-1], 1, pool[-2])])
    cs.pad_and_shrink()
    return cs, n_outer, n_loop, limit


def inputs(seed, n_outer, n_loop, batch, limit):
    rng = np.random.default_rng(seed + 1)
    outer = (rng.integers(0, 1 << 63, size=(n_outer, batch), dtype=np.uint64) % np.uint64(P)).astype(np.uint64)
    outer[3:] = rng.integers(0, 16, size=(n_outer - 3, batch))
    loop = (rng.integers(0, 1 << 63, size=(n_loop, batch * limit), dtype=np.uint64) % np.uint64(P)).astype(np.uint64)
    loop[3:] = rng.integers(0, 16, size=(n_loop - 3, batch * limit))
    loop[0] = 0   # carried accumulator: seeded
    return outer, loop


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_programs_on_the_oracle(seed):
    cs, n_outer, n_loop, limit = random_circuit(seed)
    outer, loop = inputs(seed, n_outer, n_loop, 3, limit)
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 3, 256).seed(outer, loop)
    run = zko.CircuitRun(cs.export(False), cs.export(True), 3, 256)
    run.resolve(outer, seeded)
    bad, nrel = run.check()
    assert bad == 0 and nrel == 3 * cs.stats()["constraints_per_instance"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(10, 22)))
def test_random_programs_gpu_equals_oracle(zk, seed):
    cs, n_outer, n_loop, limit = random_circuit(seed, n_ops=300 + 37 * (seed % 5), limit=3 + seed % 4)
    batch = [1, 5, 64, 65, 130][seed % 5]
    outer, loop = inputs(seed, n_outer, n_loop, batch, limit)
    cs.set_batch(batch)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, n_outer)
    cs.bind_inputs(True, d_l, n_loop)
    cs.seed_carried_inputs(d_l)
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), batch, 256).seed(outer, loop)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), seeded)
    ok, f = cs.resolve_and_check()
    assert ok, f
    run = zko.CircuitRun(cs.export(False), cs.export(True), batch, 256)
    run.resolve(outer, seeded)
    assert run.check()[0] == 0
    assert np.array_equal(cs.trace(False), run.oc) and np.array_equal(cs.trace(True), run.lc)
    # the same circuit with every program (witness phases and seeding cone) forced into strand form, then into plain form
    import os
    for strands in ("1", "0"):
        os.environ["ZKGL_STRANDS"] = os.environ["ZKGL_SEED_STRANDS"] = strands
        try:
            d_l2 = zk.DeviceBuffer.from_numpy(loop)
            cs.bind_inputs(True, d_l2, n_loop)
            cs.seed_carried_inputs(d_l2)
            assert np.array_equal(d_l2.to_numpy().reshape(loop.shape), seeded), strands
            ok, f = cs.resolve_and_check()
            assert ok, (strands, f)
            assert np.array_equal(cs.trace(False), run.oc) and np.array_equal(cs.trace(True), run.lc), strands
        finally:
            del os.environ["ZKGL_STRANDS"], os.environ["ZKGL_SEED_STRANDS"]
