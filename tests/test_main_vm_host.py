"""main_vm (SURVEY §8 a15 / a16 / a11 consumer) on the CPU oracle: the recorded circuit (libzkgl recorder -> exported program ->
oracle/zko_engine.c interpreter + checker) against the native restatement of the VM cycle written straight from the Rust
(oracle/main_vm_native.py).  Three independent things are compared: (1) the VmLocalState the circuit carries into every cycle,
derived by sequential seeding from the raw oracle words only, == the native model's state, word for word; (2) every gate /
lookup / copy / link constraint holds on the resolved trace; (3) the public input == the native input commitment."""
import numpy as np
import pytest

import zkgl
import vm_programs as vp
from oracle import main_vm_native as vn
from oracle import zko


_TOTAL_ROWS = {}


def run_oracle(cs, batch):
    if id(cs) not in _TOTAL_ROWS:
        _TOTAL_ROWS[id(cs)] = int(sum(t["n_rows"] for t in zko.parse_export(cs.export(False))["tables"]))
    return zko.CircuitRun(cs.export(False), cs.export(True), batch, _TOTAL_ROWS[id(cs)])


def check_run(cs, D, vrun, limit, n_instances):
    outer, loop = vp.pack_instance_streams(cs, D, vrun, limit, n_instances)
    lay = cs.main_vm_layout()
    first, n = lay["loop"]["state"]
    assert (first, n) == (0, 243)
    # (1) seeding: blank the carried words, let the circuit derive them from the raw witness
    raw = loop.copy()
    raw[0:243] = 0
    run = run_oracle(cs, n_instances)
    seeded = run.seed(outer, raw)
    if not np.array_equal(seeded, loop):
        bad = np.argwhere(seeded != loop)
        w, col = bad[0]
        raise AssertionError(f"carried state differs from the native restatement first at word {w} of cycle {col} "
                             f"(circuit {int(seeded[w, col])}, native {int(loop[w, col])}); {len(bad)} words differ")
    # (2) the resolved trace satisfies everything
    run = run_oracle(cs, n_instances)
    run.resolve(outer, loop)
    bad, nrel = run.check()
    assert bad == 0, f"{bad} violated relations"
    assert nrel == cs.stats()["constraints_per_instance"] * n_instances
    # (3) commitments
    for i in range(n_instances):
        assert [int(run.oc[c, i]) for c in cs.public_cells()] == vp.expected_commitment(D, vrun, limit, i)
    return run, outer, loop


@pytest.mark.parametrize("name", ["arith", "memory", "logs"])
def test_single_frame_programs(name):
    d, D = vp.defs()
    ops = dict(arith=vp.program_arith, memory=vp.program_memory_and_logs, logs=vp.program_logs)[name](D)
    limit = 16
    n_inst = (len(ops) + 6 + limit - 1) // limit
    cs = vp.vm_cs(limit)
    vrun = vn.VmRun(D, vp.make_world_factory(D, ops), n_inst * limit)
    check_run(cs, D, vrun, limit, n_inst)


def test_calls_program():
    d, D = vp.defs()
    ops, contracts = vp.program_calls(D)
    limit = 16
    cs = vp.vm_cs(limit)
    probe = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), 600)
    done_at = next(i for i, s in enumerate(probe.states) if s.depth == 0)
    n_inst = (done_at + 3 + limit - 1) // limit
    vrun = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), n_inst * limit)
    assert vrun.states[-1].depth == 0 and vrun.states[-1].ctx.pc == 0   # the bootloader frame returned ok
    check_run(cs, D, vrun, limit, n_inst)


def test_every_opcode_family_is_executed():
    """the four programs together run all eleven families of cycle.rs:73-156 (+ NOP / PANIC masking) un-masked at least once"""
    d, D = vp.defs()
    seen = set()
    ops, contracts = vp.program_calls(D)
    for make in (vp.make_world_factory(D, vp.program_arith(D)), vp.make_world_factory(D, vp.program_memory_and_logs(D)),
                 vp.make_world_factory(D, vp.program_logs(D)), vp.make_world_factory(D, ops, contracts)):
        vrun = vn.VmRun(D, make, 260)
        for st, W in vrun.rows:
            if st.depth and not W["_masked"]:
                seen.add(W["_family"])
    names = {v: k for k, v in vn.FAM.items()}
    assert {names[f] for f in seen} == set(vn.FAM) - {"INVALID"}


def test_negative_cases():
    d, D = vp.defs()
    ops, contracts = vp.program_calls(D)
    limit = 16
    cs = vp.vm_cs(limit)
    n_inst = 13
    vrun = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), n_inst * limit)
    outer, loop = vp.pack_instance_streams(cs, D, vrun, limit, n_inst)
    lay = cs.main_vm_layout()["loop"]

    def violations(o, l):
        run = run_oracle(cs, n_inst)
        run.resolve(o, l)
        return run.check()[0]

    assert violations(outer, loop) == 0
    ret_cycle = next(i for i, (st, W) in enumerate(vrun.rows) if W["_family"] == vn.FAM["RET"] and st.depth > 1)
    write_cycle = next(i for i, (st, W) in enumerate(vrun.rows) if any(W["log_rollback_queue_prev_head"]))
    # (a) a carried register word that is not the previous cycle's output
    bad = loop.copy(); bad[9, 5] ^= 1
    assert violations(outer, bad) > 0
    # (b) the popped callstack entry does not hash to the callstack sponge
    bad = loop.copy(); bad[lay["ret_popped_context"][0] + 31, ret_cycle] += 1   # ergs_remaining of the popped frame
    assert violations(outer, bad) > 0
    # (c) a rollback-queue head claim that does not hash to the current head
    bad = loop.copy(); bad[lay["log_rollback_queue_prev_head"][0], write_cycle] ^= 1
    assert violations(outer, bad) > 0
    # (d) a code word limb out of the u32 range
    bad = loop.copy(); bad[lay["code_word"][0] + 2, 3] = 1 << 32
    assert violations(outer, bad) > 0
    # (e) start_flag of a continuation instance flipped: its first cycle no longer links to the hidden FSM input
    bad = outer.copy(); bad[0, 2] = 1
    assert violations(bad, loop) > 0
