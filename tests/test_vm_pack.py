"""Product-side input path of main_vm: zk_pack_main_vm_witness takes what a host of the reference has — the
VmCircuitWitness: closed-form input + the WitnessOracle's per-getter FIFOs (src/main_vm/witness_oracle.rs:45-91,
src/fsm_input_output/circuit_inputs/main_vm.rs:64-71) — and writes the circuit's input streams.  The native walker behind it
(csrc/vm_native.hpp, also phase A of the device seeding) must place every answer at the cycle that asks for it, and — with
ZK_VM_PACK_FILL_STATE — reproduce the VmLocalState of every cycle.  The oracle (oracle/main_vm_native.py) only compares."""
import numpy as np
import pytest

import zkgl
import vm_programs as vp
from oracle import main_vm_native as vn


def _programs(D):
    out = []
    for seed in (0, 1):
        out.append(("arith", seed, vp.program_arith(D, seed), None))
        out.append(("memory", seed, vp.program_memory_and_logs(D, seed), None))
        out.append(("logs", seed, vp.program_logs(D, seed), None))
        ops, contracts = vp.program_calls(D, seed)
        out.append(("calls", seed, ops, contracts))
    return out


def _first_difference(a, b, lay):
    bad = np.argwhere(a != b)
    w, col = bad[0]
    name = next((n for n, (f, k) in lay["loop"].items() if f <= w < f + k), "?")
    return f"{len(bad)} words differ, first at word {w} ({name}+{w - lay['loop'].get(name, (0, 0))[0]}) of column {col}: packer {int(a[w, col])}, native {int(b[w, col])}"


@pytest.mark.parametrize("fill_state", [False, True])
def test_packer_places_every_oracle_answer_at_its_cycle(fill_state):
    d, D = vp.defs()
    limit = 16
    cs = vp.vm_cs(limit)
    lay = cs.main_vm_layout()
    for name, seed, ops, contracts in _programs(D):
        probe = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), 4 * len(ops) + 64)
        done_at = next(i for i, s in enumerate(probe.states) if s.depth == 0)
        n_inst = (done_at + 1 + limit - 1) // limit + 1   # one chunk past the end: skipped cycles (empty callstack)
        run = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), n_inst * limit)
        want_outer, want_loop = vp.pack_instance_streams(cs, D, run, limit, n_inst)
        outer, loop, reports = vp.pack_through_the_c_abi(cs, run, limit, n_inst, fill_state=fill_state)
        assert not any(r.underflow for r in reports), (name, seed)
        assert np.array_equal(outer, want_outer), (name, seed)
        if not fill_state:
            want_loop = want_loop.copy()
            want_loop[0:243] = 0
        assert np.array_equal(loop, want_loop), (name, seed, _first_difference(loop, want_loop, lay))
        if fill_state:   # hidden_fsm_output of every chunk == the native state after its last cycle
            for i, r in enumerate(reports):
                assert list(r.final_state) == [int(x) for x in run.states[(i + 1) * limit].flatten()], (name, seed, i)
        # every FIFO consumed exactly
        q = vp.oracle_queues(run, 0, n_inst * limit)
        used = [sum(getattr(r, f) for r in reports) for f in ("used_memory_reads", "used_storage_reads", "used_refunds", "used_rollback_queue_witness",
                                                                "used_rollback_tails_for_call", "used_callstack", "used_decommit_pages")]
        assert used == [len(q.memory_reads), len(q.storage_reads), len(q.refunds), len(q.rollback_queue_witness), len(q.rollback_tails_for_call),
                        len(q.callstack), len(q.decommit_pages)], (name, seed)


def test_packer_reports_underflow_and_rejects_foreign_circuits():
    d, D = vp.defs()
    limit = 16
    cs = vp.vm_cs(limit)
    ops = vp.program_memory_and_logs(D, 0)
    run = vn.VmRun(D, vp.make_world_factory(D, ops), limit)
    q = vp.oracle_queues(run, 0, limit)
    assert len(q.memory_reads) > 2
    q.memory_reads = q.memory_reads[:2]
    q.freeze()
    ow, lw = cs.input_words()
    outer, loop = np.zeros((ow, 1), dtype=np.uint64), np.zeros((lw, limit), dtype=np.uint64)
    rep = cs.pack_main_vm_witness(vp.closed_form_input(run, 0), q.view(), 0, 1, outer, loop)
    assert rep.underflow == 1 and rep.used_memory_reads == 2
    other = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    other.configure_ram_permutation()
    other.ram_permutation_entry_point(4)
    other.pad_and_shrink()
    with pytest.raises(zkgl.ZkError):
        other.pack_main_vm_witness(vp.closed_form_input(run, 0), q.view(), 0, 1, outer, loop)


def test_bench_fixture_packs_without_underflow():
    """tests/golden/vm_bench_witness.npz (64 executions as WitnessOracle FIFOs) -> streams: every FIFO exactly consumed, 64 distinct
    executions, raw words only (bench.py and tests/test_gpu_full_size.py feed the device from this)"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    cs, limit = bench.build_main_vm_cs(zkgl, 20)
    outer, loop, expect = bench.main_vm_streams(zkgl, cs, limit, 16)
    assert expect is not None and outer.shape[1] == 16 and loop.shape[1] == 16 * limit
    assert not loop[:243].any()
    lay = cs.main_vm_layout()["loop"]
    f, n = lay["code_word"]
    assert len({loop[f:f + n, e * limit:(e + 1) * limit].tobytes() for e in range(16)}) == 16
