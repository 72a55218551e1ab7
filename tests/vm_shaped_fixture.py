"""tests/vm_shaped_fixture.py — TEST INFRASTRUCTURE: the synthetic "main_vm-shaped" cycle (csrc/circuits/vm_shaped.cpp, round 1's stand-in
for main_vm) as a small mixed-gate circuit for the prover-stage and seeding tests.  Not a workload of bench.py (the product's VM
circuit is main_vm)."""
import numpy as np


def vm_shaped_inputs(rng, n_outer, n_loop, batch, limit):
    P = 0xFFFFFFFF00000001
    outer = rng.integers(0, 2**32, size=(n_outer, batch), dtype=np.uint64)
    outer[120:135] = rng.integers(0, 2, size=(15, batch))
    outer[135] = rng.integers(0, 2**16, size=batch)
    outer[138] = rng.integers(0, 2**30, size=batch)
    outer[139:142] = rng.integers(0, 2, size=(3, batch))
    outer[142:154] = rng.integers(0, 2**63, size=(12, batch), dtype=np.uint64) % np.uint64(P)
    outer[154] = rng.integers(0, 2**20, size=batch)
    outer[155:167] = rng.integers(0, 2**63, size=(12, batch), dtype=np.uint64) % np.uint64(P)
    loop = np.zeros((n_loop, batch * limit), dtype=np.uint64)
    loop[183:] = rng.integers(0, 2**32, size=(n_loop - 183, batch * limit), dtype=np.uint64)
    loop[183 + 16] = rng.integers(0, 2, size=batch * limit)
    return outer, loop


vm_inputs = vm_shaped_inputs


def build_vm_shaped_cs(zkgl, log2_rows):
    probe = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), max_trace_len=1 << 30)
    probe.configure_vm_shaped()
    probe.vm_shaped_entry_point(1)
    probe.pad_and_shrink()
    st = probe.stats()
    limit = ((1 << log2_rows) - st["outer_slots"]) // st["loop_slots"]
    probe.close()
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), max_trace_len=1 << log2_rows)
    cs.configure_vm_shaped()
    cs.vm_shaped_entry_point(limit)
    cs.pad_and_shrink()
    return cs, limit


build_vm_cs = build_vm_shaped_cs
