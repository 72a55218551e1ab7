"""Generates tests/golden/vm_bench_witness.npz: the raw WitnessOracle words (per-cycle input stream WITHOUT the carried
VmLocalState) of N_EXEC synthetic zkEVM executions of the endless mixed-workload program (tests/vm_programs.py
program_bench_loop), produced by the native restatement oracle/main_vm_native.py, plus the input commitment the native model
expects for the first `limit`-cycle chunk at 2^20 rows.  bench.py loads this file as DATA (it never imports oracle/ for its
workload), lets the device derive the carried state (zk_cs_seed_carried_inputs) and compares the public inputs it gets with the
commitments stored here.

    python tests/golden/make_vm_bench_witness.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import vm_programs as vp  # noqa: E402
import zkgl  # noqa: E402
from oracle import main_vm_native as vn  # noqa: E402

N_EXEC, LOG2_ROWS = 8, 20


def main():
    d, D = vp.defs()
    probe = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), 1 << 30, 1 << 28)
    probe.configure_main_vm(d)
    probe.main_vm_entry_point(1)
    probe.pad_and_shrink()
    st = probe.stats()
    limit = ((1 << LOG2_ROWS) - st["outer_slots"]) // st["loop_slots"]
    lay = probe.main_vm_layout()
    n_outer, n_loop = probe.input_words()
    assert lay["loop"]["state"] == (0, 243)
    raw = np.zeros((N_EXEC, limit, n_loop - 243), dtype=np.uint64)
    tails = np.zeros((N_EXEC, 4), dtype=np.uint64)
    commits = np.zeros((N_EXEC, 4), dtype=np.uint64)
    for e in range(N_EXEC):
        ops, contracts = vp.program_bench_loop(D, e)
        vrun = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), limit)
        for k, (state, W) in enumerate(vrun.rows):
            for name, words in W.items():
                if name.startswith("_"):
                    continue
                first, n = lay["loop"][name]
                raw[e, k, first - 243:first - 243 + n] = np.array([int(x) for x in words], dtype=np.uint64)
        tails[e] = np.array(vrun.rollback_tail_for_block, dtype=np.uint64)
        commits[e] = np.array(vp.expected_commitment(D, vrun, limit, 0), dtype=np.uint64)
        print(f"execution {e}: {limit} cycles, commitment {[hex(int(x)) for x in commits[e]]}")
    out = os.path.join(HERE, "vm_bench_witness.npz")
    np.savez_compressed(out, raw=raw, rollback_tail=tails, commitment=commits, limit=np.array([limit]), log2_rows=np.array([LOG2_ROWS]),
                        layout=np.frombuffer(json.dumps(lay).encode(), dtype=np.uint8))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
