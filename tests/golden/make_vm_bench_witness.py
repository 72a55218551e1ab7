"""Generates tests/golden/vm_bench_witness.npz: the VmCircuitWitness of N_EXEC synthetic zkEVM executions of the endless
mixed-workload program (tests/vm_programs.py program_bench_loop, one seed per execution) as a host of the reference holds it —
the WitnessOracle's per-getter FIFOs in call order (src/main_vm/witness_oracle.rs:45-91: memory reads, storage reads, refunds,
rollback-queue witnesses, rollback tails for calls, popped callstack entries, decommit pages) + the closed-form input — produced
by the native restatement oracle/main_vm_native.py, plus the input commitment the native model expects for the first
`limit`-cycle chunk at 2^20 rows.  bench.py loads this file as DATA (it never imports oracle/ for its workload), turns the FIFOs
into the circuit's input streams with the product's own zk_pack_main_vm_witness, lets the device derive the carried state
(zk_cs_seed_window_async) and compares the public inputs it gets with the commitments stored here.

    python tests/golden/make_vm_bench_witness.py [--realistic]
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

N_EXEC, LOG2_ROWS = 64, 20
FIFOS = ("memory_reads", "storage_reads", "refunds", "rollback_queue_witness", "rollback_tails_for_call", "callstack", "decommit_pages")
WIDTH = dict(memory_reads=9, storage_reads=8, refunds=1, rollback_queue_witness=4, rollback_tails_for_call=4, callstack=54, decommit_pages=1)


def main(realistic=False):
    d, D = vp.defs()
    probe = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), 1 << 30, 1 << 28)
    probe.configure_main_vm(d)
    probe.main_vm_entry_point(1)
    probe.pad_and_shrink()
    st = probe.stats()
    limit = ((1 << LOG2_ROWS) - st["outer_slots"]) // st["loop_slots"]
    fifo = {k: [] for k in FIFOS}
    offsets = {k: [0] for k in FIFOS}
    tails = np.zeros((N_EXEC, 4), dtype=np.uint64)
    commits = np.zeros((N_EXEC, 4), dtype=np.uint64)
    families = np.zeros((N_EXEC, 16), dtype=np.uint32)
    for e in range(N_EXEC):
        ops, contracts = vp.program_bench_loop(D, e, realistic=realistic)
        vrun = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), limit)
        q = vp.oracle_queues(vrun, 0, limit)
        rows = dict(memory_reads=[list(v) + [p] for v, p in q.memory_reads], storage_reads=[list(v) for v in q.storage_reads],
                    refunds=[[r] for r in q.refunds], rollback_queue_witness=[list(v) for v in q.rollback_queue_witness],
                    rollback_tails_for_call=[list(v) for v in q.rollback_tails_for_call],
                    callstack=[list(c) + list(s) for c, s in q.callstack], decommit_pages=[[p] for p in q.decommit_pages])
        for k in FIFOS:
            fifo[k] += [[int(x) for x in r] for r in rows[k]]
            offsets[k].append(len(fifo[k]))
        for _, W in vrun.rows:
            families[e, W["_family"]] += 1
        tails[e] = np.array(vrun.rollback_tail_for_block, dtype=np.uint64)
        commits[e] = np.array(vp.expected_commitment(D, vrun, limit, 0), dtype=np.uint64)
        print(f"execution {e}: {limit} cycles, {[offsets[k][-1] - offsets[k][-2] for k in FIFOS]} oracle answers, commitment {[hex(int(x)) for x in commits[e]]}", flush=True)
    out = os.path.join(HERE, "vm_bench_witness_realistic.npz" if realistic else "vm_bench_witness.npz")
    arrays = {k: np.array(fifo[k], dtype=np.uint64).reshape(-1, WIDTH[k]) for k in FIFOS}
    arrays.update({k + "_offsets": np.array(offsets[k], dtype=np.int64) for k in FIFOS})
    np.savez_compressed(out, rollback_tail=tails, commitment=commits, limit=np.array([limit]), log2_rows=np.array([LOG2_ROWS]),
                        families_per_execution=families, **arrays)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(realistic="--realistic" in sys.argv)   # --realistic: the compiled-contract-like mix (vm_programs.program_bench_loop)
