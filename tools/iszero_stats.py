"""How many of main_vm's zero-checks leave the small-inverse table, per wavefront, on the bench fixtures (no GPU: the lane harness of tests/emu runs the
product's interpreter source on one full-size instance per fixture; the statistic is read from the resolved store).  -> profiles/r5_iszero_stats.json
usage: python tools/iszero_stats.py > profiles/r5_iszero_stats.json"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    sys.path.insert(0, p)
import numpy as np
import bench, zkgl, lane_harness as LH
from oracle import zko

cs, limit = bench.build_main_vm_cs(zkgl, 20)
rows = int(sum(t["n_rows"] for t in zko.parse_export(cs.export(False))["tables"]))
out = {"circuit": "main_vm, 2^20 rows", "cycles_per_instance": limit, "small_inverse_table_entries": 4096}
for name, path in bench.FIXTURES.items():
    outer, loop, expect = bench.main_vm_streams(zkgl, cs, limit, 2, fixture=path)
    t = time.time()
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 2, rows).seed(outer, loop)
    r = LH.resolve(cs, outer, seeded, 2, variant=os.environ.get('EMU_VARIANT', ''), defs=os.environ.get('EMU_DEFS', '').split())
    st = (C.c_uint64 * 4)()
    LH.lib(os.environ.get('EMU_VARIANT', '')).zk_emu_iszero_stats(cs._h, st)
    gp = (C.c_uint64 * 4)()
    LH.lib(os.environ.get('EMU_VARIANT', '')).zk_emu_gated_p2_stats(cs._h, gp)
    mr = (C.c_uint64 * 3)()
    LH.lib(os.environ.get('EMU_VARIANT', '')).zk_emu_gated_p2_merged_rounds(cs._h, mr)
    out[name] = {"instances": 2, "gated_permutation_levels": int(mr[0]), "permutations_a_level_merged_kernel_runs_per_wavefront": round(mr[2] / mr[1], 2),
                 "gated_witness_only_permutations_per_cycle": int(gp[0]), "gated_permutations_run_per_wavefront": round(gp[2] / gp[1], 2),
                 "gated_permutations_needed_per_lane": round(gp[3] / (2 * limit), 3), "wavefronts": int(st[1]), "zero_checks_per_cycle": int(st[0]),
                 "zero_checks_taking_the_chain_per_wavefront": round(st[2] / st[1], 2), "per_lane_alone": round(st[3] / (2 * limit), 2),
                 "commitments_equal_fixture": bool(expect is not None and [int(x) for x in r.public[0]] == [int(x) for x in expect[0]]),
                 "fused_failure": r.fused_failure, "seconds": round(time.time() - t, 1)}
out["reading"] = ("gated permutations: a wavefront runs a witness-only permutation as soon as ONE of its 64 lanes (consecutive cycles, different opcodes) has its execute flag on; "
                  "round 5's -DZKGL_P2_MERGE build (deleted unmeasured in round 6, recoverable from commit f8fecc7) put the mutually independent permutations of a dependency level under one header and runs one permutation per round (a round = every lane's next member that is on).  "
                  "Zero-checks: a wavefront runs the 72-multiplication x^(p-2) chain for a zero-check as soon as ONE of its 64 lanes holds |x| >= 4096; round 5's -DZKGL_BATCH_INV build (same fate) replaced k such "
                  "chains by one chain + 3 k multiplications, eight at a time")
print(json.dumps(out, indent=1))
