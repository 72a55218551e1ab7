"""tests/emu/run_case.py <case> [<case> ...] — one process = one library (ZKGL_LIB, read when zkgl loads): run lane-harness cases against the oracle
interpreter and print one JSON line per (case, form).  Environment: EMU_VARIANT / EMU_DEFS name the harness build that matches ZKGL_LIB
(tests/emu/build.sh), plus whatever record-time switches the variant needs (ZKGL_SHA4_MACRO=1 ...).  Used by tests/test_lane_harness.py."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402

import lane_harness as LH  # noqa: E402
import zkgl  # noqa: E402
from oracle import zko  # noqa: E402

VAR = os.environ.get("EMU_VARIANT", "")
DEFS = os.environ.get("EMU_DEFS", "").split()


def compare(name, cs, outer, loop, batch, table_rows, forms=(False, True), expect_public=None):
    run = zko.CircuitRun(cs.export(False), cs.export(True), batch, table_rows)
    run.resolve(outer, loop)
    bad, _ = run.check()
    for strands in forms:
        r = LH.resolve(cs, outer, loop, batch, strands=strands, variant=VAR, defs=DEFS)
        out = {"case": name, "strands": bool(strands), "outer_equal": bool(np.array_equal(r.oc, run.oc)),
               "loop_equal": bool(np.array_equal(r.lc, run.lc)) if (r.lc.size and cs.stats()["limit"]) else True, "fused_failure": bool(r.fused_failure), "oracle_violations": int(bad),
               "loop_ops": int(cs.stats()["loop_ops"]), "features": zkgl.build_features()}
        if expect_public is not None:
            out["public_equal"] = bool(all([int(x) for x in r.public[i]] == list(expect_public[i]) for i in range(batch)))
        print(json.dumps(out), flush=True)


def case_ram():
    from helpers import ram_cs, random_instances
    from oracle import ram_native as rn
    limit = 8
    cs = ram_cs(limit)
    insts = random_instances(77, 5, 5, limit)
    outer, loop = rn.pack_streams(insts, limit)
    compare("ram", cs, outer, loop, len(insts), 65536, expect_public=[i["commitment"] for i in insts])


def case_vm():
    import vm_programs as vp
    from oracle import main_vm_native as vn
    d, D = vp.defs()
    ops = vp.program_arith(D)
    vm_limit = 16
    n_inst = (len(ops) + 6 + vm_limit - 1) // vm_limit
    vcs = vp.vm_cs(vm_limit)
    vrun = vn.VmRun(D, vp.make_world_factory(D, ops), n_inst * vm_limit)
    raw_outer, raw, _ = vp.pack_through_the_c_abi(vcs, vrun, vm_limit, n_inst)
    rows = int(sum(t["n_rows"] for t in zko.parse_export(vcs.export(False))["tables"]))
    seeded = zko.CircuitRun(vcs.export(False), vcs.export(True), n_inst, rows).seed(raw_outer, raw)
    compare("vm", vcs, raw_outer, seeded, n_inst, rows, forms=(False,))


def case_keccak():
    from test_keccak_fsm_host import REFERENCE_CASES, TABLE_ROWS, reference_case, streams
    insts = [reference_case(l, u)[1] for l, u in REFERENCE_CASES[:4]]
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_keccak(); cs.keccak256_round_function_entry_point(2); cs.pad_and_shrink()
    outer, loop = streams(insts, 2)
    compare("keccak", cs, outer, loop, len(insts), TABLE_ROWS, expect_public=[i["public_input"] for i in insts])


def _sha(ref):
    from test_sha256_host import loop_stream
    rng = np.random.default_rng(5)
    msgs = [bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in (56, 64, 119)]
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_sha256(ref); cs.sha256_blocks_entry_point(2); cs.pad_and_shrink()
    outer = np.zeros((0, len(msgs)), dtype=np.uint64)
    rows = 3 * 4096 + 2 * 16 if ref else 65536 * 3 + 7 * 256
    loop = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), rows).seed(outer, loop_stream(msgs, 2))
    import hashlib
    compare("sha4" if ref else "sha", cs, outer, loop, len(msgs), rows, expect_public=[list(hashlib.sha256(m).digest()) for m in msgs])


def case_sha():
    _sha(False)


def case_sha4():
    _sha(True)


def case_iszero():
    """zero-checks of LARGE operands (they leave the small-inverse table): 21 of them — two full batches of eight, a tail at the end of the program —
    one whose inverse a later op reads (cannot be deferred), zeros and small values in between"""
    from helpers import Rec
    from zkgl import GATE as G
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(40, 0, 8, 4))
    for k in ("CONST", "FMA", "ZEROCHECK", "PUBLIC_INPUT"):
        cs.allow_gate(G[k])
    r = Rec(cs)
    xs = [r.inp() for _ in range(21)]
    flags = []
    for i, x in enumerate(xs):
        f, aux = r.iszero(x)
        flags.append(f)
        if i == 4:
            flags.append(r.fma(1, aux, x, 0, x))       # reads the inverse: x * x^-1
    acc = flags[0]
    one = r.const(1)
    for f in flags[1:]:
        acc = r.fma(3, acc, one, 1, f)
    cs.place_gate(G["PUBLIC_INPUT"], [acc])
    cs.pad_and_shrink()
    rng = np.random.default_rng(9)
    B = 6
    inp = (rng.integers(1 << 40, 1 << 63, size=(21, B), dtype=np.uint64))
    inp[3, :] = 0; inp[7, 1] = 0; inp[9, :] = 5; inp[10, 2] = zkgl.P - 3; inp[20, 4] = 0
    compare("iszero", cs, inp, np.zeros((0, B), dtype=np.uint64), B, 1, forms=(False,))


def case_adversarial():
    """the fused-mode failure flag of the witness kernels: a macro-op input that is not a byte; a SELECT by 2 with different branches"""
    import test_macro_ownership as MO
    cs = MO.keccak_circuit(MO.honest)
    rng = np.random.default_rng(13)
    B = 3
    inp = rng.integers(0, 256, size=(200, B), dtype=np.uint64)
    empty = np.zeros((0, B), dtype=np.uint64)
    for name, mutate in (("clean", None), ("not_a_byte", (7, 1, 256))):
        x = inp.copy()
        if mutate:
            x[mutate[0], mutate[1]] = mutate[2]
        r = LH.resolve(cs, x, empty, B, strands=False, variant=VAR, defs=DEFS)
        print(json.dumps({"case": "adversarial_" + name, "strands": False, "fused_failure": bool(r.fused_failure),
                          "failing_lane": None if not r.fused_failure else int(min(v for v in r.fail[:6] if v != 0xFFFFFFFFFFFFFFFF) >> 32)}), flush=True)


def case_verdicts():
    """resolve_and_check's verdict, fused and stored, from the product's witness + check kernel source: an outsider's gate on a macro-op output (honest /
    forged), adversarial inputs — against the oracle checker.  (VERDICT r4 'mirror by trust': the forged gate must be judged in the fused mode too.)"""
    import test_macro_ownership as MO
    rng = np.random.default_rng(13)
    B = 5
    empty = np.zeros((0, B), dtype=np.uint64)
    for kind, make, n_in in (("keccak", MO.keccak_circuit, 200), ("sha256", MO.sha_circuit, 96), ("sha256_reference_tables", lambda e=None: MO.sha_circuit(e, True), 96)):
        inp = rng.integers(0, 256, size=(n_in, B), dtype=np.uint64)
        for extra_name, extra in (("honest", MO.honest), ("forged", MO.forged)):
            cs = make(extra)
            run = zko.CircuitRun(cs.export(False), cs.export(True), B, 1 << 20)
            for mut_name, mut in (("clean", None), ("not_a_byte", (n_in // 2, 3, 1 << 40))):
                x = inp.copy()
                if mut:
                    x[mut[0], mut[1]] = mut[2]
                run.resolve(x, empty)
                nbad, _ = run.check()
                LH.resolve(cs, x, empty, B, strands=False, variant=VAR, defs=DEFS)
                ok_f, lane_f = LH.check(cs, False, VAR)
                ok_s, lane_s = LH.check(cs, True, VAR)
                print(json.dumps({"case": f"verdict_{kind}_{extra_name}_{mut_name}", "oracle_accepts": nbad == 0, "fused_accepts": ok_f, "stored_accepts": ok_s,
                                  "fused_lane": lane_f, "stored_lane": lane_s}), flush=True)


def case_fuzz_verdicts():
    """tests/test_fused_differential.py's programs (random mixes of exactly the constructs whose soundness rests on a non-mirrored gate) x adversarial
    input vectors, case by case: verdict(fused step) == verdict(every relation from the store) == verdict(oracle checker) — the GPU test's
    comparison, with the product's witness + check kernel SOURCE on the lane harness in place of the device"""
    import test_fused_differential as FD
    n_prog = int(os.environ.get("EMU_FUZZ_PROGRAMS", "6")); n_cases = int(os.environ.get("EMU_FUZZ_CASES", "60"))
    total = agree = rejected = 0
    for p in range(n_prog):
        pr, outer, loop = FD.program_and_cases(p, n_cases)
        cs = pr.cs
        want = FD.oracle_verdicts(pr, outer, loop)
        eo, el = cs.export(False), cs.export(True)
        for c in range(n_cases):
            oc = np.ascontiguousarray(outer[:, c:c + 1])
            lo = np.ascontiguousarray(loop[:, c * pr.limit:(c + 1) * pr.limit]) if pr.limit else np.zeros((0, 1), dtype=np.uint64)
            if pr.limit:
                lo = zko.CircuitRun(eo, el, 1, 256).seed(oc, lo)
            LH.resolve(cs, oc, lo, 1, strands=False, variant=VAR, defs=DEFS)
            ok_f, _ = LH.check(cs, False, VAR)
            ok_s, _ = LH.check(cs, True, VAR)
            total += 1; agree += int(ok_f == ok_s == bool(want[c])); rejected += int(not want[c])
            if not (ok_f == ok_s == bool(want[c])):
                print(json.dumps({"case": "fuzz_disagreement", "program": p, "input": c, "oracle_accepts": bool(want[c]), "fused_accepts": ok_f, "stored_accepts": ok_s}), flush=True)
    print(json.dumps({"case": "fuzz_verdicts", "programs": n_prog, "cases": total, "agree": agree, "oracle_rejects": rejected}), flush=True)


if __name__ == "__main__":
    for c in sys.argv[1:]:
        globals()["case_" + c]()
