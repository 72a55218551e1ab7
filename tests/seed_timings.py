"""Seeding (zk_cs_seed_carried_inputs) timings: plain cone vs its strand form, and that both write the same stream.
Lives under tests/ because its inputs come from the oracle's native restatements.  GPU box, repo root: python tests/seed_timings.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import zkgl
import test_gpu_full_size as T
from oracle import keccak_native as kn, sha256_native as shn

zkgl.init(0)


def run(name, cs, outer, raw, batch):
    cs.set_batch(batch)
    d_o = zkgl.DeviceBuffer.from_numpy(outer)
    cs.bind_inputs(False, d_o, outer.shape[0])
    res, out = {}, {"config": name, "instances": batch}
    for mode in ("0", "1"):
        os.environ["ZKGL_SEED_STRANDS"] = mode
        d_l = zkgl.DeviceBuffer.from_numpy(raw)
        cs.bind_inputs(True, d_l, raw.shape[0])
        t0 = time.perf_counter(); cs.seed_carried_inputs(d_l); zkgl.sync(); dt = time.perf_counter() - t0
        res[mode] = d_l.to_numpy()
        out["strands_s" if mode == "1" else "plain_s"] = round(dt, 3)
    del os.environ["ZKGL_SEED_STRANDS"]
    out["same_stream"] = bool(np.array_equal(res["0"], res["1"]))
    ok, f = cs.resolve_and_check()
    out["satisfied"] = bool(ok)
    print(json.dumps(out), flush=True)


cs, limit = T.fit(lambda c: c.configure_keccak(), lambda c, l: c.keccak256_round_function_entry_point(l), 20)
reqs, _ = T._keccak_requests(np.random.default_rng(0xC3), limit)
inst = kn.instance(reqs, limit)
B = 32
outer = np.array([inst["outer"]] * B, dtype=np.uint64).T.copy(); loop = np.array(inst["rows"] * B, dtype=np.uint64).T.copy()
raw = loop.copy(); raw[:kn.CARRIED, :] = 0
run("keccak256_round_function 2^20 rows", cs, outer, raw, B)
cs, limit = T.fit(lambda c: c.configure_sha256(), lambda c, l: c.sha256_round_function_entry_point(l), 20)
rng = np.random.default_rng(1)
msgs = [bytes(rng.integers(0, 256, size=64 * 8 - 9, dtype=np.uint8)) for _ in range(limit // 8)]
reqs = [shn.request(m, 1 + 2 * i, 10 + i, 0, 9000 + i, i) for i, m in enumerate(msgs)]
inst = shn.instance(reqs, limit)
outer = np.array([inst["outer"]] * B, dtype=np.uint64).T.copy(); loop = np.array(inst["rows"] * B, dtype=np.uint64).T.copy()
raw = loop.copy(); raw[:shn.CARRIED, :] = 0
run("sha256_round_function 2^20 rows", cs, outer, raw, B)
