"""GPU box, repo root: seeding pass and step of the SURVEY 8(f) circuits (demux_log_queue, sort_decommittment_requests,
code_unpacker_sha256, linear_hasher) at a common size -> one JSON line per circuit.  Inputs come from the oracle's native restatements
(test infrastructure), so this lives with the measurement tools, not in the product.
usage: python tools/f_timings.py [log2_rows=18] [batch=32]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import zkgl
import test_gpu_full_size as T
from oracle import code_unpacker_native as cn, decommit_native as dn, demux_native as mn, linear_hasher_native as hn
from oracle.decommit_native import dq

LOG2 = int(sys.argv[1]) if len(sys.argv) > 1 else 18
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
zkgl.init(0)
rng = np.random.default_rng(0xF)


def timed(name, cs, insts, carried, limit):
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    raw = loop.copy(); raw[:carried] = 0
    cs.set_batch(len(insts))
    d_o = zkgl.DeviceBuffer.from_numpy(outer)
    cs.bind_inputs(False, d_o, outer.shape[0])
    d_l = zkgl.DeviceBuffer.from_numpy(raw); cs.bind_inputs(True, d_l, raw.shape[0]); cs.seed_carried_inputs(d_l); zkgl.sync()
    d_l = zkgl.DeviceBuffer.from_numpy(raw); cs.bind_inputs(True, d_l, raw.shape[0])
    t0 = time.perf_counter(); cs.seed_carried_inputs(d_l); zkgl.sync(); t_seed = time.perf_counter() - t0
    same = bool(np.array_equal(d_l.to_numpy().reshape(raw.shape), loop))
    ok, f = cs.resolve_and_check(); assert ok, f
    t0 = time.perf_counter(); ok, f = cs.resolve_and_check(); dt = time.perf_counter() - t0
    st = cs.stats()
    # the same instances with every carried word written by the host packer from the witness's queue states (demux: zk_pack_demux_witness_tails)
    given_ms = None
    # (demux, code_unpacker: all words; sort_decommits: all but the four grand-product words, which k_decommit_seed scans)
    if os.environ.get("F_GIVEN", "1") == "1" and name != "linear_hasher":
        device_words = [1, 2, 3, 4] if name == "sort_decommittment_requests" else []
        cs.set_seed_given([w for w in range(carried) if w not in device_words])
        part = loop.copy(); part[device_words] = 0
        d_g = zkgl.DeviceBuffer.from_numpy(part); cs.bind_inputs(True, d_g, part.shape[0]); cs.seed_carried_inputs(d_g); zkgl.sync()
        d_g = zkgl.DeviceBuffer.from_numpy(part); cs.bind_inputs(True, d_g, part.shape[0])
        t0 = time.perf_counter(); cs.seed_carried_inputs(d_g); zkgl.sync(); given_ms = round(1e3 * (time.perf_counter() - t0), 3)
        assert np.array_equal(d_g.to_numpy().reshape(loop.shape), loop)
        ok, f = cs.resolve_and_check(); assert ok, f
        cs.set_seed_given([])
    print(json.dumps({"circuit": name, "seed_ms_queue_states_from_the_witness": given_ms, "instances": len(insts), "limit": limit, "rows_per_instance": st["rows_per_instance"], "seed_ms": round(1e3 * t_seed, 2),
                      "step_ms": round(1e3 * dt, 2), "seeded_equals_native": same, "seed_ops": st["seed_ops"], "loop_ops": st["loop_ops"]}), flush=True)


def lq(**k):
    from test_linear_hasher_host import log_query
    return log_query(**k)


which = os.environ.get("F_CIRCUITS", "demux,decommit,unpacker,hasher").split(",")
if "demux" in which:
    cs, limit = T.fit(lambda c: c.configure_demux_log_queue(), lambda c, l: c.demultiplex_storage_logs_entry_point(l), LOG2)
    from test_demux_host import random_queries
    insts = [mn.instance(random_queries(np.random.default_rng(100 + k), limit - 3), limit) for k in range(B)]
    timed("demux_log_queue", cs, insts, mn.CARRIED, limit)
if "decommit" in which:
    cs, limit = T.fit(lambda c: c.configure_sort_decommits(), lambda c, l: c.sort_and_deduplicate_code_decommittments_entry_point(l), LOG2)
    insts = []
    while len(insts) < B:
        u, s = dn.random_decommits(rng, max(1, limit // 3), max_repeats=3)
        if len(u) <= limit:
            insts.append(dn.instance(u, s, limit))
    timed("sort_decommittment_requests", cs, insts, dn.CARRIED, limit)
if "unpacker" in which:
    cs, limit = T.fit(lambda c: c.configure_code_unpacker(), lambda c, l: c.unpack_code_into_memory_entry_point(l), LOG2)
    from test_code_unpacker_host import random_code
    insts = []
    for k in range(B):
        reqs, rounds = [], 0
        while True:
            n = 2 * int(rng.integers(4, 40)) + 1
            if rounds + (n + 1) // 2 > limit:
                break
            w = random_code(rng, n)
            reqs.append((dq(cn.versioned_hash(w), 2048 + 8 * len(reqs), 1, 5 + len(reqs)), w))
            rounds += (n + 1) // 2
        insts.append(cn.instance(reqs, limit))
    timed("code_unpacker_sha256", cs, insts, cn.CARRIED, limit)
if "hasher" in which:
    from test_linear_hasher_host import random_messages
    lim = 17 * max(1, int(os.environ.get("HASHER_PERIODS", "2")))
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4), 1 << 24, 1 << 28)
    cs.configure_linear_hasher(); cs.linear_hasher_entry_point(lim); cs.pad_and_shrink()
    insts = [hn.instance(random_messages(np.random.default_rng(300 + k), lim - 2), lim) for k in range(B)]
    timed("linear_hasher", cs, insts, hn.CARRIED, lim)
