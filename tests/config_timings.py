"""Timings of BASELINE.json's other configurations at full size on one GPU (C1, C3, C4, C5): one fused
resolve_and_check per configuration (after one warm-up), inputs from the native restatements used by the parity tests.
Prints one JSON line per configuration.  It lives under tests/ because its input generators are the oracle's native restatements (test infrastructure).
usage (GPU box, repo root): python tests/config_timings.py            (CONFIGS=C3k,C5 selects: C1 C3k C3s C3s4 C4s C4l C5)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file is tests/config_timings.py)
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import zkgl
import test_gpu_full_size as T
from oracle import keccak_native as kn, log_sorter_native as ln, ram_native as rn, sha256_native as shn, storage_native as sn, eip4844_native as en

zkgl.init(0)


def timed(name, cs, outer, loop, batch, seed_carried=1, given=(), stream_x=0):
    """one configuration: seeding of the carried words from the raw stream (timed, compared with the native restatement's words), then
    one fused resolve_and_check (after a warm-up); from-raw rate = constraints / (seeding + step)"""
    cs.set_batch(batch)
    carried = cs.carried_words()
    raw = loop.copy()
    raw[[w for w in carried if w not in given], :] = 0
    cs.set_seed_given(list(given))
    if given:
        pass                    # words the host packer fills (ram: the queue heads = the witness's previous tails)
    d_o, d_l = zkgl.DeviceBuffer.from_numpy(outer), zkgl.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    t_seed, seeded_ok = None, None
    if seed_carried and carried:
        cs.seed_carried_inputs(d_l)                       # warm-up (allocations)
        d_l2 = zkgl.DeviceBuffer.from_numpy(raw)
        cs.bind_inputs(True, d_l2, loop.shape[0])
        t0 = time.perf_counter(); cs.seed_carried_inputs(d_l2); zkgl.sync(); t_seed = time.perf_counter() - t0
        seeded_ok = bool(np.array_equal(d_l2.to_numpy().reshape(loop.shape), loop))
        d_l = d_l2
    t_stream = None
    if stream_x and t_seed is not None:                   # one seeding pass over a stream of stream_x batches (the pass is latency-bound:
        S = stream_x * batch                              # one workgroup per instance walks its cycles; more instances ride along for free)
        limit = loop.shape[1] // batch
        o_s = np.tile(outer, (1, stream_x))
        l_s = np.tile(raw.reshape(raw.shape[0], batch, limit), (1, stream_x, 1)).reshape(raw.shape[0], S * limit)
        d_os, d_ls = zkgl.DeviceBuffer.from_numpy(o_s), zkgl.DeviceBuffer.from_numpy(l_s)
        cs.seed_stream(S, d_os, d_ls); zkgl.sync()
        d_ls = zkgl.DeviceBuffer.from_numpy(l_s)
        t0 = time.perf_counter(); cs.seed_stream(S, d_os, d_ls); zkgl.sync(); t_stream = time.perf_counter() - t0
        got = d_ls.to_numpy().reshape(raw.shape[0], stream_x, batch * limit)
        seeded_ok = seeded_ok and all(np.array_equal(got[:, k], loop) for k in range(stream_x))
        del d_os, d_ls, got
    ok, f = cs.resolve_and_check(); assert ok or os.environ.get("ZKGL_STUB_RUN"), f
    t0 = time.perf_counter(); ok, f = cs.resolve_and_check(); dt = time.perf_counter() - t0
    st = cs.stats()
    total = batch * st["constraints_per_instance"]
    print(json.dumps({"config": name, "instances": batch, "rows_per_instance": st["rows_per_instance"], "constraints_per_instance": st["constraints_per_instance"],
                      "step_ms": round(1e3 * dt, 2), "constraints_per_s": round(total / dt), "rows_per_s": round(batch * st["rows_per_instance"] / dt),
                      "k_witness_loop_ms": round(cs.last_ms(1), 2), "carried_words": len(carried), "seed_s": None if t_seed is None else round(t_seed, 4),
                      "seeded_equals_native": seeded_ok,
                      "seed_stream": None if t_stream is None else {"instances": stream_x * batch, "seed_s": round(t_stream, 4), "seed_s_per_step": round(t_stream / stream_x, 4),
                                                                    "constraints_per_s_from_raw": round(total / (dt + t_stream / stream_x))},
                      "constraints_per_s_from_raw": None if t_seed is None else round(total / (dt + t_seed)),
                      "values_GBps_loop_kernel": round(batch * st["limit"] * st["cells_written_loop"] * 8 / (cs.last_ms(1) * 1e-3) / 1e9, 1) if cs.last_ms(1) > 0 else None}), flush=True)


rng = np.random.default_rng(1)
# CONFIG_TIMINGS_MAX_INSTANCES=n: a DRY RUN of this tool with at most n instances per configuration (the emulated device runs every line of it that way,
# tests/emu/README.md); the numbers of such a run mean nothing
_CAP = int(os.environ.get("CONFIG_TIMINGS_MAX_INSTANCES", "0"))
nb = lambda n: min(n, _CAP) if _CAP else n
WANT = os.environ.get("CONFIGS")
want = lambda tag: WANT is None or tag in WANT.split(",")
# C1: ram_permutation 2^16 rows, 512 instances
if want("C1"):
    cs, limit = T.fit(lambda c: c.configure_ram_permutation(), lambda c, l: c.ram_permutation_entry_point(l), 16)
    u, s, nd = rn.random_ram_witness(rng, limit, n_cells=64)
    inst = rn.instance(u, s, limit, nd)
    outer, loop = rn.pack_streams([inst] * nb(512), limit)
    timed("C1 ram_permutation 2^16 rows", cs, outer, loop, nb(512))
    timed("C1 ram_permutation 2^16 rows, heads from the witness's previous tails", cs, outer, loop, nb(512), given=zkgl.ram_head_words())
# C3
if want("C3k"):
    cs, limit = T.fit(lambda c: c.configure_keccak(), lambda c, l: c.keccak256_round_function_entry_point(l), 20)
    reqs, _ = T._keccak_requests(np.random.default_rng(0xC3), limit)
    inst = kn.instance(reqs, limit)
    B = nb(128)
    outer = np.array([inst["outer"]] * B, dtype=np.uint64).T.copy(); loop = np.array(inst["rows"] * B, dtype=np.uint64).T.copy()
    timed("C3 keccak256_round_function 2^20 rows", cs, outer, loop, B, stream_x=4)
    timed("C3 keccak256_round_function 2^20 rows, every carried word from the witness's queue states (zk_pack_keccak_witness_tails)", cs, outer, loop, B, given=list(range(kn.CARRIED)))
if want("C3s"):
    B = nb(128)
    cs, limit = T.fit(lambda c: c.configure_sha256(), lambda c, l: c.sha256_round_function_entry_point(l), 20)
    msgs = [bytes(rng.integers(0, 256, size=64 * 8 - 9, dtype=np.uint8)) for _ in range(limit // 8)]
    reqs = [shn.request(m, 1 + 2 * i, 10 + i, 0, 9000 + i, i) for i, m in enumerate(msgs)]
    inst = shn.instance(reqs, limit)
    outer = np.array([inst["outer"]] * B, dtype=np.uint64).T.copy(); loop = np.array(inst["rows"] * B, dtype=np.uint64).T.copy()
    timed("C3 sha256_round_function 2^20 rows", cs, outer, loop, B, stream_x=4)
    timed("C3 sha256_round_function 2^20 rows, every carried word from the witness's queue states (zk_pack_sha256_witness_tails)", cs, outer, loop, B, given=list(range(shn.CARRIED)))
if want("C3s4"):   # the SAME circuit under the reference's own table set (src/code_unpacker_sha256/mod.rs:554-566: width-4 lookups, Maj4 / TriXor4 / Ch4 / Split4BitChunk<1,2>)
    B = nb(128)      # the compression is ONE macro-op (ZK_OP_SHA256_ROUNDS a = 1: the default recording of this table set since round 6); ZKGL_SHA4_MACRO=0 interprets it op by op
    cs, limit = T.fit(lambda c: c.configure_sha256(True), lambda c, l: c.sha256_round_function_entry_point(l), 20)
    msgs = [bytes(rng.integers(0, 256, size=64 * 8 - 9, dtype=np.uint8)) for _ in range(limit // 8)]
    reqs = [shn.request(m, 1 + 2 * i, 10 + i, 0, 9000 + i, i) for i, m in enumerate(msgs)]
    inst = shn.instance(reqs, limit)
    outer = np.array([inst["outer"]] * B, dtype=np.uint64).T.copy(); loop = np.array(inst["rows"] * B, dtype=np.uint64).T.copy()
    form = "op by op" if os.environ.get("ZKGL_SHA4_MACRO") == "0" else "macro-op ZK_OP_SHA256_ROUNDS a = 1"
    timed(f"C3 sha256_round_function 2^20 rows, REFERENCE table set (4-bit chunks, {form}), every carried word from the witness's queue states", cs, outer, loop, B, given=list(range(shn.CARRIED)))
# C4 (4 instances on one GPU here; BASELINE shards them over 4 GPUs)
if want("C4s"):
    cs, limit = T.fit(lambda c: c.configure_storage_validity(), lambda c, l: c.sort_and_deduplicate_storage_access_entry_point(l, True), 22)
    u, s = sn.random_storage_witness(np.random.default_rng(0xC4), limit - 3, n_cells=512)
    inst = sn.instance(u, s, limit)
    outer, loop = sn.pack_streams([inst] * nb(4), limit)
    timed("C4 storage_validity 2^22 rows", cs, outer, loop, nb(4))
    g = [w for w in range(67) if not 2 <= w < 6]
    timed("C4 storage_validity 2^22 rows, integer state walked by the packer (previous tails), output chain on the device", cs, outer, loop, nb(4), given=[w for w in g if not 17 <= w < 21])
    timed("C4 storage_validity 2^22 rows, integer state walked by the packer, output tails from the host", cs, outer, loop, nb(4), given=g)
    o1, l1 = sn.pack_streams([inst], limit)
    timed("C4 storage_validity 2^22 rows, ONE instance (one GPU of BASELINE's 4)", cs, o1, l1, 1)
if want("C4l"):
    cs, limit = T.fit(lambda c: c.configure_log_sorter(), lambda c, l: c.sort_and_deduplicate_events_entry_point(l), 22)
    u, s = ln.random_events(np.random.default_rng(0xC4 + 1), int(limit / 1.1) - 8, rollback_frac=0.1)
    inst = ln.instance(u, s, limit)
    outer, loop = ln.pack_streams([inst] * nb(4), limit)
    timed("C4 log_sorter 2^22 rows", cs, outer, loop, nb(4))
    g = [w for w in range(57) if not 1 <= w < 5]
    timed("C4 log_sorter 2^22 rows, integer state walked by the packer (previous tails), output chain on the device", cs, outer, loop, nb(4), given=[w for w in g if not 15 <= w < 19])
    timed("C4 log_sorter 2^22 rows, integer state walked by the packer, output tails from the host", cs, outer, loop, nb(4), given=g)
    o1, l1 = ln.pack_streams([inst], limit)
    timed("C4 log_sorter 2^22 rows, ONE instance", cs, o1, l1, 1)
# C5: 8 blobs (BASELINE: one per GPU)
if want("C5"):
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4), 1 << 21, 1 << 28)
    cs.configure_eip_4844(); cs.eip_4844_entry_point(4096); cs.pad_and_shrink()
    insts = []
    for k in range(nb(8)):
        r = np.random.default_rng(0xC5 + k)
        insts.append(en.instance(bytes(r.integers(0, 256, size=31 * 4096, dtype=np.uint8)), b"\x01" + bytes(r.integers(0, 256, size=31, dtype=np.uint8)), 4096))
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r_ for i in insts for r_ in i["rows"]], dtype=np.uint64).T.copy()
    timed("C5 eip_4844 8 blobs x 4096 chunks", cs, outer, loop, nb(8), stream_x=nb(8))
    timed("C5 eip_4844 8 blobs x 4096 chunks, the 217 carried words from the host packer (zk_pack_eip4844_witness_full)", cs, outer, loop, nb(8), given=list(range(217)))
