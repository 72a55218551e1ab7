"""Host side of the packers with the witness's queue states, at BASELINE C3's full size (2^20 rows per instance): how long the C packer takes
per instance (one core) for sha256_round_function and keccak256_round_function, against the device seeding pass it replaces.  Runs on the
CPU (no GPU needed: the circuit is only recorded to learn `limit`).  Inputs come from the oracle's native restatements (test infrastructure),
so this lives with the measurement tools.   usage: python tools/host_pack_timings.py > profiles/r4_host_pack_timings.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import zkgl
import test_gpu_full_size as T
import test_witness_pack as W
from oracle import keccak_native as kn, sha256_native as shn, zko
from oracle.storage_native import encode

out = {}


def tails(inst, reqs, o, req_at, mem_at):
    head, prev = [int(v) for v in o[req_at:req_at + 4]], []
    for r in reqs:
        prev.append(head)
        head = zko.queue_tail4_push20(head, encode(r["query"]))
    mt, mtails = [int(v) for v in o[mem_at + 12:mem_at + 24]], []
    for q in inst["pushed"]:
        mt = zko.queue_full_push(mt, zko.memory_query_encode(q))
        mtails.append(mt)
    return np.array(prev or [[0] * 4], dtype=np.uint64), np.array(mtails, dtype=np.uint64).reshape(-1, 12)


def reads_array(vals):
    ra = ((zkgl.C.c_uint32 * 8) * max(len(vals), 1))()
    for dst, v in zip(ra, vals):
        dst[:] = [(v >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
    return ra


# ---- sha256
cs, limit = T.fit(lambda c: c.configure_sha256(), lambda c, l: c.sha256_round_function_entry_point(l), 20)
rng = np.random.default_rng(1)
msgs = [bytes(rng.integers(0, 256, size=64 * 8 - 9, dtype=np.uint8)) for _ in range(limit // 8)]
reqs = [shn.request(m, 1 + 2 * i, 10 + i, 0, 9000 + i, i) for i, m in enumerate(msgs)]
inst = shn.instance(reqs, limit)
o = inst["outer"]
w = zkgl.Sha256RoundFunctionWitness()
w.start_flag = 1
w.initial_log_queue_state, w.initial_memory_queue_state = W._q4(o[1:10]), W._q12(o[10:35])
n_popped = len(reqs) - len(inst["rest"][0])
qa = (zkgl.LogQueryWitness * max(n_popped, 1))(*[W._lq(r["query"]) for r in reqs[:n_popped]])
rd = [v for r in reqs for v in r["reads"]]
ra = reads_array(rd)
w.requests_queue_witness, w.n_requests, w.memory_reads_witness, w.n_reads = qa, n_popped, ra, len(rd)
w.hidden_fsm_output.log_queue_state = W._q4(inst["fsm_out"]["req"])
prev, mt = tails(inst, reqs[:n_popped], o, 1, 10)
outer = np.zeros((87, 1), dtype=np.uint64); loop = np.zeros((112, limit), dtype=np.uint64)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); zkgl.pack_sha256_witness_tails(w, limit, 0, outer, loop, prev, mt); ts.append(time.perf_counter() - t0)
el = np.array(inst["rows"], dtype=np.uint64).T
out["sha256_round_function"] = {"limit": limit, "requests": n_popped, "memory_pushes": int(mt.shape[0]), "pack_ms_per_instance_one_core": round(1e3 * min(ts), 3),
                                "equals_native_stream": bool(np.array_equal(loop, el)), "device_seeding_pass_it_replaces_ms_128_instances": 38.0}
cs.close()

# ---- keccak256
cs, limit = T.fit(lambda c: c.configure_keccak(), lambda c, l: c.keccak256_round_function_entry_point(l), 20)
reqs, _ = T._keccak_requests(np.random.default_rng(0xC3), limit)
inst = kn.instance(reqs, limit)
o = inst["outer"]
w = zkgl.KeccakRoundFunctionWitness()
w.start_flag = 1
w.initial_log_queue_state, w.initial_memory_queue_state = W._q4(o[1:10]), W._q12(o[10:35])
n_popped = len(reqs) - len(inst["rest"][0])
qa = (zkgl.LogQueryWitness * max(n_popped, 1))(*[W._lq(r["query"]) for r in reqs[:n_popped]])
rd = [v for r in reqs for v in r["reads"]]
ra = reads_array(rd)
w.requests_queue_witness, w.n_requests, w.memory_reads_witness, w.n_reads = qa, n_popped, ra, len(rd)
w.hidden_fsm_output.log_queue_state = W._q4(inst["fsm_out"]["req"])
prev, mt = tails(inst, reqs[:n_popped], o, 1, 10)
outer = np.zeros((474, 1), dtype=np.uint64); loop = np.zeros((507, limit), dtype=np.uint64)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); zkgl.pack_keccak_witness_tails(w, limit, 0, outer, loop, prev, mt); ts.append(time.perf_counter() - t0)
el = np.array(inst["rows"], dtype=np.uint64).T
out["keccak256_round_function"] = {"limit": limit, "requests": n_popped, "memory_pushes": int(mt.shape[0]), "pack_ms_per_instance_one_core": round(1e3 * min(ts), 3),
                                   "equals_native_stream": bool(np.array_equal(loop, el)), "device_seeding_pass_it_replaces_ms_128_instances": 17.6}
# ---- main_vm (bench fixture: 64 executions x 2 352 cycles): the three forms of the packer
import bench
cs, limit = bench.build_main_vm_cs(zkgl, 20)
fx = np.load(bench.FIXTURE)
E = 8                                               # eight executions are enough for a per-instance figure
n_outer, n_loop = cs.input_words()
queues, cfs = [], []
for e in range(E):
    q = zkgl.VmOracleQueues()
    sl = {k: fx[k][fx[k + "_offsets"][e]:fx[k + "_offsets"][e + 1]] for k in bench._FIFOS}
    q.memory_reads = [(r[:8], r[8]) for r in sl["memory_reads"]]; q.storage_reads = list(sl["storage_reads"]); q.refunds = [r[0] for r in sl["refunds"]]
    q.rollback_queue_witness = list(sl["rollback_queue_witness"]); q.rollback_tails_for_call = list(sl["rollback_tails_for_call"])
    q.callstack = [(r[:42], r[42:]) for r in sl["callstack"]]; q.decommit_pages = [r[0] for r in sl["decommit_pages"]]
    q.freeze()
    cf = zkgl.VmClosedFormInput(); cf.start_flag = 1
    cf.rollback_queue_tail_for_block[:] = [int(x) for x in fx["rollback_tail"][e]]
    queues.append(q); cfs.append(cf)
outer = np.zeros((n_outer, E), dtype=np.uint64); loop = np.zeros((n_loop, E * limit), dtype=np.uint64)
for warm in range(2):                               # the first pass pays the page faults of the staging array
    t0 = time.perf_counter()
    for e in range(E):
        cs.pack_main_vm_witness(cfs[e], queues[e].view(), e, E, outer, loop)
    t_plain = (time.perf_counter() - t0) / E
states, perms_hash = [], 0
filled = np.zeros_like(loop)
t0 = time.perf_counter()
for e in range(E):
    arrs = (np.zeros((8 * limit, 12), dtype=np.uint64), np.zeros((2 * limit, 12), dtype=np.uint64), np.zeros((2 * limit, 4), dtype=np.uint64))
    st = zkgl.VmQueueStates.over(*arrs)
    cs.pack_main_vm_witness_states(cfs[e], queues[e].view(), st, e, E, outer, filled, zkgl.VM_PACK_FILL_STATE | zkgl.VM_PACK_RECORD_STATES)
    states.append((arrs, (st.used_memory_tails, st.used_decommit_tails, st.used_log_forward_tails))); perms_hash += st.host_permutations
t_hash = (time.perf_counter() - t0) / E
read = np.zeros_like(loop); perms_read = 0
t0 = time.perf_counter()
for e in range(E):
    arrs, used = states[e]
    st = zkgl.VmQueueStates.over(*[np.ascontiguousarray(a[:k]) for a, k in zip(arrs, used)])
    cs.pack_main_vm_witness_states(cfs[e], queues[e].view(), st, e, E, outer, read, zkgl.VM_PACK_STATES_FROM_WITNESS)
    perms_read += st.host_permutations
t_read = (time.perf_counter() - t0) / E
out["main_vm"] = {"limit": limit, "executions": E,
                  "pack_ms_per_instance_raw_stream_device_seeds": round(1e3 * t_plain, 2),
                  "pack_ms_per_instance_fill_state_host_hashes": round(1e3 * t_hash, 2), "host_permutations_per_instance_fill_state": perms_hash // E,
                  "pack_ms_per_instance_states_from_witness": round(1e3 * t_read, 2), "host_permutations_per_instance_states_from_witness": perms_read // E,
                  "queue_pushes_per_instance": [int(sum(u[1][k] for u in states) // E) for k in range(3)],
                  "streams_equal": bool(np.array_equal(filled, read)),
                  "device_seeding_pass_it_replaces": "one pass per 5 steps inside bench.py's timed region: value 274.5 G from raw against 320.4 G with the state resident (profiles/r4_bench.json)"}
cs.close()
out["note"] = ("C packers zk_pack_{sha256,keccak}_witness_tails on one host core of this container, one full-size start instance (BASELINE C3: 2^20 rows); "
               "the device passes they replace were measured at 128 instances (profiles/r4_config_timings_mid.jsonl)")
print(json.dumps(out, indent=1))
