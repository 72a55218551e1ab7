"""GPU probe of the chain-specialised main_vm seeding: (1) native seeding == cone seeding == native restatement on the mixed test
batch; (2) timing of a bench-sized stream (default 1920 instances x 2384 cycles) with the phase split, spot-checked against the cone
kernel on the first instances.   python tools/vm_seed_probe.py [n_instances]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, zkgl
import bench
import vm_programs as vp

zkgl.init(0)
dev = torch.device("cuda", 0)
os.environ["ZKGL_SEED_PHASE_MS"] = "1"
out = {}
# ---- (1) parity on the mixed batch
d, D = vp.defs()
LIMIT = 32
cs = vp.vm_cs(LIMIT)
outer, loop, commits, info = vp.mixed_batch(cs, D, LIMIT, 64)
B = outer.shape[1]
raw = loop.copy(); raw[0:243] = 0
res = {}
for mode in ("1", "0"):
    os.environ["ZKGL_SEED_NATIVE"] = mode
    d_o, d_l = zkgl.DeviceBuffer.from_numpy(outer), zkgl.DeviceBuffer.from_numpy(raw)
    cs.seed_stream(B, d_o, d_l)
    res[mode] = d_l.to_numpy().reshape(loop.shape)
bad = np.argwhere(res["1"] != loop)
out["mixed_batch_native_equals_restatement"] = bool(len(bad) == 0)
out["mixed_batch_native_equals_cone"] = bool(np.array_equal(res["1"], res["0"]))
if len(bad):
    w, col = bad[0]
    out["differing_words"] = sorted(set(int(x) for x in bad[:, 0]))
    out["first_difference"] = {"word": int(w), "column": int(col), "instance": info[col // LIMIT], "cycle": int(col % LIMIT), "native": int(res["1"][w, col]), "want": int(loop[w, col]), "n": len(bad)}
print(json.dumps(out)); sys.stdout.flush()
# ---- (2) bench-sized stream
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
cs, limit = bench.build_main_vm_cs(zkgl, 20)
n_outer, n_loop = cs.input_words()
NE = int(os.environ.get('PROBE_EXECS', '64'))
outer8, loop8, expect8 = bench.main_vm_streams(zkgl, cs, limit, NE)
sel = torch.arange(S, device=dev) % NE
d_outer = torch.from_numpy(outer8.view(np.int64)).to(dev)[:, sel].contiguous()
l8 = torch.from_numpy(loop8.view(np.int64)).to(dev).view(n_loop, NE, limit)
d_loop = l8[:, sel, :].reshape(n_loop, S * limit).contiguous()
del l8
stream = torch.cuda.current_stream().cuda_stream
os.environ["ZKGL_SEED_NATIVE"] = "1"
times = []
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    cs.seed_stream(S, d_outer, d_loop, stream)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t)
out["stream_instances"] = S; out["limit"] = limit
out["native_seed_s"] = [round(x, 4) for x in times]
out["native_phase_ms"] = [round(cs.last_ms(k), 3) for k in (5, 6, 7)]
out["instances_per_s"] = S / min(times)
del os.environ["ZKGL_SEED_PHASE_MS"]      # the production path: chunks of cycles, chains under the walker
times = []
for rep in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    cs.seed_stream(S, d_outer, d_loop, stream)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t)
out["native_seed_s_chunked"] = [round(x, 4) for x in times]
out["instances_per_s_chunked"] = S / min(times)
state_native = d_loop[:243].view(243, S, limit)[:, :8, :].cpu().numpy().copy()
# the cone on the first 8 instances (its pass costs the same for 8 and for 1024)
os.environ["ZKGL_SEED_NATIVE"] = "0"
d_o8 = torch.from_numpy(outer8.view(np.int64)).to(dev)[:, :8].contiguous()
d_l8 = torch.from_numpy(loop8.view(np.int64)).to(dev).view(n_loop, NE, limit)[:, :8, :].reshape(n_loop, 8 * limit).contiguous()
torch.cuda.synchronize(); t = time.perf_counter()
cs.seed_stream(8, d_o8, d_l8, stream)
torch.cuda.synchronize(); out["cone_seed_s_8_instances"] = round(time.perf_counter() - t, 4)
state_cone = d_l8[:243].view(243, 8, limit).cpu().numpy()
out["bench_stream_native_equals_cone"] = bool(np.array_equal(state_native, state_cone))
if not out["bench_stream_native_equals_cone"]:
    badw = np.argwhere(state_native != state_cone)
    out["bench_first_difference"] = [int(x) for x in badw[0]] + [len(badw)]
# and the trace it seeds satisfies the circuit
os.environ["ZKGL_SEED_NATIVE"] = "1"
Bc = min(S, 64)
cs.set_batch(Bc)
cs.bind_inputs(False, d_outer, n_outer, lane_stride=S, lane_offset=0)
cs.bind_inputs(True, d_loop, n_loop, lane_stride=S * limit, lane_offset=0)
ok, f = cs.resolve_and_check(stream)
out["resolve_and_check_on_native_seeded_stream"] = bool(ok)
if expect8 is not None:
    got = np.array([cs.public_inputs(i) for i in range(8)], dtype=np.uint64)
    out["commitments_equal_fixture"] = bool(np.array_equal(got, expect8[:8]))
print(json.dumps(out))
