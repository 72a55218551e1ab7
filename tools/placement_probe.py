"""GPU probe: does the loop kernel's "mode" (39-40 vs 43-44 ms at B = 384) follow where the variable store lands?  One process, one
seeded stream; the store is re-allocated (zk_cs_set_batch) behind spacer allocations of different sizes, the loop kernel timed each time.
    python tools/placement_probe.py [B]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, zkgl, bench

zkgl.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
cs, limit = bench.build_main_vm_cs(zkgl, 20)
o64, l64, expect = bench.main_vm_streams(zkgl, cs, limit)     # the fixture's 64 executions; instance i replays execution i mod 64
E = o64.shape[1]
idx = np.arange(B) % E
outer = np.ascontiguousarray(o64[:, idx])
loop = np.ascontiguousarray(l64.reshape(l64.shape[0], E, limit)[:, idx, :].reshape(l64.shape[0], B * limit))
d_o, d_l = zkgl.DeviceBuffer.from_numpy(outer), zkgl.DeviceBuffer.from_numpy(loop)
cs.set_batch(B)
cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
cs.seed_carried_inputs(d_l)             # the per-cycle VM state from the raw words, once
zkgl.sync()
out = []
for spacer_gb in (0, 0, 1, 3, 7, 16, 0, 5, 11, 0):
    spacer = zkgl.DeviceBuffer(max(1, int(spacer_gb * (1 << 30) // 8)))
    cs.set_batch(B)                      # frees and re-allocates the stores
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    ms, mhz = [], []
    for _ in range(4):
        ok, f = cs.resolve_and_check(); assert ok, f
        ms.append(cs.last_ms(1)); mhz.append(cs.last_ms(8))
    out.append({"spacer_GB": spacer_gb, "loop_ms": [round(x, 2) for x in ms[1:]], "shader_mhz": round(float(np.mean(mhz[1:])))})
    print(json.dumps(out[-1])); sys.stdout.flush()
    spacer.free()
