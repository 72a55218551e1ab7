"""GPU probe 2: WHICH allocation carries the loop kernel's per-process mode?  One process; between measurements exactly one thing
is re-allocated at a new address (the new copy is made before the old one is freed): the input streams, or the whole constraint
system (programs, constant pool, tables, multiplicity vector, stores).    python tools/placement_probe2.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, zkgl, bench

zkgl.init(0)
B = 384
cs, limit = bench.build_main_vm_cs(zkgl, 20)
o64, l64, expect = bench.main_vm_streams(zkgl, cs, limit)
E = o64.shape[1]
idx = np.arange(B) % E
outer = np.ascontiguousarray(o64[:, idx])
loop = np.ascontiguousarray(l64.reshape(l64.shape[0], E, limit)[:, idx, :].reshape(l64.shape[0], B * limit))
d_o, d_l = zkgl.DeviceBuffer.from_numpy(outer), zkgl.DeviceBuffer.from_numpy(loop)
cs.set_batch(B)
cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
cs.seed_carried_inputs(d_l); zkgl.sync()
seeded = d_l.to_numpy().reshape(loop.shape)

def measure(tag):
    ms, mhz = [], []
    for _ in range(4):
        ok, f = cs.resolve_and_check(); assert ok, f
        ms.append(cs.last_ms(1)); mhz.append(cs.last_ms(8))
    print(json.dumps({"after": tag, "loop_ms": [round(x, 2) for x in ms[1:]], "shader_mhz": round(float(np.mean(mhz[1:]))),
                      "loop_inputs_at": hex(d_l.ptr)})); sys.stdout.flush()

measure("start")
for rep in range(3):
    n_o, n_l = zkgl.DeviceBuffer.from_numpy(outer), zkgl.DeviceBuffer.from_numpy(seeded)   # new addresses first, then free the old
    cs.bind_inputs(False, n_o, outer.shape[0]); cs.bind_inputs(True, n_l, loop.shape[0])
    d_o.free(); d_l.free(); d_o, d_l = n_o, n_l
    measure(f"inputs re-allocated ({rep})")
for rep in range(3):
    new, _ = bench.build_main_vm_cs(zkgl, 20)       # new programs / pool / tables at new addresses, then the old system goes
    new.set_batch(1)
    cs.close(); cs = new
    cs.set_batch(B)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    measure(f"constraint system rebuilt ({rep})")
