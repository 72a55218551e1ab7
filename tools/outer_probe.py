"""Outer-scope phases of the VM workload, alone (resolve(): phases in sequence on one stream) and overlapped with the loop
scope's kernels (resolve_and_check()).  GPU box, repo root: python tools/outer_probe.py [batch]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "era-zkevm_circuits_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import zkgl
from bench import build_vm_cs, vm_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 145
zkgl.init(0)
dev = torch.device("cuda", 0)
cs, limit = build_vm_cs(zkgl, 20)
n_outer, n_loop = cs.input_words()
outer, loop = vm_inputs(np.random.default_rng(0xC2), n_outer, n_loop, B, limit)
cs.set_batch(B)
d_outer = torch.from_numpy(outer.view(np.int64)).to(dev)
d_loop = torch.from_numpy(loop.view(np.int64)).to(dev)
cs.bind_inputs(False, d_outer, n_outer)
cs.bind_inputs(True, d_loop, n_loop)
stream = torch.cuda.current_stream().cuda_stream
cs.seed_carried_inputs(d_loop, stream)
for _ in range(2):
    cs.resolve(stream)
    torch.cuda.synchronize()
    print("resolve(): total %.2f ms, loop %.2f ms, outer pre+post %.2f ms" % (cs.last_ms(0), cs.last_ms(1), cs.last_ms(4)))
for _ in range(2):
    ok, f = cs.resolve_and_check(stream)
    torch.cuda.synchronize()
    assert ok
    print("resolve_and_check(): total %.2f ms, loop %.2f ms, loop checks %.2f ms (gates %.2f), outer post+checks %.2f ms"
          % (cs.last_ms(0), cs.last_ms(1), cs.last_ms(2), cs.last_ms(3), cs.last_ms(4)))
