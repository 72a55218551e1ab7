"""Prover-stage kernels on the bench workload (the REAL main_vm cycle since round 3, 2^20 rows per instance): K10 lookup accumulators, K12 copy-permutation
grand product over the whole batch, and for one instance trace columns -> coefficients -> x8 coset LDE (K11).
GPU box, repo root: python tools/prover_stage_bench.py [batch] -> one JSON line (wall-clock around synchronous calls, second call)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "era-zkevm_circuits_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import zkgl
from bench import build_main_vm_cs, main_vm_streams

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
zkgl.init(0)
dev = torch.device("cuda", 0)
cs, limit = build_main_vm_cs(zkgl, 20)
n_outer, n_loop = cs.input_words()
o64, l64, _ = main_vm_streams(zkgl, cs, limit)      # the fixture's 64 executions; instance i replays execution i mod 64
E = o64.shape[1]
idx = np.arange(B) % E
outer = np.ascontiguousarray(o64[:, idx])
loop = np.ascontiguousarray(l64.reshape(l64.shape[0], E, limit)[:, idx, :].reshape(l64.shape[0], B * limit))
cs.set_batch(B)
d_outer = torch.from_numpy(outer.view(np.int64)).to(dev)
d_loop = torch.from_numpy(loop.view(np.int64)).to(dev)
cs.bind_inputs(False, d_outer, n_outer)
cs.bind_inputs(True, d_loop, n_loop)
stream = torch.cuda.current_stream().cuda_stream
cs.seed_carried_inputs(d_loop, stream)
ok, f = cs.resolve_and_check(stream)
assert ok, f
st = cs.stats()
rows, n_cols = st["rows_per_instance"], st["copy_columns"] + st["lookup_columns"]
cells = st["cells_populated_loop"] * limit + st["cells_populated_outer"]   # trace cells that hold a value: what K12 multiplies over


def wall(fn):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, r


out = {"batch": B, "rows_per_instance": rows, "populated_cells_per_instance": cells}
ms, (bad, _) = wall(lambda: cs.lookup_argument((11, 12), (13, 14), stream))
assert bad == 0
out["K10_lookup_argument"] = {"ms": round(ms, 2), "ms_per_instance": round(ms / B, 3)}
ms, (bad, _) = wall(lambda: cs.copy_permutation((11, 12), (13, 14), None, stream))
assert bad == 0
out["K12_copy_permutation_check"] = {"ms": round(ms, 2), "ms_per_instance": round(ms / B, 3), "cells_per_s": round(B * cells / ms * 1e3 / 1e9, 2),
                                    "read_GBps": round(B * cells * 8 / ms / 1e6, 1)}
zb = min(B, 8)   # the column z itself for a few instances: 33 MB of running products per instance
log_n = 20
cols = torch.empty((n_cols, 1 << log_n), dtype=torch.int64, device=dev)
ext = torch.empty((n_cols, 8 << log_n), dtype=torch.int64, device=dev)


def pipeline():
    cs.trace_columns(0, cols, log_n, None, stream)
    zkgl.ntt(cols, log_n, n_cols, None, True, 1, stream, True)          # rows -> bit-reversed coefficients
    zkgl.lde(cols, ext, log_n, 3, n_cols, None, 7, stream, True)          # -> natural-order values on 8 cosets


ms, _ = wall(pipeline)
out["K11_columns_interpolate_lde8_one_instance"] = {"ms": round(ms, 2), "columns": n_cols, "values_out": n_cols * (8 << log_n)}
print(json.dumps(out))
