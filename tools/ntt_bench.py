"""K11 throughput: batched 2^log_n transforms over trace-column-sized batches.  GPU box, repo root:
python tools/ntt_bench.py [log_n] [n_polys] -> one JSON line (elements/s, ms, HBM fraction of the 2-pass traffic)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "era-zkevm_circuits_amd"))
import zkgl

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_polys = int(sys.argv[2]) if len(sys.argv) > 2 else 164 * 4   # four main_vm instances' worth of columns
zkgl.init(0)
n = 1 << log_n
g = torch.Generator(device="cuda").manual_seed(1)
d = torch.randint(0, 2**62, (n_polys, n), dtype=torch.int64, device="cuda", generator=g)
stream = torch.cuda.current_stream().cuda_stream
passes = (log_n + 9) // 10


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"log_n": log_n, "n_polys": n_polys, "passes": passes}
for name, inv, shift in (("forward", False, 1), ("forward_coset", False, 7), ("inverse_coset", True, 7)):
    ms = timed(lambda: zkgl.ntt(d, log_n, n_polys, n, inv, shift, stream))
    el = n_polys * n
    out[name] = {"ms": round(ms, 3), "gelem_per_s": round(el / ms / 1e6, 2),
                 "algorithmic_GBps": round(16 * el / ms / 1e6, 1),          # read once + write once
                 "moved_GBps": round(16 * passes * el / ms / 1e6, 1)}       # what the passes actually move
lb = 3
q = min(n_polys, 164)
dst = torch.empty((q, n << lb), dtype=torch.int64, device="cuda")
ms = timed(lambda: zkgl.lde(d, dst, log_n, lb, q, n, 7, stream), reps=3)
out["lde_x8"] = {"n_polys": q, "ms": round(ms, 3), "gelem_out_per_s": round(q * (n << lb) / ms / 1e6, 2)}
print(json.dumps(out))
