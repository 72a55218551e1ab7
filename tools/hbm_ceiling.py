"""Achievable HBM bandwidth on this GPU with trivial streaming kernels (torch ops on 8 GiB tensors):
write-only (fill_), read-only (sum), copy (read + write).  Context for roofline fractions quoted against the 8 TB/s spec."""
import json, time, torch
n = 1 << 30  # 8 GiB of int64
x = torch.empty(n, dtype=torch.int64, device="cuda")
y = torch.empty(n, dtype=torch.int64, device="cuda")
def bench(f, bytes_moved, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return bytes_moved / (best * 1e-3) / 1e9
out = {"write_only_GBps": bench(lambda: x.fill_(7), 8 * n), "read_only_GBps": bench(lambda: x.sum(), 8 * n),
       "copy_total_GBps": bench(lambda: y.copy_(x), 16 * n)}
print(json.dumps(out))
