"""Soak of the device Goldilocks multiply / fma / inverse on structured operands (powers of two +-k, p-k, 32-bit boundary
patterns, random): every pair of ~1500 special values against Python integers.  usage (GPU box): python tools/field_soak.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "era-zkevm_circuits_amd"))
import numpy as np
import zkgl
P = 0xFFFFFFFF00000001
zkgl.init(0)
vals = set()
for k in range(64):
    for d in (-3, -2, -1, 0, 1, 2, 3):
        for base in (1 << k, P - (1 << k), (1 << k) * 0xFFFFFFFF):
            vals.add((base + d) % P)
rng = np.random.default_rng(7)
vals |= {int(x) % P for x in rng.integers(0, 1 << 63, size=400, dtype=np.uint64)}
vals |= {P - 1 - int(x) for x in rng.integers(0, 1 << 20, size=100)}
vals = np.array(sorted(vals), dtype=np.uint64)
n = len(vals)
a = np.repeat(vals, n); b = np.tile(vals, n); c = np.roll(a, 12345)
da, db, dc = (zkgl.DeviceBuffer.from_numpy(x) for x in (a, b, c))
out = zkgl.DeviceBuffer(a.size)
zkgl.gl_mul_cols(out, da, db, a.size)
got = out.to_numpy()[:a.size]
want = np.array([(int(x) * int(y)) % P for x, y in zip(a, b)], dtype=np.uint64)
assert np.array_equal(got, want), "mul mismatch"
zkgl.gl_fma_cols(out, da, db, dc, 3, P - 5, a.size)
got = out.to_numpy()[:a.size]
want = np.array([(3 * int(x) * int(y) + (P - 5) * int(z)) % P for x, y, z in zip(a, b, c)], dtype=np.uint64)
assert np.array_equal(got, want), "fma mismatch"
dv = zkgl.DeviceBuffer.from_numpy(vals)
oi = zkgl.DeviceBuffer(n)
zkgl.gl_inv_cols(oi, dv, n)
inv = oi.to_numpy()[:n]
assert all((int(x) * int(y)) % P == (1 if int(x) else 0) for x, y in zip(vals, inv)), "inverse mismatch"
print(f"field soak ok: {n} special values, {a.size} pairs (mul, fma), {n} inverses")
