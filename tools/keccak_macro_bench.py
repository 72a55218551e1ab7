"""GPU box: K8 micro-benchmark — the keccak256_blocks circuit (a chain of Keccak-f permutations, nothing else) resolved with the
macro-op ZK_OP_KECCAK_F and, under ZKGL_NO_HASH_MACROS=1, with one interpreted op per value.
usage: python tools/keccak_macro_bench.py [instances] [blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import zkgl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
zkgl.init(0)
cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4), max_trace_len=1 << 22)
cs.configure_keccak()
cs.keccak256_blocks_entry_point(NB)
cs.pad_and_shrink()
st = cs.stats()
rng = np.random.default_rng(1)
n_outer, n_loop = cs.input_words()
loop = np.zeros((n_loop, B * NB), dtype=np.uint64)
loop[200:] = rng.integers(0, 256, size=(n_loop - 200, B * NB))
cs.set_batch(B)
d_l = zkgl.DeviceBuffer.from_numpy(loop)
cs.bind_inputs(True, d_l, n_loop)
cs.seed_carried_inputs(d_l)
for mode in ("default", "ZKGL_STRANDS=0", "ZKGL_STRANDS=1"):
    if "=" in mode:
        k, v = mode.split("="); os.environ[k] = v
    ok, f = cs.resolve_and_check(); assert ok, f
    t0 = time.perf_counter(); ok, f = cs.resolve_and_check(); dt = time.perf_counter() - t0
    vals = B * NB * st["cells_written_loop"] * 8
    print(f"{'macro' if not os.environ.get('ZKGL_NO_HASH_MACROS') else 'interpreted'} {mode}: {B} x {NB} permutations, loop kernel {cs.last_ms(1):.2f} ms, step {1e3 * dt:.2f} ms, "
          f"{vals / cs.last_ms(1) / 1e6:.0f} GB/s of values, ops/lane {st['loop_ops']}, values/lane {st['cells_written_loop']}")
