"""Round 5's GPU tests of paths that had not executed on a device when they were written (the GPU was closed to the build for the whole round).
They live in ONE file that sorts last: the driver runs `pytest -x`, and a surprise in new code must not hide the established parity evidence
(whole traces of every circuit against the oracle) behind the first failure.

  * who evaluates a gate on a macro-op's output (tests/test_macro_ownership.py): forged / honest outside gates, adversarial inputs, both modes
  * C3 keccak, C3 sha256 (2^20 rows) and C5 (8 blobs x 4096 chunks) under check_if_satisfied's semantics (every relation from the stored values)
  * the seeding cone with gated witness-only permutations (tests/test_seed_program.py)
  * macro-op backends with kernels of their own (round 6: in the one library): whole-trace parity (was: tools/ab_r5.sh)
  * bench.py end to end on a small configuration, one rank and two"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import zkgl
from oracle import zko
from zkgl import GATE as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


# ---- macro-op ownership (tests/test_macro_ownership.py holds the circuits and the software references)
import test_macro_ownership as MO  # noqa: E402


@pytest.mark.parametrize("kind", ["keccak", "sha256", "sha256_reference_tables"])
@pytest.mark.parametrize("stored", [False, True])
def test_forged_gate_on_a_macro_output_is_rejected_in_both_check_modes(zk, kind, stored):
    make = {"keccak": MO.keccak_circuit, "sha256": MO.sha_circuit, "sha256_reference_tables": lambda e=None: MO.sha_circuit(e, True)}[kind]
    n_in = 200 if kind == "keccak" else 96
    rng = np.random.default_rng(13)
    B = 70
    inp = rng.integers(0, 256, size=(n_in, B), dtype=np.uint64)
    for extra, want_ok in ((MO.honest, True), (MO.forged, False)):
        cs = make(extra)
        cs.set_check_mode(stored)
        cs.set_batch(B)
        d = zkgl.DeviceBuffer.from_numpy(inp)
        cs.bind_inputs(False, d, n_in)
        ok, f = cs.resolve_and_check()
        assert ok == want_ok, (kind, stored, f)
        if not want_ok:
            assert f.kind == G["REDUCTION4"]
        else:
            for i in (0, B - 1):
                col = bytes(int(x) for x in inp[:, i])
                want = MO.keccak_f_bytes(col)[:32] if kind == "keccak" else MO.sha_compress_bytes(col[:32], col[32:])
                assert bytes(cs.public_inputs(i)) == want
            # every cell of the trace equals the oracle interpreter's
            run = zko.CircuitRun(cs.export(False), cs.export(True), B, 1 << 20)
            run.resolve(inp, np.zeros((0, B), dtype=np.uint64))
            assert np.array_equal(cs.trace(False), run.oc)
            # differential, adversarial inputs: an input that is not a byte makes a tuple of the gadget a non-row of its table; the macro-op
            # tests its inputs (fused), the check program finds the tuple (stored), the oracle checker counts it — same verdict, same instance
            for word, inst, value in ((0, 5, 256), (n_in - 1, B - 1, zkgl.P - 1), (n_in // 2, 33, 1 << 40)):
                bad = inp.copy(); bad[word, inst] = value
                d = zkgl.DeviceBuffer.from_numpy(bad)
                cs.bind_inputs(False, d, n_in)
                ok, f = cs.resolve_and_check()
                assert not ok and f.instance == inst, (kind, stored, word, f)
                run = zko.CircuitRun(cs.export(False), cs.export(True), B, 1 << 20)
                run.resolve(bad, np.zeros((0, B), dtype=np.uint64))
                nbad, _ = run.check()
                assert nbad > 0


# ---- check_if_satisfied's semantics on the macro-op circuits (/root/reference/src/ram_permutation/mod.rs:556)
def _stored_mode_agrees(cs, good, bad=None):
    """the same batch with EVERY relation re-evaluated from the stored values (zk_cs_set_check_mode(ZK_CHECK_STORED)): satisfied where the fused
    step was, and a rejected batch rejected at the same instance"""
    cs.set_check_mode(True)
    try:
        for (d_o, d_l, n_o, n_l), want_instance in ((good, None),) + (((bad[:4], bad[4]),) if bad else ()):
            cs.bind_inputs(False, d_o, n_o); cs.bind_inputs(True, d_l, n_l)
            ok, f = cs.resolve_and_check()
            if want_instance is None:
                assert ok, f
            else:
                assert not ok and f.instance == want_instance, f
    finally:
        cs.set_check_mode(False)


def test_c3_keccak_under_the_stored_mode(zk):
    import test_gpu_full_size as T
    from oracle import keccak_native as kn
    cs, limit = T.fit(lambda c: c.configure_keccak(), lambda c, l: c.keccak256_round_function_entry_point(l), 20)
    insts = []
    for seed in (0xC3, 0xC3 + 1):
        reqs, _ = T._keccak_requests(np.random.default_rng(seed), limit)
        insts.append(kn.instance(reqs, limit))
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    ok, f, keep = T.run_gpu(zk, cs, outer, loop, len(insts))
    assert ok, f
    bad = loop.copy(); bad[459, limit + 3] ^= 1   # a memory word read by instance 1 differs from the one its queue chain was built with
    ok, f, keep2 = T.run_gpu(zk, cs, outer, bad, len(insts))
    assert not ok and f.instance == 1
    _stored_mode_agrees(cs, keep + (outer.shape[0], loop.shape[0]), keep2 + (outer.shape[0], loop.shape[0], 1))


def test_c3_sha256_under_the_stored_mode(zk):
    import test_gpu_full_size as T
    from oracle import sha256_native as shn
    cs, limit = T.fit(lambda c: c.configure_sha256(), lambda c, l: c.sha256_round_function_entry_point(l), 20)
    rng = np.random.default_rng(0xC3 + 2)
    msgs = [bytes(rng.integers(0, 256, size=64 * 8 - 9, dtype=np.uint8)) for _ in range(limit // 8)]
    reqs = [shn.request(m, 1 + 2 * i, 10 + i, 0, 9000 + i, i) for i, m in enumerate(msgs)]
    inst = shn.instance(reqs, limit)
    assert inst["satisfiable"]
    outer = np.array([inst["outer"]] * 2, dtype=np.uint64).T.copy()
    loop = np.array(inst["rows"] * 2, dtype=np.uint64).T.copy()
    ok, f, keep = T.run_gpu(zk, cs, outer, loop, 2)
    assert ok, f
    _stored_mode_agrees(cs, keep + (outer.shape[0], loop.shape[0]))


def test_c5_eip4844_under_the_stored_mode(zk):
    from test_eip4844_host import make_instances, streams
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4), 1 << 21, 1 << 28)
    cs.configure_eip_4844()
    cs.eip_4844_entry_point(4096)
    cs.pad_and_shrink()
    insts = make_instances(4096, [11, 12, 13, 14])
    outer, loop = streams(insts)
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    for stored in (False, True):
        cs.set_check_mode(stored)
        ok, f = cs.resolve_and_check()
        assert ok, (stored, f)
    lanes = loop.shape[1] // len(insts)
    bad = loop.copy(); bad[217 + 5, 3 * lanes + 7] ^= 1          # a block byte of blob 3 differs from the one the sponge state was walked with
    d_b = zk.DeviceBuffer.from_numpy(bad)
    cs.bind_inputs(True, d_b, bad.shape[0])
    for stored in (True, False):
        cs.set_check_mode(stored)
        ok, f = cs.resolve_and_check()
        assert not ok and f.instance == 3, (stored, f)
    cs.set_check_mode(False)


# ---- the seeding cone with gated witness-only permutations
def test_cone_with_gated_permutations_is_verified_not_assumed(zk, monkeypatch):
    import test_seed_program as SP
    monkeypatch.setenv("ZKGL_SEED_NATIVE", "0")
    B, limit = 70, 5
    rng = np.random.default_rng(3)
    outer = rng.integers(1, 1 << 60, size=(1, B), dtype=np.uint64)
    loop = np.zeros((2, B * limit), dtype=np.uint64)
    loop[1] = rng.integers(0, 2, size=B * limit)
    for ok_form in (True, False):
        cs = SP._gated_chain(ok_form)
        cs.set_batch(B)
        d_o, d_l = zkgl.DeviceBuffer.from_numpy(outer), zkgl.DeviceBuffer.from_numpy(loop)
        cs.bind_inputs(False, d_o, 1); cs.bind_inputs(True, d_l, 2)
        if not ok_form:      # the carried word would be seeded from an ungated output: the cone is not offered
            with pytest.raises(zkgl.ZkError):
                cs.seed_carried_inputs(d_l)
            continue
        cs.seed_carried_inputs(d_l)
        seeded = d_l.to_numpy().reshape(loop.shape)
        want = zko.CircuitRun(cs.export(False), cs.export(True), B, 1).seed(outer, loop)
        assert np.array_equal(seeded, want)
        ok, f = cs.resolve_and_check()
        assert ok, f


# ---- macro-op backends with kernels of their own
def test_bytebuf_macro_recording_on_the_gpu_equals_the_oracle(zk, monkeypatch):
    """whole trace of the macro recording (ZKGL_BYTEBUF_MACRO=1 at record time), plain and strand kernels, both check modes; seeding through the native
    FSM seeder.  Since round 6 the op's device backend is in the one library, in kernels of its own (k_witness_strands2<.., X_BYTEBUF>, k_witness_plain_x<X_BYTEBUF>)."""
    import zkgl
    import test_bytebuf_macro as BB
    from test_keccak_fsm_host import REFERENCE_CASES, TABLE_ROWS, reference_case, streams
    from oracle import keccak_native as N
    cs = BB.record(monkeypatch, True)
    assert zkgl.build_features() & zkgl.BUILD_BYTEBUF_KERNEL
    insts = [reference_case(l, u)[1] for l, u in REFERENCE_CASES] * 8        # 80 instances x 2 cycles: a few wavefronts
    outer, loop = streams(insts, 2)
    r = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS)
    r.resolve(outer, loop)
    for strands in ("0", "1"):
        monkeypatch.setenv("ZKGL_STRANDS", strands)
        cs.set_batch(len(insts))
        raw = loop.copy(); raw[:N.CARRIED] = 0
        d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
        cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
        cs.seed_carried_inputs(d_l)
        assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
        for stored in (False, True):
            cs.set_check_mode(stored)
            ok, f = cs.resolve_and_check()
            assert ok, (strands, stored, f)
        from test_gpu_cs import assert_trace_equal
        assert_trace_equal(cs, r)
        bad = loop.copy(); bad[300, 5] = 256                                  # a buffer byte that is not a byte: rejected in both modes
        d_b = zk.DeviceBuffer.from_numpy(bad)
        cs.bind_inputs(True, d_b, loop.shape[0])
        for stored in (False, True):
            cs.set_check_mode(stored)
            ok, f = cs.resolve_and_check()
            assert not ok
    cs.set_check_mode(False)


def test_sha4_macro_recording_on_the_gpu_equals_the_oracle(zk, monkeypatch):
    """whole trace of the macro recording (the default recording of the reference's table set since round 6) against the oracle interpreter, both check
    modes; adversarial inputs rejected in both.  Its kernels: k_witness_strands2<.., X_SHA4>, k_witness_plain_x<X_SHA4>."""
    import test_sha4_macro as S4
    from test_sha256_host import loop_stream
    REF_TABLE_ROWS = S4.REF_TABLE_ROWS
    cs = S4.record(monkeypatch, True)
    assert zkgl.build_features() & zkgl.BUILD_SHA4_KERNEL
    rng = np.random.default_rng(45)
    msgs = [bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in rng.integers(56, 120, size=70)]
    outer = np.zeros((0, len(msgs)), dtype=np.uint64)
    raw = loop_stream(msgs, 2)
    loop = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS).seed(outer, raw)
    r = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS)
    r.resolve(outer, loop)
    for strands in ("0", "1"):
        monkeypatch.setenv("ZKGL_STRANDS", strands)
        cs.set_batch(len(msgs))
        d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
        cs.bind_inputs(False, d_o, 0); cs.bind_inputs(True, d_l, raw.shape[0])
        cs.seed_carried_inputs(d_l)
        assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
        for stored in (False, True):
            cs.set_check_mode(stored)
            ok, f = cs.resolve_and_check()
            assert ok, (strands, stored, f)
        for i, m in enumerate(msgs):
            assert bytes(cs.public_inputs(i)) == hashlib.sha256(m).digest()
        from test_gpu_cs import assert_trace_equal
        assert_trace_equal(cs, r)
        bad = loop.copy(); bad[40, 5] = 256                                   # a block byte that is not a byte: rejected in both modes
        d_b = zk.DeviceBuffer.from_numpy(bad)
        cs.bind_inputs(True, d_b, loop.shape[0])
        for stored in (False, True):
            cs.set_check_mode(stored)
            ok, f = cs.resolve_and_check()
            assert not ok
    cs.set_check_mode(False)


def test_gpu_equals_oracle_with_the_reference_tables(zk):
    """tests/test_sha256_reference_tables.py's device half (sha256 blocks + the code_unpacker reference fixture under the reference's width-4 tables)"""
    from test_sha256_reference_tables import gpu_equals_oracle_with_the_reference_tables
    gpu_equals_oracle_with_the_reference_tables(zk)


# ---- bench.py end to end
def run_bench(*args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def _needs_hardware(zk):
    # bench.py times a GPU through torch; on the emulated device of tests/emu (ZKGL_LIB = tests/emu/_gen/dev/libzkgl.so) there is nothing to time
    if __import__("helpers").emulated_device():
        pytest.skip("bench.py needs the hardware: nothing to time on the emulated device")


def test_one_gpu_line_carries_roofline_and_host_fed_figures(zk):
    _needs_hardware(zk)
    d = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "6", "--log2-rows", "16", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["unit"] == "constraints/s"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "zke::k_witness_loop" and 0 < r["frac"] < 1 and r["avg_launch_ms"] > 0
    assert d["config"]["commitment_gather"].startswith("zk_cs_gather_commitments")
    for key in ("value_including_host_pack", "value_states_from_witness"):
        h = d[key]
        assert h is not None and "error" not in h, h
        assert h["value"] > 0 and h["pack_ms_per_instance_one_core"] > 0 and h["h2d_GBps"] > 0 and h["host_cores_per_gpu_to_sustain_value"] > 0
    # device_seeds stages 117 of the 360 rows, states_from_witness all of them
    assert d["value_states_from_witness"]["staged_bytes_per_window"] > 2.5 * d["value_including_host_pack"]["staged_bytes_per_window"]
    assert d["distinct_commitments"] == 6


def test_two_rank_launch_preflight(zk):
    import zkgl
    _needs_hardware(zk)
    d = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--log2-rows", "16", "--no-cpu-baseline", "--headline-only")
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["config"]["per_rank_ms_per_step"]) == 2
    if zkgl.device_count() >= 2:
        assert d["config"]["commitment_gather"].startswith("zk_cs_gather_commitments"), d["config"]["commitment_gather"]
    else:
        assert "ranks share one GPU" in d["config"]["commitment_gather"]
    assert d["distinct_commitments"] == 8
