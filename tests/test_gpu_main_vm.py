"""-m gpu: main_vm on a real MI355X through the C ABI: >= 64 instances x 32 cycles drawn from programs that execute
all eleven opcode families (far_call / ret / UMA / log included).  The device derives the per-cycle VmLocalState from the raw
oracle words (zk_cs_seed_carried_inputs) == the native restatement; the resolved trace is bit-exact, cell for cell, against
oracle/zko_engine.c; check_if_satisfied agrees; public inputs == the native input commitments; a tampered witness is reported at
the right instance."""
import numpy as np
import pytest

import vm_programs as vp
from oracle import zko
from test_main_vm_host import run_oracle

pytestmark = pytest.mark.gpu

LIMIT = 32


@pytest.fixture(scope="module")
def batch():
    d, D = vp.defs()
    cs = vp.vm_cs(LIMIT)
    outer, loop, commits, info = vp.mixed_batch(cs, D, LIMIT, 64)
    return cs, D, outer, loop, commits, info


def test_main_vm_gpu_bit_exact(zk, batch):
    cs, D, outer, loop, commits, info = batch
    B = outer.shape[1]
    assert B >= 64 and loop.shape[1] == B * LIMIT
    raw = loop.copy()
    raw[0:243] = 0
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    seeded = d_l.to_numpy().reshape(loop.shape)
    assert np.array_equal(seeded, loop), "device-seeded VmLocalState differs from the native restatement"
    ok, f = cs.resolve_and_check()
    assert ok, f
    run = run_oracle(cs, B)
    run.resolve(outer, loop)
    assert run.check()[0] == 0
    assert np.array_equal(cs.trace(False), run.oc), "outer-scope trace differs from the oracle"
    assert np.array_equal(cs.trace(True), run.lc), "loop-scope trace differs from the oracle"
    for i in range(B):
        assert cs.public_inputs(i) == commits[i], info[i]
    total = run.mult.size // B
    for i in (0, B // 2, B - 1):
        assert np.array_equal(cs.multiplicities(i), run.mult[i * total:(i + 1) * total])


def test_main_vm_gpu_generic_seeding_and_plain_program(zk, batch, monkeypatch):
    """the generic sequential seeding mode and the non-strand program form give the same streams / trace"""
    cs, D, outer, loop, commits, info = batch
    B = 8
    o, l = outer[:, :B].copy(), loop[:, :B * LIMIT].copy()
    raw = l.copy()
    raw[0:243] = 0
    monkeypatch.setenv("ZKGL_SEED_GENERIC", "1")
    monkeypatch.setenv("ZKGL_STRANDS", "0")
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(o), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, o.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(l.shape), l)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i in range(B):
        assert cs.public_inputs(i) == commits[i]


def test_main_vm_gpu_reports_tampered_witness(zk, batch):
    cs, D, outer, loop, commits, info = batch
    B = outer.shape[1]
    lay = cs.main_vm_layout()["loop"]
    victim = next(i for i, (name, seed, chunk) in enumerate(info) if name == "calls" and chunk == 1)
    bad = loop.copy()
    bad[lay["state"][0] + 9, victim * LIMIT + 7] ^= 1     # a carried register limb that is not the previous cycle's output
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(bad)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, bad.shape[0])
    ok, f = cs.resolve_and_check()
    assert not ok and f.instance == victim


def test_main_vm_gpu_stream_seeding_and_windows(zk, batch):
    """zk_cs_seed_stream over the whole stream (no batch set for it), then the stream resolved in windows of 24 instances
    through zk_cs_bind_inputs_window: same seeded words, every window satisfied, same commitments"""
    cs, D, outer, loop, commits, info = batch
    S = outer.shape[1]
    raw = loop.copy()
    raw[0:243] = 0
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.seed_stream(S, d_o, d_l)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop), "stream-seeded VmLocalState differs from the native restatement"
    W = 24
    cs.set_batch(W)
    for first in range(0, S - W + 1, W):
        cs.bind_inputs(False, d_o, outer.shape[0], lane_stride=S, lane_offset=first)
        cs.bind_inputs(True, d_l, raw.shape[0], lane_stride=S * LIMIT, lane_offset=first * LIMIT)
        ok, f = cs.resolve_and_check()
        assert ok, (first, f)
        for i in range(W):
            assert cs.public_inputs(i) == commits[first + i], info[first + i]


def test_gather_commitments_through_the_c_abi(zk, batch):
    """zk_comm_* + zk_cs_gather_commitments (RCCL all-gather behind the C ABI) on a one-rank communicator: the gathered
    [world, batch, 4] words are the public inputs of every instance"""
    cs, D, outer, loop, commits, info = batch
    B = 16
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer[:, :B].copy()), zk.DeviceBuffer.from_numpy(loop[:, :B * LIMIT].copy())
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    ok, f = cs.resolve_and_check()
    assert ok, f
    comm = zk.Comm(zk.Comm.unique_id(), 0, 1)
    try:
        got = cs.gather_commitments(comm)
    finally:
        comm.close()
    assert got.shape == (1, B, 4)
    for i in range(B):
        assert [int(x) for x in got[0, i]] == commits[i]


@pytest.mark.parametrize("mode", ["inline", "pass"])
def test_multiplicities_in_both_modes(zk, batch, monkeypatch, mode):
    """lookup multiplicities by wave-aggregated atomics inside the witness kernels (inline) and by the k_multiplicities pass over the
    stored keys (pass: what hash-style circuits use by default) — the same vectors, equal to the oracle's"""
    cs, D, outer, loop, commits, info = batch
    monkeypatch.setenv("ZKGL_MULT_MODE", mode)
    B = 16
    o, l = outer[:, :B].copy(), loop[:, :B * LIMIT].copy()
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(o), zk.DeviceBuffer.from_numpy(l)
    cs.bind_inputs(False, d_o, o.shape[0]); cs.bind_inputs(True, d_l, l.shape[0])
    ok, f = cs.resolve_and_check()
    assert ok, f
    run = run_oracle(cs, B)
    run.resolve(o, l)
    total = run.mult.size // B
    for i in range(B):
        assert np.array_equal(cs.multiplicities(i), run.mult[i * total:(i + 1) * total]), (mode, i)
    cs.resolve()   # the plain resolve path counts them too
    assert np.array_equal(cs.multiplicities(3), run.mult[3 * total:4 * total])


def test_native_seeding_equals_cone_seeding(zk, batch, monkeypatch):
    """the chain-specialised seeding (native walker + Poseidon2 chain kernels + fill: kernels_vm_seed.hpp) and the recorded cone of the
    carried outputs (k_seed_wave) derive the same 243 words for every cycle — and both equal the native restatement; the window
    entry point (zk_cs_seed_window_async) seeds a slice of a longer stream and leaves the rest alone"""
    cs, D, outer, loop, commits, info = batch
    S = outer.shape[1]
    raw = loop.copy()
    raw[0:243] = 0
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ZKGL_SEED_NATIVE", mode)
        d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
        cs.seed_stream(S, d_o, d_l)
        got[mode] = d_l.to_numpy().reshape(loop.shape)
    assert np.array_equal(got["1"], got["0"]), "native seeding differs from the cone kernels"
    assert np.array_equal(got["1"], loop), "seeded VmLocalState differs from the native restatement"
    monkeypatch.setenv("ZKGL_SEED_NATIVE", "1")
    monkeypatch.setenv("ZKGL_VM_SEED_CHUNKS", "1")        # one chunk, phase after phase on the caller's stream
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.seed_stream(S, d_o, d_l)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
    monkeypatch.delenv("ZKGL_VM_SEED_CHUNKS")
    first, n = 5, 17
    d_l2 = zk.DeviceBuffer.from_numpy(raw)
    cs.seed_window_async(n, d_o, S, d_l2, S * LIMIT, first)
    zk.sync()
    win = d_l2.to_numpy().reshape(loop.shape)
    want = raw.copy()
    want[:, first * LIMIT:(first + n) * LIMIT] = loop[:, first * LIMIT:(first + n) * LIMIT]
    assert np.array_equal(win, want)


@pytest.mark.parametrize("ni", ["2", "4"])
def test_native_seeding_with_several_walkers_per_wavefront(zk, batch, monkeypatch, ni):
    """k_vm_walk<2> / <4>: two or four instances share a walking wavefront once a pass holds more than 1 024 / 2 048 instances (bench.py's 1 920-instance passes
    take <2>).  A test batch is far smaller, so the form is forced (ZKGL_VM_WALK_NI): same 243 words per cycle as the native restatement, with an instance count
    that leaves the last wavefront partly empty, in chunks and as one chunk."""
    cs, D, outer, loop, commits, info = batch
    S = outer.shape[1]
    assert S % 4 != 0 or S > 4
    raw = loop.copy()
    raw[0:243] = 0
    monkeypatch.setenv("ZKGL_SEED_NATIVE", "1")
    monkeypatch.setenv("ZKGL_VM_WALK_NI", ni)
    n = S - 1 if (S - 1) % int(ni) else S - 2          # the last walking wavefront is not full
    for chunks in (None, "1"):
        if chunks:
            monkeypatch.setenv("ZKGL_VM_SEED_CHUNKS", chunks)
        d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
        cs.seed_window_async(n, d_o, S, d_l, S * LIMIT, 0)
        zk.sync()
        got = d_l.to_numpy().reshape(loop.shape)
        want = raw.copy()
        want[:, :n * LIMIT] = loop[:, :n * LIMIT]
        assert np.array_equal(got, want), (ni, chunks)


def test_main_vm_hook_compare_witness(zk):
    """structured_input.hook_compare_witness (src/main_vm/mod.rs:218): the circuit's hidden_fsm_output / observable_output groups against
    the closed-form input a host holds; input streams through zk_pack_main_vm_witness only"""
    from oracle import main_vm_native as vn
    d, D = vp.defs()
    cs = vp.vm_cs(LIMIT)
    ops, contracts = vp.program_calls(D, 3)
    n_inst = 4
    run = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), n_inst * LIMIT)
    outer, loop, reports = vp.pack_through_the_c_abi(cs, run, LIMIT, n_inst)
    assert not any(r.underflow for r in reports)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.seed_stream(n_inst, d_o, d_l)
    cs.set_batch(n_inst)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    ok, f = cs.resolve_and_check()
    assert ok, f
    hv = cs.hook_vars("hidden_fsm_output")
    assert len(hv) == 243
    expected = np.array([[int(x) for x in run.states[(i + 1) * LIMIT].flatten()] for i in range(n_inst)], dtype=np.uint64).T.copy()
    ok, where = cs.hook_compare_witness(hv, zk.DeviceBuffer.from_numpy(expected))
    assert ok, where
    bad = expected.copy(); bad[146, 2] ^= 1     # the timestamp of instance 2
    ok, where = cs.hook_compare_witness(hv, zk.DeviceBuffer.from_numpy(bad))
    assert not ok and where == (2, 146)
    assert len(cs.hook_vars("observable_output")) == 59   # VmOutputData: 9 + 25 + 25 (circuit_inputs/main_vm.rs:32-38)
    for i in range(n_inst):
        assert cs.public_inputs(i) == vp.expected_commitment(D, run, LIMIT, i)


def test_main_vm_deferred_poseidon2_intermediates(zk, batch, monkeypatch):
    """ZK_CHECK_FUSED_DEFER_P2: the loop kernel writes only the 12 final outputs of every in-circuit permutation; the verdict and the
    public inputs are the fused mode's, and the first reader of the store (here the trace) finds every intermediate regenerated bit
    for bit (k_fill_p2) — whole trace == oracle, the full re-evaluation passes, a corrupted intermediate is still caught"""
    cs, D, outer, loop, commits, info = batch
    B = outer.shape[1]
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    for strands in ("0", "1"):
        monkeypatch.setenv("ZKGL_STRANDS", strands)
        cs.set_check_mode(False, defer_p2=True)
        try:
            ok, f = cs.resolve_and_check()
            assert ok, f
            for i in (0, B - 1):
                assert cs.public_inputs(i) == commits[i]
            ok, f = cs.check_if_satisfied()          # every gate from the stored values: needs the intermediates
            assert ok, f
            run = run_oracle(cs, B)
            run.resolve(outer, loop)
            assert np.array_equal(cs.trace(True), run.lc), "loop-scope trace differs from the oracle after the deferred fill"
        finally:
            cs.set_check_mode(False)
    # a tampered carried word is reported in the deferred mode like in the others
    bad = loop.copy()
    lay = cs.main_vm_layout()["loop"]
    bad[lay["state"][0] + 9, 5 * LIMIT + 7] ^= 1
    d_b = zk.DeviceBuffer.from_numpy(bad)
    cs.bind_inputs(True, d_b, bad.shape[0])
    cs.set_check_mode(False, defer_p2=True)
    try:
        ok, f = cs.resolve_and_check()
        assert not ok and f.instance == 5
    finally:
        cs.set_check_mode(False)
