"""Native seeding of the precompile FSM circuits (csrc/kernels_fsm_seed.hpp: one wavefront per instance walks
keccak256_precompile_inner / sha256_precompile_inner natively, reference src/keccak256_round_function/mod.rs:215-640,
src/sha256_round_function/mod.rs:139-340) against the recorded-cone seeding (ZKGL_SEED_NATIVE=0) and the native restatements
(oracle/*_native.py, test infrastructure): bit-equal carried words on start, continuation, multi-request, zero-length, unaligned
and empty instances; then the seeded stream is resolved, satisfied, and yields the restatement's commitments."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _seed_both_ways(zk, cs, outer, loop, carried):
    raw = loop.copy()
    raw[:carried, :] = 0
    cs.set_batch(outer.shape[1])
    d_o = zk.DeviceBuffer.from_numpy(outer)
    cs.bind_inputs(False, d_o, outer.shape[0])
    got = {}
    for mode in ("1", "0"):
        os.environ["ZKGL_SEED_NATIVE"] = mode
        try:
            d_l = zk.DeviceBuffer.from_numpy(raw)
            cs.bind_inputs(True, d_l, raw.shape[0])
            cs.seed_carried_inputs(d_l)
            got[mode] = d_l.to_numpy().reshape(raw.shape)
        finally:
            os.environ.pop("ZKGL_SEED_NATIVE", None)
    bad = sorted({int(w) for w in np.nonzero((got["1"] != loop).any(axis=1))[0]})
    assert not bad, f"native seeding: loop words {bad[:20]} differ from the restatement"
    assert np.array_equal(got["0"], loop), "cone seeding differs from the restatement"
    d_l = zk.DeviceBuffer.from_numpy(got["1"])
    cs.bind_inputs(True, d_l, raw.shape[0])
    ok, f = cs.resolve_and_check()
    assert ok, f
    return d_o, d_l


def test_keccak_fsm_native_seeding(zk):
    from oracle import keccak_native as kn
    from test_keccak_fsm_host import fsm_cs, make_requests, streams
    limit = 7
    cs = fsm_cs(limit)
    rng = np.random.default_rng(71)
    data = lambda n: bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
    insts = []
    for lengths, offsets in (((300, 0, 272, 1), (70, 5, 0, 63)), ((135,), (31,)), ((136, 136), (0, 1)), ((), ()), ((700,), (17,)), ((1, 2, 3, 4, 5), (0, 31, 32, 33, 95)),
                             ((408,), (0,)), ((271, 137), (8, 24))):
        reqs = make_requests([data(n) for n in lengths], list(offsets))
        a = kn.instance(reqs, limit)
        insts.append(a)
        if not a["fsm_out"]["completed"]:      # its continuation: FSM input from the previous instance, the rest of the queue
            b = kn.instance(a["rest"][0], limit, start_flag=False, fsm_in=a["fsm_out"], obs_req=a["obs_req"], obs_mem=a["obs_mem"], pending=a["rest"][1])
            insts.append(b)
    assert any(not i["outer"][0] for i in insts), "no continuation instance in the batch"
    insts = insts * 7                          # > one wave tile of instances
    outer, loop = streams(insts, limit)
    _seed_both_ways(zk, cs, outer, loop, kn.CARRIED)
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]


def test_sha256_fsm_native_seeding(zk):
    from oracle import sha256_native as sn
    from test_sha256_fsm_host import fsm_cs, make_requests, messages, streams
    limit = 9
    cs = fsm_cs(limit)
    rng = np.random.default_rng(72)
    insts = []
    for lengths in ((3,), (0, 55, 56), (150, 64), (), (119, 1), (600,), (64, 64, 64, 64, 64), (1000,)):
        reqs = make_requests(messages(rng, lengths))
        a = sn.instance(reqs, limit)
        insts.append(a)
        if not a["fsm_out"]["completed"]:
            insts.append(sn.instance(a["rest"][0], limit, start_flag=False, fsm_in=a["fsm_out"], obs_req=a["obs_req"], obs_mem=a["obs_mem"], pending=a["rest"][1]))
    assert any(not i["outer"][0] for i in insts)
    insts = insts * 8
    outer, loop = streams(insts, limit)
    _seed_both_ways(zk, cs, outer, loop, sn.CARRIED)
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]


@pytest.mark.parametrize("n_chunks", [27, 100, 9])
def test_eip4844_native_seeding(zk, n_chunks):
    """eip_4844: the sponge lane + the Horner lane (Montgomery products in the BLS12-381 scalar field) against the cone and the restatement"""
    import zkgl
    from oracle import eip4844_native as en
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_eip_4844()
    cs.eip_4844_entry_point(n_chunks)
    cs.pad_and_shrink()
    rng = np.random.default_rng(4844 + n_chunks)
    insts = []
    for k in range(70):
        blob = bytes(rng.integers(0, 256, size=31 * n_chunks, dtype=np.uint8)) if k % 5 else b"\xff" * (31 * n_chunks)
        vh = b"\x01" + bytes(rng.integers(0, 256, size=31, dtype=np.uint8))
        insts.append(en.instance(blob, vh, n_chunks))
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    _seed_both_ways(zk, cs, outer, loop, 217)
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
