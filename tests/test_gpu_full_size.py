"""-m gpu: BASELINE.json's configurations at their FULL sizes through size-independent properties
(the oracle interpreter would take minutes there; parity at oracle-sized cases is in test_gpu_cs.py):

  C1  ram_permutation, 2^16 rows        : a permutation is accepted and its commitment equals the native restatement;
                                          swapping two sorted items (order broken) or changing a value is rejected
  C3  keccak256 + sha256 round functions, 2^20 rows : every digest the FSM writes equals the software hash (native
                                          model) and the device reproduces the native public input; satisfied
  C4  storage_validity + log_sorter, 2^22 rows      : accepted, commitments equal the native restatement; a broken
                                          permutation is rejected
  C2  main_vm, 2^20 rows (2 384 cycles)  : the raw witness of the committed fixture (8 executions of synthetic zkEVM programs,
                                          tests/golden/vm_bench_witness.npz) is seeded on the device, resolved and satisfied; the
                                          public inputs equal the native restatement's commitments stored with the fixture; the
                                          seeded stream is idempotent under a second seeding pass; a flipped oracle word is rejected
                                          at its instance
  C5 (eip_4844, 4096 chunks) is in test_gpu_cs.py.

Loop streams come from the native restatements (oracle/*_native.py: they hold the per-cycle state the device seeding
reproduces, which is checked at small sizes)."""
import hashlib

import numpy as np
import pytest

import zkgl
from oracle import keccak_native as kn
from oracle import log_sorter_native as ln
from oracle import ram_native as rn
from oracle import sha256_native as shn
from oracle import storage_native as sn
from oracle import zko

pytestmark = pytest.mark.gpu


def fit(configure, entry, log2_rows, cols=100):
    """largest `limit` whose trace fits 2^log2_rows rows, and the recorded circuit"""
    def make(limit, max_rows):
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(cols, 0, 8, 4), max_trace_len=max_rows, max_variables=1 << 28)
        configure(cs)
        entry(cs, limit)
        cs.pad_and_shrink()
        return cs
    probe = make(1, 1 << 30)
    st = probe.stats()
    limit = ((1 << log2_rows) - st["outer_slots"]) // st["loop_slots"]
    probe.close()
    cs = make(limit, 1 << log2_rows)
    assert cs.stats()["rows_per_instance"] <= 1 << log2_rows and cs.stats()["rows_per_instance"] > (1 << log2_rows) - 2 * st["loop_slots"]
    return cs, limit


def run_gpu(zk, cs, outer, loop, batch):
    cs.set_batch(batch)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    ok, f = cs.resolve_and_check()
    return ok, f, (d_o, d_l)


def assert_whole_trace_equals_oracle(zk, cs, outer, loop, k, limit, log_n):
    """instance k of the resolved batch: every cell of its trace columns (zk_cs_trace_columns: loop rows, outer rows, zero padding)
    against the oracle interpreter run on that instance's own streams — the witness columns at BASELINE's full size, not only the
    commitments they hash to"""
    st = cs.stats()
    tables = zko.parse_export(cs.export(False))["tables"]
    run = zko.CircuitRun(cs.export(False), cs.export(True), 1, int(sum(t["n_rows"] for t in tables)))
    run.resolve(np.ascontiguousarray(outer[:, k:k + 1]), np.ascontiguousarray(loop[:, k * limit:(k + 1) * limit]))
    n_cols, ls, osl = st["copy_columns"] + st["lookup_columns"], st["loop_slots"], st["outer_slots"]
    cols = zk.DeviceBuffer(n_cols << log_n)
    cs.trace_columns(k, cols, log_n)
    got = cols.to_numpy().reshape(n_cols, 1 << log_n)
    want_loop = run.lc[:ls * n_cols, :limit].reshape(ls, n_cols, limit).transpose(1, 2, 0).reshape(n_cols, limit * ls)
    assert np.array_equal(got[:, :limit * ls], want_loop), "loop rows of the full-size trace differ from the oracle interpreter"
    want_outer = run.oc[:osl * n_cols, 0].reshape(osl, n_cols).T
    assert np.array_equal(got[:, limit * ls:limit * ls + osl], want_outer), "outer rows of the full-size trace differ from the oracle"
    assert not got[:, limit * ls + osl:].any()


def test_c1_ram_permutation_2_16_rows(zk):
    cs, limit = fit(lambda c: c.configure_ram_permutation(), lambda c, l: c.ram_permutation_entry_point(l), 16)
    rng = np.random.default_rng(0xC1)
    insts = []
    for n in (limit, limit - 7, limit // 2, 0):
        u, s, nd = rn.random_ram_witness(rng, n, n_cells=64) if n else ([], [], 0)
        insts.append(rn.instance(u, s, limit, nd))
    assert all(i["completed"] for i in insts)
    outer, loop = rn.pack_streams(insts, limit)
    ok, f, keep = run_gpu(zk, cs, outer, loop, len(insts))
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["commitment"]
    assert_whole_trace_equals_oracle(zk, cs, outer, loop, 1, limit, 16)
    # order broken: swap two adjacent sorted items of instance 1 (heads/accumulators re-derived natively)
    u, s, nd = rn.random_ram_witness(np.random.default_rng(5), limit, n_cells=64)
    j = next(k for k in range(1, limit - 1) if s[k][:3] != s[k + 1][:3])
    s[j], s[j + 1] = s[j + 1], s[j]
    bad = rn.instance(u, s, limit, nd)
    outer_b, loop_b = rn.pack_streams([insts[0], bad], limit)
    ok, f, keep2 = run_gpu(zk, cs, outer_b, loop_b, 2)
    assert not ok and f.instance == 1
    del keep, keep2


def _keccak_requests(rng, limit):
    """lengths uniform in [0, 1024] B, misalignment uniform in [0, 31]; as many as fit `limit` cycles"""
    reqs, datas, cycles = [], [], 0
    while True:
        n, off = int(rng.integers(0, 1025)), int(rng.integers(0, 32))
        need = n // 136 + 2            # upper bound on the cycles a request takes (reads lag absorbs by < 1 cycle)
        if cycles + need > limit:
            break
        d = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        reqs.append(kn.request(d, timestamp=1 + 2 * len(reqs), input_page=100 + len(reqs), input_offset=off + 32 * int(rng.integers(0, 50)),
                               output_page=5000 + len(reqs), output_offset=len(reqs)))
        datas.append(d)
        cycles += need
    return reqs, datas


def test_c3_keccak256_round_function_2_20_rows(zk):
    cs, limit = fit(lambda c: c.configure_keccak(), lambda c, l: c.keccak256_round_function_entry_point(l), 20)
    insts = []
    for seed in (0xC3, 0xC3 + 1):
        reqs, datas = _keccak_requests(np.random.default_rng(seed), limit)
        inst = kn.instance(reqs, limit)
        assert inst["satisfiable"] and inst["fsm_out"]["completed"] == 1
        writes = [q for q in inst["pushed"] if q[3] == 1]
        assert [sum(l << (32 * i) for i, l in enumerate(q[5:13])).to_bytes(32, "big") for q in writes] == [zko.keccak256(d) for d in datas]
        insts.append(inst)
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    ok, f, keep = run_gpu(zk, cs, outer, loop, len(insts))
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    assert_whole_trace_equals_oracle(zk, cs, outer, loop, 1, limit, 20)
    loop[459, limit + 3] ^= 1   # a memory word read by instance 1 differs from the one its queue chain was built with
    ok, f, keep2 = run_gpu(zk, cs, outer, loop, len(insts))
    assert not ok and f.instance == 1
    del keep, keep2


def test_c3_sha256_round_function_2_20_rows(zk):
    cs, limit = fit(lambda c: c.configure_sha256(), lambda c, l: c.sha256_round_function_entry_point(l), 20)
    insts = []
    for seed in (0xC3 + 2, 0xC3 + 3):
        rng = np.random.default_rng(seed)
        reqs, msgs, cycles = [], [], 0
        while True:
            rounds = int(rng.integers(1, 17))                 # 1-16 rounds per request
            if cycles + rounds > limit:
                break
            n = 64 * rounds - 9 - int(rng.integers(0, 55))
            m = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
            r = shn.request(m, timestamp=1 + 2 * len(reqs), input_page=10 + len(reqs), input_offset=int(rng.integers(0, 1000)),
                            output_page=9000 + len(reqs), output_offset=len(reqs))
            assert r["rounds"] == rounds
            reqs.append(r); msgs.append(m); cycles += rounds
        inst = shn.instance(reqs, limit)
        assert inst["satisfiable"] and inst["fsm_out"]["completed"] == 1
        writes = [q for q in inst["pushed"] if q[3] == 1]
        assert [sum(l << (32 * i) for i, l in enumerate(q[5:13])).to_bytes(32, "big") for q in writes] == [hashlib.sha256(m).digest() for m in msgs]
        insts.append(inst)
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    ok, f, keep = run_gpu(zk, cs, outer, loop, len(insts))
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    assert_whole_trace_equals_oracle(zk, cs, outer, loop, 1, limit, 20)
    del keep


def test_c4_storage_validity_2_22_rows(zk):
    cs, limit = fit(lambda c: c.configure_storage_validity(),
                    lambda c, l: c.sort_and_deduplicate_storage_access_entry_point(l, True), 22)
    # BASELINE config C4: four sharded instances (one per GPU there; one batch of four different witnesses here)
    insts, wit = [], []
    for k in range(4):
        rng = np.random.default_rng(0xC4 + 16 * k)
        u, s = sn.random_storage_witness(rng, limit - 3 - 5 * k, n_cells=512 - 64 * k)
        inst = sn.instance(u, s, limit)
        assert inst["satisfiable"] and inst["completed"]
        insts.append(inst); wit.append((u, s))
    outer, loop = sn.pack_streams(insts, limit)
    ok, f, keep = run_gpu(zk, cs, outer, loop, 4)
    assert ok, f
    for k in range(4):
        assert cs.public_inputs(k) == insts[k]["commitment"]
    assert len({tuple(i["commitment"]) for i in insts}) == 4
    del keep
    u, s = wit[0]
    # not a permutation: one sorted record's written value changed -> grand products differ (entry-point check)
    q, ts = s[len(s) // 2]
    q = list(q); q[21] ^= 1
    s2 = list(s); s2[len(s) // 2] = (q, ts)
    bad = sn.instance(u, s2, limit)
    outer, loop = sn.pack_streams([bad], limit)
    ok, f, keep = run_gpu(zk, cs, outer, loop, 1)
    assert not ok
    del keep


def test_c4_log_sorter_2_22_rows(zk):
    cs, limit = fit(lambda c: c.configure_log_sorter(), lambda c, l: c.sort_and_deduplicate_events_entry_point(l), 22)
    insts = []
    for k in range(4):   # four sharded instances, as in BASELINE's C4
        rng = np.random.default_rng(0xC4 + 1 + 16 * k)
        u, s = ln.random_events(rng, int(limit / 1.1) - 8 - 3 * k, rollback_frac=0.1)   # 10 % rollbacks, paired
        assert len(u) <= limit
        inst = ln.instance(u, s, limit)
        assert inst["satisfiable"] and inst["completed"]
        insts.append(inst)
    outer, loop = ln.pack_streams(insts, limit)
    ok, f, keep = run_gpu(zk, cs, outer, loop, 4)
    assert ok, f
    for k in range(4):
        assert cs.public_inputs(k) == insts[k]["commitment"]
    del keep


def test_c2_main_vm_2_20_rows(zk):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    cs, limit = bench.build_main_vm_cs(zkgl, 20)
    st = cs.stats()
    assert st["rows_per_instance"] <= 1 << 20 and st["rows_per_instance"] > (1 << 20) - 2 * st["loop_slots"]
    B = 8
    outer, loop, expect = bench.main_vm_streams(zkgl, cs, limit, B)   # the fixture's VmCircuitWitnesses through zk_pack_main_vm_witness
    assert expect is not None, "the fixture was generated for another limit"
    assert not loop[0:bench.VM_STATE_WORDS].any()                      # raw witness only: the carried VmLocalState words are blank
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.seed_stream(B, d_o, d_l)
    seeded = d_l.to_numpy().reshape(loop.shape)
    assert np.array_equal(seeded[bench.VM_STATE_WORDS:], loop[bench.VM_STATE_WORDS:])   # seeding writes carried words only
    cs.seed_stream(B, d_o, d_l)                                         # idempotent: the same recurrence from the same raw words
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), seeded)
    cs.set_batch(B)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i in range(B):
        assert cs.public_inputs(i) == [int(x) for x in expect[i]], i
    assert int(cs.multiplicities(0).sum()) == st["lookups_per_instance"]
    # the same batch with EVERY relation re-evaluated from the stored values (ZK_CHECK_STORED = check_if_satisfied's semantics,
    # /root/reference/src/ram_permutation/mod.rs:556) and in the deferred-intermediates mode: same verdict, same public inputs
    for mode in ((True, False), (False, True)):
        cs.set_check_mode(mode[0], defer_p2=mode[1])
        try:
            ok, f = cs.resolve_and_check()
            assert ok, (mode, f)
            assert cs.public_inputs(B - 1) == [int(x) for x in expect[B - 1]]
        finally:
            cs.set_check_mode(False)
    ok, f = cs.resolve_and_check()
    assert ok, f
    # the WHOLE trace of one full-size instance — 164 columns x 2^20 rows, every cell — against the oracle interpreter run on that
    # instance's streams: the witness columns the prover would commit to (zk_cs_trace_columns), not only the commitments
    assert_whole_trace_equals_oracle(zk, cs, outer, seeded, 3, limit, 20)
    # a flipped limb of an opcode word in the middle of instance 5
    lay = cs.main_vm_layout()["loop"]
    bad = seeded.copy()
    bad[lay["code_word"][0] + 1, 5 * limit + limit // 2] ^= 1
    d_b = zk.DeviceBuffer.from_numpy(bad)
    cs.bind_inputs(True, d_b, bad.shape[0])
    ok, f = cs.resolve_and_check()
    assert not ok and f.instance == 5
    cs.set_check_mode(True)
    try:
        ok, f2 = cs.resolve_and_check()
        assert not ok and f2.instance == 5
    finally:
        cs.set_check_mode(False)
