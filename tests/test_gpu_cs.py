"""-m gpu: the engine on a real MI355X through the C ABI — witness generation bit-exact against the
CPU oracle interpreter (whole trace, every cell), check_if_satisfied agreeing with the oracle
checker, commitments equal to the native restatement, fault injection reporting the failing place."""
import json
import os

import numpy as np
import pytest

import zkgl
from helpers import GOLD, LINK, G, P, Rec, load_fixture, new_cs, oracle_run, ram_cs, rand_fe, random_instances
from oracle import ram_native as rn
from oracle import zko
from test_cs_host import all_ops_circuit, all_ops_inputs

pytestmark = pytest.mark.gpu


def gpu_run(zk, cs, outer, loop, batch):
    cs.set_batch(batch)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    cs.resolve()
    return d_o, d_l


def assert_trace_equal(cs, run):
    assert np.array_equal(cs.trace(False), run.oc), "outer-scope trace differs from the oracle"
    got = cs.trace(True)
    assert got.shape == run.lc.shape or run.lc.shape[0] == 1
    if got.size:
        assert np.array_equal(got, run.lc), "loop-scope trace differs from the oracle"


def test_ram_fixture_trace_bit_exact(zk):
    u, s, limit = load_fixture()
    cs = ram_cs(limit)
    inst = rn.instance(u, s, limit, 1)
    outer, loop = rn.pack_streams([inst], limit)
    keep = gpu_run(zk, cs, outer, loop, 1)
    run = oracle_run(cs, outer, loop, 1)
    assert_trace_equal(cs, run)
    ok, f = cs.check_if_satisfied()
    assert ok, f
    assert cs.public_inputs(0) == inst["commitment"]
    gold = json.load(open(os.path.join(GOLD, "ram_commitments.json")))
    assert cs.public_inputs(0) == [int(x, 16) for x in gold["fixture_limit16"]]
    assert np.array_equal(cs.multiplicities(0), run.mult[:65536])
    assert int(cs.multiplicities(0).sum()) == cs.stats()["lookups_per_instance"]
    del keep


def test_ram_batch_of_instances(zk):
    limit, batch = 16, 37  # batch*limit = 592 lanes: not a multiple of the wave or the workgroup
    cs = ram_cs(limit)
    insts = random_instances(11, batch, 13, limit)
    insts[3] = rn.instance([], [], limit, 0)                      # empty queue
    rng = np.random.default_rng(5)
    u, s, nd = rn.random_ram_witness(rng, limit)
    insts[7] = rn.instance(u, s, limit, nd)                        # queue exactly fills the chunk
    outer, loop = rn.pack_streams(insts, limit)
    keep = gpu_run(zk, cs, outer, loop, batch)
    run = oracle_run(cs, outer, loop, batch)
    assert_trace_equal(cs, run)
    assert run.check()[0] == 0
    ok, f = cs.check_if_satisfied()
    assert ok, f
    for i in (0, 3, 7, batch - 1):
        assert cs.public_inputs(i) == insts[i]["commitment"]
        assert np.array_equal(cs.multiplicities(i), run.mult[i * 65536:(i + 1) * 65536])
    del keep


def test_ram_unsatisfied_witnesses_are_rejected_like_the_oracle(zk):
    u, s, limit = load_fixture()
    cs = ram_cs(limit)
    good = rn.instance(u, s, limit, 1)
    s_bad = [list(x) for x in s]; s_bad[1][5] ^= 4; s_bad[2][5] ^= 4
    not_perm = rn.instance(u, s_bad, limit, 1)
    not_sorted = rn.instance(u, [s[1], s[0], s[2]], limit, 1)
    insts = [good, not_perm, good, not_sorted]
    outer, loop = rn.pack_streams(insts, limit)
    keep = gpu_run(zk, cs, outer, loop, 4)
    run = oracle_run(cs, outer, loop, 4)
    assert_trace_equal(cs, run)     # the witness is deterministic even when it does not satisfy
    assert run.check()[0] > 0
    ok, f = cs.check_if_satisfied()
    assert not ok and f.instance == 1  # first failing instance
    del keep


def test_fault_injection_reports_place(zk):
    """flip one cell => first failing (scope, instance, iteration, slot) is reported"""
    limit, batch = 8, 3
    cs = ram_cs(limit)
    insts = random_instances(3, batch, 6, limit)
    outer, loop = rn.pack_streams(insts, limit)
    keep = gpu_run(zk, cs, outer, loop, batch)
    assert cs.check_if_satisfied()[0]
    st = cs.stats()
    n_cols = st["copy_columns"] + st["lookup_columns"]
    tr = cs.trace(True)
    # pick a populated trace cell in the loop scope: column 2, slot 5, lane of (instance 1, iteration 4)
    cell, lane = 5 * n_cols + 2, 1 * limit + 4
    old = int(tr[cell, lane])
    cs.write_cell(True, cell, lane, (old + 1) % P)
    ok, f = cs.check_if_satisfied()
    assert not ok and (f.scope, f.instance, f.iteration) == (1, 1, 4)
    cs.write_cell(True, cell, lane, old)
    assert cs.check_if_satisfied()[0]
    # outer scope
    tro = cs.trace(False)
    cell_o = 3 * n_cols + 1
    old = int(tro[cell_o, 2])
    cs.write_cell(False, cell_o, 2, old ^ 1)
    ok, f = cs.check_if_satisfied()
    assert not ok and (f.scope, f.instance) == (0, 2)
    cs.write_cell(False, cell_o, 2, old)
    # carried-state seed tampering is caught by the link (copy-constraint) check
    loop_bad = loop.copy(); loop_bad[27, 2 * limit + 3] ^= 1   # lhs accumulator entering iteration 3 of instance 2
    d_l = zk.DeviceBuffer.from_numpy(loop_bad)
    cs.bind_inputs(True, d_l, loop.shape[0])
    cs.resolve()
    ok, f = cs.check_if_satisfied()
    assert not ok and f.instance == 2
    del keep


def test_copy_constraint_failures_name_the_pair(zk):
    """The gate checker verifies the copy constraint of every cell it reads (chain partner = nearest earlier cell of the
    variable); cells of rows without relations are left to k_check_copies.  Flipping a boolean cell keeps its gate
    satisfied, so the report must be the copy pair (kind 0x200, slot = the pair's cell, relation = pair index) — the
    smallest index among the pairs the cell takes part in, as the oracle's pair scan finds it."""
    limit, batch = 8, 3
    cs = ram_cs(limit)
    insts = random_instances(5, batch, 6, limit)
    outer, loop = rn.pack_streams(insts, limit)
    keep = gpu_run(zk, cs, outer, loop, batch)
    assert cs.check_if_satisfied()[0]
    st = cs.stats()
    n_cols = st["copy_columns"] + st["lookup_columns"]
    h = zko.parse_export(cs.export(True))
    tr = cs.trace(True)
    pairs = h["copies"]
    in_pair = {}
    for pi, (c, partner) in enumerate(pairs):
        in_pair.setdefault(c, []).append(pi)
        in_pair.setdefault(partner, []).append(pi)
    tried = 0
    for slot, (kind, ninst, _, _) in enumerate(h["rows"]):
        if kind != G["BOOLEAN"]:
            continue
        for j in range(ninst):
            cell, lane = slot * n_cols + j, 2 * limit + 5
            if cell not in in_pair:
                continue
            old = int(tr[cell, lane])
            assert old in (0, 1)
            cs.write_cell(True, cell, lane, 1 - old)
            ok, f = cs.check_if_satisfied()
            want = min(in_pair[cell])
            assert not ok and (f.scope, f.instance, f.iteration, f.kind) == (1, 2, 5, 0x200), (f.scope, f.instance, f.iteration, hex(f.kind))
            assert (f.relation, f.slot) == (want, pairs[want][0])
            cs.write_cell(True, cell, lane, old)
            tried += 1
            if tried == 6:
                break
        if tried == 6:
            break
    assert tried > 0 and cs.check_if_satisfied()[0]
    # a public-input cell sits in a row without relations: only the residual pair list covers it
    ho = zko.parse_export(cs.export(False))
    pub_slot = [slot for slot, row in enumerate(ho["rows"]) if row[0] == G["PUBLIC_INPUT"]][0]
    pub = pub_slot * n_cols + 2
    mine = [pi for pi, (c, partner) in enumerate(ho["copies"]) if pub in (c, partner)]
    assert mine
    tro = cs.trace(False)
    old = int(tro[pub, 1])
    cs.write_cell(False, pub, 1, (old + 1) % P)
    ok, f = cs.check_if_satisfied()
    assert not ok and (f.scope, f.instance, f.kind, f.relation) == (0, 1, 0x200, min(mine))
    cs.write_cell(False, pub, 1, old)
    assert cs.check_if_satisfied()[0]
    del keep


def test_all_ops_circuit_gpu_equals_oracle(zk):
    """every op / gate kind; carried state seeded by the GPU's sequential mode == the oracle's"""
    limit, batch = 4, 130
    rng = np.random.default_rng(17)
    cs = new_cs()
    n_outer, n_loop = all_ops_circuit(cs, limit)
    cs.pad_and_shrink()
    outer, loop_raw = all_ops_inputs(rng, batch, limit, n_outer, n_loop)
    seeded_oracle = zko.CircuitRun(cs.export(False), cs.export(True), batch, 306).seed(outer, loop_raw)
    cs.set_batch(batch)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop_raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop_raw.shape[0])
    cs.seed_carried_inputs(d_l)
    loop = d_l.to_numpy().reshape(loop_raw.shape)
    assert np.array_equal(loop, seeded_oracle)
    cs.resolve()
    run = oracle_run(cs, outer, loop, batch, 306)
    assert_trace_equal(cs, run)
    assert run.check()[0] == 0
    ok, f = cs.check_if_satisfied()
    assert ok, f
    for i in (0, 1, batch - 1):
        assert np.array_equal(cs.multiplicities(i), run.mult[i * 306:(i + 1) * 306])


@pytest.mark.parametrize("mode", ["cone", "generic"])
def test_ram_generic_seeding_equals_native_streams(zk, mode, monkeypatch):
    """raw witness only (items + is_first flag): the engine's sequential seeding reproduces the
    per-iteration state the native restatement derives from the reference code — both with the cone program
    (backward slice of the carried outputs over LDS slots, k_seed_cone) and with the generic mode (k_witness_seq)"""
    monkeypatch.setenv("ZKGL_SEED_GENERIC", "1" if mode == "generic" else "0")
    limit, batch = 8, 21
    assert ram_cs(limit).stats()["seed_ops"] > 0
    cs = ram_cs(limit)
    insts = random_instances(31, batch, 6, limit)
    outer, loop = rn.pack_streams(insts, limit)
    raw = loop.copy()
    raw[0:46] = 0  # drop every carried word (layout: DESIGN.md, ram_permutation loop stream)
    cs.set_batch(batch)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
    cs.resolve()
    assert cs.check_if_satisfied()[0]
    for i in range(batch):
        assert cs.public_inputs(i) == insts[i]["commitment"]


def test_lookup_absent_key_gpu(zk):
    cs = new_cs()
    rows = np.array([[k * 7 + 3, k] for k in range(50)], dtype=np.uint64)
    t = cs.add_lookup_table(5, 1, 1, rows)
    key = cs.input(0)
    cs.perform_lookup(t, [key], 1)
    cs.pad_and_shrink()
    keys = np.array([[10, 11, 3 + 49 * 7, 3 + 50 * 7, P - 1]], dtype=np.uint64)
    cs.set_batch(5)
    d = zk.DeviceBuffer.from_numpy(keys)
    cs.bind_inputs(False, d, 1)
    cs.resolve()
    ok, f = cs.check_if_satisfied()
    assert not ok and f.instance == 1 and f.kind == 0x100


def test_resolve_is_repeatable_and_timed(zk):
    limit, batch = 16, 8
    cs = ram_cs(limit)
    insts = random_instances(23, batch, 16, limit)
    outer, loop = rn.pack_streams(insts, limit)
    keep = gpu_run(zk, cs, outer, loop, batch)
    t1 = cs.trace(True).copy()
    cs.resolve()
    assert np.array_equal(cs.trace(True), t1)
    assert cs.last_ms(0) > 0 and cs.last_ms(1) > 0
    assert cs.check_if_satisfied()[0] and cs.last_ms(2) > 0
    del keep


def test_vm_shaped_gpu_equals_oracle(zk):
    from vm_shaped_fixture import vm_inputs
    from test_cs_host import VM_TABLE_ROWS, vm_cs
    limit, batch = 5, 70
    cs = vm_cs(limit)
    n_outer, n_loop = cs.input_words()
    outer, loop_raw = vm_inputs(np.random.default_rng(0xC2), n_outer, n_loop, batch, limit)
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), batch, VM_TABLE_ROWS).seed(outer, loop_raw)
    cs.set_batch(batch)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop_raw)
    cs.bind_inputs(False, d_o, n_outer)
    cs.bind_inputs(True, d_l, n_loop)
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(loop_raw.shape), seeded)
    cs.resolve()
    run = oracle_run(cs, outer, seeded, batch, VM_TABLE_ROWS)
    assert_trace_equal(cs, run)
    ok, f = cs.check_if_satisfied()
    assert ok, f
    assert run.check()[0] == 0
    for i in (0, batch - 1):
        assert cs.public_inputs(i) == [int(run.oc[c, i]) for c in cs.public_cells()]
        assert np.array_equal(cs.multiplicities(i), run.mult[i * VM_TABLE_ROWS:(i + 1) * VM_TABLE_ROWS])


def test_fused_pipeline_equals_separate_calls(zk):
    limit, batch = 8, 9
    cs = ram_cs(limit)
    insts = random_instances(41, batch, 7, limit)
    outer, loop = rn.pack_streams(insts, limit)
    keep = gpu_run(zk, cs, outer, loop, batch)
    ref_loop, ref_outer = cs.trace(True).copy(), cs.trace(False).copy()
    assert cs.check_if_satisfied()[0]
    ok, f = cs.resolve_and_check()
    assert ok, f
    assert np.array_equal(cs.trace(True), ref_loop) and np.array_equal(cs.trace(False), ref_outer)
    assert cs.last_ms(0) > 0 and cs.last_ms(1) > 0
    for i in range(batch):
        assert cs.public_inputs(i) == insts[i]["commitment"]
    # an unsatisfiable instance is reported by the fused path as well
    u, s, limit16 = load_fixture()
    bad = rn.instance(u, [s[1], s[0], s[2]], limit, 1)
    insts[4] = bad
    outer, loop = rn.pack_streams(insts, limit)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    ok, f = cs.resolve_and_check()
    assert not ok and f.instance == 4
    del keep


def test_storage_validity_gpu_equals_oracle(zk):
    """config C4 circuit: reference fixture (inner logic) + seeded random logs, GPU trace bit-exact vs oracle"""
    from helpers import load_storage_fixture, storage_cs
    from oracle import storage_native as sn
    unsorted, sorted_records, limit = load_storage_fixture()
    cs = storage_cs(limit, enforce_permutation=False)
    rng = np.random.default_rng(8)
    insts = [sn.instance(unsorted, sorted_records, limit, enforce_permutation=False)]
    for n in (0, 5, 16, 11):
        u, s = sn.random_storage_witness(rng, n)
        insts.append(sn.instance(u, s, limit))
    outer, loop = sn.pack_streams(insts, limit)
    keep = gpu_run(zk, cs, outer, loop, len(insts))
    run = oracle_run(cs, outer, loop, len(insts))
    assert_trace_equal(cs, run)
    assert run.check()[0] == 0
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["commitment"]
    # entry-point semantics (lhs == rhs enforced): the non-permutation fixture must now be rejected, instance 0
    cs2 = storage_cs(limit, enforce_permutation=True)
    keep2 = gpu_run(zk, cs2, outer, loop, len(insts))
    ok, f = cs2.check_if_satisfied()
    assert not ok and f.instance == 0 and f.scope == 0
    del keep, keep2


def test_keccak256_gpu_digests(zk):
    """K8 on the GPU: lookup-table Keccak-f, digests equal the software Keccak-256 for the reference's lengths
    (src/keccak256_round_function/mod.rs:1096-1144), trace bit-exact vs the oracle interpreter"""
    from test_keccak_host import TABLE_ROWS, keccak_cs, loop_stream
    n_blocks = 2
    cs = keccak_cs(n_blocks)
    rng = np.random.default_rng(136)
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (136, 166, 180, 200, 137, 271)] * 11   # 66 instances
    outer = np.zeros((0, len(msgs)), dtype=np.uint64)
    raw = loop_stream(msgs, n_blocks)
    cs.set_batch(len(msgs))
    d_l = zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, m in enumerate(msgs):
        assert bytes(cs.public_inputs(i)) == zko.keccak256(m)
    seeded = d_l.to_numpy().reshape(raw.shape)
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), TABLE_ROWS)
    run.resolve(outer, seeded)
    assert_trace_equal(cs, run)
    assert np.array_equal(cs.multiplicities(3), run.mult[3 * TABLE_ROWS:4 * TABLE_ROWS])


def test_sha256_gpu_digests(zk):
    import hashlib
    from test_sha256_host import TABLE_ROWS, loop_stream, sha_cs
    n_blocks = 2
    cs = sha_cs(n_blocks)
    rng = np.random.default_rng(256)
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (56, 64, 100, 119, 57)] * 14   # 70 instances
    raw = loop_stream(msgs, n_blocks)
    cs.set_batch(len(msgs))
    d_l = zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, m in enumerate(msgs):
        assert bytes(cs.public_inputs(i)) == hashlib.sha256(m).digest()
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), TABLE_ROWS)
    run.resolve(np.zeros((0, len(msgs)), dtype=np.uint64), d_l.to_numpy().reshape(raw.shape))
    assert_trace_equal(cs, run)


def test_log_sorter_gpu_equals_oracle(zk):
    from oracle import log_sorter_native as ln
    from test_log_sorter_host import load_log_sorter_fixture, log_sorter_cs
    u, s, limit = load_log_sorter_fixture()
    cs = log_sorter_cs(limit)
    rng = np.random.default_rng(4)
    insts = [ln.instance(u, s, limit), ln.instance([], [], limit)]
    for n in (3, 7, 5):
        uu, ss = ln.random_events(rng, n, rollback_frac=0.4)
        insts.append(ln.instance(uu, ss, limit))
    assert all(i["satisfiable"] for i in insts)
    outer, loop = ln.pack_streams(insts, limit)
    keep = gpu_run(zk, cs, outer, loop, len(insts))
    run = oracle_run(cs, outer, loop, len(insts))
    assert_trace_equal(cs, run)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["commitment"]
    del keep


def test_sha256_round_function_fsm_gpu(zk):
    """a18 on the GPU: the precompile FSM (request pop, 2 reads + 1 write per cycle, compression); carried words
    seeded on the device from the raw request / memory-read stream; trace bit-exact vs the oracle interpreter,
    public inputs equal the native restatement, digests in the pushed writes equal hashlib (native side)."""
    import hashlib
    from oracle import sha256_native as sn
    from test_sha256_fsm_host import TABLE_ROWS, fsm_cs, make_requests, messages, streams
    limit = 5
    cs = fsm_cs(limit)
    rng = np.random.default_rng(18)
    insts, all_msgs = [], []
    for k in range(66):
        lengths = [(3,), (0, 55, 56), (150, 64), (), (119, 1)][k % 5]
        msgs = messages(rng, lengths)
        insts.append(sn.instance(make_requests(msgs), limit))
        all_msgs.append(msgs)
    for inst, msgs in zip(insts, all_msgs):
        writes = [q for q in inst["pushed"] if q[3] == 1]
        assert [sum(l << (32 * i) for i, l in enumerate(q[5:13])).to_bytes(32, "big") for q in writes] == \
            [hashlib.sha256(m).digest() for m in msgs]
    outer, loop = streams(insts, limit)
    raw = loop.copy()
    raw[:sn.CARRIED, :] = 0
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(raw.shape), loop), "device seeding differs from the native FSM trajectory"
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS)
    run.resolve(outer, loop)
    assert_trace_equal(cs, run)
    # a corrupted memory read is rejected in the loop scope of the right instance
    bad = loop.copy()
    bad[97, 7 * limit] ^= 1
    d_b = zk.DeviceBuffer.from_numpy(bad)
    cs.bind_inputs(True, d_b, bad.shape[0])
    ok, f = cs.resolve_and_check()
    assert not ok and f.instance == 7


def test_keccak256_round_function_fsm_gpu(zk):
    """a17 on the GPU: the ten reference cases (src/keccak256_round_function/mod.rs:1096-1144) plus multi-request
    instances, x8 to fill two wave tiles; carried words seeded on the device; trace bit-exact vs the oracle interpreter."""
    from oracle import keccak_native as kn
    from test_keccak_fsm_host import REFERENCE_CASES, TABLE_ROWS, fsm_cs, make_requests, reference_case, streams
    limit = 2
    cs = fsm_cs(limit)
    cases = [reference_case(l, u) for l, u in REFERENCE_CASES]
    for data, inst in cases:
        q = inst["pushed"][-1]
        assert q[3] == 1 and sum(l << (32 * i) for i, l in enumerate(q[5:13])).to_bytes(32, "big") == zko.keccak256(data)
    insts = [c[1] for c in cases]
    rng = np.random.default_rng(17)
    insts.append(kn.instance(make_requests([bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (0, 40)], [9, 33]), limit))
    insts.append(kn.instance([], limit))
    insts = insts * 6   # 66 instances
    outer, loop = streams(insts, limit)
    raw = loop.copy()
    raw[:kn.CARRIED, :] = 0
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(raw.shape), loop), "device seeding differs from the native FSM trajectory"
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS)
    run.resolve(outer, loop)
    assert_trace_equal(cs, run)
    # the strand form of the programs (8 wavefronts per tile, barriers between dependency levels) and the plain form fill
    # the same cells: force each in turn (by default this circuit's loop body runs as strands, its outer phases do not)
    traces = {}
    for mode in ("0", "1"):
        os.environ["ZKGL_STRANDS"] = mode
        try:
            cs.resolve()
            ok, f = cs.check_if_satisfied()
            assert ok, (mode, f)
            traces[mode] = (cs.trace(False).copy(), cs.trace(True).copy())
        finally:
            del os.environ["ZKGL_STRANDS"]
    assert np.array_equal(traces["0"][0], traces["1"][0]) and np.array_equal(traces["0"][1], traces["1"][1])
    assert np.array_equal(traces["1"][1], run.lc) and np.array_equal(traces["1"][0], run.oc)
    bad = loop.copy()
    bad[460, 12 * limit] ^= 1   # instance 12 = reference case (180, 0): corrupt its first memory read
    d_b = zk.DeviceBuffer.from_numpy(bad)
    cs.bind_inputs(True, d_b, bad.shape[0])
    ok, f = cs.resolve_and_check()
    assert not ok and f.instance == 12


def test_bigint_witness_ops_gpu(zk):
    """ZK_OP_NN_MULMOD (Knuth D on the device, bit-serial division in the oracle) and ZK_OP_DIVREM vs Python integers"""
    from test_eip4844_host import bigint_ops_cs, bigint_ops_expected, bigint_ops_inputs
    cs, outs, qr = bigint_ops_cs()
    inp = bigint_ops_inputs(192)
    cs.set_batch(inp.shape[1])
    d = zk.DeviceBuffer.from_numpy(inp)
    cs.bind_inputs(False, d, inp.shape[0])
    cs.resolve()
    tr = cs.trace(False)
    for i in range(inp.shape[1]):
        e_nn, e_dr = bigint_ops_expected(inp, i)
        assert [int(tr[cs.var_cell(v), i]) for v in outs] == e_nn
        assert [int(tr[cs.var_cell(v), i]) for v in qr] == e_dr


def test_eip4844_gpu(zk):
    """a19 on the GPU: 27-chunk blobs (7 Keccak blocks, partially active last Horner iteration), 66 instances; carried
    words seeded on the device; trace bit-exact vs the oracle; the stream link catches a blob byte that differs between
    its block view and its chunk view"""
    from test_eip4844_host import TABLE_ROWS, blob_cs, make_instances, streams
    n_chunks = 27
    cs = blob_cs(n_chunks)
    insts = make_instances(n_chunks, range(66))
    outer, loop = streams(insts)
    limit = cs.stats()["limit"]
    raw = loop.copy()
    raw[:217, :] = 0
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(raw.shape), loop), "device seeding differs from the native trajectory"
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS)
    run.resolve(outer, loop)
    assert_trace_equal(cs, run)
    bad = loop.copy()
    bad[217 + 136 + 40, 5 * limit + 1] ^= 1    # chunk view of a blob byte of instance 5, iteration 1
    d_b = zk.DeviceBuffer.from_numpy(bad)
    cs.bind_inputs(True, d_b, bad.shape[0])
    ok, f = cs.resolve_and_check()
    assert not ok and f.instance == 5


def test_eip4844_full_size_blobs_gpu(zk):
    """the reference's size (4096 chunks, 934 blocks, ~1.17 M rows per blob): BASELINE C5's 8 blobs seeded, resolved and checked on the
    device; linear hash / opening value / output hash agree with keccak256 + Python big-integer Horner through the public input"""
    from test_eip4844_host import make_instances, streams
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4), 1 << 21, 1 << 28)
    cs.configure_eip_4844()
    cs.eip_4844_entry_point(4096)
    cs.pad_and_shrink()
    insts = make_instances(4096, [11, 12, 13, 14, 15, 16, 17, 18])
    outer, loop = streams(insts)
    raw = loop.copy()
    raw[:217, :] = 0
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(raw.shape), loop)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]


def test_demux_log_queue_gpu(zk):
    """8(f)-1 on the GPU: reference fixture + random mixes of the six classes, device seeding, trace bit-exact vs oracle"""
    from oracle import demux_native as dn
    from test_demux_host import demux_cs, load_demux_fixture, random_queries, streams
    qs, limit = load_demux_fixture()
    cs = demux_cs(limit)
    rng = np.random.default_rng(66)
    insts = [dn.instance(qs, limit)] + [dn.instance(random_queries(rng, int(rng.integers(0, limit + 1))), limit) for _ in range(69)]
    assert all(i["satisfiable"] for i in insts)
    outer, loop = streams(insts, limit)
    raw = loop.copy()
    raw[:dn.CARRIED] = 0
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(raw.shape), loop)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), 65536)
    run.resolve(outer, loop)
    assert_trace_equal(cs, run)


def test_sort_decommittment_requests_gpu(zk):
    """8(f)-1 on the GPU: reference fixture (partial pass) + random request logs, device seeding, trace bit-exact vs oracle"""
    from oracle import decommit_native as dn
    from test_decommit_host import decommit_cs, load_decommit_fixture, streams
    u, s, limit = load_decommit_fixture()
    cs = decommit_cs(limit)
    rng = np.random.default_rng(67)
    insts = [dn.instance(u, s, limit)]
    while len(insts) < 66:
        uu, ss = dn.random_decommits(rng, int(rng.integers(1, 6)), max_repeats=3)
        if len(uu) <= limit:
            insts.append(dn.instance(uu, ss, limit))
    assert all(i["satisfiable"] for i in insts)
    outer, loop = streams(insts, limit)
    raw = loop.copy()
    raw[:dn.CARRIED] = 0
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(raw.shape), loop)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), 65536)
    run.resolve(outer, loop)
    assert_trace_equal(cs, run)


def test_code_unpacker_sha256_gpu(zk):
    """8(f)-4 on the GPU: the reference's SHA-256 known-answer fixture + random bytecodes; device seeding; trace parity"""
    from oracle import code_unpacker_native as cn
    from oracle.decommit_native import dq
    from test_code_unpacker_host import TABLE_ROWS, load_code_unpacker_fixture, random_code, streams, unpacker_cs
    req, words, limit = load_code_unpacker_fixture()
    cs = unpacker_cs(limit)
    rng = np.random.default_rng(68)
    insts = [cn.instance([(req, words)], limit)]
    while len(insts) < 66:
        reqs, rounds = [], 0
        while True:
            n = 2 * int(rng.integers(0, 8)) + 1
            if rounds + (n + 1) // 2 > limit:
                break
            w = random_code(rng, n)
            reqs.append((dq(cn.versioned_hash(w), 2048 + 8 * len(reqs), 1, 5 + len(reqs)), w))
            rounds += (n + 1) // 2
            if rng.random() < 0.3:
                break
        insts.append(cn.instance(reqs, limit))
    assert all(i["satisfiable"] for i in insts)
    outer, loop = streams(insts, limit)
    raw = loop.copy()
    raw[:cn.CARRIED] = 0
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(raw.shape), loop)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS)
    run.resolve(outer, loop)
    assert_trace_equal(cs, run)


def test_linear_hasher_gpu(zk):
    """8(f)-4 on the GPU: 17-cycle period body (28 Keccak permutations, 10.7 M cells per lane -> the `_wide` interpreter
    kernels with 64-bit addressing); digests equal software Keccak through the public input; device seeding"""
    from oracle import linear_hasher_native as hn
    from test_linear_hasher_host import TABLE_ROWS, hasher_cs, random_messages, streams
    cs = hasher_cs(17)
    rng = np.random.default_rng(90)
    insts = []
    for k in range(66):
        qs = random_messages(rng, [0, 1, 3, 17, 9, 16][k % 6])
        inst = hn.instance(qs, 17)
        assert inst["satisfiable"] and inst["digest"] == zko.keccak256(b"".join(hn.into_bytes(q) for q in qs))
        insts.append(inst)
    outer, loop = streams(insts)
    raw = loop.copy()
    raw[:hn.CARRIED] = 0
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, raw.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(raw.shape), loop)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]
    # oracle parity on the first wave tile only (the oracle needs ~40 s per 64 lanes of this circuit)
    run = zko.CircuitRun(cs.export(False), cs.export(True), 2, TABLE_ROWS)
    run.resolve(outer[:, :2], loop[:, :2])
    assert np.array_equal(cs.trace(True)[:, :2], run.lc[:, :2]) and np.array_equal(cs.trace(False)[:, :2], run.oc[:, :2])


def test_narrow_strand_form_gpu(zk, monkeypatch):
    """loop scopes with a wide op graph keep a second strand program dealt over 8 wavefronts per tile (launch_phase picks it when
    the tiles outnumber what the chip keeps resident at 16); forced here at test sizes: same traces, digests and verdicts"""
    monkeypatch.setenv("ZKGL_STRANDS", "1")
    monkeypatch.setenv("ZKGL_STRANDS_NARROW", "1")
    test_keccak256_round_function_fsm_gpu(zk)
    test_sha256_round_function_fsm_gpu(zk)
    test_eip4844_gpu(zk)
    test_keccak256_gpu_digests(zk)
