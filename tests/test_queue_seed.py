"""Chain-free seeding of the LogQuery sorters (storage_validity_by_grand_product, log_sorter).

The reference's queue witnesses hold (item, previous tail) pairs (CircuitQueueRawWitness, src/storage_validity_by_grand_product/input.rs:131-136,
src/log_sorter/input.rs:101-106): with them zk_pack_storage_witness / zk_pack_log_sorter_witness walk the integer carried state on the
host, and the device computes the grand-product accumulators as scans (csrc/kernels_queue_seed.hpp) and — unless the host also supplied
them — the output queue's tails.  CPU: the packer's words == the native restatement's.  GPU: the seeded streams == the restatement's
(start instances) and == the recorded-cone seeding (continuation instances, which the restatements do not model)."""
import ctypes as C

import numpy as np
import pytest

import zkgl
from oracle import log_sorter_native as ln, storage_native as sn, zko
from test_witness_pack import _lq, _q4

LIMIT = 40


def _tails_array(tails):
    arr = ((C.c_uint64 * 4) * max(len(tails), 1))()
    for a, t in zip(arr, tails):
        a[:] = [int(x) for x in t]
    return arr


def _storage_fsm(f, x):
    f.lhs_accumulator[:] = x[0:2]; f.rhs_accumulator[:] = x[2:4]
    f.current_unsorted_queue_state, f.current_intermediate_sorted_queue_state, f.current_final_sorted_queue_state = _q4(x[4:13]), _q4(x[13:22]), _q4(x[22:31])
    f.cycle_idx = int(x[31]); f.previous_packed_key[:] = x[32:45]; f.previous_key[:] = x[45:53]; f.previous_address[:] = x[53:58]
    f.previous_timestamp = int(x[58]); f.this_cell_has_explicit_read_and_rollback_depth_zero = int(x[59])
    f.this_cell_base_value[:] = x[60:68]; f.this_cell_current_value[:] = x[68:76]; f.this_cell_current_depth = int(x[76])


def storage_witness(u, s, inst, first=0, fsm_in=None, with_output_tails=False):
    """the witness struct of the instance that starts at element `first` of the two queues (0: a start instance; otherwise a
    continuation with `fsm_in` = the flattened FSM state there), previous tails included"""
    o = inst["outer"]
    w = zkgl.StorageValidityWitness()
    w.start_flag, w.completion_flag, w.shard_id_to_process = int(first == 0), 1, int(o[1])
    w.unsorted_log_queue_state, w.intermediate_sorted_queue_state = _q4(o[2:11]), _q4(o[11:20])
    _storage_fsm(w.hidden_fsm_input, o[20:97] if fsm_in is None else fsm_in)
    ub, _ = sn.queue4_simulate([sn.encode(q) for q in u])
    sb, _ = sn.queue4_simulate([sn.encode_timestamped(q, t) for q, t in s])
    ua = (zkgl.LogQueryWitness * max(len(u) - first, 1))(*[_lq(q) for q in u[first:]])
    sa = (zkgl.TimestampedLogRecordWitness * max(len(s) - first, 1))()
    for rec, (q, t) in zip(sa, s[first:]):
        rec.record, rec.timestamp = _lq(q), int(t)
    ut, st = _tails_array(ub[first:]), _tails_array(sb[first:])
    w.unsorted_queue_witness, w.n_unsorted, w.intermediate_sorted_queue_witness, w.n_sorted = ua, len(u) - first, sa, len(s) - first
    w.unsorted_previous_tails, w.sorted_previous_tails = C.cast(ut, C.POINTER(C.c_uint64 * 4)), C.cast(st, C.POINTER(C.c_uint64 * 4))
    w._keep = (ua, sa, ut, st)
    if with_output_tails:
        _, tails = _queue4_tails([sn.encode(q) for q in inst["final_items"]])
        zkgl.set_output_tails(w, tails)
    return w


def _queue4_tails(encodings):
    tail, after = [0] * 4, []
    for e in encodings:
        tail = zko.queue_tail4_push20(tail, e)
        after.append(tail)
    return tail, after


def log_sorter_witness(u, s, inst, first=0, fsm_in=None, with_output_tails=False):
    o = inst["outer"]
    w = zkgl.LogSorterWitness()
    w.start_flag, w.completion_flag = int(first == 0), 1
    w.initial_log_queue_state, w.intermediate_sorted_queue_state = _q4(o[1:10]), _q4(o[10:19])
    f, x = w.hidden_fsm_input, (o[19:87] if fsm_in is None else fsm_in)
    f.lhs_accumulator[:] = x[0:2]; f.rhs_accumulator[:] = x[2:4]
    f.initial_unsorted_queue_state, f.intermediate_sorted_queue_state, f.final_result_queue_state = _q4(x[4:13]), _q4(x[13:22]), _q4(x[22:31])
    f.previous_key = int(x[31]); f.previous_item = _lq(x[32:68])
    ub, _ = sn.queue4_simulate([sn.encode(q) for q in u])
    sb, _ = sn.queue4_simulate([sn.encode(q) for q in s])
    ua = (zkgl.LogQueryWitness * max(len(u) - first, 1))(*[_lq(q) for q in u[first:]])
    sa = (zkgl.LogQueryWitness * max(len(s) - first, 1))(*[_lq(q) for q in s[first:]])
    ut, st = _tails_array(ub[first:]), _tails_array(sb[first:])
    w.initial_queue_witness, w.n_initial, w.intermediate_sorted_queue_witness, w.n_sorted = ua, len(u) - first, sa, len(s) - first
    w.initial_previous_tails, w.sorted_previous_tails = C.cast(ut, C.POINTER(C.c_uint64 * 4)), C.cast(st, C.POINTER(C.c_uint64 * 4))
    w._keep = (ua, sa, ut, st)
    if with_output_tails:
        _, tails = _queue4_tails([sn.encode(q) for q in inst["result_items"]])
        zkgl.set_output_tails(w, tails)
    return w


def storage_cases():
    out = []
    for seed, n, cells in ((11, LIMIT, 5), (12, LIMIT - 7, 3), (13, 1, 1), (14, 0, 1), (15, LIMIT - 1, 12)):
        u, s = sn.random_storage_witness(np.random.default_rng(seed), n, n_cells=cells)
        out.append((u, s, sn.instance(u, s, LIMIT)))
    return out


def log_sorter_cases():
    out = []
    for seed, n in ((21, 24), (22, 4), (23, 0), (24, 30)):
        u, s = ln.random_events(np.random.default_rng(seed), n, rollback_frac=0.3)
        u, s = u[:LIMIT], s[:LIMIT]
        if len(u) == LIMIT:   # a truncated batch is no longer a permutation with all its twins: keep it consistent instead
            u, s = ln.random_events(np.random.default_rng(seed), 20, rollback_frac=0.5)
        out.append((u, s, ln.instance(u, s, LIMIT)))
    return out


@pytest.mark.parametrize("with_output_tails", [False, True])
def test_storage_packer_walks_the_integer_state(with_output_tails):
    cases = storage_cases()
    B = len(cases)
    outer = np.zeros((97, B), dtype=np.uint64); loop = np.full((140, B * LIMIT), 7, dtype=np.uint64)
    for i, (u, s, inst) in enumerate(cases):
        assert inst["satisfiable"]
        w = storage_witness(u, s, inst, with_output_tails=with_output_tails)
        zkgl.pack_storage_witness(w, LIMIT, i, outer, loop)
        given = zkgl.storage_given_words(w)
    eo, el = sn.pack_streams([c[2] for c in cases], LIMIT)
    assert np.array_equal(outer, eo) and np.array_equal(loop[67:], el[67:])
    assert len(given) == (63 if with_output_tails else 59) and not set(given) & {2, 3, 4, 5}
    bad = [w for w in given if not np.array_equal(loop[w], el[w])]
    assert not bad, f"carried words {bad} differ from the native restatement"
    rest = [w for w in range(67) if w not in given]
    assert not loop[rest].any()


@pytest.mark.parametrize("with_output_tails", [False, True])
def test_log_sorter_packer_walks_the_integer_state(with_output_tails):
    cases = log_sorter_cases()
    B = len(cases)
    outer = np.zeros((87, B), dtype=np.uint64); loop = np.full((129, B * LIMIT), 7, dtype=np.uint64)
    for i, (u, s, inst) in enumerate(cases):
        assert inst["satisfiable"]
        w = log_sorter_witness(u, s, inst, with_output_tails=with_output_tails)
        zkgl.pack_log_sorter_witness(w, LIMIT, i, outer, loop)
        given = zkgl.log_sorter_given_words(w)
    eo, el = ln.pack_streams([c[2] for c in cases], LIMIT)
    assert np.array_equal(outer, eo) and np.array_equal(loop[57:], el[57:])
    assert len(given) == (53 if with_output_tails else 49)
    bad = [w for w in given if not np.array_equal(loop[w], el[w])]
    assert not bad, f"carried words {bad} differ from the native restatement"


def test_too_few_output_tails_are_rejected():
    u, s, inst = storage_cases()[0]
    w = storage_witness(u, s, inst, with_output_tails=True)
    assert w.n_output_tails > 1
    w.n_output_tails = 1
    outer = np.zeros((97, 1), dtype=np.uint64); loop = np.zeros((140, LIMIT), dtype=np.uint64)
    with pytest.raises(zkgl.ZkError):
        zkgl.pack_storage_witness(w, LIMIT, 0, outer, loop)


def _storage_cs(limit):
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_storage_validity()
    cs.sort_and_deduplicate_storage_access_entry_point(limit, True)
    cs.pad_and_shrink()
    return cs


def _log_sorter_cs(limit):
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_log_sorter()
    cs.sort_and_deduplicate_events_entry_point(limit)
    cs.pad_and_shrink()
    return cs


def _seed(zk, cs, outer, loop, given):
    cs.set_batch(outer.shape[1])
    cs.set_seed_given(given)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    cs.seed_carried_inputs(d_l)
    return d_o, d_l, d_l.to_numpy().reshape(loop.shape)


@pytest.mark.gpu
@pytest.mark.parametrize("circuit", ["storage", "log_sorter"])
@pytest.mark.parametrize("with_output_tails", [False, True])
def test_scan_seeding_equals_the_native_restatement(zk, circuit, with_output_tails):
    storage = circuit == "storage"
    cases = (storage_cases() if storage else log_sorter_cases()) * 17   # more than one wave tile of instances
    B = len(cases)
    n_outer, n_loop, carried = (97, 140, 67) if storage else (87, 129, 57)
    outer = np.zeros((n_outer, B), dtype=np.uint64); loop = np.zeros((n_loop, B * LIMIT), dtype=np.uint64)
    for i, (u, s, inst) in enumerate(cases):
        if storage:
            w = storage_witness(u, s, inst, with_output_tails=with_output_tails)
            zkgl.pack_storage_witness(w, LIMIT, i, outer, loop)
            given = zkgl.storage_given_words(w)
        else:
            w = log_sorter_witness(u, s, inst, with_output_tails=with_output_tails)
            zkgl.pack_log_sorter_witness(w, LIMIT, i, outer, loop)
            given = zkgl.log_sorter_given_words(w)
    cs = _storage_cs(LIMIT) if storage else _log_sorter_cs(LIMIT)
    eo, el = (sn if storage else ln).pack_streams([c[2] for c in cases], LIMIT)
    d_o, d_l, got = _seed(zk, cs, outer, loop, given)
    bad = sorted({int(w) for w in np.nonzero((got != el).any(axis=1))[0]})
    assert not bad, f"loop words {bad} differ from the native restatement"
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, c in enumerate(cases):
        assert cs.public_inputs(i) == c[2]["commitment"]
    cs.set_seed_given([])


@pytest.mark.gpu
@pytest.mark.parametrize("circuit", ["storage", "log_sorter"])
def test_continuation_instances_equal_the_cone_seeding(zk, circuit):
    """a continuation instance (start_flag = 0, FSM input = the state after `first` elements, the rest of both queues): the packer's
    walk + the scan kernels against the recorded cone on the same raw stream"""
    storage = circuit == "storage"
    limit = 16
    n_outer, n_loop, carried = (97, 140, 67) if storage else (87, 129, 57)
    cs = _storage_cs(limit) if storage else _log_sorter_cs(limit)
    packs = []
    for seed in range(6):
        first = 9 + seed % 3
        if storage:
            u, s = sn.random_storage_witness(np.random.default_rng(300 + seed), first + limit - seed, n_cells=4)
        else:
            u, s = ln.random_events(np.random.default_rng(400 + seed), 9 + seed, rollback_frac=0.4)
        packs.append((u, s, first))
    # the FSM state after `first` elements: run the long instance on the device once (cone seeding), read cycle `first`'s carried words
    B = len(packs)
    long_limit = limit + 12
    cs_long = _storage_cs(long_limit) if storage else _log_sorter_cs(long_limit)
    o_long = np.zeros((n_outer, B), dtype=np.uint64); l_long = np.zeros((n_loop, B * long_limit), dtype=np.uint64)
    insts = []
    for i, (u, s, first) in enumerate(packs):
        assert len(u) <= long_limit
        inst = (sn if storage else ln).instance(u, s, long_limit)
        insts.append(inst)
        w = (storage_witness if storage else log_sorter_witness)(u, s, inst)
        (zkgl.pack_storage_witness if storage else zkgl.pack_log_sorter_witness)(w, long_limit, i, o_long, l_long)
        given = (zkgl.storage_given_words if storage else zkgl.log_sorter_given_words)(w)
    _, _, full = _seed(zk, cs_long, o_long, l_long, given)
    cs_long.set_seed_given([])
    el = (sn if storage else ln).pack_streams(insts, long_limit)[1]
    assert np.array_equal(full, el)
    # continuation witnesses from cycle `first` on
    outer = np.zeros((n_outer, B), dtype=np.uint64); loop = np.zeros((n_loop, B * limit), dtype=np.uint64)
    for i, (u, s, first) in enumerate(packs):
        state = [int(x) for x in full[:carried, i * long_limit + first]]
        if storage:
            o = insts[i]["outer"]
            # StorageDeduplicatorFSMInputOutput from the carried words: lhs, rhs, the three queue states (head, full tail, length), ...
            fsm = state[2:6] + state[7:11] + list(o[6:10]) + [state[11]] + state[12:16] + list(o[15:19]) + [state[16]] + [0] * 4 + state[17:21] + [state[21]] + \
                [state[6]] + state[22:35] + state[35:43] + state[43:48] + [state[48], state[49]] + state[50:58] + state[58:66] + [state[66]]
            w = storage_witness(u, s, insts[i], first=first, fsm_in=fsm)
            zkgl.pack_storage_witness(w, limit, i, outer, loop)
            given = zkgl.storage_given_words(w)
        else:
            o = insts[i]["outer"]
            fsm = state[1:5] + state[5:9] + list(o[5:9]) + [state[9]] + state[10:14] + list(o[14:18]) + [state[14]] + [0] * 4 + state[15:19] + [state[19]] + \
                [state[20]] + state[21:57]
            w = log_sorter_witness(u, s, insts[i], first=first, fsm_in=fsm)
            zkgl.pack_log_sorter_witness(w, limit, i, outer, loop)
            given = zkgl.log_sorter_given_words(w)
    _, _, got = _seed(zk, cs, outer, loop, given)
    raw = loop.copy(); raw[:carried] = 0
    _, d_c, cone = _seed(zk, cs, outer, raw, [])
    bad = sorted({int(w) for w in np.nonzero((got != cone).any(axis=1))[0]})
    assert not bad, f"loop words {bad}: packer walk + scans differ from the cone seeding"
    # and the continuation carries on exactly where the long instance was
    for i, (u, s, first) in enumerate(packs):
        n = min(limit, long_limit - first)
        assert np.array_equal(got[2 if storage else 1:carried, i * limit:i * limit + n], full[2 if storage else 1:carried, i * long_limit + first:i * long_limit + first + n])
    ok, f = cs.resolve_and_check()
    assert ok, f
