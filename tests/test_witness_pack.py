"""Product-side witness packers (include/zkgl_witness.h): the streams of ram_permutation built THROUGH THE C ABI
from the reference's witness struct (and from its bincode bytes) must equal what the oracle's packer builds from the same witness;
on the GPU the C-ABI-built streams are seeded, resolved and satisfied, with the oracle's commitment.  The oracle packer
(oracle/ram_native.py) is only the comparison here."""
import ctypes as C
import struct

import numpy as np
import pytest

import zkgl

C_u32x8 = C.c_uint32 * 8
from oracle import ram_native as rn

LIMIT = 24


def _qstate(head, tail, length):
    q = zkgl.FullQueueStateWitness()
    q.head[:] = [int(x) for x in head]
    q.tail[:] = [int(x) for x in tail]
    q.length = int(length)
    return q


def _fsm(f):
    w = zkgl.RamFsmWitness()
    w.lhs_accumulator[:] = f["lhs"]; w.rhs_accumulator[:] = f["rhs"]
    w.current_unsorted_queue_state = _qstate(f["unsorted"][0:12], f["unsorted"][12:24], f["unsorted"][24])
    w.current_sorted_queue_state = _qstate(f["sorted"][0:12], f["sorted"][12:24], f["sorted"][24])
    w.previous_sorting_key[:] = f["prev_sorting_key"]; w.previous_full_key[:] = f["prev_full_key"]
    w.previous_value[:] = f["prev_value"]; w.previous_is_ptr = int(f["prev_is_ptr"]); w.num_nondeterministic_writes = int(f["nondet"])
    return w


def _queries(items):
    arr = (zkgl.MemoryQueryWitness * max(len(items), 1))()
    for a, it in zip(arr, items):
        a.timestamp, a.memory_page, a.index, a.rw_flag, a.is_ptr = it[0], it[1], it[2], it[3], it[4]
        a.value[:] = it[5:13]
    return arr


def witness_struct(inst, unsorted, sorted_, nondet, fsm_in):
    w = zkgl.RamPermutationWitness()
    w.start_flag, w.completion_flag = 1, int(inst["completed"])
    ou, os_ = inst["obs_unsorted"], inst["obs_sorted"]
    w.unsorted_queue_initial_state = _qstate(ou[0:12], ou[12:24], ou[24])
    w.sorted_queue_initial_state = _qstate(os_[0:12], os_[12:24], os_[24])
    w.non_deterministic_bootloader_memory_snapshot_length = nondet
    w.hidden_fsm_input, w.hidden_fsm_output = _fsm(fsm_in), _fsm(inst["fsm_out"])
    ua, sa = _queries(unsorted), _queries(sorted_)
    w.unsorted_queue_witness, w.n_unsorted, w.sorted_queue_witness, w.n_sorted = ua, len(unsorted), sa, len(sorted_)
    w._keep = (ua, sa)
    return w


# ---- a bincode 1.x writer for the same struct (test side): the decoder under test is the C one
def _b_u256(limbs):
    v = sum(int(x) << (32 * i) for i, x in enumerate(limbs))
    s = ("0x%x" % v).encode()
    return struct.pack("<Q", len(s)) + s


def _b_qstate(q):
    return b"".join(struct.pack("<Q", int(x)) for x in list(q.head) + list(q.tail)) + struct.pack("<I", q.length)


def _b_fsm(f):
    return (b"".join(struct.pack("<Q", int(x)) for x in list(f.lhs_accumulator) + list(f.rhs_accumulator)) + _b_qstate(f.current_unsorted_queue_state) +
            _b_qstate(f.current_sorted_queue_state) + b"".join(struct.pack("<I", int(x)) for x in list(f.previous_sorting_key) + list(f.previous_full_key)) +
            _b_u256(f.previous_value) + struct.pack("<B", f.previous_is_ptr) + struct.pack("<I", f.num_nondeterministic_writes))


def bincode_bytes(w, tails_u, tails_s):
    out = struct.pack("<BB", w.start_flag, w.completion_flag) + _b_qstate(w.unsorted_queue_initial_state) + _b_qstate(w.sorted_queue_initial_state)
    out += struct.pack("<I", w.non_deterministic_bootloader_memory_snapshot_length) + _b_fsm(w.hidden_fsm_input) + _b_fsm(w.hidden_fsm_output)
    for q, n, tails in ((w.unsorted_queue_witness, w.n_unsorted, tails_u), (w.sorted_queue_witness, w.n_sorted, tails_s)):
        out += struct.pack("<Q", n)
        for i in range(n):
            m = q[i]
            out += struct.pack("<IIIBB", m.timestamp, m.memory_page, m.index, m.rw_flag, m.is_ptr) + _b_u256(m.value)
            out += b"".join(struct.pack("<Q", int(x)) for x in tails[i])
    return out


def _case(seed, n_items):
    rng = np.random.default_rng(seed)
    u, s, nd = rn.random_ram_witness(rng, n_items, n_cells=6)
    inst = rn.instance(u, s, LIMIT, nd)
    return u, s, nd, inst


def _expected(insts):
    outer, loop = rn.pack_streams(insts, LIMIT)
    loop = loop.copy()
    loop[0:46] = 0          # the packer leaves the carried words to the device seeding
    return outer, loop


def test_ram_packer_equals_the_oracle_packer():
    cases = [_case(100 + i, n) for i, n in enumerate((LIMIT, LIMIT - 5, 1))]
    B = len(cases)
    outer = np.zeros((zkgl.RAM_OUTER_WORDS, B), dtype=np.uint64)
    loop = np.full((zkgl.RAM_LOOP_WORDS, B * LIMIT), 0xDEAD, dtype=np.uint64)
    for i, (u, s, nd, inst) in enumerate(cases):
        zkgl.pack_ram_witness(witness_struct(inst, u, s, nd, rn.empty_fsm()), LIMIT, i, outer, loop)
    eo, el = _expected([c[3] for c in cases])
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)


def test_ram_bincode_round_trip_and_errors():
    u, s, nd, inst = _case(7, LIMIT - 2)
    w = witness_struct(inst, u, s, nd, rn.empty_fsm())
    ub, sb = inst["heads"]
    data = bincode_bytes(w, ub, sb)
    d, used = zkgl.decode_ram_witness_bincode(data + b"tail", LIMIT)
    assert used == len(data)
    outer = np.zeros((zkgl.RAM_OUTER_WORDS, 1), dtype=np.uint64)
    loop = np.zeros((zkgl.RAM_LOOP_WORDS, LIMIT), dtype=np.uint64)
    zkgl.pack_ram_witness(d, LIMIT, 0, outer, loop)
    eo, el = _expected([inst])
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)
    with pytest.raises(zkgl.ZkError):
        zkgl.decode_ram_witness_bincode(data[:-9], LIMIT)          # truncated
    with pytest.raises(zkgl.ZkError):
        zkgl.decode_ram_witness_bincode(data, LIMIT - 10)          # more elements than the caller's buffers
    with pytest.raises(zkgl.ZkError):
        zkgl.pack_ram_witness(d, LIMIT - 10, 0, np.zeros((121, 1), dtype=np.uint64), np.zeros((72, LIMIT - 10), dtype=np.uint64))


def _packed_with_tails(cases):
    B = len(cases)
    outer = np.zeros((zkgl.RAM_OUTER_WORDS, B), dtype=np.uint64)
    loop = np.zeros((zkgl.RAM_LOOP_WORDS, B * LIMIT), dtype=np.uint64)
    for i, (u, s, nd, inst) in enumerate(cases):
        ub, sb = inst["heads"]
        w, _ = zkgl.decode_ram_witness_bincode(bincode_bytes(witness_struct(inst, u, s, nd, rn.empty_fsm()), ub, sb), LIMIT, keep_tails=True)
        zkgl.pack_ram_witness(w, LIMIT, i, outer, loop)
    return outer, loop


def test_ram_packer_writes_the_queue_heads_from_the_previous_tails():
    """the decoder that keeps the witness's previous tails (CircuitQueueRawWitness elements are (item, previous_tail),
    reference src/base_structures/vm_state/mod.rs FullStateCircuitQueueRawWitness) lets the packer write the head of both queues in
    every cycle: those 24 words equal the native restatement's, the other carried words stay with the device"""
    cases = [_case(500 + i, n) for i, n in enumerate((LIMIT, LIMIT - 9, 1, 0))]
    outer, loop = _packed_with_tails(cases)
    full_o, full_l = rn.pack_streams([c[3] for c in cases], LIMIT)
    heads = zkgl.ram_head_words()
    assert heads == list(range(1, 13)) + list(range(14, 26))
    assert np.array_equal(loop[heads], full_l[heads])
    rest = [w for w in range(46) if w not in heads]
    assert not loop[rest].any()
    assert np.array_equal(loop[46:], full_l[46:]) and np.array_equal(outer, full_o)


@pytest.mark.gpu
def test_ram_scan_seeding_from_given_heads_equals_the_cone_seeding(zk):
    """kernels_queue_seed.hpp: with the heads given, the remaining carried words (accumulators, lengths, previous keys) are scans over
    the cycles — bit-equal to the recorded-cone seeding and to the native restatement, instances ending early / empty included"""
    from helpers import ram_cs
    cases = [_case(600 + i, n) for i, n in enumerate((LIMIT, LIMIT - 7, LIMIT - 1, 3, 0, 1))]
    outer, loop = _packed_with_tails(cases)
    full_o, full_l = rn.pack_streams([c[3] for c in cases], LIMIT)
    cs = ram_cs(LIMIT)
    cs.set_batch(len(cases))
    d_o = zk.DeviceBuffer.from_numpy(outer)
    cs.bind_inputs(False, d_o, outer.shape[0])
    # the cone (nothing given): from a stream without the heads
    raw = loop.copy(); raw[0:46] = 0
    d_cone = zk.DeviceBuffer.from_numpy(raw)
    cs.bind_inputs(True, d_cone, loop.shape[0])
    cs.seed_carried_inputs(d_cone)
    assert np.array_equal(d_cone.to_numpy().reshape(loop.shape), full_l)
    # the scan kernel
    cs.set_seed_given(zkgl.ram_head_words())
    d_l = zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(True, d_l, loop.shape[0])
    cs.seed_carried_inputs(d_l)
    got = d_l.to_numpy().reshape(loop.shape)
    bad = [w for w in range(46) if not np.array_equal(got[w], full_l[w])]
    assert not bad, f"carried words {bad} differ from the native restatement"
    assert np.array_equal(got, full_l)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, c in enumerate(cases):
        assert cs.public_inputs(i) == c[3]["commitment"]
    cs.set_seed_given([])


@pytest.mark.gpu
def test_ram_streams_built_through_the_c_abi_run_on_the_gpu(zk):
    from helpers import ram_cs
    cases = [_case(300 + i, n) for i, n in enumerate((LIMIT, LIMIT - 7, LIMIT - 1, 3))]
    B = len(cases)
    outer = np.zeros((zkgl.RAM_OUTER_WORDS, B), dtype=np.uint64)
    loop = np.zeros((zkgl.RAM_LOOP_WORDS, B * LIMIT), dtype=np.uint64)
    for i, (u, s, nd, inst) in enumerate(cases):
        ub, sb = inst["heads"]
        w, _ = zkgl.decode_ram_witness_bincode(bincode_bytes(witness_struct(inst, u, s, nd, rn.empty_fsm()), ub, sb), LIMIT)
        zkgl.pack_ram_witness(w, LIMIT, i, outer, loop)
    cs = ram_cs(LIMIT)
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    cs.seed_carried_inputs(d_l)
    full_o, full_l = rn.pack_streams([c[3] for c in cases], LIMIT)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), full_l), "device-seeded carried words differ from the native restatement"
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, c in enumerate(cases):
        assert cs.public_inputs(i) == c[3]["commitment"]


@pytest.mark.gpu
def test_hook_compare_witness_on_the_device(zk):
    """zk_cs_hook_compare_witness: the circuit's hidden_fsm_output against the expectation (here: the native restatement's), equal for
    the true witness, first difference (instance, position) for a tampered expectation"""
    from helpers import ram_cs
    cases = [_case(500 + i, n) for i, n in enumerate((LIMIT, LIMIT - 3, 5))]
    B = len(cases)
    outer, loop = rn.pack_streams([c[3] for c in cases], LIMIT)
    cs = ram_cs(LIMIT)
    cs.set_batch(B)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    cs.resolve()
    hv = cs.hook_vars("hidden_fsm_output")
    expected = np.array([rn.flatten_fsm(c[3]["fsm_out"]) for c in cases], dtype=np.uint64).T.copy()   # [69, B]
    assert expected.shape == (len(hv), B)
    ok, where = cs.hook_compare_witness(hv, zk.DeviceBuffer.from_numpy(expected))
    assert ok, where
    bad = expected.copy(); bad[40, 2] ^= 1; bad[50, 2] ^= 1
    ok, where = cs.hook_compare_witness(hv, zk.DeviceBuffer.from_numpy(bad))
    assert not ok and where == (2, 40)


@pytest.mark.gpu
def test_non_canonical_input_word_is_reported(zk):
    """an input stream word >= p is reported by check_if_satisfied (kind ZK_FAILURE_NONCANONICAL_INPUT), whatever the gates make of it"""
    from helpers import ram_cs
    u, s, nd, inst = _case(900, LIMIT)
    outer, loop = rn.pack_streams([inst, inst], LIMIT)
    loop = loop.copy()
    loop[50, LIMIT + 3] = np.uint64(0xFFFFFFFF00000001)   # == p: the same field element as 0, but not canonical
    cs = ram_cs(LIMIT)
    cs.set_batch(2)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    ok, f = cs.resolve_and_check()
    assert not ok
    cs.resolve()
    ok, f = cs.check_if_satisfied()
    assert not ok and f.scope == 1 and f.instance == 1
    # (the first failure reported for the lane may be a gate that chokes on the word; when it is the input check itself it carries its
    # own kind, distinct from the stream links' 0x400)
    assert zk.FAILURE_NONCANONICAL_INPUT != zk.FAILURE_STREAM_LINK and (f.kind < 0x100 or f.kind == zk.FAILURE_NONCANONICAL_INPUT)


# ---------------------------------------------------------------- storage_validity / log_sorter packers
def _lq(words):
    q = zkgl.LogQueryWitness()
    q.address[:] = words[0:5]; q.key[:] = words[5:13]; q.read_value[:] = words[13:21]; q.written_value[:] = words[21:29]
    q.aux_byte, q.rw_flag, q.rollback, q.is_service, q.shard_id, q.tx_number_in_block, q.timestamp = [int(x) for x in words[29:36]]
    return q


def _q4(words):
    q = zkgl.QueueStateWitness()
    q.head[:] = [int(x) for x in words[0:4]]; q.tail[:] = [int(x) for x in words[4:8]]; q.length = int(words[8])
    return q


def test_storage_packer_equals_the_oracle_packer():
    from oracle import storage_native as sn
    limit = 20
    cases = []
    for seed, n in ((11, 17), (12, 5)):
        u, s = sn.random_storage_witness(np.random.default_rng(seed), n, n_cells=3)
        cases.append((u, s, sn.instance(u, s, limit)))
    B = len(cases)
    outer = np.zeros((97, B), dtype=np.uint64); loop = np.full((140, B * limit), 7, dtype=np.uint64)
    for i, (u, s, inst) in enumerate(cases):
        o = inst["outer"]
        w = zkgl.StorageValidityWitness()
        w.start_flag, w.completion_flag, w.shard_id_to_process = int(o[0]), int(inst["completed"]), int(o[1])
        w.unsorted_log_queue_state, w.intermediate_sorted_queue_state = _q4(o[2:11]), _q4(o[11:20])
        f, x = w.hidden_fsm_input, o[20:97]
        f.lhs_accumulator[:] = x[0:2]; f.rhs_accumulator[:] = x[2:4]
        f.current_unsorted_queue_state, f.current_intermediate_sorted_queue_state, f.current_final_sorted_queue_state = _q4(x[4:13]), _q4(x[13:22]), _q4(x[22:31])
        f.cycle_idx = int(x[31]); f.previous_packed_key[:] = x[32:45]; f.previous_key[:] = x[45:53]; f.previous_address[:] = x[53:58]
        f.previous_timestamp = int(x[58]); f.this_cell_has_explicit_read_and_rollback_depth_zero = int(x[59])
        f.this_cell_base_value[:] = x[60:68]; f.this_cell_current_value[:] = x[68:76]; f.this_cell_current_depth = int(x[76])
        ua = (zkgl.LogQueryWitness * len(u))(*[_lq(q) for q in u])
        sa = (zkgl.TimestampedLogRecordWitness * len(s))()
        for rec, (q, t) in zip(sa, s):
            rec.record, rec.timestamp = _lq(q), int(t)
        w.unsorted_queue_witness, w.n_unsorted, w.intermediate_sorted_queue_witness, w.n_sorted = ua, len(u), sa, len(s)
        zkgl.pack_storage_witness(w, limit, i, outer, loop)
    eo, el = sn.pack_streams([c[2] for c in cases], limit)
    el = el.copy(); el[0:67] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)


def test_log_sorter_packer_equals_the_oracle_packer():
    from oracle import log_sorter_native as ln
    limit = 20
    cases = []
    for seed, n in ((21, 12), (22, 4)):
        u, s = ln.random_events(np.random.default_rng(seed), n, rollback_frac=0.3)
        u, s = u[:limit], s[:limit]
        cases.append((u, s, ln.instance(u, s, limit)))
    B = len(cases)
    outer = np.zeros((87, B), dtype=np.uint64); loop = np.full((129, B * limit), 7, dtype=np.uint64)
    for i, (u, s, inst) in enumerate(cases):
        o = inst["outer"]
        w = zkgl.LogSorterWitness()
        w.start_flag, w.completion_flag = int(o[0]), int(inst["completed"])
        w.initial_log_queue_state, w.intermediate_sorted_queue_state = _q4(o[1:10]), _q4(o[10:19])
        f, x = w.hidden_fsm_input, o[19:87]
        f.lhs_accumulator[:] = x[0:2]; f.rhs_accumulator[:] = x[2:4]
        f.initial_unsorted_queue_state, f.intermediate_sorted_queue_state, f.final_result_queue_state = _q4(x[4:13]), _q4(x[13:22]), _q4(x[22:31])
        f.previous_key = int(x[31]); f.previous_item = _lq(x[32:68])
        ua = (zkgl.LogQueryWitness * len(u))(*[_lq(q) for q in u])
        sa = (zkgl.LogQueryWitness * len(s))(*[_lq(q) for q in s])
        w.initial_queue_witness, w.n_initial, w.intermediate_sorted_queue_witness, w.n_sorted = ua, len(u), sa, len(s)
        zkgl.pack_log_sorter_witness(w, limit, i, outer, loop)
    eo, el = ln.pack_streams([c[2] for c in cases], limit)
    el = el.copy(); el[0:57] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)


def test_eip4844_packer_equals_the_oracle_rows():
    from oracle import eip4844_native as en
    rng = np.random.default_rng(44)
    cases = []
    for n_chunks in (27, 27):
        blob = bytes(rng.integers(0, 256, size=31 * n_chunks, dtype=np.uint8))
        vh = b"\x01" + bytes(rng.integers(0, 256, size=31, dtype=np.uint8))
        cases.append((blob, vh, en.instance(blob, vh, n_chunks)))
    it, lw = zkgl.eip4844_stream_shape(27)
    assert (it, lw) == (len(cases[0][2]["rows"]), len(cases[0][2]["rows"][0]))
    B = len(cases)
    outer = np.zeros((64, B), dtype=np.uint64); loop = np.full((lw, B * it), 9, dtype=np.uint64)
    for i, (blob, vh, inst) in enumerate(cases):
        zkgl.pack_eip4844_witness(blob, vh, inst["linear_hash"], i, outer, loop)
    eo = np.array([c[2]["outer"] for c in cases], dtype=np.uint64).T
    el = np.array([r for c in cases for r in c[2]["rows"]], dtype=np.uint64).T.copy()
    full = el.copy()
    el[0:217] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)
    # zk_pack_eip4844_witness_full: sponge states and opening limbs (BLS12-381 scalar field Horner) walked on the host — all words
    for i, (blob, vh, inst) in enumerate(cases):
        zkgl.pack_eip4844_witness(blob, vh, inst["linear_hash"], i, outer, loop, full=True)
    assert np.array_equal(outer, eo) and np.array_equal(loop, full), np.argwhere(loop != full)[:8]


@pytest.mark.gpu
def test_eip4844_with_the_carried_words_from_the_host_needs_no_device_seeding(zk):
    """zk_pack_eip4844_witness_full + zk_eip4844_given_words: the product path with no seeding kernel; commitments equal the restatement's"""
    from oracle import eip4844_native as en
    n_chunks = 27
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_eip_4844(); cs.eip_4844_entry_point(n_chunks); cs.pad_and_shrink()
    rng = np.random.default_rng(4845)
    cases = []
    for k in range(5):
        blob = bytes(rng.integers(0, 256, size=31 * n_chunks, dtype=np.uint8)) if k else b"\xff" * (31 * n_chunks)
        vh = b"\x01" + bytes(rng.integers(0, 256, size=31, dtype=np.uint8))
        cases.append((blob, vh, en.instance(blob, vh, n_chunks)))
    it, lw = zkgl.eip4844_stream_shape(n_chunks)
    B = len(cases)
    outer = np.zeros((64, B), dtype=np.uint64); loop = np.zeros((lw, B * it), dtype=np.uint64)
    for i, (blob, vh, inst) in enumerate(cases):
        zkgl.pack_eip4844_witness(blob, vh, inst["linear_hash"], i, outer, loop, full=True)
    cs.set_batch(B)
    cs.set_seed_given(list(range(217)))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, (_, _, inst) in enumerate(cases):
        assert cs.public_inputs(i) == inst["public_input"]


def _q12(words):
    q = zkgl.FullQueueStateWitness()
    q.head[:] = [int(x) for x in words[0:12]]; q.tail[:] = [int(x) for x in words[12:24]]; q.length = int(words[24])
    return q


def _sha_fsm(f, x):
    f.read_precompile_call, f.read_words_for_round, f.completed = int(x[0]), int(x[1]), int(x[2])
    f.sha256_inner_state[:] = x[3:11]; f.timestamp_to_use_for_read, f.timestamp_to_use_for_write = int(x[11]), int(x[12])
    f.input_page, f.input_offset, f.output_page, f.output_offset, f.num_rounds = [int(v) for v in x[13:18]]
    f.log_queue_state, f.memory_queue_state = _q4(x[18:27]), _q12(x[27:52])


def test_sha256_packer_walks_the_fsm_schedule():
    """requests and read values are placed at the cycles that consume them (the reference pops them lazily): several requests in one
    instance, and a continuation instance that starts in the middle of a request"""
    from oracle import sha256_native as shn
    rng = np.random.default_rng(256)
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (10, 100, 64 * 3 - 9, 0)]
    reqs = [shn.request(m, 1 + 2 * i, 10 + i, 3 * i, 9000 + i, i) for i, m in enumerate(msgs)]
    limit = 5
    first = shn.instance(reqs, limit)
    second = shn.instance(first["rest"][0], limit, start_flag=False, fsm_in=first["fsm_out"], obs_req=first["obs_req"], obs_mem=first["obs_mem"],
                          pending=first["rest"][1])
    insts = [first, second]
    B = 2
    outer = np.zeros((87, B), dtype=np.uint64); loop = np.full((112, B * limit), 5, dtype=np.uint64)
    all_reads = [v for r in reqs for v in r["reads"]]
    consumed_reads = 0
    consumed_reqs = 0
    for i, inst in enumerate(insts):
        o = inst["outer"]
        w = zkgl.Sha256RoundFunctionWitness()
        w.start_flag = int(o[0])
        w.initial_log_queue_state, w.initial_memory_queue_state = _q4(o[1:10]), _q12(o[10:35])
        _sha_fsm(w.hidden_fsm_input, o[35:87])
        # what the witness generator would hand over for this instance: the requests / reads from here on
        rq = reqs[consumed_reqs:]
        rd = all_reads[consumed_reads:]
        qa = (zkgl.LogQueryWitness * max(len(rq), 1))(*[_lq(r["query"]) for r in rq])
        ra = ((C_u32x8 := (zkgl.C.c_uint32 * 8)) * max(len(rd), 1))()
        for dst, v in zip(ra, rd):
            dst[:] = [(v >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
        w.requests_queue_witness, w.n_requests, w.memory_reads_witness, w.n_reads = qa, len(rq), ra, len(rd)
        zkgl.pack_sha256_witness(w, limit, i, outer, loop)
        rows = np.array(inst["rows"], dtype=np.uint64)
        consumed_reqs += int(sum(1 for r in rows if r[60:96].any()))
        consumed_reads += int(sum((1 if r[96:104].any() else 0) + (1 if r[104:112].any() else 0) for r in rows))
    eo = np.array([x["outer"] for x in insts], dtype=np.uint64).T
    el = np.array([r for x in insts for r in x["rows"]], dtype=np.uint64).T.copy()
    el[0:60] = 0
    assert np.array_equal(outer, eo)
    assert np.array_equal(loop, el)


def _sha256_packed_with_tails():
    """a start instance that stops in the middle of a call and its continuation, packed with the queue states the reference's
    witnesses hold: previous tails of the requests, the memory queue's tail after every push (= the RAM permutation's witness)"""
    from oracle import sha256_native as shn, zko
    from oracle.storage_native import encode
    rng = np.random.default_rng(257)
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (10, 100, 64 * 3 - 9, 0)]
    reqs = [shn.request(m, 1 + 2 * i, 10 + i, 3 * i, 9000 + i, i) for i, m in enumerate(msgs)]
    limit = 5
    first = shn.instance(reqs, limit)
    second = shn.instance(first["rest"][0], limit, start_flag=False, fsm_in=first["fsm_out"], obs_req=first["obs_req"], obs_mem=first["obs_mem"],
                          pending=first["rest"][1])
    insts = [first, second]
    outer = np.zeros((87, 2), dtype=np.uint64); loop = np.full((112, 2 * limit), 5, dtype=np.uint64)
    all_reads = [v for r in reqs for v in r["reads"]]
    consumed_reads = consumed_reqs = 0
    given = None
    for i, inst in enumerate(insts):
        o = inst["outer"]
        w = zkgl.Sha256RoundFunctionWitness()
        w.start_flag = int(o[0])
        w.initial_log_queue_state, w.initial_memory_queue_state = _q4(o[1:10]), _q12(o[10:35])
        _sha_fsm(w.hidden_fsm_input, o[35:87])
        w.hidden_fsm_output.log_queue_state = _q4(inst["fsm_out"]["req"])
        rows = np.array(inst["rows"], dtype=np.uint64)
        n_req = int(sum(1 for r in rows if r[60:96].any()))
        n_rd = int(sum((1 if r[96:104].any() else 0) + (1 if r[104:112].any() else 0) for r in rows))
        rq = reqs[consumed_reqs:consumed_reqs + n_req]                  # what this instance pops
        rd = all_reads[consumed_reads:]
        qa = (zkgl.LogQueryWitness * max(len(rq), 1))(*[_lq(r["query"]) for r in rq])
        ra = ((zkgl.C.c_uint32 * 8) * max(len(rd), 1))()
        for dst, v in zip(ra, rd):
            dst[:] = [(v >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
        w.requests_queue_witness, w.n_requests, w.memory_reads_witness, w.n_reads = qa, len(rq), ra, len(rd)
        start_req = o[1:10] if o[0] else o[35 + 18:35 + 27]
        start_mem = o[10:35] if o[0] else o[35 + 27:35 + 52]
        head, prev = [int(v) for v in start_req[0:4]], []
        for r in rq:
            prev.append(head)
            head = zko.queue_tail4_push20(head, encode(r["query"]))
        mt, mtails = [int(v) for v in start_mem[12:24]], []
        for q in inst["pushed"]:
            mt = zko.queue_full_push(mt, zko.memory_query_encode(q))
            mtails.append(mt)
        given = zkgl.pack_sha256_witness_tails(w, limit, i, outer, loop, np.array(prev or [[0] * 4], dtype=np.uint64), np.array(mtails, dtype=np.uint64).reshape(-1, 12))
        consumed_reqs += n_req; consumed_reads += n_rd
    return outer, loop, insts, limit, given


def test_sha256_packer_with_the_witness_queue_states_writes_every_carried_word():
    """zk_pack_sha256_witness_tails: flags, call parameters and the SHA-256 inner state walked natively, queue states from the witness —
    the stream equals the native restatement's in all 112 words of every cycle (the device pass it replaces is a chain of dependent
    Poseidon2 permutations: 38 ms against a 10 ms step at full size)"""
    outer, loop, insts, limit, given = _sha256_packed_with_tails()
    assert given == list(range(60))
    eo = np.array([x["outer"] for x in insts], dtype=np.uint64).T
    el = np.array([r for x in insts for r in x["rows"]], dtype=np.uint64).T.copy()
    assert np.array_equal(outer, eo)
    assert np.array_equal(loop, el), np.argwhere(loop != el)[:8]


@pytest.mark.gpu
def test_sha256_fsm_with_the_witness_queue_states_needs_no_device_seeding(zk):
    outer, loop, insts, limit, given = _sha256_packed_with_tails()
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_sha256()
    cs.sha256_round_function_entry_point(limit)
    cs.pad_and_shrink()
    cs.set_batch(len(insts))
    cs.set_seed_given(given)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    cs.seed_carried_inputs(d_l)                      # every carried word is declared given: no kernel runs
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]


def test_keccak_packer_walks_the_fsm_schedule():
    """the keccak precompile: six conditional unaligned reads per cycle through the 192-byte buffer — requests and read values placed
    at their cycles by the schedule walk; several requests (aligned, unaligned, empty, block-sized) and a continuation instance"""
    outer, loop, insts = _keccak_packed(False)
    eo = np.array([x["outer"] for x in insts], dtype=np.uint64).T
    el = np.array([r for x in insts for r in x["rows"]], dtype=np.uint64).T.copy()
    el[0:423] = 0
    assert np.array_equal(outer, eo)
    assert np.array_equal(loop, el)


def test_keccak_packer_with_the_witness_queue_states_writes_every_carried_word():
    """zk_pack_keccak_witness_tails: flags, parameters, the ByteBuffer and the sponge state (one native Keccak-f per cycle) walked on the
    host, queue states from the witness — all 507 words of every cycle equal the native restatement's"""
    outer, loop, insts = _keccak_packed(True)
    eo = np.array([x["outer"] for x in insts], dtype=np.uint64).T
    el = np.array([r for x in insts for r in x["rows"]], dtype=np.uint64).T.copy()
    assert np.array_equal(outer, eo)
    assert np.array_equal(loop, el), np.argwhere(loop != el)[:8]


@pytest.mark.gpu
def test_keccak_fsm_with_the_witness_queue_states_needs_no_device_seeding(zk):
    from test_keccak_fsm_host import fsm_cs
    outer, loop, insts = _keccak_packed(True)
    limit = loop.shape[1] // len(insts)
    cs = fsm_cs(limit)
    cs.set_batch(len(insts))
    cs.set_seed_given(list(range(423)))
    try:
        d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
        cs.bind_inputs(False, d_o, outer.shape[0])
        cs.bind_inputs(True, d_l, loop.shape[0])
        cs.seed_carried_inputs(d_l)                      # every carried word is declared given: no kernel runs
        assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
        ok, f = cs.resolve_and_check()
        assert ok, f
        for i, inst in enumerate(insts):
            assert cs.public_inputs(i) == inst["public_input"]
    finally:
        cs.set_seed_given([])


def _keccak_packed(tails):
    from oracle import keccak_native as kn, zko
    from oracle.storage_native import encode
    rng = np.random.default_rng(1600)
    specs = [(50, 0), (135, 31), (0, 0), (136, 0), (200, 7)]
    reqs = [kn.request(bytes(rng.integers(0, 256, size=n, dtype=np.uint8)), 1 + 2 * i, 10 + i, 64 * i + mis, 9000 + i, i) for i, (n, mis) in enumerate(specs)]
    limit = 4
    first = kn.instance(reqs, limit)
    second = kn.instance(first["rest"][0], limit, start_flag=False, fsm_in=first["fsm_out"], obs_req=first["obs_req"], obs_mem=first["obs_mem"],
                         pending=first["rest"][1])
    third = kn.instance(second["rest"][0], limit, start_flag=False, fsm_in=second["fsm_out"], obs_req=second["obs_req"], obs_mem=second["obs_mem"],
                        pending=second["rest"][1])
    insts = [first, second, third]
    B = len(insts)
    outer = np.zeros((474, B), dtype=np.uint64); loop = np.full((507, B * limit), 5, dtype=np.uint64)
    all_reads = [v for r in reqs for v in r["reads"]]
    used_reqs = used_reads = 0
    for i, inst in enumerate(insts):
        o = inst["outer"]
        w = zkgl.KeccakRoundFunctionWitness()
        w.start_flag = int(o[0])
        w.initial_log_queue_state, w.initial_memory_queue_state = _q4(o[1:10]), _q12(o[10:35])
        f, x = w.hidden_fsm_input, o[35:474]
        f.read_precompile_call, f.read_unaligned_words_for_round, f.padding_round, f.completed = [int(v) for v in x[0:4]]
        st = x[4:204]
        for a in range(5):
            for b in range(5):
                for k in range(8):
                    f.keccak_internal_state[a][b][k] = int(st[(a * 5 + b) * 8 + k])
        f.timestamp_to_use_for_read, f.timestamp_to_use_for_write = int(x[204]), int(x[205])
        (f.input_page, f.input_memory_byte_offset, f.input_memory_byte_length, f.output_page, f.output_word_offset, f.needs_full_padding_round) = [int(v) for v in x[206:212]]
        f.buffer_bytes[:] = [int(v) for v in x[212:404]]; f.buffer_filled = int(x[404])
        f.log_queue_state, f.memory_queue_state = _q4(x[405:414]), _q12(x[414:439])
        rq, rd = reqs[used_reqs:], all_reads[used_reads:]
        qa = (zkgl.LogQueryWitness * max(len(rq), 1))(*[_lq(r["query"]) for r in rq])
        ra = ((zkgl.C.c_uint32 * 8) * max(len(rd), 1))()
        for dst, v in zip(ra, rd):
            dst[:] = [(v >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
        w.requests_queue_witness, w.n_requests, w.memory_reads_witness, w.n_reads = qa, len(rq), ra, len(rd)
        n_popped = (len(reqs) - len(inst["rest"][0])) - used_reqs
        if not tails:
            zkgl.pack_keccak_witness(w, limit, i, outer, loop)
        else:
            # the queue states the reference's witnesses hold: the request queue's head before every pop of this instance, the memory
            # queue's tail after every push (the RAM permutation's unsorted queue witness)
            qa2 = (zkgl.LogQueryWitness * max(n_popped, 1))(*[_lq(r["query"]) for r in rq[:n_popped]])
            w.requests_queue_witness, w.n_requests = qa2, n_popped
            w.hidden_fsm_output.log_queue_state = _q4(inst["fsm_out"]["req"])
            start_req = o[1:10] if o[0] else x[405:414]
            start_mem = o[10:35] if o[0] else x[414:439]
            head, prev = [int(v) for v in start_req[0:4]], []
            for r in rq[:n_popped]:
                prev.append(head)
                head = zko.queue_tail4_push20(head, encode(r["query"]))
            mt, mtails = [int(v) for v in start_mem[12:24]], []
            for q in inst["pushed"]:
                mt = zko.queue_full_push(mt, zko.memory_query_encode(q))
                mtails.append(mt)
            given = zkgl.pack_keccak_witness_tails(w, limit, i, outer, loop, np.array(prev or [[0] * 4], dtype=np.uint64), np.array(mtails, dtype=np.uint64).reshape(-1, 12))
            assert given == list(range(423))
        used_reqs = len(reqs) - len(inst["rest"][0])
        used_reads = len(all_reads) - len(inst["rest"][1]) - sum(len(r["reads"]) for r in inst["rest"][0])
    return outer, loop, insts


@pytest.mark.gpu
def test_corrupted_poseidon2_intermediate_is_caught_by_the_macro_packet(zk):
    """the checker evaluates an in-circuit Poseidon2 permutation as ONE macro packet (all 621 relations on the stored values, each value
    loaded once); a corrupted stored intermediate must be caught there and named by the gate-by-gate re-run"""
    from helpers import ram_cs
    cases = [_case(700 + i, LIMIT) for i in range(2)]
    outer, loop = rn.pack_streams([c[3] for c in cases], LIMIT)
    cs = ram_cs(LIMIT)
    cs.set_batch(2)
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    n = cs.store_slots(True)
    rng = np.random.default_rng(5)
    kinds = set()
    for slot in sorted(set(int(x) for x in rng.integers(0, n, size=12))):
        cs.resolve()
        ok, f = cs.check_if_satisfied()
        assert ok, f
        lane = LIMIT + 7                       # instance 1, cycle 7
        cs.debug_poke_store(True, slot, lane, 0x1234567)
        ok, f = cs.check_if_satisfied()
        assert not ok and f.scope == 1 and f.instance == 1 and f.iteration == 7, (slot, ok, f)
        kinds.add(f.kind)
    assert zkgl.GATE["MATMUL12_EXT"] in kinds or zkgl.GATE["MATMUL12_INT"] in kinds or zkgl.GATE["FMA"] in kinds, kinds


# ---------------------------------------------------------------- demux / sort_decommits / code_unpacker / linear_hasher packers
def _streams(insts):
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    return outer, loop


def _dq(words):
    q = zkgl.DecommitQueryWitness()
    q.code_hash[:] = [int(x) for x in words[0:8]]; q.page, q.is_first, q.timestamp = int(words[8]), int(words[9]), int(words[10])
    return q


def _demux_packed():
    from oracle import demux_native as dn
    from oracle.storage_native import log_query
    rng = np.random.default_rng(61)
    qs = []
    for t in range(11):
        kind = int(rng.integers(0, 6))
        address = {3: 0x8010, 4: 0x02, 5: 0x01}.get(kind, int(rng.integers(1 << 20, 1 << 40)))
        qs.append(log_query(address=address, key=int.from_bytes(rng.bytes(32), "little"), read_value=int.from_bytes(rng.bytes(32), "little"),
                            written_value=int.from_bytes(rng.bytes(32), "little"), rw_flag=int(rng.integers(0, 2)), aux_byte=[0, 1, 2, 3, 3, 3][kind],
                            rollback=int(rng.integers(0, 2)), is_service=int(rng.integers(0, 2)), shard_id=0,
                            tx_number_in_block=int(rng.integers(0, 1000)), timestamp=100 + t))
    limit = 7
    a = dn.instance(qs, limit)
    b = dn.instance(a["rest"], limit, start_flag=False, fsm_in=a["fsm_out"], obs_initial=a["obs_initial"])
    assert a["satisfiable"] and b["satisfiable"] and b["completed"]
    insts, queues = [a, b], [qs, a["rest"]]
    outer = np.zeros((73, 2), dtype=np.uint64); loop = np.full((71, 2 * limit), 9, dtype=np.uint64)
    for i, (inst, queue) in enumerate(zip(insts, queues)):
        o = inst["outer"]
        w = zkgl.DemuxLogQueueWitness()
        w.start_flag, w.completion_flag = int(o[0]), int(inst["completed"])
        w.initial_log_queue_state = _q4(o[1:10])
        w.hidden_fsm_input.initial_log_queue_state = _q4(o[10:19])
        for k in range(6):
            w.hidden_fsm_input.output_queue_states[k] = _q4(o[19 + 9 * k:28 + 9 * k])
        popped = queue[:limit]
        arr = (zkgl.LogQueryWitness * max(len(popped), 1))(*[_lq(q) for q in popped])
        w.initial_queue_witness, w.n_initial = arr, len(popped)
        zkgl.pack_demux_witness(w, limit, i, outer, loop)
    return outer, loop, insts, limit


def test_demux_packer_equals_the_oracle_streams():
    """a start instance and its continuation (the reference's witness of the second instance holds the rest of the queue)"""
    outer, loop, insts, limit = _demux_packed()
    eo, el = _streams(insts)
    el = el.copy(); el[0:35] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)


def _demux_tails(inst, queue, limit):
    """what the reference's witnesses hold beside the items: the previous tail of every popped element (the input queue's head before
    the pop) and, for every element, the tail of its target queue after the push (the next circuit's previous tails)"""
    from oracle import demux_native as dn
    rows = inst["rows"]
    n = min(len(queue), limit)
    prev = np.zeros((max(n, 1), 4), dtype=np.uint64)
    out = np.zeros((max(n, 1), 4), dtype=np.uint64)
    fsm_out = inst["fsm_out"]["out"]
    for c in range(n):
        prev[c] = rows[c][0:4]
        k, good = dn.target_queue(queue[c])
        if k is not None:
            nxt = rows[c + 1][5 + 5 * k:9 + 5 * k] if c + 1 < limit else fsm_out[k][4:8]
            out[c] = nxt
    return prev, out


def _demux_packed_with_tails():
    outer, loop, insts, limit = _demux_packed()
    from oracle import demux_native as dn   # noqa: F401
    # rebuild the two witnesses of _demux_packed and pack them again with the queue states
    rng = np.random.default_rng(61)
    from oracle.storage_native import log_query
    qs = []
    for t in range(11):
        kind = int(rng.integers(0, 6))
        address = {3: 0x8010, 4: 0x02, 5: 0x01}.get(kind, int(rng.integers(1 << 20, 1 << 40)))
        qs.append(log_query(address=address, key=int.from_bytes(rng.bytes(32), "little"), read_value=int.from_bytes(rng.bytes(32), "little"),
                            written_value=int.from_bytes(rng.bytes(32), "little"), rw_flag=int(rng.integers(0, 2)), aux_byte=[0, 1, 2, 3, 3, 3][kind],
                            rollback=int(rng.integers(0, 2)), is_service=int(rng.integers(0, 2)), shard_id=0,
                            tx_number_in_block=int(rng.integers(0, 1000)), timestamp=100 + t))
    queues = [qs, insts[0]["rest"]]
    outer2 = np.zeros((73, 2), dtype=np.uint64); loop2 = np.full((71, 2 * limit), 9, dtype=np.uint64)
    given = None
    for i, (inst, queue) in enumerate(zip(insts, queues)):
        o = inst["outer"]
        w = zkgl.DemuxLogQueueWitness()
        w.start_flag, w.completion_flag = int(o[0]), int(inst["completed"])
        w.initial_log_queue_state = _q4(o[1:10])
        w.hidden_fsm_input.initial_log_queue_state = _q4(o[10:19])
        for k in range(6):
            w.hidden_fsm_input.output_queue_states[k] = _q4(o[19 + 9 * k:28 + 9 * k])
        popped = queue[:limit]
        arr = (zkgl.LogQueryWitness * max(len(popped), 1))(*[_lq(q) for q in popped])
        w.initial_queue_witness, w.n_initial = arr, len(popped)
        prev, out = _demux_tails(inst, queue, limit)
        given = zkgl.pack_demux_witness_tails(w, limit, i, outer2, loop2, prev, out)
    return outer2, loop2, insts, limit, given


def test_demux_packer_with_the_witness_queue_states_writes_every_carried_word():
    """zk_pack_demux_witness_tails: with the previous tails of the input witness and the tails the next circuits' witnesses hold, the
    packer's stream — carried words included — IS the native restatement's: nothing is left to seed"""
    outer, loop, insts, limit, given = _demux_packed_with_tails()
    eo, el = _streams(insts)
    assert given == list(range(35))
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)


@pytest.mark.gpu
def test_demux_with_the_witness_queue_states_needs_no_device_seeding(zk):
    from test_demux_host import demux_cs
    outer, loop, insts, limit, given = _demux_packed_with_tails()
    cs = demux_cs(limit)
    cs.set_batch(len(insts))
    cs.set_seed_given(given)
    try:
        d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
        cs.bind_inputs(False, d_o, outer.shape[0])
        cs.bind_inputs(True, d_l, loop.shape[0])
        cs.seed_carried_inputs(d_l)                      # every carried word is declared given: no kernel runs
        assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
        ok, f = cs.resolve_and_check()
        assert ok, f
        for i, inst in enumerate(insts):
            assert cs.public_inputs(i) == inst["public_input"]
        bad = loop.copy(); bad[7, 3] ^= 1                # a wrong tail word from the host: the circuit's own constraints reject it
        d_b = zk.DeviceBuffer.from_numpy(bad)
        cs.bind_inputs(True, d_b, bad.shape[0])
        ok, f = cs.resolve_and_check()
        assert not ok
    finally:
        cs.set_seed_given([])


def test_sort_decommits_packer_equals_the_oracle_streams():
    from oracle import decommit_native as dn
    u, s = dn.random_decommits(np.random.default_rng(62), 5)
    limit = len(u) + 3
    inst = dn.instance(u, s, limit)
    assert inst["satisfiable"] and inst["completed"]
    o = inst["outer"]
    w = zkgl.SortDecommitsWitness()
    w.start_flag, w.completion_flag = int(o[0]), 1
    w.initial_queue_state, w.sorted_queue_initial_state = _q12(o[1:26]), _q12(o[26:51])
    f, x = w.hidden_fsm_input, o[51:151]
    f.initial_queue_state, f.sorted_queue_state, f.final_queue_state = _q12(x[0:25]), _q12(x[25:50]), _q12(x[50:75])
    f.lhs_accumulator[:] = x[75:77]; f.rhs_accumulator[:] = x[77:79]; f.previous_packed_key[:] = x[79:88]
    f.first_encountered_timestamp = int(x[88]); f.previous_record = _dq(x[89:100])
    ua = (zkgl.DecommitQueryWitness * len(u))(*[_dq(q) for q in u])
    sa = (zkgl.DecommitQueryWitness * len(s))(*[_dq(q) for q in s])
    w.initial_queue_witness, w.n_initial, w.sorted_queue_witness, w.n_sorted = ua, len(u), sa, len(s)
    outer = np.zeros((151, 1), dtype=np.uint64); loop = np.full((87, limit), 9, dtype=np.uint64)
    zkgl.pack_sort_decommits_witness(w, limit, 0, outer, loop)
    eo, el = _streams([inst])
    el = el.copy(); el[0:65] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)
    w.n_sorted = len(s) - 1
    with pytest.raises(zkgl.ZkError):
        zkgl.pack_sort_decommits_witness(w, limit, 0, outer, loop)


def _sort_decommits_packed_with_tails():
    """a start instance that stops in the middle of the queues and the continuation that finishes them, packed with the queue states
    the reference's witnesses hold"""
    from oracle import decommit_native as dn, zko
    u, s = dn.random_decommits(np.random.default_rng(65), 6, max_repeats=3)
    limit = (len(u) + 3) // 2 + 1
    a = dn.instance(u, s, limit)
    b = dn.instance(a["rest"][0], a["rest"][1], limit, start_flag=False, fsm_in=a["fsm_out"], obs=a["obs"])
    assert a["satisfiable"] and b["satisfiable"] and not a["completed"] and b["completed"]
    insts, queues = [a, b], [(u, s), a["rest"]]
    outer = np.zeros((151, 2), dtype=np.uint64); loop = np.full((87, 2 * limit), 9, dtype=np.uint64)
    given = None
    for i, (inst, (uq, sq)) in enumerate(zip(insts, queues)):
        o = inst["outer"]
        w = zkgl.SortDecommitsWitness()
        w.start_flag, w.completion_flag = int(o[0]), int(inst["completed"])
        w.initial_queue_state, w.sorted_queue_initial_state = _q12(o[1:26]), _q12(o[26:51])
        f, x = w.hidden_fsm_input, o[51:151]
        f.initial_queue_state, f.sorted_queue_state, f.final_queue_state = _q12(x[0:25]), _q12(x[25:50]), _q12(x[50:75])
        f.lhs_accumulator[:] = x[75:77]; f.rhs_accumulator[:] = x[77:79]; f.previous_packed_key[:] = x[79:88]
        f.first_encountered_timestamp = int(x[88]); f.previous_record = _dq(x[89:100])
        fo = inst["fsm_out"]
        w.hidden_fsm_output.initial_queue_state, w.hidden_fsm_output.sorted_queue_state = _q12(fo["initial"]), _q12(fo["sorted"])
        uq, sq = uq[:limit], sq[:limit]                      # the elements this instance pops
        ua = (zkgl.DecommitQueryWitness * max(len(uq), 1))(*[_dq(q) for q in uq])
        sa = (zkgl.DecommitQueryWitness * max(len(sq), 1))(*[_dq(q) for q in sq])
        w.initial_queue_witness, w.n_initial, w.sorted_queue_witness, w.n_sorted = ua, len(uq), sa, len(sq)
        st_u = (o[1:26] if o[0] else x[0:25]); st_s = (o[26:51] if o[0] else x[25:50]); st_r = [0] * 25 if o[0] else x[50:75]
        hu, hs, pu, ps = [int(v) for v in st_u[0:12]], [int(v) for v in st_s[0:12]], [], []
        for q, r in zip(uq, sq):
            pu.append(hu); ps.append(hs)
            hu, hs = zko.queue_full_push(hu, dn.encode(q)), zko.queue_full_push(hs, dn.encode(r))
        rt, rts = [int(v) for v in st_r[12:24]], []
        # pushes inside the loop (a completed instance pushes its last record after it, mod.rs:347-360): previous record not
        # trivial and its hash differs from the sorted element's
        n_loop_pushes = sum(1 for r in inst["rows"] if not r[0] and r[54:62] != r[76:84])
        for q in inst["result"][:n_loop_pushes]:
            rt = zko.queue_full_push(rt, dn.encode(q))
            rts.append(rt)
        given = zkgl.pack_sort_decommits_witness_tails(w, limit, i, outer, loop, np.array(pu or [[0] * 12], dtype=np.uint64),
                                                       np.array(ps or [[0] * 12], dtype=np.uint64), np.array(rts, dtype=np.uint64).reshape(-1, 12))
    return outer, loop, insts, limit, given


def test_sort_decommits_packer_with_the_witness_queue_states_leaves_only_the_grand_products():
    """zk_pack_sort_decommits_witness_tails: 61 of the 65 carried words of every cycle equal the native restatement's; words 1..4 (the
    grand-product accumulators, which need the circuit's challenges) stay for the device scan"""
    outer, loop, insts, limit, given = _sort_decommits_packed_with_tails()
    eo, el = _streams(insts)
    assert given == [0] + list(range(5, 65))
    el = el.copy(); el[1:5] = 0
    assert np.array_equal(outer, eo)
    assert np.array_equal(loop, el), np.argwhere(loop != el)[:8]


@pytest.mark.gpu
def test_sort_decommits_with_the_witness_queue_states_seeds_by_a_scan(zk):
    from test_decommit_host import decommit_cs
    outer, loop, insts, limit, given = _sort_decommits_packed_with_tails()
    cs = decommit_cs(limit)
    cs.set_batch(len(insts))
    cs.set_seed_given(given)
    try:
        d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
        cs.bind_inputs(False, d_o, outer.shape[0])
        cs.bind_inputs(True, d_l, loop.shape[0])
        cs.seed_carried_inputs(d_l)                      # k_decommit_seed: the two grand products per repetition
        _, el = _streams(insts)
        assert np.array_equal(d_l.to_numpy().reshape(loop.shape), el)
        ok, f = cs.resolve_and_check()
        assert ok, f
        for i, inst in enumerate(insts):
            assert cs.public_inputs(i) == inst["public_input"]
    finally:
        cs.set_seed_given([])


def _code_unpacker_packed(tails=False):
    from oracle import code_unpacker_native as cn, zko
    from oracle.decommit_native import dq, encode
    rng = np.random.default_rng(63)
    reqs = []
    for k, n in enumerate((1, 5, 3)):
        words = [int.from_bytes(rng.bytes(32), "big") for _ in range(n)]
        reqs.append((dq(cn.versioned_hash(words), 2000 + 8 * k, 1, 100 + k), words))
    limit = 3
    a = cn.instance(reqs, limit)
    b = cn.instance(a["rest"][0], limit, start_flag=False, fsm_in=a["fsm_out"], obs=a["obs"], pending=a["rest"][1])
    assert a["satisfiable"] and b["satisfiable"] and b["fsm_out"]["finished"] == 1
    insts = [a, b]
    remaining = [(reqs, []), (a["rest"][0], a["rest"][1])]
    outer = np.zeros((125, 2), dtype=np.uint64); loop = np.full((101, 2 * limit), 9, dtype=np.uint64)
    for i, (inst, (rq, pending)) in enumerate(zip(insts, remaining)):
        o = inst["outer"]
        w = zkgl.CodeUnpackerWitness()
        w.start_flag = int(o[0])
        w.sorted_requests_queue_initial_state, w.memory_queue_initial_state = _q12(o[1:26]), _q12(o[26:51])
        f, x = w.hidden_fsm_input, o[51:125]
        f.sha256_inner_state[:] = x[0:8]; f.hash_to_compare_against[:] = x[8:16]
        f.current_index, f.current_page, f.timestamp, f.num_rounds_left, f.length_in_bits = [int(v) for v in x[16:21]]
        f.state_get_from_queue, f.state_decommit, f.finished = [int(v) for v in x[21:24]]
        f.decommittment_requests_queue_state, f.memory_queue_state = _q12(x[24:49]), _q12(x[49:74])
        qa = (zkgl.DecommitQueryWitness * max(len(rq), 1))(*[_dq(q) for q, _ in rq])
        words = list(pending) + [wd for _, ws in rq for wd in ws]   # the words of the code in flight, then the queued bytecodes
        wa = ((C_u32x8) * max(len(words), 1))()
        for dst, v in zip(wa, words):
            dst[:] = [(v >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
        w.sorted_requests_queue_witness, w.n_requests, w.code_words, w.n_code_words = qa, len(rq), wa, len(words)
        if not tails:
            zkgl.pack_code_unpacker_witness(w, limit, i, outer, loop)
            continue
        # what the reference's witnesses hold beside the elements: the requests queue's head before every pop, the memory queue's
        # tail after every push (= the previous tails of the RAM permutation's unsorted queue witness)
        start = o[1:26] if o[0] else x[24:49]
        head, prev = [int(v) for v in start[0:12]], []
        for q, _ in rq:
            prev.append(head)
            head = zko.queue_full_push(head, encode(q))
        mt, mtails = [int(v) for v in (o[26:51] if o[0] else x[49:74])[12:24]], []
        for m in inst["pushed"]:
            mt = zko.queue_full_push(mt, zko.memory_query_encode(m))
            mtails.append(mt)
        w.code_words, w.n_code_words = wa, len(inst["pushed"])      # the words this instance consumes
        w.hidden_fsm_output.decommittment_requests_queue_state = _q12(inst["fsm_out"]["req"])
        given = zkgl.pack_code_unpacker_witness_tails(w, limit, i, outer, loop, np.array(prev or [[0] * 12], dtype=np.uint64),
                                                      np.array(mtails or [[0] * 12], dtype=np.uint64))
        assert given == list(range(74))
    return outer, loop, insts, limit


def test_code_unpacker_packer_walks_the_fsm_schedule():
    """requests and code words land on the cycles that consume them: three bytecodes (1, 5 and 3 words: 1 + 3 + 2 rounds) over a start
    instance and a continuation that begins in the middle of the second bytecode"""
    outer, loop, insts, limit = _code_unpacker_packed()
    eo, el = _streams(insts)
    el = el.copy(); el[0:74] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)


def test_code_unpacker_packer_with_the_witness_queue_states_writes_every_carried_word():
    """zk_pack_code_unpacker_witness_tails: FSM scalars and the SHA-256 state walked natively, queue states taken from the witness — the
    stream equals the native restatement's in all 101 words of every cycle"""
    outer, loop, insts, limit = _code_unpacker_packed(tails=True)
    eo, el = _streams(insts)
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)


@pytest.mark.gpu
def test_code_unpacker_with_the_witness_queue_states_needs_no_device_seeding(zk):
    from test_code_unpacker_host import unpacker_cs
    outer, loop, insts, limit = _code_unpacker_packed(tails=True)
    cs = unpacker_cs(limit)
    cs.set_batch(len(insts))
    cs.set_seed_given(list(range(74)))
    try:
        d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
        cs.bind_inputs(False, d_o, outer.shape[0])
        cs.bind_inputs(True, d_l, loop.shape[0])
        cs.seed_carried_inputs(d_l)
        assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
        ok, f = cs.resolve_and_check()
        assert ok, f
        for i, inst in enumerate(insts):
            assert cs.public_inputs(i) == inst["public_input"]
        bad = loop.copy(); bad[3, 2] ^= 1                # a wrong byte of the SHA state from the host: the links reject it
        d_b = zk.DeviceBuffer.from_numpy(bad)
        cs.bind_inputs(True, d_b, bad.shape[0])
        ok, f = cs.resolve_and_check()
        assert not ok
    finally:
        cs.set_seed_given([])


@pytest.mark.gpu
@pytest.mark.parametrize("circuit", ["demux", "code_unpacker"])
def test_packed_streams_of_a_start_and_a_continuation_instance_run_on_the_gpu(zk, circuit):
    """the product path end to end for two more circuits: C-ABI packers -> device seeding -> resolve + check; commitments and the seeded
    streams equal the native restatement's (instance 1 continues instance 0: its hidden_fsm_input is instance 0's output)"""
    if circuit == "demux":
        from test_demux_host import demux_cs
        outer, loop, insts, limit = _demux_packed()
        cs = demux_cs(limit)
    else:
        from test_code_unpacker_host import unpacker_cs
        outer, loop, insts, limit = _code_unpacker_packed()
        cs = unpacker_cs(limit)
    cs.set_batch(len(insts))
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    cs.seed_carried_inputs(d_l)
    _, el = _streams(insts)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), el)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["public_input"]


def test_linear_hasher_packer_equals_the_oracle_streams():
    from oracle import linear_hasher_native as hn
    from oracle.storage_native import log_query
    rng = np.random.default_rng(64)
    qs = [log_query(address=int(rng.integers(1, 1 << 60)), key=int.from_bytes(rng.bytes(32), "little"),
                    written_value=int.from_bytes(rng.bytes(32), "little"), rw_flag=1, aux_byte=2, is_service=int(rng.integers(0, 2)),
                    shard_id=int(rng.integers(0, 2)), tx_number_in_block=int(rng.integers(0, 65536)), timestamp=5 + t) for t in range(20)]
    limit = 34
    inst = hn.instance(qs, limit)
    assert inst["satisfiable"]
    w = zkgl.LinearHasherWitness()
    w.start_flag, w.completion_flag = 1, 1
    w.queue_state = _q4(inst["outer"][1:10])
    arr = (zkgl.LogQueryWitness * len(qs))(*[_lq(q) for q in qs])
    w.queue_witness, w.n_queue = arr, len(qs)
    outer = np.zeros((10, 1), dtype=np.uint64); loop = np.full((818, 2), 9, dtype=np.uint64)
    zkgl.pack_linear_hasher_witness(w, limit, 0, outer, loop)
    eo, el = _streams([inst])
    el = el.copy(); el[0:206] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)
    with pytest.raises(zkgl.ZkError):
        zkgl._check(zkgl.lib().zk_pack_linear_hasher_witness(C.byref(w), 33, 0, 1, outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p)))
    # with the previous tails of the queue witness the packer also walks the sponge: all 818 words of every period
    from oracle import zko
    from oracle.storage_native import encode
    for n_q, lim in ((20, 34), (0, 17), (17, 17), (31, 51)):    # queue ends inside a period / empty / exactly at the period's end / block-aligned total (31 * 88 = 2728 = 20 * 136 + 8)
        qq = qs[:n_q] if n_q <= len(qs) else qs + qs[:n_q - len(qs)]
        inst = hn.instance(qq, lim)
        w = zkgl.LinearHasherWitness()
        w.start_flag, w.completion_flag = 1, 1
        w.queue_state = _q4(inst["outer"][1:10])
        arr = (zkgl.LogQueryWitness * max(len(qq), 1))(*[_lq(q) for q in qq])
        w.queue_witness, w.n_queue = arr, len(qq)
        head, prev = [0] * 4, []
        for q in qq:
            prev.append(head)
            head = zko.queue_tail4_push20(head, encode(q))
        outer = np.zeros((10, 1), dtype=np.uint64); loop = np.full((818, lim // 17), 9, dtype=np.uint64)
        given = zkgl.pack_linear_hasher_witness_tails(w, lim, 0, outer, loop, np.array(prev or [[0] * 4], dtype=np.uint64))
        eo, el = _streams([inst])
        assert given == list(range(206))
        assert np.array_equal(outer, eo) and np.array_equal(loop, el), (n_q, lim, np.argwhere(loop != el)[:6])


# ---------------------------------------------------------------- bincode of the LogQuery-queue witnesses (test-side writer, C decoder)
def _b_h160(limbs):
    v = sum(int(x) << (32 * i) for i, x in enumerate(limbs))
    s_ = ("0x%040x" % v).encode()
    return struct.pack("<Q", len(s_)) + s_


def _b_q4(q):
    return b"".join(struct.pack("<Q", int(x)) for x in list(q.head) + list(q.tail)) + struct.pack("<I", q.length)


def _b_lq(q):
    return (_b_h160(q.address) + _b_u256(q.key) + _b_u256(q.read_value) + _b_u256(q.written_value) +
            struct.pack("<BBBBBII", q.aux_byte, q.rw_flag, q.rollback, q.is_service, q.shard_id, q.tx_number_in_block, q.timestamp))


def _b_log_queue(arr, n, with_ts=False, tails=None):
    out = struct.pack("<Q", n)
    for i in range(n):
        out += (_b_lq(arr[i].record) + struct.pack("<I", arr[i].timestamp)) if with_ts else _b_lq(arr[i])
        # the tail before the push (the plain decoders skip it, the _tails ones keep it)
        out += b"".join(struct.pack("<Q", int(tails[i][t]) if tails is not None else 1000 + 4 * i + t) for t in range(4))
    return out


def _b_storage_fsm(f):
    return (b"".join(struct.pack("<Q", int(x)) for x in list(f.lhs_accumulator) + list(f.rhs_accumulator)) + _b_q4(f.current_unsorted_queue_state) +
            _b_q4(f.current_intermediate_sorted_queue_state) + _b_q4(f.current_final_sorted_queue_state) + struct.pack("<I", f.cycle_idx) +
            b"".join(struct.pack("<I", int(x)) for x in f.previous_packed_key) + _b_u256(f.previous_key) + _b_h160(f.previous_address) +
            struct.pack("<IB", f.previous_timestamp, f.this_cell_has_explicit_read_and_rollback_depth_zero) + _b_u256(f.this_cell_base_value) +
            _b_u256(f.this_cell_current_value) + struct.pack("<I", f.this_cell_current_depth))


def _b_log_sorter_fsm(f):
    return (b"".join(struct.pack("<Q", int(x)) for x in list(f.lhs_accumulator) + list(f.rhs_accumulator)) + _b_q4(f.initial_unsorted_queue_state) +
            _b_q4(f.intermediate_sorted_queue_state) + _b_q4(f.final_result_queue_state) + struct.pack("<I", f.previous_key) + _b_lq(f.previous_item))


def test_storage_bincode_round_trip():
    """a continuation-shaped witness (non-trivial FSM input: accumulators, previous key / address / values) through bytes and back"""
    from oracle import storage_native as sn
    limit = 12
    u, s_ = sn.random_storage_witness(np.random.default_rng(71), 9, n_cells=3)
    a = sn.instance(u, s_, limit)
    o = a["outer"]
    w = zkgl.StorageValidityWitness()
    w.start_flag, w.completion_flag, w.shard_id_to_process = int(o[0]), int(a["completed"]), int(o[1])
    w.unsorted_log_queue_state, w.intermediate_sorted_queue_state = _q4(o[2:11]), _q4(o[11:20])
    for f, x in ((w.hidden_fsm_input, o[20:97]), (w.hidden_fsm_output, sn.flatten_fsm(a["fsm_out"]) if hasattr(sn, "flatten_fsm") else o[20:97])):
        f.lhs_accumulator[:] = x[0:2]; f.rhs_accumulator[:] = x[2:4]
        f.current_unsorted_queue_state, f.current_intermediate_sorted_queue_state, f.current_final_sorted_queue_state = _q4(x[4:13]), _q4(x[13:22]), _q4(x[22:31])
        f.cycle_idx = int(x[31]); f.previous_packed_key[:] = x[32:45]; f.previous_key[:] = x[45:53]; f.previous_address[:] = x[53:58]
        f.previous_timestamp = int(x[58]); f.this_cell_has_explicit_read_and_rollback_depth_zero = int(x[59])
        f.this_cell_base_value[:] = x[60:68]; f.this_cell_current_value[:] = x[68:76]; f.this_cell_current_depth = int(x[76])
    ua = (zkgl.LogQueryWitness * len(u))(*[_lq(q) for q in u])
    sa = (zkgl.TimestampedLogRecordWitness * len(s_))()
    for rec, (q, t) in zip(sa, s_):
        rec.record, rec.timestamp = _lq(q), int(t)
    w.unsorted_queue_witness, w.n_unsorted, w.intermediate_sorted_queue_witness, w.n_sorted = ua, len(u), sa, len(s_)
    data = (struct.pack("<BBB", w.start_flag, w.completion_flag, w.shard_id_to_process) + _b_q4(w.unsorted_log_queue_state) + _b_q4(w.intermediate_sorted_queue_state) +
            _b_q4(w.hidden_fsm_output.current_final_sorted_queue_state) + _b_storage_fsm(w.hidden_fsm_input) + _b_storage_fsm(w.hidden_fsm_output) +
            _b_log_queue(ua, len(u)) + _b_log_queue(sa, len(s_), with_ts=True))
    d, used = zkgl.decode_storage_witness_bincode(data + b"xx", limit)
    assert used == len(data)
    outer = np.zeros((97, 1), dtype=np.uint64); loop = np.zeros((140, limit), dtype=np.uint64)
    zkgl.pack_storage_witness(d, limit, 0, outer, loop)
    eo, el = sn.pack_streams([a], limit)
    el = el.copy(); el[0:67] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)
    assert bytes(d.hidden_fsm_output) == bytes(w.hidden_fsm_output)
    with pytest.raises(zkgl.ZkError):
        zkgl.decode_storage_witness_bincode(data[:-5], limit)
    with pytest.raises(zkgl.ZkError):
        zkgl.decode_storage_witness_bincode(data, len(u) - 1)


def test_log_sorter_demux_linear_hasher_bincode_round_trips():
    from oracle import demux_native as dn, linear_hasher_native as hn, log_sorter_native as ln
    # log_sorter
    limit = 14
    u, s_ = ln.random_events(np.random.default_rng(72), 10, rollback_frac=0.3)
    u, s_ = u[:limit], s_[:limit]
    inst = ln.instance(u, s_, limit)
    o = inst["outer"]
    w = zkgl.LogSorterWitness()
    w.start_flag, w.completion_flag = int(o[0]), int(inst["completed"])
    w.initial_log_queue_state, w.intermediate_sorted_queue_state = _q4(o[1:10]), _q4(o[10:19])
    f, x = w.hidden_fsm_input, o[19:87]
    f.lhs_accumulator[:] = x[0:2]; f.rhs_accumulator[:] = x[2:4]
    f.initial_unsorted_queue_state, f.intermediate_sorted_queue_state, f.final_result_queue_state = _q4(x[4:13]), _q4(x[13:22]), _q4(x[22:31])
    f.previous_key = int(x[31]); f.previous_item = _lq(x[32:68])
    ua = (zkgl.LogQueryWitness * len(u))(*[_lq(q) for q in u]); sa = (zkgl.LogQueryWitness * len(s_))(*[_lq(q) for q in s_])
    data = (struct.pack("<BB", w.start_flag, w.completion_flag) + _b_q4(w.initial_log_queue_state) + _b_q4(w.intermediate_sorted_queue_state) +
            _b_q4(zkgl.QueueStateWitness()) + _b_log_sorter_fsm(w.hidden_fsm_input) + _b_log_sorter_fsm(w.hidden_fsm_output) +
            _b_log_queue(ua, len(u)) + _b_log_queue(sa, len(s_)))
    d, used = zkgl.decode_log_sorter_witness_bincode(data, limit)
    assert used == len(data)
    outer = np.zeros((87, 1), dtype=np.uint64); loop = np.zeros((129, limit), dtype=np.uint64)
    zkgl.pack_log_sorter_witness(d, limit, 0, outer, loop)
    eo, el = ln.pack_streams([inst], limit)
    el = el.copy(); el[0:57] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)
    # demux: the packed start instance of _demux_packed, through bytes
    pouter, ploop, insts, limit = _demux_packed()
    o = insts[0]["outer"]
    w = zkgl.DemuxLogQueueWitness()
    w.start_flag, w.completion_flag = int(o[0]), int(insts[0]["completed"])
    w.initial_log_queue_state = _q4(o[1:10])
    w.hidden_fsm_input.initial_log_queue_state = _q4(o[10:19])
    for k in range(6):
        w.hidden_fsm_input.output_queue_states[k] = _q4(o[19 + 9 * k:28 + 9 * k])
    rows = [r[35:71] for r in insts[0]["rows"]]
    n = sum(1 for r in rows if any(r))
    qa = (zkgl.LogQueryWitness * max(n, 1))(*[_lq(r) for r in rows[:n]])
    fsm = lambda f: _b_q4(f.initial_log_queue_state) + b"".join(_b_q4(q) for q in f.output_queue_states)
    data = (struct.pack("<BB", w.start_flag, w.completion_flag) + _b_q4(w.initial_log_queue_state) + b"".join(_b_q4(zkgl.QueueStateWitness()) for _ in range(6)) +
            fsm(w.hidden_fsm_input) + fsm(w.hidden_fsm_output) + _b_log_queue(qa, n))
    d, used = zkgl.decode_demux_witness_bincode(data, limit)
    assert used == len(data)
    outer = np.zeros((73, 2), dtype=np.uint64); loop = np.full((71, 2 * limit), 9, dtype=np.uint64)
    zkgl.pack_demux_witness(d, limit, 0, outer, loop)
    assert np.array_equal(outer[:, 0], pouter[:, 0]) and np.array_equal(loop[:, :limit], ploop[:, :limit])
    d, used = zkgl.decode_demux_witness_bincode(data, limit, keep_tails=True)   # the previous tails bincode carries beside the elements
    assert used == len(data) and all(list(d._keep[-1][i]) == [1000 + 4 * i + t for t in range(4)] for i in range(n))
    # end to end with the REAL previous tails on the wire: bytes -> decode (tails kept) -> packer with tails == the native stream, carried
    # words included (the output tails are the next circuits' previous tails; taken from the native restatement here)
    irows = insts[0]["rows"]
    prev = [irows[c][0:4] for c in range(n)]
    data2 = (struct.pack("<BB", w.start_flag, w.completion_flag) + _b_q4(w.initial_log_queue_state) + b"".join(_b_q4(zkgl.QueueStateWitness()) for _ in range(6)) +
             fsm(w.hidden_fsm_input) + fsm(w.hidden_fsm_output) + _b_log_queue(qa, n, tails=prev))
    d, used = zkgl.decode_demux_witness_bincode(data2, limit, keep_tails=True)
    from oracle import demux_native as dmn
    out_t = np.zeros((max(n, 1), 4), dtype=np.uint64)
    for c in range(n):
        k, _ = dmn.target_queue([int(x) for x in irows[c][35:71]])
        if k is not None:
            out_t[c] = irows[c + 1][5 + 5 * k:9 + 5 * k] if c + 1 < limit else insts[0]["fsm_out"]["out"][k][4:8]
    o3 = np.zeros((73, 2), dtype=np.uint64); l3 = np.full((71, 2 * limit), 9, dtype=np.uint64)
    zkgl.pack_demux_witness_tails(d, limit, 0, o3, l3, np.array([list(t) for t in d._keep[-1]], dtype=np.uint64), out_t)
    eo, el = _streams(insts)
    assert np.array_equal(o3[:, 0], eo[:, 0]) and np.array_equal(l3[:, :limit], el[:, :limit])
    # linear_hasher
    from oracle.storage_native import log_query
    rng = np.random.default_rng(73)
    qs = [log_query(address=int(rng.integers(1, 1 << 60)), key=int.from_bytes(rng.bytes(32), "little"), written_value=int.from_bytes(rng.bytes(32), "little"),
                    rw_flag=1, aux_byte=2, shard_id=int(rng.integers(0, 2)), tx_number_in_block=int(rng.integers(0, 65536)), timestamp=5 + t) for t in range(9)]
    inst = hn.instance(qs, 17)
    w = zkgl.LinearHasherWitness()
    w.start_flag, w.completion_flag = 1, 1
    w.queue_state = _q4(inst["outer"][1:10])
    qa = (zkgl.LogQueryWitness * len(qs))(*[_lq(q) for q in qs])
    data = struct.pack("<BB", 1, 1) + _b_q4(w.queue_state) + inst["digest"] + _b_log_queue(qa, len(qs))
    d, used = zkgl.decode_linear_hasher_witness_bincode(data, 17)
    assert used == len(data)
    outer = np.zeros((10, 1), dtype=np.uint64); loop = np.zeros((818, 1), dtype=np.uint64)
    zkgl.pack_linear_hasher_witness(d, 17, 0, outer, loop)
    eo, el = _streams([inst])
    el = el.copy(); el[0:206] = 0
    assert np.array_equal(outer, eo) and np.array_equal(loop, el)
    with pytest.raises(zkgl.ZkError):
        zkgl.decode_linear_hasher_witness_bincode(data[:40], 17)


# ---------------------------------------------------------------- bincode of the precompile / decommitment witnesses
P_GL = 0xFFFFFFFF00000001


def _rand_q(rng, cls, width):
    q = cls()
    q.head[:] = [int(rng.integers(0, P_GL, dtype=np.uint64)) for _ in range(width)]
    q.tail[:] = [int(rng.integers(0, P_GL, dtype=np.uint64)) for _ in range(width)]
    q.length = int(rng.integers(0, 1 << 32))
    return q


def _rand_limbs(rng, n, zero_top=0):
    v = [int(x) for x in rng.integers(0, 1 << 32, size=n)]
    for i in range(zero_top):
        v[n - 1 - i] = 0          # short hex strings: U256 is written without leading zeros
    return v


def _rand_lq_struct(rng):
    q = zkgl.LogQueryWitness()
    q.address[:] = _rand_limbs(rng, 5, int(rng.integers(0, 3))); q.key[:] = _rand_limbs(rng, 8, int(rng.integers(0, 9)))
    q.read_value[:] = _rand_limbs(rng, 8); q.written_value[:] = _rand_limbs(rng, 8, 8)
    q.aux_byte, q.shard_id = int(rng.integers(0, 256)), int(rng.integers(0, 256))
    q.rw_flag, q.rollback, q.is_service = [int(x) for x in rng.integers(0, 2, size=3)]
    q.tx_number_in_block, q.timestamp = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
    return q


def _rand_dq_struct(rng):
    q = zkgl.DecommitQueryWitness()
    q.code_hash[:] = _rand_limbs(rng, 8, int(rng.integers(0, 2)))
    q.page, q.is_first, q.timestamp = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 2)), int(rng.integers(0, 1 << 32))
    return q


def _b_q12(q):
    return _b_q4(q)   # same shape: head, tail, length


def _b_dq(q):
    return _b_u256(q.code_hash) + struct.pack("<IBI", q.page, q.is_first, q.timestamp)


def _b_dq_queue(arr, n):
    out = struct.pack("<Q", n)
    for i in range(n):
        out += _b_dq(arr[i]) + b"".join(struct.pack("<Q", 77 + 12 * i + t) for t in range(12))
    return out


def _b_u256_seq(arr, n):
    return struct.pack("<Q", n) + b"".join(_b_u256(arr[i]) for i in range(n))


def _words(rng, n):
    arr = (C_u32x8 * max(n, 1))()
    for i in range(n):
        arr[i][:] = _rand_limbs(rng, 8, int(rng.integers(0, 9)))
    return arr


def test_sha256_and_keccak_bincode_round_trips():
    rng = np.random.default_rng(81)
    for kind in ("sha256", "keccak"):
        w = zkgl.Sha256RoundFunctionWitness() if kind == "sha256" else zkgl.KeccakRoundFunctionWitness()
        w.start_flag, w.completion_flag = 0, 1
        w.initial_log_queue_state, w.initial_memory_queue_state = _rand_q(rng, zkgl.QueueStateWitness, 4), _rand_q(rng, zkgl.FullQueueStateWitness, 12)
        fsm_bytes = []
        for f in (w.hidden_fsm_input, w.hidden_fsm_output):
            f.log_queue_state, f.memory_queue_state = _rand_q(rng, zkgl.QueueStateWitness, 4), _rand_q(rng, zkgl.FullQueueStateWitness, 12)
            f.timestamp_to_use_for_read, f.timestamp_to_use_for_write = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
            if kind == "sha256":
                f.read_precompile_call, f.read_words_for_round, f.completed = [int(x) for x in rng.integers(0, 2, size=3)]
                f.sha256_inner_state[:] = _rand_limbs(rng, 8)
                f.input_page, f.input_offset, f.output_page, f.output_offset, f.num_rounds = _rand_limbs(rng, 5)
                b = (struct.pack("<BBB", f.read_precompile_call, f.read_words_for_round, f.completed) + struct.pack("<8I", *f.sha256_inner_state) +
                     struct.pack("<7I", f.timestamp_to_use_for_read, f.timestamp_to_use_for_write, f.input_page, f.input_offset, f.output_page, f.output_offset, f.num_rounds))
            else:
                f.read_precompile_call, f.read_unaligned_words_for_round, f.padding_round, f.completed = [int(x) for x in rng.integers(0, 2, size=4)]
                st = bytes(rng.integers(0, 256, size=200, dtype=np.uint8))
                for a in range(5):
                    for c in range(5):
                        for k in range(8):
                            f.keccak_internal_state[a][c][k] = st[(a * 5 + c) * 8 + k]
                f.input_page, f.input_memory_byte_offset, f.input_memory_byte_length, f.output_page, f.output_word_offset = _rand_limbs(rng, 5)
                f.needs_full_padding_round = int(rng.integers(0, 2))
                buf = bytes(rng.integers(0, 256, size=192, dtype=np.uint8))
                f.buffer_bytes[:] = list(buf); f.buffer_filled = int(rng.integers(0, 193))
                b = (struct.pack("<BBBB", f.read_precompile_call, f.read_unaligned_words_for_round, f.padding_round, f.completed) + st +
                     struct.pack("<7I", f.timestamp_to_use_for_read, f.timestamp_to_use_for_write, f.input_page, f.input_memory_byte_offset,
                                 f.input_memory_byte_length, f.output_page, f.output_word_offset) + struct.pack("<B", f.needs_full_padding_round) + buf +
                     struct.pack("<B", f.buffer_filled))
            fsm_bytes.append(b + _b_q4(f.log_queue_state) + _b_q12(f.memory_queue_state))
        nq, nr = 3, 7
        qa = (zkgl.LogQueryWitness * nq)(*[_rand_lq_struct(rng) for _ in range(nq)])
        ra = _words(rng, nr)
        data = (struct.pack("<BB", w.start_flag, w.completion_flag) + _b_q4(w.initial_log_queue_state) + _b_q12(w.initial_memory_queue_state) +
                _b_q12(_rand_q(rng, zkgl.FullQueueStateWitness, 12)) + fsm_bytes[0] + fsm_bytes[1] + _b_log_queue(qa, nq) + _b_u256_seq(ra, nr))
        dec = zkgl.decode_sha256_witness_bincode if kind == "sha256" else zkgl.decode_keccak_witness_bincode
        d, used = dec(data + b"!", nq, nr)
        assert used == len(data) and (d.start_flag, d.completion_flag, d.n_requests, d.n_reads) == (0, 1, nq, nr)
        assert bytes(d.initial_log_queue_state) == bytes(w.initial_log_queue_state) and bytes(d.initial_memory_queue_state) == bytes(w.initial_memory_queue_state)
        assert bytes(d.hidden_fsm_input) == bytes(w.hidden_fsm_input) and bytes(d.hidden_fsm_output) == bytes(w.hidden_fsm_output)
        assert all(bytes(d.requests_queue_witness[i]) == bytes(qa[i]) for i in range(nq))
        assert all(list(d.memory_reads_witness[i]) == list(ra[i]) for i in range(nr))
        d2, used2 = dec(data, nq, nr, keep_tails=True)       # the previous tails bincode carries beside the requests
        assert used2 == len(data) and all(list(d2._keep[-1][i]) == [1000 + 4 * i + t for t in range(4)] for i in range(nq))
        with pytest.raises(zkgl.ZkError):
            dec(data, nq, nr - 1)
        with pytest.raises(zkgl.ZkError):
            dec(data[:-3], nq, nr)


def test_sort_decommits_and_code_unpacker_bincode_round_trips():
    rng = np.random.default_rng(82)
    # sort_decommittment_requests
    w = zkgl.SortDecommitsWitness()
    w.start_flag, w.completion_flag = 1, 0
    w.initial_queue_state, w.sorted_queue_initial_state = _rand_q(rng, zkgl.FullQueueStateWitness, 12), _rand_q(rng, zkgl.FullQueueStateWitness, 12)
    fb = []
    for f in (w.hidden_fsm_input, w.hidden_fsm_output):
        f.initial_queue_state, f.sorted_queue_state, f.final_queue_state = [_rand_q(rng, zkgl.FullQueueStateWitness, 12) for _ in range(3)]
        f.lhs_accumulator[:] = [int(rng.integers(0, P_GL, dtype=np.uint64)) for _ in range(2)]
        f.rhs_accumulator[:] = [int(rng.integers(0, P_GL, dtype=np.uint64)) for _ in range(2)]
        f.previous_packed_key[:] = _rand_limbs(rng, 9); f.first_encountered_timestamp = int(rng.integers(0, 1 << 32)); f.previous_record = _rand_dq_struct(rng)
        fb.append(_b_q12(f.initial_queue_state) + _b_q12(f.sorted_queue_state) + _b_q12(f.final_queue_state) +
                  struct.pack("<4Q", *f.lhs_accumulator, *f.rhs_accumulator) + struct.pack("<9I", *f.previous_packed_key) +
                  struct.pack("<I", f.first_encountered_timestamp) + _b_dq(f.previous_record))
    n = 5
    ia = (zkgl.DecommitQueryWitness * n)(*[_rand_dq_struct(rng) for _ in range(n)]); sa = (zkgl.DecommitQueryWitness * n)(*[_rand_dq_struct(rng) for _ in range(n)])
    data = (struct.pack("<BB", 1, 0) + _b_q12(w.initial_queue_state) + _b_q12(w.sorted_queue_initial_state) + _b_q12(_rand_q(rng, zkgl.FullQueueStateWitness, 12)) +
            fb[0] + fb[1] + _b_dq_queue(ia, n) + _b_dq_queue(sa, n))
    d, used = zkgl.decode_sort_decommits_witness_bincode(data, n)
    assert used == len(data) and (d.n_initial, d.n_sorted) == (n, n)
    assert bytes(d.hidden_fsm_input) == bytes(w.hidden_fsm_input) and bytes(d.hidden_fsm_output) == bytes(w.hidden_fsm_output)
    assert bytes(d.initial_queue_state) == bytes(w.initial_queue_state) and bytes(d.sorted_queue_initial_state) == bytes(w.sorted_queue_initial_state)
    assert all(bytes(d.initial_queue_witness[i]) == bytes(ia[i]) and bytes(d.sorted_queue_witness[i]) == bytes(sa[i]) for i in range(n))
    with pytest.raises(zkgl.ZkError):
        zkgl.decode_sort_decommits_witness_bincode(data, n - 1)
    # ..._tails keeps the 12-word previous tail bincode carries beside every element (the inputs of the packers with tails)
    d, used = zkgl.decode_sort_decommits_witness_bincode(data, n, keep_tails=True)
    it, st = d._keep[-2], d._keep[-1]
    assert used == len(data) and all(list(it[i]) == [77 + 12 * i + t for t in range(12)] and list(st[i]) == list(it[i]) for i in range(n))
    assert all(bytes(d.sorted_queue_witness[i]) == bytes(sa[i]) for i in range(n))
    # code_unpacker_sha256: bytes of the start instance of _code_unpacker_packed -> decode -> pack == the oracle's streams
    from oracle import code_unpacker_native as cn
    outer, loop, insts, limit = _code_unpacker_packed()
    o = insts[0]["outer"]
    rows = insts[0]["rows"]
    reqs = [r[74:85] for r in rows if any(r[74:85])]
    words = []
    for r in rows:
        for lo in (85, 93):
            if any(r[lo:lo + 8]):
                words.append(r[lo:lo + 8])
    qstate = lambda x: struct.pack("<24Q", *[int(v) for v in x[0:24]]) + struct.pack("<I", int(x[24]))
    x = o[51:125]
    fsm = (struct.pack("<8I", *[int(v) for v in x[0:8]]) + _b_u256(x[8:16]) + struct.pack("<III", int(x[16]), int(x[17]), int(x[18])) + struct.pack("<H", int(x[19])) +
           struct.pack("<I", int(x[20])) + struct.pack("<BBB", int(x[21]), int(x[22]), int(x[23])) + qstate(x[24:49]) + qstate(x[49:74]))
    code_words = struct.pack("<Q", 2) + struct.pack("<Q", 1) + _b_u256(words[0]) + struct.pack("<Q", len(words) - 1) + b"".join(_b_u256(wd) for wd in words[1:])
    dq_queue = struct.pack("<Q", len(reqs)) + b"".join(_b_u256(q[0:8]) + struct.pack("<IBI", int(q[8]), int(q[9]), int(q[10])) + bytes(96) for q in reqs)
    data = struct.pack("<BB", 1, 0) + qstate(o[26:51]) + qstate(o[1:26]) + qstate([0] * 25) + fsm + fsm + dq_queue + code_words
    d, used = zkgl.decode_code_unpacker_witness_bincode(data, len(reqs), len(words))
    assert used == len(data) and d.n_requests == len(reqs) and d.n_code_words == len(words)
    o2 = np.zeros((125, 2), dtype=np.uint64); l2 = np.full((101, 2 * limit), 9, dtype=np.uint64)
    zkgl.pack_code_unpacker_witness(d, limit, 0, o2, l2)
    assert np.array_equal(o2[:, 0], outer[:, 0]) and np.array_equal(l2[:, :limit], loop[:, :limit])
    with pytest.raises(zkgl.ZkError):
        zkgl.decode_code_unpacker_witness_bincode(data, len(reqs), len(words) - 1)
    d, used = zkgl.decode_code_unpacker_witness_bincode(data, len(reqs), len(words), keep_tails=True)
    assert used == len(data) and all(list(d._keep[-1][i]) == [0] * 12 for i in range(len(reqs)))
    # end to end with the REAL previous tails on the wire: bytes -> decode (tails kept) -> packer with tails == the native stream with its
    # carried words (the memory queue's tails are the RAM permutation's witness; taken from the native restatement here)
    from oracle import zko
    from oracle.decommit_native import encode as dq_encode
    head, prev = [int(v) for v in o[1:13]], []
    for q in reqs:
        prev.append(head)
        head = zko.queue_full_push(head, dq_encode([int(v) for v in q]))
    dq_queue2 = struct.pack("<Q", len(reqs)) + b"".join(_b_u256(q[0:8]) + struct.pack("<IBI", int(q[8]), int(q[9]), int(q[10])) +
                                                        b"".join(struct.pack("<Q", int(t)) for t in prev[i]) for i, q in enumerate(reqs))
    fsm_out = insts[0]["fsm_out"]
    xo = cn.flatten_fsm(fsm_out)
    fsm2 = (struct.pack("<8I", *[int(v) for v in xo[0:8]]) + _b_u256(xo[8:16]) + struct.pack("<III", int(xo[16]), int(xo[17]), int(xo[18])) + struct.pack("<H", int(xo[19])) +
            struct.pack("<I", int(xo[20])) + struct.pack("<BBB", int(xo[21]), int(xo[22]), int(xo[23])) + qstate(xo[24:49]) + qstate(xo[49:74]))
    n_used = len(insts[0]["pushed"])
    cw2 = struct.pack("<Q", 1) + struct.pack("<Q", n_used) + b"".join(_b_u256(wd) for wd in words[:n_used])
    data2 = struct.pack("<BB", 1, 0) + qstate(o[26:51]) + qstate(o[1:26]) + qstate([0] * 25) + fsm + fsm2 + dq_queue2 + cw2
    d, used = zkgl.decode_code_unpacker_witness_bincode(data2, len(reqs), len(words), keep_tails=True)
    assert used == len(data2)
    mt, mtails = [int(v) for v in o[26:51][12:24]], []
    for m in insts[0]["pushed"]:
        mt = zko.queue_full_push(mt, zko.memory_query_encode(m))
        mtails.append(mt)
    o3 = np.zeros((125, 2), dtype=np.uint64); l3 = np.full((101, 2 * limit), 9, dtype=np.uint64)
    zkgl.pack_code_unpacker_witness_tails(d, limit, 0, o3, l3, np.array([list(t) for t in d._keep[-1]], dtype=np.uint64), np.array(mtails, dtype=np.uint64))
    eo, el = _streams(insts)
    assert np.array_equal(o3[:, 0], eo[:, 0]) and np.array_equal(l3[:, :limit], el[:, :limit])


# ---------------------------------------------------------------- a byte vector NOT written by this file's writer
def test_hand_derived_ram_bincode_vector_decodes_to_the_reference_fixture():
    """tests/golden/ram_witness_bincode_vector.hex: the reference's ram_permutation fixture (src/ram_permutation/mod.rs:559-634) as
    bincode bytes derived field by field from the serde derive order (tests/golden/make_ram_bincode_vector.py; every byte group is
    listed with its rule in ram_witness_bincode_vector.md).  The C decoder must read exactly these bytes into the fixture's witness."""
    import json, os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = bytes.fromhex(open(os.path.join(here, "ram_witness_bincode_vector.hex")).read().strip())
    fx = json.load(open(os.path.join(here, "ram_fixture.json")))
    w, used = zkgl.decode_ram_witness_bincode(data, 16)
    assert used == len(data) == 2118
    assert w.start_flag == 1 and w.completion_flag == 0 and w.n_unsorted == 3 and w.n_sorted == 3
    assert w.unsorted_queue_initial_state.length == 3 and not any(w.unsorted_queue_initial_state.head)

    def rec(row):
        d = dict(zip(fx["fields"], row))
        if d["memory_page"] == "BOOTLOADER_HEAP_PAGE":
            d["memory_page"] = rn.BOOTLOADER_HEAP_PAGE
        return d
    for got, rows in ((w.unsorted_queue_witness, fx["unsorted"]), (w.sorted_queue_witness, fx["sorted"])):
        for i, row in enumerate(rows):
            d, m = rec(row), got[i]
            assert (m.timestamp, m.memory_page, m.index, m.rw_flag, m.is_ptr) == (d["timestamp"], d["memory_page"], d["index"], d["rw_flag"], d["is_ptr"])
            assert sum(int(x) << (32 * k) for k, x in enumerate(m.value)) == d["value"]
    # the decoded witness packs into the same streams as the oracle's packer builds from the fixture
    limit = 16
    u = [rn.mq(*[rec(r)[k] for k in fx["fields"]]) for r in fx["unsorted"]]
    s = [rn.mq(*[rec(r)[k] for k in fx["fields"]]) for r in fx["sorted"]]
    inst = rn.instance(u, s, limit, 1)
    outer = np.zeros((zkgl.RAM_OUTER_WORDS, 1), dtype=np.uint64)
    loop = np.zeros((zkgl.RAM_LOOP_WORDS, limit), dtype=np.uint64)
    w.hidden_fsm_input.num_nondeterministic_writes = 0
    zkgl.pack_ram_witness(w, limit, 0, outer, loop)
    eo, el = rn.pack_streams([inst], limit)
    el = el.copy(); el[0:46] = 0
    assert np.array_equal(loop, el)
    tail_words = slice(12, 24)     # unsorted_queue_initial_state.tail inside the outer stream: the chain of the fixture's three pushes
    assert np.array_equal(outer[tail_words, 0], eo[tail_words, 0])
