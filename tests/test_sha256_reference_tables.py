"""the SHA circuits under the REFERENCE's own table set (/root/reference/src/code_unpacker_sha256/mod.rs:484-566:
lookup width 4 x 8 repetitions; Maj4Table, TriXor4Table, Ch4Table, Split4BitChunkTable<1>, Split4BitChunkTable<2>; no 8-bit table).
zk_circuit_sha256_configure_reference_tables + the same entry points record the 4-bit-chunk compression
(csrc/circuits/sha256_gadget4.hpp); everything observable — digests, memory-queue tails, commitments, input streams — must equal the
8-bit decomposition's and the software hash, and the code_unpacker reference fixture (the crate's only SHA-256 known answer) must pass."""
import hashlib

import numpy as np
import pytest

import zkgl
from oracle import code_unpacker_native as N
from oracle import zko
from oracle.ram_native import mq
from test_code_unpacker_host import load_code_unpacker_fixture, streams
from test_sha256_host import loop_stream

REF_TABLE_ROWS = 3 * 4096 + 2 * 16
_CS = {}


def sha_cs4(n_blocks):
    key = ("blocks", n_blocks)
    if key not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_sha256(reference_tables=True)
        cs.sha256_blocks_entry_point(n_blocks)
        cs.pad_and_shrink()
        _CS[key] = cs
    return _CS[key]


def unpacker_cs4(limit):
    key = ("unpacker", limit)
    if key not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))      # the geometry of the reference test (mod.rs:470-476)
        cs.configure_sha256(reference_tables=True)
        cs.unpack_code_into_memory_entry_point(limit)
        cs.pad_and_shrink()
        _CS[key] = cs
    return _CS[key]


def test_the_table_set_is_the_reference_one():
    cs = sha_cs4(1)
    ex = zko.parse_export(cs.export(False))
    shapes = sorted((t["n_keys"], t["n_vals"], t["n_rows"]) for t in ex["tables"] if t["n_rows"])
    assert shapes == [(1, 3, 16), (1, 3, 16), (3, 1, 4096), (3, 1, 4096), (3, 1, 4096)]
    assert cs.stats()["lookup_columns"] == 4 * 8


@pytest.mark.parametrize("lengths,n_blocks", [((0, 3, 55), 1), ((56, 64, 119), 2)])
def test_digest_equals_hashlib_with_the_reference_tables(lengths, n_blocks):
    rng = np.random.default_rng(sum(lengths) + 7)
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in lengths]
    if 3 in lengths:
        msgs[lengths.index(3)] = b"abc"
    cs = sha_cs4(n_blocks)
    outer = np.zeros((0, len(msgs)), dtype=np.uint64)
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS).seed(outer, loop_stream(msgs, n_blocks))
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS)
    run.resolve(outer, seeded)
    bad, nrel = run.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * len(msgs)
    for i, m in enumerate(msgs):
        assert bytes(int(run.oc[c, i]) for c in cs.public_cells()) == hashlib.sha256(m).digest()
    # negatives: a message word that is not a byte; a state byte that differs from the IV
    for word, value in ((40, 300), (0, int(seeded[0, 0]) ^ 1)):
        b = seeded.copy(); b[word, 0] = value
        r = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS); r.resolve(outer, b)
        assert r.check()[0] > 0


def test_rows_per_compression_of_both_decompositions():
    from test_sha256_host import sha_cs
    s8, s4 = sha_cs(1).stats(), sha_cs4(1).stats()
    rows8, rows4 = s8["loop_slots"], s4["loop_slots"]
    print(f"rows per SHA-256 compression: 8-bit engine tables {rows8} (lookups {s8['lookups_per_instance']}), "
          f"reference width-4 tables {rows4} (lookups {s4['lookups_per_instance']})")
    assert 600 < rows8 < 1000 and 700 < rows4 < 1400


def test_code_unpacker_reference_fixture_with_the_reference_tables():
    req, words, limit = load_code_unpacker_fixture()
    cs = unpacker_cs4(limit)
    assert cs.input_words() == (N.OUTER_WORDS, N.LOOP_WORDS)          # same streams as with the 8-bit tables
    inst = N.instance([(req, words)], limit)
    tail = [0] * 12                                                     # compute_memory_queue_state (mod.rs:640-660)
    for i, w in enumerate(words):
        tail = zko.queue_full_push(tail, zko.memory_query_encode(mq(40973, 2368, i, 1, 0, w)))
    assert inst["memory_state"][12:24] == tail and inst["memory_state"][24] == 33
    outer, loop = streams([inst], limit)
    blank = loop.copy()
    blank[:N.CARRIED] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 1, REF_TABLE_ROWS).seed(outer, blank)
    assert np.array_equal(seeded, loop)
    r = zko.CircuitRun(cs.export(False), cs.export(True), 1, REF_TABLE_ROWS)
    r.resolve(outer, loop)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]
    # a corrupted bytecode word no longer hashes to the versioned hash
    bad_inst = N.instance([(req, [words[0] ^ 1] + words[1:])], limit)
    bo, bl = streams([bad_inst], limit)
    r = zko.CircuitRun(cs.export(False), cs.export(True), 1, REF_TABLE_ROWS)
    r.resolve(bo, bl)
    assert r.check()[0] > 0


def gpu_equals_oracle_with_the_reference_tables(zk):
    """the device's trace of both circuits == the oracle interpreter's, cell for cell; fused and stored verdicts.
    Run as a test from tests/test_zz_round5_gpu.py (round 6): since the macro-op became the default recording of this table set, this is a device path no
    device has run, and such tests sort last so that `pytest -x` cannot hide the established parity evidence behind them."""
    msgs = [b"abc", b"", bytes(range(55))]
    cs = sha_cs4(1)
    outer = np.zeros((0, len(msgs)), dtype=np.uint64)
    loop = loop_stream(msgs, 1)
    cs.set_batch(len(msgs))
    d_o, d_l = zk.DeviceBuffer(1), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, 0)
    cs.bind_inputs(True, d_l, loop.shape[0])
    cs.seed_carried_inputs(d_l)
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS).seed(outer, loop)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), seeded)
    for stored in (False, True):
        cs.set_check_mode(stored)
        ok, f = cs.resolve_and_check()
        assert ok, f
    cs.set_check_mode(False)
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), REF_TABLE_ROWS)
    run.resolve(outer, seeded)
    assert np.array_equal(cs.trace(False), run.oc) and np.array_equal(cs.trace(True), run.lc)
    for i, m in enumerate(msgs):
        assert bytes(cs.public_inputs(i)) == hashlib.sha256(m).digest()
    # code_unpacker: the reference fixture
    req, words, limit = load_code_unpacker_fixture()
    ucs = unpacker_cs4(limit)
    inst = N.instance([(req, words)], limit)
    uo, ul = streams([inst], limit)
    ucs.set_batch(1)
    blank = ul.copy(); blank[:N.CARRIED] = 0
    d_uo, d_ul = zk.DeviceBuffer.from_numpy(uo), zk.DeviceBuffer.from_numpy(blank)
    ucs.bind_inputs(False, d_uo, uo.shape[0])
    ucs.bind_inputs(True, d_ul, ul.shape[0])
    ucs.seed_carried_inputs(d_ul)
    assert np.array_equal(d_ul.to_numpy().reshape(ul.shape), ul)
    ok, f = ucs.resolve_and_check()
    assert ok, f
    assert ucs.public_inputs(0) == inst["public_input"]
    r = zko.CircuitRun(ucs.export(False), ucs.export(True), 1, REF_TABLE_ROWS)
    r.resolve(uo, ul)
    assert np.array_equal(ucs.trace(False), r.oc) and np.array_equal(ucs.trace(True), r.lc)


def test_sha256_precompile_fsm_with_the_reference_tables():
    """sha256_round_function_entry_point under the reference table set: same streams, same public input, digests == hashlib"""
    from oracle import sha256_native as SN
    from test_sha256_fsm_host import make_requests, messages
    from test_sha256_fsm_host import streams as fsm_streams
    limit = 5
    msgs = messages(np.random.default_rng(128), (0, 55, 56))
    inst = SN.instance(make_requests(msgs), limit)
    assert inst["satisfiable"] and inst["fsm_out"]["completed"] == 1
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_sha256(reference_tables=True)
    cs.sha256_round_function_entry_point(limit)
    cs.pad_and_shrink()
    assert cs.input_words() == (SN.OUTER_WORDS, SN.LOOP_WORDS)
    outer, loop = fsm_streams([inst], limit)
    blank = loop.copy()
    blank[:SN.CARRIED, :] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 1, REF_TABLE_ROWS).seed(outer, blank)
    assert np.array_equal(seeded, loop)
    r = zko.CircuitRun(cs.export(False), cs.export(True), 1, REF_TABLE_ROWS)
    r.resolve(outer, loop)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]
