"""8(f)-4: linear_hasher_entry_point (/root/reference/src/linear_hasher/mod.rs:35-212) recorded through the C-ABI and executed on
the CPU oracle interpreter: the digest equals keccak256 of the concatenated 88-byte serialisations (software Keccak) for queues
that end inside / at the end of / across blocks, the empty queue yields keccak256(""), an unfinished queue is rejected."""
import numpy as np
import pytest

import zkgl
from oracle import linear_hasher_native as N
from oracle import zko
from oracle.storage_native import log_query

TABLE_ROWS = 65536 * 2 + 7 * 256
_CS = {}


def hasher_cs(limit):
    if limit not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4), 1 << 22, 1 << 28)
        cs.configure_linear_hasher()
        cs.linear_hasher_entry_point(limit)
        cs.pad_and_shrink()
        _CS[limit] = cs
    return _CS[limit]


def random_messages(rng, n):
    return [log_query(address=int(rng.integers(1, 1 << 60)), key=int.from_bytes(rng.bytes(32), "little"),
                      written_value=int.from_bytes(rng.bytes(32), "little"), rw_flag=1, aux_byte=2, is_service=int(rng.integers(0, 2)),
                      shard_id=int(rng.integers(0, 2)), tx_number_in_block=int(rng.integers(0, 65536)), timestamp=5 + t) for t in range(n)]


def streams(insts):
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    return outer, loop


def test_limit_must_be_a_multiple_of_17():
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_linear_hasher()
    with pytest.raises(zkgl.ZkError):
        cs.linear_hasher_entry_point(16)


def test_digest_equals_software_keccak():
    cs = hasher_cs(17)
    assert cs.input_words() == (N.OUTER_WORDS, N.LOOP_WORDS)
    rng = np.random.default_rng(88)
    insts = []
    for n in (0, 1, 3, 17):          # empty; ends inside a block; after the 2nd absorb; full period (ends exactly at a block end)
        qs = random_messages(rng, n)
        inst = N.instance(qs, 17)
        assert inst["satisfiable"] and inst["digest"] == zko.keccak256(b"".join(N.into_bytes(q) for q in qs))
        insts.append(inst)
    outer, loop = streams(insts)
    blank = loop.copy()
    blank[:N.CARRIED] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS).seed(outer, blank)
    assert np.array_equal(seeded, loop)
    r = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS)
    r.resolve(outer, loop)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * len(insts)
    for i, inst in enumerate(insts):
        assert [int(r.oc[c, i]) for c in cs.public_cells()] == inst["public_input"]


def test_two_periods_and_unfinished_queue():
    cs = hasher_cs(34)
    rng = np.random.default_rng(89)
    good = N.instance(random_messages(rng, 20), 34)      # crosses into the second period
    assert good["satisfiable"]
    too_long = N.instance(random_messages(rng, 36), 34)  # queue not exhausted within `limit`: completion is enforced (mod.rs:172-173)
    assert not too_long["satisfiable"]
    outer, loop = streams([good, too_long])
    r = zko.CircuitRun(cs.export(False), cs.export(True), 2, TABLE_ROWS)
    r.resolve(outer, loop)
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == good["public_input"]
    r1 = zko.CircuitRun(cs.export(False), cs.export(True), 1, TABLE_ROWS)
    r1.resolve(outer[:, :1], loop[:, :2])
    assert r1.check()[0] == 0
    r2 = zko.CircuitRun(cs.export(False), cs.export(True), 1, TABLE_ROWS)
    r2.resolve(outer[:, 1:], loop[:, 2:])
    assert r2.check()[0] > 0
