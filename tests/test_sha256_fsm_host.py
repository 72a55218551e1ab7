"""a18: sha256_round_function_entry_point (the precompile FSM, /root/reference/src/sha256_round_function/mod.rs:88-468)
recorded through the C-ABI and executed on the CPU oracle interpreter.  The reference holds no fixture for this
circuit; the checks are the ones its sibling keccak test makes (keccak256_round_function/mod.rs:1000-1094): the
last memory-queue item is a write of the software digest, and the assembly is satisfied — plus equality with the
native restatement (oracle/sha256_native.py) cycle by cycle and on the public input."""
import hashlib

import numpy as np
import pytest

import zkgl
from oracle import sha256_native as N
from oracle import zko

TABLE_ROWS = 65536 * 3 + 7 * 256
_CS = {}


def fsm_cs(limit):
    if limit not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_sha256()
        cs.sha256_round_function_entry_point(limit)
        cs.pad_and_shrink()
        _CS[limit] = cs
    return _CS[limit]


def streams(instances, limit):
    outer = np.array([i["outer"] for i in instances], dtype=np.uint64).T.copy()
    loop = np.array([r for i in instances for r in i["rows"]], dtype=np.uint64).T.copy()
    assert outer.shape == (N.OUTER_WORDS, len(instances)) and loop.shape == (N.LOOP_WORDS, len(instances) * limit)
    return outer, loop


def run(cs, outer, loop, batch):
    r = zko.CircuitRun(cs.export(False), cs.export(True), batch, TABLE_ROWS)
    r.resolve(outer, loop)
    return r


def messages(rng, lengths):
    return [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in lengths]


def make_requests(msgs):
    return [N.request(m, timestamp=10 + 7 * i, input_page=100 + i, input_offset=5 * i, output_page=200 + i, output_offset=3 + i)
            for i, m in enumerate(msgs)]


def test_layout():
    cs = fsm_cs(4)
    assert cs.input_words() == (N.OUTER_WORDS, N.LOOP_WORDS)


@pytest.mark.parametrize("lengths,limit", [((3,), 2), ((0, 55, 56), 5), ((150, 64), 6)])
def test_fsm_writes_sha256_digests(lengths, limit):
    rng = np.random.default_rng(sum(lengths) + 17)
    msgs = messages(rng, lengths)
    if lengths == (3,):
        msgs = [b"abc"]
    inst = N.instance(make_requests(msgs), limit)
    assert inst["satisfiable"] and inst["fsm_out"]["completed"] == 1
    # native model: one write per request carrying the software digest, big-endian
    writes = [q for q in inst["pushed"] if q[3] == 1]
    assert len(writes) == len(msgs)
    for q, m in zip(writes, msgs):
        assert sum(l << (32 * i) for i, l in enumerate(q[5:13])).to_bytes(32, "big") == hashlib.sha256(m).digest()
    cs = fsm_cs(limit)
    outer, loop = streams([inst], limit)
    # generic seeding of the carried words from the non-carried stream reproduces the native FSM trajectory
    blank = loop.copy()
    blank[:N.CARRIED, :] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 1, TABLE_ROWS).seed(outer, blank)
    assert np.array_equal(seeded, loop)
    r = run(cs, outer, loop, 1)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]


def test_fsm_continuation_across_instances():
    """two chained instances (limit 3 each) == one long instance: same final memory queue state"""
    rng = np.random.default_rng(5)
    msgs = messages(rng, (100, 10, 70))  # 2 + 1 + 2 rounds
    whole = N.instance(make_requests(msgs), 6)
    a = N.instance(make_requests(msgs), 3)
    assert a["fsm_out"]["completed"] == 0
    b = N.instance(a["rest"][0], 3, start_flag=False, fsm_in=a["fsm_out"], obs_req=a["obs_req"], obs_mem=a["obs_mem"],
                   pending=a["rest"][1])
    assert a["satisfiable"] and b["satisfiable"] and b["fsm_out"]["completed"] == 1
    assert b["memory_state"] == whole["memory_state"]
    cs = fsm_cs(3)
    outer, loop = streams([a, b], 3)
    r = run(cs, outer, loop, 2)
    assert r.check()[0] == 0
    for i, inst in enumerate((a, b)):
        assert [int(r.oc[c, i]) for c in cs.public_cells()] == inst["public_input"]


def test_fsm_empty_queue_finishes_immediately():
    inst = N.instance([], 2)
    assert inst["satisfiable"] and inst["fsm_out"]["completed"] == 1 and inst["pushed"] == []
    cs = fsm_cs(2)
    outer, loop = streams([inst], 2)
    r = run(cs, outer, loop, 1)
    assert r.check()[0] == 0
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]


@pytest.mark.parametrize("kind", ["address", "aux_byte", "read_value", "digest_state"])
def test_fsm_negative(kind):
    msgs = [b"abc"]
    if kind == "address":
        reqs = [N.request(msgs[0], 1, 2, 3, 4, 5, address=0x8010)]
    elif kind == "aux_byte":
        reqs = [N.request(msgs[0], 1, 2, 3, 4, 5, aux_byte=0)]
    else:
        reqs = [N.request(msgs[0], 1, 2, 3, 4, 5)]
    inst = N.instance(reqs, 2)
    cs = fsm_cs(2)
    outer, loop = streams([inst], 2)
    if kind in ("address", "aux_byte"):
        assert not inst["satisfiable"]
    elif kind == "read_value":
        loop[96, 0] ^= 1      # the memory value read differs from what the carried chain was computed with
    else:
        loop[3, 1] ^= 1       # carried SHA state of cycle 1 differs from the output of cycle 0
    r = run(cs, outer, loop, 1)
    assert r.check()[0] > 0
