"""8(f)-4: unpack_code_into_memory_entry_point (/root/reference/src/code_unpacker_sha256/mod.rs:33-442) recorded through the
C-ABI and executed on the CPU oracle interpreter.  The reference fixture (mod.rs:618-718: 33 bytecode words whose SHA-256
must match the versioned code hash; limit 40) carries the only SHA-256 known answer of the crate: the circuit accepts it and
the final memory-queue tail equals the one recomputed from 33 plain pushes, which is what the reference test asserts
(:594-606).  Plus several requests, continuation and negatives."""
import json
import os

import numpy as np
import pytest

import zkgl
from helpers import GOLD
from oracle import code_unpacker_native as N
from oracle import zko
from oracle.decommit_native import dq
from oracle.ram_native import mq

TABLE_ROWS = 65536 * 3 + 7 * 256
_CS = {}


def unpacker_cs(limit):
    if limit not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_code_unpacker()
        cs.unpack_code_into_memory_entry_point(limit)
        cs.pad_and_shrink()
        _CS[limit] = cs
    return _CS[limit]


def load_code_unpacker_fixture():
    f = json.load(open(os.path.join(GOLD, "code_unpacker_fixture.json")))
    words = [int(w) for w in f["code_words"]]
    return dq(int(f["code_hash"]), f["page"], 1, f["timestamp"]), words, f["limit"]


def streams(insts, limit):
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    assert outer.shape == (N.OUTER_WORDS, len(insts)) and loop.shape == (N.LOOP_WORDS, len(insts) * limit)
    return outer, loop


def run(cs, outer, loop, batch):
    r = zko.CircuitRun(cs.export(False), cs.export(True), batch, TABLE_ROWS)
    r.resolve(outer, loop)
    return r


def random_code(rng, n_words):
    return [int.from_bytes(rng.bytes(32), "big") for _ in range(n_words)]


def test_reference_fixture_sha256_known_answer():
    req, words, limit = load_code_unpacker_fixture()
    assert N.versioned_hash(words) == sum(l << (32 * i) for i, l in enumerate(req[0:8]))     # the fixture's hash IS sha256 of its bytecode
    cs = unpacker_cs(limit)
    assert cs.input_words() == (N.OUTER_WORDS, N.LOOP_WORDS)
    inst = N.instance([(req, words)], limit)
    assert inst["satisfiable"] and inst["fsm_out"]["finished"] == 1
    tail = [0] * 12                                   # compute_memory_queue_state (mod.rs:640-660)
    for i, w in enumerate(words):
        tail = zko.queue_full_push(tail, zko.memory_query_encode(mq(40973, 2368, i, 1, 0, w)))
    assert inst["memory_state"][12:24] == tail and inst["memory_state"][24] == 33
    outer, loop = streams([inst], limit)
    blank = loop.copy()
    blank[:N.CARRIED] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 1, TABLE_ROWS).seed(outer, blank)
    assert np.array_equal(seeded, loop)
    r = run(cs, outer, loop, 1)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]


def test_several_requests_and_continuation():
    rng = np.random.default_rng(44)
    reqs = []
    for k, n in enumerate((1, 5, 3)):
        words = random_code(rng, n)
        reqs.append((dq(N.versioned_hash(words), 2000 + 8 * k, 1, 100 + k), words))
    whole = N.instance(reqs, 8)          # 1 + 3 + 2 rounds
    assert whole["satisfiable"] and whole["fsm_out"]["finished"] == 1 and len(whole["pushed"]) == 9
    a = N.instance(reqs, 4)
    b = N.instance(a["rest"][0], 4, start_flag=False, fsm_in=a["fsm_out"], obs=a["obs"], pending=a["rest"][1])
    assert a["fsm_out"]["finished"] == 0 and b["fsm_out"]["finished"] == 1 and b["memory_state"] == whole["memory_state"]
    cs = unpacker_cs(4)
    outer, loop = streams([a, b], 4)
    r = run(cs, outer, loop, 2)
    assert r.check()[0] == 0
    for i, inst in enumerate((a, b)):
        assert [int(r.oc[c, i]) for c in cs.public_cells()] == inst["public_input"]


@pytest.mark.parametrize("kind", ["wrong_hash", "wrong_version", "even_length", "code_word"])
def test_negative(kind):
    rng = np.random.default_rng(45)
    words = random_code(rng, 3)
    h = N.versioned_hash(words)
    if kind == "wrong_hash":
        h ^= 1
    elif kind == "wrong_version":
        h ^= 1 << 248
    elif kind == "even_length":
        words = random_code(rng, 4)
        h = N.versioned_hash(words)
    inst = N.instance([(dq(h, 2048, 1, 7), words)], 4)
    cs = unpacker_cs(4)
    outer, loop = streams([inst], 4)
    if kind == "code_word":
        loop[N.CARRIED + 11, 1] ^= 1     # a bytecode word differs from the one the digest / queue chain was computed with
    else:
        assert not inst["satisfiable"]
    assert run(cs, outer, loop, 1).check()[0] > 0
