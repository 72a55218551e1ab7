"""8(f)-1: sort_and_deduplicate_code_decommittments_entry_point (/root/reference/src/sort_decommittment_requests/mod.rs:40-372)
recorded through the C-ABI and executed on the CPU oracle interpreter: the reference fixture (mod.rs:565-1390: 29 + 29 queries,
limit 16 — a partial pass, like its test :485-563) is accepted; random request logs deduplicate to one record per code hash
with the first timestamp; continuation; order / is_first / page / permutation violations rejected."""
import json
import os

import numpy as np
import pytest

import zkgl
from helpers import GOLD
from oracle import decommit_native as N
from oracle import zko

_CS = {}


def decommit_cs(limit):
    if limit not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_sort_decommits()
        cs.sort_and_deduplicate_code_decommittments_entry_point(limit)
        cs.pad_and_shrink()
        _CS[limit] = cs
    return _CS[limit]


def load_decommit_fixture():
    f = json.load(open(os.path.join(GOLD, "decommit_fixture.json")))
    conv = lambda lst: [N.dq(int(d["code_hash"]), int(d["page"]), int(d["is_first"]), int(d["timestamp"])) for d in lst]
    return conv(f["unsorted"]), conv(f["sorted"]), f["limit"]


def streams(insts, limit):
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    assert outer.shape == (N.OUTER_WORDS, len(insts)) and loop.shape == (N.LOOP_WORDS, len(insts) * limit)
    return outer, loop


def run(cs, outer, loop, batch):
    r = zko.CircuitRun(cs.export(False), cs.export(True), batch, 65536)
    r.resolve(outer, loop)
    return r


def test_layout_and_reference_fixture():
    u, s, limit = load_decommit_fixture()
    cs = decommit_cs(limit)
    assert cs.input_words() == (N.OUTER_WORDS, N.LOOP_WORDS)
    inst = N.instance(u, s, limit)
    assert inst["satisfiable"] and not inst["completed"]
    outer, loop = streams([inst], limit)
    blank = loop.copy()
    blank[:N.CARRIED] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 1, 65536).seed(outer, blank)
    assert np.array_equal(seeded, loop)
    r = run(cs, outer, loop, 1)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]


def test_deduplication_and_continuation():
    rng = np.random.default_rng(40)
    u, s = N.random_decommits(rng, 7)
    assert 7 <= len(u) <= 24
    whole = N.instance(u, s, 24)
    assert whole["satisfiable"] and whole["completed"]
    firsts = {}
    for q in u:
        firsts.setdefault(tuple(q[0:8]), q[10])
    assert [(tuple(q[0:8]), q[10]) for q in whole["result"]] == sorted(firsts.items(), key=lambda kv: sum(l << (32 * i) for i, l in enumerate(kv[0])))
    a = N.instance(u, s, 8)
    b = N.instance(*a["rest"], 8, start_flag=False, fsm_in=a["fsm_out"], obs=a["obs"])
    c = N.instance(*b["rest"], 8, start_flag=False, fsm_in=b["fsm_out"], obs=a["obs"])
    assert c["completed"] and c["satisfiable"] and c["fsm_out"]["final"] == whole["fsm_out"]["final"]
    cs = decommit_cs(8)
    outer, loop = streams([a, b, c], 8)
    r = run(cs, outer, loop, 3)
    assert r.check()[0] == 0
    for i, inst in enumerate((a, b, c)):
        assert [int(r.oc[cc, i]) for cc in cs.public_cells()] == inst["public_input"]


@pytest.mark.parametrize("kind", ["order", "is_first", "page", "permutation"])
def test_negative(kind):
    rng = np.random.default_rng(41)
    u, s = N.random_decommits(rng, 3, max_repeats=3)
    s = [list(q) for q in s]
    j = next(k for k in range(1, len(s)) if s[k][0:8] == s[k - 1][0:8])     # second request of some hash
    if kind == "order":
        s[j - 1], s[j] = s[j], s[j - 1]
    elif kind == "is_first":
        i = next(k for k in range(len(s)) if s[k][9] == 1 and k > 0)
        s[i][9] = 0
    elif kind == "page":
        s[j][8] += 8
    else:
        s[-1][10] += 1   # same order, but no longer the multiset of the original queue
    inst = N.instance(u, s, 16)
    assert not inst["satisfiable"]
    cs = decommit_cs(16)
    outer, loop = streams([inst], 16)
    assert run(cs, outer, loop, 1).check()[0] > 0
