"""log_sorter (config C4) on the CPU oracle interpreter: reference fixture
(/root/reference/src/log_sorter/mod.rs:493-636, 638-815) + rollback-collapse positives / negatives."""
import json
import os

import numpy as np
import pytest

import zkgl
from helpers import GOLD, oracle_run
from oracle import log_sorter_native as ln
from oracle import zko

_CS = {}


def log_sorter_cs(limit):
    if limit not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_log_sorter()
        cs.sort_and_deduplicate_events_entry_point(limit)
        cs.pad_and_shrink()
        _CS[limit] = cs
    return _CS[limit]


def load_log_sorter_fixture():
    f = json.load(open(os.path.join(GOLD, "log_sorter_fixture.json")))
    conv = lambda d: ln.log_query(**{k: int(v) for k, v in d.items()})
    return [conv(d) for d in f["unsorted"]], [conv(d) for d in f["sorted"]], f["limit"]


def run(cs, insts, limit):
    outer, loop = ln.pack_streams(insts, limit)
    return oracle_run(cs, outer, loop, len(insts)), outer, loop


def test_reference_fixture():
    u, s, limit = load_log_sorter_fixture()
    inst = ln.instance(u, s, limit)
    assert inst["satisfiable"] and inst["completed"] and inst["permutation_ok"]
    assert len(inst["result_items"]) == 4   # no rollbacks in the fixture: every event survives
    cs = log_sorter_cs(limit)
    assert cs.input_words() == (87, 129)
    r, _, _ = run(cs, [inst], limit)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["commitment"]
    st = cs.stats()   # 9 permutations per item (2 pops + 1 push, 3 rounds each) + 35 per instance
    assert st["gate_instances"]["MATMUL12_EXT"] // 9 == 9 * limit + 6 + 3 + 26


@pytest.mark.parametrize("seed,n_events,limit", [(1, 4, 8), (2, 6, 16), (3, 9, 16)])
def test_rollbacks_are_collapsed(seed, n_events, limit):
    rng = np.random.default_rng(seed)
    u, s = ln.random_events(rng, n_events, rollback_frac=0.5)
    if len(u) > limit:
        u, s = ln.random_events(rng, n_events, rollback_frac=0.0)
    inst = ln.instance(u, s, limit)
    assert inst["satisfiable"] and inst["completed"]
    n_rolled = sum(q[31] for q in u)
    assert len(inst["result_items"]) == n_events - n_rolled
    cs = log_sorter_cs(limit)
    r, outer, loop = run(cs, [inst], limit)
    assert r.check()[0] == 0
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["commitment"]
    raw = loop.copy(); raw[0:57] = 0
    assert np.array_equal(zko.CircuitRun(cs.export(False), cs.export(True), 1, 65536).seed(outer, raw), loop)


def test_negative_cases():
    limit = 8
    cs = log_sorter_cs(limit)
    rng = np.random.default_rng(5)
    u, s = ln.random_events(rng, 4, rollback_frac=0.0)
    # (a) a read (rw = 0) in the events queue
    u_bad = [list(q) for q in u]; u_bad[1][30] = 0
    s_bad = [list(q) for q in s]; s_bad[1][30] = 0
    inst = ln.instance(u_bad, s_bad, limit)
    assert not inst["satisfiable"] and run(cs, [inst], limit)[0].check()[0] > 0
    # (b) not sorted by timestamp
    inst = ln.instance(u, [s[1], s[0], s[2], s[3]], limit)
    assert not inst["satisfiable"] and run(cs, [inst], limit)[0].check()[0] > 0
    # (c) a lone rollback without its forward twin
    lone = [ln.log_query(address=7, key=9, written_value=5, rw_flag=1, rollback=1, timestamp=50)]
    inst = ln.instance(lone, lone, limit)
    assert not inst["satisfiable"] and run(cs, [inst], limit)[0].check()[0] > 0
    # (d) sorted side drops an element (not a permutation)
    inst = ln.instance(u, s[:3] + [s[2]], limit)
    assert not inst["satisfiable"] and run(cs, [inst], limit)[0].check()[0] > 0
