"""8(f)-1: demultiplex_storage_logs_enty_point (/root/reference/src/demux_log_queue/mod.rs:38-396) recorded through the C-ABI
and executed on the CPU oracle interpreter: the reference's 16-query fixture (mod.rs:595-923, all rollup-storage reads) is
accepted like its test asserts (:563-592); random mixes of all six classes route to the right queues (queue tails equal the
native restatement through the public input); continuation; porter-shard / unknown aux byte rejected."""
import json
import os

import numpy as np
import pytest

import zkgl
from helpers import GOLD
from oracle import demux_native as N
from oracle import zko
from oracle.storage_native import log_query

_CS = {}


def demux_cs(limit):
    if limit not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_demux_log_queue()
        cs.demultiplex_storage_logs_entry_point(limit)
        cs.pad_and_shrink()
        _CS[limit] = cs
    return _CS[limit]


def load_demux_fixture():
    f = json.load(open(os.path.join(GOLD, "demux_fixture.json")))
    qs = [log_query(**{k: int(v) for k, v in d.items()}) for d in f["unsorted"]]
    return qs, f["limit"]


def streams(insts, limit):
    outer = np.array([i["outer"] for i in insts], dtype=np.uint64).T.copy()
    loop = np.array([r for i in insts for r in i["rows"]], dtype=np.uint64).T.copy()
    assert outer.shape == (N.OUTER_WORDS, len(insts)) and loop.shape == (N.LOOP_WORDS, len(insts) * limit)
    return outer, loop


def run(cs, outer, loop, batch):
    r = zko.CircuitRun(cs.export(False), cs.export(True), batch, 65536)
    r.resolve(outer, loop)
    return r


def random_queries(rng, n):
    qs = []
    for t in range(n):
        kind = int(rng.integers(0, 6))
        aux = [0, 1, 2, 3, 3, 3][kind]
        address = {3: 0x8010, 4: 0x02, 5: 0x01}.get(kind, int(rng.integers(1 << 20, 1 << 40)))
        qs.append(log_query(address=address, key=int.from_bytes(rng.bytes(32), "little"), read_value=int.from_bytes(rng.bytes(32), "little"),
                            written_value=int.from_bytes(rng.bytes(32), "little"), rw_flag=int(rng.integers(0, 2)), aux_byte=aux,
                            rollback=int(rng.integers(0, 2)), is_service=int(rng.integers(0, 2)), shard_id=0,
                            tx_number_in_block=int(rng.integers(0, 1000)), timestamp=100 + t))
    return qs


def test_layout_and_reference_fixture():
    qs, limit = load_demux_fixture()
    cs = demux_cs(limit)
    assert cs.input_words() == (N.OUTER_WORDS, N.LOOP_WORDS)
    inst = N.instance(qs, limit)
    assert inst["satisfiable"] and inst["completed"] and [len(r) for r in inst["routed"]] == [16, 0, 0, 0, 0, 0]
    outer, loop = streams([inst], limit)
    blank = loop.copy()
    blank[:N.CARRIED] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 1, 65536).seed(outer, blank)
    assert np.array_equal(seeded, loop)
    r = run(cs, outer, loop, 1)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]


def test_all_six_classes_and_continuation():
    rng = np.random.default_rng(6)
    qs = random_queries(rng, 21)
    whole = N.instance(qs, 24)
    assert whole["satisfiable"] and whole["completed"] and all(len(r) > 0 for r in whole["routed"])
    a = N.instance(qs, 8)
    b = N.instance(a["rest"], 8, start_flag=False, fsm_in=a["fsm_out"], obs_initial=a["obs_initial"])
    c = N.instance(b["rest"], 8, start_flag=False, fsm_in=b["fsm_out"], obs_initial=a["obs_initial"])
    assert (a["completed"], b["completed"], c["completed"]) == (0, 0, 1) and c["fsm_out"]["out"] == whole["fsm_out"]["out"]
    cs = demux_cs(8)
    outer, loop = streams([a, b, c], 8)
    r = run(cs, outer, loop, 3)
    assert r.check()[0] == 0
    for i, inst in enumerate((a, b, c)):
        assert [int(r.oc[cc, i]) for cc in cs.public_cells()] == inst["public_input"]


@pytest.mark.parametrize("kind", ["porter_shard", "aux_byte", "tail"])
def test_negative(kind):
    rng = np.random.default_rng(2)
    qs = random_queries(rng, 5)
    if kind == "porter_shard":
        qs[2] = log_query(address=77, key=1, aux_byte=0, shard_id=1, timestamp=9)
    elif kind == "aux_byte":
        qs[2] = log_query(address=77, key=1, aux_byte=7, timestamp=9)
    inst = N.instance(qs, 8)
    cs = demux_cs(8)
    outer, loop = streams([inst], 8)
    if kind == "tail":
        loop[5 + 2, 3] ^= 1   # carried storage-queue tail of cycle 3 differs from cycle 2's output
    else:
        assert not inst["satisfiable"]
    assert run(cs, outer, loop, 1).check()[0] > 0
