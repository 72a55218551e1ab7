"""storage_validity_by_grand_product (config C4) on the CPU oracle interpreter: the reference's own fixture
(pattern A, /root/reference/src/storage_validity_by_grand_product/mod.rs:1034-1135) plus the
permutation-positive / -negative entry-point tests SURVEY.md Appendix D asks for."""
import numpy as np
import pytest

from helpers import load_storage_fixture, oracle_run, storage_cs
from oracle import storage_native as sn
from oracle import zko


def run(cs, insts, limit):
    outer, loop = sn.pack_streams(insts, limit)
    r = oracle_run(cs, outer, loop, len(insts))
    return r, outer, loop


def test_log_query_encoding_restatement():
    rng = np.random.default_rng(3)
    q = sn.log_query(address=int.from_bytes(rng.bytes(20), "little"), key=int.from_bytes(rng.bytes(32), "little"),
                     read_value=int.from_bytes(rng.bytes(32), "little"), written_value=int.from_bytes(rng.bytes(32), "little"),
                     rw_flag=1, aux_byte=0xAB, rollback=1, is_service=1, shard_id=0xCD, tx_number_in_block=0x11223344, timestamp=0x55667788)
    e = sn.encode(q)
    f = sn.fields(q)
    # spot checks against src/base_structures/log_query/mod.rs:131-150 (v0), :477-495 (v17), :497-510 (v18, v19)
    kb0 = [(f["key"][0] >> (8 * i)) & 0xFF for i in range(4)]
    assert e[0] == f["read"][0] + (kb0[0] << 32) + (kb0[1] << 40) + (kb0[2] << 48)
    ab4 = [(f["address"][4] >> (8 * i)) & 0xFF for i in range(4)]
    assert e[17] == 0x11223344 + (ab4[3] << 32) + (0xAB << 40) + (0xCD << 48)
    assert e[18] == 3 and e[19] == 1
    assert sn.encode_timestamped(q, 77)[19] == 1 + (77 << 8)


def test_reference_fixture_inner_logic_is_satisfiable():
    unsorted, sorted_records, limit = load_storage_fixture()
    assert len(unsorted) == 16 and limit == 16
    inst = sn.instance(unsorted, sorted_records, limit, enforce_permutation=False)
    assert inst["satisfiable"] and inst["completed"]
    assert not inst["permutation_ok"]      # the fixture's sorted side is not a permutation
    cs = storage_cs(limit, enforce_permutation=False)
    assert cs.input_words() == (97, 140)
    r, _, _ = run(cs, [inst], limit)
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"]
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["commitment"]
    # with the entry point's lhs == rhs enforcement the same witness must be rejected
    r2, _, _ = run(storage_cs(limit, True), [sn.instance(unsorted, sorted_records, limit)], limit)
    assert r2.check()[0] > 0
    # dedup result: one final record per touched cell that needs an update
    assert len(inst["final_items"]) >= 1


@pytest.mark.parametrize("seed,n_items,limit", [(1, 6, 8), (2, 8, 8), (3, 13, 16)])
def test_random_storage_log_positive(seed, n_items, limit):
    rng = np.random.default_rng(seed)
    u, s = sn.random_storage_witness(rng, n_items)
    inst = sn.instance(u, s, limit)
    assert inst["satisfiable"] and inst["completed"] and inst["permutation_ok"]
    cs = storage_cs(limit, True)
    r, outer, loop = run(cs, [inst], limit)
    assert r.check()[0] == 0
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["commitment"]
    # generic sequential seeding reproduces the natively derived carried state
    raw = loop.copy(); raw[0:67] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), 1, 65536).seed(outer, raw)
    assert np.array_equal(seeded, loop)


def test_storage_negative_cases():
    limit = 8
    cs = storage_cs(limit, True)
    A, B, X, Y = (5, 77), (9, 3), 0x1111 << 100, 0x2222 << 60
    mk = lambda cell, t, **kw: sn.log_query(address=cell[0], key=cell[1], timestamp=100 + t, **kw)
    u = [mk(A, 0, rw_flag=1, read_value=0, written_value=X), mk(B, 1, rw_flag=0, read_value=0, written_value=0),
         mk(A, 2, rw_flag=0, read_value=X, written_value=X), mk(A, 3, rw_flag=1, read_value=X, written_value=Y),
         mk(A, 4, rw_flag=1, read_value=X, written_value=Y, rollback=1), mk(B, 5, rw_flag=1, read_value=0, written_value=Y)]
    order = [0, 2, 3, 4, 1, 5]                      # (address, key) then position
    s = [(u[i], i) for i in order]
    good = sn.instance(u, s, limit)
    assert good["satisfiable"] and good["completed"] and good["permutation_ok"]
    assert run(cs, [good], limit)[0].check()[0] == 0
    # cell A: written X then Y then rolled back -> final record (read 0, written X); cell B: 0 -> Y
    assert [(q[13], q[21], q[30]) for q in good["final_items"]] == [(0, X & 0xFFFFFFFF, 1), (0, Y & 0xFFFFFFFF, 1)]
    # (a) a read returning a value that was never written (still a permutation: same change on both sides)
    u_bad = [list(q) for q in u]; u_bad[2][13] ^= 1
    s_bad = [(u_bad[i], i) for i in order]
    inst = sn.instance(u_bad, s_bad, limit)
    assert inst["permutation_ok"] and not inst["satisfiable"]
    assert run(cs, [inst], limit)[0].check()[0] > 0
    # (b) sorted side in the wrong order
    s_swapped = [s[1], s[0]] + s[2:]
    inst = sn.instance(u, s_swapped, limit)
    assert not inst["satisfiable"]
    assert run(cs, [inst], limit)[0].check()[0] > 0
    # (c) empty instance is fine
    inst = sn.instance([], [], limit)
    assert inst["satisfiable"] and inst["completed"]
    assert run(cs, [inst], limit)[0].check()[0] == 0
