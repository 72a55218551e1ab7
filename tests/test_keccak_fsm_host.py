"""a17: keccak256_round_function_entry_point (the precompile FSM, /root/reference/src/keccak256_round_function/mod.rs:155-794)
recorded through the C-ABI and executed on the CPU oracle interpreter.  The parametrised cases are the reference's own
tests (mod.rs:1096-1144: (length, unalignment) = (50,0) (135,0) (200,0) (180,0) (136,0) (50,31) (135,31) (136,31) (200,31),
two cycles, request at page 123 -> 456): the last memory-queue item must be a write of Keccak256(input) and the
assembly must be satisfied.  On top: equality with the native restatement (oracle/keccak_native.py) cycle by cycle and
on the public input, multi-request / continuation / zero-length cases, negatives."""
import numpy as np
import pytest

import zkgl
from oracle import keccak_native as N
from oracle import zko

TABLE_ROWS = 65536 * 2 + 7 * 256
_CS = {}


def fsm_cs(limit):
    if limit not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_keccak()
        cs.keccak256_round_function_entry_point(limit)
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


def last_write_digest(inst):
    q = inst["pushed"][-1]
    assert q[3] == 1
    return sum(l << (32 * i) for i, l in enumerate(q[5:13])).to_bytes(32, "big")


def reference_case(length, unalignment):
    """test_for_length_and_unalignment (mod.rs:1000-1094): StdRng bytes are replaced by numpy's (any input works)"""
    rng = np.random.default_rng(1000 * length + unalignment)
    data = bytes(rng.integers(0, 256, size=length, dtype=np.uint8))
    req = N.request(data, timestamp=0, input_page=123, input_offset=unalignment, output_page=456, output_offset=0)
    return data, N.instance([req], 2)


# the reference's TEN (length, misalignment) cases, src/keccak256_round_function/mod.rs:1096-1144 (the tenth, keccak_256_unaligned_two_rounds_but_one_read_round,
# was missing here until round 6)
REFERENCE_CASES = [(50, 0), (135, 0), (200, 0), (180, 0), (136, 0), (50, 31), (135, 31), (136, 31), (200, 31), (166, 22)]


def test_layout():
    assert fsm_cs(2).input_words() == (N.OUTER_WORDS, N.LOOP_WORDS)


def test_reference_cases_on_the_oracle_interpreter():
    """all ten reference cases as one batch of instances"""
    cs = fsm_cs(2)
    cases = [reference_case(l, u) for l, u in REFERENCE_CASES]
    for (data, inst) in cases:
        assert inst["satisfiable"] and inst["fsm_out"]["completed"] == 1
        assert last_write_digest(inst) == zko.keccak256(data)
    insts = [c[1] for c in cases]
    outer, loop = streams(insts, 2)
    blank = loop.copy()
    blank[:N.CARRIED, :] = 0
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), TABLE_ROWS).seed(outer, blank)
    assert np.array_equal(seeded, loop), "generic seeding differs from the native FSM trajectory"
    r = run(cs, outer, loop, len(insts))
    bad, nrel = r.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * len(insts)
    for i, inst in enumerate(insts):
        assert [int(r.oc[c, i]) for c in cs.public_cells()] == inst["public_input"]


def make_requests(datas, offsets):
    return [N.request(d, timestamp=3 + 5 * i, input_page=50 + i, input_offset=o, output_page=90 + i, output_offset=i)
            for i, (d, o) in enumerate(zip(datas, offsets))]


def test_several_requests_zero_length_and_continuation():
    rng = np.random.default_rng(7)
    datas = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (300, 0, 272, 1)]
    offsets = [70, 5, 0, 63]
    whole = N.instance(make_requests(datas, offsets), 9)
    assert whole["satisfiable"] and whole["fsm_out"]["completed"] == 1
    writes = [q for q in whole["pushed"] if q[3] == 1]
    assert [sum(l << (32 * i) for i, l in enumerate(q[5:13])).to_bytes(32, "big") for q in writes] == [zko.keccak256(d) for d in datas]
    a = N.instance(make_requests(datas, offsets), 3)
    b = N.instance(a["rest"][0], 3, start_flag=False, fsm_in=a["fsm_out"], obs_req=a["obs_req"], obs_mem=a["obs_mem"], pending=a["rest"][1])
    c = N.instance(b["rest"][0], 3, start_flag=False, fsm_in=b["fsm_out"], obs_req=a["obs_req"], obs_mem=a["obs_mem"], pending=b["rest"][1])
    assert a["fsm_out"]["completed"] == 0 and b["fsm_out"]["completed"] == 0 and c["fsm_out"]["completed"] == 1
    assert c["memory_state"] == whole["memory_state"]
    cs = fsm_cs(3)
    outer, loop = streams([a, b, c], 3)
    r = run(cs, outer, loop, 3)
    assert r.check()[0] == 0
    for i, inst in enumerate((a, b, c)):
        assert [int(r.oc[c_, i]) for c_ in cs.public_cells()] == inst["public_input"]


def test_empty_queue_finishes_immediately():
    inst = N.instance([], 2)
    assert inst["satisfiable"] and inst["fsm_out"]["completed"] == 1 and inst["pushed"] == []
    cs = fsm_cs(2)
    outer, loop = streams([inst], 2)
    r = run(cs, outer, loop, 1)
    assert r.check()[0] == 0
    assert [int(r.oc[c, 0]) for c in cs.public_cells()] == inst["public_input"]


@pytest.mark.parametrize("kind", ["address", "aux_byte", "read_value", "buffer_byte"])
def test_fsm_negative(kind):
    data = b"zkgl" * 40
    kw = dict(address=0x02) if kind == "address" else (dict(aux_byte=0) if kind == "aux_byte" else {})
    inst = N.instance([N.request(data, 1, 2, 3, 4, 5, **kw)], 2)
    cs = fsm_cs(2)
    outer, loop = streams([inst], 2)
    if kind in ("address", "aux_byte"):
        assert not inst["satisfiable"]
    elif kind == "read_value":
        loop[459, 0] ^= 1     # a memory value differs from what the carried chain was computed with
    else:
        loop[212 + 3, 1] ^= 1   # carried buffer byte of cycle 1 differs from the output of cycle 0
    r = run(cs, outer, loop, 1)
    assert r.check()[0] > 0
