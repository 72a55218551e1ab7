"""K8: SHA-256 compression through 8-bit lookup tables + block chain, on the CPU oracle interpreter; the
circuit's digest equals hashlib.sha256 (the reference's own SHA-256 fixture lives in code_unpacker_sha256,
/root/reference/src/code_unpacker_sha256/mod.rs:604-612: a bytecode hash must match)."""
import hashlib

import numpy as np
import pytest

import zkgl
from oracle import zko

TABLE_ROWS = 65536 * 3 + 7 * 256
_CS = {}


def sha_cs(n_blocks):
    if n_blocks not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_sha256()
        cs.sha256_blocks_entry_point(n_blocks)
        cs.pad_and_shrink()
        _CS[n_blocks] = cs
    return _CS[n_blocks]


def pad_blocks(msg: bytes, n_blocks: int):
    p = bytearray(msg) + b"\x80"
    while len(p) % 64 != 56:
        p.append(0)
    p += (8 * len(msg)).to_bytes(8, "big")
    assert len(p) == 64 * n_blocks, (len(msg), n_blocks)
    return np.frombuffer(bytes(p), dtype=np.uint8).astype(np.uint64).reshape(n_blocks, 64)


def loop_stream(msgs, n_blocks):
    loop = np.zeros((96, len(msgs) * n_blocks), dtype=np.uint64)
    for i, m in enumerate(msgs):
        loop[32:, i * n_blocks:(i + 1) * n_blocks] = pad_blocks(m, n_blocks).T
    return loop


def run_on_oracle(cs, msgs, n_blocks):
    outer = np.zeros((0, len(msgs)), dtype=np.uint64)
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), TABLE_ROWS).seed(outer, loop_stream(msgs, n_blocks))
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), TABLE_ROWS)
    run.resolve(outer, seeded)
    return run, outer, seeded


@pytest.mark.parametrize("lengths,n_blocks", [((0, 3, 55), 1), ((56, 64, 100, 119), 2), ((120, 183), 3)])
def test_digest_equals_hashlib_sha256(lengths, n_blocks):
    rng = np.random.default_rng(sum(lengths) + 1)
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in lengths]
    if 3 in lengths:
        msgs[lengths.index(3)] = b"abc"
    cs = sha_cs(n_blocks)
    run, _, _ = run_on_oracle(cs, msgs, n_blocks)
    bad, nrel = run.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * len(msgs)
    for i, m in enumerate(msgs):
        assert bytes(int(run.oc[c, i]) for c in cs.public_cells()) == hashlib.sha256(m).digest()


def test_sha256_negative():
    cs = sha_cs(1)
    run, outer, seeded = run_on_oracle(cs, [b"abc"], 1)
    assert run.check()[0] == 0
    bad = seeded.copy(); bad[40, 0] = 300                        # non-byte message word
    r = zko.CircuitRun(cs.export(False), cs.export(True), 1, TABLE_ROWS); r.resolve(outer, bad)
    assert r.check()[0] > 0
    bad = seeded.copy(); bad[0, 0] ^= 1                          # initial state differs from the IV
    r = zko.CircuitRun(cs.export(False), cs.export(True), 1, TABLE_ROWS); r.resolve(outer, bad)
    assert r.check()[0] > 0
