"""K8: Keccak-f[1600] through 8-bit lookup tables + Keccak-256 sponge over pre-padded blocks, on the CPU oracle
interpreter.  Pattern B of the reference's tests: the circuit's digest equals a software Keccak-256
(/root/reference/src/keccak256_round_function/mod.rs:1007-1011, 1087) for the reference's message lengths
(:1096-1144), and the trace is satisfiable."""
import hashlib

import numpy as np
import pytest

import zkgl
from oracle import zko

TABLE_ROWS = 65536 * 2 + 7 * 256
_CS = {}


def keccak_cs(n_blocks):
    if n_blocks not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_keccak()
        cs.keccak256_blocks_entry_point(n_blocks)
        cs.pad_and_shrink()
        _CS[n_blocks] = cs
    return _CS[n_blocks]


def pad_blocks(msg: bytes, n_blocks: int):
    """Keccak-256 padding (0x01 .. 0x80) to exactly n_blocks blocks of 136 bytes -> u64 array [n_blocks, 136]"""
    p = bytearray(msg) + b"\x01"
    while len(p) % 136:
        p.append(0)
    p[-1] ^= 0x80
    assert len(p) == 136 * n_blocks, (len(msg), n_blocks)
    return np.frombuffer(bytes(p), dtype=np.uint8).astype(np.uint64).reshape(n_blocks, 136)


def loop_stream(msgs, n_blocks):
    """raw loop input stream [336, B*n_blocks]: carried state words left 0 (seeded), block bytes filled"""
    B = len(msgs)
    loop = np.zeros((336, B * n_blocks), dtype=np.uint64)
    for i, m in enumerate(msgs):
        loop[200:, i * n_blocks:(i + 1) * n_blocks] = pad_blocks(m, n_blocks).T
    return loop


def run_on_oracle(cs, msgs, n_blocks):
    outer = np.zeros((0, len(msgs)), dtype=np.uint64)
    raw = loop_stream(msgs, n_blocks)
    seeded = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), TABLE_ROWS).seed(outer, raw)
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(msgs), TABLE_ROWS)
    run.resolve(outer, seeded)
    return run, outer, seeded


def digest_of(run, cs, i):
    return bytes(int(run.oc[c, i]) for c in cs.public_cells())


def test_keccak_circuit_shape():
    cs = keccak_cs(1)
    st = cs.stats()
    assert cs.input_words() == (0, 336)
    # ~1030 lookups per round x 24 rounds + 136 absorb + 32 capacity range checks
    assert 24000 < st["lookups_per_instance"] < 26500
    assert st["gate_instances"]["MATMUL12_EXT"] == 0          # no Poseidon2 on this path


@pytest.mark.parametrize("lengths,n_blocks", [((0, 50, 135), 1), ((136, 166, 180, 200), 2)])
def test_digest_equals_software_keccak256(lengths, n_blocks):
    rng = np.random.default_rng(sum(lengths))
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in lengths]
    cs = keccak_cs(n_blocks)
    run, _, _ = run_on_oracle(cs, msgs, n_blocks)
    bad, nrel = run.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * len(msgs)
    for i, m in enumerate(msgs):
        assert digest_of(run, cs, i) == zko.keccak256(m)
    assert digest_of(run, cs, 0).hex() == zko.keccak256(msgs[0]).hex()
    if lengths[0] == 0:
        assert digest_of(run, cs, 0).hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"


def test_state_carry_tampering_and_non_byte_inputs_are_rejected():
    n_blocks = 2
    cs = keccak_cs(n_blocks)
    run, outer, seeded = run_on_oracle(cs, [b"x" * 150], n_blocks)
    assert run.check()[0] == 0
    bad_state = seeded.copy(); bad_state[17, 1] ^= 1           # a carried state byte entering block 1
    r2 = zko.CircuitRun(cs.export(False), cs.export(True), 1, TABLE_ROWS); r2.resolve(outer, bad_state)
    assert r2.check()[0] > 0
    bad_in = seeded.copy(); bad_in[205, 0] = 256               # "byte" out of range: the xor lookup must fail
    r3 = zko.CircuitRun(cs.export(False), cs.export(True), 1, TABLE_ROWS); r3.resolve(outer, bad_in)
    assert r3.check()[0] > 0
