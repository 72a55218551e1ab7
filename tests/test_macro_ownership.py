"""Who evaluates a gate on a macro-op's output (VERDICT r4 "mirror by trust", ADVICE r4 cs.cpp:720).

The fused mode of resolve_and_check leaves a gate to the witness kernels when the op producing its output computes the gate's relation.
For the hash macro-ops (ZK_OP_KECCAK_F, ZK_OP_SHA256_ROUNDS, ZK_OP_BYTEBUF_FILL) that is true of the gates THE GADGET places while it walks
the op's structure — and of nothing else.  The recorder tags those gates inside the gadget's window (CS::emit_macro_op .. end_macro_op,
GateRec::owner); a gate placed from outside on a macro output (zk_cs_place_gate is public ABI, variables are plain integers) has no
owner and must be read from the store: fused verdict == stored verdict == oracle verdict.

The circuits are recorded through the C ABI with the gadget-level entries zk_gadget_keccak_f1600 / zk_gadget_sha256_compress."""
import hashlib

import numpy as np
import pytest

import zkgl
from oracle import zko
from zkgl import GATE as G


def keccak_circuit(extra=None):
    """outer scope only: 200 input bytes -> Keccak-f[1600]; public inputs = the first 32 output bytes.  extra(cs, ins, outs) may add gates."""
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_keccak()
    ins = [cs.input(w) for w in range(200)]
    outs = cs.keccak_f1600(ins)
    if extra:
        extra(cs, ins, outs)
    for v in outs[:32]:
        cs.place_gate(G["PUBLIC_INPUT"], [v])
    cs.pad_and_shrink()
    return cs


def sha_circuit(extra=None, reference_tables=False):
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_sha256(reference_tables)
    st = [cs.input(w) for w in range(32)]
    blk = [cs.input(32 + w) for w in range(64)]
    outs = cs.sha256_compress(st, blk)
    if extra:
        extra(cs, st + blk, outs)
    for v in outs:
        cs.place_gate(G["PUBLIC_INPUT"], [v])
    cs.pad_and_shrink()
    return cs


def forged(cs, ins, outs):
    # 4 * outs[1] - outs[0] == 0: false for (almost) every state; its output variable is a macro-op output
    cs.place_gate(G["REDUCTION4"], [outs[1], outs[1], outs[1], outs[1], outs[0]], [1, 1, 1, 1])


def honest(cs, ins, outs):
    # outs[0] - outs[0] == 0 written as a reduction: holds for every witness
    z = cs.allocate_constant(0)
    cs.place_gate(G["REDUCTION4"], [outs[0], z, z, z, outs[0]], [1, 0, 0, 0])


@pytest.mark.parametrize("circuit", [keccak_circuit, sha_circuit])
def test_a_foreign_gate_on_a_macro_output_stays_in_the_fused_check_program(circuit):
    base = circuit().stats()
    more = circuit(forged).stats()
    for st in (base, more):
        assert st["constraints_per_instance"] == st["constraints_from_store_fused"] + st["constraints_in_witness_fused"]
    assert more["constraints_per_instance"] == base["constraints_per_instance"] + 1
    # the gadget's own gates and tuples are evaluated where the op produces them ...
    assert more["constraints_in_witness_fused"] == base["constraints_in_witness_fused"] > 0
    # ... the outsider's gate is not
    assert more["constraints_from_store_fused"] == base["constraints_from_store_fused"] + 1


def keccak_f_bytes(state200: bytes) -> bytes:
    """software Keccak-f[1600] on 200 bytes (lanes little-endian)"""
    RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
          0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
          0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
          0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
    ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
    M = (1 << 64) - 1
    rol = lambda x, n: ((x << n) | (x >> (64 - n))) & M if n else x
    A = [[int.from_bytes(state200[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rc in RC:
        Cc = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [Cc[(x - 1) % 5] ^ rol(Cc[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        Bm = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                Bm[y][(2 * x + 3 * y) % 5] = rol(A[x][y], ROT[x][y])
        A = [[Bm[x][y] ^ ((~Bm[(x + 1) % 5][y]) & Bm[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= rc
    out = bytearray(200)
    for x in range(5):
        for y in range(5):
            out[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = A[x][y].to_bytes(8, "little")
    return bytes(out)


def test_keccak_f_gadget_on_the_oracle_equals_software_keccak_f():
    cs = keccak_circuit(honest)
    rng = np.random.default_rng(11)
    B = 3
    inp = rng.integers(0, 256, size=(200, B), dtype=np.uint64)
    run = zko.CircuitRun(cs.export(False), cs.export(True), B, 65536 * 2 + 7 * 256)
    run.resolve(inp, np.zeros((0, B), dtype=np.uint64))
    bad, nrel = run.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * B
    for i in range(B):
        want = keccak_f_bytes(bytes(int(x) for x in inp[:, i]))
        assert bytes(int(run.oc[c, i]) for c in cs.public_cells()) == want[:32]
    # the outsider's false gate is a false relation for the oracle checker too
    cs2 = keccak_circuit(forged)
    run2 = zko.CircuitRun(cs2.export(False), cs2.export(True), B, 65536 * 2 + 7 * 256)
    run2.resolve(inp, np.zeros((0, B), dtype=np.uint64))
    bad2, _ = run2.check()
    assert bad2 == B


def sha_compress_bytes(state: bytes, block: bytes) -> bytes:
    """software compression: 8 LE-byte words of state, 16 LE-byte message words -> 8 LE-byte words"""
    K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
         0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
         0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
         0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
         0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
         0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
    M = 0xffffffff
    ror = lambda x, n: ((x >> n) | (x << (32 - n))) & M
    h = [int.from_bytes(state[4 * i:4 * i + 4], "little") for i in range(8)]
    w = [int.from_bytes(block[4 * i:4 * i + 4], "little") for i in range(16)]
    for t in range(16, 64):
        s0 = ror(w[t - 15], 7) ^ ror(w[t - 15], 18) ^ (w[t - 15] >> 3)
        s1 = ror(w[t - 2], 17) ^ ror(w[t - 2], 19) ^ (w[t - 2] >> 10)
        w.append((w[t - 16] + s0 + w[t - 7] + s1) & M)
    a, b, c, d, e, f, g, hh = h
    for t in range(64):
        t1 = (hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g & M)) + K[t] + w[t]) & M
        t2 = ((ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & M
        hh, g, f, e, d, c, b, a = g, f, e, (d + t1) & M, c, b, a, (t1 + t2) & M
    return b"".join(((x + y) & M).to_bytes(4, "little") for x, y in zip(h, [a, b, c, d, e, f, g, hh]))


def sha_inputs(rng, B):
    return rng.integers(0, 256, size=(96, B), dtype=np.uint64)


@pytest.mark.parametrize("reference_tables", [False, True])
def test_sha256_compress_gadget_on_the_oracle_equals_software_compression(reference_tables):
    cs = sha_circuit(honest, reference_tables)
    rng = np.random.default_rng(12)
    B = 3
    inp = sha_inputs(rng, B)
    run = zko.CircuitRun(cs.export(False), cs.export(True), B, 1 << 20)
    run.resolve(inp, np.zeros((0, B), dtype=np.uint64))
    bad, nrel = run.check()
    assert bad == 0 and nrel == cs.stats()["constraints_per_instance"] * B
    for i in range(B):
        col = bytes(int(x) for x in inp[:, i])
        assert bytes(int(run.oc[c, i]) for c in cs.public_cells()) == sha_compress_bytes(col[:32], col[32:])
    # one block of a real message: IV + padded "abc" gives the known digest
    iv = b"".join(x.to_bytes(4, "little") for x in (0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19))
    msg = b"abc" + b"\x80" + b"\x00" * 52 + (24).to_bytes(8, "big")
    le_block = b"".join(msg[4 * i:4 * i + 4][::-1] for i in range(16))
    out = sha_compress_bytes(iv, le_block)
    assert b"".join(out[4 * i:4 * i + 4][::-1] for i in range(8)) == hashlib.sha256(b"abc").digest()


# (the device half — forged / honest gates and adversarial inputs in both check modes: tests/test_zz_round5_gpu.py)
