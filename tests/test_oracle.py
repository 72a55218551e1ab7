"""Pins the CPU oracle itself (no GPU): known constants, independent pure-Python restatements,
hashlib known answers, the reference's own fixture.  SURVEY.md §8c."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import ram_native as rn
from oracle import zko

P = zko.P
GOLD = os.path.join(os.path.dirname(__file__), "golden")
rng = np.random.default_rng(0xC0FFEE)


def rand_fe(n):
    return [int(x) for x in (rng.integers(0, 2**63, size=n, dtype=np.uint64).astype(object) * 2 + rng.integers(0, 2, size=n)) % P]


EDGE = [0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 1 << 63]


# ---------------------------------------------------------------- Goldilocks
def test_field_ops_against_python_ints():
    vals = EDGE + rand_fe(200)
    for a in vals:
        for b in vals[:12]:
            assert zko.gl_add(a, b) == (a + b) % P
            assert zko.gl_sub(a, b) == (a - b) % P
            assert zko.gl_mul(a, b) == (a * b) % P
        inv = zko.gl_inv(a)
        assert inv == (pow(a, P - 2, P) if a else 0)
        if a:
            assert inv * a % P == 1


# ---------------------------------------------------------------- Poseidon constants
def test_round_constants_match_published_values():
    known = json.load(open(os.path.join(GOLD, "poseidon_rc_known.json")))
    rc = zko.round_constants()
    assert len(rc) == 360 and all(int(x) < P for x in rc)
    assert [int(x) for x in rc[:14]] == [int(x, 16) for x in known["first"]]
    assert int(rc[359]) == int(known["last"], 16)


# ---------------------------------------------------------------- Poseidon2: pure-python twin
M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]
SHIFTS = [4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12]


def py_mds_external(s):
    me = [[(2 if (i // 4) == (j // 4) else 1) * M4[i % 4][j % 4] for j in range(12)] for i in range(12)]
    return [sum(me[i][j] * s[j] for j in range(12)) % P for i in range(12)]


def py_mds_inner(s):
    tot = sum(s)
    return [(tot + (s[i] << SHIFTS[i])) % P for i in range(12)]


def py_poseidon2(s):
    rc = [int(x) for x in zko.round_constants()]
    s = py_mds_external(list(s))
    for r in range(30):
        if r < 4 or r >= 26:
            s = py_mds_external([pow((s[i] + rc[12 * r + i]) % P, 7, P) for i in range(12)])
        else:
            s[0] = pow((s[0] + rc[12 * r]) % P, 7, P)
            s = py_mds_inner(s)
    return s


def test_poseidon2_c_oracle_equals_python_matrix_form():
    for case in ([0] * 12, list(range(12)), [P - 1] * 12, rand_fe(12), rand_fe(12)):
        assert zko.mds_external(case) == py_mds_external(case)
        assert zko.mds_inner(case) == py_mds_inner(case)
        assert zko.poseidon2_permute(case) == py_poseidon2(case)


def test_poseidon2_golden_vectors():
    g = json.load(open(os.path.join(GOLD, "poseidon2_vectors.json")))
    for v in g["permute"]:
        assert zko.poseidon2_permute([int(x, 16) for x in v["in"]]) == [int(x, 16) for x in v["out"]]
    for v in g["commit_encoding"]:
        assert zko.commit_encoding([int(x, 16) for x in v["in"]]) == [int(x, 16) for x in v["out"]]


def test_poseidon2_is_a_permutation_on_a_sample():
    outs = {tuple(zko.poseidon2_permute([i] + [0] * 11)) for i in range(64)}
    assert len(outs) == 64


# ---------------------------------------------------------------- sponge rules
def py_commit(values):
    s = [0] * 12
    s[11] = len(values) % P
    for c in range((len(values) + 7) // 8):
        chunk = values[8 * c: 8 * c + 8]
        s[:8] = chunk + [0] * (8 - len(chunk))
        s = py_poseidon2(s)
    return s[:4]


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 16, 18, 69])
def test_commit_encoding_rule(n):
    v = rand_fe(n)
    assert zko.commit_encoding(v) == py_commit(v)


def test_fs_challenges_rule():
    # ram: 26 inputs -> [[..9]; 2] ; storage: 10 inputs -> [[..21]; 2]   (src/utils.rs:12-78)
    for n_in, nchal in ((26, 9), (10, 21)):
        v = rand_fe(n_in)
        s = [0] * 12
        s[11] = n_in
        for c in range((n_in + 7) // 8):
            chunk = v[8 * c: 8 * c + 8]
            s[:8] = chunk + [0] * (8 - len(chunk))
            s = py_poseidon2(s)
        exp, can = [], 8
        for r in range(2):
            row = [1]
            for _ in range(1, nchal):
                if can == 0:
                    s = py_poseidon2(s); can = 8
                row.append(s[8 - can]); can -= 1
            exp.append(row)
        assert zko.fs_challenges(v, 2, nchal) == exp


def test_queue_rules():
    tail, enc = rand_fe(12), rand_fe(8)
    assert zko.queue_full_push(tail, enc) == py_poseidon2(enc + tail[8:])
    t4, e20 = rand_fe(4), rand_fe(20)
    r0 = py_poseidon2(e20[:8] + [0] * 4)
    r1 = py_poseidon2(e20[8:16] + r0[8:])
    r2 = py_poseidon2(e20[16:20] + t4 + r1[8:])
    assert zko.queue_tail4_push20(t4, e20) == r2[:4]


# ---------------------------------------------------------------- encodings / grand product
def test_memory_query_encode():
    for _ in range(50):
        q = rn.mq(int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)),
                  int(rng.integers(0, 2)), int(rng.integers(0, 2)), int.from_bytes(rng.bytes(32), "little"))
        v = q[5:]
        by = lambda x: [(x >> (8 * i)) & 0xFF for i in range(4)]
        b5, b6, b7 = by(v[5]), by(v[6]), by(v[7])
        exp = [q[0], q[1], q[2] + (q[3] << 32) + (q[4] << 33),
               v[0] + (b5[0] << 32) + (b5[1] << 40) + (b5[2] << 48),
               v[1] + (b5[3] << 32) + (b6[0] << 40) + (b6[1] << 48),
               v[2] + (b6[2] << 32) + (b6[3] << 40) + (b7[0] << 48),
               v[3] + (b7[1] << 32) + (b7[2] << 40) + (b7[3] << 48), v[4]]
        assert zko.memory_query_encode(q) == exp
        assert all(e < P for e in exp)


def test_execution_context_record_encode():
    """a11: 42 variables -> 32 elements (src/base_structures/vm_state/saved_context.rs:111-266)"""
    names = (["this"] * 5 + ["caller"] * 5 + ["code_address"] * 5 + ["code_page", "base_page", "heap_ub", "aux_heap_ub"] +
             ["rq_head"] * 4 + ["rq_tail"] * 4 + ["seg_len", "pc", "sp", "eh_loc", "ergs", "is_static", "is_kernel",
                                                   "this_shard", "caller_shard", "code_shard"] + ["ctx128"] * 4 + ["is_local"])
    assert len(names) == 42
    for _ in range(20):
        rec = []
        for nm in names:
            if nm in ("rq_head", "rq_tail"): rec.append(rand_fe(1)[0])
            elif nm in ("pc", "sp", "eh_loc"): rec.append(int(rng.integers(0, 2**16)))
            elif nm.startswith("is_"): rec.append(int(rng.integers(0, 2)))
            elif nm.endswith("_shard"): rec.append(int(rng.integers(0, 256)))
            else: rec.append(int(rng.integers(0, 2**32)))
        f = {nm: [rec[i] for i in range(42) if names[i] == nm] for nm in set(names)}
        one = lambda k: f[k][0]
        sb = [(one("seg_len") >> (8 * i)) & 0xFF for i in range(4)]
        exp = (f["rq_head"] + f["rq_tail"] + f["code_address"] + f["this"] + f["caller"] + f["ctx128"] + [
            one("code_page") + (one("pc") << 32) + (one("this_shard") << 48) + (one("is_static") << 56),
            one("base_page") + (one("sp") << 32) + (one("caller_shard") << 48) + (one("is_kernel") << 56),
            one("ergs") + (one("eh_loc") << 32) + (one("code_shard") << 48) + (one("is_local") << 56),
            one("heap_ub") + (sb[0] << 32) + (sb[1] << 40),
            one("aux_heap_ub") + (sb[2] << 32) + (sb[3] << 40)])
        assert zko.execution_context_encode(rec) == exp and all(e < P for e in exp)


def test_grand_product_and_permutation_property():
    n, L = 40, 8
    enc = np.array([rand_fe(L) for _ in range(n)], dtype=np.uint64)
    ch = rand_fe(L + 1)
    flags = [1] * n
    acc = zko.grand_product(enc, flags, ch, 1)
    exp, run = [], 1
    for i in range(n):
        c = (ch[L] + sum(int(enc[i, j]) * ch[j] for j in range(L))) % P
        run = run * c % P
        exp.append(run)
    assert [int(x) for x in acc] == exp
    perm = rng.permutation(n)
    assert int(zko.grand_product(enc[perm], flags, ch, 1)[-1]) == exp[-1]  # permutation invariant
    enc2 = enc.copy(); enc2[3, 2] ^= 1
    assert int(zko.grand_product(enc2, flags, ch, 1)[-1]) != exp[-1]
    fl = [i % 3 != 0 for i in range(n)]
    run = 7
    for i in range(n):
        if fl[i]:
            run = run * ((ch[L] + sum(int(enc[i, j]) * ch[j] for j in range(L))) % P) % P
    assert int(zko.grand_product(enc, fl, ch, 7)[-1]) == run


# ---------------------------------------------------------------- hashes pinned by hashlib
def sha3_256_via_oracle_permutation(msg: bytes) -> bytes:
    st = [0] * 25
    padded = bytearray(msg) + b"\x06"
    while len(padded) % 136:
        padded.append(0)
    padded[-1] ^= 0x80
    for off in range(0, len(padded), 136):
        for i in range(17):
            st[i] ^= int.from_bytes(padded[off + 8 * i: off + 8 * i + 8], "little")
        st = zko.keccak_f1600(st)
    return b"".join(x.to_bytes(8, "little") for x in st)[:32]


def test_keccak_f_pinned_by_sha3_256():
    for n in (0, 1, 135, 136, 137, 300):
        m = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        assert sha3_256_via_oracle_permutation(m) == hashlib.sha3_256(m).digest()


def test_keccak256_known_answers():
    assert zko.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert zko.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # reference test shapes (len, misalignment): src/keccak256_round_function/mod.rs:1096-1144
    for n in (50, 135, 136, 166, 180, 200):
        m = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        st = [0] * 25  # independent sponge around the pinned permutation, 0x01 padding
        padded = bytearray(m) + b"\x01"
        while len(padded) % 136:
            padded.append(0)
        padded[-1] ^= 0x80
        for off in range(0, len(padded), 136):
            for i in range(17):
                st[i] ^= int.from_bytes(padded[off + 8 * i: off + 8 * i + 8], "little")
            st = zko.keccak_f1600(st)
        assert zko.keccak256(m) == b"".join(x.to_bytes(8, "little") for x in st)[:32]


def test_sha256_compression_pinned_by_hashlib():
    iv = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
    msg = bytes(rng.integers(0, 256, size=55, dtype=np.uint8))
    block = msg + b"\x80" + (len(msg) * 8).to_bytes(8, "big")
    assert len(block) == 64
    out = zko.sha256_compress(iv, block)
    assert b"".join(x.to_bytes(4, "big") for x in out) == hashlib.sha256(msg).digest()


# ---------------------------------------------------------------- reference fixture, native path
def load_fixture():
    f = json.load(open(os.path.join(GOLD, "ram_fixture.json")))
    conv = lambda lst: [rn.mq(a, rn.BOOTLOADER_HEAP_PAGE if b == "BOOTLOADER_HEAP_PAGE" else b, c, d, e, v) for a, b, c, d, e, v in lst]
    return conv(f["unsorted"]), conv(f["sorted"]), f["limit"]


def test_reference_ram_fixture_is_accepted_natively():
    u, s, limit = load_fixture()
    inst = rn.instance(u, s, limit, 1)
    assert inst["satisfiable"] and inst["completed"]
    gold = json.load(open(os.path.join(GOLD, "ram_commitments.json")))
    assert inst["commitment"] == [int(x, 16) for x in gold["fixture_limit16"]]
    # grand-product equality is what the entry point (not the _inner test) enforces: mod.rs:164-168
    assert inst["fsm_out"]["lhs"] == inst["fsm_out"]["rhs"]


def test_native_rejects_bad_witnesses():
    u, s, limit = load_fixture()
    assert not rn.instance(u, [s[1], s[0], s[2]], limit, 1)["satisfiable"]          # not sorted
    bad = [list(x) for x in s]; bad[2][5] ^= 1                                        # read returns a different value
    assert not rn.instance(u, bad, limit, 1)["satisfiable"]
    assert not rn.instance(u, s, limit, 0)["satisfiable"]                             # wrong non-det write count
