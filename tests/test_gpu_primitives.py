"""-m gpu: the hand-written HIP primitives (K1-K4, a9), called through the C ABI, bit-exact against
the CPU oracle on seeded inputs, edge cases included (empty, single, odd, non-multiple-of-tile sizes,
values adjacent to p and to 2^32)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLD, P, rand_fe

pytestmark = pytest.mark.gpu

EDGE = [0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, (1 << 63), (1 << 63) + 1]


def fe_array(rng, n):
    a = np.array(rand_fe(rng, n), dtype=np.uint64)
    k = min(n, len(EDGE))
    a[:k] = np.array(EDGE[:k], dtype=np.uint64)
    return a


@pytest.mark.parametrize("n", [0, 1, 2, 3, 255, 1000, 4097, 1 << 18])
def test_column_ops(zk, oracle, n):
    rng = np.random.default_rng(100 + n)
    a, b, c = fe_array(rng, n), fe_array(rng, n)[::-1].copy(), fe_array(rng, n)
    rng.shuffle(c)
    q, l = rand_fe(rng, 2)
    da, db, dc = (zk.DeviceBuffer.from_numpy(x) for x in (a, b, c))
    dst = zk.DeviceBuffer(n)
    ai, bi, ci = a.astype(object), b.astype(object), c.astype(object)
    zk.gl_fma_cols(dst, da, db, dc, q, l, n); zk.sync()
    assert np.array_equal(dst.to_numpy(), np.array((q * ai * bi + l * ci) % P, dtype=np.uint64).reshape(n))
    if n:
        assert np.array_equal(dst.to_numpy(), oracle.gl_fma_cols(a, b, c, q, l))
    zk.gl_add_cols(dst, da, db, n); zk.sync()
    assert np.array_equal(dst.to_numpy(), np.array((ai + bi) % P, dtype=np.uint64).reshape(n))
    zk.gl_sub_cols(dst, da, db, n); zk.sync()
    assert np.array_equal(dst.to_numpy(), np.array((ai - bi) % P, dtype=np.uint64).reshape(n))
    zk.gl_mul_cols(dst, da, db, n); zk.sync()
    assert np.array_equal(dst.to_numpy(), np.array((ai * bi) % P, dtype=np.uint64).reshape(n))
    sel = np.array(rng.integers(0, 2, size=n), dtype=np.uint64)
    ds = zk.DeviceBuffer.from_numpy(sel)
    zk.gl_select_cols(dst, ds, da, db, n); zk.sync()
    assert np.array_equal(dst.to_numpy(), np.where(sel == 1, a, b))
    if n <= 4097:
        zk.gl_inv_cols(dst, da, n); zk.sync()
        inv = dst.to_numpy()
        assert all(int(inv[i]) == (pow(int(a[i]), P - 2, P) if a[i] else 0) for i in range(n))


def test_column_ops_in_place(zk):
    rng = np.random.default_rng(7)
    n = 1001
    a, b = fe_array(rng, n), fe_array(rng, n)
    da, db = zk.DeviceBuffer.from_numpy(a), zk.DeviceBuffer.from_numpy(b)
    zk.gl_mul_cols(da, da, db, n); zk.sync()
    assert np.array_equal(da.to_numpy(), np.array((a.astype(object) * b.astype(object)) % P, dtype=np.uint64))


@pytest.mark.parametrize("n", [0, 1, 63, 64, 255, 256, 257, 1000, 1 << 16])
def test_poseidon2_batched(zk, oracle, n):
    rng = np.random.default_rng(200 + n)
    st = np.array([rand_fe(rng, 12) for _ in range(n)], dtype=np.uint64).reshape(n, 12)
    if n >= 3:
        st[0] = 0; st[1] = P - 1; st[2] = np.arange(12)
    exp = oracle.poseidon2_permute_batch(st) if n else st
    d = zk.DeviceBuffer.from_numpy(st)
    zk.poseidon2_permute_aos(d, n); zk.sync()
    assert np.array_equal(d.to_numpy().reshape(n, 12), exp)
    stride = n + 5
    soa = np.zeros((12, stride), dtype=np.uint64)
    soa[:, :n] = st.T
    d2 = zk.DeviceBuffer.from_numpy(soa)
    zk.poseidon2_permute_soa(d2, n, stride); zk.sync()
    got = d2.to_numpy().reshape(12, stride)
    assert np.array_equal(got[:, :n].T, exp)
    assert not got[:, n:].any()  # padding untouched


def test_poseidon2_golden_vectors_on_gpu(zk):
    g = json.load(open(os.path.join(GOLD, "poseidon2_vectors.json")))
    st = np.array([[int(x, 16) for x in v["in"]] for v in g["permute"]], dtype=np.uint64)
    d = zk.DeviceBuffer.from_numpy(st)
    zk.poseidon2_permute_aos(d, st.shape[0]); zk.sync()
    exp = np.array([[int(x, 16) for x in v["out"]] for v in g["permute"]], dtype=np.uint64)
    assert np.array_equal(d.to_numpy().reshape(exp.shape), exp)


@pytest.mark.parametrize("length", [0, 1, 8, 9, 18, 51, 69])
def test_commit_encoding_batch(zk, oracle, length):
    rng = np.random.default_rng(300 + length)
    n = 300
    enc = np.array([rand_fe(rng, length) for _ in range(n)], dtype=np.uint64).reshape(n, length)
    din = zk.DeviceBuffer.from_numpy(enc.T.copy())
    dout = zk.DeviceBuffer(4 * n)
    zk.commit_encoding_batch(din, length, n, dout); zk.sync()
    got = dout.to_numpy().reshape(4, n).T
    for i in (0, 1, n // 2, n - 1):
        assert [int(x) for x in got[i]] == oracle.commit_encoding([int(x) for x in enc[i]])


def test_queue_full_push_chain(zk, oracle):
    rng = np.random.default_rng(41)
    nq, items = 70, 9
    enc = np.array(rand_fe(rng, nq * items * 8), dtype=np.uint64)
    tails = np.array(rand_fe(rng, nq * 12), dtype=np.uint64)
    tails[:12] = 0
    d_enc, d_tail = zk.DeviceBuffer.from_numpy(enc), zk.DeviceBuffer.from_numpy(tails)
    d_states = zk.DeviceBuffer(nq * items * 12)
    zk.queue_full_push_chain(d_enc, nq, items, d_tail, d_states); zk.sync()
    got_t, got_s = d_tail.to_numpy().reshape(nq, 12), d_states.to_numpy().reshape(nq, items, 12)
    for q in (0, 1, 63, 64, nq - 1):
        t = [int(x) for x in tails[12 * q: 12 * q + 12]]
        for k in range(items):
            assert [int(x) for x in got_s[q, k]] == t
            t = oracle.queue_full_push(t, [int(x) for x in enc[(q * items + k) * 8: (q * items + k) * 8 + 8]])
        assert [int(x) for x in got_t[q]] == t
    zk.queue_full_push_chain(d_enc, nq, 0, d_tail, None); zk.sync()  # zero items: tails unchanged
    assert np.array_equal(d_tail.to_numpy().reshape(nq, 12), got_t)


def test_memory_query_encode(zk, oracle):
    rng = np.random.default_rng(43)
    n = 777
    q = np.zeros((13, n), dtype=np.uint64)
    q[0:3] = rng.integers(0, 2**32, size=(3, n), dtype=np.uint64)
    q[3:5] = rng.integers(0, 2, size=(2, n), dtype=np.uint64)
    q[5:13] = rng.integers(0, 2**32, size=(8, n), dtype=np.uint64)
    q[:, 0] = [0xFFFFFFFF] * 3 + [1, 1] + [0xFFFFFFFF] * 8
    q[:, 1] = 0
    d_q, d_e = zk.DeviceBuffer.from_numpy(q), zk.DeviceBuffer(8 * n)
    zk.memory_query_encode(d_q, n, d_e); zk.sync()
    got = d_e.to_numpy().reshape(8, n)
    for i in range(n):
        assert [int(x) for x in got[:, i]] == oracle.memory_query_encode([int(x) for x in q[:, i]])


def test_execution_context_encode(zk, oracle):
    rng = np.random.default_rng(47)
    n = 513
    rec = np.zeros((42, n), dtype=np.uint64)
    rec[:] = rng.integers(0, 2**32, size=(42, n), dtype=np.uint64)
    rec[19:27] = rng.integers(0, 2**63, size=(8, n), dtype=np.uint64) % np.uint64(P)   # reverted queue head / tail
    rec[28:31] = rng.integers(0, 2**16, size=(3, n))                                   # pc, sp, exception handler
    rec[32:34] = rng.integers(0, 2, size=(2, n)); rec[41] = rng.integers(0, 2, size=n)  # booleans
    rec[34:37] = rng.integers(0, 256, size=(3, n))                                     # shard ids
    rec[:, 0] = 0
    d_r, d_e = zk.DeviceBuffer.from_numpy(rec), zk.DeviceBuffer(32 * n)
    zk.execution_context_encode(d_r, n, d_e); zk.sync()
    got = d_e.to_numpy().reshape(32, n)
    for i in range(n):
        assert [int(x) for x in got[:, i]] == oracle.execution_context_encode([int(x) for x in rec[:, i]])


@pytest.mark.parametrize("n,enc_len", [(1, 8), (5, 8), (1023, 8), (1024, 8), (1025, 20), (300000, 8), (70000, 20)])
def test_grand_product(zk, oracle, n, enc_len):
    rng = np.random.default_rng(500 + n)
    enc = np.array(rand_fe(rng, n * enc_len), dtype=np.uint64).reshape(n, enc_len)
    flags = np.array(rng.integers(0, 4, size=n) != 0, dtype=np.uint64)
    if n > 4:
        flags[:2] = 0
    ch = np.array([1] + rand_fe(rng, enc_len), dtype=np.uint64)
    init = rand_fe(rng, 1)[0]
    d_enc, d_fl, d_ch = zk.DeviceBuffer.from_numpy(enc.T.copy()), zk.DeviceBuffer.from_numpy(flags), zk.DeviceBuffer.from_numpy(ch)
    d_acc, d_scr = zk.DeviceBuffer(n), zk.DeviceBuffer(n + 1024)
    zk.grand_product(d_enc, d_fl, d_ch, enc_len, n, init, d_acc, d_scr); zk.sync()
    exp = oracle.grand_product(enc, flags.astype(np.uint8), ch, init)
    assert np.array_equal(d_acc.to_numpy(), exp)


def test_grand_product_permutation_invariance_at_scale(zk):
    """size-independent property at 2^20 items: the product over a permuted multiset is equal,
    and changes when one element changes (the argument of src/ram_permutation/mod.rs:164-168)."""
    rng = np.random.default_rng(77)
    n, L = 1 << 20, 8
    enc = rng.integers(0, 2**63, size=(L, n), dtype=np.uint64) % np.uint64(P)
    perm = rng.permutation(n)
    ch = np.array([1] + rand_fe(rng, L), dtype=np.uint64)
    ones = np.ones(n, dtype=np.uint64)
    d_fl, d_ch = zk.DeviceBuffer.from_numpy(ones), zk.DeviceBuffer.from_numpy(ch)
    d_acc, d_scr = zk.DeviceBuffer(n), zk.DeviceBuffer(n)

    def final(e):
        d = zk.DeviceBuffer.from_numpy(np.ascontiguousarray(e))
        zk.grand_product(d, d_fl, d_ch, L, n, 1, d_acc, d_scr); zk.sync()
        return int(d_acc.to_numpy()[-1])

    a = final(enc)
    assert final(enc[:, perm]) == a
    enc2 = enc.copy(); enc2[3, 12345] ^= np.uint64(1)
    assert final(enc2) != a
