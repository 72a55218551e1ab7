"""K11: Goldilocks NTT / coset LDE.  CPU part: the oracle's fast transform is pinned by the definition (Horner
evaluation at every domain point, in C and independently in Python integers) and by algebraic identities.  GPU part (-m gpu):
zk_ntt / zk_lde equal the oracle bit for bit at every size up to 2^16, and satisfy size-independent properties at 2^20-2^22."""
import numpy as np
import pytest

from helpers import P
from oracle import zko

SHIFT = 7  # the multiplicative generator: the coset used by the usual quotient domains


def bitrev(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2) if bits else 0


def rand_poly(rng, *shape):
    return rng.integers(0, P, size=shape, dtype=np.uint64)


# ------------------------------------------------------------------ CPU: oracle pins
def test_two_adic_root_has_exact_order():
    w32 = zko.two_adic_root(32)
    assert w32 == pow(7, (P - 1) >> 32, P)
    assert pow(w32, 1 << 32, P) == 1 and pow(w32, 1 << 31, P) == P - 1
    for n in range(0, 33):
        assert zko.two_adic_root(n) == pow(w32, 1 << (32 - n), P)
    # 2 has order 192 in GF(p): the 64th roots of unity are powers of two, omega_64 = 8^k for an odd k
    w64 = zko.two_adic_root(6)
    assert any(pow(8, k, P) == w64 for k in range(1, 64, 2))


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 7])
@pytest.mark.parametrize("shift", [1, SHIFT, 0x123456789ABCDEF])
def test_fast_transform_equals_the_definition(log_n, shift):
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    a = rand_poly(rng, n)
    w = zko.two_adic_root(log_n)
    # the definition in Python integers
    want = [0] * n
    for k in range(n):
        x = shift * pow(w, k, P) % P
        acc = 0
        for c in reversed([int(v) for v in a]):
            acc = (acc * x + c) % P
        want[bitrev(k, log_n)] = acc
    assert [int(v) for v in zko.ntt_naive(a, shift)] == want
    got = zko.ntt(a, False, shift)
    assert [int(v) for v in got] == want
    assert np.array_equal(zko.ntt(got, True, shift), a)


def test_oracle_identities():
    rng = np.random.default_rng(5)
    a, b = rand_poly(rng, 4, 1 << 10), rand_poly(rng, 4, 1 << 10)
    fa, fb = zko.ntt(a), zko.ntt(b)
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    assert np.array_equal(zko.ntt(s), ((fa.astype(object) + fb.astype(object)) % P).astype(np.uint64))  # linearity
    const = np.zeros(1 << 10, dtype=np.uint64); const[0] = 12345
    assert np.all(zko.ntt(const, False, SHIFT) == 12345)                                                # constant polynomial
    # convolution theorem on a subgroup twice the size: (a * b)(x) = a(x) b(x)
    pa, pb = np.zeros(1 << 11, dtype=np.uint64), np.zeros(1 << 11, dtype=np.uint64)
    pa[: 1 << 10], pb[: 1 << 10] = a[0], b[0]
    prod = (zko.ntt(pa).astype(object) * zko.ntt(pb).astype(object)) % P
    c = zko.ntt(prod.astype(np.uint64), True)
    i = 777
    want = sum(int(a[0][j]) * int(b[0][i - j]) for j in range(i + 1)) % P
    assert int(c[i]) == want
    # LDE blocks == one big transform of the zero-padded coefficients
    big = np.zeros(1 << 13, dtype=np.uint64); big[: 1 << 10] = a[1]
    assert np.array_equal(zko.lde(a[1], 3, SHIFT), zko.ntt(big, False, SHIFT))


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def zk():
    import zkgl
    if zkgl.device_count() == 0:
        pytest.skip("needs a GPU")
    zkgl.init(0)
    return zkgl


def dev_ntt(zk, a, log_n, inverse=False, shift=1, stride=None, natural_values=False):
    stride = stride or (1 << log_n)
    flat = np.zeros(a.shape[0] * stride, dtype=np.uint64)
    for q in range(a.shape[0]):
        flat[q * stride: q * stride + (1 << log_n)] = a[q]
    d = zk.DeviceBuffer.from_numpy(flat)
    zk.ntt(d, log_n, a.shape[0], stride, inverse, shift, None, natural_values)
    zk.sync()
    out = d.to_numpy()
    return np.stack([out[q * stride: q * stride + (1 << log_n)] for q in range(a.shape[0])]), out


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", list(range(0, 17)))
def test_gpu_transform_equals_oracle(zk, log_n):
    rng = np.random.default_rng(200 + log_n)
    n_polys = 3 if log_n < 14 else 2
    a = rand_poly(rng, n_polys, 1 << log_n)
    for shift in (1, SHIFT):
        f, _ = dev_ntt(zk, a, log_n, False, shift)
        assert np.array_equal(f, zko.ntt(a, False, shift)), (log_n, shift)
        b, _ = dev_ntt(zk, f, log_n, True, shift)
        assert np.array_equal(b, a), (log_n, shift)
    back, _ = dev_ntt(zk, a, log_n, True, 0x123456789ABCDEF)
    assert np.array_equal(back, zko.ntt(a, True, 0x123456789ABCDEF))


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [1, 4, 9, 10, 11, 13, 15])
def test_gpu_natural_value_order(zk, log_n):
    """ZK_NTT_NATURAL_VALUES: values at their natural index (trace rows), coefficients at bitrev(i) — the oracle's transform
    with the bit-reversal permutation moved to the other side"""
    rng = np.random.default_rng(400 + log_n)
    n = 1 << log_n
    perm = np.array([bitrev(i, log_n) for i in range(n)])
    rows = rand_poly(rng, 2, n)                                     # natural-order values on SHIFT * <omega>
    coeffs_brev, _ = dev_ntt(zk, rows, log_n, True, SHIFT, None, True)
    want = zko.ntt(rows[:, perm], True, SHIFT)                      # oracle: bit-reversed values -> natural coefficients
    assert np.array_equal(coeffs_brev, want[:, perm])
    back, _ = dev_ntt(zk, coeffs_brev, log_n, False, SHIFT, None, True)
    assert np.array_equal(back, rows)
    # the two orders chain: bit-reversed coefficients -> natural coefficients is just the permutation
    f, _ = dev_ntt(zk, np.ascontiguousarray(coeffs_brev[:, perm]), log_n, False, SHIFT)
    assert np.array_equal(f[:, perm], rows)
    # coset LDE from bit-reversed coefficients: natural-order values of every coset
    src = zk.DeviceBuffer.from_numpy(coeffs_brev.reshape(-1))
    out = zk.DeviceBuffer(2 * (n << 2))
    zk.lde(src, out, log_n, 2, 2, None, 3, None, True)
    zk.sync()
    got = out.to_numpy().reshape(2, 4, n)
    for q in range(2):
        ref = zko.lde(want[q], 2, 3).reshape(4, n)
        assert np.array_equal(got[q], ref[:, perm])


@pytest.mark.gpu
def test_gpu_stride_leaves_the_gaps_alone(zk):
    rng = np.random.default_rng(9)
    log_n, stride = 11, (1 << 11) + 40
    a = rand_poly(rng, 5, 1 << log_n)
    f, raw = dev_ntt(zk, a, log_n, False, SHIFT, stride)
    assert np.array_equal(f, zko.ntt(a, False, SHIFT))
    for q in range(5):
        assert not raw[q * stride + (1 << log_n): (q + 1) * stride].any()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,log_blowup", [(1, 1), (5, 3), (10, 2), (12, 3), (14, 1)])
def test_gpu_lde_equals_oracle_and_the_big_transform(zk, log_n, log_blowup):
    rng = np.random.default_rng(300 + log_n)
    n, n_polys = 1 << log_n, 3
    a = rand_poly(rng, n_polys, n)
    src = zk.DeviceBuffer.from_numpy(a.reshape(-1))
    out = zk.DeviceBuffer(n_polys * (n << log_blowup))
    zk.lde(src, out, log_n, log_blowup, n_polys, None, SHIFT)
    zk.sync()
    got = out.to_numpy().reshape(n_polys, n << log_blowup)
    assert np.array_equal(src.to_numpy().reshape(n_polys, n), a)  # source untouched
    for q in range(n_polys):
        assert np.array_equal(got[q], zko.lde(a[q], log_blowup, SHIFT))
    big = np.zeros((n_polys, n << log_blowup), dtype=np.uint64)
    big[:, :n] = a
    f, _ = dev_ntt(zk, big, log_n + log_blowup, False, SHIFT)
    assert np.array_equal(got, f)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [20, 22])
def test_gpu_full_size_properties(zk, log_n):
    """trace-sized transforms (2^20 rows main_vm / hashes, 2^22 storage): round trip, linearity, oracle on one polynomial"""
    if __import__("helpers").emulated_device() and __import__("os").environ.get("ZKGL_EMU_TORCH") != "1":
        pytest.skip("device memory of this test is a torch CUDA tensor: the hardware, or the emulated device with the torch.cuda stand-ins (tools/emulated_gpu_suite.sh)")
    import torch
    n, n_polys = 1 << log_n, 6
    rng = np.random.default_rng(log_n)
    a = rand_poly(rng, n_polys, n)
    a[1] = 0; a[1][0] = 42                       # constant
    a[2] = ((a[3].astype(object) + a[4].astype(object)) % P).astype(np.uint64)
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    zk.ntt(d, log_n, n_polys, n, False, SHIFT)
    torch.cuda.synchronize()
    f = d.cpu().numpy().view(np.uint64)
    assert np.all(f[1] == 42)
    assert np.array_equal(f[2], ((f[3].astype(object) + f[4].astype(object)) % P).astype(np.uint64))
    assert np.array_equal(f[0], zko.ntt(a[0], False, SHIFT))
    zk.ntt(d, log_n, n_polys, n, True, SHIFT)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), a)


@pytest.mark.gpu
def test_gpu_argument_errors(zk):
    d = zk.DeviceBuffer(16)
    with pytest.raises(zk.ZkError):
        zk.ntt(d, 3, 1, 8, False, 0)          # zero shift
    with pytest.raises(zk.ZkError):
        zk.ntt(d, 3, 2, 4, False, 1)          # stride smaller than the polynomial
    with pytest.raises(zk.ZkError):
        zk.ntt(d, 31, 1, None, False, 1)
    with pytest.raises(zk.ZkError):
        zk._check(zk.lib().zk_ntt(zk._ptr(d), 3, 1, 8, 4, 1, None))   # unknown mode bit


@pytest.mark.gpu
def test_gpu_trace_columns_feed_the_transform(zk):
    """zk_cs_trace_columns lays one instance's resolved trace out as column polynomials (loop rows, outer rows, zero padding);
    interpolation in natural-value order and re-evaluation give the rows back, and every LDE coset of a column is consistent
    with the oracle's transform of the same coefficients."""
    from helpers import ram_cs, random_instances
    from oracle import ram_native as rn
    from test_gpu_cs import gpu_run
    limit, batch, inst = 70, 3, 1                 # 70 iterations: the instance's lanes straddle wave tiles
    cs = ram_cs(limit)
    outer, loop = rn.pack_streams(random_instances(21, batch, 40, limit), limit)
    keep = gpu_run(zk, cs, outer, loop, batch)
    assert cs.check_if_satisfied()[0]
    st = cs.stats()
    n_cols = st["copy_columns"] + st["lookup_columns"]
    S, So, rows = st["loop_slots"], st["outer_slots"], st["rows_per_instance"]
    assert rows == limit * S + So
    log_n = int(rows - 1).bit_length()
    n, stride = 1 << log_n, (1 << log_n) + 24
    out = zk.DeviceBuffer(n_cols * stride)
    out.zero()
    cs.trace_columns(inst, out, log_n, stride)
    zk.sync()
    got = out.to_numpy().reshape(n_cols, stride)
    tl, to = cs.trace(True), cs.trace(False)
    want = np.zeros((n_cols, stride), dtype=np.uint64)
    for c in range(n_cols):
        loop_part = tl[c::n_cols][:S, inst * limit:(inst + 1) * limit]      # [slot, iteration]
        want[c, :limit * S] = loop_part.T.reshape(-1)
        want[c, limit * S: rows] = to[c::n_cols][:So, inst]
    assert np.array_equal(got, want)
    with pytest.raises(zk.ZkError):
        cs.trace_columns(inst, out, log_n - 1, stride)
    with pytest.raises(zk.ZkError):
        cs.trace_columns(batch, out, log_n, stride)
    # the batch form (zk_cs_trace_columns_batch): instances 1..2 in one pass == the per-instance calls; instance 0's lanes share a
    # store tile with instance 1's (70 iterations) and must not leak into the output
    istride = n_cols * stride + 40
    outb = zk.DeviceBuffer(2 * istride)
    outb.zero()
    cs.trace_columns_batch(1, 2, outb, log_n, n_cols, stride, istride)
    zk.sync()
    gb = outb.to_numpy()
    assert np.array_equal(gb[:n_cols * stride].reshape(n_cols, stride), want)
    assert not gb[n_cols * stride:istride].any()
    out2 = zk.DeviceBuffer(n_cols * stride)
    out2.zero()
    cs.trace_columns(2, out2, log_n, stride)
    zk.sync()
    assert np.array_equal(gb[istride:istride + n_cols * stride], out2.to_numpy())
    with pytest.raises(zk.ZkError):
        cs.trace_columns_batch(2, 2, outb, log_n, n_cols, stride, istride)      # past the batch
    with pytest.raises(zk.ZkError):
        cs.trace_columns_batch(0, 2, outb, log_n, n_cols, stride, stride)       # instances would overlap
    # rows -> bit-reversed coefficients -> rows
    zk.ntt(out, log_n, n_cols, stride, True, 1, None, True)
    zk.sync()
    coeffs_brev = out.to_numpy().reshape(n_cols, stride)[:, :n].copy()
    perm = np.array([bitrev(i, log_n) for i in range(n)])
    c0 = 5
    assert np.array_equal(coeffs_brev[c0][perm], zko.ntt(want[c0, :n][perm], True, 1))
    lde_out = zk.DeviceBuffer(n_cols * (n << 1))
    zk.lde(out, lde_out, log_n, 1, n_cols, stride, SHIFT, None, True)
    zk.ntt(out, log_n, n_cols, stride, False, 1, None, True)
    zk.sync()
    assert np.array_equal(out.to_numpy().reshape(n_cols, stride), want)
    ext = lde_out.to_numpy().reshape(n_cols, 2, n)
    ref = zko.lde(coeffs_brev[c0][perm], 1, SHIFT).reshape(2, n)
    assert np.array_equal(ext[c0], ref[:, perm])
    del keep


@pytest.mark.gpu
def test_gpu_degenerate_shapes(zk):
    """empty batches, constants (log_n = 0), blow-up factor 1, a 2^17 transform whose passes split 9 + 8"""
    d = zk.DeviceBuffer.from_numpy(np.array([5, 6, 7, 8], dtype=np.uint64))
    zk.ntt(d, 2, 0, 4, False, 1)                       # no polynomials: nothing happens
    zk.ntt(d, 0, 4, 1, False, SHIFT)                   # four constants: a[0] g^0
    zk.ntt(d, 0, 4, 1, True, SHIFT)
    zk.sync()
    assert d.to_numpy().tolist() == [5, 6, 7, 8]
    rng = np.random.default_rng(77)
    a = rand_poly(rng, 2, 1 << 6)
    src = zk.DeviceBuffer.from_numpy(a.reshape(-1))
    out = zk.DeviceBuffer(2 << 6)
    zk.lde(src, out, 6, 0, 2, None, SHIFT)             # blow-up 1 = the coset transform itself, out of place
    zk.sync()
    assert np.array_equal(out.to_numpy().reshape(2, -1), zko.ntt(a, False, SHIFT))
    with pytest.raises(zk.ZkError):
        zk.lde(src, out, 0, 1, 1, None, 1)
    b = rand_poly(rng, 1, 1 << 17)
    f, _ = dev_ntt(zk, b, 17, False, SHIFT)
    assert np.array_equal(f, zko.ntt(b, False, SHIFT))
    g, _ = dev_ntt(zk, b, 17, True, SHIFT, None, True)
    perm = np.array([bitrev(i, 17) for i in range(1 << 17)])
    assert np.array_equal(g[:, perm], zko.ntt(np.ascontiguousarray(b[:, perm]), True, SHIFT))
