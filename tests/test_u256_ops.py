"""ZK_OP_U256_MULWIDE / ZK_OP_U256_DIVREM — the witness ops behind the VM's mul / div / shift closures
(/root/reference/src/main_vm/opcodes/mul_div.rs:20-172: U256::full_mul, U256::div_mod with the b == 0 convention).
Oracle interpreter (Knuth D, row-wise product) and device (bit-serial division, column-wise product) use different
algorithms; both are compared with Python integers."""
import numpy as np
import pytest

import zkgl
from helpers import G, OP, Rec
from oracle import zko


def u256_ops_cs():
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(30, 0, 4, 4))
    for k in ("CONST", "FMA"):
        cs.allow_gate(G[k])
    r = Rec(cs)
    a = [r.inp() for _ in range(8)]
    b = [r.inp() for _ in range(8)]
    prod = cs.alloc_multiple_variables_without_values(16)
    cs.emit_op(OP["U256_MULWIDE"], a + b, prod)
    qr = cs.alloc_multiple_variables_without_values(16)
    cs.emit_op(OP["U256_DIVREM"], a + b, qr)
    cs.pad_and_shrink()
    return cs, prod, qr


def limbs(x, n=8):
    return [(x >> (32 * i)) & 0xffffffff for i in range(n)]


def u256_ops_inputs(B=96):
    rng = np.random.default_rng(2301)
    vals = []
    M = (1 << 256) - 1
    special = [(0, 0), (M, M), (M, 1), (1, M), (M, 0), (12345, 0), (0, 7), (M, (1 << 255) + 1), (M - 1, M), (1 << 255, 1 << 255),
               ((1 << 256) - (1 << 128), (1 << 128) + 1), (3 << 200, 1 << 64), ((1 << 224) - 1, (1 << 32) - 1), (M, (1 << 33) - 1),
               (0x8000000000000000_0000000000000000_0000000000000000_0000000000000000, 0x80000000_00000000_00000000)]
    vals += special
    while len(vals) < B:
        ba, bb = int(rng.integers(1, 257)), int(rng.integers(1, 257))
        a = int.from_bytes(rng.bytes(32), "little") >> (256 - ba)
        b = int.from_bytes(rng.bytes(32), "little") >> (256 - bb)
        vals.append((a, b))
    inp = np.zeros((16, B), dtype=np.uint64)
    for i, (a, b) in enumerate(vals):
        inp[:8, i] = limbs(a)
        inp[8:, i] = limbs(b)
    return inp, vals


def expected(a, b):
    q, r = (0, a) if b == 0 else divmod(a, b)
    return limbs(a * b, 16), limbs(q) + limbs(r)


def test_u256_ops_oracle_vs_python():
    cs, prod, qr = u256_ops_cs()
    inp, vals = u256_ops_inputs()
    B = inp.shape[1]
    run_ = zko.CircuitRun(cs.export(False), cs.export(True), B, 0)
    run_.resolve(inp, np.zeros((0, 0), dtype=np.uint64))
    for i, (a, b) in enumerate(vals):
        e_mul, e_div = expected(a, b)
        assert [int(run_.oc[cs.var_cell(v), i]) for v in prod] == e_mul, (hex(a), hex(b))
        assert [int(run_.oc[cs.var_cell(v), i]) for v in qr] == e_div, (hex(a), hex(b))


@pytest.mark.gpu
def test_u256_ops_gpu(zk):
    cs, prod, qr = u256_ops_cs()
    inp, vals = u256_ops_inputs(320)
    cs.set_batch(inp.shape[1])
    d = zk.DeviceBuffer.from_numpy(inp)
    cs.bind_inputs(False, d, inp.shape[0])
    cs.resolve()
    tr = cs.trace(False)
    for i, (a, b) in enumerate(vals):
        e_mul, e_div = expected(a, b)
        assert [int(tr[cs.var_cell(v), i]) for v in prod] == e_mul, (hex(a), hex(b))
        assert [int(tr[cs.var_cell(v), i]) for v in qr] == e_div, (hex(a), hex(b))
