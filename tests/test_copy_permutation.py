"""K12: copy-permutation grand product.  CPU part: the engine's sigma (zk_cs_sigma, host side) is a permutation
whose cycles are exactly the copy classes the oracle derives from the exported structure alone (copy pairs, links, stream
links).  GPU part (-m gpu): z equals the oracle's integers row by row, closes at 1 on satisfied traces and does not on a
tampered copy."""
import numpy as np
import pytest

from helpers import G, P, ram_cs, random_instances
from oracle import ram_native as rn
from oracle import zko

BETA, GAMMA = (0x1234567890ABCDEF % P, 77), (0xFEDCBA0987654321 % P, 5)


def full_sigma(cs, limit):
    parts = [cs.sigma(False)] + [cs.sigma(True, k) for k in range(limit)]
    return np.concatenate(parts)


def test_sigma_cycles_are_the_copy_classes_ram():
    limit = 5
    cs = ram_cs(limit)
    ho, hl = zko.parse_export(cs.export(False)), zko.parse_export(cs.export(True))
    sigma = full_sigma(cs, limit)
    assert sigma.size == ho["n_trace_cells"] + limit * hl["n_trace_cells"]
    assert zko.sigma_matches_classes(sigma, ho, hl, limit)
    # the links really joined iterations: some loop label maps outside its own iteration
    nto, ntl = ho["n_trace_cells"], hl["n_trace_cells"]
    k1 = sigma[nto + ntl: nto + 2 * ntl]
    assert ((k1 < nto + ntl) | (k1 >= nto + 2 * ntl)).any()
    # and a broken sigma is rejected by the checker itself
    bad = sigma.copy(); bad[[3, 4]] = bad[[4, 3]]
    assert not zko.sigma_matches_classes(bad, ho, hl, limit) or sigma[3] == 3 and sigma[4] == 4


def test_sigma_cycles_are_the_copy_classes_with_stream_links():
    import zkgl
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4), max_trace_len=1 << 26, max_variables=1 << 26)
    cs.configure_eip_4844()
    cs.eip_4844_entry_point(9)     # 9 chunks = 279 bytes: 3 Keccak blocks, chunk and block views tied by a stream link
    cs.pad_and_shrink()
    limit = cs.stats()["limit"]
    ho, hl = zko.parse_export(cs.export(False)), zko.parse_export(cs.export(True))
    assert hl["streams"]
    assert zko.sigma_matches_classes(full_sigma(cs, limit), ho, hl, limit)


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def zk():
    import zkgl
    if zkgl.device_count() == 0:
        pytest.skip("needs a GPU")
    zkgl.init(0)
    return zkgl


@pytest.mark.gpu
def test_gpu_z_equals_oracle_and_detects_broken_copies(zk):
    from test_gpu_cs import gpu_run
    limit, batch, inst = 6, 3, 1
    cs = ram_cs(limit)
    outer, loop = rn.pack_streams(random_instances(31, batch, 5, limit), limit)
    keep = gpu_run(zk, cs, outer, loop, batch)
    assert cs.check_if_satisfied()[0]
    st = cs.stats()
    n_cols, rows = st["copy_columns"] + st["lookup_columns"], st["rows_per_instance"]
    zbuf = zk.DeviceBuffer(batch * (rows + 1) * 2)
    bad, out = cs.copy_permutation(BETA, GAMMA, zbuf)
    assert bad == 0 and np.array_equal(out[:, :2], out[:, 2:])
    z = zbuf.to_numpy().reshape(batch, rows + 1, 2)
    assert all(tuple(int(x) for x in z[i, rows]) == (1, 0) for i in range(batch))
    ho, hl = zko.parse_export(cs.export(False)), zko.parse_export(cs.export(True))
    sigma = full_sigma(cs, limit)
    to, tl = cs.trace(False), cs.trace(True)
    want = zko.copy_permutation_z(ho, hl, limit, to[:, inst], [tl[:, inst * limit + k] for k in range(limit)], sigma, BETA, GAMMA, n_cols)
    assert [tuple(int(x) for x in r) for r in z[inst]] == want
    # the check-only call agrees with the call that also writes z
    bad2, out2 = cs.copy_permutation(BETA, GAMMA)
    assert bad2 == 0 and np.array_equal(out, out2)
    # one copy of a boolean flipped: its gate still holds, the grand product of that instance no longer closes
    for slot, row in enumerate(hl["rows"]):
        if row[0] == G["BOOLEAN"]:
            break
    cell, lane = slot * n_cols, 2 * limit + 3
    in_pairs = [c for pr in hl["copies"] for c in pr]
    assert cell in in_pairs
    old = int(tl[cell, lane])
    cs.write_cell(True, cell, lane, 1 - old)
    bad, out = cs.copy_permutation(BETA, GAMMA)
    assert bad == 1 and not np.array_equal(out[2, :2], out[2, 2:]) and np.array_equal(out[:2, :2], out[:2, 2:])
    cs.write_cell(True, cell, lane, old)
    # a carried value changed between iterations: every gate and in-iteration copy holds, only the link class is broken
    kind, lc, oc = next(l for l in hl["links"] if l[0] == zko.LINK_CARRY and l[1] < hl["n_trace_cells"] and l[2] < hl["n_trace_cells"])
    cls = [c for c, partner in hl["copies"] if partner == lc] + [lc]
    lane = 0 * limit + 2
    olds = [int(tl[c, lane]) for c in cls]
    for c in cls:
        cs.write_cell(True, c, lane, (int(tl[c, lane]) + 1) % P)   # the whole class of the input variable moves together
    bad, out = cs.copy_permutation(BETA, GAMMA)
    assert bad == 1 and not np.array_equal(out[0, :2], out[0, 2:])
    for c, v in zip(cls, olds):
        cs.write_cell(True, c, lane, v)
    assert cs.copy_permutation(BETA, GAMMA)[0] == 0
    del keep


@pytest.mark.gpu
def test_gpu_grand_product_closes_on_the_vm_cycle(zk):
    """main_vm-shaped cycle (BASELINE config C2 at a short limit): 183 carried words per iteration, broadcast imports, lookups"""
    if __import__("helpers").emulated_device() and __import__("os").environ.get("ZKGL_EMU_TORCH") != "1":
        pytest.skip("device memory of this test is a torch CUDA tensor: the hardware, or the emulated device with the torch.cuda stand-ins (tools/emulated_gpu_suite.sh)")
    import torch
    from vm_shaped_fixture import build_vm_cs, vm_inputs
    cs, limit = build_vm_cs(zk, 12)   # 2^12 rows
    n_outer, n_loop = cs.input_words()
    B = 5
    outer, loop = vm_inputs(np.random.default_rng(3), n_outer, n_loop, B, limit)
    cs.set_batch(B)
    d_o = torch.from_numpy(outer.view(np.int64)).cuda()
    d_l = torch.from_numpy(loop.view(np.int64)).cuda()
    cs.bind_inputs(False, d_o, n_outer)
    cs.bind_inputs(True, d_l, n_loop)
    cs.seed_carried_inputs(d_l)
    ok, f = cs.resolve_and_check()
    assert ok, f
    bad, out = cs.copy_permutation(BETA, GAMMA)
    assert bad == 0 and np.array_equal(out[:, :2], out[:, 2:])
    ho, hl = zko.parse_export(cs.export(False)), zko.parse_export(cs.export(True))
    assert zko.sigma_matches_classes(full_sigma(cs, limit), ho, hl, limit)
