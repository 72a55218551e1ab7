"""(Sorts last, like tests/test_zz_round5_gpu.py: the driver runs `pytest -x`, and a surprise in code no device has run must not hide the established parity
evidence behind the first failure.)

The NARROW STORE of a loop scope (csrc/store_geom.hpp, cs.cpp build_narrow_layout; opt-in per batch: ZKGL_NARROW_STORE=1 at zk_cs_set_batch).

CS::bound_values proves, from the constraints alone, which values are bytes in EVERY satisfying witness (main_vm: 5 103 of a cycle's 17 700);
the narrow layout keeps them in one-byte slots of the store the fused step writes and reads (k_witness_loop_narrow, k_check_prog_t<true>, links),
every other reader sees the ordinary store k_widen_store expands it into.  Checked here:
  host (no GPU): the layout exists, is a prefix sum of the classes, only narrow-capable ops own byte slots, the written bytes fall to <= 0.80 x;
  -m gpu: main_vm over the narrow store == the oracle, cell for cell, for the whole trace (through the widening), same commitments and
          multiplicities as the ordinary store; a value that does not fit its byte slot / a tampered witness gives the ordinary store's
          verdict and failure report; the deferred-Poseidon2 mode on top; wide lane tilings; ram_permutation (reference fixture shape)."""
import numpy as np
import pytest

import vm_programs as vp
import zkgl
from oracle import zko

LIMIT = 32
AW_BYTE = 1 << 28


def _ask(monkeypatch, narrow=True):
    """the narrow store is a decision of zk_cs_set_batch (ZKGL_NARROW_STORE=1); every loop scope that can use it carries the layout"""
    if narrow:
        monkeypatch.setenv("ZKGL_NARROW_STORE", "1")
    else:
        monkeypatch.delenv("ZKGL_NARROW_STORE", raising=False)


# ------------------------------------------------------------------------------------------------ host
def test_main_vm_narrow_layout_meets_the_byte_budget(monkeypatch):
    cs = vp.vm_cs(LIMIT)
    st = cs.stats()
    assert st["store_bytes_per_lane_loop"] == 8 * st["cells_written_loop"]
    assert 0 < st["narrow_store_bytes_per_lane_loop"] <= 0.80 * st["store_bytes_per_lane_loop"], st      # <= 0.80 x
    assert st["narrow_store_bytes_per_lane_loop"] == 8 * (st["cells_written_loop"] - st["narrow_byte_values_loop"]) + st["narrow_byte_values_loop"]
    assert st["narrow_byte_values_loop"] <= st["values_below_2_32_loop"]
    assert st["narrow_store_active"] == 0          # a batch decides (zk_cs_set_batch)
    monkeypatch.setenv("ZKGL_NARROW_STORE", "0")   # at finalize: no layout at all
    cs0 = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), 1 << 22, 1 << 28)
    cs0.configure_main_vm(vp.defs()[0])
    cs0.main_vm_entry_point(4)
    cs0.pad_and_shrink()
    assert cs0.stats()["narrow_store_bytes_per_lane_loop"] == 0 and cs0.narrow_byte_input_words() == []


def test_circuits_without_a_plain_loop_kernel_get_no_narrow_layout(monkeypatch):
    """hash circuits run their loop scope in strand form with macro-ops that stream their own outputs: the layout is not offered"""
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_keccak()
    cs.keccak256_round_function_entry_point(2)
    cs.pad_and_shrink()
    assert cs.stats()["narrow_store_bytes_per_lane_loop"] == 0
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_ram_permutation()
    cs.ram_permutation_entry_point(8)
    cs.pad_and_shrink()
    st = cs.stats()
    assert 0 < st["narrow_store_bytes_per_lane_loop"] < st["store_bytes_per_lane_loop"]


# ------------------------------------------------------------------------------------------------ device
@pytest.fixture(scope="module")
def vm_batch():
    d, D = vp.defs()
    cs = vp.vm_cs(LIMIT)
    outer, loop, commits, info = vp.mixed_batch(cs, D, LIMIT, 64)
    return cs, D, outer, loop, commits, info


def _bind(zk, cs, outer, loop):
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    return d_o, d_l


@pytest.mark.gpu
@pytest.mark.parametrize("tile_log2", [None, "8", "12"], ids=["tiles64", "tiles256", "tiles4096"])
def test_main_vm_over_the_narrow_store_equals_the_oracle(zk, vm_batch, monkeypatch, tile_log2):
    from test_main_vm_host import run_oracle
    cs, D, outer, loop, commits, info = vm_batch
    B = outer.shape[1]
    monkeypatch.setenv("ZKGL_STRANDS", "0")          # the plain loop kernel (what a full batch takes): the narrow store's kernel
    _ask(monkeypatch)
    if tile_log2:
        monkeypatch.setenv("ZKGL_STORE_TILE_LOG2", tile_log2)
    cs.set_batch(B)
    assert cs.stats()["narrow_store_active"] == 1
    raw = loop.copy()
    raw[0:243] = 0
    d_o, d_l = _bind(zk, cs, outer, raw)
    cs.seed_carried_inputs(d_l)
    assert np.array_equal(d_l.to_numpy().reshape(loop.shape), loop)
    before = cs.stats()
    ok, f = cs.resolve_and_check()
    assert ok, f
    st = cs.stats()
    assert st["narrow_steps"] == before["narrow_steps"] + 1 and st["narrow_repeats"] == before["narrow_repeats"]    # ran over the narrow store, nothing repeated
    for i in range(B):
        assert cs.public_inputs(i) == commits[i], info[i]
    run = run_oracle(cs, B)
    run.resolve(outer, loop)
    assert run.check()[0] == 0
    # every cell of the trace — through k_widen_store (the ordinary store is what trace readers address)
    assert np.array_equal(cs.trace(True), run.lc), "loop-scope trace read back from the narrow store differs from the oracle"
    assert np.array_equal(cs.trace(False), run.oc)
    total = run.mult.size // B
    for i in (0, B // 2, B - 1):
        assert np.array_equal(cs.multiplicities(i), run.mult[i * total:(i + 1) * total])
    # the full (stored) check reads the widened values
    ok, f = cs.check_if_satisfied()
    assert ok, f
    # a second step after the readers: narrow again
    ok, f = cs.resolve_and_check()
    assert ok, f
    assert cs.stats()["narrow_steps"] == st["narrow_steps"] + 1


@pytest.mark.gpu
def test_issued_buffer_operations_of_the_two_stores(zk, vm_batch, monkeypatch):
    """EMULATED DEVICE only (it counts the lane-level buffer operations the kernels issue; no such counter exists on hardware, where the PMC passes of
    tools/evidence_r6b.sh read WRITE_SIZE / FETCH_SIZE): one fused step of main_vm over each store.  Every byte-class value is one 1-byte store instead of
    one 8-byte store, every operand read one load of its class — the stored bytes fall by exactly what zk_stats says, and the issued load bytes
    (the upper bound of what a cache-less device would fetch) are reported for profiles/r6_predictions.md."""
    from helpers import emulated_device
    if not emulated_device():
        pytest.skip("lane-level operation counts exist on the emulated device only")
    import ctypes as C
    cs, D, outer, loop, commits, info = vm_batch
    B = outer.shape[1]
    monkeypatch.setenv("ZKGL_STRANDS", "0")

    def counts():
        out = (C.c_ulonglong * 4)()
        zkgl.lib().zk_emu_buffer_ops(out)
        return np.array(list(out), dtype=np.int64)

    ops = {}
    for mode in ("ordinary", "narrow"):
        _ask(monkeypatch, mode == "narrow")
        cs.set_batch(B)
        assert cs.stats()["narrow_store_active"] == (1 if mode == "narrow" else 0)
        keep = _bind(zk, cs, outer, loop)
        c0 = counts()
        ok, f = cs.resolve_and_check()
        assert ok, f
        ops[mode] = counts() - c0
        del keep
    st = cs.stats()
    ld8_o, st8_o, ldb_o, stb_o = (int(x) for x in ops["ordinary"])
    ld8_n, st8_n, ldb_n, stb_n = (int(x) for x in ops["narrow"])
    lanes = -(-B * LIMIT // 64) * 64          # the lanes of the last wavefront beyond the batch redo the last lane's work
    assert ldb_o == 0 and stb_o == 0
    assert stb_n == lanes * st["narrow_byte_values_loop"]                       # one 1-byte store per byte-class value and lane ...
    assert st8_o - st8_n == stb_n                                               # ... in place of one 8-byte store
    assert (st8_o * 8) - (st8_n * 8 + stb_n) == lanes * (st["store_bytes_per_lane_loop"] - st["narrow_store_bytes_per_lane_loop"])
    assert ld8_o == ld8_n + ldb_n                                               # every operand read is one load of the value's class
    print(f"[narrow store, issued by the kernels per lane] stores {st8_o * 8 / lanes:.0f} -> {(st8_n * 8 + stb_n) / lanes:.0f} B "
          f"({(st8_n * 8 + stb_n) / (st8_o * 8):.3f}); operand loads {ld8_o / lanes:.0f} of 8 B -> {ld8_n / lanes:.0f} of 8 B + {ldb_n / lanes:.0f} of 1 B "
          f"({(ld8_n * 8 + ldb_n) / (ld8_o * 8):.3f} of the bytes)")
    _ask(monkeypatch, False)
    cs.set_batch(B)


@pytest.mark.gpu
def test_narrow_store_failures_are_the_ordinary_stores(zk, vm_batch, monkeypatch):
    """a tampered carried word, and an oracle word that does not fit the byte slot its range check gives it (written truncated, it would BE a
    boolean): verdict and failure report of the narrow batch == those of the same circuit recorded without the layout (the step is repeated
    over the ordinary store)"""
    cs, D, outer, loop, commits, info = vm_batch
    B = outer.shape[1]
    monkeypatch.setenv("ZKGL_STRANDS", "0")
    lay = cs.main_vm_layout()["loop"]
    victim = next(i for i, (name, seed, chunk) in enumerate(info) if name == "calls" and chunk == 1)
    bad_a = loop.copy()
    bad_a[lay["state"][0] + 9, victim * LIMIT + 7] ^= 1
    byte_words = cs.narrow_byte_input_words()
    w = lay["src0_read_is_ptr"][0]
    assert w in byte_words, "the range-checked oracle flag is expected in a one-byte slot"
    bad_b = loop.copy()
    bad_b[w, 5 * LIMIT + 3] += 256          # & 0xff it is the boolean the circuit wants: only the overflow test can object
    reports = {}
    c = cs
    for mode in ("narrow", "ordinary"):
        _ask(monkeypatch, mode == "narrow")
        c.set_batch(B)
        assert c.stats()["narrow_store_active"] == (1 if mode == "narrow" else 0)
        for name, bad in (("carried", bad_a), ("byte", bad_b)):
            keep = _bind(zk, c, outer, bad)
            r0 = c.stats()["narrow_repeats"]
            ok, f = c.resolve_and_check()
            assert not ok, (mode, name)
            reports[(mode, name)] = (f.scope, f.instance, f.iteration, f.slot, f.kind, f.relation)
            if mode == "narrow":
                assert c.stats()["narrow_repeats"] == r0 + 1
            del keep
    assert reports[("narrow", "carried")] == reports[("ordinary", "carried")] and reports[("narrow", "carried")][1] == victim
    assert reports[("narrow", "byte")] == reports[("ordinary", "byte")] and reports[("narrow", "byte")][1] == 5


@pytest.mark.gpu
def test_narrow_store_with_deferred_poseidon2_intermediates(zk, vm_batch, monkeypatch):
    """ZK_CHECK_FUSED_DEFER_P2 over the narrow store: the loop kernel leaves the permutations' intermediates out, readers get them from
    k_fill_p2 AFTER the widening — whole trace == the oracle"""
    from test_main_vm_host import run_oracle
    cs, D, outer, loop, commits, info = vm_batch
    B = 16
    o, l = outer[:, :B].copy(), loop[:, :B * LIMIT].copy()
    monkeypatch.setenv("ZKGL_STRANDS", "0")
    _ask(monkeypatch)
    cs.set_batch(B)
    cs.set_check_mode(False, defer_p2=True)
    try:
        _bind(zk, cs, o, l)
        ok, f = cs.resolve_and_check()
        assert ok, f
        assert cs.stats()["narrow_store_active"] == 1
        run = run_oracle(cs, B)
        run.resolve(o, l)
        assert np.array_equal(cs.trace(True), run.lc)
        for i in range(B):
            assert cs.public_inputs(i) == commits[i]
    finally:
        cs.set_check_mode(False)


@pytest.mark.gpu
def test_ram_permutation_over_the_narrow_store(zk, monkeypatch):
    """a queue circuit (two Poseidon2 chains per item, LOOP_LAST values in the outer post phase): k_widen_last feeds the outer scope; the witness
    COLUMNS of the batch are read straight from the narrow store (k_trace_columns_batch decodes address words: no widened copy is made for them)"""
    from helpers import random_instances
    from oracle import ram_native as rn
    monkeypatch.setenv("ZKGL_NARROW_STORE", "1")
    monkeypatch.setenv("ZKGL_STRANDS", "0")
    limit = 70                      # an instance's lanes straddle the 64-lane tiles
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_ram_permutation()
    cs.ram_permutation_entry_point(limit)
    cs.pad_and_shrink()
    insts = random_instances(11, 9, 40, limit)
    insts[3] = rn.instance([], [], limit, 0)                      # empty queue
    outer, loop = rn.pack_streams(insts, limit)
    cs.set_batch(len(insts))
    assert cs.stats()["narrow_store_active"] == 1
    keep = _bind(zk, cs, outer, loop)
    ok, f = cs.resolve_and_check()
    assert ok, f
    for i, inst in enumerate(insts):
        assert cs.public_inputs(i) == inst["commitment"]
    # ---- columns of instances 1..3 while the values are in the narrow store only
    st = cs.stats()
    assert st["narrow_store_pending"] == 1
    n_cols = st["copy_columns"] + st["lookup_columns"]
    S, So, rows = st["loop_slots"], st["outer_slots"], st["rows_per_instance"]
    log_n = int(rows - 1).bit_length()
    stride = (1 << log_n) + 8
    istride = n_cols * stride
    outb = zk.DeviceBuffer(3 * istride)
    outb.zero()
    cs.trace_columns_batch(1, 3, outb, log_n, n_cols, stride, istride)
    zk.sync()
    assert cs.stats()["narrow_store_pending"] == 1, "reading the columns expanded the narrow store"
    got = outb.to_numpy().reshape(3, n_cols, stride)
    # ---- every cell against the oracle (the trace readers below DO expand the store: the ordinary one is what they address)
    run = zko.CircuitRun(cs.export(False), cs.export(True), len(insts), 65536)
    run.resolve(outer, loop)
    tl, to = cs.trace(True), cs.trace(False)
    assert cs.stats()["narrow_store_pending"] == 0
    assert np.array_equal(tl, run.lc) and np.array_equal(to, run.oc)
    for k, inst in enumerate((1, 2, 3)):
        want = np.zeros((n_cols, stride), dtype=np.uint64)
        for c in range(n_cols):
            want[c, :limit * S] = tl[c::n_cols][:S, inst * limit:(inst + 1) * limit].T.reshape(-1)
            want[c, limit * S: rows] = to[c::n_cols][:So, inst]
        assert np.array_equal(got[k], want), inst
    del keep


@pytest.mark.gpu
def test_c2_main_vm_2_20_rows_over_the_narrow_store(zk, monkeypatch):
    """BASELINE's C2 at full size (2^20 rows per instance, the bench fixture through zk_pack_main_vm_witness) with the batch on the narrow store: commitments ==
    the fixture's, and the WHOLE trace of one instance — 164 columns x 2^20 rows, read by zk_cs_trace_columns straight from the one-byte / eight-byte slots —
    == the oracle interpreter's, cell for cell; then the deferred-Poseidon2 mode on top (columns after widening + k_fill_p2)"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    import test_gpu_full_size as T
    monkeypatch.setenv("ZKGL_NARROW_STORE", "1")
    cs, limit = bench.build_main_vm_cs(zkgl, 20)
    B = 4
    outer, loop, expect = bench.main_vm_streams(zkgl, cs, limit, B)
    assert expect is not None
    d_o, d_l = zk.DeviceBuffer.from_numpy(outer), zk.DeviceBuffer.from_numpy(loop)
    cs.seed_stream(B, d_o, d_l)
    seeded = d_l.to_numpy().reshape(loop.shape)
    cs.set_batch(B)
    st = cs.stats()
    assert st["narrow_store_active"] == 1 and st["narrow_store_bytes_per_lane_loop"] <= 0.80 * st["store_bytes_per_lane_loop"]
    cs.bind_inputs(False, d_o, outer.shape[0]); cs.bind_inputs(True, d_l, loop.shape[0])
    ok, f = cs.resolve_and_check()
    assert ok, f
    assert cs.stats()["narrow_repeats"] == 0 and cs.stats()["narrow_store_pending"] == 1
    for i in range(B):
        assert cs.public_inputs(i) == [int(x) for x in expect[i]], i
    T.assert_whole_trace_equals_oracle(zk, cs, outer, seeded, 1, limit, 20)
    assert cs.stats()["narrow_store_pending"] == 1, "the columns were not read from the narrow store"
    cs.set_check_mode(False, defer_p2=True)
    try:
        ok, f = cs.resolve_and_check()
        assert ok, f
        T.assert_whole_trace_equals_oracle(zk, cs, outer, seeded, 2, limit, 20)
        assert cs.stats()["narrow_store_pending"] == 0      # the fill works in the ordinary store: widened first
    finally:
        cs.set_check_mode(False)
    cs.close()
