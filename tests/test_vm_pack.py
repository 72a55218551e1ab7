"""Product-side input path of main_vm: zk_pack_main_vm_witness takes what a host of the reference has — the
VmCircuitWitness: closed-form input + the WitnessOracle's per-getter FIFOs (src/main_vm/witness_oracle.rs:45-91,
src/fsm_input_output/circuit_inputs/main_vm.rs:64-71) — and writes the circuit's input streams.  The native walker behind it
(csrc/vm_native.hpp, also phase A of the device seeding) must place every answer at the cycle that asks for it, and — with
ZK_VM_PACK_FILL_STATE — reproduce the VmLocalState of every cycle.  The oracle (oracle/main_vm_native.py) only compares."""
import numpy as np
import pytest

import zkgl
import vm_programs as vp
from oracle import main_vm_native as vn


def _programs(D):
    out = []
    for seed in (0, 1):
        out.append(("arith", seed, vp.program_arith(D, seed), None))
        out.append(("memory", seed, vp.program_memory_and_logs(D, seed), None))
        out.append(("logs", seed, vp.program_logs(D, seed), None))
        ops, contracts = vp.program_calls(D, seed)
        out.append(("calls", seed, ops, contracts))
    return out


def _first_difference(a, b, lay):
    bad = np.argwhere(a != b)
    w, col = bad[0]
    name = next((n for n, (f, k) in lay["loop"].items() if f <= w < f + k), "?")
    return f"{len(bad)} words differ, first at word {w} ({name}+{w - lay['loop'].get(name, (0, 0))[0]}) of column {col}: packer {int(a[w, col])}, native {int(b[w, col])}"


@pytest.mark.parametrize("fill_state", [False, True])
def test_packer_places_every_oracle_answer_at_its_cycle(fill_state):
    d, D = vp.defs()
    limit = 16
    cs = vp.vm_cs(limit)
    lay = cs.main_vm_layout()
    for name, seed, ops, contracts in _programs(D):
        probe = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), 4 * len(ops) + 64)
        done_at = next(i for i, s in enumerate(probe.states) if s.depth == 0)
        n_inst = (done_at + 1 + limit - 1) // limit + 1   # one chunk past the end: skipped cycles (empty callstack)
        run = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), n_inst * limit)
        want_outer, want_loop = vp.pack_instance_streams(cs, D, run, limit, n_inst)
        outer, loop, reports = vp.pack_through_the_c_abi(cs, run, limit, n_inst, fill_state=fill_state)
        assert not any(r.underflow for r in reports), (name, seed)
        assert np.array_equal(outer, want_outer), (name, seed)
        if not fill_state:
            want_loop = want_loop.copy()
            want_loop[0:243] = 0
        assert np.array_equal(loop, want_loop), (name, seed, _first_difference(loop, want_loop, lay))
        if fill_state:   # hidden_fsm_output of every chunk == the native state after its last cycle
            for i, r in enumerate(reports):
                assert list(r.final_state) == [int(x) for x in run.states[(i + 1) * limit].flatten()], (name, seed, i)
        # every FIFO consumed exactly
        q = vp.oracle_queues(run, 0, n_inst * limit)
        used = [sum(getattr(r, f) for r in reports) for f in ("used_memory_reads", "used_storage_reads", "used_refunds", "used_rollback_queue_witness",
                                                                "used_rollback_tails_for_call", "used_callstack", "used_decommit_pages")]
        assert used == [len(q.memory_reads), len(q.storage_reads), len(q.refunds), len(q.rollback_queue_witness), len(q.rollback_tails_for_call),
                        len(q.callstack), len(q.decommit_pages)], (name, seed)


def test_packer_with_the_queue_states_of_the_witness_hashes_only_the_callstack_pushes():
    """zk_pack_main_vm_witness_states: a first pass plays the witness generator (FILL_STATE | RECORD_STATES: the memory / decommitment /
    forward-log queue tails after every push are written out), a second pass reads them (STATES_FROM_WITNESS) — the same 243 state words
    of every cycle as the native VM, with only the callstack sponge of a call (4 permutations) and of the bootloader frame hashed"""
    d, D = vp.defs()
    limit = 16
    cs = vp.vm_cs(limit)
    lay = cs.main_vm_layout()
    total_hash, total_read = 0, 0
    for name, seed, ops, contracts in _programs(D):
        probe = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), 4 * len(ops) + 64)
        done_at = next(i for i, s in enumerate(probe.states) if s.depth == 0)
        n_inst = (done_at + 1 + limit - 1) // limit + 1
        run = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), n_inst * limit)
        want_outer, want_loop = vp.pack_instance_streams(cs, D, run, limit, n_inst)
        ow, lw = cs.input_words()
        q = vp.oracle_queues(run, 0, n_inst * limit)
        recorded = []
        for mode in ("record", "read"):
            outer = np.zeros((ow, n_inst), dtype=np.uint64); loop = np.zeros((lw, n_inst * limit), dtype=np.uint64)
            used = [0] * 7
            for i in range(n_inst):
                if mode == "record":
                    arrs = (np.zeros((8 * limit, 12), dtype=np.uint64), np.zeros((2 * limit, 12), dtype=np.uint64), np.zeros((2 * limit, 4), dtype=np.uint64))
                    st = zkgl.VmQueueStates.over(*arrs)
                    flags = zkgl.VM_PACK_FILL_STATE | zkgl.VM_PACK_RECORD_STATES
                else:
                    arrs, n_used = recorded[i]
                    st = zkgl.VmQueueStates.over(*[np.ascontiguousarray(a[:k]) for a, k in zip(arrs, n_used)])   # exactly what was pushed: nothing to spare
                    flags = zkgl.VM_PACK_STATES_FROM_WITNESS
                rep = cs.pack_main_vm_witness_states(vp.closed_form_input(run, i * limit), q.view(used), st, i, n_inst, outer, loop, flags)
                assert not rep.underflow, (name, seed, mode, i)
                got = [rep.used_memory_reads, rep.used_storage_reads, rep.used_refunds, rep.used_rollback_queue_witness, rep.used_rollback_tails_for_call,
                       rep.used_callstack, rep.used_decommit_pages]
                used = [a + b for a, b in zip(used, got)]
                n_used = (st.used_memory_tails, st.used_decommit_tails, st.used_log_forward_tails)
                if mode == "record":
                    recorded.append((arrs, n_used))
                    total_hash += st.host_permutations
                else:
                    assert n_used == recorded[i][1], (name, seed, i)
                    total_read += st.host_permutations
                    # what is still hashed: the callstack sponge — four permutations per pushed frame (+ the bootloader's formal frame)
                    assert st.host_permutations % 4 == 0 and st.host_permutations <= 4 * (limit + 1), (name, seed, i, st.host_permutations)
                assert list(rep.final_state) == [int(x) for x in run.states[(i + 1) * limit].flatten()], (name, seed, mode, i)
            assert np.array_equal(outer, want_outer), (name, seed, mode)
            assert np.array_equal(loop, want_loop), (name, seed, mode, _first_difference(loop, want_loop, lay))
    assert total_read * 4 < total_hash, (total_read, total_hash)


def test_packer_reports_underflow_and_rejects_foreign_circuits():
    d, D = vp.defs()
    limit = 16
    cs = vp.vm_cs(limit)
    ops = vp.program_memory_and_logs(D, 0)
    run = vn.VmRun(D, vp.make_world_factory(D, ops), limit)
    q = vp.oracle_queues(run, 0, limit)
    assert len(q.memory_reads) > 2
    q.memory_reads = q.memory_reads[:2]
    q.freeze()
    ow, lw = cs.input_words()
    outer, loop = np.zeros((ow, 1), dtype=np.uint64), np.zeros((lw, limit), dtype=np.uint64)
    rep = cs.pack_main_vm_witness(vp.closed_form_input(run, 0), q.view(), 0, 1, outer, loop)
    assert rep.underflow == 1 and rep.used_memory_reads == 2
    other = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    other.configure_ram_permutation()
    other.ram_permutation_entry_point(4)
    other.pad_and_shrink()
    with pytest.raises(zkgl.ZkError):
        other.pack_main_vm_witness(vp.closed_form_input(run, 0), q.view(), 0, 1, outer, loop)


def test_bench_fixture_packs_without_underflow():
    """tests/golden/vm_bench_witness.npz (64 executions as WitnessOracle FIFOs) -> streams: every FIFO exactly consumed, 64 distinct
    executions, raw words only (bench.py and tests/test_gpu_full_size.py feed the device from this)"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    cs, limit = bench.build_main_vm_cs(zkgl, 20)
    outer, loop, expect = bench.main_vm_streams(zkgl, cs, limit, 16)
    assert expect is not None and outer.shape[1] == 16 and loop.shape[1] == 16 * limit
    assert not loop[:243].any()
    lay = cs.main_vm_layout()["loop"]
    f, n = lay["code_word"]
    assert len({loop[f:f + n, e * limit:(e + 1) * limit].tobytes() for e in range(16)}) == 16


# ---- VmCircuitInputOutputWitness from bincode bytes (test-side writer in serde's derive order, C decoder)
def _vm_state_bincode(st):
    import struct
    f = [int(x) for x in st.flatten()]
    u256 = lambda limbs: (lambda s: struct.pack("<Q", len(s)) + s)(("0x%x" % sum(v << (32 * i) for i, v in enumerate(limbs))).encode())
    h160 = lambda limbs: struct.pack("<Q", 42) + ("0x%040x" % sum(v << (32 * i) for i, v in enumerate(limbs))).encode()
    u8, u16, u32, u64 = (lambda v: struct.pack("<B", v)), (lambda v: struct.pack("<H", v)), (lambda v: struct.pack("<I", v)), (lambda v: struct.pack("<Q", v))
    o, n = b"", 0
    o += u256(f[0:8]); n = 8
    for r in range(15):
        o += u8(f[n]) + u256(f[n + 1:n + 9]); n += 9
    o += b"".join(u8(x) for x in f[n:n + 3]); n += 3
    o += b"".join(u32(x) for x in f[n:n + 4]); n += 4
    o += u16(f[n]) + u8(f[n + 1]) + u32(f[n + 2]); n += 3
    for a in range(3):
        o += h160(f[n:n + 5]); n += 5
    o += b"".join(u32(x) for x in f[n:n + 4]); n += 4
    o += b"".join(u64(x) for x in f[n:n + 8]); n += 8
    o += u32(f[n]); n += 1
    o += b"".join(u16(x) for x in f[n:n + 3]); n += 3
    o += u32(f[n]); n += 1
    o += u8(f[n]) + u8(f[n + 1]); n += 2
    o += b"".join(u8(x) for x in f[n:n + 3]); n += 3
    o += b"".join(u32(x) for x in f[n:n + 4]); n += 4
    o += u8(f[n]); n += 1
    o += b"".join(u64(x) for x in f[n:n + 4]); n += 4
    o += u32(f[n]) + u32(f[n + 1]); n += 2
    o += b"".join(u64(x) for x in f[n:n + 12]); n += 12
    o += b"".join(u64(x) for x in f[n:n + 12]) + u32(f[n + 12]); n += 13
    o += b"".join(u64(x) for x in f[n:n + 12]) + u32(f[n + 12]); n += 13
    o += b"".join(u32(x) for x in f[n:n + 4]); n += 4
    assert n == 243
    return o


def test_vm_closed_form_input_bincode_decoder():
    import struct
    d, D = vp.defs()
    ops, contracts = vp.program_calls(D, 1)
    limit = 16
    run = vn.VmRun(D, vp.make_world_factory(D, ops, contracts), 3 * limit, default_aa=0x0100000000000000000000000000000000000000000000000000000000001234)
    c0 = limit                                   # the second chunk: a non-trivial hidden_fsm_input
    st_in, st_out = run.states[c0], run.states[c0 + limit]
    u64 = lambda v: struct.pack("<Q", int(v))
    aa = ("0x%x" % run.gctx[1]).encode()
    data = struct.pack("<BB", 0, 0)                                                            # start_flag, completion_flag
    data += b"".join(u64(x) for x in run.rollback_tail_for_block)                                # VmInputData
    data += b"".join(u64(0) for _ in range(12)) + struct.pack("<I", 0)
    data += b"".join(u64(0) for _ in range(12)) + struct.pack("<I", 0)
    data += struct.pack("<B", run.gctx[0]) + struct.pack("<Q", len(aa)) + aa
    data += b"".join(u64(0) for _ in range(8)) + struct.pack("<I", 0)                            # VmOutputData: placeholders
    data += (b"".join(u64(0) for _ in range(24)) + struct.pack("<I", 0)) * 2
    data += _vm_state_bincode(st_in) + _vm_state_bincode(st_out)
    cf, rest, used = zkgl.decode_vm_closed_form_input_bincode(data + b"xyz")
    assert used == len(data)
    want = vp.closed_form_input(run, c0)
    for name in ("start_flag", "zkporter_is_available", "memory_queue_initial_length"):
        assert getattr(cf, name) == getattr(want, name)
    assert list(cf.rollback_queue_tail_for_block) == list(want.rollback_queue_tail_for_block)
    assert list(cf.default_aa_code_hash) == list(want.default_aa_code_hash)
    assert list(cf.hidden_fsm_input) == list(want.hidden_fsm_input) == [int(x) for x in st_in.flatten()]
    assert list(rest.hidden_fsm_output) == [int(x) for x in st_out.flatten()]
    with pytest.raises(zkgl.ZkError):
        zkgl.decode_vm_closed_form_input_bincode(data[:-7])
    # decoded input + the oracle FIFOs -> the same streams as the directly built closed-form input
    cs = vp.vm_cs(limit)
    ow, lw = cs.input_words()
    q = vp.oracle_queues(run, c0, limit)
    o1, l1 = np.zeros((ow, 1), dtype=np.uint64), np.zeros((lw, limit), dtype=np.uint64)
    o2, l2 = np.zeros((ow, 1), dtype=np.uint64), np.zeros((lw, limit), dtype=np.uint64)
    cs.pack_main_vm_witness(cf, q.view(), 0, 1, o1, l1)
    cs.pack_main_vm_witness(want, q.view(), 0, 1, o2, l2)
    assert np.array_equal(o1, o2) and np.array_equal(l1, l2)


def test_batch_packer_on_host_threads_equals_the_per_instance_packer():
    """zk_pack_main_vm_witness_batch (host pool, include/zkgl_witness.h zk_parallel_for): the same words as one zk_pack_main_vm_witness
    per chunk, whatever the thread count; a failing chunk is named"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    cs, limit = bench.build_main_vm_cs(zkgl, 16)
    fx = np.load(bench.FIXTURE)
    E, B = 6, 9                                     # chunks land in instances 2..7 of a batch of 9
    n_outer, n_loop = cs.input_words()
    cfs, queues = bench.fixture_witnesses(zkgl, fx, E)
    want_o = np.zeros((n_outer, B), dtype=np.uint64); want_l = np.zeros((n_loop, B * limit), dtype=np.uint64)
    want_reps = [cs.pack_main_vm_witness(cfs[e], queues[e].view(), 2 + e, B, want_o, want_l) for e in range(E)]
    for threads in (1, 3, 0):
        o = np.zeros_like(want_o); l = np.zeros_like(want_l)
        reps = cs.pack_main_vm_witness_batch(cfs, [q.view() for q in queues], 2, B, o, l, n_threads=threads)
        assert np.array_equal(o, want_o) and np.array_equal(l, want_l)
        for a, b in zip(reps, want_reps):
            assert bytes(a) == bytes(b)
    # with the host-side chains too (FILL_STATE): still the same words
    fo = np.zeros_like(want_o); fl = np.zeros_like(want_l)
    for e in range(E):
        cs.pack_main_vm_witness(cfs[e], queues[e].view(), 2 + e, B, fo, fl, zkgl.VM_PACK_FILL_STATE)
    o = np.zeros_like(want_o); l = np.zeros_like(want_l)
    cs.pack_main_vm_witness_batch(cfs, [q.view() for q in queues], 2, B, o, l, flags=zkgl.VM_PACK_FILL_STATE, n_threads=4)
    assert np.array_equal(o, fo) and np.array_equal(l, fl)
    # out of range: refused before any thread starts
    with pytest.raises(zkgl.ZkError):
        cs.pack_main_vm_witness_batch(cfs, [q.view() for q in queues], B - 2, B, o, l)


def test_parallel_for_runs_every_job_and_names_the_lowest_failure():
    C = zkgl.C
    seen = (C.c_uint32 * 40)()
    JOB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32)

    def job(ctx, j):
        seen[j] += 1
        return 0 if j not in (7, 31) else 2    # ZK_ERR_INVALID-like codes
    first = C.c_uint32()
    rc = zkgl.lib().zk_parallel_for(40, 5, JOB(job), None, C.byref(first))
    assert rc == 2 and first.value == 7 and list(seen) == [1] * 40
    assert zkgl.lib().zk_last_error().decode().startswith("job 7: ")
    rc = zkgl.lib().zk_parallel_for(0, 0, JOB(job), None, C.byref(first))
    assert rc == 0 and first.value == 0xFFFFFFFF
    assert zkgl.lib().zk_host_threads() >= 1


def test_oracle_words_only_packs_exactly_the_rows_behind_the_vm_state():
    """ZK_VM_PACK_ORACLE_WORDS_ONLY: the array starts at row 243 of the loop stream; its content equals rows 243.. of the full packer's"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    cs, limit = bench.build_main_vm_cs(zkgl, 16)
    fx = np.load(bench.FIXTURE)
    E = 5
    n_outer, n_loop = cs.input_words()
    cfs, queues = bench.fixture_witnesses(zkgl, fx, E)
    views = [q.view() for q in queues]
    full_o = np.zeros((n_outer, E), dtype=np.uint64); full_l = np.zeros((n_loop, E * limit), dtype=np.uint64)
    cs.pack_main_vm_witness_batch(cfs, views, 0, E, full_o, full_l, n_threads=2)
    assert not full_l[:243].any()
    o = np.zeros_like(full_o); l = np.full((n_loop - 243, E * limit), 0xdead, dtype=np.uint64)
    reps = cs.pack_main_vm_witness_batch(cfs, views, 0, E, o, l, flags=zkgl.VM_PACK_ORACLE_WORDS_ONLY, n_threads=3)
    assert np.array_equal(o, full_o) and np.array_equal(l, full_l[243:]) and not any(r.underflow for r in reps)
    with pytest.raises(zkgl.ZkError):   # the state rows are the device seeder's in this mode
        cs.pack_main_vm_witness_batch(cfs, views, 0, E, o, l, flags=zkgl.VM_PACK_ORACLE_WORDS_ONLY | zkgl.VM_PACK_FILL_STATE)
    # the tile leaves with non-temporal stores for whole cache lines and plain stores for the partial lines at the ends of a run: every
    # alignment of the staging array (pinned memory: 0; a numpy array: 16 bytes into a line; the worst case: 56) gives the same words
    rows, cols = n_loop - 243, E * limit
    for off_words in (0, 1, 2, 5, 7):
        raw = np.full(rows * cols + 16, 0xbeef, dtype=np.uint64)
        base = (-raw.ctypes.data % 64) // 8 + off_words
        dst = raw[base:base + rows * cols].reshape(rows, cols)
        assert dst.ctypes.data % 64 == 8 * off_words
        cs.pack_main_vm_witness_batch(cfs, views, 0, E, o, dst, flags=zkgl.VM_PACK_ORACLE_WORDS_ONLY, n_threads=1)
        assert np.array_equal(dst, full_l[243:]), off_words
        assert (raw[:base] == 0xbeef).all() and (raw[base + rows * cols:] == 0xbeef).all()   # nothing outside the array


@pytest.mark.parametrize("mode", ["device_seeds", "states_from_witness"])
def test_bench_host_feed_packs_the_words_the_resident_stream_holds(mode):
    """bench.py's HostFeed (the host half of value_including_host_pack / value_states_from_witness): window k of the stream, packed on
    the host pool, equals the rows of the same instances packed one by one — the raw rows (device_seeds) or all 360 with the host-side
    chains (states_from_witness: the queue tails are READ from the recorded queue states, only callstack pushes are hashed)"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    cs, limit = bench.build_main_vm_cs(zkgl, 16)
    B, K, first = 5, 3, 7
    feed = bench.HostFeed(zkgl, cs, limit, B, bench.FIXTURE, mode, n_threads=3, first_execution=first)
    n_outer, n_loop = cs.input_words()
    so = np.zeros((n_outer, B), dtype=np.uint64); sl = np.zeros((feed.rows(), B * limit), dtype=np.uint64)
    for k in (0, 2):
        assert feed.pack_window(k, so, sl) > 0
        wo = np.zeros((n_outer, B), dtype=np.uint64); wl = np.zeros((n_loop, B * limit), dtype=np.uint64)
        for j in range(B):
            e = (first + k * B + j) % feed.n_exec
            cs.pack_main_vm_witness(feed.cfs[e], feed.views[e], j, B, wo, wl, zkgl.VM_PACK_FILL_STATE if mode == "states_from_witness" else 0)
        assert np.array_equal(so, wo)
        assert np.array_equal(sl, wl[feed.first_row:])
    assert feed.rows() == (n_loop - 243 if mode == "device_seeds" else n_loop)
