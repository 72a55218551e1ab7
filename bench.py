#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: constraints/s (+ witness-rows/s) of main_vm at 2^20 rows per instance
(config C2), N MI355X, one process per GPU.

Workload = the REAL main_vm circuit (csrc/circuits/main_vm.cpp: vm_cycle of /root/reference/src/main_vm/cycle.rs:28-795 with
all eleven opcode families) executing synthetic zkEVM programs.  tests/golden/vm_bench_witness.npz holds the VmCircuitWitness of 64
distinct executions of an endless mixed program (far calls, returns, reverts, UMA, logs, arithmetic; generator:
tests/golden/make_vm_bench_witness.py) the way a host of the reference holds it: the WitnessOracle's per-getter FIFOs.  The
product's own packer (zk_pack_main_vm_witness) turns them into the circuit's input streams; instance i of a rank replays execution
(rank * S + i) mod 64.  Instance = one `limit`-cycle chunk filling 2^20 trace rows.

A "step" = one pass of the WHOLE hot path over one batch of B independent circuit instances per GPU whose RAW witness (oracle
words placed at their cycles, no VM state) already lives in HBM:
  * seeding — the sequential half: the per-cycle VmLocalState from the raw words (native walker + Poseidon2 chains + fill,
    zk_cs_seed_window_async) of the NEXT window of the stream, on a second HIP stream;
  * witness generation (outer pre, loop, outer post kernels) + the full satisfiability check (gate + lookup + copy + link kernels)
    of THIS window (zk_cs_resolve_and_check).
`value` = constraints / time of K such steps: from the raw witness, seeding inside the timed region.  `value_inputs_resident` (the
figure rounds 1-2 quoted) is measured right after it without the seeding, `value_from_raw_witness_serial` counts the seeding of
a window as if nothing overlapped.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--log2-rows 20] [--no-cpu-baseline]
(--gpus N > 1 without a launcher re-executes itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (without this RCCL's hipIpcGetMemHandle fails)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
VM_STATE_WORDS = 243   # VmLocalState, the carried part of the loop stream
FIXTURES = {"default": os.path.join(ROOT, "tests", "golden", "vm_bench_witness.npz"),
            # the compiled-contract-like opcode mix (tests/vm_programs.py program_bench_loop(realistic=True)): ~1 % logs, ~0.5 % calls,
            # ~10 % heap accesses, the rest arithmetic / jumps / stack traffic; same circuit, same limit
            "realistic": os.path.join(ROOT, "tests", "golden", "vm_bench_witness_realistic.npz")}
FIXTURE = FIXTURES["default"]


# ------------------------------------------------------------------------------------------------ main_vm workload
def build_main_vm_cs(zkgl, log2_rows):
    """record the cycle once, with the largest `limit` that fits 2^log2_rows trace rows"""
    probe = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), max_trace_len=1 << 30, max_variables=1 << 28)
    probe.configure_main_vm()
    probe.main_vm_entry_point(1)
    probe.pad_and_shrink()
    st = probe.stats()
    limit = ((1 << log2_rows) - st["outer_slots"]) // st["loop_slots"]
    probe.close()
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), max_trace_len=1 << log2_rows, max_variables=1 << 28)  # src/main_vm/cycle.rs:959-966
    cs.configure_main_vm()
    cs.main_vm_entry_point(limit)
    cs.pad_and_shrink()
    return cs, limit


_FIFOS = ("memory_reads", "storage_reads", "refunds", "rollback_queue_witness", "rollback_tails_for_call", "callstack", "decommit_pages")


def fixture_witnesses(zkgl, fx, E):
    """the fixture's first E VmCircuitWitnesses as the C ABI's structs: (closed-form inputs, WitnessOracle FIFO holders)"""
    arr = {k: fx[k] for k in _FIFOS}
    off = {k: fx[k + "_offsets"] for k in _FIFOS}
    cfs, queues = [], []
    for e in range(E):
        q = zkgl.VmOracleQueues()
        sl = {k: arr[k][off[k][e]:off[k][e + 1]] for k in _FIFOS}
        q.memory_reads = [(r[:8], r[8]) for r in sl["memory_reads"]]
        q.storage_reads = list(sl["storage_reads"])
        q.refunds = [r[0] for r in sl["refunds"]]
        q.rollback_queue_witness = list(sl["rollback_queue_witness"])
        q.rollback_tails_for_call = list(sl["rollback_tails_for_call"])
        q.callstack = [(r[:42], r[42:]) for r in sl["callstack"]]
        q.decommit_pages = [r[0] for r in sl["decommit_pages"]]
        q.freeze()
        cf = zkgl.VmClosedFormInput()
        cf.start_flag = 1
        cf.rollback_queue_tail_for_block[:] = [int(x) for x in fx["rollback_tail"][e]]
        cfs.append(cf); queues.append(q)
    return cfs, queues


def main_vm_streams(zkgl, cs, limit, n_exec=None, fixture=None, n_threads=0):
    """(outer [words, E], loop [words, E * limit] with the carried VM state zero, expected commitments [E, 4] or None): the fixture's
    VmCircuitWitnesses — closed-form input + the WitnessOracle's per-getter FIFOs — through the product's packer
    (zk_pack_main_vm_witness_batch: one chunk per host thread)"""
    fx = np.load(fixture or FIXTURE)
    E = int(fx["commitment"].shape[0]) if n_exec is None else min(int(n_exec), int(fx["commitment"].shape[0]))
    n_outer, n_loop = cs.input_words()
    outer = np.zeros((n_outer, E), dtype=np.uint64)
    loop = np.zeros((n_loop, E * limit), dtype=np.uint64)
    cfs, queues = fixture_witnesses(zkgl, fx, E)
    reps = cs.pack_main_vm_witness_batch(cfs, [q.view() for q in queues], 0, E, outer, loop, n_threads=n_threads)
    for e, rep in enumerate(reps):
        if rep.underflow:
            raise RuntimeError(f"fixture execution {e}: the packer ran out of oracle answers (fixture made for another circuit?)")
    expect = fx["commitment"][:E].copy() if int(fx["limit"][0]) == limit else None
    return outer, loop, expect


# ------------------------------------------------------------------------------------------------ the host side of a step
class HostFeed:
    """What the host does per step when every batch is NEW witness data: the B chunks of a window packed on the host pool
    (zk_pack_main_vm_witness_batch) into a staging array, the array copied to its place in the device stream.  Two forms:
      "device_seeds"        ZK_VM_PACK_ORACLE_WORDS_ONLY: 117 of the 360 rows per cycle; the device pass derives the VmLocalState rows
      "states_from_witness" ZK_VM_PACK_STATES_FROM_WITNESS: all 360 rows, the queue tails read from the witness generator's queue
                            states (src/fsm_input_output/circuit_inputs/main_vm.rs:64-71, src/ram_permutation/input.rs:105-116): no device pass
    The packing half needs no GPU (tests/test_vm_pack.py drives it with numpy staging)."""

    def __init__(self, zkgl, cs, limit, B, fixture, mode, n_threads, first_execution=0):
        self.zkgl, self.cs, self.limit, self.B, self.mode, self.n_threads = zkgl, cs, limit, B, mode, max(1, int(n_threads))
        self.fx = np.load(fixture)
        self.n_exec = int(self.fx["commitment"].shape[0])
        self.first_execution = first_execution
        self.n_outer, self.n_loop = cs.input_words()
        self.cfs, self.queues = fixture_witnesses(zkgl, self.fx, self.n_exec)
        self.views = [q.view() for q in self.queues]
        self.flags = zkgl.VM_PACK_ORACLE_WORDS_ONLY if mode == "device_seeds" else zkgl.VM_PACK_STATES_FROM_WITNESS
        self.first_row = VM_STATE_WORDS if mode == "device_seeds" else 0
        self.states = None
        if mode == "states_from_witness":       # the witness generator's queue states, produced once by the packer's RECORD mode (hashing, outside any timing)
            self.states, self._keep = [], []
            o = np.zeros((self.n_outer, 1), dtype=np.uint64); l = np.zeros((self.n_loop, limit), dtype=np.uint64)
            for e in range(self.n_exec):
                arrs = (np.zeros((8 * limit, 12), dtype=np.uint64), np.zeros((2 * limit, 12), dtype=np.uint64), np.zeros((2 * limit, 4), dtype=np.uint64))
                st = zkgl.VmQueueStates.over(*arrs)
                cs.pack_main_vm_witness_states(self.cfs[e], self.views[e], st, 0, 1, o, l, zkgl.VM_PACK_FILL_STATE | zkgl.VM_PACK_RECORD_STATES)
                cut = [np.ascontiguousarray(a[:max(int(k), 1)]) for a, k in zip(arrs, (st.used_memory_tails, st.used_decommit_tails, st.used_log_forward_tails))]
                self._keep.append(cut)
                self.states.append(zkgl.VmQueueStates.over(*cut))
        self._arrays = {}

    def window_arrays(self, k):
        """ctypes arrays of the B chunks of window k: chunk j replays execution (first_execution + k * B + j) % n_exec"""
        if k not in self._arrays:
            C, z = self.zkgl.C, self.zkgl
            idx = [(self.first_execution + k * self.B + j) % self.n_exec for j in range(self.B)]
            cfa = (z.VmClosedFormInput * self.B)(*[self.cfs[e] for e in idx])
            oa = (z.VmWitnessOracle * self.B)(*[self.views[e] for e in idx])
            sa = None if self.states is None else (z.VmQueueStates * self.B)(*[self.states[e] for e in idx])
            self._arrays[k] = (cfa, oa, sa)
        return self._arrays[k]

    def rows(self):
        return self.n_loop - self.first_row

    def pack_window(self, k, stage_outer, stage_loop, n_threads=None):
        """window k into stage_outer [n_outer, B] / stage_loop [rows(), B * limit] (u64, C-contiguous); seconds spent in the packer"""
        cfa, oa, sa = self.window_arrays(k)
        t = time.perf_counter()
        reps = self.cs.pack_main_vm_witness_batch(cfa, oa, 0, self.B, stage_outer, stage_loop, flags=self.flags, states=sa,
                                                  n_threads=self.n_threads if n_threads is None else n_threads)
        dt = time.perf_counter() - t
        if any(r.underflow for r in reps):
            raise RuntimeError("host feed: the packer ran out of oracle answers / queue states")
        return dt


def host_fed_steps(torch, zkgl, cs, feed, dev, K, steps, step_stream, expect, gather_fn):
    """`steps` steps in which every window is packed on the host and copied to the device WHILE the GPU resolves the previous one.
    Two device streams of K windows: the GPU works through one (seeding pass at its first window in the "device_seeds" form) while the
    feeder thread fills the other window by window.  Returns the measured figures; commitments of the last window are compared with
    the fixture's (so the words the GPU consumed are the words the host packed: the device streams start out zeroed)."""
    import threading
    B, limit, n_outer, n_loop = feed.B, feed.limit, feed.n_outer, feed.n_loop
    S = B * K
    stage_o = torch.zeros((n_outer, B), dtype=torch.int64).pin_memory()
    stage_l = torch.zeros((feed.rows(), B * limit), dtype=torch.int64).pin_memory()
    so_np, sl_np = stage_o.numpy().view(np.uint64), stage_l.numpy().view(np.uint64)
    d_o = [torch.zeros((n_outer, S), dtype=torch.int64, device=dev) for _ in range(2)]
    d_l = [torch.zeros((n_loop, S * limit), dtype=torch.int64, device=dev) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    pack_s, h2d_s, errors = [], [], []

    def fill(buf, k):
        try:
            pack_s.append(feed.pack_window(k, so_np, sl_np))
            t = time.perf_counter()
            with torch.cuda.device(dev), torch.cuda.stream(copy_stream):
                d_l[buf][feed.first_row:, k * B * limit:(k + 1) * B * limit].copy_(stage_l, non_blocking=True)
                d_o[buf][:, k * B:(k + 1) * B].copy_(stage_o, non_blocking=True)
            copy_stream.synchronize()
            h2d_s.append(time.perf_counter() - t)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def seed(buf):
        if feed.mode == "device_seeds":
            cs.seed_window_async(S, d_o[buf], S, d_l[buf], S * limit, 0, step_stream.cuda_stream)

    def run(buf, k):
        cs.bind_inputs(False, d_o[buf], n_outer, lane_stride=S, lane_offset=k * B)
        cs.bind_inputs(True, d_l[buf], n_loop, lane_stride=S * limit, lane_offset=k * B * limit)
        ok, failure = cs.resolve_and_check(step_stream.cuda_stream)
        if not ok:
            raise RuntimeError(f"host-fed trace not satisfied: {failure}")
        gather_fn()

    one_core_s = feed.pack_window(0, so_np, sl_np, n_threads=1) / B      # what ONE host core needs per chunk (staging already touched)
    for k in range(K):          # the first stream, untimed
        fill(0, k)
    if errors:
        raise RuntimeError(errors[0])
    n_prologue = len(pack_s)
    cur = 0
    seed(cur)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last_k = 0
    for i in range(steps):
        k = i % K
        if k == 0 and i > 0:
            cur ^= 1
            seed(cur)           # queued in front of the step kernels, like the resident-stream measurement
        th = threading.Thread(target=fill, args=(cur ^ 1, k))
        th.start()
        run(cur, k)
        th.join()
        last_k = k
        if errors:
            raise RuntimeError(errors[0])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    local = np.array([cs.public_inputs(i) for i in range(B)], dtype=np.uint64)
    parity = None if expect is None else bool(np.array_equal(local, expect[last_k * B:(last_k + 1) * B]))
    pk, hd = pack_s[n_prologue:], h2d_s[n_prologue:]
    staged_bytes = (stage_l.numel() + stage_o.numel()) * 8
    del d_o, d_l, stage_o, stage_l
    return {"elapsed": elapsed, "steps": steps, "pack_ms_per_window": 1e3 * float(np.mean(pk)), "h2d_ms_per_window": 1e3 * float(np.mean(hd)),
            "pack_ms_per_instance_one_core": 1e3 * one_core_s, "staged_bytes_per_window": staged_bytes, "h2d_GBps": staged_bytes / float(np.mean(hd)) / 1e9, "commitments_equal_native_restatement": parity}


# ------------------------------------------------------------------------------------------------ CPU baseline (the only leg that may touch oracle/)
def cpu_baseline(log2_rows, seconds_target=20.0):
    """CPU restatement ("port"): the oracle's IR interpreter + checker (oracle/zko_engine.c, gcc -O3 -march=native -flto, OpenMP over
    the lanes = instances x cycles on every host core) on a bounded sample of the SAME workload: full-size main_vm instances from
    the same fixture.  Median of the passes that fit the time budget (>= 3, <= 5).  NOT the reference Rust binary (unbuildable here)."""
    import zkgl
    from oracle import zko

    cores = os.cpu_count() or 1
    cs, limit = build_main_vm_cs(zkgl, log2_rows)
    n_inst = 8 if cores >= 16 else 2
    outer, loop, _ = main_vm_streams(zkgl, cs, limit, n_inst)
    total_rows = int(sum(t["n_rows"] for t in zko.parse_export(cs.export(False))["tables"]))
    run = zko.CircuitRun(cs.export(False), cs.export(True), n_inst, total_rows)
    t0 = time.perf_counter()
    loop = run.seed(outer, loop)  # sequential seeding of the carried state, reported separately like the GPU leg
    t_seed = time.perf_counter() - t0
    times = []
    t_all = time.perf_counter()
    for rep in range(5):
        t0 = time.perf_counter()
        run.resolve(outer, loop)
        t1 = time.perf_counter()
        bad, nrel = run.check()
        t2 = time.perf_counter()
        assert bad == 0
        times.append((t2 - t0, t1 - t0, t2 - t1))
        if rep >= 2 and time.perf_counter() - t_all > seconds_target:
            break
    times.sort()
    med = times[len(times) // 2]
    st = cs.stats()
    cs.close()
    return {"value": nrel / med[0], "unit": "constraints/s", "cores": cores, "kind": "port",
            "witness_rows_per_s": n_inst * st["rows_per_instance"] / med[1],
            "value_from_raw_witness": nrel / (med[0] + t_seed),
            "sample": f"{n_inst} full-size main_vm instances ({limit} cycles, {st['rows_per_instance']} rows, {nrel} constraints in total), "
                      f"median of {len(times)} passes: resolve {med[1]:.2f}s + check {med[2]:.2f}s (+ seeding {t_seed:.2f}s: one thread per instance walks its cycles in order, outside `value`), "
                      f"OpenMP {cores} threads over instances x cycles, Goldilocks reduction by the 2^64 = 2^32 - 1 identity; "
                      f"CPU restatement (oracle), not the reference Rust binary"}


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=384, help="independent circuit instances per GPU per step")
    ap.add_argument("--seed-windows", type=int, default=0, help="batches per stream of raw witness (0: = --steps, clamped to 2..8); two streams are resident per GPU")
    ap.add_argument("--log2-rows", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="only the timed default-mode steps (profiler runs: every k_witness_loop launch of the process is then a default-mode launch); the secondary figures are null")
    ap.add_argument("--no-host-feed", action="store_true", help="skip the host-fed figures (packing + H2D of every window overlapped with the steps)")
    ap.add_argument("--fixture", default="default", choices=sorted(FIXTURES), help="which synthetic executions main_vm replays (default: every opcode family every ~150 cycles)")
    ap.add_argument("--with-narrow-store-mode", action="store_true", help="also measure the labelled mode `mode_narrow_store` (the same steps over the narrow store, then with the Poseidon2 "
                    "intermediates deferred on top).  Off by default: its kernels have not run on a device yet, and a secondary figure must never be able to cost the line")
    ap.add_argument("--narrow-store", action="store_true", help="run the HEADLINE steps over the narrow store (ZKGL_NARROW_STORE=1 at zk_cs_set_batch: byte-class values in one-byte slots, "
                    "csrc/store_geom.hpp); without it the narrow store is a labelled mode beside `value` (mode_narrow_store)")
    args = ap.parse_args()
    # ---- N > 1 without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                   "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:])
    # stdout carries the ONE JSON line and nothing else: libraries that write to fd 1 on their own (librccl prints a version banner
    # at its first communicator, possibly from another thread) are sent to stderr; the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import zkgl

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise RuntimeError("bench.py needs a GPU: libzkgl has no CPU fallback")
    dev_index = local_rank % n_dev
    shared_gpu = world > n_dev            # only in smoke tests of the N>1 path on a 1-GPU box: RCCL refuses duplicate devices
    if world > 1:
        # control plane (barriers, the communicator id, max over ranks) over gloo; the ONE RCCL communicator of the run is the
        # product's (zk_comm_create), used by the path's only collective: the commitment gather
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(dev_index)
    zkgl.init(dev_index)
    dev = torch.device("cuda", dev_index)

    B = args.batch
    # raw witness: a stream of K batches is resident; ONE seeding pass derives the VM state of all its K * B instances (a pass is a
    # latency chain: it costs about the same for 384 and for 1920 instances, so long passes are the cheap way to do it), then K steps
    # resolve and check its K windows.  K = --steps: any K consecutive steps contain exactly one pass.  (Running the pass of the next
    # stream underneath the steps on a second HIP stream was measured and lost: its 2 000 long-lived wavefronts take registers from
    # the bandwidth-bound step kernels for longer than the pass takes alone — profiles/r3_summary.md.)
    K = args.seed_windows if args.seed_windows >= 2 else min(max(args.steps, 2), 8)
    S = B * K
    expect = None
    step_stream = torch.cuda.current_stream()
    seed_stream = step_stream
    stream = step_stream.cuda_stream
    cs, limit = build_main_vm_cs(zkgl, args.log2_rows)
    n_outer, n_loop = cs.input_words()
    t_pack = time.perf_counter()
    outer_e, loop_e, expect_e = main_vm_streams(zkgl, cs, limit, fixture=FIXTURES[args.fixture])   # zk_pack_main_vm_witness: FIFOs -> streams, all executions
    t_pack = time.perf_counter() - t_pack
    n_exec = outer_e.shape[1]
    # the stream is assembled on the device: instance i replays execution (rank * S + i) % n_exec of the fixture
    sel = (torch.arange(S, device=dev) + rank * S) % n_exec
    d_outer = torch.from_numpy(outer_e.view(np.int64)).to(dev)[:, sel].contiguous()
    le = torch.from_numpy(loop_e.view(np.int64)).to(dev).view(n_loop, n_exec, limit)
    d_loop = le[:, sel, :].reshape(n_loop, S * limit).contiguous()
    del le, loop_e
    expect = None if expect_e is None else expect_e[((np.arange(S) + rank * S) % n_exec)]
    bufs = [d_loop]
    st = cs.stats()
    if args.narrow_store:
        os.environ["ZKGL_NARROW_STORE"] = "1"
    cs.set_batch(B)
    headline_narrow = bool(cs.stats()["narrow_store_active"])
    if args.narrow_store and not headline_narrow:
        raise RuntimeError("--narrow-store: zk_cs_set_batch did not take the narrow store (no layout, strand-form loop launch, or no room for both stores)")

    # ---- the path's only collective: ONE all-gather of the 4-element input commitments of the batch PER STEP, behind the
    # C ABI (zk_comm_* + zk_cs_gather_commitments = k_pack_public + one ncclAllGather over RCCL / xGMI on the step's stream), INSIDE the
    # timed region for every N — at N = 1 it is a one-rank all-gather.  The launcher's part (rank 0's unique id to the other ranks)
    # goes over the gloo control plane.  Ranks that share a device (world > device_count: only the N>1 smoke test on a 1-GPU box) cannot
    # form an RCCL communicator (duplicate devices are refused) and gather over gloo — the line says so.  Everywhere else a failing RCCL
    # path is an ERROR: a scaling curve measured over a silent gloo fallback would not be the north star's.
    comm = None
    if shared_gpu:
        gather_path = "torch.distributed (gloo) all_gather per step — ranks share one GPU (smoke test), RCCL refuses duplicate devices"
    else:
        gather_path = "zk_cs_gather_commitments per step (k_pack_public + one RCCL all-gather behind the C ABI, on the step stream)"
        ids = [zkgl.Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        try:
            comm = zkgl.Comm(bytes(ids[0]), rank, world)
        except Exception as e:  # noqa: BLE001
            if not os.environ.get("ZKGL_BENCH_ALLOW_GLOO_GATHER"):
                raise RuntimeError(f"rank {rank}: the RCCL communicator could not be created on {n_dev} visible device(s): {e} "
                                   "(ZKGL_BENCH_ALLOW_GLOO_GATHER=1 falls back to gloo and labels the line)") from e
            gather_path = f"torch.distributed (gloo) all_gather per step — RCCL communicator FAILED on rank {rank}: {e}"
        if world > 1:   # every rank learns whether any rank fell back: one mode for the whole job
            from zkgl.dist import gather_floats as gather_floats_early
            oks = gather_floats_early(0.0 if comm is None else 1.0)
            if min(oks) < 1.0 and comm is not None:
                comm.close(); comm = None
                gather_path = "torch.distributed (gloo) all_gather per step — RCCL communicator FAILED on another rank"
    assert comm is not None or shared_gpu or os.environ.get("ZKGL_BENCH_ALLOW_GLOO_GATHER"), "RCCL gather expected"
    gather_events = []

    cur = [0]   # the stream being resolved; the other one is being seeded

    def seed(buf, first, n, sync=True):
        with torch.cuda.stream(seed_stream):
            cs.seed_window_async(n, d_outer, S, bufs[buf], S * limit, first, seed_stream.cuda_stream)
        if sync:
            seed_stream.synchronize()

    def bind(k):
        cs.bind_inputs(False, d_outer, n_outer, lane_stride=S, lane_offset=k * B)
        cs.bind_inputs(True, bufs[cur[0]], n_loop, lane_stride=S * limit, lane_offset=k * B * limit)

    # ---- seeding alone (not overlapped with anything): one window, and a whole stream in one pass
    torch.cuda.synchronize()
    seed(0, 0, S)                                   # first call allocates the scratch
    t = time.perf_counter(); seed(0, 0, B); t_seed_window = time.perf_counter() - t
    t = time.perf_counter(); seed(0, 0, S); t_seed_stream = time.perf_counter() - t
    window = [0]
    step_no = [0]

    def resolve(k):
        bind(k)
        window[0] = k
        ok, failure = cs.resolve_and_check(stream)  # witness generation + full satisfiability check, one pipeline
        if not ok and not os.environ.get("ZKGL_STUB_RUN"):  # ZKGL_STUB_RUN: tools/stub_bench.sh times deliberately wrong kernel variants
            raise RuntimeError(f"trace not satisfied: {failure}")

    def gather(timed=False):
        """the step's collective: all ranks' commitments of the window just resolved"""
        if comm is not None:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(step_stream)
            cs.gather_commitments_async(comm, stream)
            if timed:
                e1.record(step_stream)
                gather_events.append((e0, e1))
            return None
        from zkgl.dist import gather_commitments
        t = time.perf_counter()
        c = gather_commitments(np.array([cs.public_inputs(i) for i in range(B)], dtype=np.uint64))
        if timed:
            gather_events.append(time.perf_counter() - t)
        return c

    def step(timed=False):
        """one batch from the raw witness: at window 0 the stream's seeding pass (all K batches), then window k is resolved and checked,
        then the batch's commitments are all-gathered"""
        k = step_no[0] % K
        if k == 0:
            seed(0, 0, S, sync=False)       # same stream order as the step kernels: queued, not waited for
        step_no[0] += 1
        resolve(k)
        return gather(timed)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    loop_ms, check_ms, gate_ms, outer_ms, shader_mhz, p2_skipped = [], [], [], [], [], []
    commits = None
    for _ in range(args.steps):
        commits = step(timed=True)
        loop_ms.append(cs.last_ms(1)); check_ms.append(cs.last_ms(2)); gate_ms.append(cs.last_ms(3)); outer_ms.append(cs.last_ms(4))
        shader_mhz.append(cs.last_ms(8)); p2_skipped.append(cs.last_ms(9))
    fence()
    elapsed_local = time.perf_counter() - t0
    local = np.array([cs.public_inputs(i) for i in range(B)], dtype=np.uint64)
    last_window = window[0]
    if comm is not None:
        commits = comm.gathered().copy()          # [world, B, 4]: the last step's gather
        gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_events]))
    else:
        gather_ms = 1e3 * float(np.mean(gather_events))
    if not np.array_equal(commits[rank], local) and not os.environ.get("ZKGL_STUB_RUN"):  # stub variants store garbage
        raise RuntimeError("gathered commitments differ from this rank's public inputs")
    # (--headline-only: profiler runs skip everything below, so that every k_witness_loop launch of the process is a default-mode launch)
    raw_stored_local = resident_local = stored_local = float("nan")
    deferred, mat_s, mat_chunk, n_cols_trace = None, None, 0, int(st["copy_columns"] + st["lookup_columns"])
    if not args.headline_only:
        # ---- the same K steps from the raw witness (seeding pass, gather) with EVERY relation re-evaluated from the stored values
        # (zk_cs_set_check_mode(ZK_CHECK_STORED): what check_if_satisfied does, /root/reference/src/ram_permutation/mod.rs:556)
        cs.set_check_mode(True)
        step_no[0] = 0
        for _ in range(min(args.warmup, 1)):
            step()
        step_no[0] = 0
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        raw_stored_local = time.perf_counter() - t1
        cs.set_check_mode(False)
        resolve(last_window)
        # ---- LABELLED MODE, not `value`: ZK_CHECK_FUSED_DEFER_P2 — the loop kernel leaves out the 950 intermediates of every in-circuit Poseidon2
        # permutation (nothing in the fused step reads them); whoever reads the store later (the full check, the column readers) gets them
        # regenerated bit for bit by k_fill_p2, timed here on its own (zk_cs_complete_store).  Its unit differs: fewer values per cycle.
        deferred = None
        cs.set_check_mode(False, defer_p2=True)
        step_no[0] = 0
        step(); fence()
        step_no[0] = 0
        d_loop_ms = []
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
            d_loop_ms.append(cs.last_ms(1))
        fence()
        d_elapsed = time.perf_counter() - t1
        d_local = np.array([cs.public_inputs(i) for i in range(min(B, 8))], dtype=np.uint64)   # reads go through the fill
        resolve(window[0]); torch.cuda.synchronize()
        tf = time.perf_counter(); cs.complete_store(stream); torch.cuda.synchronize(); fill_s = time.perf_counter() - tf
        deferred = {"elapsed": d_elapsed, "loop_ms": float(np.mean(d_loop_ms)), "fill_s": fill_s,
                    "commitments_equal": bool(expect is None or np.array_equal(d_local, expect[window[0] * B: window[0] * B + d_local.shape[0]]))}
        cs.set_check_mode(False)
        resolve(last_window)
        # ---- the same K steps with the per-cycle state already resident (what rounds 1-2 reported as `value`)
        t1 = time.perf_counter()
        for i in range(args.steps):
            resolve((last_window + 1 + i) % K)
        torch.cuda.synchronize()
        resident_local = time.perf_counter() - t1
        # ---- and with every gate re-evaluated from the stored values (the mode of rounds 1-2; resolve_and_check's default is fused: the
        # gates mirrored by their producing witness op are evaluated by the witness kernels)
        cs.set_check_mode(True)
        resolve(0); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            resolve((last_window + 1 + i) % K)
        torch.cuda.synchronize()
        stored_local = time.perf_counter() - t1
        cs.set_check_mode(False)
        resolve(last_window)   # back to the window the commitments were read from, default mode
        # ---- materialised witness columns: the variable store is what the step writes; the trace proper (every cell of every column of
        # every instance, as column polynomials) is produced on demand by zk_cs_trace_columns_batch — timed here for the WHOLE batch, in
        # chunks of as many instances as fit beside the store (a batch's columns are 4x its store)
        mat_s, mat_chunk, n_cols_trace = None, 0, int(st["copy_columns"] + st["lookup_columns"])
        try:
            free_b, _total_b = torch.cuda.mem_get_info(dev)
            per_inst = n_cols_trace << (args.log2_rows + 3)
            mat_chunk = int(max(1, min(B, 64, (free_b - (8 << 30)) // per_inst)))
            cols = torch.empty(mat_chunk * (n_cols_trace << args.log2_rows), dtype=torch.int64, device=dev)
            cs.trace_columns_batch(0, mat_chunk, cols, args.log2_rows, n_cols_trace, stream=stream); torch.cuda.synchronize()
            tm = time.perf_counter()
            for i0 in range(0, B, mat_chunk):
                cs.trace_columns_batch(i0, min(mat_chunk, B - i0), cols, args.log2_rows, n_cols_trace, stream=stream)
            torch.cuda.synchronize()
            mat_s = (time.perf_counter() - tm) / B
            del cols
        except Exception as e:  # noqa: BLE001
            print(f"[bench] trace_columns timing unavailable: {e}", file=sys.stderr)

    # ---- the SAME steps on the compiled-contract-like opcode mix (tests/golden/vm_bench_witness_realistic.npz: ~1 % logs, ~0.5 % calls, ~10 % heap
    # accesses, large operands): a second labelled value beside the headline's synthetic mix.  Same circuit, same batch, from the raw witness
    # (seeding pass + K windows + gather).  The interpreter's data-dependent costs differ: more gated permutations skipped, but zero-checks
    # of large operands and U256 divisions in every wavefront (profiles/r4_loop_variants.md).
    realistic = None
    if not args.headline_only and args.fixture == "default" and not os.environ.get("ZKGL_STUB_RUN"):
        keep = (d_outer, bufs[0], expect)
        try:
            outer_r, loop_r, expect_r = main_vm_streams(zkgl, cs, limit, fixture=FIXTURES["realistic"])
            n_exec_r = outer_r.shape[1]
            sel_r = (torch.arange(S, device=dev) + rank * S) % n_exec_r
            d_outer = torch.from_numpy(outer_r.view(np.int64)).to(dev)[:, sel_r].contiguous()
            le = torch.from_numpy(loop_r.view(np.int64)).to(dev).view(n_loop, n_exec_r, limit)
            bufs[0] = le[:, sel_r, :].reshape(n_loop, S * limit).contiguous()
            del le, loop_r
            expect = None
            cs.set_check_mode(False)
            step_no[0] = 0
            step(); fence()
            step_no[0] = 0
            r_loop_ms, r_skip = [], []
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
                r_loop_ms.append(cs.last_ms(1)); r_skip.append(cs.last_ms(9))
            fence()
            r_elapsed = time.perf_counter() - t1
            r_local = np.array([cs.public_inputs(i) for i in range(min(B, 8))], dtype=np.uint64)
            w = window[0]
            want = None if expect_r is None else expect_r[((np.arange(S) + rank * S) % n_exec_r)][w * B: w * B + r_local.shape[0]]
            realistic = {"elapsed": r_elapsed, "loop_ms": float(np.mean(r_loop_ms)), "skipped": float(np.mean(r_skip)),
                         "commitments_equal": None if want is None else bool(np.array_equal(r_local, want)), "n_exec": int(n_exec_r)}
        except Exception as e:  # noqa: BLE001
            realistic = {"error": repr(e)}
            print(f"[bench] realistic-fixture figure unavailable: {e}", file=sys.stderr)
        d_outer, bufs[0], expect = keep
        del keep
        torch.cuda.empty_cache()
        try:
            cs.set_check_mode(False)
            step_no[0] = 0
            resolve(last_window)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] re-resolve after the realistic-fixture run failed: {e}", file=sys.stderr)
    # ---- LABELLED MODE, not `value`: the NARROW STORE (csrc/store_geom.hpp).  The values the circuit's own constraints bound below 2^8 in every
    # satisfying witness (zk_stats.narrow_byte_values_loop of a cycle's values) live in one-byte slots of the store the fused step writes and
    # reads; every other reader gets the ordinary store through k_widen_store (timed here on its own).  Same steps from the raw witness
    # (seeding pass, fused check, gather), then the same with the Poseidon2 intermediates deferred on top (the two byte levers together).
    narrow = None
    if args.with_narrow_store_mode and not args.headline_only and not headline_narrow and not os.environ.get("ZKGL_STUB_RUN") and st["narrow_store_bytes_per_lane_loop"]:
        try:
            os.environ["ZKGL_NARROW_STORE"] = "1"
            cs.set_batch(B)
            if not cs.stats()["narrow_store_active"]:
                raise RuntimeError("zk_cs_set_batch did not take the narrow store at this batch")
            narrow = {}
            for label, defer in (("plain", False), ("p2_deferred", True)):
                cs.set_check_mode(False, defer_p2=defer)
                step_no[0] = 0
                step(); fence()
                step_no[0] = 0
                n_loop_ms, n_gate_ms = [], []
                r0 = cs.stats()["narrow_repeats"]
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                    n_loop_ms.append(cs.last_ms(1)); n_gate_ms.append(cs.last_ms(3))
                fence()
                n_elapsed = time.perf_counter() - t1
                resolve(window[0]); torch.cuda.synchronize()
                tw = time.perf_counter(); cs.complete_store(stream); torch.cuda.synchronize(); widen_s = time.perf_counter() - tw
                n_local = np.array([cs.public_inputs(i) for i in range(min(B, 8))], dtype=np.uint64)
                narrow[label] = {"elapsed": n_elapsed, "loop_ms": float(np.mean(n_loop_ms)), "gate_ms": float(np.mean(n_gate_ms)),
                                 "complete_store_s": widen_s, "repeats": int(cs.stats()["narrow_repeats"] - r0),
                                 "commitments_equal": bool(expect is None or np.array_equal(n_local, expect[window[0] * B: window[0] * B + n_local.shape[0]]))}
        except Exception as e:  # noqa: BLE001
            narrow = {"error": repr(e)}
            print(f"[bench] narrow-store figure unavailable: {e}", file=sys.stderr)
        os.environ.pop("ZKGL_NARROW_STORE", None)
        try:
            cs.set_check_mode(False)
            cs.set_batch(B)          # back to the ordinary store for what follows
            step_no[0] = 0
            resolve(last_window)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] re-resolve after the narrow-store run failed: {e}", file=sys.stderr)
    # ---- the host side of the step, measured: every window packed from the WitnessOracle FIFOs on this rank's share of the host threads and
    # copied to the device while the GPU resolves the previous window (HostFeed / host_fed_steps above).  Labelled figures beside `value`
    # (whose inputs are resident, as the contract asks).  No collective inside: a failure on one rank must not hang the others.
    host_feed = None
    if not args.headline_only and not args.no_host_feed and not os.environ.get("ZKGL_STUB_RUN"):
        host_feed = {}
        n_threads = max(1, zkgl.host_threads() // max(1, world))
        for mode in ("device_seeds", "states_from_witness"):
            try:
                feed = HostFeed(zkgl, cs, limit, B, FIXTURES[args.fixture], mode, n_threads, first_execution=rank * S)
                cs.set_check_mode(False)
                r = host_fed_steps(torch, zkgl, cs, feed, dev, K, args.steps, step_stream, expect, lambda: gather())
                r["host_threads_used"] = n_threads
                host_feed[mode] = r
                del feed
            except Exception as e:  # noqa: BLE001
                host_feed[mode] = {"error": repr(e)}
                print(f"[bench] host feed ({mode}) unavailable: {e}", file=sys.stderr)
            torch.cuda.empty_cache()
    from zkgl.dist import gather_floats, max_over_ranks
    elapsed = max_over_ranks(elapsed_local)
    resident = max_over_ranks(resident_local)
    stored = max_over_ranks(stored_local)
    raw_stored = max_over_ranks(raw_stored_local)
    gather_ms = max_over_ranks(gather_ms)
    if deferred is not None:
        deferred["elapsed"] = max_over_ranks(deferred["elapsed"])
    per_rank_ms = gather_floats(1e3 * elapsed_local / args.steps)
    parity = None
    if expect is not None:
        parity = bool(np.array_equal(local, expect[last_window * B:(last_window + 1) * B]))
        if not parity and not os.environ.get("ZKGL_STUB_RUN"):
            raise RuntimeError("public inputs differ from the native restatement's commitments stored in the fixture")
    if comm is not None:
        comm.close()
    if rank == 0:
        n_inst = B * world
        per_step_constraints = st["constraints_per_instance"] * n_inst
        constraints = per_step_constraints * args.steps
        rows = st["rows_per_instance"] * n_inst * args.steps
        # dominant kernel: the loop-scope witness interpreter.  ALGORITHMIC bytes per launch = every witness VALUE of the loop
        # rows written once (8 B; one per variable — the trace is a view of the variable store) + every input word
        # read once.  SURVEY §8(d) counts every trace CELL (a variable occupies 3.1 cells on average): that figure is reported
        # beside it as trace_cell_equivalent_GBps, it is not what the kernel has to move.
        lane_bytes = (st["narrow_store_bytes_per_lane_loop"] if headline_narrow else st["cells_written_loop"] * 8) + n_loop * 8
        algo_bytes = B * st["limit"] * lane_bytes
        cell_bytes = B * st["limit"] * (st["cells_populated_loop"] + n_loop) * 8
        k_ms = float(np.mean(loop_ms))
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        valu_busy, valu_src = None, None
        for name in ("pmc_r6n.json" if headline_narrow else "pmc_r6.json", "pmc_r5.json", "pmc_r4.json", "pmc_r3.json", "pmc_r2.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                if headline_narrow != pmc.get("kernel", "").endswith("_narrow"):
                    continue     # a ratio measured on the other store's kernel says nothing about this one
                traffic = pmc["traffic_over_algorithmic"] * algo_bytes
                traffic_src = f"profiles/{name} ratio {pmc['traffic_over_algorithmic']:.3f} measured at batch {pmc['batch']}"
                vi = pmc.get("valu_issue") or {}
                if vi.get("valu_busy_frac") is not None:
                    valu_busy, valu_src = float(vi["valu_busy_frac"]), f"profiles/{name} (SQ_ACTIVE_INST_VALU x 4 / 32 / GRBM_GUI_ACTIVE, its own --pmc pass)"
                break
            except Exception:
                pass
        step_s = elapsed / args.steps
        res_s = resident / args.steps
        p2_per_cycle = 9   # in-circuit Poseidon2 permutations of a main_vm cycle (the code-word read + the 8 enforced sponges, cycle.rs:670-784), 962 values each
        flat = commits.reshape(-1).astype(np.uint64)
        def feed_line(r, what):
            try:
                return _feed_line(r, what)
            except Exception as e:  # noqa: BLE001  (a secondary figure must never cost the line)
                return {"error": repr(e)}

        def _feed_line(r, what):
            if r is None or "error" in r:
                return r
            ms = 1e3 * r["elapsed"] / r["steps"]
            cores = r["pack_ms_per_instance_one_core"] * B / (1e3 * step_s)     # host cores that pack one window per step of `value`
            return {"what": what, "value": st["constraints_per_instance"] * B * r["steps"] / r["elapsed"], "unit": "constraints/s (this rank's GPU)",
                    "ms_per_step": ms, "host_threads_used": r["host_threads_used"], "host_threads_of_the_box": zkgl.host_threads(),
                    "pack_ms_per_window": r["pack_ms_per_window"], "pack_ms_per_instance_one_core": r["pack_ms_per_instance_one_core"],
                    "h2d_ms_per_window": r["h2d_ms_per_window"], "h2d_GBps": r["h2d_GBps"], "staged_bytes_per_window": r["staged_bytes_per_window"],
                    "host_cores_per_gpu_to_sustain_value": cores, "gpus_one_host_can_feed": zkgl.host_threads() / cores if cores > 0 else None,
                    "commitments_equal_native_restatement": r["commitments_equal_native_restatement"]}
        try:
            realistic_line = None if realistic is None else realistic if "error" in realistic else {
                "what": "the same steps (from the raw witness: seeding pass per K windows, fused check, gather) replaying tests/golden/vm_bench_witness_realistic.npz — "
                        "the compiled-contract-like opcode mix (~1 % logs, ~0.5 % calls, ~10 % heap accesses, large operands)",
                "value": st["constraints_per_instance"] * B * args.steps / realistic["elapsed"], "unit": "constraints/s (this rank's GPU)",
                "ms_per_step": 1e3 * realistic["elapsed"] / args.steps, "k_witness_loop_ms": realistic["loop_ms"],
                "achieved_GBps": algo_bytes / (realistic["loop_ms"] * 1e-3) / 1e9, "frac": algo_bytes / (realistic["loop_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "witness_only_permutations_skipped_frac": realistic["skipped"], "distinct_executions": realistic["n_exec"],
                "commitments_equal_native_restatement": realistic["commitments_equal"]}
        except Exception as e:  # noqa: BLE001
            realistic_line = {"error": repr(e)}
        out = {
            "metric": "constraints/s + witness-rows/s, main_vm 2^20 rows", "value": constraints / elapsed, "unit": "constraints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * step_s,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks)", "data": "synthetic",
            "value_is": f"from the raw witness: every K = {K} steps start with ONE seeding pass that derives the per-cycle VM state of the stream's {S} instances, then each step resolves + checks one window and all-gathers its commitments",
            "witness_rows_per_s": rows / elapsed,
            "vm_cycles_per_s": n_inst * limit * args.steps / elapsed,
            "value_inputs_resident": per_step_constraints / res_s,
            "value_inputs_resident_verify_stored": per_step_constraints / (stored / args.steps),
            "value_from_raw_witness_verify_stored": per_step_constraints / (raw_stored / args.steps),
            # where the relations counted in constraints_per_instance are evaluated in the default (fused) mode (zk_stats; the two add up):
            "constraints_evaluated_from_store_per_instance": st["constraints_from_store_fused"],
            "constraints_evaluated_in_witness_kernels_per_instance": st["constraints_in_witness_fused"],
            "gather_ms_per_step": gather_ms,
            "check_mode": "`value` is the FUSED mode: a gate mirrored by the witness op that produces its output (same variables and constants, proved "
                          "per gate at finalize) and every lookup tuple is evaluated by the witness kernel on the values it holds "
                          "(constraints_evaluated_in_witness_kernels_per_instance); the check kernels read the rest from the store "
                          "(constraints_evaluated_from_store_per_instance: enforcements, range checks of inputs, integer relations).  "
                          "value_from_raw_witness_verify_stored / value_inputs_resident_verify_stored: the same steps with EVERY relation "
                          "re-evaluated from the stored values (zk_cs_set_check_mode(ZK_CHECK_STORED)); verdicts are identical "
                          "(tests/test_fused_differential.py)",
            "value_from_raw_witness_serial": per_step_constraints / (res_s + t_seed_stream / K),
            "witness_rows_materialised_per_s": None if mat_s is None else st["rows_per_instance"] * n_inst / (step_s + B * mat_s),
            # every window packed on the host pool and copied in WHILE the previous one is resolved (rank 0's figures; every rank runs it on
            # host_threads / world threads, so at N = 8 this is what one box can do for its eight GPUs)
            "value_realistic_fixture": realistic_line,
            "value_including_host_pack": None if not host_feed else feed_line(host_feed.get("device_seeds"),
                "every step's window is NEW: B chunks packed from the WitnessOracle FIFOs (zk_pack_main_vm_witness_batch, ZK_VM_PACK_ORACLE_WORDS_ONLY: "
                "117 of 360 rows) on the host pool + one H2D copy, overlapped with the GPU step; the device pass derives the VmLocalState rows"),
            "value_states_from_witness": None if not host_feed else feed_line(host_feed.get("states_from_witness"),
                "as above with ZK_VM_PACK_STATES_FROM_WITNESS: the host writes all 360 rows, queue tails read from the witness's queue states "
                "(circuit_inputs/main_vm.rs:64-71); NO device seeding pass"),
            "config": {"workload": ("main_vm (real vm_cycle, 11 opcode families; synthetic zkEVM programs from tests/golden/" + os.path.basename(FIXTURES[args.fixture]) + ", "
                                    f"{n_exec} distinct executions through zk_pack_main_vm_witness)") +
                                   f", geometry 140/0/8/deg8 + 3x8 lookups, 2^{args.log2_rows} rows/instance",
                       "instances_per_gpu": B, "cycles_per_instance": limit, "rows_per_instance": st["rows_per_instance"],
                       "constraints_per_instance": st["constraints_per_instance"], "parallelism": f"independent instances x{world}",
                       "windows_per_stream": K, "steps_is_a_multiple_of_windows": args.steps % K == 0,
                       "seed_one_window_alone_s": round(t_seed_window, 4), "seed_whole_stream_alone_s": round(t_seed_stream, 4),
                       "seeding_instances_per_s": S / t_seed_stream, "ms_per_step_inputs_resident": 1e3 * res_s,
                       "host_pack_s": round(t_pack, 3), "trace_columns_s_per_instance": mat_s,
                       "trace_columns_measured_over": f"the whole batch ({B} instances) in chunks of {mat_chunk} (zk_cs_trace_columns_batch)",
                       "trace_columns_GBps": None if mat_s is None else (n_cols_trace << (args.log2_rows + 3)) / mat_s / 1e9,
                       "per_rank_ms_per_step": [round(x, 3) for x in per_rank_ms],
                       "commitments_equal_native_restatement": parity, "commitment_gather": gather_path},
            "mode_p2_intermediates_deferred": None if deferred is None else {
                "what": "zk_cs_set_check_mode(ZK_CHECK_FUSED_DEFER_P2): same steps from the raw witness (seeding pass, fused check, gather); k_witness_loop writes the 12 "
                        "outputs of an in-circuit Poseidon2 permutation and skips its 950 intermediates; k_fill_p2 regenerates them for the first reader",
                "value": constraints / deferred["elapsed"], "ms_per_step": 1e3 * deferred["elapsed"] / args.steps,
                "k_witness_loop_ms": deferred["loop_ms"],
                "values_written_per_cycle": st["cells_written_loop"] - 950 * p2_per_cycle,
                "algorithmic_bytes_per_launch": B * st["limit"] * (st["cells_written_loop"] - 950 * p2_per_cycle + n_loop) * 8,
                "achieved_GBps": B * st["limit"] * (st["cells_written_loop"] - 950 * p2_per_cycle + n_loop) * 8 / (deferred["loop_ms"] * 1e-3) / 1e9,
                "k_fill_p2_ms_per_batch": 1e3 * deferred["fill_s"],
                "ms_per_step_with_fill": 1e3 * (deferred["elapsed"] / args.steps + deferred["fill_s"]),
                "commitments_equal_native_restatement": deferred["commitments_equal"]},
            "mode_narrow_store": None if narrow is None else narrow if "error" in narrow else {
                "what": "ZKGL_NARROW_STORE=1 at zk_cs_set_batch (csrc/store_geom.hpp): the values the constraints bound below 2^8 in every satisfying witness live in one-byte "
                        "slots of the store the fused step writes and reads (k_witness_loop_narrow, k_check_prog_t<true>, links); same steps from the raw witness (seeding pass, "
                        "fused check, gather).  Every reader outside the step gets the ordinary store through k_widen_store (complete_store_ms_per_batch).  p2_deferred: "
                        "ZK_CHECK_FUSED_DEFER_P2 on top (its complete_store = widening + k_fill_p2)",
                "bytes_written_per_cycle": st["narrow_store_bytes_per_lane_loop"], "bytes_written_per_cycle_ordinary_store": st["store_bytes_per_lane_loop"],
                "bytes_ratio": st["narrow_store_bytes_per_lane_loop"] / st["store_bytes_per_lane_loop"], "byte_values_per_cycle": st["narrow_byte_values_loop"],
                **{label: {"value": st["constraints_per_instance"] * B * args.steps / r["elapsed"], "unit": "constraints/s (this rank's GPU)",
                           "ms_per_step": 1e3 * r["elapsed"] / args.steps, "k_witness_loop_narrow_ms": r["loop_ms"], "k_check_prog_t_true_ms": r["gate_ms"],
                           "algorithmic_bytes_per_launch": B * st["limit"] * (st["narrow_store_bytes_per_lane_loop"] - (950 * p2_per_cycle * 8 if label == "p2_deferred" else 0) + n_loop * 8),
                           "achieved_GBps": B * st["limit"] * (st["narrow_store_bytes_per_lane_loop"] - (950 * p2_per_cycle * 8 if label == "p2_deferred" else 0) + n_loop * 8) / (r["loop_ms"] * 1e-3) / 1e9,
                           "values_per_s_vs_ordinary_store_kernel": k_ms / r["loop_ms"],
                           "complete_store_ms_per_batch": 1e3 * r["complete_store_s"], "steps_repeated_over_the_ordinary_store": r["repeats"],
                           "commitments_equal_native_restatement": r["commitments_equal"]} for label, r in narrow.items()}},
            "roofline": {"bound": "hbm", "kernel": "zke::k_witness_loop_narrow" if headline_narrow else "zke::k_witness_loop", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the OTHER roofline that binds this kernel (co-limited): fraction of the cycles in which the vector ALUs of a
                         # SIMD were issuing, from the PMC pass of the committed profile (peak = 1: every SIMD issues every cycle it can)
                         "frac_valu_issue": valu_busy, "frac_valu_issue_source": valu_src,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": algo_bytes,
                         "unit_of_work": ("values (one per variable; narrow store: 1 B for the values the constraints bound below 2^8, 8 B otherwise)" if headline_narrow else
                                          "values (one per variable, 8 B): the variable store; trace cells are a view of it"),
                         "bytes_written_per_cycle": lane_bytes - n_loop * 8,
                         "avg_launch_ms": k_ms,
                         # clock probe inside the kernel (s_memtime / s_memrealtime of its first wavefront): the clock the power management
                         # granted this launch (2.09-2.27 GHz seen; the kernel's time has not followed it, profiles/r3_loop_probe.md §4)
                         "shader_clock_mhz": float(np.mean(shader_mhz)),
                         # simulate_round_function(cs, state, execute): 18 of the cycle's 27 permutations are witness-only and gated by the
                         # reference's flag; a wavefront (64 consecutive cycles of one instance) whose cycles all have it off skips the permutation
                         "witness_only_permutations_skipped_frac": float(np.mean(p2_skipped)),
                         "values_written_per_cycle": st["cells_written_loop"], "trace_cells_populated_per_cycle": st["cells_populated_loop"],
                         "trace_cell_equivalent_GBps": cell_bytes / (k_ms * 1e-3) / 1e9,
                         "hbm_busy_GBps": None if traffic is None else traffic / (k_ms * 1e-3) / 1e9,
                         "other_kernels_ms": {"loop_gates_plus_copies_check": float(np.mean(check_ms)), "k_check_gates_loop": float(np.mean(gate_ms)),
                                              "outer_post_and_checks_overlapped": float(np.mean(outer_ms))}},
            # sum (mod 2^64) of every gathered commitment word times its position + 1: cannot cancel like an XOR of repeated rows
            "commitment_checksum": int((flat * (np.arange(flat.size, dtype=np.uint64) + np.uint64(1))).sum(dtype=np.uint64)),
            "distinct_commitments": int(len({tuple(r) for r in commits.reshape(-1, commits.shape[-1]).tolist()})),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.log2_rows)
        else:
            out["cpu_baseline"] = None
        def _clean(x):   # --headline-only leaves the secondary figures unmeasured: null, not NaN
            if isinstance(x, dict): return {k: _clean(v) for k, v in x.items()}
            if isinstance(x, list): return [_clean(v) for v in x]
            if isinstance(x, float) and x != x: return None
            return x
        json_out.write(json.dumps(_clean(out)) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
