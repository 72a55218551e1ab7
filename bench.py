#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: constraints/s (+ witness-rows/s) of main_vm at 2^20 rows per instance
(config C2), N MI355X, one process per GPU.

Workload = the REAL main_vm circuit (csrc/circuits/main_vm.cpp: vm_cycle of /root/reference/src/main_vm/cycle.rs:28-795 with
all eleven opcode families) executing synthetic zkEVM programs: tests/golden/vm_bench_witness.npz holds the raw WitnessOracle
words of 8 executions of an endless mixed program (far calls, returns, reverts, UMA, logs, arithmetic; generator:
tests/golden/make_vm_bench_witness.py), tiled over the B instances of the batch.  Instance = one `limit`-cycle chunk filling
2^20 trace rows.

A "step" = one pass of the hot path over one batch of B independent circuit instances per GPU whose inputs already live in HBM:
witness generation (outer pre, loop, outer post kernels) followed by the full satisfiability check (gate + lookup + copy + link
kernels).  The per-cycle VmLocalState the loop scope consumes is derived on the device from the raw oracle words
(zk_cs_seed_carried_inputs, a sequential chain per instance) BEFORE the timed region; its time is reported as
config.input_seeding_s and folded into `value_from_raw_witness` = constraints / (seeding + step).  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--log2-rows 20] [--no-cpu-baseline] [--workload main_vm|vm_shaped]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
VM_STATE_WORDS = 243   # VmLocalState, the carried part of the loop stream
FIXTURE = os.path.join(ROOT, "tests", "golden", "vm_bench_witness.npz")


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


def main_vm_streams(cs, limit, batch, first=0):
    """(outer [words, B], loop [words, B * limit] with the carried words zero, expected commitments [B, 4] or None) from the fixture"""
    fx = np.load(FIXTURE)
    lay = cs.main_vm_layout()
    if json.loads(bytes(fx["layout"]).decode()) != {k: {n: list(v) for n, v in d.items()} for k, d in lay.items()}:
        raise RuntimeError("tests/golden/vm_bench_witness.npz was generated for another stream layout: re-run tests/golden/make_vm_bench_witness.py")
    raw, tails, commits = fx["raw"], fx["rollback_tail"], fx["commitment"]
    n_exec, cycles, n_raw = raw.shape
    if cycles < limit:
        raise RuntimeError(f"fixture holds {cycles} cycles per execution, the circuit needs {limit}")
    n_outer, n_loop = cs.input_words()
    assert n_loop == VM_STATE_WORDS + n_raw
    outer = np.zeros((n_outer, batch), dtype=np.uint64)
    loop = np.zeros((n_loop, batch * limit), dtype=np.uint64)
    per_exec = [np.ascontiguousarray(raw[e, :limit].T) for e in range(n_exec)]   # [words, limit]
    t0 = lay["outer"]["rollback_queue_tail_for_block"][0]
    for i in range(batch):
        e = (first + i) % n_exec
        outer[lay["outer"]["start_flag"][0], i] = 1
        outer[t0:t0 + 4, i] = tails[e]
        loop[VM_STATE_WORDS:, i * limit:(i + 1) * limit] = per_exec[e]
    expect = np.stack([commits[(first + i) % n_exec] for i in range(batch)]) if int(fx["limit"][0]) == limit else None
    return outer, loop, expect


# ------------------------------------------------------------------------------------------------ round-1 micro-workload (kept for A/B)
def vm_shaped_inputs(rng, n_outer, n_loop, batch, limit):
    P = 0xFFFFFFFF00000001
    outer = rng.integers(0, 2**32, size=(n_outer, batch), dtype=np.uint64)
    outer[120:135] = rng.integers(0, 2, size=(15, batch))
    outer[135] = rng.integers(0, 2**16, size=batch)
    outer[138] = rng.integers(0, 2**30, size=batch)
    outer[139:142] = rng.integers(0, 2, size=(3, batch))
    outer[142:154] = rng.integers(0, 2**63, size=(12, batch), dtype=np.uint64) % np.uint64(P)
    outer[154] = rng.integers(0, 2**20, size=batch)
    outer[155:167] = rng.integers(0, 2**63, size=(12, batch), dtype=np.uint64) % np.uint64(P)
    loop = np.zeros((n_loop, batch * limit), dtype=np.uint64)
    loop[183:] = rng.integers(0, 2**32, size=(n_loop - 183, batch * limit), dtype=np.uint64)
    loop[183 + 16] = rng.integers(0, 2, size=batch * limit)
    return outer, loop


vm_inputs = vm_shaped_inputs  # name used by tests/test_gpu_cs.py


def build_vm_shaped_cs(zkgl, log2_rows):
    probe = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), max_trace_len=1 << 30)
    probe.configure_vm_shaped()
    probe.vm_shaped_entry_point(1)
    probe.pad_and_shrink()
    st = probe.stats()
    limit = ((1 << log2_rows) - st["outer_slots"]) // st["loop_slots"]
    probe.close()
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), max_trace_len=1 << log2_rows)
    cs.configure_vm_shaped()
    cs.vm_shaped_entry_point(limit)
    cs.pad_and_shrink()
    return cs, limit


build_vm_cs = build_vm_shaped_cs


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
    outer, loop, _ = main_vm_streams(cs, limit, n_inst)
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
                      f"median of {len(times)} passes: resolve {med[1]:.2f}s + check {med[2]:.2f}s (+ sequential seeding {t_seed:.2f}s, outside `value`), "
                      f"OpenMP {cores} threads over instances x cycles, Goldilocks reduction by the 2^64 = 2^32 - 1 identity; "
                      f"CPU restatement (oracle), not the reference Rust binary"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=384, help="independent circuit instances per GPU per step")
    ap.add_argument("--seed-windows", type=int, default=5, help="batches of raw witness seeded in one zk_cs_seed_stream pass (the stream = batch x windows instances)")
    ap.add_argument("--log2-rows", type=int, default=20)
    ap.add_argument("--workload", default="main_vm", choices=["main_vm", "vm_shaped"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise RuntimeError("bench.py needs a GPU: libzkgl has no CPU fallback")
    dev_index = local_rank % n_dev
    shared_gpu = world > n_dev            # only in smoke tests of the N>1 path on a 1-GPU box: RCCL refuses duplicate devices
    if world > 1:
        dist.init_process_group("gloo" if shared_gpu else "nccl", rank=rank, world_size=world)
    torch.cuda.set_device(dev_index)
    zkgl.init(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = torch.device("cpu") if shared_gpu else dev

    B = args.batch
    K = max(1, args.seed_windows)     # the raw witness stream holds K batches: seeded in ONE pass, resolved window by window
    S = B * K
    expect = None
    stream = torch.cuda.current_stream().cuda_stream
    if args.workload == "main_vm":
        cs, limit = build_main_vm_cs(zkgl, args.log2_rows)
        n_outer, n_loop = cs.input_words()
        # the stream is assembled on the device: instance i replays execution (rank * S + i) % n_exec of the fixture
        outer8, loop8, expect8 = main_vm_streams(cs, limit, 8, first=0)
        n_exec = 8
        sel = (torch.arange(S, device=dev) + rank * S) % n_exec
        d_outer = torch.from_numpy(outer8.view(np.int64)).to(dev)[:, sel].contiguous()
        l8 = torch.from_numpy(loop8.view(np.int64)).to(dev).view(n_loop, n_exec, limit)
        d_loop = l8[:, sel, :].reshape(n_loop, S * limit).contiguous()
        del l8, loop8
        expect = None if expect8 is None else expect8[((np.arange(S) + rank * S) % n_exec)]
        state_words = VM_STATE_WORDS
    else:
        cs, limit = build_vm_shaped_cs(zkgl, args.log2_rows)
        n_outer, n_loop = cs.input_words()
        outer, loop = vm_shaped_inputs(np.random.default_rng(0xC2 + rank), n_outer, n_loop, S, limit)
        d_outer = torch.from_numpy(outer.view(np.int64)).to(dev)
        d_loop = torch.from_numpy(loop.view(np.int64)).to(dev)
        del loop
        state_words = 183
    st = cs.stats()
    cs.set_batch(B)
    seed_s = []
    for _ in range(2):  # the second pass overwrites the carried words with the same values: a clean timing of the seeding alone
        torch.cuda.synchronize()
        t = time.perf_counter()
        cs.seed_stream(S, d_outer, d_loop, stream)   # zk_cs_seed_stream: the sequential part, all K windows at once
        torch.cuda.synchronize()
        seed_s.append(time.perf_counter() - t)
    t_seed = min(seed_s)
    window = [0]

    def bind(k):
        cs.bind_inputs(False, d_outer, n_outer, lane_stride=S, lane_offset=k * B)
        cs.bind_inputs(True, d_loop, n_loop, lane_stride=S * limit, lane_offset=k * B * limit)
        window[0] = k

    bind(0)

    step_no = [0]

    def step():
        bind(step_no[0] % K)   # every step takes the next window of the seeded stream
        step_no[0] += 1
        ok, failure = cs.resolve_and_check(stream)  # witness generation + full satisfiability check, one pipeline
        if not ok and not os.environ.get("ZKGL_STUB_RUN"):  # ZKGL_STUB_RUN: tools/stub_bench.sh times deliberately wrong kernel variants
            raise RuntimeError(f"trace not satisfied: {failure}")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    loop_ms, check_ms, gate_ms, outer_ms, step_ms = [], [], [], [], []
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
        loop_ms.append(cs.last_ms(1)); check_ms.append(cs.last_ms(2)); gate_ms.append(cs.last_ms(3)); outer_ms.append(cs.last_ms(4))
    fence()
    elapsed_local = time.perf_counter() - t0
    # the path's only collective: gather the 4-element input commitments of every instance
    from zkgl.dist import gather_commitments, gather_floats, max_over_ranks
    elapsed = max_over_ranks(elapsed_local, coll_dev)
    seed_all = max_over_ranks(t_seed, coll_dev)
    per_rank_ms = gather_floats(1e3 * elapsed_local / args.steps, coll_dev)
    local = np.array([cs.public_inputs(i) for i in range(B)], dtype=np.uint64)
    parity = None
    if expect is not None:
        parity = bool(np.array_equal(local, expect[window[0] * B:(window[0] + 1) * B]))
        if not parity and not os.environ.get("ZKGL_STUB_RUN"):
            raise RuntimeError("public inputs differ from the native restatement's commitments stored in the fixture")
    # the collective behind the C ABI: zk_comm_* + zk_cs_gather_commitments (one ncclAllGather of the packed public inputs over
    # RCCL / xGMI); the launcher's part — handing rank 0's unique id to the other ranks — is a torch.distributed broadcast here
    gather_path = "zk_cs_gather_commitments (RCCL all-gather behind the C ABI)"
    import threading
    attempt = {}

    def c_abi_gather():   # on its own thread with a deadline: a communicator that cannot be formed must not cost the bench line
        try:
            torch.cuda.set_device(dev_index)   # the current device is per thread
            uid = torch.zeros(128, dtype=torch.uint8, device=coll_dev)
            if rank == 0:
                uid = torch.frombuffer(bytearray(zkgl.Comm.unique_id()), dtype=torch.uint8).to(coll_dev)
            if world > 1:
                dist.broadcast(uid, src=0)
            comm = zkgl.Comm(bytes(uid.cpu().numpy().tobytes()), rank, world)
            got = cs.gather_commitments(comm, stream)          # [world, B, 4] u64
            torch.cuda.synchronize()
            comm.close()
            if not np.array_equal(got[rank], local):
                raise RuntimeError("gathered commitments differ from this rank's public inputs")
            attempt["commits"] = got
        except Exception as e:  # noqa: BLE001
            attempt["error"] = e

    hung = False
    if shared_gpu:
        attempt["error"] = RuntimeError("ranks share one GPU (smoke test): RCCL refuses duplicate devices")
    else:
        th = threading.Thread(target=c_abi_gather, daemon=True)
        th.start()
        th.join(180.0)
        if th.is_alive():
            hung = True
            attempt["error"] = TimeoutError("no communicator within 180 s")
    ok_everywhere = 1 if "commits" in attempt else 0
    if world > 1 and not hung:   # every rank takes the same path
        flag = torch.tensor([ok_everywhere], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_everywhere = int(flag.item())
    if ok_everywhere:
        commits = attempt["commits"]
    else:
        print(f"[bench] C-ABI gather unavailable ({attempt.get('error', 'another rank failed')}); using torch.distributed.all_gather", file=sys.stderr)
        gather_path = "torch.distributed all_gather (fallback)"
        commits = gather_commitments(local, coll_dev)   # [world, B, 4] u64: RCCL all_gather over xGMI when world > 1
    if rank == 0:
        n_inst = B * world
        constraints = st["constraints_per_instance"] * n_inst * args.steps
        rows = st["rows_per_instance"] * n_inst * args.steps
        # dominant kernel: the loop-scope witness interpreter.  ALGORITHMIC bytes per launch = every witness VALUE of the loop
        # rows written once (8 B; one per variable — the trace is a view of the variable store) + every input word
        # read once.  SURVEY §8(d) counts every trace CELL (a variable occupies 3.1 cells on average): that figure is reported
        # beside it as trace_cell_equivalent_GBps, it is not what the kernel has to move.
        algo_bytes = B * st["limit"] * (st["cells_written_loop"] + n_loop) * 8
        cell_bytes = B * st["limit"] * (st["cells_populated_loop"] + n_loop) * 8
        k_ms = float(np.mean(loop_ms))
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_r2.json")))
            traffic = pmc["traffic_over_algorithmic"] * algo_bytes
            traffic_src = f"profiles/pmc_r2.json ratio {pmc['traffic_over_algorithmic']:.3f} measured at batch {pmc['batch']}"
        except Exception:
            pass
        step_s = elapsed / args.steps
        out = {
            "metric": "constraints/s + witness-rows/s, main_vm 2^20 rows", "value": constraints / elapsed, "unit": "constraints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * step_s,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks)", "data": "synthetic",
            "witness_rows_per_s": rows / elapsed,
            # the whole path from the raw witness: one seeding pass over the K-window stream + K steps
            "value_from_raw_witness": st["constraints_per_instance"] * n_inst * K / (seed_all + K * step_s),
            "config": {"workload": ("main_vm (real vm_cycle, 11 opcode families; synthetic zkEVM programs from tests/golden/vm_bench_witness.npz)"
                                    if args.workload == "main_vm" else "main_vm-shaped micro-workload (round 1)") +
                                   f", geometry 140/0/8/deg8 + 3x8 lookups, 2^{args.log2_rows} rows/instance",
                       "instances_per_gpu": B, "cycles_per_instance": limit, "rows_per_instance": st["rows_per_instance"],
                       "constraints_per_instance": st["constraints_per_instance"], "parallelism": f"independent instances x{world}",
                       "input_seeding_s": round(seed_all, 4), "seeded_stream_instances_per_gpu": S, "seeding_s_per_batch": round(seed_all / K, 4), "per_rank_ms_per_step": [round(x, 3) for x in per_rank_ms],
                       "commitments_equal_native_restatement": parity, "commitment_gather": gather_path},
            "roofline": {"bound": "hbm", "kernel": "zke::k_witness_loop", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": algo_bytes,
                         "avg_launch_ms": k_ms,
                         "values_written_per_cycle": st["cells_written_loop"], "trace_cells_populated_per_cycle": st["cells_populated_loop"],
                         "trace_cell_equivalent_GBps": cell_bytes / (k_ms * 1e-3) / 1e9,
                         "hbm_busy_GBps": None if traffic is None else traffic / (k_ms * 1e-3) / 1e9,
                         "other_kernels_ms": {"loop_gates_plus_copies_check": float(np.mean(check_ms)), "k_check_gates_loop": float(np.mean(gate_ms)),
                                              "outer_post_and_checks_overlapped": float(np.mean(outer_ms))}},
            "commitment_checksum": int(np.bitwise_xor.reduce(commits.reshape(-1))) & 0xFFFFFFFFFFFF,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.log2_rows)
        else:
            out["cpu_baseline"] = None
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if hung:
        os._exit(0)   # a thread is still inside a collective that will never complete
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
