"""N>1 path on CPU: world_size-2 gloo processes shard independent circuit instances (no data-path
collective) and all-gather the 4-element commitments; the result must equal the single-process run."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, limit, q):
    for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import oracle_run, ram_cs, random_instances
    from oracle import ram_native as rn
    from zkgl.dist import gather_commitments, max_over_ranks, shard_instances

    insts = random_instances(77, n_total, 5, limit)          # same seeded global work list on every rank
    mine = shard_instances(n_total, rank, world)
    cs = ram_cs(limit)
    outer, loop = rn.pack_streams([insts[i] for i in mine], limit)
    run = oracle_run(cs, outer, loop, len(mine))             # CPU stand-in for the GPU engine in this test
    assert run.check()[0] == 0
    local = np.array([[int(run.oc[c, j]) for c in cs.public_cells()] for j in range(len(mine))], dtype=np.uint64)
    allc = gather_commitments(local)
    t = max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((allc.tolist(), t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_commitment_gather():
    n_total, limit, world = 6, 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, limit, q)) for r in range(world)]
    for p in procs:
        p.start()
    allc, t = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0  # max over ranks
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import random_instances
    insts = random_instances(77, n_total, 5, limit)
    allc = np.array(allc, dtype=np.uint64)
    assert allc.shape == (world, n_total // world, 4)
    for r in range(world):
        for j, i in enumerate(range(r, n_total, world)):
            assert [int(x) for x in allc[r, j]] == insts[i]["commitment"]


# ---- the same on the GPU: every rank drives libzkgl (the product path) on the visible device; the two ranks of this test share one
# GPU, where RCCL refuses duplicate devices, so the gather here is the gloo face of zkgl.dist — the RCCL collective behind the C ABI
# (zk_cs_gather_commitments) is covered by tests/test_gpu_main_vm.py on a one-rank communicator and by bench.py --gpus N.
import pytest  # noqa: E402


def _gpu_worker(rank, world, port, n_total, limit, q):
    for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zkgl
    from helpers import ram_cs, random_instances
    from oracle import ram_native as rn
    from zkgl.dist import gather_commitments, shard_instances

    zkgl.init(0)
    insts = random_instances(77, n_total, 5, limit)
    mine = shard_instances(n_total, rank, world)
    cs = ram_cs(limit)
    outer, loop = rn.pack_streams([insts[i] for i in mine], limit)
    cs.set_batch(len(mine))
    d_o, d_l = zkgl.DeviceBuffer.from_numpy(outer), zkgl.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    ok, failure = cs.resolve_and_check()
    assert ok, failure
    local = np.array([cs.public_inputs(j) for j in range(len(mine))], dtype=np.uint64)
    allc = gather_commitments(local)
    if rank == 0:
        q.put(allc.tolist())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_drive_libzkgl_on_the_gpu():
    n_total, limit, world = 6, 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, n_total, limit, q)) for r in range(world)]
    for p in procs:
        p.start()
    allc = np.array(q.get(timeout=300), dtype=np.uint64)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import random_instances
    insts = random_instances(77, n_total, 5, limit)
    for r in range(world):
        for j, i in enumerate(range(r, n_total, world)):
            assert [int(x) for x in allc[r, j]] == insts[i]["commitment"]


# ---- the RCCL collective itself on two real devices: one process per GPU, rank 0's unique id over the gloo control plane, ONE
# communicator (zk_comm_create), zk_cs_gather_commitments = one ncclAllGather.  Needs two GPUs: skipped on a one-GPU box.
def _rccl_worker(rank, world, port, n_total, limit, q):
    for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zkgl
    from helpers import ram_cs, random_instances
    from oracle import ram_native as rn
    from zkgl.dist import shard_instances

    zkgl.init(rank % zkgl.device_count())   # one process per GPU (the emulated device of tests/emu has one: both ranks meet there)
    insts = random_instances(78, n_total, 5, limit)
    mine = shard_instances(n_total, rank, world)
    cs = ram_cs(limit)
    outer, loop = rn.pack_streams([insts[i] for i in mine], limit)
    cs.set_batch(len(mine))
    d_o, d_l = zkgl.DeviceBuffer.from_numpy(outer), zkgl.DeviceBuffer.from_numpy(loop)
    cs.bind_inputs(False, d_o, outer.shape[0])
    cs.bind_inputs(True, d_l, loop.shape[0])
    ok, failure = cs.resolve_and_check()
    assert ok, failure
    ids = [zkgl.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    comm = zkgl.Comm(bytes(ids[0]), rank, world)
    got = cs.gather_commitments(comm)    # [world, batch, 4]
    comm.close()
    q.put((rank, got.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_gather_behind_the_c_abi_on_two_gpus():
    import zkgl
    from helpers import emulated_device
    if zkgl.device_count() < 2 and not emulated_device():
        pytest.skip("needs two GPUs (the driver's multi-GPU tier); a one-GPU box covers the collective on a one-rank communicator")
    # (on the emulated device the two PROCESSES meet in the stand-in collective of tests/emu/dev/rccl/rccl.h: what is checked there is that comm.cpp and the
    #  host above it hand a world of two the right ranks, counts and buffers — not RCCL)
    n_total, limit, world = 8, 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, n_total, limit, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import random_instances
    insts = random_instances(78, n_total, 5, limit)
    for viewer in range(world):            # every rank holds every rank's commitments
        allc = np.array(results[viewer], dtype=np.uint64)
        for r in range(world):
            for j, i in enumerate(range(r, n_total, world)):
                assert [int(x) for x in allc[r, j]] == insts[i]["commitment"]
