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
