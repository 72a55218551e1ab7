"""Multi-GPU plumbing of the hot path: independent circuit instances are sharded across ranks with
no data-path collective; the only exchange is ONE all-gather of the 4-element input commitments
(`input_commitment`, /root/reference/src/ram_permutation/mod.rs:203-209) — RCCL over xGMI on GPUs
(backend "nccl"), gloo in the CPU tests.  Payload is 32 B per instance: latency-bound."""
from __future__ import annotations

import numpy as np


def shard_instances(n_total: int, rank: int, world: int):
    """instance i -> rank i % world; returns the global instance ids of `rank`"""
    return list(range(rank, n_total, world))


def gather_commitments(local: np.ndarray, device=None):
    """local: u64 [n_local, 4] -> u64 [world, n_local, 4] on every rank (all ranks hold equal n_local)"""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint64).view(np.int64))
    if device is not None:
        t = t.to(device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t.cpu().numpy().view(np.uint64)[None]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy().view(np.uint64)


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device=None):
    """one float per rank -> list over ranks (per-rank step times of bench.py, so that a multi-GPU run is diagnosable)"""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
