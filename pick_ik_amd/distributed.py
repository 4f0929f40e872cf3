"""Multi-GPU decomposition of a batch of IK problems.

Every target pose is an independent problem: the batch is cut into contiguous shards, one per rank
(one process per GPU), each shard is solved with `problem_offset` = its first global index -- the
random streams are keyed by the GLOBAL problem index, so the sharded job returns exactly what a
single call over the whole batch returns -- and the only collective is the final gather of
solutions / status / cost (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous range [lo, hi) of rank's problems; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def solve_shard(solve_fn, goals, seeds, rank: int, world: int, rng_seed: int = 0):
    """solve_fn(goal_shard, seed_shard, rng_seed=..., problem_offset=...) -> (sol, status, cost, ...)"""
    lo, hi = shard_bounds(len(goals), rank, world)
    out = solve_fn(goals[lo:hi], seeds[lo:hi], rng_seed=rng_seed, problem_offset=lo)
    return (lo, hi), out


def all_gather_results(sol, status, cost, total: int, group=None, device=None):
    """Gathers the ragged shards of every rank into full [total, ...] arrays on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dof = sol.shape[1]
    cap = -(-total // world)  # ceil: pad shards to a common size for all_gather_into_tensor
    dev = device or "cpu"

    def pad(a, width, dtype):
        t = torch.zeros((cap,) + ((width,) if width else ()), dtype=dtype, device=dev)
        t[: len(a)] = torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)
        return t

    parts = []
    for a, width, dtype in ((sol, dof, torch.float64), (status, 0, torch.int32), (cost, 0, torch.float64)):
        mine = pad(a, width, dtype)
        full = torch.empty((world * cap,) + tuple(mine.shape[1:]), dtype=dtype, device=dev)
        dist.all_gather_into_tensor(full, mine, group=group)
        keep = []
        for r in range(world):
            lo, hi = shard_bounds(total, r, world)
            keep.append(full[r * cap: r * cap + (hi - lo)])
        parts.append(torch.cat(keep).cpu().numpy())
    del rank
    return tuple(parts)


shard_range = shard_bounds


def gather_results(dist, sol, status, out_sol, out_status, group=None):
    """bench.py's final gather: equal-sized shards already on the device, one all-gather each for
    the solutions [n][dof] and the status words [n] into [world * n ...] buffers (rank r's shard at
    position r).  `dist` = torch.distributed (RCCL when the group's backend is "nccl")."""
    dist.all_gather_into_tensor(out_sol, sol.contiguous(), group=group)
    dist.all_gather_into_tensor(out_status, status.contiguous(), group=group)
    return out_sol, out_status
