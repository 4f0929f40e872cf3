"""N > 1 path on CPU: two processes (gloo), contiguous shards with global problem offsets, final
all-gather.  The per-shard solver here is the ORACLE (test infrastructure) because the product has no
CPU path; what is under test is the decomposition + gather logic bench.py / a multi-GPU caller uses:
the gathered result must equal the single-process solve of the whole batch bit for bit."""
import os
import socket

import numpy as np
import pytest

from pick_ik_amd import robots
from pick_ik_amd.distributed import all_gather_results, gather_results, shard_bounds, solve_shard


def test_shard_bounds_cover_without_overlap():
    for total in (0, 1, 7, 8, 4096, 1048576 + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, out_dir):
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ch = robots.panda()
    o = O.Oracle(ch)
    rng = np.random.default_rng(2024)  # same data on every rank
    q = rng.uniform(ch.qmin, ch.qmax, size=(total, 7))
    goals = o.fk(q)
    seeds = np.tile(robots.PANDA_HOME, (total, 1))
    p = O.default_params()

    def solve_fn(g, s, rng_seed, problem_offset):
        return o.solve_batch(p, g, s, rng_seed=rng_seed, problem_offset=problem_offset)

    (lo, hi), (sol, status, cost, _) = solve_shard(solve_fn, goals, seeds, rank, world, rng_seed=77)
    assert (lo, hi) == shard_bounds(total, rank, world)
    full = all_gather_results(sol, status, cost, total)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), sol=full[0], status=full[1], cost=full[2])
    if total % world == 0:
        # the gather bench.py itself uses (equal shards, preallocated buffers, one collective each)
        import torch
        n = total // world
        buf = (torch.empty(total, 7, dtype=torch.float64), torch.empty(total, dtype=torch.int32))
        gather_results(dist, torch.from_numpy(sol), torch.from_numpy(status), buf[0], buf[1])
        assert np.array_equal(buf[0].numpy(), full[0]) and np.array_equal(buf[1].numpy(), full[1])
        assert np.array_equal(buf[0].numpy()[rank * n:(rank + 1) * n], sol)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [37, 64])
def test_two_rank_sharded_solve_equals_single_call(tmp_path, oracle_mod, total):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    O = oracle_mod
    ch = robots.panda()
    o = O.Oracle(ch)
    rng = np.random.default_rng(2024)
    goals = o.fk(rng.uniform(ch.qmin, ch.qmax, size=(total, 7)))
    seeds = np.tile(robots.PANDA_HOME, (total, 1))
    ref = o.solve_batch(O.default_params(), goals, seeds, rng_seed=77, num_threads=4)
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_array_equal(got["sol"], ref[0])
        np.testing.assert_array_equal(got["status"], ref[1])
        np.testing.assert_array_equal(got["cost"], ref[2])
