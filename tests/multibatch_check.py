"""Body of tests/test_gpu_multibatch.py::test_device_multi_batch_pool (own interpreter: torch first,
then the library).  K HBM-resident batches solved as ONE pool by pikamd_solve_batches_device must
each equal, bit for bit, the batch solved by a call of its own; the per-batch completion counters
must read B when the call is done."""
import sys

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
import pick_ik_amd as pk  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pick_ik_amd import robots  # noqa: E402
from pick_ik_amd.solver import Batch, STATS_DTYPE  # noqa: E402
from tests.common import random_targets  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)


def run(chain, strict, cases, sizes, exact=None):
    s = pk.Solver(chain, device=0, strict=strict, exact=exact)
    o = O.Oracle(chain)
    D, T = s.dof, s.n_tips
    for mode, kw in cases:
        p = pk.default_params(**kw)
        host, devb, recs = [], [], []
        off = 1000
        for B in sizes:
            q = rng.uniform(chain.qmin, chain.qmax, size=(B, D))
            goal = o.fk(q).reshape(B, 7 * T)
            seed = np.clip(q + rng.normal(0, 0.3, size=q.shape), chain.qmin, chain.qmax)
            guess = seed.copy()
            guess[::4] = rng.uniform(chain.qmin, chain.qmax, size=guess[::4].shape)
            t = dict(goal=torch.from_numpy(goal).to(dev), seed=torch.from_numpy(seed).to(dev),
                     guess=torch.from_numpy(guess).to(dev),
                     sol=torch.empty(B, D, dtype=torch.float64, device=dev),
                     st=torch.zeros(B, dtype=torch.int32, device=dev),
                     c=torch.empty(B, dtype=torch.float64, device=dev),
                     stats=torch.zeros(B, 3, dtype=torch.int64, device=dev),
                     done=torch.zeros(1, dtype=torch.int32, device=dev))
            host.append((goal, seed, guess, off))
            devb.append(t)
            recs.append(Batch(B, t["goal"].data_ptr(), t["seed"].data_ptr(), t["guess"].data_ptr(), off,
                              t["sol"].data_ptr(), t["st"].data_ptr(), t["c"].data_ptr(),
                              t["stats"].data_ptr(), t["done"].data_ptr()))
            off += B + 5
        torch.cuda.synchronize()
        st = torch.cuda.Stream(device=dev)
        for rep in range(2):  # twice on the same slot: the kernels re-arm their queue counters
            for t in devb:
                t["done"].zero_()
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                s.solve_batches_device(p, recs, rng_seed=77, stream=st.cuda_stream, slot=3)
            torch.cuda.synchronize()
            for k, ((goal, seed, guess, off_k), t) in enumerate(zip(host, devb)):
                ref = s.solve_batch(p, goal, seed, rng_seed=77, problem_offset=off_k, initial_guess=guess)
                what = f"{chain.name} strict={strict} exact={exact} mode {mode} batch {k}"
                np.testing.assert_array_equal(t["sol"].cpu().numpy(), ref[0], err_msg=what)
                np.testing.assert_array_equal(t["st"].cpu().numpy(), ref[1], err_msg=what)
                np.testing.assert_array_equal(t["c"].cpu().numpy(), ref[2], err_msg=what)
                got_stats = t["stats"].cpu().numpy().view(STATS_DTYPE).reshape(-1)
                np.testing.assert_array_equal(got_stats, ref[3], err_msg=what)
                assert int(t["done"].item()) == len(goal), (what, int(t["done"].item()), len(goal))
    s.close()


for exact in (None, False):  # the library's default arithmetic (exact), then the opt-in fast flavour
    run(robots.panda(), False,
        ((0, dict(memetic_population_size=32, memetic_max_generations=24)), (1, dict(mode=1))),
        [700, 1, 333, 0, 64, 1500, 17], exact=exact)
# several tip frames, strict build (the kernels with the highest register pressure)
run(robots.torso_dual_arm(), True,
    ((0, dict(memetic_population_size=24, memetic_max_generations=12)), (1, dict(mode=1, gd_max_iters=30))),
    [96, 5, 130])
print("multi-batch check OK")
