#!/usr/bin/env python3
"""Where a generation's time goes: the same 4096-problem Panda batch with parts of the algorithm
switched off through its own parameters (single launch, PIK_PASSES=none, fixed generation count).
The batch is sized to one wavefront per SIMD for each lanes-per-elite setting (1024 wavefronts), so
the figures are the latency of ONE wavefront's generation.  usage: tools/ablate.py [lpe ...]"""
import os
import sys
import time

import numpy as np

os.environ["PIK_PASSES"] = "none"
sys.path.insert(0, ".")
import torch  # noqa: E402
import pick_ik_amd as pk  # noqa: E402

ch = pk.robots.panda()
s = pk.Solver(ch)
rng = np.random.default_rng(0)
B = 16384
q = rng.uniform(ch.qmin, ch.qmax, size=(B, 7))
goal = s.fk(q)
goal[:, :3] *= 3.0  # unreachable: nobody finishes early, every generation is run
dev = torch.device("cuda", 0)
g = torch.from_numpy(goal).to(dev)
sd = torch.from_numpy(np.tile(pk.robots.PANDA_HOME, (B, 1))).to(dev)
sol = torch.empty(B, 7, dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev)
G = 8
WAVES = int(os.environ.get("PIK_ABLATE_WAVES", "1024"))  # wavefronts in flight (1024 = one per SIMD)
for lpe in (sys.argv[1:] or ["1", "4", "16"]):
    os.environ["PIK_LPE"] = lpe
    B = 16 * WAVES // int(lpe)
    for name, kw in (("full generation", dict()), ("no gradient descent", dict(memetic_gd_max_iters=0)),
                     ("one child (P = E + 1)", dict(memetic_population_size=5)),
                     ("one child, 1 GD iteration", dict(memetic_population_size=5, memetic_gd_max_iters=1)),
                     ("one child, no GD", dict(memetic_population_size=5, memetic_gd_max_iters=0))):
        kw = dict(dict(memetic_population_size=128, memetic_max_generations=G, memetic_wipeout_fitness_tol=-1e300), **kw)
        p = pk.default_params(**kw)
        ts = []
        for r in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.solve_batch_device(p, B, g.data_ptr(), sd.data_ptr(), sol.data_ptr(), st.data_ptr(), rng_seed=r)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"waves {WAVES:5d} LPE {lpe}  {name:28s} {min(ts[1:]) * 1e3 / G:8.3f} ms per generation")
