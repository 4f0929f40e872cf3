"""Body of tests/test_gpu_parity.py::test_device_entry_point_overlapped_slots (own interpreter:
torch first, then the library)."""
import sys

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
import pick_ik_amd as pk  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pick_ik_amd import robots  # noqa: E402
from tests.common import random_targets  # noqa: E402

import os  # noqa: E402
s = pk.Solver(robots.panda(), device=0, exact=(None if os.environ.get("PIK_CHECK_FLAVOUR", "exact") == "exact" else False))
o = O.Oracle(s.chain)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(12)
S, rounds = 4, 5
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
p = pk.default_params(memetic_population_size=32, memetic_max_generations=20)
jobs = []
for r in range(rounds):
    for slot in range(S):
        B = int(rng.integers(1, 700))
        _, goal = random_targets(o.fk, s.chain, rng, B)
        seed = np.tile(robots.PANDA_HOME, (B, 1))
        seed[::5] = rng.uniform(s.chain.qmin, s.chain.qmax, size=seed[::5].shape)
        g, sd = torch.from_numpy(goal).to(dev), torch.from_numpy(seed).to(dev)
        sol = torch.empty(B, 7, dtype=torch.float64, device=dev)
        st = torch.zeros(B, dtype=torch.int32, device=dev)
        c = torch.empty(B, dtype=torch.float64, device=dev)
        jobs.append((slot, r, goal, seed, g, sd, sol, st, c))
torch.cuda.synchronize()
for slot, r, goal, seed, g, sd, sol, st, c in jobs:  # enqueue everything, no waiting in between
    with torch.cuda.stream(streams[slot]):
        s.solve_batch_device(p, len(goal), g.data_ptr(), sd.data_ptr(), sol.data_ptr(), st.data_ptr(),
                             c.data_ptr(), 0, rng_seed=100 + r, problem_offset=7 * slot,
                             stream=streams[slot].cuda_stream, slot=slot)
torch.cuda.synchronize()
for slot, r, goal, seed, g, sd, sol, st, c in jobs:
    ref = s.solve_batch(p, goal, seed, rng_seed=100 + r, problem_offset=7 * slot)
    np.testing.assert_array_equal(sol.cpu().numpy(), ref[0], err_msg=f"slot {slot} round {r}")
    np.testing.assert_array_equal(st.cpu().numpy(), ref[1])
    np.testing.assert_array_equal(c.cpu().numpy(), ref[2])
s.close()
print("overlap check OK")
