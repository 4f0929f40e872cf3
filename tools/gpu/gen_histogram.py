#!/usr/bin/env python3
"""How many problems of the driver's pool (20 batches of 4096 Panda targets, population 128) are still running at every
generation mark of the schedule -- the occupancy of each compaction pass.  usage: gen_histogram.py [exact|fast]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pick_ik_amd as pk
from pick_ik_amd.solver import Batch

which = sys.argv[1] if len(sys.argv) > 1 else "exact"
chain = pk.robots.panda()
dev = torch.device("cuda", 0)
s = pk.Solver(chain, device=0, exact=(which == "exact"))
B, K, D = 4096, 20, 7
rng = np.random.default_rng(0x5049434B)
f64 = dict(dtype=torch.float64, device=dev)
seed_t = torch.from_numpy(np.tile(pk.robots.PANDA_HOME, (B, 1))).to(dev)
goals, sols, status, costs, stats = [], [], [], [], []
for _ in range(K):
    q = torch.from_numpy(rng.uniform(chain.qmin, chain.qmax, size=(B, D))).to(dev)
    g = torch.empty(B, 7, **f64)
    s.fk_device(B, q.data_ptr(), g.data_ptr(), torch.cuda.current_stream().cuda_stream)
    goals.append(g); sols.append(torch.empty(B, D, **f64)); status.append(torch.zeros(B, dtype=torch.int32, device=dev))
    costs.append(torch.empty(B, **f64)); stats.append(torch.zeros(B, 3, dtype=torch.int64, device=dev))
torch.cuda.synchronize()
params = pk.default_params(memetic_population_size=128, memetic_elite_size=4, memetic_max_generations=100)
recs = [Batch(B, goals[i].data_ptr(), seed_t.data_ptr(), None, i * B, sols[i].data_ptr(), status[i].data_ptr(),
              costs[i].data_ptr(), stats[i].data_ptr(), None) for i in range(K)]
s.solve_batches_device(params, recs, rng_seed=1234, stream=0, slot=0)
torch.cuda.synchronize()
st = torch.cat(stats).cpu().numpy()
gens = (st[:, 1] & 0xffffffff).astype(np.int64)
evals = st[:, 0]
ok = torch.cat(status).cpu().numpy() == pk.SUCCESS
n = len(gens)
print(f"{which}: {n} problems, success {ok.mean():.4f}, mean generations {gens.mean():.2f}, mean evaluations {evals.mean():.0f}")
print("problems that run generation g (i.e. concluded in a generation > g, or never): mark, count, share, wavefronts at 1/2/4/8/16 lanes per elite")
for m in (0, 1, 2, 4, 8, 12, 16, 24, 32, 40, 48, 64, 80, 99):
    c = int((gens > m).sum())
    print(f"  gen {m:3d}: {c:6d}  {c / n:7.4f}   " + " / ".join(str((c * 4 * l + 63) // 64) for l in (1, 2, 4, 8, 16)))
h = np.bincount(gens, minlength=101)
print("concluded in generation g:", " ".join(f"{g}:{int(h[g])}" for g in range(101) if h[g]))
