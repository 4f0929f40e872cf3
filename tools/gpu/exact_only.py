#!/usr/bin/env python3
"""The exact kernels alone on the driver's pool (20 batches of 4096 Panda targets, population 128): one warm-up
call and one timed call -- a short program to put under rocprofv3.  usage: exact_only.py [exact|strict|fast]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pick_ik_amd as pk
from pick_ik_amd.solver import Batch

which = sys.argv[1] if len(sys.argv) > 1 else "exact"
chain = pk.robots.panda()
dev = torch.device("cuda", 0)
s = pk.Solver(chain, device=0, strict=(which == "strict"), exact=(which == "exact"))
B, K, D = 4096, 20, 7
rng = np.random.default_rng(0x5049434B)
f64 = dict(dtype=torch.float64, device=dev)
seed_t = torch.from_numpy(np.tile(pk.robots.PANDA_HOME, (B, 1))).to(dev)
goals, sols, status, costs, stats = [], [], [], [], []
for _ in range(2 * K):
    q = torch.from_numpy(rng.uniform(chain.qmin, chain.qmax, size=(B, D))).to(dev)
    g = torch.empty(B, 7, **f64)
    s.fk_device(B, q.data_ptr(), g.data_ptr(), torch.cuda.current_stream().cuda_stream)
    goals.append(g); sols.append(torch.empty(B, D, **f64)); status.append(torch.zeros(B, dtype=torch.int32, device=dev))
    costs.append(torch.empty(B, **f64)); stats.append(torch.zeros(B, 3, dtype=torch.int64, device=dev))
torch.cuda.synchronize()
params = pk.default_params(memetic_population_size=128, memetic_elite_size=4, memetic_max_generations=100)
st = torch.cuda.Stream(device=dev)
s.reserve(params, B * K, slot=0, stream=st.cuda_stream)
for rep in range(2):
    recs = [Batch(B, goals[i].data_ptr(), seed_t.data_ptr(), None, i * B, sols[i].data_ptr(), status[i].data_ptr(),
                  costs[i].data_ptr(), stats[i].data_ptr(), None) for i in range(rep * K, rep * K + K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(st):
        s.solve_batches_device(params, recs, rng_seed=1234, stream=st.cuda_stream, slot=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = sum(int((status[i] == pk.SUCCESS).sum()) for i in range(rep * K, rep * K + K))
    print(f"{which} rep {rep}: {dt * 1e3:.2f} ms, {ok / dt / 1e6:.3f} M solves/s, success {ok / (K * B):.4f}", flush=True)

# ... and one isolated 4096-target batch at a time (the tail alone: 100 generations of the widest variants), median of 7
if "--single" in sys.argv:
    ms = []
    for rep in range(9):
        i = rep % K
        recs = [Batch(B, goals[i].data_ptr(), seed_t.data_ptr(), None, i * B, sols[i].data_ptr(), status[i].data_ptr(),
                      costs[i].data_ptr(), stats[i].data_ptr(), None)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            s.solve_batches_device(params, recs, rng_seed=1234, stream=st.cuda_stream, slot=0)
        torch.cuda.synchronize()
        if rep >= 2:
            ms.append((time.perf_counter() - t0) * 1e3)
    ms.sort()
    print(f"{which} single batch: median {ms[len(ms) // 2]:.2f} ms (min {ms[0]:.2f}, max {ms[-1]:.2f})", flush=True)
