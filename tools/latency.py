#!/usr/bin/env python3
"""Wall-clock latency of the synchronous host-pointer entry point (pikamd_solve_batch: H2D, all
passes, D2H) for small batches -- what a MoveIt plugin call (B = 1) sees -- with the CPU oracle
(native timing build, one thread per problem up to the host's cores) beside it on the same problems.
yaml defaults (population 16, elites 4).  usage: tools/latency.py [robot] [exact|fast]   (exact = the default arithmetic)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import pick_ik_amd as pk  # noqa: E402
from oracle import oracle as O  # noqa: E402  (CPU baseline leg of a measurement tool)

name = sys.argv[1] if len(sys.argv) > 1 else "panda"
ch = pk.robots.by_name(name)
flavour = sys.argv[2] if len(sys.argv) > 2 else "exact"
s = pk.Solver(ch, exact=(flavour == "exact"))
print(f"# {name}, arithmetic = {flavour}")
try:
    o = O.Oracle(ch, timing_build=True)
except Exception:
    o = O.Oracle(ch)
rng = np.random.default_rng(0)
home = {"panda": pk.robots.PANDA_HOME, "ur5": pk.robots.UR5_HOME}.get(name, np.zeros(ch.dof))
for P in (16, 128):
    p = pk.default_params(memetic_population_size=P)
    for B in (1, 16, 256, 4096):
        q = rng.uniform(ch.qmin, ch.qmax, size=(B, ch.dof))
        goal = s.fk(q)
        seed = np.tile(home, (B, 1))
        s.solve_batch(p, goal, seed, rng_seed=1)  # warm-up (allocations, constants)
        ts, ok = [], []
        for r in range(20 if B <= 256 else 5):
            t0 = time.perf_counter()
            _, st, _, stats = s.solve_batch(p, goal, seed, rng_seed=2 + r)
            ts.append(time.perf_counter() - t0)
            ok.append((st == pk.SUCCESS).mean())
        ts = np.array(ts) * 1e3
        po = O.default_params(memetic_population_size=P)
        tc, ook = [], []
        for r in range(10 if B <= 256 else 3):
            t0 = time.perf_counter()
            _, ost, _, _ = o.solve_batch(po, goal, seed, rng_seed=2 + r, num_threads=min(B, O.max_threads()))
            tc.append(time.perf_counter() - t0)
            ook.append((ost == 1).mean())
        tc = np.array(tc) * 1e3
        print(f"{name} P={P:4d} B={B:5d}: GPU median {np.median(ts):8.2f} ms  min {ts.min():8.2f}  max {ts.max():8.2f}  "
              f"success {np.mean(ok):.3f} (mean of the repetitions)  mean generations {stats['generations'].mean():.1f} | CPU oracle "
              f"({min(B, O.max_threads())} threads) median {np.median(tc):8.2f} ms  min {tc.min():8.2f}  "
              f"success {np.mean(ook):.3f}")

# local mode (ik_gradient, the plugin's `mode: local`): seeds 0.05 rad (std) away from a solution
p = pk.default_params(mode=1)
po = O.default_params(mode=1)
for B in (1, 16, 256, 4096):
    q = rng.uniform(ch.qmin, ch.qmax, size=(B, ch.dof))
    goal = s.fk(q)
    seed = np.clip(q + rng.normal(0.0, 0.05, size=q.shape), ch.qmin, ch.qmax)
    s.solve_batch(p, goal, seed)
    ts, ok = [], []
    for r in range(20):
        t0 = time.perf_counter()
        _, st, _, stats = s.solve_batch(p, goal, seed)
        ts.append(time.perf_counter() - t0)
        ok.append((st == pk.SUCCESS).mean())
    ts = np.array(ts) * 1e3
    tc, ook = [], []
    for r in range(5):
        t0 = time.perf_counter()
        _, ost, _, _ = o.solve_batch(po, goal, seed, num_threads=min(B, O.max_threads()))
        tc.append(time.perf_counter() - t0)
        ook.append((ost == 1).mean())
    tc = np.array(tc) * 1e3
    print(f"{name} local mode B={B:5d}: GPU median {np.median(ts):8.3f} ms  min {ts.min():8.3f}  success {np.mean(ok):.3f}  "
          f"mean steps {stats['generations'].mean():.1f} | CPU oracle ({min(B, O.max_threads())} threads) median "
          f"{np.median(tc):8.3f} ms  min {tc.min():8.3f}  success {np.mean(ook):.3f}")

# What a host callback INSIDE the search would cost: one host round trip = joint vectors
# device -> host, the callback, a value host -> device, the next kernel.  pikamd_cost_batch of 4 candidates is
# that round trip without any callback work (H2D copies, one tiny kernel, D2H copies, a synchronise):
p = pk.default_params()
q4 = rng.uniform(ch.qmin, ch.qmax, size=(4, ch.dof))
g4 = s.fk(q4)
s.cost(p, g4, q4, q4)
ts = []
for r in range(200):
    t0 = time.perf_counter()
    s.cost(p, g4, q4, q4)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
_, _, _, st = s.solve_batch(pk.default_params(memetic_population_size=128), g4, np.tile(home, (4, 1)), rng_seed=1)
ev = st["cost_evals"].mean()
print(f"{name} host round trip of 4 joint vectors (pikamd_cost_batch, no callback work): median {np.median(ts):.1f} us  min "
      f"{ts.min():.1f} us; a solve at P=128 makes {ev:.0f} cost evaluations: one round trip per evaluation = "
      f"{ev * np.median(ts) * 1e-3:.0f} ms per solve, per gradient-descent iteration (25 x generations) = "
      f"{25 * st['generations'].mean() * np.median(ts) * 1e-3:.1f} ms")
