#!/usr/bin/env python3
"""The problems of BASELINE config 2 that run all 100 generations -- the tail every pool waits for -- solved ALONE
with a fixed number of lanes per elite and no compaction passes: one wavefront per problem group, nothing else on
the chip.  A short program to put under rocprofv3 (tools/gpu/critical_path.sh): the kernel of the measured calls
is memetic_kernel<7, LPE, false, 1>, the search for the long runners runs under other variants.
usage: long_runners.py <fast|exact> <lanes per elite> [repetitions]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pick_ik_amd as pk  # noqa: E402

flavour, lpe = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ch = pk.robots.panda()
s = pk.Solver(ch, device=0, exact=(flavour == "exact"))
s.set_option("self_test", "off")
rng = np.random.default_rng(0)
B0 = 32768
goal0 = s.fk(rng.uniform(ch.qmin, ch.qmax, size=(B0, 7)))
seed0 = np.tile(pk.robots.PANDA_HOME, (B0, 1))
p = pk.default_params(memetic_population_size=128)
# the search: a width the measured calls do not use, so that the profile separates the two by kernel name
s.set_option("lanes_per_elite", "1" if lpe != 1 else "2")
s.set_option("passes", "none")
_, st0, _, stats0 = s.solve_batch(p, goal0, seed0, rng_seed=1)
long_run = np.flatnonzero(stats0["generations"] >= 100)
goal, seed = goal0[long_run], seed0[long_run]
n = len(goal)
s.set_option("lanes_per_elite", str(lpe))
s.set_option("two_per_simd", "0")
s.set_option("regime", "latency")
# every long runner as a batch of ONE problem with its own problem_offset -- the random streams are keyed by the
# problem's index, so this is what keeps it the same 100-generation run --, 64 / (4 lanes) of them per pool so that
# a pool is a whole number of wavefronts (pools of at most 64 records)
per_wave = 64 // (4 * lpe)
group = max(per_wave, 64 // per_wave * per_wave)
n = len(long_run) // group * group
long_run = long_run[:n]
ms, launches = [], 0
for _ in range(reps + 1):
    t0 = time.perf_counter()
    launches = 0
    for k in range(0, n, group):
        recs = [(goal0[i:i + 1], seed0[i:i + 1], None, int(i)) for i in long_run[k:k + group]]
        out = s.solve_batches(p, recs, rng_seed=1)
        launches += 1
        assert all(int(o[3]["generations"][0]) == 100 for o in out)
    ms.append((time.perf_counter() - t0) * 1e3)
waves = n // per_wave
print(f"{flavour} lanes {lpe}: {len(np.flatnonzero(stats0['generations'] >= 100))} of {B0} problems run all 100 generations; "
      f"{n} of them solved as pools of {group} one-problem batches ({group // per_wave} wavefront(s) per pool, {launches} pools per "
      f"repetition), {reps + 1} repetitions: {' '.join(f'{m:.2f}' for m in ms)} ms (host pointers, staging included)")
print(f"RESULT flavour={flavour} lpe={lpe} problems={n} waves={waves} calls={reps + 1} launches={launches} generations=100 "
      f"best_ms={min(ms):.3f}")
