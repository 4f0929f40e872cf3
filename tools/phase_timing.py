#!/usr/bin/env python3
"""Where a generation's cycles go, phase by phase (s_memtime instrumentation compiled in with
-DPIK_PHASE_TIMING into a SEPARATE library: build it with
  python tools/phase_timing.py --build
then run this on the GPU).  Same workload as tools/ablate.py: unreachable targets, 8 generations."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pick_ik_amd", "libpick_ik_amd_phases.so")
if "--build" in sys.argv:
    # chain length 7 only: the kernels of the common configuration (what the Panda default call runs) with the
    # counters, the general ones without (fk and the selection run need them), everything else stubs
    src = os.path.join(ROOT, "pick_ik_amd", "csrc")
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include")]
    jobs, objs = [], []
    for flavour, fl in (("fast", ["-ffp-contract=on"]), ("common", ["-ffp-contract=on", "-DPIK_COMMON=1"]),
                        ("strict", ["-DPIK_STRICT", "-ffp-contract=off"])):
        for n in range(1, 17):
            o = f"/tmp/phases_{flavour}_d{n}.o"
            real_obj = n == 7 and flavour != "strict"
            flags = (["-DPIK_PHASE_TIMING=1"] if flavour == "common" else []) if real_obj else ["-DPIK_INST_STUB=1"]
            jobs.append(subprocess.Popen([*base, *fl, "-c", f"-DPIK_INST_D={n}", *flags, "-o", o, os.path.join(src, "pik_inst.hip")]))
            objs.append(o)
    jobs.append(subprocess.Popen([*base, "-ffp-contract=on", "-c", "-o", "/tmp/phases_abi.o", os.path.join(src, "pik_amd.hip")]))
    assert all(j.wait() == 0 for j in jobs)
    subprocess.run([*base, "-shared", "-o", LIB, "/tmp/phases_abi.o", *objs], check=True)
    print("built", LIB)
    sys.exit(0)

# --lib=<path>: another instrumented library, e.g. the exact flavour's kernels with the counters:
#   python tools/build_variant.py phases_exact --objs exact:7 -DPIK_PHASE_TIMING=1
#   python tools/phase_timing.py --exact --lib=pick_ik_amd/_variants/lib_phases_exact.so [--real] 1 4 16
for a in sys.argv[1:]:
    if a.startswith("--lib="):
        LIB = os.path.abspath(a[len("--lib="):])
EXACT = "--exact" in sys.argv
os.environ["PIK_LIB"] = LIB
os.environ["PIK_PASSES"] = "none"
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pick_ik_amd as pk  # noqa: E402

NAMES = ["gradient descent", "publish + worst", "round head", "child genes (RNG + mixing)", "child evaluation",
         "accept / erase", "insert into kept set", "after the loop", "sort / rank / extinctions",
         "termination / resolve"]
ROBOT = next((a[len("--robot="):] for a in sys.argv[1:] if a.startswith("--robot=")), "panda")  # e.g. --robot=ur5
POP = int(next((a[len("--population="):] for a in sys.argv[1:] if a.startswith("--population=")), "128"))
# --goals: BASELINE config 3's joint goals (centre + minimal displacement, cost threshold 0.01)
GOALS = dict(center_joints_weight=0.01, minimal_displacement_weight=0.001, cost_threshold=0.01) if "--goals" in sys.argv else {}
ch = pk.robots.by_name(ROBOT)
HOME = pk.robots.PANDA_HOME if ROBOT == "panda" else 0.5 * (ch.qmin + ch.qmax)
s = pk.Solver(ch, exact=EXACT)
L = C.CDLL(LIB)
rng = np.random.default_rng(0)
real = "--real" in sys.argv  # the long-runners of a real batch instead of unreachable targets
lpes = [a for a in sys.argv[1:] if not a.startswith("--")] or ["1", "4", "16"]
if real:
    # problems of BASELINE config 2 that run all 100 generations: the tail every pool waits for
    B0 = 32768
    g0 = s.fk(rng.uniform(ch.qmin, ch.qmax, size=(B0, ch.dof)))
    sd0 = np.tile(HOME, (B0, 1))
    p0 = pk.default_params(memetic_population_size=POP, **GOALS)
    _, st0, _, stats0 = s.solve_batch(p0, g0, sd0, rng_seed=1)
    long_run = np.flatnonzero(stats0["generations"] >= 100)
    print(f"{len(long_run)} of {B0} problems run all 100 generations; erasures per generation "
          f"{stats0['pool_erasures'][long_run].sum() / (100.0 * len(long_run)):.2f}, wipeouts per generation "
          f"{stats0['wipeouts'][long_run].sum() / (100.0 * len(long_run)):.2f}")
G = 100 if real else 8
s.set_option("passes", "none")
for lpe in lpes:
    s.set_option("lanes_per_elite", lpe)
    if real:
        goal, seed = g0[long_run], sd0[long_run]
        B = len(goal)
        p = pk.default_params(memetic_population_size=POP, **GOALS)
        waves = -(-B * int(lpe) // 16)
    else:
        B = 16 * 1024 // int(lpe)
        q = rng.uniform(ch.qmin, ch.qmax, size=(B, ch.dof))
        goal = s.fk(q)
        goal[:, :3] *= 3.0
        seed = np.tile(HOME, (B, 1))
        p = pk.default_params(memetic_population_size=POP, memetic_max_generations=G, memetic_wipeout_fitness_tol=-1e300, **GOALS)
        waves = 1024
    s.solve_batch(p, goal, seed, rng_seed=1)
    buf = (C.c_ulonglong * 16)()
    L.pik_debug_phase_cycles(buf)
    import time
    t0 = time.perf_counter()
    s.solve_batch(p, goal, seed, rng_seed=1)
    dt = time.perf_counter() - t0
    L.pik_debug_phase_cycles(buf)
    tot = sum(buf[:10])
    print(f"LPE {lpe}: {B} problems on {waves} wavefronts, {G} generations, call {dt * 1e3:.2f} ms = {dt * 1e3 / G:.4f} ms per "
          f"generation; {tot / waves / G:.0f} counter ticks per wavefront-generation, by phase:")
    for k, nme in enumerate(NAMES):
        print(f"   {nme:30s} {buf[k] / waves / G:12.0f} ticks per generation  {100.0 * buf[k] / tot:5.1f} %")
