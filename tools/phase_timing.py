#!/usr/bin/env python3
"""Where a generation's cycles go, phase by phase (s_memtime instrumentation compiled in with
-DPIK_PHASE_TIMING into a SEPARATE library: build it with
  PIK_ONLY_D=7 PIK_EXTRA_HIPCC_FLAGS=-DPIK_PHASE_TIMING python tools/phase_timing.py --build
then run this on the GPU).  Same workload as tools/ablate.py: unreachable targets, 8 generations."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pick_ik_amd", "libpick_ik_amd_phases.so")
if "--build" in sys.argv:
    src = os.path.join(ROOT, "pick_ik_amd", "csrc")
    objs = []
    for n in range(1, 13):
        o = f"/tmp/phases_d{n}.o"
        flags = ["-DPIK_PHASE_TIMING=1"] if n == 7 else ["-DPIK_INST_STUB=1"]
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f"-DPIK_INST_D={n}",
                        *flags, "-o", o, os.path.join(src, "pik_inst.hip")], check=True)
        objs.append(o)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", "/tmp/phases_abi.o",
                    os.path.join(src, "pik_amd.hip")], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, "/tmp/phases_abi.o", *objs], check=True)
    print("built", LIB)
    sys.exit(0)

os.environ["PIK_LIB"] = LIB
os.environ["PIK_PASSES"] = "none"
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pick_ik_amd as pk  # noqa: E402

NAMES = ["gradient descent", "publish + worst", "round head", "child genes (RNG + mixing)", "child evaluation",
         "accept / erase", "insert into kept set", "after the loop", "sort / rank / extinctions",
         "termination / resolve"]
ch = pk.robots.panda()
s = pk.Solver(ch)
L = C.CDLL(LIB)
rng = np.random.default_rng(0)
G = 8
for lpe in (sys.argv[1:] or ["1", "4", "16"]):
    os.environ["PIK_LPE"] = lpe
    B = 16 * 1024 // int(lpe)
    q = rng.uniform(ch.qmin, ch.qmax, size=(B, 7))
    goal = s.fk(q)
    goal[:, :3] *= 3.0
    seed = np.tile(pk.robots.PANDA_HOME, (B, 1))
    p = pk.default_params(memetic_population_size=128, memetic_max_generations=G, memetic_wipeout_fitness_tol=-1e300)
    s.solve_batch(p, goal, seed, rng_seed=1)
    buf = (C.c_ulonglong * 16)()
    L.pik_debug_phase_cycles(buf)
    s.solve_batch(p, goal, seed, rng_seed=2)
    L.pik_debug_phase_cycles(buf)
    waves = 1024
    tot = sum(buf[:10])
    print(f"LPE {lpe}: {tot / waves / G / 100e6 * 1e3:.3f} ms per generation at 100 MHz counter (cycle counter units), by phase:")
    for k, nme in enumerate(NAMES):
        print(f"   {nme:30s} {buf[k] / waves / G:12.0f} ticks per generation  {100.0 * buf[k] / tot:5.1f} %")
