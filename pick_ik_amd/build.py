"""Builds pick_ik_amd/libpick_ik_amd.so from csrc/ with hipcc for gfx950 (in-tree, so the .so
travels with the repository snapshot to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libpick_ik_amd.so")
# Verification build of the SAME sources: no FMA contraction, generic joint rotations -- IEEE
# arithmetic in the reference's operation order, bit-comparable with the CPU oracle's portable
# math mode (tests/test_gpu_strict_parity.py).  ~2x slower; never used for measurements.
LIB_STRICT = os.path.join(_HERE, "libpick_ik_amd_strict.so")
SOURCES = ["pik_amd.hip", "pik_kernels.hpp", "pik_math.hpp", "pik_host.hpp"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pick_ik_amd.h")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(lib: str, extra, verbose: bool):
    extra = list(extra) + os.environ.get("PIK_EXTRA_HIPCC_FLAGS", "").split()  # experiments only
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *extra,
           "-o", lib + ".tmp", os.path.join(CSRC, "pik_amd.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(lib + ".tmp", lib)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Builds the product library and the strict-arithmetic verification library."""
    procs = []
    import concurrent.futures as cf
    jobs = []
    if force or is_stale(LIB):
        jobs.append((LIB, []))
    if force or is_stale(LIB_STRICT):
        jobs.append((LIB_STRICT, ["-DPIK_STRICT=1", "-ffp-contract=off"]))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for f in [ex.submit(_compile, lib, extra, verbose) for lib, extra in jobs]:
                f.result()
    del procs
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
