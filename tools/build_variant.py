#!/usr/bin/env python3
"""An experiment build of the product library for interleaved A/B runs on one GPU box (tools/gpu/ab_exact.sh):
the objects named on the command line are recompiled from the working tree with extra flags, the rest are taken from
the standard object cache (pick_ik_amd/_build), and the result is linked to pick_ik_amd/_variants/lib_<name>.so
(git-ignored; it travels with the gpurun snapshot).  Select it with PIK_LIB=<path>.

usage: tools/build_variant.py <name> [--objs exact:7,fast:amd,...] [extra hipcc flags...]
  exact:7  = the exact flavour's kernels for 7 variables     common:7 = the common-configuration flavour's
  fast:amd = the C ABI translation unit (host code + launch tables)
"""
import concurrent.futures as cf
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pick_ik_amd import build as B  # noqa: E402

name = sys.argv[1]
args = sys.argv[2:]
objs_spec = "exact:7,fast:amd"
if args and args[0] == "--objs":
    objs_spec = args[1]
    args = args[2:]
extra = args
want = set(objs_spec.split(","))
all_objs = B._objects(False) + B._exact_objects() + B._common_objects() + B._common_objects(True)
vdir = os.path.join(B.BUILD_DIR, "var_" + name)
os.makedirs(vdir, exist_ok=True)
os.makedirs(os.path.join(os.path.dirname(B.LIB), "_variants"), exist_ok=True)


def key_of(o):
    flavour = os.path.basename(os.path.dirname(o[0]))
    base = os.path.basename(o[0])
    if base == "pik_amd.o":
        return flavour + ":amd"
    if base.startswith("pik_inst_d"):
        return flavour + ":" + base[len("pik_inst_d"):-2]
    return flavour + ":" + base[:-2]


jobs, link = [], []
for o in all_objs:
    if key_of(o) in want:
        obj = os.path.join(vdir, key_of(o).replace(":", "_") + ".o")
        cmd = B._cmd(obj, o[1], o[2], False) + extra
        jobs.append((obj, cmd))
        link.append(obj)
    else:
        if not os.path.exists(o[0]):
            raise SystemExit(f"missing cached object {o[0]}: run python -m pick_ik_amd.build first")
        link.append(o[0])
missing = want - {key_of(o) for o in all_objs}
if missing:
    raise SystemExit(f"unknown objects {missing}")


def run(job):
    obj, cmd = job
    r = subprocess.run(cmd, cwd=B.CSRC, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(" ".join(cmd) + "\n" + r.stderr[-4000:])
    open(obj + ".res", "w").write(r.stderr)


with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as ex:
    list(ex.map(run, jobs))
lib = os.path.join(os.path.dirname(B.LIB), "_variants", f"lib_{name}.so")
subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *link], check=True)
print(lib)
