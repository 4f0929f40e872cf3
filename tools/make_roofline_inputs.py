#!/usr/bin/env python3
"""Gathers per-shape profile summaries (tools/profile_driver_cmd.sh -> gpurun_out/prof_<tag>/summary.json) into
profiles/roofline_inputs.json, the file bench.py reads, and copies summary + kernel statistics into profiles/.
usage: tools/make_roofline_inputs.py <round tag, e.g. r05> key=tag [key=tag ...]
  keys: <shape> for the fast flavour, <shape>_exact for the exact flavour;
        shapes: driver_cmd default_run single_batch config3 config4 config5
Records that are not named on the command line are KEPT (a re-profile of one flavour leaves the other's inputs alone).
Every record stores `flavour_sha`: the hash of the preprocessed device source of the kernel flavour that served the
profiled run (pick_ik_amd/build.py flavour_sha), which bench.py compares with the sources it runs (roofline.inputs_stale)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pick_ik_amd import build  # noqa: E402

rnd = sys.argv[1]
path = os.path.join(ROOT, "profiles", "roofline_inputs.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out.pop("csrc_sha", None)  # (rounds 3-4: one hash over all of csrc/; replaced by the per-record flavour_sha)
out["note"] = ("per benchmarked shape and flavour: executed work per solved problem from rocprofv3 PMC passes of bench.py's own "
               "command (tools/profile_driver_cmd.sh -> tools/read_prof.py); the kernels a run executes depend on its shape "
               "(latency- or throughput-greedy variants) and on the arithmetic option, so each has its own record; "
               "flavour_sha = hash of the flavour's preprocessed device source at profile time")
for a in sys.argv[2:]:
    key, tag = a.split("=")
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    rec = json.load(open(os.path.join(src, "summary.json")))
    names = [k["name"] for k in rec.get("solver_kernels", [])]
    ns = sorted({n.split("::")[0] for n in names})
    rec["kernel_namespaces"] = ns
    rec["flavour_sha"] = build.flavour_sha(ns[0]) if len(ns) == 1 and ns[0] in build.FLAVOUR_FLAGS else None
    out[key] = rec
    base = os.path.join(ROOT, "profiles", f"{rnd}_{key}")
    shutil.copy(os.path.join(src, "summary.json"), base + "_summary.json")
    shutil.copy(os.path.join(src, "summary.txt"), base + "_summary.txt")
    for f in glob.glob(src + "/kt/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, base + "_kernel_stats.csv")
    for name in ("bench_line.json", "bench_line_full.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 2:
            shutil.copy(p, base + "_" + name)
    print(key, ns, "flop/problem", rec["executed_fp64_flop_per_problem"], "valu/problem", rec["valu_wave_instructions_per_problem"],
          "hbm B/problem", rec["hbm_bytes_per_problem"], "sha", rec["flavour_sha"])
# records of earlier rounds without a hash: the flavour they were taken with, if its source is still what it was
for key, rec in out.items():
    if isinstance(rec, dict) and "solver_kernels" in rec and "flavour_sha" not in rec:
        ns = sorted({k["name"].split("::")[0] for k in rec["solver_kernels"]})
        rec["kernel_namespaces"] = ns
        rec["flavour_sha"] = None
json.dump(out, open(path, "w"), indent=1)
print("wrote profiles/roofline_inputs.json")
