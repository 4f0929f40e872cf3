#!/usr/bin/env python3
"""Gathers the per-shape profile summaries (tools/profile_driver_cmd.sh -> gpurun_out/prof_<tag>/summary.json) into
profiles/roofline_inputs.json, the file bench.py reads, and copies summary + kernel statistics into profiles/.
usage: tools/make_roofline_inputs.py <round tag, e.g. r03> shape=tag [shape=tag ...]
  shapes: driver_cmd default_run single_batch config3 config4 config5"""
import json
import os
import shutil
import sys
import glob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
out = {"note": "per benchmarked shape: executed work per solved problem from rocprofv3 PMC passes of bench.py's own command "
               "(tools/profile_driver_cmd.sh -> tools/read_prof.py); the kernels a run executes depend on its shape "
               "(latency- or throughput-greedy variants), so each shape has its own record"}
for a in sys.argv[2:]:
    shape, tag = a.split("=")
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    rec = json.load(open(os.path.join(src, "summary.json")))
    out[shape] = rec
    base = os.path.join(ROOT, "profiles", f"{rnd}_{shape}")
    shutil.copy(os.path.join(src, "summary.json"), base + "_summary.json")
    shutil.copy(os.path.join(src, "summary.txt"), base + "_summary.txt")
    for f in glob.glob(src + "/kt/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, base + "_kernel_stats.csv")
    for name in ("bench_line.json", "bench_line_full.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 2:
            shutil.copy(p, base + "_" + name)
    print(shape, "flop/problem", rec["executed_fp64_flop_per_problem"], "valu/problem", rec["valu_wave_instructions_per_problem"],
          "hbm B/problem", rec["hbm_bytes_per_problem"])
# the kernel sources the counters belong to (bench.py compares it with the sources it runs: roofline.inputs_stale)
sys.path.insert(0, ROOT)
import bench  # noqa: E402
out["csrc_sha"] = bench.csrc_sha()
json.dump(out, open(os.path.join(ROOT, "profiles", "roofline_inputs.json"), "w"), indent=1)
print("wrote profiles/roofline_inputs.json")
