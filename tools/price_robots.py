#!/usr/bin/env python3
"""Roofline of one profiled bench.py run of a non-default robot (tools/gpu/price_robots.sh): the PMC summary of
tools/read_prof.py + the bench line of the kernel-trace run -> one JSON object on stdout.
usage: tools/price_robots.py gpurun_out/prof_<tag>"""
import json
import os
import sys

d = sys.argv[1]
s = json.load(open(os.path.join(d, "summary.json")))
b = json.load(open(os.path.join(d, "bench_line.json")))
PEAK = 78.6e12
sr = b["config"].get("success_rate") or 1.0
pps = b["value"] / max(sr, 1e-9)  # problems per second (solved or not)
fl = s["executed_fp64_flop_per_problem"]
out = {
    "robot": b["config"].get("robot"), "arithmetic": b["config"].get("arithmetic"), "dof": b["config"].get("dof"),
    "command": s["command"], "value": b["value"], "unit": b["unit"], "ms_per_step": b["ms_per_step"],
    "success_rate": sr, "tip_frames": b["config"].get("tip_frames"), "problems_per_s": pps,
    "identical_to_oracle_on_sample": (b.get("parity") or {}).get("identical_to_oracle_on_sample"),
    "kernel": (b.get("roofline") or {}).get("kernel"),
    "roofline": {"bound": "fp64_valu", "peak": 78.6, "unit": "TFLOP/s", "achieved": fl * pps / 1e12,
                 "frac": fl * pps / PEAK, "executed_fp64_flop_per_problem": fl,
                 "valu_wave_instructions_per_problem": s["valu_wave_instructions_per_problem"],
                 "fp64_share_of_valu_instructions": s["fp64_share_of_valu_instructions"],
                 "salu_instructions_per_problem": s.get("salu_instructions_per_problem"),
                 "memory_instructions_per_problem": s.get("memory_instructions_per_problem"),
                 "hbm_bytes_per_problem": s.get("hbm_bytes_per_problem"),
                 "note": "counters are per PROBLEM of the profiled process (solved or not); achieved = flop per problem x problems per second"},
    "source": s["source"],
}
print(json.dumps(out))
