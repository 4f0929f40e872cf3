#!/usr/bin/env python3
"""The tables of DESIGN.md sections 5 and 6 from the committed inputs: profiles/roofline_inputs.json (per-problem
counters per shape and flavour) and the bench lines under profiles/<round>_bench_*.json.  usage: tools/design_tables.py [round tag, default r06]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rin = json.load(open(os.path.join(ROOT, "profiles", "roofline_inputs.json")))
SHAPES = ["driver_cmd", "default_run", "single_batch", "config3", "config4", "config5"]
TITLE = {"driver_cmd": "driver's command", "default_run": "default run", "single_batch": "one batch at a time",
         "config3": "config 3", "config4": "config 4", "config5": "config 5"}


def k(x):
    return f"{x / 1e3:.1f} k"


for fl, sfx in (("exact", "_exact"), ("fast", "")):
    cols = [s for s in SHAPES if s + sfx in rin]
    print(f"\n| per solved problem ({fl} flavour) | " + " | ".join(TITLE[s] for s in cols) + " |")
    print("|---|" + "---|" * len(cols))
    R = [rin[s + sfx] for s in cols]
    rows = [
        ("vector instructions (`SQ_INSTS_VALU`, wave level)", lambda r: k(r["valu_wave_instructions_per_problem"])),
        ("... FP64 arithmetic (ADD + MUL + FMA)", lambda r: k(r["valu_classes_per_problem"]["fp64_arith"])),
        ("... FP64 transcendental (`v_rsq` / `v_rcp`)", lambda r: k(r["valu_classes_per_problem"]["fp64_trans"])),
        ("... INT64 / INT32 / convert", lambda r: " / ".join(f"{r['valu_classes_per_problem'][c] / 1e3:.1f}" for c in ("int64", "int32", "cvt")) + " k"),
        ("FP64 share of the vector instructions", lambda r: f"{100 * r['fp64_share_of_valu_instructions']:.1f} %"),
        ("scalar / LDS instructions", lambda r: f"{k(r['salu_instructions_per_problem'])} / {k(r['lds_instructions_per_problem'])}"),
        ("memory instructions of any kind (FLAT + VMEM)", lambda r: k(r["memory_instructions_per_problem"]) if r.get("memory_instructions_per_problem") else "--"),
        ("executed FP64 flop ((ADD + MUL + TRANS + 2 FMA) x 64 lanes)", lambda r: f"{r['executed_fp64_flop_per_problem'] / 1e6:.2f} M"),
        ("HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE)", lambda r: f"{r['hbm_bytes_per_problem'] / 1e3:.1f} KB"),
        ("profile", lambda r: "`" + r["source"] + "`"),
    ]
    for name, f in rows:
        print(f"| {name} | " + " | ".join(f(r) for r in R) + " |")

print()
RND = sys.argv[1] if len(sys.argv) > 1 else "r06"
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", RND + "_bench_*.json"))):
    d = json.load(open(f))
    r = d.get("roofline") or {}
    fast = d.get("fast") or {}
    fr = fast.get("roofline") or {}
    vi = (r.get("valu_issue") or {}).get("calibrated") or {}
    print(f"{os.path.basename(f)}: {d['config'].get('arithmetic')} {d['value'] / 1e6:.3f} M {d['unit']} {d['ms_per_step']:.3f} ms/step "
          f"frac {r.get('frac')} issue {vi.get('frac_low')}..{vi.get('frac_high')} stale {r.get('inputs_stale')} "
          f"identical {(d.get('parity') or {}).get('identical_to_oracle_on_sample')} | fast {fast.get('value', 0) / 1e6:.3f} M frac {fr.get('frac')}")
    for leg in ("sustained", "single_batch", "config3", "config4"):
        for where, dd in (("exact", d), ("fast", fast)):
            if leg in dd:
                x = dd[leg]
                print(f"    {where}.{leg}: value {x.get('value')} {x.get('unit')} ms {x.get('ms_per_step', x.get('median_ms'))} frac {(x.get('roofline') or {}).get('frac')}")
    if "cpu_baseline" in d:
        print("    cpu_baseline:", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cores"))
