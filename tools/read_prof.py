#!/usr/bin/env python3
"""Summarises the rocprofv3 outputs of tools/profile_driver_cmd.sh: per-kernel duration stats (CSV
or rocpd .db), PMC counter sums over the solver kernels, and -- with --json -- the record bench.py
reads for `roofline` (executed FP64 flop and HBM bytes per solved problem).

usage: tools/read_prof.py <prof dir> [--json out.json] [--problems N]
  N = problems the profiled process solved (default: read from the bench line: (steps + warmup) x batch)

Definitions (so that the numbers can be recomputed from the committed summary):
  executed FP64 flop = (SQ_INSTS_VALU_ADD_F64 + _MUL_F64 + _TRANS_F64 + 2 x _FMA_F64) x 64 lanes
      -- wave-level instruction counts x the full wave width, i.e. an UPPER bound on useful flops
      (lanes masked off by divergence are counted as if they worked).
  HBM bytes          = 2 x FETCH_SIZE + WRITE_SIZE, both reported in KiB by rocprofv3; the factor 2
      is MI355X_MICROARCH.md's gfx950 correction for FETCH_SIZE (128-byte requests tallied at 64).
"""
import csv
import glob
import json
import os
import sqlite3
import sys

SOLVER_KERNELS = ("memetic_kernel", "ik_gradient_kernel")


def kernel_stats_csv(root):
    for f in glob.glob(root + "/kt/**/*kernel_stats.csv", recursive=True):
        with open(f) as fh:
            return list(csv.DictReader(fh))
    return []


def main(root, json_out=None, problems=None):
    ks = kernel_stats_csv(root)
    kernels = []
    if ks:
        print("== rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
        print(f"  {'kernel':72s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'min us':>10s} {'max us':>10s} {'%':>6s}")
        for r in ks[:8]:
            print(f"  {r['Name'][:72]:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.2f} "
                  f"{float(r['AverageNs'])/1e3:10.1f} {float(r['MinNs'])/1e3:10.1f} {float(r['MaxNs'])/1e3:10.1f} "
                  f"{float(r['Percentage']):6.2f}")
            if any(k in r["Name"] for k in SOLVER_KERNELS):
                kernels.append({"name": r["Name"].split("(")[0].replace("void ", ""), "calls": int(r["Calls"]),
                                "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                "avg_us": float(r["AverageNs"]) / 1e3})
    sums = {}
    for db in sorted(glob.glob(root + "/*/*.db") + glob.glob(root + "/*/*/*.db")):
        con = sqlite3.connect(db)
        print("==", os.path.relpath(db, root))
        try:
            cols = [c[1] for c in con.execute("pragma table_info(counters_collection)")]
            name_col = "counter_name" if "counter_name" in cols else "name"
            q = (f"select kernel_name, {name_col}, count(*), sum(value) from counters_collection "
                 f"group by kernel_name, {name_col}")
            for kname, cname, n, total in con.execute(q).fetchall():
                if any(k in (kname or "") for k in SOLVER_KERNELS):
                    print(f"  {kname[:48]:48s} {cname:28s} dispatches {n:5d} sum {total:.6g}")
                    sums[cname] = sums.get(cname, 0.0) + total
        except Exception as e:  # a kernel-trace-only database
            print("  (no counters:", e, ")")
    line = None
    bl = os.path.join(root, "bench_line.json")
    if os.path.exists(bl) and os.path.getsize(bl) > 2:
        line = json.load(open(bl))
        if problems is None:
            problems = (line["steps"] + line["warmup"]) * line["config"]["batch_per_gpu"]
    if json_out and problems:
        f64 = {k: sums.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS")}
        flop = (f64["ADD"] + f64["MUL"] + f64["TRANS"] + 2.0 * f64["FMA"]) * 64.0
        fetch, write = sums.get("FETCH_SIZE", 0.0) * 1024.0, sums.get("WRITE_SIZE", 0.0) * 1024.0
        rec = {
            "source": os.path.basename(root.rstrip("/")),
            "command": "python bench.py " + (open(os.path.join(root, "args.txt")).read().strip()
                                              if os.path.exists(os.path.join(root, "args.txt")) else ""),
            "problems_in_profiled_process": problems,
            "solver_kernels": kernels,
            "counters_summed_over_solver_kernels": sums,
            "fp64_wave_instructions": f64,
            "executed_fp64_flop_total": flop,
            "executed_fp64_flop_per_problem": flop / problems,
            "valu_wave_instructions_per_problem": sums.get("SQ_INSTS_VALU", 0.0) / problems,
            "fp64_share_of_valu_instructions": (sum(f64.values()) / sums["SQ_INSTS_VALU"]) if sums.get("SQ_INSTS_VALU") else None,
            # instruction classes per problem, for the issue roof (bench.py prices each class with the cycles
            # tools/valu_rates.hip measured for it: profiles/r03_valu_rates.json)
            "valu_classes_per_problem": {
                "fp64_arith": (f64["ADD"] + f64["MUL"] + f64["FMA"]) / problems,
                "fp64_trans": f64["TRANS"] / problems,
                "int64": sums.get("SQ_INSTS_VALU_INT64", 0.0) / problems,
                "int32": sums.get("SQ_INSTS_VALU_INT32", 0.0) / problems,
                "cvt": sums.get("SQ_INSTS_VALU_CVT", 0.0) / problems,
                "fp32": sum(sums.get("SQ_INSTS_VALU_" + k + "_F32", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS")) / problems,
                "all": sums.get("SQ_INSTS_VALU", 0.0) / problems,
            },
            # memory instructions of ANY kind (scratch spills and reloads, parked state, the problem's own I/O): the
            # price of the exact kernels' stack frames and of the two-per-SIMD kernels' spills, counted
            "memory_instructions_per_problem": (sums.get("SQ_INSTS_FLAT", 0.0) + sums.get("SQ_INSTS_VMEM_RD", 0.0) +
                                                sums.get("SQ_INSTS_VMEM_WR", 0.0)) / problems,
            "salu_instructions_per_problem": sums.get("SQ_INSTS_SALU", 0.0) / problems,
            "lds_instructions_per_problem": sums.get("SQ_INSTS_LDS", 0.0) / problems,
            "fetch_bytes_per_problem_raw": fetch / problems,
            "write_bytes_per_problem": write / problems,
            "hbm_bytes_per_problem": (2.0 * fetch + write) / problems,
            "bench_line_of_the_kernel_trace_run": line,
            "definitions": __doc__.split("Definitions")[1].strip(),
        }
        json.dump(rec, open(json_out, "w"), indent=1)
        print("wrote", json_out)
        for k in ("executed_fp64_flop_per_problem", "valu_wave_instructions_per_problem",
                  "fp64_share_of_valu_instructions", "memory_instructions_per_problem", "hbm_bytes_per_problem"):
            print(f"  {k}: {rec[k]}")


if __name__ == "__main__":
    a = sys.argv[1:]
    root = a[0] if a else "gpurun_out/prof_r02"
    jo = a[a.index("--json") + 1] if "--json" in a else None
    nb = int(a[a.index("--problems") + 1]) if "--problems" in a else None
    main(root, jo, nb)
