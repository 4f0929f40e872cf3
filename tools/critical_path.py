#!/usr/bin/env python3
"""One wavefront's generation from its instruction counts: the output of tools/gpu/critical_path.sh (the long
runners of config 2 alone on the chip, memetic_kernel<7, LPE, false, 1> under rocprofv3) against the issue rates
of a LONE wavefront measured by tools/valu_rates.hip (profiles/r03_valu_rates.json).

A lone wavefront issues one instruction of any kind per 4.3-5.3 cycles and a dependent FP64 chain runs as fast as
an independent one (section 5 of DESIGN.md), so the time of a generation is
    sum over classes (instructions of the class x cycles per instruction of a lone wavefront) + exposed waits
and the first term is a FLOOR for this instruction stream.  The counts are wave-level PMC sums of the measured
kernel divided by (wavefronts x calls x 100 generations); the measured time per generation is the kernel's average
duration / 100.  usage: tools/critical_path.py <dir> <fast|exact> <lanes>..."""
import csv
import glob
import json
import os
import re
import sqlite3
import sys

root, flavour = sys.argv[1], sys.argv[2]
lanes = [int(x) for x in sys.argv[3:]]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rates = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_rates.json")))


def lone(instr, form="thr"):
    for r in rates["rows"]:
        if r["instr"] == instr and r["form"] == form and r["waves_per_simd"] == 1:
            return r["cycles_per_instr_per_wave"]
    raise KeyError(instr)


CYC = {"fp64_arith": lone("v_fma_f64"), "fp64_trans": lone("v_rsq_f64", "dep"), "int32": lone("v_mov_b32", "dep"),
       "int64": lone("v_fma_f64"), "cvt": lone("v_rndne_f64"), "other_valu": None, "salu": 4.25, "smem": 4.25, "lds": 8.6}
OTHER_LO, OTHER_HI = lone("v_mov_b32", "dep"), lone("v_fma_f64")  # moves / selects ... lane reads, 64-bit moves


def counters(d, kernel_pat):
    sums, n_disp = {}, {}
    for db in glob.glob(d + "/*/*.db") + glob.glob(d + "/*/*/*.db"):
        con = sqlite3.connect(db)
        try:
            cols = [c[1] for c in con.execute("pragma table_info(counters_collection)")]
            name_col = "counter_name" if "counter_name" in cols else "name"
            for kname, cname, n, total in con.execute(
                    f"select kernel_name, {name_col}, count(*), sum(value) from counters_collection group by kernel_name, {name_col}"):
                if kname and re.search(kernel_pat, kname):
                    sums[cname] = sums.get(cname, 0.0) + total
                    n_disp[cname] = n
        except Exception:
            pass
    return sums, n_disp


for L in lanes:
    d = os.path.join(root, f"{flavour}_lpe{L}")
    res = {}
    for line in open(os.path.join(d, "plain.log")):
        if line.startswith("RESULT"):
            res = dict(kv.split("=") for kv in line.split()[1:])
        elif line.strip():
            print(line.rstrip())
    if not res:
        print(f"lanes {L}: no result"); continue
    waves, calls, launches = int(res["waves"]), int(res["calls"]), int(res["launches"])
    pat = rf"memetic_kernel<7, {L}, false, 1>"
    avg_us = None
    for f in glob.glob(d + "/kt/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if re.search(pat, r["Name"]):
                # (the search for the long runners may use the same template at another width: not this row)
                avg_us, n_calls = float(r["AverageNs"]) / 1e3, int(r["Calls"])
    s, _ = counters(d, pat)
    if avg_us is None or not s:
        print(f"lanes {L}: profile incomplete ({avg_us}, {len(s)} counters)"); continue
    per = waves * calls * 100.0  # wavefront-generations in the profiled calls
    f64 = sum(s.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) for k in ("ADD", "MUL", "FMA"))
    cls = {"fp64_arith": f64, "fp64_trans": s.get("SQ_INSTS_VALU_TRANS_F64", 0.0),
           "int32": s.get("SQ_INSTS_VALU_INT32", 0.0), "int64": s.get("SQ_INSTS_VALU_INT64", 0.0),
           "cvt": s.get("SQ_INSTS_VALU_CVT", 0.0)}
    cls["other_valu"] = s.get("SQ_INSTS_VALU", 0.0) - sum(cls.values())
    cls.update(salu=s.get("SQ_INSTS_SALU", 0.0), smem=s.get("SQ_INSTS_SMEM", 0.0), lds=s.get("SQ_INSTS_LDS", 0.0))
    lo = hi = 0.0
    print(f"\n== {flavour} flavour, {L} lanes per elite: memetic_kernel<7, {L}, false, 1>, {waves} wavefronts x {calls} repetitions x 100 generations "
          f"in {launches * calls} launches of {waves // launches} wavefront(s)")
    print(f"   {'class':14s} {'instr / wavefront-generation':>30s} {'cycles each (lone wavefront)':>30s} {'cycles':>12s}")
    for k, v in cls.items():
        n = v / per
        if k == "other_valu":
            lo += n * OTHER_LO; hi += n * OTHER_HI
            print(f"   {k:14s} {n:30.0f} {f'{OTHER_LO:.2f} .. {OTHER_HI:.2f}':>30s} {n * OTHER_LO:8.0f}..{n * OTHER_HI:.0f}")
        else:
            lo += n * CYC[k]; hi += n * CYC[k]
            print(f"   {k:14s} {n:30.0f} {CYC[k]:30.2f} {n * CYC[k]:12.0f}")
    total_instr = sum(cls.values()) / per
    clock_mhz = 2400.0
    wave_cycles = s.get("SQ_WAVE_CYCLES", 0.0) / per
    wait_inst = s.get("SQ_WAIT_INST_ANY", 0.0) / per
    active = s.get("SQ_ACTIVE_INST_ANY", 0.0) / per
    gen_us = avg_us / 100.0
    print(f"   instructions per wavefront-generation: {total_instr:.0f}; issue cycles by the table: {lo:.0f} .. {hi:.0f}")
    print(f"   measured: kernel {avg_us:.1f} us per call = {gen_us:.3f} us per generation = {gen_us * clock_mhz:.0f} cycles at {clock_mhz:.0f} MHz "
          f"(SQ_WAVE_CYCLES {wave_cycles:.0f}, SQ_ACTIVE_INST_ANY {active:.0f}, SQ_WAIT_INST_ANY {wait_inst:.0f} per wavefront-generation, counter units)")
    mid = 0.5 * (lo + hi)
    print(f"   issue floor / measured = {lo / (gen_us * clock_mhz):.2f} .. {hi / (gen_us * clock_mhz):.2f}"
          f"  (the rest: exposed waits on LDS round trips, scalar loads and s_waitcnt; a clock below {clock_mhz:.0f} MHz under FP64 load)")
    print(f"   => a generation at {L} lanes cannot take less than {lo / clock_mhz:.1f} us with this instruction stream; "
          f"one config-2 call of 4096 targets waits for ~{100 * gen_us / 1e3:.2f} ms of such generations")
