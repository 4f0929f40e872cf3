#!/usr/bin/env python3
"""Parses hipcc's `-Rpass-analysis=kernel-resource-usage` remarks (stderr of a compile) into rows
(kernel, VGPRs, AGPRs, SGPRs, scalar / vector spills, scratch bytes per lane, occupancy, LDS bytes).

usage: python -m pick_ik_amd.kernel_resources <remarks.txt> [flavour]      -> csv rows on stdout
Used by pick_ik_amd/build.py (every object is compiled with the remarks on; the ledger of every shipped kernel is
pick_ik_amd/_build/kernel_resources.csv, its committed copy profiles/r04_kernel_resources.csv) and by
tests/test_kernel_resources_cpu.py (a kernel whose scratch / spills grow past the committed ledger fails)."""
from __future__ import annotations

import re
import subprocess
import sys

FIELDS = ("vgprs", "agprs", "sgprs", "sgpr_spills", "vgpr_spills", "scratch_bytes_per_lane", "occupancy", "lds_bytes")
_KEYS = {"vgprs": r"VGPRs", "agprs": r"AGPRs", "sgprs": r"TotalSGPRs", "sgpr_spills": r"SGPRs Spill",
         "vgpr_spills": r"VGPRs Spill", "scratch_bytes_per_lane": r"ScratchSize \[bytes/lane\]",
         "occupancy": r"Occupancy \[waves/SIMD\]", "lds_bytes": r"LDS Size \[bytes/block\]"}


def demangle(names):
    if not names:
        return []
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    # template arguments are what tells the variants apart; the parameter list is noise
    short = []
    for d in out:
        d = re.sub(r"^void ", "", d)
        depth, cut = 0, len(d)
        for i, ch in enumerate(d):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        short.append(d[:cut].replace(" ", ""))
    return short


def parse(text: str):
    """[(mangled, {field: int})] in the order the compiler reported them"""
    rows = []
    for b in re.split(r"remark: Function Name: ", text)[1:]:
        name = b.split(" [-Rpass")[0].strip()
        vals = {}
        for f, k in _KEYS.items():
            m = re.search(r"remark:\s+" + k + r": (\d+)", b)
            vals[f] = int(m.group(1)) if m else -1
        rows.append((name, vals))
    return rows


def rows_of(text: str, flavour: str = ""):
    parsed = parse(text)
    names = demangle([n for n, _ in parsed])
    return [(flavour, nm, v) for nm, (_, v) in zip(names, parsed)]


if __name__ == "__main__":
    fl = sys.argv[2] if len(sys.argv) > 2 else ""
    print("flavour,kernel," + ",".join(FIELDS))
    for f, n, v in rows_of(open(sys.argv[1]).read(), fl):
        print(f'{f},"{n}",' + ",".join(str(v[k]) for k in FIELDS))
