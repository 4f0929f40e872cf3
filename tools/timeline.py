#!/usr/bin/env python3
"""Timeline of the solver kernels of ONE stream from a rocprofv3 kernel-trace csv: for every kernel
that did real work (longer than --min-us) its variant, start offset and duration -- the passes of a
pool in the order they ran -- plus the sum per variant.
usage: tools/timeline.py <kt_kernel_trace.csv> [--min-us 20] [--last N]"""
import csv
import re
import sys
from collections import defaultdict

a = sys.argv[1:]
min_us = float(a[a.index("--min-us") + 1]) if "--min-us" in a else 20.0
last = int(a[a.index("--last") + 1]) if "--last" in a else 0
rows = [r for r in csv.DictReader(open(a[0])) if "memetic_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if last:
    rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
tot = defaultdict(float)
skipped = 0
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"memetic_kernel<([^>]*)>", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    d = (e - s) / 1e3
    if d < min_us:
        skipped += 1
        continue
    tot[name] += d
    print(f"  +{(s - t0) / 1e6:8.3f} ms  gap {(s - prev_end) / 1e3:7.1f} us  <{name:18s}>  {d / 1e3:8.3f} ms  grid {r.get('Grid_Size_X', '?')}")
    prev_end = e
print(f"({skipped} kernels shorter than {min_us} us: variants that were not their pass's choice)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  <{k:18s}> {v / 1e3:8.3f} ms")
print(f"  first start -> last end: {(int(rows[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
