#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace csv and prints how many solver kernels were in flight over time,
their duration by pass (position within the batch) and the implied wave-slot demand.
usage: tools/trace_concurrency.py <kt_kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "memetic" in r["Kernel_Name"]]
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
cur, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[cur] += t - last
    last = t
    cur += d
tot = sum(hist.values())
mean = sum(k * v for k, v in hist.items()) / tot
print(f"{len(rows)} kernels over {(t1 - t0) / 1e6:.1f} ms; mean kernels in flight {mean:.1f}; max {max(hist)}")
cum = 0
for k in sorted(hist):
    cum += hist[k]
    if k % 8 == 0 or k == max(hist):
        print(f"   <= {k:3d} in flight: {100.0 * cum / tot:5.1f}% of the time")
# per stream, kernels in order -> pass index
by_stream = defaultdict(list)
for r in rows:
    by_stream[(r["Queue_Id"], r.get("Stream_Id", ""))].append(r)
dur = defaultdict(list)
for q, rs in by_stream.items():
    rs.sort(key=lambda r: int(r["Start_Timestamp"]))
    k = 0
    for r in rs:
        lpe4 = "<7, 4>" in r["Kernel_Name"] or ", 4>" in r["Kernel_Name"]
        dur[(k, lpe4)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        k += 1
print(f"{len(by_stream)} queues/streams")
alld = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows)
print(f"kernel duration ms: median {alld[len(alld) // 2]:.2f} mean {sum(alld) / len(alld):.2f} max {alld[-1]:.2f}; "
      f"sum of durations / wall = {sum(alld) / ((t1 - t0) / 1e6):.1f}")
