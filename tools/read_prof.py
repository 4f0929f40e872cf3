#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd (.db) outputs: per-kernel duration stats and PMC counter sums.
usage: tools/read_prof.py <dir with */*.db>"""
import glob
import sqlite3
import sys


def main(root):
    for db in sorted(glob.glob(root + "/*/*.db")):
        con = sqlite3.connect(db)
        print("==", db)
        try:
            rows = con.execute(
                "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                "from kernels group by name order by 6 desc").fetchall()
            for r in rows[:6]:
                print(f"  {r[0][:70]:70s} calls {r[1]:4d} avg {r[2]/1e3:12.1f} us min {r[3]/1e3:12.1f} "
                      f"max {r[4]/1e3:12.1f} total {r[5]/1e6:10.2f} ms")
        except Exception as e:
            print("  kernels:", e)
        try:
            cols = [c[1] for c in con.execute("pragma table_info(counters_collection)")]
            name_col = "counter_name" if "counter_name" in cols else "name"
            kcol = "kernel_name" if "kernel_name" in cols else None
            q = (f"select {kcol}, {name_col}, count(*), sum(value), avg(value) from counters_collection "
                 f"group by {kcol}, {name_col}")
            for r in con.execute(q).fetchall():
                if "memetic" in (r[0] or "") or "gradient" in (r[0] or ""):
                    print(f"  {r[0][:40]:40s} {r[1]:32s} n {r[2]:4d} sum {r[3]:.6g} avg/dispatch {r[4]:.6g}")
        except Exception as e:
            print("  counters:", e)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r01a")
