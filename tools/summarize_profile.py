#!/usr/bin/env python3
"""Summarise a rocprofv3 output directory written by tools/profile_bench.sh:
per-kernel launch count / average / total time from the kernel trace, and per-launch FETCH_SIZE /
WRITE_SIZE (KB as reported by rocprofv3; see MI355X_MICROARCH.md for the gfx950 correction)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.split("::")[-1][:60]


def main(root):
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats:", os.path.relpath(f, root))
        print(open(f).read())
    tr = glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        agg = defaultdict(lambda: [0, 0.0])
        with open(tr[0]) as fh:
            for r in csv.DictReader(fh):
                d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += d
        tot = sum(v[1] for v in agg.values())
        print("== kernel trace summary (us)")
        print(f"{'kernel':40s} {'calls':>7s} {'avg_us':>10s} {'total_us':>12s} {'pct':>6s}")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:40s} {v[0]:7d} {v[1]/v[0]:10.1f} {v[1]:12.1f} {100*v[1]/tot:6.1f}")
    for which in ("pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(root, which, "**", "*counter_collection.csv"), recursive=True):
            agg = defaultdict(lambda: [0, 0.0])
            cname = None
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    cname = r["Counter_Name"]
                    a = agg[short(r["Kernel_Name"])]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
            print(f"== {which}: {cname} per launch (raw counter units as reported)")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"{k:40s} launches {v[0]:6d}  per-launch {v[1]/v[0]:14.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
