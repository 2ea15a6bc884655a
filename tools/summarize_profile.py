#!/usr/bin/env python3
"""Summarise a rocprofv3 output directory written by tools/profile_bench.sh (rocpd sqlite output):
per-kernel launch count / average / total time from the kernel trace, and per-launch FETCH_SIZE /
WRITE_SIZE (KB as rocprofv3 reports them; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads exactly half
of a wide coalesced stream, other widths uncalibrated)."""
import glob
import os
import sqlite3
import sys


def short(name):
    name = name.split("(anonymous namespace)::")[-1]
    return name.split("(")[0][:48]


def main(root):
    for f in glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        print("== rocprofv3 --kernel-trace --stats:", os.path.relpath(f, root))
        print(f"{'kernel':48s} {'calls':>7s} {'avg_us':>10s} {'total_us':>12s} {'pct':>6s} {'vgpr':>5s} {'lds':>6s}")
        rows = db.execute("select name, count(*), avg(duration), sum(duration), max(vgpr_count), max(lds_size) "
                          "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[3] for r in rows) or 1
        for n, c, a, s, v, l in rows:
            print(f"{short(n):48s} {c:7d} {a/1e3:10.1f} {s/1e3:12.1f} {100*s/tot:6.1f} {v:5d} {l:6d}")
    for which in ("pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(root, which, "**", "*.db"), recursive=True):
            db = sqlite3.connect(f)
            rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                              "group by kernel_name, counter_name order by avg(value) desc").fetchall()
            print(f"== {which}: per-launch counter value (KB as reported by rocprofv3)")
            for n, cn, c, a in rows:
                print(f"{short(n):48s} {cn:12s} launches {c:6d}  per-launch {a:14.1f} KB")


if __name__ == "__main__":
    main(sys.argv[1])
