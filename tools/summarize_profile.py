#!/usr/bin/env python3
"""Summarise a rocprofv3 output directory written by tools/profile_bench.sh (rocpd sqlite output):
per-kernel launch count / average / total time from the kernel trace, per-launch FETCH_SIZE / WRITE_SIZE
(KB as rocprofv3 reports them; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 128-B requests at 64 B,
so it is DOUBLED before it is compared with bytes - tools/traffic_from_profile.py does that) and the SQ
instruction counters per wave and per launch."""
import glob
import os
import sqlite3
import sys


def short(name):
    # "void (anonymous namespace)::k_name<true>((anonymous namespace)::Rebase, int, ...)" -> "k_name<true>"
    name = name.split("(anonymous namespace)::", 1)[-1]
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
    for f in glob.glob(os.path.join(root, "pmc_sq", "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                          "group by kernel_name, counter_name").fetchall()
        d = {}
        for n, c, k, v in rows:
            e = d.setdefault(short(n), {})
            e[c] = v
            e["launches"] = k
        print("== pmc_sq: SQ counters, averages per launch (cycle counters are quad-cycles summed over waves)")
        for n, c in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
            if n.startswith("__amd"):
                continue
            w = c.get("SQ_WAVES", 1) or 1
            wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
            print(f"{n:48s} SQ launches {c['launches']:6d} waves {w:11.0f} valu_per_launch {c.get('SQ_INSTS_VALU', 0):14.0f} "
                  f"valu/wave {c.get('SQ_INSTS_VALU', 0)/w:8.1f} salu/wave {c.get('SQ_INSTS_SALU', 0)/w:8.1f} lds/wave {c.get('SQ_INSTS_LDS', 0)/w:7.1f} "
                  f"cyc/wave {4*c.get('SQ_WAVE_CYCLES', 0)/w:9.0f} wait_any% {100*c.get('SQ_WAIT_ANY', 0)/wc:5.1f} "
                  f"active% {100*c.get('SQ_ACTIVE_INST_ANY', 0)/wc:5.1f} wait_inst% {100*c.get('SQ_WAIT_INST_ANY', 0)/wc:5.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
