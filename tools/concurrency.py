#!/usr/bin/env python3
"""Which kernels run beside which: from a rocprofv3 --kernel-trace rocpd database of the default bench, over the middle
half of the traced interval, (a) per kernel the time it is resident and the share of that time it is ALONE among the wide
kernels, (b) the share of wall time by the set of wide kernels resident, (c) the share of wall time nothing wide is resident.
usage: python3 tools/concurrency.py <dir with *.db>"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

WIDE = ("k_fast_cells", "k_orient_desc", "k_pyr_rows", "k_stereo_match", "k_octree", "k_compact")


def short(n):
    n = n.split("(anonymous namespace)::")[-1].split("(")[0].split("<")[0]
    return n


def main(root):
    for f in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        rows = [(short(n), s, e) for n, s, e in db.execute("select name, start, end from kernels order by start")]
        # the timed region: the span of the last two thirds of the FAST launches, its middle 70 %
        fast = [r for r in rows if r[0] == "k_fast_cells"]
        fast = fast[len(fast) // 3:]
        t0, t1 = fast[0][1], fast[-1][2]
        lo, hi = t0 + (t1 - t0) * 0.15, t0 + (t1 - t0) * 0.85
        ev = []
        for n, s, e in rows:
            if e <= lo or s >= hi or n not in WIDE:
                continue
            ev.append((max(s, lo), 1, n))
            ev.append((min(e, hi), -1, n))
        ev.sort()
        cur = defaultdict(int)
        last = lo
        by_set = defaultdict(float)
        for t, d, n in ev:
            if t > last:
                key = tuple(sorted(k for k, v in cur.items() if v > 0))
                by_set[key] += t - last
                last = t
            cur[n] += d
        by_set[()] += hi - last
        tot = hi - lo
        print(f"window {tot/1e6:.2f} ms of {os.path.basename(f)}")
        res = defaultdict(float)
        alone = defaultdict(float)
        for key, dt in by_set.items():
            for k in key:
                res[k] += dt
                if len(key) == 1:
                    alone[k] += dt
        for k in sorted(res, key=lambda k: -res[k]):
            print(f"  {k:18s} resident {100*res[k]/tot:5.1f} % of wall, alone {100*alone[k]/max(res[k],1):5.1f} % of that")
        print("  sets of wide kernels resident, share of wall:")
        for key, dt in sorted(by_set.items(), key=lambda kv: -kv[1])[:14]:
            print(f"    {100*dt/tot:5.1f} %  {' + '.join(k[2:] for k in key) or '(none)'}")


if __name__ == "__main__":
    main(sys.argv[1])
