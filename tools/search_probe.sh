#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-search_probe}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t -o trace -- python3 $REPO/tools/search_probe.py > $OUT/log.txt 2>&1
grep "^th" $OUT/log.txt
python3 - <<PY
import sqlite3, glob
for f in glob.glob("$OUT/t/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    prev = None
    out = []
    for n, s, e in rows:
        n = n.split("(anonymous namespace)::")[-1].split("(")[0]
        if n == "k_search_local" and prev != "k_search_local":
            out.append([])
        if n == "k_search_local":
            out[-1].append((e - s) / 1e3)
        prev = n
    for o in out:
        print("first pass %6.1f us, second %5.1f, later avg %5.1f (%d launches)" % (o[0], o[1] if len(o) > 1 else 0, sum(o[2:]) / max(len(o) - 2, 1), len(o)))
PY
rm -rf $OUT/t
