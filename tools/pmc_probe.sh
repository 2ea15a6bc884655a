#!/bin/bash
# One rocprofv3 counter pass (kernel-trace + --pmc only) over any python script of the repo; prints per-kernel averages.
# usage: [ENV=...] tools/pmc_probe.sh <tag> "<COUNTER ...>" <script.py> [args]
TAG=${1:-any}; CNT=${2:-SQ_WAVES}; SCRIPT=$3; shift; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT -d $OUT/sq -o pmc -- python3 $REPO/$SCRIPT "$@" > $OUT/run.log 2>&1
python3 - <<PY
import sqlite3, glob
for f in glob.glob("$OUT/sq/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    d = {}
    for n, c, k, v in rows:
        n = n.split("(anonymous namespace)::")[-1].split("(")[0]
        d.setdefault(n, {})[c] = v
    names = sorted({c for v in d.values() for c in v})
    print("kernel".ljust(26), " ".join(c.replace("SQ_", "")[:16].rjust(17) for c in names))
    for n, c in sorted(d.items()):
        if n.startswith("__amd"): continue
        print(n[:26].ljust(26), " ".join(f"{c.get(k, 0):17.0f}" for k in names))
PY
rm -rf $OUT/sq
