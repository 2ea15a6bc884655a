#!/bin/bash
# kernel trace of the tracking-search latency tool: per-kernel order, start offsets and durations of the last frames
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_tracking
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t -o trace -- python3 $REPO/tests/tools/bench_tracking.py 5 > $OUT/log.txt 2>&1
python3 - <<PY
import sqlite3, glob
for f in glob.glob("$OUT/t/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-120:]
    t0 = rows[0][1]
    prev = t0
    for n, s, e in rows:
        n = n.split("(anonymous namespace)::")[-1].split("(")[0][:40]
        print(f"{(s-t0)/1e3:10.1f} us  gap {(s-prev)/1e3:7.1f}  dur {(e-s)/1e3:7.1f}  {n}")
        prev = e
PY
rm -rf $OUT/t
