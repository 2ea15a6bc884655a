#!/bin/bash
# kernel timeline of the last frames of the configs[3] tracking leg: start offset, gap to the previous kernel's end, duration
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-trace_trk}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t -o trace -- python3 $REPO/tools/profile_tracking_leg.py 6 > $OUT/log.txt 2>&1
python3 - <<PY > $OUT/timeline.txt
import sqlite3, glob
for f in glob.glob("$OUT/t/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-${2:-140}:]
    t0 = rows[0][1]
    prev = t0
    for n, s, e in rows:
        n = n.split("(anonymous namespace)::")[-1].split("(")[0][:40]
        print(f"{(s-t0)/1e3:10.1f} us  gap {(s-prev)/1e3:7.1f}  dur {(e-s)/1e3:7.1f}  {n}")
        prev = e
PY
rm -rf $OUT/t
cat $OUT/timeline.txt
