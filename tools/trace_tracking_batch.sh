#!/bin/bash
# rocprofv3 kernel trace + stats of the configs[3] throughput leg (B frames per launch): per-kernel table, and the timeline of
# one step.  usage (on the GPU box): bash tools/trace_tracking_batch.sh <tag> [B] [steps]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-trk_batch}
B=${2:-128}; STEPS=${3:-4}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/tests/tools/bench_tracking_batch.py $B $STEPS > $OUT/trace.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 tests/tools/bench_tracking_batch.py $B $STEPS   (configs[3] throughput leg: $B frames per launch,"
  echo "# 512x512 KB8 two-camera frames, nFeatures 2000, th 7 and th 15, $STEPS steps each + warm-up)"
  python3 $REPO/tools/summarize_profile.py $OUT; } > $OUT/summary.txt
cat $OUT/summary.txt
python3 - <<PY > $OUT/timeline.txt
import sqlite3, glob
for f in glob.glob("$OUT/trace/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-260:]
    t0 = rows[0][1]
    prev = t0
    for n, s, e in rows:
        n = n.split("(anonymous namespace)::", 1)[-1].split("(")[0][:40]
        print(f"{(s-t0)/1e3:10.1f} us  gap {(s-prev)/1e3:7.1f}  dur {(e-s)/1e3:7.1f}  {n}")
        prev = e
PY
tail -1 $OUT/trace.log | cut -c1-1500
rm -rf $OUT/trace
