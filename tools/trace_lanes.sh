#!/bin/bash
# Where the GPU idles while configs[3]'s throughput leg runs with several steps in flight: rocprofv3 kernel + memory-copy trace of
# tests/tools/bench_tracking_batch.py, then (steady state = the last 60 % of the trace) the busy fraction of the device (union of
# kernel and copy intervals), the sum of kernel durations, and the largest idle gaps with the operations on either side.
# usage (on the GPU box): bash tools/trace_lanes.sh <tag> [B=128] [steps=10] [lanes=2]
#        TRACE_CMD="tools/extract_512_alone.py" bash tools/trace_lanes.sh <tag>      (any other script instead of the leg)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-lanes}
B=${2:-128}; STEPS=${3:-10}; LANES=${4:-2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o trace -- python3 $REPO/${TRACE_CMD:-tests/tools/bench_tracking_batch.py $B $STEPS $LANES 1 0} > $OUT/trace.log 2>&1
timeout 200 python3 - $OUT <<'PY' > $OUT/lanes.txt
import sqlite3, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    ev = [(s, e, n.split("(anonymous namespace)::", 1)[-1].split("(")[0][:36], "k") for n, s, e in db.execute("select name, start, end from kernels")]
    try:
        ev += [(s, e, "copy " + str(n)[:30], "c") for n, s, e in db.execute("select name, start, end from memory_copies")]
    except Exception as ex:
        print("no memory_copies view:", ex)
    ev.sort()
    t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
    cut = t0 + 0.4 * (t1 - t0)
    ss = [x for x in ev if x[0] >= cut]
    busy, cs, ce = 0, ss[0][0], ss[0][1]
    gaps = []
    last = ss[0]
    for x in ss[1:]:
        if x[0] > ce:
            busy += ce - cs
            gaps.append((x[0] - ce, last[2], x[2]))
            cs, ce = x[0], x[1]
            last = x
        elif x[1] > ce:
            ce = x[1]
            last = x
    busy += ce - cs
    wall = ce - ss[0][0]
    ksum = sum(e - s for s, e, _, k in ss if k == "k")
    print("steady state: wall %.1f ms, device busy %.1f ms (%.1f %%), sum of kernel durations %.1f ms (%.2f x wall)" % (wall / 1e6, busy / 1e6, 100 * busy / wall, ksum / 1e6, ksum / wall))
    print("idle: %.1f ms in %d gaps; by the operation that follows the gap:" % ((wall - busy) / 1e6, len(gaps)))
    by = {}
    for g, a, b in gaps:
        by.setdefault(b, [0, 0.0]); by[b][0] += 1; by[b][1] += g
    for b, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-40s %5d gaps %8.2f ms" % (b, c, t / 1e6))
    print("by the operation that precedes the gap:")
    by = {}
    for g, a, b in gaps:
        by.setdefault(a, [0, 0.0]); by[a][0] += 1; by[a][1] += g
    for b, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-40s %5d gaps %8.2f ms" % (b, c, t / 1e6))
    print("operations in the steady state (count, average us, total ms):")
    agg = {}
    for s0, e0, n, k in ss:
        agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += e0 - s0
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print("  %-40s %6d %9.1f %9.2f" % (n, c, t / c / 1e3, t / 1e6))
PY
cat $OUT/lanes.txt
tail -1 $OUT/trace.log | cut -c1-300
rm -rf $OUT/trace
