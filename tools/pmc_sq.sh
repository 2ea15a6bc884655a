#!/bin/bash
# SQ instruction / cycle counters per kernel (own pass, kernel-trace only).  usage: tools/pmc_sq.sh <tag> [bench args]
TAG=${1:-sq}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
  -d $OUT/sq -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import sqlite3, glob
for f in glob.glob("$OUT/sq/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    d = {}
    for n, c, k, v in rows:
        n = n.split("(anonymous namespace)::")[-1].split("(")[0]
        d.setdefault(n, {})[c] = v
    for n, c in d.items():
        w = c.get("SQ_WAVES", 1) or 1
        print(f"{n:28s} waves {w:9.0f} valu/wave {c.get('SQ_INSTS_VALU',0)/w:8.1f} salu/wave {c.get('SQ_INSTS_SALU',0)/w:8.1f} lds/wave {c.get('SQ_INSTS_LDS',0)/w:7.1f} "
              f"cyc/wave {4*c.get('SQ_WAVE_CYCLES',0)/w:9.0f} wait_any% {100*c.get('SQ_WAIT_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):5.1f} "
              f"active% {100*c.get('SQ_ACTIVE_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):5.1f} wait_inst% {100*c.get('SQ_WAIT_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):5.1f}")
PY
rm -rf $OUT/sq
