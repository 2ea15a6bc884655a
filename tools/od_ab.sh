#!/bin/bash
# A/B of k_orient_desc between the default library and a variant (runs on the GPU box): the kernel's own duration (rocprofv3
# kernel trace), its SQ counters, and its marginal cost inside the pipeline (FT_DEBUG_REPEAT=orient: the step time with the kernel
# enqueued twice minus the plain step time).
# usage: tools/od_ab.sh <variant .so>
VAR=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/od_ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 12 --warmup 3 --no-cpu-baseline --no-host-in --no-workloads"
for v in default variant; do
  if [ $v = variant ]; then export FT_LIB=$REPO/$VAR; else unset FT_LIB; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/tr_$v -o t -- python3 $REPO/bench.py $ARGS > $OUT/tr_$v.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $OUT/sq_$v -o p -- python3 $REPO/bench.py $ARGS > $OUT/sq_$v.log 2>&1
  echo "== $v"
  timeout 120 python3 - $OUT/tr_$v $OUT/sq_$v <<'PY'
import glob, sqlite3, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    for n, c, a in sqlite3.connect(f).execute("select name, count(*), avg(duration) from kernels where name like '%orient_desc%' or name like '%fast_cells%' group by name"):
        print("%-60s launches %4d  avg %8.1f us" % (n.split("(anonymous namespace)::", 1)[-1][:60], c, a / 1e3))
for f in glob.glob(sys.argv[2] + "/**/*.db", recursive=True):
    rows = sqlite3.connect(f).execute("select counter_name, avg(value) from counters_collection where kernel_name like '%orient_desc%' group by counter_name").fetchall()
    d = dict(rows); w = d.get("SQ_WAVES", 1)
    print("k_orient_desc per wave:", {k: round(v / w, 1) for k, v in d.items() if k != "SQ_WAVES"}, "(cycle counters: quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES: cycles)")
PY
  for rep in "" orient "" orient; do
    FT_DEBUG_REPEAT=$rep timeout 300 python3 $REPO/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-host-in --no-workloads 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('repeat=[$rep] ms_per_step %.4f  %.0f frames/s' % (d['ms_per_step'], d['value']))"
  done
  rm -rf $OUT/tr_$v $OUT/sq_$v
done
