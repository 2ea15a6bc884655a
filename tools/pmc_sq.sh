#!/bin/bash
# SQ instruction / cycle counters per kernel (own pass: --kernel-trace + --pmc only).  usage: tools/pmc_sq.sh <tag> [bench args]
TAG=${1:-sq}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
  -d $OUT/pmc_sq -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-in "$@" > $OUT/bench_sq.log 2>&1
python3 $REPO/tools/summarize_profile.py $OUT
rm -rf $OUT/pmc_sq
