#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench, then the
# HBM-traffic counters and the SQ instruction counters, each in its own pass (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; --pmc must not be combined with trace domains other than --kernel-trace).
# usage: tools/profile_bench.sh <tag> [bench args...]     (the program after `--` is python3 itself, no wrapper)
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-host-in --no-workloads $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
  -d $OUT/pmc_sq -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq.log 2>&1
python3 $REPO/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
grep -a '^{"metric"' $OUT/bench_trace.log | tail -1 > $OUT/bench_line_under_profiler.json
# keep only what is needed (the merge back is capped at 64 MiB)
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
