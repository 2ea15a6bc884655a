#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench, then the
# HBM-traffic counters in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass;
# --pmc must not be combined with trace domains other than --kernel-trace).
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -50
python3 $REPO/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only what is needed (the merge back is capped at 64 MiB)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
