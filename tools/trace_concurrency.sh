#!/bin/bash
# kernel trace of the default bench -> tools/concurrency.py
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-conc}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t -o trace -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-in --no-workloads "$@" > $OUT/bench.log 2>&1
python3 $REPO/tools/concurrency.py $OUT/t > $OUT/concurrency.txt 2>&1
cat $OUT/concurrency.txt
rm -rf $OUT/t
