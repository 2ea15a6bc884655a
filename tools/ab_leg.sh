#!/bin/bash
# A/B of the configs[3] throughput leg's host loop on one box: lanes x overlap (bench.tracking_batch_leg(overlap=...)), alternating.
# usage (on the GPU box): bash tools/ab_leg.sh [rounds=3] [steps=24] "2:0 2:1 4:0 4:1"
cd ${GRAFT_REPO_ROOT:-.}
ROUNDS=${1:-3}; STEPS=${2:-24}; VARS=${3:-"2:0 2:1 4:0 4:1"}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    l=${v%%:*}; o=${v##*:}
    timeout 300 python3 tests/tools/bench_tracking_batch.py 128 $STEPS $l 1 0 $o 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('lanes $l overlap $o', ' '.join('%s %6.0f' % (k, v['value']) for k, v in d['by_th'].items()))"
  done
done | sort | awk '{k=$1" "$2" "$3" "$4; a[k]=a[k]" "$6"/"$8; s7[k]+=$6; s15[k]+=$8; n[k]++} END {for (k in a) printf "%-22s mean th7 %6.0f th15 %6.0f  runs%s\n", k, s7[k]/n[k], s15[k]/n[k], a[k]}' | sort
