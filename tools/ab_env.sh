#!/bin/bash
# A/B of one environment switch on one box: the default bench alternating with `VAR=value`, <runs> runs of <steps> steps each.
# usage: tools/ab_env.sh "FT_DEBUG_STEREO_ORDER=0" [runs=4] [steps=60] [bench args...]      (runs on the GPU box)
SW=$1; RUNS=${2:-4}; STEPS=${3:-60}; shift; shift; shift
cd ${GRAFT_REPO_ROOT:-.}
for r in $(seq 1 $RUNS); do
  a=$(python3 bench.py --steps $STEPS --warmup 6 --no-cpu-baseline --no-host-in --no-workloads "$@" 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().split('\n')[-1])['value'])")
  b=$(env $SW python3 bench.py --steps $STEPS --warmup 6 --no-cpu-baseline --no-host-in --no-workloads "$@" 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().split('\n')[-1])['value'])")
  echo "run $r: default $a   $SW $b"
done
