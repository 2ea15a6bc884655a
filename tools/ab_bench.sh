#!/bin/bash
# A/B of whole-pipeline throughput on the GPU box: alternates bench.py runs between named environment variants.
# usage: tools/ab_bench.sh <rounds> <steps> "<bench args>" name1="ENV=.. ENV=.." name2="..." ...   (an empty value = default build)
ROUNDS=$1; STEPS=$2; ARGS=$3; shift 3
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%=*}; envs=${v#*=}
    val=$(env $envs python bench.py --no-cpu-baseline --no-host-in --no-workloads --steps $STEPS --warmup 5 $ARGS 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']))")
    echo "$name $val"
  done
done | sort | awk '{a[$1]=a[$1]" "$2; s[$1]+=$2; n[$1]++} END {for (k in a) printf "%-12s mean %8.0f  runs%s\n", k, s[k]/n[k], a[k]}'
