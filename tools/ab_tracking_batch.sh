#!/bin/bash
# A/B of the configs[3] throughput leg on ONE box (boxes differ by ~15 %: host cores): alternates named environment variants.
# usage: tools/ab_tracking_batch.sh <rounds> name1="ENV=.." name2="ENV=.." ...     (an empty value = the default build)
ROUNDS=${1:-3}; shift
for r in $(seq 1 $ROUNDS); do for v in "$@"; do
  name=${v%%=*}; envs=${v#*=}
  env $envs python3 tests/tools/bench_tracking_batch.py 128 12 4 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['by_th']['7']['value']), round(d['by_th']['15']['value']))"
done; done | sort
