#!/bin/bash
# The round's measurement set (runs on the GPU box): profiles of the default bench, the dense and the planes scenes, marginal
# costs, the driver's default line, a long run, the latency tool, the tracking-leg trace.  usage: bash tools/final_measure.sh [r04]
set -u
R=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
bash tools/profile_bench.sh $R > gpurun_out/final/prof_$R.log 2>&1
bash tools/profile_bench.sh ${R}_dense --mosaic 10 > gpurun_out/final/prof_${R}_dense.log 2>&1
bash tools/profile_bench.sh ${R}_planes --scene planes > gpurun_out/final/prof_${R}_planes.log 2>&1
python3 tools/marginal_costs.py gpurun_out/final/${R}_marginal_costs.json --no-workloads > /dev/null 2>&1
python3 tools/marginal_costs.py gpurun_out/final/${R}_dense_marginal_costs.json --no-workloads --mosaic 10 > /dev/null 2>&1
python3 tools/marginal_costs.py gpurun_out/final/${R}_planes_marginal_costs.json --no-workloads --scene planes > /dev/null 2>&1
python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-host-in --no-workloads > gpurun_out/final/bench_long.log 2>&1
python3 tools/bench_latency.py > gpurun_out/final/latency.log 2>&1
bash tools/ab_persistent.sh ${R}_trk_final > gpurun_out/final/tracking.log 2>&1
tail -1 gpurun_out/final/bench_long.log | cut -c1-300
