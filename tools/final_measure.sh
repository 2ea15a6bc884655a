set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
bash tools/profile_bench.sh r03 > gpurun_out/final/prof_r03.log 2>&1
bash tools/profile_bench.sh r03_dense --mosaic 10 > gpurun_out/final/prof_r03_dense.log 2>&1
python3 tools/marginal_costs.py gpurun_out/final/r03_marginal_costs.json --no-workloads > /dev/null 2>&1
python3 tools/marginal_costs.py gpurun_out/final/r03_dense_marginal_costs.json --no-workloads --mosaic 10 > /dev/null 2>&1
python3 bench.py > gpurun_out/final/bench_line.log 2>&1
python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-host-in --no-workloads > gpurun_out/final/bench_long.log 2>&1
python3 tools/bench_latency.py > gpurun_out/final/latency.log 2>&1
tail -1 gpurun_out/final/bench_line.log | cut -c1-400
