#!/bin/bash
# The round's measurement set (runs on the GPU box): rocprofv3 profiles (kernel trace + FETCH / WRITE / SQ passes) of the default
# bench, the dense and the planes scenes, with the per-launch traffic stamped with the library's csrc hash; marginal costs; the
# driver's default line and a long run; the configs[3] legs (one frame at a time, and B frames per launch) as kernel traces; the
# LDS / issue counters of the pipeline.  usage: bash tools/final_measure.sh [r06]     -> gpurun_out/final/
set -u
R=${1:-r06}
cd $GRAFT_REPO_ROOT
F=gpurun_out/final
mkdir -p $F
CSRC=$(python3 -c "from fasttrack_amd import orb; print(orb.version().rsplit('csrc:', 1)[1])")
echo "library csrc $CSRC" > $F/library.txt
for v in "" dense planes; do
  tag=$R${v:+_$v}
  args=""; [ "$v" = dense ] && args="--mosaic 10"; [ "$v" = planes ] && args="--scene planes"
  bash tools/profile_bench.sh $tag $args > $F/prof_$tag.log 2>&1
  cp gpurun_out/prof_$tag/summary.txt $F/${tag}_rocprof_summary.txt
  cp gpurun_out/prof_$tag/bench_line_under_profiler.json $F/${tag}_bench_line_under_profiler.json 2>/dev/null
  python3 tools/traffic_from_profile.py $F/${tag}_rocprof_summary.txt $F/${tag}_traffic.json --steps 7 --csrc $CSRC > /dev/null 2>&1
  python3 tools/marginal_costs.py $F/${tag}_marginal_costs.json --no-workloads $args > /dev/null 2>&1
done
# the traffic / marginal costs of the default workload are what bench.py reports beside its live numbers
mkdir -p profiles
cp $F/${R}_traffic.json $F/${R}_marginal_costs.json profiles/ 2>/dev/null
# the SQ instruction counts of the tracking leg's kernels (bench.py prices k_resolve_batch and the leg's extraction with them)
bash tools/pmc_tracking_batch.sh final/trk_sq 128 > $F/trk_sq.log 2>&1
cp $F/trk_sq/tracking_sq.json profiles/${R}_tracking_sq.json 2>/dev/null
cp $F/trk_sq/tracking_sq.json $F/${R}_tracking_sq.json 2>/dev/null
python3 bench.py > $F/${R}_bench_line.json 2> $F/bench_line.err
python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-host-in --no-workloads > $F/${R}_bench_long.json 2>/dev/null
python3 tools/bench_latency.py > $F/${R}_latency.json 2>/dev/null
bash tools/trace_tracking_batch.sh final/trk_batch 128 4 > /dev/null 2>&1
cp $F/trk_batch/summary.txt $F/${R}_tracking_batch_rocprof_summary.txt
head -150 $F/trk_batch/timeline.txt > $F/${R}_tracking_batch_timeline.txt
grep -a '^{"metric"' $F/trk_batch/trace.log | tail -1 > $F/${R}_tracking_batch_under_profiler.json
python3 tests/tools/bench_tracking_batch.py 128 12 4 2>/dev/null | tail -1 > $F/${R}_tracking_batch.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$F/trk_single/trace -o trace -- python3 $GRAFT_REPO_ROOT/tools/profile_tracking_leg.py 24 > $GRAFT_REPO_ROOT/$F/trk_single.log 2>&1 )
{ echo "# rocprofv3 --kernel-trace --stats -- python3 tools/profile_tracking_leg.py 24   (configs[3] one frame at a time: 512x512 KB8 two-camera frames,"
  echo "# nFeatures 2000, th 7 and th 15, 24 frames each + warm-up)"
  python3 tools/summarize_profile.py $F/trk_single; } > $F/${R}_tracking_rocprof_summary.txt
rm -rf $F/trk_single/trace
bash tools/issue_gaps.sh final/issue_gaps > /dev/null 2>&1
cp $F/issue_gaps/issue_gaps.txt $F/${R}_issue_gaps.txt 2>/dev/null
tail -c 400 $F/${R}_bench_long.json
