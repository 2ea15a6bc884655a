#!/bin/bash
# A/B of the two forms of the projection searches' claim iteration on bench.py's configs[3] tracking leg: every pass in one
# persistent launch (FT_SEARCH_PERSISTENT=1, default) against a launch per pass (=0); three alternating rounds, then a
# rocprofv3 kernel trace + stats of the leg in the default form (summary kept as profiles/r04_tracking_rocprof_summary.txt).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r04_trk}
mkdir -p $OUT
cd $REPO
for r in 1 2 3; do for v in 1 0; do
  echo -n "persistent=$v " >> $OUT/ab.txt
  FT_SEARCH_PERSISTENT=$v python3 tools/profile_tracking_leg.py 24 2>/dev/null | tail -1 >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
[ -n "$NO_TRACE" ] && exit 0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/tools/profile_tracking_leg.py 24 > $OUT/trace.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 tools/profile_tracking_leg.py 24   (configs[3] tracking leg: 512x512 KB8 two-camera frames,"
  echo "# nFeatures 2000, th 7 and th 15, 24 frames each + 2 warm-up: extraction + fisheye match + upload + search_last_frame + track_local_map per frame)"
  python3 $REPO/tools/summarize_profile.py $OUT; } > $OUT/tracking_rocprof_summary.txt
cat $OUT/tracking_rocprof_summary.txt
tail -1 $OUT/trace.log
ls -R $OUT/trace | head -20
rm -rf $OUT/trace
