#!/bin/bash
# runs on the GPU box: builds tools/hbm_calib.hip, collects FETCH_SIZE and WRITE_SIZE in separate passes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/hbm_calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/tools/hbm_calib.hip -o /tmp/hbm_calib || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o pmc -- /tmp/hbm_calib > $OUT/expected.txt 2>/dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o pmc -- /tmp/hbm_calib > /dev/null 2>&1
cat $OUT/expected.txt
python3 - <<PY
import sqlite3, glob
for which in ("f", "w"):
    for f in glob.glob("$OUT/%s/**/*.db" % which, recursive=True):
        db = sqlite3.connect(f)
        for n, c, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            if "rocclr" in n: continue
            print(f"{n.split('(')[0]:14s} {c:11s} {v:14.1f} KB")
PY
rm -rf $OUT/f $OUT/w
