#!/bin/bash
# SQ instruction counters of the kernels of configs[3]'s throughput leg (B frames per launch, one lane): a pass of its own
# (--kernel-trace + --pmc only) -> gpurun_out/<tag>/tracking_sq.json, stamped with the library's csrc hash (bench.py prices the
# vector issue of k_resolve_batch and of the leg's extraction kernels with it while it runs that library).
# usage (on the GPU box): bash tools/pmc_tracking_batch.sh <tag> [B]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-trk_sq}
B=${2:-128}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
  -d $OUT/pmc_sq -o pmc -- python3 $REPO/tests/tools/bench_tracking_batch.py $B 2 1 > $OUT/sq.log 2>&1
cd $REPO
python3 - <<PY
import glob, json, sqlite3, sys
sys.path.insert(0, "$REPO")
from fasttrack_amd import orb
d = {}
for f in glob.glob("$OUT/pmc_sq/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for n, c, k, v in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        n = n.split("(anonymous namespace)::", 1)[-1].split("(")[0]
        if n.startswith("__amd"):
            continue
        e = d.setdefault(n, {})
        e[c] = v
        e["launches"] = k
out = {"csrc": orb.version().rsplit("csrc:", 1)[1].strip(), "batch_frames": $B,
       "source": "rocprofv3 --kernel-trace --pmc SQ_* -- python3 tests/tools/bench_tracking_batch.py $B 2 1 (one lane; averages per launch)",
       "kernels": {n: {"launches": int(e["launches"]), "waves_per_launch": e.get("SQ_WAVES"), "valu_per_launch": e.get("SQ_INSTS_VALU"),
                       "salu_per_launch": e.get("SQ_INSTS_SALU"), "lds_per_launch": e.get("SQ_INSTS_LDS"),
                       "wave_quad_cycles_per_launch": e.get("SQ_WAVE_CYCLES"), "wait_any_frac": (e.get("SQ_WAIT_ANY", 0) / max(e.get("SQ_WAVE_CYCLES", 1), 1))}
                   for n, e in sorted(d.items())}}
json.dump(out, open("$OUT/tracking_sq.json", "w"), indent=1)
for n, e in out["kernels"].items():
    print("%-44s launches %5d  valu/launch %14.0f  waves %10.0f" % (n[:44], e["launches"], e["valu_per_launch"] or 0, e["waves_per_launch"] or 0))
PY
rm -rf $OUT/pmc_sq
