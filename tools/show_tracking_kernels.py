#!/usr/bin/env python3
"""HIP-event durations of the kernels of configs[3]'s throughput leg, one lane alone (bench.tracking_batch_leg's "kernels" table).
usage: python3 tools/show_tracking_kernels.py [B=128] [th=7]"""
import importlib.util, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from fasttrack_amd import orb
ctx = orb.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
th = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
out = bench.tracking_batch_leg(orb, ctx, B=B, steps=4, warmup=2, ths=(th,), in_flight=1)
print("one lane: %.0f frames/s" % out["value"])
for k, v in out["kernels"].items():
    print("%-44s %5.1f launches/step  %7.3f ms/launch  %7.3f ms/step" % (k, v["launches_per_step"], v["avg_launch_ms"], v["ms_per_step"]))
