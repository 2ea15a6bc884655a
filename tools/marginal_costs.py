#!/usr/bin/env python3
"""Marginal cost of every kernel inside the overlapped pipeline (runs on the GPU box): FT_DEBUG_REPEAT=<kernel> makes the
launcher of that kernel enqueue it twice (all kernels are idempotent, results unchanged); the increase of the step time of
the default bench is what that kernel costs where it runs - which neither its event time nor its rocprofv3 duration show,
both being inflated by the kernels it shares the chip with.
usage: python3 tools/marginal_costs.py <out.json> [bench args...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path, extra = sys.argv[1], sys.argv[2:]


def run(repeat):
    env = dict(os.environ)
    if repeat:
        env["FT_DEBUG_REPEAT"] = repeat
    else:
        env.pop("FT_DEBUG_REPEAT", None)
    best = None
    for _ in range(2):  # best of two runs: a marginal cost is a difference of two noisy numbers
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-host-in", "--steps", "40"] + extra,
                           capture_output=True, text=True, env=env, check=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        best = d if best is None or d["ms_per_step"] < best["ms_per_step"] else best
    return best


base = run("")
names = {"fast": "k_fast_cells", "pyr": "k_pyr_rows (7 launches)", "octree": "k_octree", "orient": "k_orient_desc",
         "stereo": "k_stereo_match", "compact": "k_compact", "rowsort": "k_stereo_rowsort", "median": "k_stereo_median"}
doubled, marg = {}, {}
for key, name in names.items():
    d = run(key)
    doubled[key] = d["ms_per_step"]
    marg[name] = d["ms_per_step"] - base["ms_per_step"]
B = base["config"]["batch_pairs_per_gpu"]
lib = base.get("library", "")
json.dump({"method": __doc__.split("usage")[0].strip(), "csrc": lib.rsplit("csrc:", 1)[1].strip() if "csrc:" in lib else None,
           "workload": base["config"]["workload"], "batch_pairs": B,
           "distinct_pairs": base["config"]["distinct_pairs"], "baseline_ms_per_step": base["ms_per_step"],
           "baseline_frames_per_s": base["value"], "ms_per_step_with_kernel_doubled": doubled, "marginal_ms_per_step": marg,
           "marginal_ms_per_128_pairs": {k: v * 128.0 / B for k, v in marg.items()},
           "sum_of_marginals_over_step": sum(marg.values()) / base["ms_per_step"]}, open(out_path, "w"), indent=1)
print(open(out_path).read())
