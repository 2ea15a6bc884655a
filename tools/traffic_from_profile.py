#!/usr/bin/env python3
"""profiles/<tag>_rocprof_summary.txt -> profiles/<tag>_traffic.json: per-launch HBM-side bytes of every kernel
from the FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts
128-byte read requests at 64 bytes each, so it is DOUBLED for every kernel (tools/hbm_calib.sh confirms it for the
access patterns of this repo: a coalesced dword / 16-byte stream and the 48-byte row segments of the tile loads both
read back as half of the 128-byte requests they touch); WRITE_SIZE is exact.  Also records, per kernel, the average
rocprofv3 duration and the VALU issue fraction = SQ_INSTS_VALU x the mean issue cycles of the kernel's opcode mix
(profiles/r06_valu_mix.json: tools/valu_mix.py over the measured table profiles/r06_valu_rates.txt; 4.33 - the dear class - for a
kernel the file does not hold; rounds 1 - 5 priced every instruction at 4) / (1024 SIMDs x duration x 2.4 GHz).

usage: traffic_from_profile.py <summary.txt> <out.json> --workload W --batch B --distinct D --steps S"""
import argparse
import json
import os
import re

FETCH_FACTOR = 2.0
SIMDS, CLOCK_HZ = 1024, 2.4e9
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def valu_cycles(kernel):
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "r06_valu_mix.json")))["kernels"]
    except Exception:
        mix = {}
    for k, v in mix.items():
        if k == kernel or k.split("<")[0] == kernel:
            return float(v["mean_cycles_per_valu"])
    return 4.33


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("out")
    ap.add_argument("--workload", default="stereo_1280x720_nf2000")
    ap.add_argument("--batch", type=int, default=512, help="pairs per step")
    ap.add_argument("--distinct", type=int, default=512)
    ap.add_argument("--steps", type=int, default=7, help="passes over the batch the profiled run made (warmup + steps)")
    ap.add_argument("--csrc", default=None, help="hash of fasttrack_amd/csrc the profiled library was built from (ft_version); bench.py "
                                                 "uses the file only while it runs that library")
    a = ap.parse_args()
    d = {}
    for line in open(a.summary):
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+launches\s+(\d+)\s+per-launch\s+([\d.]+) KB", line)
        if m:
            e = d.setdefault(m.group(1).split("<")[0], {})
            e["launches"] = int(m.group(3))
            e["fetch_kb_raw" if m.group(2) == "FETCH_SIZE" else "write_kb"] = float(m.group(4))
            continue
        m = re.match(r"(.+?)\s+SQ launches\s+(\d+)\s+waves\s+([\d.]+)\s+valu_per_launch\s+([\d.]+)\s+valu/wave\s+([\d.]+)\s+salu/wave\s+([\d.]+)\s+lds/wave\s+([\d.]+)", line)
        if m:
            e = d.setdefault(m.group(1).split("<")[0], {})
            e.update(waves_per_launch=float(m.group(3)), valu_per_launch=float(m.group(4)), valu_per_wave=float(m.group(5)),
                     salu_per_wave=float(m.group(6)), lds_per_wave=float(m.group(7)))
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(\d+)\s*$", line)
        if m and not line.startswith("kernel"):
            e = d.setdefault(m.group(1).split("<")[0], {})
            e.update(trace_launches=int(m.group(2)), avg_us=float(m.group(3)), pct_of_kernel_time=float(m.group(5)))
    images_per_step = 2 * a.batch
    for k, e in d.items():
        if "launches" in e:
            e["fetch_factor"] = FETCH_FACTOR
            e["traffic_bytes_per_launch"] = (e.get("fetch_kb_raw", 0) * FETCH_FACTOR + e.get("write_kb", 0)) * 1024
            e["images_per_launch"] = images_per_step * a.steps / e["launches"] if e["launches"] else None
        if "valu_per_launch" in e and e.get("avg_us"):
            e["valu_cycles_per_instruction"] = valu_cycles(k)
            e["valu_issue_frac"] = e["valu_per_launch"] * e["valu_cycles_per_instruction"] / (SIMDS * e["avg_us"] * 1e-6 * CLOCK_HZ)
    json.dump({"source": a.summary, "csrc": a.csrc, "workload": a.workload, "batch_pairs": a.batch, "distinct_pairs": a.distinct,
               "note": "per launch; bench of %d pairs per step, %d passes profiled; FETCH_SIZE doubled (128-B requests counted at 64 B)"
                       % (a.batch, a.steps), "kernels": d}, open(a.out, "w"), indent=1)
    print(json.dumps(d.get("k_fast_cells"), indent=1))


if __name__ == "__main__":
    main()
