#!/usr/bin/env python3
"""profiles/<tag>_rocprof_summary.txt -> profiles/<tag>_traffic.json: per-launch HBM-side bytes of every kernel
from the FETCH_SIZE / WRITE_SIZE passes, with the gfx950 corrections measured by tools/hbm_calib.sh
(profiles/r01_hbm_calibration_expected.txt): FETCH_SIZE is halved for coalesced streams, exact for the
row-segment tile loads k_fast_cells / k_pyr_down / k_orient_desc issue; WRITE_SIZE is exact."""
import json
import re
import sys

FETCH_FACTOR = {"k_fast_cells": 1.0, "k_pyr_down": 1.0, "k_orient_desc": 1.0}  # row-segment tile loads


def main(summary, out, images_per_step=256, steps=7):
    d = {}
    for line in open(summary):
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+launches\s+(\d+)\s+per-launch\s+([\d.]+) KB", line)
        if not m:
            continue
        k = m.group(1).split("<")[0]
        e = d.setdefault(k, {"launches": int(m.group(3))})
        e["fetch_kb" if m.group(2) == "FETCH_SIZE" else "write_kb"] = float(m.group(4))
    for k, e in d.items():
        f = FETCH_FACTOR.get(k, 2.0)
        e["fetch_factor"] = f
        e["traffic_bytes_per_launch"] = (e.get("fetch_kb", 0) * f + e.get("write_kb", 0)) * 1024
    for k, e in d.items():  # images one launch covers (bench under the profiler: `steps` passes over the batch)
        e["images_per_launch"] = images_per_step * steps / e["launches"] if e["launches"] else None
    json.dump({"source": summary, "note": "per launch; default bench (%d pairs of 1280x720 per step, %d steps profiled)" % (images_per_step // 2, steps),
               "kernels": d}, open(out, "w"), indent=1)
    print(json.dumps(d.get("k_fast_cells"), indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(int(a) for a in sys.argv[3:5]))
