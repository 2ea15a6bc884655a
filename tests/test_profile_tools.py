"""The committed profile artefacts are reproducible from each other (CPU): profiles/r02_traffic.json is what
tools/traffic_from_profile.py makes of profiles/r02_rocprof_summary.txt, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("tag", ["r02", "r04"])
def test_traffic_json_follows_from_the_rocprof_summary(tmp_path, tag):
    out = tmp_path / "t.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_from_profile.py"),
                    os.path.join(ROOT, "profiles", tag + "_rocprof_summary.txt"), str(out), "--steps", "7"], check=True, capture_output=True)
    new, old = json.load(open(out)), json.load(open(os.path.join(ROOT, "profiles", tag + "_traffic.json")))
    assert new["kernels"] == old["kernels"] and new["batch_pairs"] == old["batch_pairs"] == 512 and old["distinct_pairs"] == 512
    f = new["kernels"]["k_fast_cells"]
    assert f["fetch_factor"] == 2.0 and f["images_per_launch"] == 256.0
    assert abs(f["traffic_bytes_per_launch"] - (2 * f["fetch_kb_raw"] + f["write_kb"]) * 1024) < 1
    # FAST + NMS reads every pyramid pixel once: 256 images x 2 853 088 px; the memory side sees about that, not a third of it
    assert 0.9 < f["traffic_bytes_per_launch"] / (256 * 2853088) < 1.15
    assert 0.5 < f["valu_issue_frac"] < 1.0 and f["valu_per_wave"] > 500 and f["salu_per_wave"] > 100


@pytest.mark.parametrize("tag", ["r02", "r04"])
def test_bench_line_reports_the_committed_traffic(tag):
    d = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_line.json")))
    t = json.load(open(os.path.join(ROOT, "profiles", tag + "_traffic.json")))["kernels"]["k_fast_cells"]
    r = d["roofline"]
    assert r["kernel"] == "k_fast_cells" and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["traffic"] - t["traffic_bytes_per_launch"]) / t["traffic_bytes_per_launch"] < 0.01
    assert d["config"]["distinct_pairs"] == d["config"]["batch_pairs_per_gpu"] == 512
    assert d["host_in"]["value"] < d["value"] and d["cpu_baseline"]["kind"] == "port" and d["device_octree_fallbacks"] == 0
