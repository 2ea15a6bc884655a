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
    # (rounds 1 - 5 priced a vector instruction at 4 cycles; from round 6 on the tool prices it with the measured opcode mix:
    # everything else of a regenerated file equals the committed one)
    strip = lambda ks: {k: {f: v for f, v in e.items() if f not in ("valu_issue_frac", "valu_cycles_per_instruction")} for k, e in ks.items()}
    assert strip(new["kernels"]) == strip(old["kernels"]) and new["batch_pairs"] == old["batch_pairs"] == 512 and old["distinct_pairs"] == 512
    for k, e in new["kernels"].items():
        if "valu_issue_frac" in e:
            assert abs(e["valu_issue_frac"] / old["kernels"][k]["valu_issue_frac"] - e["valu_cycles_per_instruction"] / 4.0) < 1e-6
    f = new["kernels"]["k_fast_cells"]
    assert f["fetch_factor"] == 2.0 and f["images_per_launch"] == 256.0
    assert abs(f["traffic_bytes_per_launch"] - (2 * f["fetch_kb_raw"] + f["write_kb"]) * 1024) < 1
    # FAST + NMS reads every pyramid pixel once: 256 images x 2 853 088 px; the memory side sees about that, not a third of it
    assert 0.9 < f["traffic_bytes_per_launch"] / (256 * 2853088) < 1.15
    assert 0.5 < f["valu_issue_frac"] < 1.05 and 4.0 < f["valu_cycles_per_instruction"] < 4.5 and f["valu_per_wave"] > 500 and f["salu_per_wave"] > 100


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


def test_a_profile_of_another_library_build_is_dropped(tmp_path, monkeypatch):
    """bench.py reports PMC traffic / marginal costs from committed artefacts only while it runs the library they were measured
    on: the artefacts carry the hash of fasttrack_amd/csrc (ft_version), a stale or unstamped one yields nothing"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.csrc_stamp("fasttrack_amd 0.5 (gfx950) csrc:0123456789ab") == "0123456789ab" and bench.csrc_stamp("old library") == ""
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "a.json").write_text(json.dumps({"csrc": "0123456789ab", "kernels": {"k": {"traffic_bytes_per_launch": 1}}}))
    (prof / "b.json").write_text(json.dumps({"kernels": {}}))
    d, state = bench.load_profile("a.json", "0123456789ab")
    assert state == "current" and d["kernels"]["k"]["traffic_bytes_per_launch"] == 1
    d, state = bench.load_profile("a.json", "ffffffffffff")
    assert d == {} and state.startswith("stale")
    assert bench.load_profile("b.json", "0123456789ab")[0] == {}           # never stamped
    assert bench.load_profile("a.json", "")[0] == {}                        # a library without a stamp
    assert bench.load_profile("missing.json", "0123456789ab") == ({}, "missing")
    # the library's own stamp is what its Makefile computes: the hash of the csrc sources
    from fasttrack_amd import orb
    import hashlib, glob
    csrc = os.path.join(ROOT, "fasttrack_amd", "csrc")
    files = sorted(f for ext in ("*.cpp", "*.hip", "*.h", "*.inc") for f in glob.glob(os.path.join(csrc, ext)) if not f.endswith("version.inc"))
    h = hashlib.sha1(b"".join(open(f, "rb").read() for f in files)).hexdigest()[:12]
    assert bench.csrc_stamp(orb.version()) == h


def test_committed_r06_artefacts_belong_to_this_library():
    """the round's traffic / marginal-cost artefacts were measured on the library this tree builds (same csrc hash): bench.py's
    driver line will carry them; and the committed bench line reports them as current.  (Skipped while the round's measurement
    set has not been committed yet; a committed set of ANOTHER build fails.)"""
    import importlib.util
    import pytest
    if not os.path.exists(os.path.join(ROOT, "profiles", "r06_bench_line.json")):
        pytest.skip("profiles/r06_bench_line.json not committed yet (tools/final_measure.sh r06)")
    from fasttrack_amd import orb
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stamp = bench.csrc_stamp(orb.version())
    for name in (bench.TRAFFIC_JSON, bench.MARGINAL_JSON):
        d, state = bench.load_profile(name, stamp)
        assert state == "current", (name, state)
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")))
    assert bench.csrc_stamp(line["library"]) == stamp and set(line["profiles"].values()) == {"current"}
    assert line["roofline"]["traffic"] and 0.9 < line["roofline"]["traffic_over_algorithmic"] < 1.1
    thr = line["workloads"]["tracking_512x512_nf2000"]["throughput"]
    assert thr["by_th"]["7"]["value"] >= 15000 and thr["batch_frames"] >= 64 and thr["launches_per_frame"] < 2
