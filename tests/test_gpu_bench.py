"""bench.py on the GPU box: the N>1 launch path through the real library (two ranks folded onto the one device
of the box with FT_BENCH_DEVICE_MOD=1), the contract fields of the JSON line, and the host-in leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_folded_onto_one_device():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16", "--workload", "stereo_640x480_nf1000",
              "--no-cpu-baseline"], {"FT_BENCH_DEVICE_MOD": "1"})
    assert r["workloads"] is None  # N = 1 only
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 3
    assert len(r["per_rank_frames_per_s"]) == 2 and min(r["per_rank_frames_per_s"]) > 0
    # whole-job value = frames of both ranks / max-over-ranks time: never above the sum of the per-rank rates
    assert r["value"] <= sum(r["per_rank_frames_per_s"]) * 1.0001
    assert r["config"]["distinct_pairs"] == r["config"]["batch_pairs_per_gpu"] == 16
    assert r["host_in"]["value"] > 0 and r["cpu_baseline"] is None
    assert r["device_octree_fallbacks"] == 0


def test_bench_eight_ranks_folded_onto_one_device():
    """the shape of the driver's 8-GPU run on the one device of the box: eight processes, eight contexts with their lane
    tables and thread pools (cpus / 8 each), gloo rendezvous, NUMA pinning of every rank, the cpu_baseline leg on rank 0
    after the last barrier - so that the first real 8-GPU run does not die on ports, pinned-memory limits or queue counts"""
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--batch", "24", "--workload", "stereo_752x480_nf1200", "--no-host-in"],
             {"FT_BENCH_DEVICE_MOD": "1"}, timeout=1500)
    assert r["n_gpus"] == 8 and r["scaling"] == "weak" and len(r["per_rank_frames_per_s"]) == 8 and min(r["per_rank_frames_per_s"]) > 0
    assert r["value"] <= sum(r["per_rank_frames_per_s"]) * 1.0001 and r["device_octree_fallbacks"] == 0
    assert r["workloads"] is None and r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["kind"] == "port"
    assert r["config"]["host_threads_per_gpu"] >= 1 and r["config"]["hw_queues"] == 10


def test_bench_four_ranks_folded_onto_one_device():
    """the N = 4 point of the driver's scaling curve, folded: whole-job value over the slowest rank's time, per-rank rates, NUMA
    fields, no data-path collective"""
    r = _run(["--gpus", "4", "--steps", "3", "--warmup", "1", "--batch", "16", "--workload", "stereo_640x480_nf1000", "--no-host-in",
              "--no-cpu-baseline"], {"FT_BENCH_DEVICE_MOD": "1"}, timeout=600)
    assert r["n_gpus"] == 4 and r["scaling"] == "weak" and len(r["per_rank_frames_per_s"]) == 4 and min(r["per_rank_frames_per_s"]) > 0
    assert r["value"] <= sum(r["per_rank_frames_per_s"]) * 1.0001 and r["device_octree_fallbacks"] == 0
    assert "numa_node_of_rank0" in r["config"] and "no collective" in r["config"]["parallelism"]


def test_bench_single_rank_line_has_the_contract_fields():
    r = _run(["--steps", "3", "--warmup", "1", "--batch", "32", "--workload", "stereo_752x480_nf1200", "--no-cpu-baseline",
              "--workload-batch", "20", "--workload-frames", "6", "--tracking-batch", "24"])
    assert r["n_gpus"] == 1 and r["metric"] == "frames/sec extract+match" and r["unit"] == "frames/s"
    assert r["dtype"] == "u8" and r["vs_baseline"] is None and r["higher_is_better"] is True
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["kernel"] == "k_fast_cells" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert r["config"]["distinct_pairs"] == 32
    assert 0 < r["host_in"]["value"] <= r["value"] * 1.05   # uploading inside the timed region cannot be faster
    # the other north_star workloads ride on the same line (N = 1): 752x480 stereo, configs[3] tracking, dense and planes scenes
    wl = r["workloads"]
    assert set(wl) == {"stereo_752x480_nf1200", "tracking_512x512_nf2000", "dense_1280x720_nf2000", "planes_1280x720_nf2000"}
    for k in ("stereo_752x480_nf1200", "dense_1280x720_nf2000", "planes_1280x720_nf2000"):
        assert wl[k]["value"] > 0 and wl[k]["device_octree_fallbacks"] == 0 and wl[k]["keypoints_per_frame"] > 1000, k
    assert wl["planes_1280x720_nf2000"]["stereo_match_fraction"] > 0.3 > r["stereo_match_fraction"] > 0
    t = wl["tracking_512x512_nf2000"]
    assert set(t["by_th"]) == {"7", "15"} and t["value"] == t["by_th"]["7"]["value"] > 0
    for th in ("7", "15"):
        assert t["by_th"][th]["map_points_per_s"] > 0 and t["by_th"][th]["hamming_compares_per_frame"] > 1000 and t["by_th"][th]["matches_per_frame"] > 50
    assert t["by_th"]["15"]["hamming_compares_per_frame"] > t["by_th"]["7"]["hamming_compares_per_frame"]  # wider windows
    # ... every stereo leg with the roofline of its dominant kernel
    for k in ("stereo_752x480_nf1200", "dense_1280x720_nf2000", "planes_1280x720_nf2000"):
        ro = wl[k]["roofline"]
        assert ro["bound"] == "hbm" and ro["kernel"] == "k_fast_cells" and ro["peak"] == 8000.0 and 0 < ro["frac"] < 1
        assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12 and ro["avg_launch_ms"] > 0
    # configs[3] in its throughput form: B frames per launch (ft_tracked_batch), its kernels' event times and its roofline
    thr = t["throughput"]
    assert thr["batch_frames"] == thr["distinct_frames"] == 24 and set(thr["by_th"]) == {"7", "15"}
    assert thr["value"] == thr["by_th"]["7"]["value"] > t["value"]          # one launch set for 24 frames beats a frame at a time
    assert thr["launches_per_frame"] < 4
    ro = thr["roofline"]
    assert ro["bound"] == "valu" and "k_fisheye_2nn_batch" in ro["kernel"] and 0 < ro["frac"] < 1 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert thr["hamming_compares_per_frame"]["fisheye_2nn"] > 1e6 and thr["hamming_compares_per_frame"]["searches(first passes)"] > 1e4
    for kname in ("search_last_batch(first pass)", "search_local_batch(first pass)", "lap_gather+fisheye_2nn_batch", "cache_partition_batch",
                  "resolve_batch(last frame)", "resolve_batch(local map)"):
        assert thr["kernels"][kname]["avg_launch_ms"] > 0, kname
    # the reference's real call shape: one stereo pair in, results out
    lat = r["latency"]
    for shape in ("752x480_nf1200", "1280x720_nf2000"):
        assert 0 < lat[shape]["pinned_frames_ms"] < 20 and 0 < lat[shape]["pageable_frames_ms"] < 20
    # the line names the library it ran and says whether the committed profile artefacts belong to it
    assert "csrc:" in r["library"] and set(r["profiles"]) == {"r06_traffic.json", "r06_marginal_costs.json"}
    for state in r["profiles"].values():
        assert state == "current" or state == "missing" or state.startswith("stale")
    if not all(v == "current" for v in r["profiles"].values()):
        assert roof["traffic"] is None or r["profiles"]["r06_traffic.json"] == "current"


def test_tracking_leg_with_one_host_thread_per_lane_on_two_cpus():
    """the batched searches without the host in them.  configs[3]'s throughput leg (64 frames per launch, four
    lanes) with ONE host thread per lane, the point arrays in pinned memory (read in place), the process pinned to two CPUs
    before the context exists: the host's share of a search call (staging + what is left of the replay) stays a fraction of a millisecond (1.1 - 1.5 ms in round 5), and
    the two-CPU process keeps at least half of what the same box does with all its CPUs and three threads per lane (boxes differ
    by 15 %, so the bound is relative)."""
    import json, subprocess, sys
    tool = os.path.join(ROOT, "tests", "tools", "bench_tracking_batch.py")

    def run(env, *args):
        out = subprocess.run([sys.executable, tool, *args], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().split("\n")[-1])
    free = run({}, "64", "8", "4", "1", "0")
    two = run({"FT_BENCH_CPUS": "0,1"}, "64", "8", "4", "1", "1")
    # the host's share is WORK, measured where a thread is not waiting for a CPU: with all CPUs 0.06 - 0.2 ms per call (bound 0.5; 1.1 - 1.5 in round 5); on two CPUs
    # (four lane threads and the context's workers take turns on them, so a wall-clock section includes being descheduled) under 1.5 ms
    for name, res, bound in (("all CPUs", free, 0.5), ("two CPUs", two, 1.5)):
        lib = res["by_th"]["7"]["inside_the_library"]
        for call in ("search_last_frame", "track_local_map"):
            host = lib[f"tracked_batch.{call}.stage_ms_per_call"] + lib[f"tracked_batch.{call}.replay_ms_per_call"]
            assert host < bound, (name, call, host)
    assert two["host_threads"] == 4 and two["by_th"]["7"]["value"] > 0.5 * free["by_th"]["7"]["value"], (two["by_th"]["7"]["value"], free["by_th"]["7"]["value"])
