#!/usr/bin/env python3
"""bench.py - frames/s of the tracking front end hot path (extract + stereo match) on MI355X.

A "step" is one pass of the hot path over one batch of synthetic rectified stereo pairs that are
already resident in HBM: ORB extraction of the left and right image (pyramid, FAST-9 + NMS per cell,
octree distribution, orientation, blur, rBRIEF) followed by Frame::ComputeStereoMatches, with the
keypoints / descriptors / mvuRight / mvDepth delivered to host arrays as the reference's callers
expect.  Default workload = BASELINE.json config 5's per-GPU shard: one 1280x720 stereo stream,
nFeatures 2000, 8 levels, scale 1.2, FAST 20/7.  One process per GPU; streams are independent, so N
GPUs run N shards with no collective on the data path (weak scaling); torch.distributed (gloo) is
used only for the barrier and the max-over-ranks time.

`python bench.py --gpus N` without a torch.distributed.run environment starts the N ranks itself (a child
`python -m torch.distributed.run`, started before this process touches the GPU; never an exec); under the
launcher `--gpus` must equal WORLD_SIZE.

Inputs: every pair of a step is a DISTINCT frame (`distinct_pairs == batch_pairs_per_gpu` by default), so
level 0 streams from HBM: `--scenes` seeded synthetic scenes, each presented at distinct cyclic (dx, dy)
shifts - the same shift for the left and the right image, which keeps the pair rectified.

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline      dominant kernel (FAST cells): algorithmic bytes per launch / HIP-event duration vs 8 TB/s
  cpu_baseline  the oracle (restated CPU path of the reference) timed on this box's host cores (rank 0, after the last barrier)
  host_in       the same workload with the frames in pinned host memory (H2D inside the timed region)
  workloads     (N = 1) the other workloads BASELINE.json's north_star names, a few steps each after the headline region:
                stereo_752x480_nf1200 (configs[2]), tracking_512x512_nf2000 (configs[3]: fisheye extraction with the YAML's
                lapping area + both SearchByProjection searches per frame, latency mode), dense_1280x720_nf2000 (mosaic
                frames with ~10 k FAST survivors at level 0, SURVEY a4's range) and planes_1280x720_nf2000 (60 % of the
                left keypoints find a stereo partner) - each with its own cpu_baseline sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fasttrack_amd import shard, synth  # noqa: E402  (orb - the HIP library - is imported in main(), after the launcher decision)

WORKLOADS = {
    # name: (width, height, nfeatures, BASELINE.json config it mirrors)
    "stereo_1280x720_nf2000": (1280, 720, 2000, "configs[4] per-GPU shard: synthetic 1280x720 stereo stream, nFeatures=2000"),
    "stereo_752x480_nf1200": (752, 480, 1200, "configs[2] shape: EuRoC-like 752x480 stereo, extract + ComputeStereoMatches"),
    "stereo_640x480_nf1000": (640, 480, 1000, "configs[1] size with stereo matching"),
}
NLEVELS, SCALE, INI_TH, MIN_TH = 8, 1.2, 20, 7
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_JSON = "r06_traffic.json"
MARGINAL_JSON = "r06_marginal_costs.json"
TRACK_SQ_JSON = "r06_tracking_sq.json"   # tools/pmc_tracking_batch.sh: SQ instruction counts of the tracking leg's kernels per launch
VALU_MIX_JSON = "r06_valu_mix.json"   # tools/valu_mix.py: mean issue cycles per vector instruction of each kernel's stream
VALU_DEAR_CYCLES = 4.33               # profiles/r06_valu_rates.txt: the dear class (v_sad, v_pk_*, v_perm, v_dot*, v_cmp, ...) at 2.4 GHz
SIMDS, CLOCK_HZ = 1024, 2.4e9
_valu_mix = None


def valu_cycles(kernel):
    """mean cycles (at the nominal 2.4 GHz) a SIMD spends issuing one vector instruction of `kernel`: the opcode-weighted cost of its
    instruction stream by the issue-cost table measured on this chip (profiles/r06_valu_rates.txt -> tools/valu_mix.py).  Rounds 1 - 5
    priced every instruction at 4 cycles; the table has a ~2.3 - 3.2-cycle class (v_mov, 32-bit add / logic, f32) beside the ~4.3-cycle
    one the hot loops are built from.  A kernel the file does not hold is priced at the dear class."""
    global _valu_mix
    if _valu_mix is None:
        try:
            _valu_mix = json.load(open(os.path.join(ROOT, "profiles", VALU_MIX_JSON)))["kernels"]
        except Exception:
            _valu_mix = {}
    for k, v in _valu_mix.items():
        if k == kernel or k.split("<")[0] == kernel:
            return float(v["mean_cycles_per_valu"])
    return VALU_DEAR_CYCLES


def csrc_stamp(version: str) -> str:
    """the hash of fasttrack_amd/csrc a library was built from (ft_version: '... csrc:<hash>')"""
    return version.rsplit("csrc:", 1)[1].strip() if "csrc:" in version else ""


def load_profile(name, lib_stamp):
    """a committed profile artefact (profiles/<name>) - only if it was measured on THIS library: traffic per launch and marginal
    costs are properties of the kernels' code, so an artefact stamped with another csrc hash (or with none) yields {}"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}, "missing"
    if not lib_stamp or d.get("csrc") != lib_stamp:
        return {}, f"stale: profiled on csrc {d.get('csrc')}, this library is {lib_stamp}"
    return d, "current"


def usable_cpus():
    """CPUs this process may use: affinity, capped by a cgroup v2 CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        pass
    return max(n, 1)


def level_pixels(w, h):
    P = []
    sf = np.float32(1.0)
    for l in range(NLEVELS):
        inv = np.float32(1.0) / sf
        lw = int(np.rint(np.float32(w) * inv))
        lh = int(np.rint(np.float32(h) * inv))
        P.append(lw * lh)
        sf = np.float32(sf * np.float32(SCALE))
    return P


def cpu_baseline(w, h, nf, pairs, budget_s=12.0):
    """Oracle (oracle/, kind 'port') on host cores: extract L + R + ComputeStereoMatches per pair.
    One worker per pair across all cores (ctypes releases the GIL); bounded by ~budget_s of wall time.
    Also times the reference's own threading (left/right extraction on two threads, src/Frame.cc:127-130)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    intr = synth.intrinsics(w, h)
    cores = usable_cpus()

    def one(pair):
        L, R = pair
        exL, exR = ob.Extractor(nf, SCALE, NLEVELS, INI_TH, MIN_TH), ob.Extractor(nf, SCALE, NLEVELS, INI_TH, MIN_TH)
        kL, dL, _ = exL.extract(L)
        kR, dR, _ = exR.extract(R)
        ob.stereo_match(exL, exR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
        return len(kL) + len(kR)

    t0 = time.perf_counter()
    one(pairs[0])
    t1 = time.perf_counter() - t0
    # reference threading: two threads for the two extractions of ONE pair
    exL, exR = ob.Extractor(nf, SCALE, NLEVELS, INI_TH, MIN_TH), ob.Extractor(nf, SCALE, NLEVELS, INI_TH, MIN_TH)
    n_ref = max(2, min(16, int(4.0 / max(t1, 1e-3))))
    with ThreadPoolExecutor(2) as tp:
        t0 = time.perf_counter()
        for i in range(n_ref):
            L, R = pairs[i % len(pairs)]
            fa, fb = tp.submit(exL.extract, L), tp.submit(exR.extract, R)
            (kL, dL, _), (kR, dR, _) = fa.result(), fb.result()
            ob.stereo_match(exL, exR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
        ref_fps = n_ref / (time.perf_counter() - t0)
    # all cores, one pair per worker; rounds of `cores` pairs until ~budget_s of wall time is spent
    n_all, dt = 0, 0.0
    with ThreadPoolExecutor(cores) as tp:
        t0 = time.perf_counter()
        while True:
            list(tp.map(one, [pairs[(n_all + i) % len(pairs)] for i in range(cores)]))
            n_all += cores
            dt = time.perf_counter() - t0
            if dt >= budget_s or n_all >= 4096:
                break
    return {"value": n_all / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_all} synthetic {w}x{h} stereo pairs (nFeatures {nf}), oracle extract L+R + stereo match, "
                      f"{cores} worker threads one pair each, {dt:.1f} s; reference threading (2 threads per pair, "
                      f"matcher single-threaded): {ref_fps:.2f} frames/s over {n_ref} pairs"}


def spawn_ranks(n):
    """`bench.py --gpus N` outside a launcher: start N ranks as a child job (this process has not touched the GPU
    and does not exec), relay its output and exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_stream(alloc, w, h, rank, D, scenes, density=1.0, mosaic=0, planes=False):
    """D distinct stereo pairs of this rank's stream in pinned host memory: `scenes` seeded scenes (synth.make_stereo_pair),
    scene s of variant k shifted cyclically by (53 k mod w, 29 k mod h) px in BOTH images (rectification and disparities
    are kept; the wrap-around seam is one more edge).  alloc(shape, dtype) provides the (pinned) arrays.  Returns (hostL,
    hostR) of shape (D, h, w) and the seeded scenes."""
    S = max(1, min(scenes, D))
    if mosaic:
        base = [synth.make_mosaic_pair(w, h, seed=s, block=mosaic) for s in shard.stream_seeds(rank, S)]
    elif planes:
        base = [synth.make_planes_pair(w, h, seed=s, density=density) for s in shard.stream_seeds(rank, S)]
    else:
        base = [synth.make_stereo_pair(w, h, seed=s, density=density) for s in shard.stream_seeds(rank, S)]
    hostL, hostR = alloc((D, h, w), np.uint8), alloc((D, h, w), np.uint8)
    for d in range(D):
        L, R = base[d % S]
        k = d // S
        dx, dy = (53 * k) % w, (29 * k) % h
        if k == 0:
            hostL[d], hostR[d] = L, R
        else:
            hostL[d] = np.roll(L, (dy, dx), (0, 1))
            hostR[d] = np.roll(R, (dy, dx), (0, 1))
    return hostL, hostR, base


def stereo_leg(orb, ctx, name, w, h, nf, B, steps, warmup, mosaic=0, planes=False, scenes=8, cpu_budget_s=4.0, cpu=True):
    """One of the extra stereo workloads: the headline pipeline (two front ends used alternately, frames resident in HBM,
    results to host arrays) on another shape or scene kind, a few steps; buffers are released afterwards."""
    import ctypes as C
    intr = synth.intrinsics(w, h)
    fes = [orb.StereoFrontend(ctx, nf, SCALE, NLEVELS, INI_TH, MIN_TH, w, h, B, intr["mbf"], intr["mb"]) for _ in range(2)]
    hostL, hostR, pairs = make_stream(lambda shape, dt: np.empty(shape, dt), w, h, 0, B, scenes, 1.0, mosaic, planes)
    devL, devR = ctx.to_device(hostL), ctx.to_device(hostR)
    fb = w * h
    pL = (C.c_void_p * B)(*[devL.ptr.value + b * fb for b in range(B)])
    pR = (C.c_void_p * B)(*[devR.ptr.value + b * fb for b in range(B)])

    done_at = []

    def run(n):
        del done_at[:]
        for k in range(n):
            if k >= 2:
                fes[k % 2].wait()
                done_at.append(time.perf_counter())
            fes[k % 2].submit_raw(pL, pR, B, True, w)
        for k in range(max(n - 2, 0), n):
            fes[k % 2].wait()
            done_at.append(time.perf_counter())
    f0 = ctx.get_stat("stereo.device_octree_fallbacks")[1]
    # warm-up: at least `warmup` steps (the first dense batches size the octree's histogram tier) AND at least 0.4 s - a leg
    # starts behind CPU-side work (scene generation, the previous leg's CPU baseline), and a leg of 0.2 s measured on a GPU that
    # is still ramping its clocks up read 20 % low (122 k against 154 k frames/s for the 752x480 leg in two otherwise equal runs)
    run(max(warmup, 2))
    tw0 = time.perf_counter()
    while time.perf_counter() - tw0 < 0.4:
        run(4)
    f1 = ctx.get_stat("stereo.device_octree_fallbacks")[1]
    ctx.synchronize()
    t0 = time.perf_counter()
    run(steps)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    # the leg's dominant kernel against its roofline: a short run of its own with HIP events on the launching streams
    ctx.reset_stats()
    ctx.set_kernel_timing(True)
    run(4)
    ctx.synchronize()
    ctx.set_kernel_timing(False)
    roof = None
    try:
        tot_, n_ = ctx.get_stat("kernel.fast_cells")
        if n_:
            bytes_per_launch = 4.0 * 2.0 * B * sum(level_pixels(w, h)) / n_   # 4 steps x 2 B images, every pyramid pixel read once
            ach = bytes_per_launch / (tot_ / n_ / 1e3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_fast_cells", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": None, "bytes_per_launch": bytes_per_launch, "avg_launch_ms": tot_ / n_, "launches_timed": n_,
                    "note": "algorithmic bytes (every pyramid pixel once) / HIP-event duration on the launching stream; the duration is the "
                            "kernel's residency beside the other lanes' kernels"}
    except Exception:
        pass
    fe = fes[(steps - 1) % 2]
    kps = int(fe._nL[:B].sum() + fe._nR[:B].sum())
    nL = int(fe._nL[:B].sum())
    matches = int(fe._nm[:B].sum())
    a_, b_ = len(done_at) // 4, len(done_at) - 1 - len(done_at) // 8
    out = {"value": B * steps / dt, "unit": "frames/s", "metric": "frames/sec extract+match", "steps": steps, "warmup": max(warmup, 2),
           "steady_state_frames_per_s": B * (b_ - a_) / (done_at[b_] - done_at[a_]) if b_ > a_ else None,
           "ms_per_step": 1e3 * dt / steps, "batch_pairs": B, "distinct_pairs": B, "image": [w, h], "nfeatures": nf,
           "scene_kind": f"mosaic of {mosaic}-px tiles" if mosaic else "object scene on fronto-parallel planes" if planes else "objects on a smooth background",
           "inputs": "resident in HBM before the timed region", "keypoints_per_s": kps * steps / dt, "keypoints_per_frame": kps / B,
           "stereo_matches_per_frame": matches / B, "stereo_match_fraction": matches / max(nL, 1),
           "device_octree_fallbacks": ctx.get_stat("stereo.device_octree_fallbacks")[1] - f1,
           "device_octree_fallbacks_during_warmup": f1 - f0, "roofline": roof,
           "pipeline_hbm_read_frac": (B * steps / dt) * 2 * (3 * sum(level_pixels(w, h)) - level_pixels(w, h)[-1]) / (HBM_PEAK_GBS * 1e9)}
    for f in fes:
        f.close()
    devL.free()
    devR.free()
    if cpu:
        out["cpu_baseline"] = cpu_baseline(w, h, nf, pairs, budget_s=cpu_budget_s)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def latency_leg(orb, ctx, reps=40):
    """The reference's real call shape: ONE stereo pair in, results out (Frame::Frame, src/Frame.cc:102-147) - host images in
    pageable and in pinned memory, keypoints / descriptors / mvuRight / mvDepth in host arrays on return; median of `reps` calls
    through the latency-mode front end (one captured HIP graph per call shape)."""
    out = {"metric": "latency of one stereo pair, extract + ComputeStereoMatches", "unit": "ms", "statistic": f"median of {reps} calls"}
    for (w, h, nf) in ((752, 480, 1200), (1280, 720, 2000)):
        intr = synth.intrinsics(w, h)
        L, R = synth.make_stereo_pair(w, h, 5)
        fe = orb.StereoFrontend(ctx, nf, SCALE, NLEVELS, INI_TH, MIN_TH, w, h, 1, intr["mbf"], intr["mb"])
        Lp, Rp = ctx.pinned_array(L.shape, np.uint8), ctx.pinned_array(R.shape, np.uint8)
        Lp[:], Rp[:] = L, R

        def med(fn):
            for _ in range(5):
                fn()
            t = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                t.append(time.perf_counter() - t0)
            return 1e3 * float(np.median(t))
        out[f"{w}x{h}_nf{nf}"] = {"pageable_frames_ms": med(lambda: fe.process([L], [R])), "pinned_frames_ms": med(lambda: fe.process([Lp], [Rp]))}
        fe.close()
    return out


def tracking_leg(orb, ctx, frames=48, warmup=4, M=2000, cpu_budget_s=4.0, cpu=True):
    """BASELINE.json configs[3] (TUM-VI stereo-inertial fisheye, 512x512, nFeatures 2000, Examples/Stereo-Inertial/TUM-VI.yaml:
    45-53,86): per frame, as a SLAM front end calls it - one frame at a time, host images in, host results out -
      extraction of the left and the right image with the YAML's lapping areas [0, 511]  (ORBextractor::operator(), Frame.cc:1144)
      left <-> right descriptor matching of the lapping subsets                            (ComputeStereoFishEyeMatches, Frame.cc:1231-1271)
      SearchByProjection(CurrentFrame, LastFrame, th)                                       (ORBmatcher.cc:1775-1990; th 7 for stereo, 15 otherwise, Tracking.cc:2941-2946)
      isInFrustum + SearchByProjection(CurrentFrame, local map points, th)                  (Frame.cc:536-610, ORBmatcher.cc:49-225)
    on a two-camera KannalaBrandt8 frame resident in HBM (ft_tracked_frame_*), M local map points and one last-frame point
    per left keypoint, for th in {7, 15}.  Reported: frames/s, map points/s (points handed to the two searches) and Hamming
    compares/s (keypoints returned by GetFeaturesInArea for the searched points' windows, both cameras, counted once outside
    the timed region through ft_features_in_area with the searches' own radii and level ranges)."""
    from fasttrack_amd import scenarios as sc
    w, h, nf, D = 512, 512, 2000, 8
    lap = (0, 511)
    cam = list(sc.KB8_CAM)
    intr = dict(fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
    Trl = np.concatenate([np.eye(3), [[-0.101], [0.0], [0.0]]], 1).astype(np.float32)  # x_r = x_l + trl: 10.1 cm baseline (TUM-VI)
    TLR = (0.101, 0.0, 0.0)
    LOG_SF = float(np.float32(np.log(np.float32(SCALE))))
    ex2 = orb.ORBextractor(ctx, nf, SCALE, NLEVELS, INI_TH, MIN_TH, w, h, max_batch=2)
    exL = ex2
    sf = np.asarray(exL.GetScaleFactors(), np.float32)
    pairs = [synth.make_planes_pair(w, h, seed=7000 + i) for i in range(D)]
    # Round 6: the frame goes through the batch API with ONE frame (ft_tracked_batch_*): the keypoints stay where the extractor
    # left them (bind_fisheye: lapping-area order, left <-> right matching and grids on the device - no fisheye-match call, no
    # upload of the frame), the point arrays sit in pinned memory, the searches' writes are replayed on the device.
    tb = orb.TrackedBatch(ctx, max_frames=1, max_keypoints=2 * exL.max_keypoints + 64, max_points=max(M, exL.max_keypoints) + 64, pinned=True)

    part = {"extract_left_right": 0.0, "bind_fisheye (order, match, grids)": 0.0, "search_last_frame": 0.0, "track_local_map": 0.0}

    def clock(name, t0):
        t1 = time.perf_counter()
        part[name] += t1 - t0
        return t1

    def extract(i):
        # both cameras in ONE call (a batch of two through the captured graph), as Frame's two extraction threads overlap them
        # in the reference (Frame.cc:1144-1147); the lapping areas of the two cameras are the same in TUM-VI.yaml
        return ex2.extract_batch([pairs[i][0], pairs[i][1]], lap)

    def host_view(kL, dL, kR, dR):
        """the frame as a host FrameView (fisheye matching through the one-shot entry point): count_compares' queries only"""
        m = orb.KernelController.launchFisheyeStereoMatchKernel(ctx, dL, dR)
        l2r = np.ascontiguousarray(m["matches"], np.int32)
        r2l = np.full(len(kR), -1, np.int32)
        ok = l2r >= 0
        r2l[l2r[ok]] = np.nonzero(ok)[0].astype(np.int32)
        return orb.FrameView(keys=kL, keys_right=kR, descriptors=np.concatenate([dL, dR]), scale_factors=sf, bounds=sc.frame_bounds(w, h),
                             left_to_right=l2r, right_to_left=r2l, cam_model=1, cam=cam, Trl=Trl)
    # per distinct frame: the last frame's points and the local map, built from the frame's own keypoints; the frame
    # constants and the marshalled point arrays (pinned) once - extraction is deterministic, the counts do not change
    scen, metas, pl_last, pl_local, hostF = [], [], [], [], []
    for i in range(D):
        (kL, dL, _), (kR, dR, _) = extract(i)
        depth = np.zeros(len(kL), np.float32)
        last, Tcw_last = sc.last_frame_scenario(kL, dL, None, depth, intr, w, h, seed=40 + i)
        pts, Rcw, tcw = sc.map_points_scenario(kL, dL, depth, intr, NLEVELS, sf, 90 + i, M=M)
        scen.append((last, Tcw_last, pts, Rcw, tcw))
        metas.append(tb.prepare_frames([orb.FrameView(keys=kL, keys_right=kR, descriptors=np.zeros((len(kL) + len(kR), 32), np.uint8), scale_factors=sf,
                                                     bounds=sc.frame_bounds(w, h), left_to_right=np.zeros(max(len(kL), 1), np.int32),
                                                     right_to_left=np.zeros(max(len(kR), 1), np.int32), cam_model=1, cam=cam, Trl=Trl)]))
        pl_last.append(tb.prepare_last([last], [Tcw_last], ctx=ctx))
        pl_local.append(tb.prepare_local([orb.make_pose(Rcw, tcw, TLR)], [pts], ctx=ctx))
        hostF.append(host_view(kL, dL, kR, dR))

    def frame(i, th):
        t = time.perf_counter()
        (kL, dL, _), (kR, dR, _) = extract(i)
        t = clock("extract_left_right", t)
        tb.bind_fisheye(ex2, ex2, metas[i], lap, lap, slot0=0, slot0_right=1, want_tables=False)
        t = clock("bind_fisheye (order, match, grids)", t)
        a = tb.search_last_frame(pl_last[i], th=th)[0]
        t = clock("search_last_frame", t)
        b = tb.track_local_map(pl_local[i], viewing_cos_limit=0.5, log_scale_factor=LOG_SF, th=th)[0]
        clock("track_local_map", t)
        return len(kL), len(kR), a, b, hostF[i]
    out = {"metric": "frames/sec extract + SearchByProjection (last frame, local map)", "unit": "frames/s", "image": [w, h], "nfeatures": nf,
           "camera": "KannalaBrandt8 stereo rig, lapping areas [0, 511]", "local_map_points": M, "mode": "one frame at a time (latency mode) through ft_tracked_batch_* with one frame, host images in, host results out",
           "scene_kind": "object scene on fronto-parallel planes", "distinct_frames": D, "by_th": {}}
    for th in (7.0, 15.0):
        tw0 = time.perf_counter()
        k = 0
        while k < warmup or time.perf_counter() - tw0 < 0.3:  # (clocks: see stereo_leg)
            frame(k % D, th)
            k += 1
        ctx.synchronize()
        npts = ncmp = nmatch = 0
        ctx.reset_stats()
        for k_ in part:
            part[k_] = 0.0
        t0 = time.perf_counter()
        for k in range(frames):
            nl, nr, a, b, F = frame(k % D, th)
            npts += len(scen[k % D][0]["valid"]) + M
            nmatch += a["n"] + b["n"]
        ctx.synchronize()
        dt = time.perf_counter() - t0
        parts_ms = {k_: 1e3 * v_ / frames for k_, v_ in part.items()}
        # passes of the exact in-call claiming (EXPERIMENTS 3.2: one launch per pass, as many as the longest chain of map points
        # that take a keypoint from one another) and the time inside the two C entry points
        lib_stats = {}
        for nm_ in ("tracked_batch.search_last_frame.passes", "tracked_batch.track_local_map.passes", "tracked_batch.search_last_frame.total",
                    "tracked_batch.track_local_map.total"):
            tot_, n_ = ctx.get_stat(nm_)
            lib_stats[nm_ + ("_per_call" if nm_.endswith("passes") else "_ms_per_call")] = tot_ / n_ if n_ else None
        # Hamming compares of one pass over the distinct frames, counted outside the timed region
        for i in range(D):
            nl, nr, a, b, F = frame(i, th)
            ncmp += count_compares(orb, ctx, F, sf, scen[i], b, th, cam, Trl)
        out["by_th"][str(int(th))] = {"value": frames / dt, "ms_per_frame": 1e3 * dt / frames, "map_points_per_s": npts / dt,
                                      "hamming_compares_per_frame": ncmp / D, "hamming_compares_per_s": ncmp / D * frames / dt,
                                      "matches_per_frame": nmatch / frames, "ms_per_frame_by_part": parts_ms, "inside_the_library": lib_stats}
    out["value"] = out["by_th"]["7"]["value"]
    out["keypoints_per_frame"] = nl + nr
    if cpu:
        out["cpu_baseline"] = tracking_cpu_baseline(pairs, scen, cam, Trl, sf, lap, w, h, nf, LOG_SF, cpu_budget_s)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    tb.close()
    ex2.close()
    return out


def tracking_batch_leg(orb, ctx, B=128, steps=6, warmup=2, M=2000, ths=(7.0, 15.0), pipelined=True, in_flight=4, pinned=True,
                       one_thread_per_lane=False, overlap=True):
    """BASELINE.json configs[3] as a THROUGHPUT workload: B independent 512x512 KannalaBrandt8 stereo frames per step (the
    frames B camera streams deliver for one time step), every stage ONE launch over all of them (ft_tracked_batch_*):
      extraction of the B left and the B right images, frames resident in HBM, lapping areas [0, 511]      (ft_extract_batch x 2,
                                                                             two host threads as in Frame's constructor)
      keypoints into the reference's order + left <-> right matching of the lapping subsets + grids, on the device,
      from what the extractors left in HBM                                                                   (ft_tracked_batch_bind_fisheye)
      SearchByProjection(CurrentFrame, LastFrame, th)                                                       (ft_tracked_batch_search_last_frame)
      isInFrustum + SearchByProjection(CurrentFrame, local map points, th)                                   (ft_tracked_batch_track_local_map)
    pipelined: `in_flight` steps in flight, like the two stereo front ends of the headline - that many host threads, each with its
    own pair of extractors and its own batch object (a batch has a stream of its own), take the steps in turn, so that the kernels
    of one step run beside the extraction and the host side of the others.  Host in / out per step: the map points of every frame up
    (pinned = True: arrays in pinned host memory, read in place by the device - the host stages nothing and the writes of a search
    are replayed on the device; False: pageable arrays, packed into pinned staging by the context's host threads), keypoints,
    descriptors, assignments, match counts and frustum fields down.  one_thread_per_lane: the lane's thread runs the two
    extractions itself, one after the other (otherwise two helper threads per lane, as Frame's constructor has them).  overlap: a lane
    enqueues the extraction of its next step beside the local-map search of its current one (the searches in their submit / wait
    form; + 5 - 9 % on one box, EXPERIMENTS 11.15).  Every frame of the
    batch is a distinct image pair with its own last-frame points, local map (M points) and poses, built once from the frame's
    own keypoints (untimed)."""
    import ctypes as C
    import concurrent.futures
    from fasttrack_amd import scenarios as sc
    w, h, nf = 512, 512, 2000
    lap = (0, 511)
    cam = list(sc.KB8_CAM)
    intr = dict(fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
    Trl = np.concatenate([np.eye(3), [[-0.101], [0.0], [0.0]]], 1).astype(np.float32)
    TLR = (0.101, 0.0, 0.0)
    LOG_SF = float(np.float32(np.log(np.float32(SCALE))))
    nlanes = max(in_flight, 1) if pipelined else 1
    pairs = [synth.make_planes_pair(w, h, seed=7000 + i) for i in range(B)]
    devL, devR = ctx.to_device(np.stack([p[0] for p in pairs])), ctx.to_device(np.stack([p[1] for p in pairs]))
    fb = w * h
    pL = (C.c_void_p * B)(*[devL.ptr.value + b * fb for b in range(B)])
    pR = (C.c_void_p * B)(*[devR.ptr.value + b * fb for b in range(B)])

    class Lane:
        pass
    lanes = []
    for _ in range(nlanes):
        ln = Lane()
        ln.exL = orb.ORBextractor(ctx, nf, SCALE, NLEVELS, INI_TH, MIN_TH, w, h, max_batch=B)
        ln.exR = orb.ORBextractor(ctx, nf, SCALE, NLEVELS, INI_TH, MIN_TH, w, h, max_batch=B)
        cap = ln.exL.max_keypoints
        # (pinned: the device writes the keypoints and descriptors of a step into these arrays itself, in the reference's output order)
        alloc = ctx.pinned_array if pinned else (lambda shape, dt: np.zeros(shape, dt))
        ln.kL, ln.kR = alloc((B, cap), orb.KP_DTYPE), alloc((B, cap), orb.KP_DTYPE)
        ln.dL, ln.dR = alloc((B, cap, 32), np.uint8), alloc((B, cap, 32), np.uint8)
        ln.nL, ln.nR, ln.mL, ln.mR = (np.zeros(B, np.int32) for _ in range(4))
        lanes.append(ln)
    sf = np.asarray(lanes[0].exL.GetScaleFactors(), np.float32)
    cap = lanes[0].exL.max_keypoints
    for ln in lanes:
        ln.tb = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=2 * cap + 64, max_points=max(M, cap) + 64, pinned=pinned)
        ln.part = {"extract_left_right": 0.0, "bind_fisheye (order, match, grids)": 0.0, "search_last_frame": 0.0, "track_local_map": 0.0}
    tb = lanes[0].tb
    two = concurrent.futures.ThreadPoolExecutor(2 * nlanes)
    ahead = concurrent.futures.ThreadPoolExecutor(nlanes)

    def extract(ln):
        if one_thread_per_lane:
            ln.exL.extract_batch_into(pL, B, True, w, h, w, lap, ln.kL, ln.dL, ln.nL, ln.mL)
            ln.exR.extract_batch_into(pR, B, True, w, h, w, lap, ln.kR, ln.dR, ln.nR, ln.mR)
            return
        # the two cameras on two host threads, as Frame's constructor runs them (src/Frame.cc:1144-1147)
        a = two.submit(ln.exL.extract_batch_into, pL, B, True, w, h, w, lap, ln.kL, ln.dL, ln.nL, ln.mL)
        b = two.submit(ln.exR.extract_batch_into, pR, B, True, w, h, w, lap, ln.kR, ln.dR, ln.nR, ln.mR)
        a.result()
        b.result()
    for ln in lanes:
        extract(ln)
        # the frame views of the lane: constants + the host copies of the keypoints (extraction is deterministic: the counts and
        # the arrays the views point to are the same in every step)
        views = [orb.FrameView(keys=ln.kL[f, :ln.nL[f]], keys_right=ln.kR[f, :ln.nR[f]],
                               descriptors=np.zeros((int(ln.nL[f] + ln.nR[f]), 32), np.uint8), scale_factors=sf, bounds=sc.frame_bounds(w, h),
                               left_to_right=np.zeros(max(int(ln.nL[f]), 1), np.int32), right_to_left=np.zeros(max(int(ln.nR[f]), 1), np.int32),
                               cam_model=1, cam=cam, Trl=Trl) for f in range(B)]
        ln.meta = ln.tb.prepare_frames(views)
    l0 = lanes[0]
    nL, nR = l0.nL.copy(), l0.nR.copy()
    scen = []
    for f in range(B):
        depth = np.zeros(int(nL[f]), np.float32)
        last, Tcw_last = sc.last_frame_scenario(l0.kL[f, :nL[f]], l0.dL[f, :nL[f]], None, depth, intr, w, h, seed=40 + f)
        pts, Rcw, tcw = sc.map_points_scenario(l0.kL[f, :nL[f]], l0.dL[f, :nL[f]], depth, intr, NLEVELS, sf, 90 + f, M=M)
        scen.append((last, Tcw_last, pts, Rcw, tcw))
    pc = ctx if pinned else None
    pl_last = tb.prepare_last([s_[0] for s_ in scen], [s_[1] for s_ in scen], ctx=pc)
    pl_local = tb.prepare_local([orb.make_pose(s_[3], s_[4], TLR) for s_ in scen], [s_[2] for s_ in scen], ctx=pc)
    def pipeline(ln, n, th):
        for _ in range(n):
            t0 = time.perf_counter()
            extract(ln)
            t1 = time.perf_counter()
            ln.tb.bind_fisheye(ln.exL, ln.exR, ln.meta, lap, lap, want_tables=False)
            t2 = time.perf_counter()
            ln.tb.search_last_frame(pl_last, th=th, copy=False)
            t3 = time.perf_counter()
            ln.tb.track_local_map(pl_local, viewing_cos_limit=0.5, log_scale_factor=LOG_SF, th=th, copy=False)
            t4 = time.perf_counter()
            for k_, v_ in zip(ln.part, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                ln.part[k_] += v_

    def pipeline_overlapped(ln, n, th):
        """the same calls, the searches in their submit / wait form: the extraction of a lane's NEXT step is enqueued beside the
        local-map search of its current one (a frame's extraction needs nothing of the frame before it; the batch object holds the
        current step's keypoints by then - the bind has gathered them - so the extractors' slots are free)"""
        if n <= 0:
            return
        t0 = time.perf_counter()
        extract(ln)
        t1 = time.perf_counter()
        ln.part["extract_left_right"] += t1 - t0
        for k in range(n):
            t1 = time.perf_counter()
            ln.tb.bind_fisheye(ln.exL, ln.exR, ln.meta, lap, lap, want_tables=False)
            t2 = time.perf_counter()
            ln.tb.search_last_frame(pl_last, th=th, copy=False, submit=True)
            ln.tb.wait(copy=False)
            t3 = time.perf_counter()
            ln.tb.track_local_map(pl_local, viewing_cos_limit=0.5, log_scale_factor=LOG_SF, th=th, copy=False, submit=True)
            t4 = time.perf_counter()
            if k + 1 < n:
                extract(ln)
            t5 = time.perf_counter()
            ln.tb.wait(copy=False)
            t6 = time.perf_counter()
            for k_, v_ in zip(ln.part, (t5 - t4, t2 - t1, t3 - t2, (t4 - t3) + (t6 - t5))):
                ln.part[k_] += v_

    def run(n, th):
        """n steps, dealt to the lanes; every lane works through its steps on a host thread of its own"""
        fn = pipeline_overlapped if overlap else pipeline
        futs = [ahead.submit(fn, ln, n // nlanes + (1 if i < n % nlanes else 0), th) for i, ln in enumerate(lanes)]
        for f_ in futs:
            f_.result()
    out = {"metric": "frames/sec extract + SearchByProjection (last frame, local map), B frames per launch", "unit": "frames/s",
           "batch_frames": B, "distinct_frames": B, "steps": steps, "image": [w, h], "nfeatures": nf, "local_map_points": M,
           "mode": (f"{nlanes} steps in flight ({nlanes} host threads, each with its extractors and its batch)" if nlanes > 1 else "one step at a time") +
                   ("; a lane's next extraction enqueued beside its current local-map search" if overlap else ""),
           "host_threads": nlanes * (1 if one_thread_per_lane else 3),
           "inputs": "images resident in HBM before the timed region; map points in %s host memory, uploaded inside it" % ("pinned" if pinned else "pageable"),
           "outputs": "keypoints, descriptors, assignments, match counts, frustum fields in host memory", "by_th": {}}
    for th in ths:
        run(max(warmup, 2), th)
        for ln in lanes:
            assert np.array_equal(ln.nL, nL) and np.array_equal(ln.nR, nR)
        ctx.synchronize()
        ctx.reset_stats()
        for ln in lanes:
            for k_ in ln.part:
                ln.part[k_] = 0.0
        t0 = time.perf_counter()
        run(steps, th)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        stats = {}
        for nm_ in ("tracked_batch.search_last_frame.passes", "tracked_batch.track_local_map.passes", "tracked_batch.search_last_frame.total",
                    "tracked_batch.track_local_map.total", "tracked_batch.search_last_frame.stage", "tracked_batch.search_last_frame.device",
                    "tracked_batch.search_last_frame.replay", "tracked_batch.track_local_map.stage", "tracked_batch.track_local_map.device",
                    "tracked_batch.track_local_map.replay", "tracked_batch.bind_fisheye.total"):
            try:
                tot_, n_ = ctx.get_stat(nm_)
            except Exception:
                continue
            stats[nm_ + ("_per_call" if nm_.endswith("passes") else "_ms_per_call")] = tot_ / n_ if n_ else None
        nmatch = int(tb._nm.sum())
        part = {k_: sum(ln.part[k_] for ln in lanes) for k_ in lanes[0].part}   # host-thread time per step (the lanes' steps overlap)
        out["by_th"][str(int(th))] = {"value": B * steps / dt, "ms_per_step": 1e3 * dt / steps, "us_per_frame": 1e6 * dt / (B * steps),
                                      "map_points_per_s": (int(nL.sum()) + B * M) * steps / dt,
                                      "local_map_matches_per_frame": nmatch / B,
                                      "ms_per_step_by_part": {k_: 1e3 * v_ / steps for k_, v_ in part.items()},
                                      "inside_the_library": stats}
    out["value"] = out["by_th"][str(int(ths[0]))]["value"]
    out["keypoints_per_frame"] = float(nL.sum() + nR.sum()) / B
    # ---- what the kernels do with it: durations by HIP events on the launching stream (a run of its own, th = ths[0]), the
    # Hamming compares of a step, and the issue bound they are priced against ----
    th0 = ths[0]
    ctx.reset_stats()
    ctx.set_kernel_timing(True)
    pipeline(lanes[0], 2, th0)   # (one lane alone: an event pair brackets the kernels of ONE stream)
    ctx.synchronize()
    ctx.set_kernel_timing(False)
    kern = {}
    for nm_ in ("kernel.pyr_down(all levels)", "kernel.fast_cells", "kernel.compact", "kernel.octree", "kernel.orient_desc",
                "kernel.lap_gather+fisheye_2nn_batch", "kernel.build_grid_batch", "kernel.frustum_batch", "kernel.search_last_batch(first pass)",
                "kernel.search_last_batch(later pass)", "kernel.search_local_batch(first pass)", "kernel.search_local_batch(later pass)",
                "kernel.cache_partition_batch", "kernel.resolve_batch(last frame)", "kernel.resolve_batch(local map)",
                "kernel.replay_batch(last frame)", "kernel.replay_batch(local map)"):
        try:
            tot_, n_ = ctx.get_stat(nm_)
        except Exception:
            continue
        if n_:
            kern[nm_[7:]] = {"launches_per_step": n_ / 2.0, "avg_launch_ms": tot_ / n_, "ms_per_step": tot_ / 2.0}
    # compares of the searches = keypoints GetFeaturesInArea returns for the searched points' windows (count_compares), a sample of frames
    fr = tb.track_local_map(pl_local, viewing_cos_limit=0.5, log_scale_factor=LOG_SF, th=th0, copy=True)
    sample = list(range(0, B, max(B // 8, 1)))[:8]
    views0 = lanes[0].meta[1]
    cmp_search = float(np.mean([count_compares(orb, ctx, views0[f], sf, scen[f], fr[f], th0, cam, Trl) for f in sample]))
    cmp_2nn = float(np.mean(nL.astype(np.float64) * nR.astype(np.float64)))   # lapping areas cover the images: every left x every right keypoint
    fps = out["value"]
    # 256-bit Hamming distance = 8 v_xor_b32 + 8 v_bcnt_u32_b32 (accumulating) per lane.  Measured as an alternating pair on this chip
    # (profiles/r06_valu_rates.txt, "pair v_xor_b32 + v_bcnt_u32_b32"): 3.20 cycles at 2.4 GHz per instruction of the pair - the xor is
    # of the cheap class, the popcount of the dear one - so a wave's 64 compares cost 16 x 3.20 = 51.2 cycles of its SIMD
    XOR_BCNT_PAIR_CYCLES = 3.20
    VALU_COMPARES_PER_S = SIMDS * CLOCK_HZ * 64 / (16 * XOR_BCNT_PAIR_CYCLES)
    k2 = kern.get("lap_gather+fisheye_2nn_batch")
    out["kernels"] = kern
    out["hamming_compares_per_frame"] = {"fisheye_2nn": cmp_2nn, "searches(first passes)": cmp_search}
    out["hamming_compares_per_s"] = (cmp_2nn + cmp_search) * fps
    if k2:
        ach = cmp_2nn * B / (k2["avg_launch_ms"] / 1e3)
        out["roofline"] = {"bound": "valu", "kernel": "k_fisheye_2nn_batch (timed together with k_lap_gather_batch, ~10 % of the pair)",
                           "achieved": ach / 1e9, "peak": VALU_COMPARES_PER_S / 1e9, "unit": "G Hamming compares/s", "frac": ach / VALU_COMPARES_PER_S,
                           "traffic": None, "compares_per_launch": cmp_2nn * B, "avg_launch_ms": k2["avg_launch_ms"],
                           "peak_source": "16 vector instructions (8 v_xor_b32 + 8 v_bcnt_u32_b32) per 256-bit compare and lane at the measured "
                                          "3.20 cycles per instruction of the alternating pair (profiles/r06_valu_rates.txt), 1 024 SIMDs x 2.4 GHz "
                                          "= 3.07 T compares/s (rounds 1 - 5 priced both opcodes at 4 cycles: 2.46 T)"}
    # ---- the kernels that dominate the step (VERDICT r5 missing 6): the extraction's k_fast_cells against HBM (its algorithmic
    # bytes = the pixels of all pyramid levels of the launch's images) and k_resolve_batch against the vector issue of the CUs it
    # runs on (a workgroup = a CU per frame: instructions of the committed SQ pass x the opcode-weighted cycles of its stream)
    Pimg = sum(int(round(w / SCALE ** l)) * int(round(h / SCALE ** l)) for l in range(NLEVELS))
    kf = kern.get("fast_cells")
    out["roofline_by_kernel"] = {}
    if kf:
        ach = B * Pimg / (kf["avg_launch_ms"] / 1e3) / 1e9
        out["roofline_by_kernel"]["k_fast_cells (512x512, per camera)"] = {
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "bytes_per_launch": B * Pimg,
            "avg_launch_ms": kf["avg_launch_ms"], "launches_per_step": kf["launches_per_step"], "traffic": None}
    sq, sq_state = load_profile(TRACK_SQ_JSON, csrc_stamp(orb.version()))
    out["profiles"] = {TRACK_SQ_JSON: sq_state}
    if sq.get("batch_frames") == B:
        for nm_, kname, mixname in (("resolve_batch(last frame)", "k_resolve_batch<false, true>", "k_resolve_batch<false>"),
                                    ("resolve_batch(local map)", "k_resolve_batch<true, true>", "k_resolve_batch<true>"),
                                    ("fast_cells", "k_fast_cells<48, false>", "k_fast_cells<48, false>"),
                                    ("orient_desc", "k_orient_desc<2>", "k_orient_desc<2>")):
            k_, e_ = kern.get(nm_), sq["kernels"].get(kname)
            if not k_ or not e_ or not e_.get("valu_per_launch"):
                continue
            cus = min(B, 256) if nm_.startswith("resolve") else 256
            cyc = e_["valu_per_launch"] * valu_cycles(mixname)
            frac = cyc / (cus * 4 * CLOCK_HZ * k_["avg_launch_ms"] / 1e3)
            out["roofline_by_kernel"].setdefault(kname, {}).update({
                "bound": "valu", "valu_instructions_per_launch": e_["valu_per_launch"], "cycles_per_instruction": valu_cycles(mixname),
                "simds_it_can_use": cus * 4, "avg_launch_ms": k_["avg_launch_ms"], "valu_issue_frac": frac,
                "note": "a workgroup (1 024 lanes) per frame: %d of 256 CUs" % cus if nm_.startswith("resolve") else "whole chip"})
    first = [kern.get("search_last_batch(first pass)"), kern.get("search_local_batch(first pass)")]
    if all(first):
        ms = first[0]["avg_launch_ms"] + first[1]["avg_launch_ms"]
        ach = cmp_search * B / (ms / 1e3)
        out["searches_first_pass"] = {"compares_per_s": ach, "frac_of_valu_bound": ach / VALU_COMPARES_PER_S, "ms_per_step": ms,
                                      "note": "window scans: grid ranges, records, descriptors of ~%d candidates per frame behind dependent loads - latency, not issue" % int(cmp_search)}
    launches = sum(v_["launches_per_step"] for v_ in kern.values()) + 2 * 8 + 26 + 10   # + pyramid / FAST / octree / descriptors of the two extractor pairs, copies, fills, deliveries
    out["launches_per_frame"] = launches / B
    out["_scen"] = scen
    out["_frames"] = (l0.kL, l0.kR, l0.dL, l0.dR, nL, nR)
    for ln in lanes:
        ln.tb.close()
        ln.exL.close()
        ln.exR.close()
    devL.free()
    devR.free()
    two.shutdown()
    ahead.shutdown()
    return out


def count_compares(orb, ctx, F, sf, scen, b, th, cam, Trl):
    """DescriptorDistance calls of the two searches of one frame = keypoints GetFeaturesInArea returns for every searched point:
    last frame (ORBmatcher.cc:1830-1846, 1905-1915: radius th * sf[octave], levels [octave - 1, octave + 1], left and right
    camera; projections recomputed here in float64) and local map (ORBmatcher.cc:76-90, 150-160: radius RadiusByViewingCos *
    th * sf[predicted level], levels [level - 1, level], from the frustum fields the search itself used)."""
    from fasttrack_amd import scenarios as sc
    last, Tcw_last, pts, Rcw, tcw = scen
    nlevels = len(sf)
    T = np.asarray(Tcw_last, np.float64).reshape(3, 4)
    Pc = np.asarray(last["world_pos"], np.float64) @ T[:, :3].T + T[:, 3]
    Pr = Pc @ np.asarray(Trl, np.float64)[:, :3].T + np.asarray(Trl, np.float64)[:, 3]
    octv = np.asarray(last["octave"])
    qs = []
    for P, right in ((Pc, None), (Pr, 1)):
        ok = (np.asarray(last["valid"]) > 0) & (P[:, 2] > 0)
        uv = sc.kb8_project64(cam, P[ok])
        qs.append((uv[:, 0], uv[:, 1], np.float32(th) * sf[octv[ok]], octv[ok] - 1, octv[ok] + 1, None if right is None else np.ones(int(ok.sum()), np.uint8)))
    for side in ("", "_r"):
        inv = b["in_view" + side] > 0
        lv = np.clip(b["level" + side], 0, nlevels - 1)
        r = np.where(b["view_cos" + side] > np.float32(0.998), 2.5, 4.0).astype(np.float32) * np.float32(th)
        px, py = (b["proj_x"], b["proj_y"]) if side == "" else (b["proj_xr"], b["proj_yr"])
        qs.append((px[inv], py[inv], (r * sf[lv])[inv], lv[inv] - 1, lv[inv], None if side == "" else np.ones(int(inv.sum()), np.uint8)))
    total = 0
    for x, y, rad, lo, hi, right in qs:
        if len(x) == 0:
            continue
        _, cnt = orb.features_in_area(ctx, F, np.asarray(x, np.float32), np.asarray(y, np.float32), np.asarray(rad, np.float32),
                                      np.asarray(lo, np.int32), np.asarray(hi, np.int32), right, capacity=8)
        total += int(np.asarray(cnt).sum())
    return total


def tracking_cpu_baseline(pairs, scen, cam, Trl, sf, lap, w, h, nf, LOG_SF, budget_s):
    """the oracle on ONE host core (the reference's tracking thread runs these stages one after the other; its two
    extraction threads are the only parallelism, Frame.cc:127-130): same frames, same sequence, th = 7"""
    from oracle import binding as ob
    from fasttrack_amd import scenarios as sc
    exL, exR = ob.Extractor(nf, SCALE, NLEVELS, INI_TH, MIN_TH), ob.Extractor(nf, SCALE, NLEVELS, INI_TH, MIN_TH)
    n, t0 = 0, time.perf_counter()
    while True:
        i = n % len(pairs)
        kL, dL, _ = exL.extract(pairs[i][0], lap)
        kR, dR, _ = exR.extract(pairs[i][1], lap)
        l2r = ob.fisheye_match(dL, dR)["matches"].astype(np.int32)
        r2l = np.full(len(kR), -1, np.int32)
        ok = l2r >= 0
        r2l[l2r[ok]] = np.nonzero(ok)[0].astype(np.int32)
        F = ob.FrameView(keys=kL, keys_right=kR, descriptors=np.concatenate([dL, dR]), scale_factors_=sf, bounds=sc.frame_bounds(w, h),
                         left_to_right=l2r, right_to_left=r2l, cam_model=1, cam=cam, Trl=Trl)
        last, Tcw_last, pts, Rcw, tcw = scen[i]
        ob.search_last_frame(F, last, Tcw_last, 7.0, False, False, True)
        fr = ob.is_in_frustum(F, ob.make_pose(Rcw, tcw, (0.101, 0.0, 0.0)), pts, 0.5, LOG_SF)
        ob.search_local_points(F, sc.local_points_from_frustum(fr, pts), 7.0)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 256:
            break
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} synthetic 512x512 fisheye pairs (nFeatures {nf}): oracle extract L + R with lapping areas, fisheye match, "
                      f"SearchByProjection(last frame) + isInFrustum + SearchByProjection(local map, {len(scen[0][2]['world_pos'])} points), th 7, one thread, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="stereo_1280x720_nf2000", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=512, help="stereo pairs per step per GPU")
    ap.add_argument("--distinct", type=int, default=0, help="distinct frames cycled through the batch (0 = the batch size)")
    ap.add_argument("--scenes", type=int, default=16, help="seeded synthetic scenes behind the distinct frames")
    ap.add_argument("--density", type=float, default=1.0, help="object density of the synthetic scenes (synth.py)")
    ap.add_argument("--scene", default="objects", choices=["objects", "planes"],
                    help="objects: every object at a disparity of its own (keypoints on occlusion edges, ~10 %% of the left keypoints "
                         "match); planes: the same scene on fronto-parallel planes (synth.make_planes_pair: ~60 %% match)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the extra workloads (752x480, tracking 512x512, dense, planes)")
    ap.add_argument("--workload-batch", type=int, default=512, help="pairs per step of the extra stereo workloads")
    ap.add_argument("--workload-frames", type=int, default=48, help="frames per threshold of the tracking workload")
    ap.add_argument("--tracking-batch", type=int, default=128, help="frames per launch of the tracking workload's throughput form (ft_tracked_batch)")
    ap.add_argument("--mosaic", type=int, default=0, metavar="BLOCK",
                    help="dense-corner scenes instead (synth.make_mosaic_pair with tiles of BLOCK px: 8 gives > 8 k FAST "
                         "candidates at level 0 of a 1280x720 frame)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-in", action="store_true", help="skip the second timed region with frames in pinned host memory")
    ap.add_argument("--sync-steps", action="store_true", help="one front end, every step fully drained before the next")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight (front ends used round-robin)")
    ap.add_argument("--stats", default="", help="write per-stage timings to this file")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank, local_rank, world = shard.env()
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: a run must not pass as a {args.gpus}-GPU result")
    from fasttrack_amd import orb  # the HIP library: loaded by the ranks only
    # plumbing only: CPU tensors over gloo; the data path has no exchange step
    dist = shard.init(rank, world)

    w, h, nf, cfg_note = WORKLOADS[args.workload]
    B = args.batch
    host_threads = args.host_threads or max(1, usable_cpus() // max(world, 1))
    # one device per rank; FT_BENCH_DEVICE_MOD=<n> folds the ranks onto n devices (plumbing check on a smaller box)
    ndev_mod = int(os.environ.get("FT_BENCH_DEVICE_MOD", "0"))
    device = local_rank % ndev_mod if ndev_mod > 0 else local_rank
    # host threads next to the GPU: the rank pins itself to the NUMA node its device hangs off (ranks that share a node share
    # its CPUs evenly) before the context creates its thread pool; FT_BENCH_NUMA=0 leaves the affinity alone
    numa = {"node": -1, "cpus": None}
    if world > 1 and os.environ.get("FT_BENCH_NUMA", "1") != "0":
        import ctypes as _C
        from fasttrack_amd import _capi
        ndev = max(_capi.lib().ft_device_count(), 1)
        nodes = []
        for d in range(ndev):
            buf = _C.create_string_buffer(64)
            nodes.append(shard.numa_node_of_pci(buf.value.decode()) if _capi.lib().ft_device_pci_bus_id(d, buf, 64) == 0 else -1)
        numa["node"] = nodes[device] if device < len(nodes) else -1
        peers = [r for r in range(world) if nodes[(r % ndev_mod if ndev_mod > 0 else r) % ndev] == numa["node"]]
        cpus = shard.pin_to_numa_node(numa["node"], len(peers), peers.index(local_rank) if local_rank in peers else 0)
        numa["cpus"] = len(cpus) if cpus else None
        if cpus and not args.host_threads:
            host_threads = max(1, min(host_threads, len(cpus)))
    ctx = orb.Context(device, host_threads)
    intr = synth.intrinsics(w, h)
    # two front ends used alternately: while one batch drains (last descriptors, matching, result copies) the
    # next batch's pyramid / FAST / octree already run (ft_stereo_frontend_submit / _wait)
    fes = [orb.StereoFrontend(ctx, nf, SCALE, NLEVELS, INI_TH, MIN_TH, w, h, B, intr["mbf"], intr["mb"])
           for _ in range(1 if args.sync_steps else max(1, args.in_flight))]
    fe = fes[0]

    # synthetic stream (seeds are per rank: one stream per GPU): D distinct pairs in pinned host memory and, for the
    # headline number, resident in HBM before the timed region
    D = max(1, min(args.distinct or B, B))
    hostL, hostR, pairs = make_stream(ctx.pinned_array, w, h, rank, D, args.scenes, args.density, args.mosaic, args.scene == "planes")
    devL, devR = ctx.to_device(hostL), ctx.to_device(hostR)
    import ctypes as C
    fb = w * h
    ptrsL = (C.c_void_p * B)(*[devL.ptr.value + (b % D) * fb for b in range(B)])
    ptrsR = (C.c_void_p * B)(*[devR.ptr.value + (b % D) * fb for b in range(B)])
    hptrsL = (C.c_void_p * B)(*[hostL.ctypes.data + (b % D) * fb for b in range(B)])
    hptrsR = (C.c_void_p * B)(*[hostR.ctypes.data + (b % D) * fb for b in range(B)])

    def barrier():
        ctx.synchronize()  # hipDeviceSynchronize on this rank's device (the library owns its HIP runtime)
        shard.barrier(dist)

    done_at = []  # completion time of every step of the last run() (results of the step in host arrays)

    def run(steps, pL, pR, on_device):
        """`steps` passes over the batch; every pass is complete (results in host arrays) on return"""
        del done_at[:]
        if len(fes) == 1:
            for _ in range(steps):
                fe.process_raw(pL, pR, B, on_device, w)
                done_at.append(time.perf_counter())
            return
        F = len(fes)
        for k in range(steps):
            if k >= F:
                fes[k % F].wait()  # the batch submitted F steps ago
                done_at.append(time.perf_counter())
            fes[k % F].submit_raw(pL, pR, B, on_device, w)
        for k in range(max(steps - F, 0), steps):
            fes[k % F].wait()
            done_at.append(time.perf_counter())

    def steady_rate():
        """frames/s between the completions of the steps in the middle of the timed region: without the fill of the pipeline at
        its start (nothing completes for the first ~1.5 steps) and the drain at its end - what a stream of batches sees"""
        if len(done_at) < 6:
            return None
        a_, b_ = len(done_at) // 4, len(done_at) - 1 - len(done_at) // 8
        return B * (b_ - a_) / (done_at[b_] - done_at[a_])

    def timed(steps, pL, pR, on_device):
        barrier()
        t0 = time.perf_counter()
        run(steps, pL, pR, on_device)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        return dt

    run(max(args.warmup, 1), ptrsL, ptrsR, True)
    ctx.reset_stats()
    ctx.set_kernel_timing(True)
    elapsed_rank = timed(args.steps, ptrsL, ptrsR, True)
    steady_rank = steady_rate()
    ctx.set_kernel_timing(False)
    kps_rank = int(fe._nL[:B].sum() + fe._nR[:B].sum())  # keypoints of one pass over this rank's batch
    matches = int(fe._nm[:B].sum())

    elapsed = shard.reduce_max(dist, elapsed_rank)
    kps, matches, kpsL = [int(v) for v in shard.reduce_sum(dist, [kps_rank, matches, int(fe._nL[:B].sum())])]
    rank_fps = shard.gather_floats(dist, B * args.steps / elapsed_rank, world)
    rank_steady = shard.gather_floats(dist, steady_rank or 0.0, world)

    if args.stats:
        ctx.save_stats(args.stats + (f".rank{rank}" if world > 1 else ""))
    # statistics of the headline region (rank 0's kernels), read before the host-in region runs
    kern = {}
    for name in ("kernel.pyr_down(all levels)", "kernel.fast_cells", "kernel.compact", "kernel.octree", "kernel.orient_desc",
                 "kernel.stereo_rowsort", "kernel.stereo_match", "kernel.stereo_median"):
        m, n = ctx.get_stat(name)
        kern[name] = {"ms_per_launch": (m / n) if n else None, "launches": n}
    host = {}
    for name in ("stereo.octree(host,both)", "stereo.host_wait_stageA", "stereo.host_launch_stageB",
                 "stereo.host_tail_sync", "stereo.submit.total", "stereo.device_octree_fallbacks"):
        m, n = ctx.get_stat(name)
        host[name] = (m / n) if n else None
    fallbacks = ctx.get_stat("stereo.device_octree_fallbacks")[1]

    # ---- host-in: the same workload with every frame handed over in pinned host memory, as the reference's call sites
    # do (cv::Mat, src/Frame.cc:442-449; the reference uploads per call, src/ORBextractor.cc:1522-1541): the H2D copies
    # of all 2 B frames of a step are inside the timed region.  Never the headline value.
    host_in = None
    if not args.no_host_in:
        hsteps = max(2, min(args.steps, 12))
        run(2, hptrsL, hptrsR, False)
        ctx.reset_stats()
        dt_h = shard.reduce_max(dist, timed(hsteps, hptrsL, hptrsR, False))
        if args.stats:
            ctx.save_stats(args.stats + ".host_in" + (f".rank{rank}" if world > 1 else ""))
        host_in = {"value": B * hsteps * world / dt_h, "unit": "frames/s", "steps": hsteps, "ms_per_step": 1e3 * dt_h / hsteps,
                   "h2d_bytes_per_step_per_gpu": 2 * B * fb, "h2d_GBps_per_gpu": 2 * B * fb * hsteps / dt_h / 1e9,
                   "note": "frames in pinned host memory (ft_host_malloc), uploaded inside the timed region; results to host as in the headline"}

    if rank == 0:
        frames = B * args.steps * world
        fps = frames / elapsed
        P = level_pixels(w, h)
        sumP = sum(P)
        # Roofline leg.  Algorithmic bytes per kernel: FAST + NMS reads every pyramid pixel once (sum P_l per
        # image); the 7 pyramid launches of a sub-batch read P - P7 and write P - P0 per image; orientation + descriptor
        # read a 43x43 patch and write 60 B per keypoint.  The batch is processed in sub-batches, so bytes per launch =
        # algorithmic bytes of rank 0's timed region / its launches; durations are HIP events on the launching stream.
        # `roofline` is k_fast_cells (see below), the others follow in roofline_other_kernels.  HBM-side traffic per launch
        # comes from the committed PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read live; FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes) and is reported when this run's launch shape equals the profiled one.
        lib_stamp = csrc_stamp(orb.version())
        tj, tj_state = load_profile(TRAFFIC_JSON, lib_stamp)
        tk = tj.get("kernels", {})
        same_inputs = (tj.get("distinct_pairs") == D and tj.get("batch_pairs") == B and tj.get("workload") == args.workload and
                       not args.mosaic and args.density == 1.0 and args.scene == "objects")  # the profiled scenes are the default ones

        def leg(stat, kernel, total_bytes, per_group=1):
            k = kern[stat]  # read before the host-in region reset the statistics
            n = k["launches"]
            if not n:
                return None
            m = k["ms_per_launch"] * n
            achieved = total_bytes / (m / 1e3) / 1e9
            traffic = None
            t = tk.get(kernel)
            imgs = args.steps * 2.0 * B / n
            if t and same_inputs and abs(imgs - t["images_per_launch"] * per_group) < 0.5:
                traffic = t["traffic_bytes_per_launch"] * per_group
            return {"bound": "hbm", "kernel": kernel + (" (%d launches per sub-batch, timed as a group)" % per_group if per_group > 1 else ""),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic,
                    "traffic_over_algorithmic": (traffic / (total_bytes / n)) if traffic else None,
                    "traffic_source": f"profiles/{TRAFFIC_JSON} (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes, "
                                      "same workload and launch shape)" if traffic else None,
                    "valu_issue_frac": (t or {}).get("valu_issue_frac"),
                    "bytes_per_launch": total_bytes / n, "avg_launch_ms": m / n, "launches_timed": n, "total_ms": m}
        legs = [leg("kernel.fast_cells", "k_fast_cells", float(args.steps) * 2.0 * B * sumP),
                leg("kernel.pyr_down(all levels)", "k_pyr_rows", float(args.steps) * 2.0 * B * ((sumP - P[-1]) + (sumP - P[0])), 7),
                leg("kernel.orient_desc", "k_orient_desc", float(kps_rank) * args.steps * (43 * 43 + 60))]
        # Dominant kernel = k_fast_cells: the largest cost inside the overlapped pipeline (profiles/*_marginal_costs.json).
        # Kernels of eight streams share the chip, so a launch's duration (HIP events and rocprofv3 alike) is the time
        # it was resident, stretched by whatever ran beside it - the better the overlap, the longer every kernel "takes".
        # `achieved` / `frac` follow the contract (algorithmic bytes / that duration); `at_marginal_cost` repeats them
        # with the kernel's cost inside the pipeline (the step time it adds when it is enqueued twice), from the
        # committed measurement of the same workload.
        legs = [x for x in legs if x]
        roof = legs[0] if legs else None
        mc, mc_state = load_profile(MARGINAL_JSON, lib_stamp)
        if roof and mc.get("workload") == args.workload and mc.get("batch_pairs") == B and not args.mosaic and args.scene == "objects":
            ms = mc["marginal_ms_per_step"]["k_fast_cells"] / (roof["launches_timed"] / args.steps)
            ach = roof["bytes_per_launch"] / (ms / 1e3) / 1e9
            roof["at_marginal_cost"] = {"ms_per_launch": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                                        "source": f"profiles/{MARGINAL_JSON} (tools/marginal_costs.py: FT_DEBUG_REPEAT=fast, same workload, same csrc)"}
            # the same for the vector-issue bound: the launch's vector instructions (committed SQ pass) x the opcode-weighted
            # cycles of the kernel's stream (valu_cycles) / (1 024 SIMDs x 2.4 GHz x the launch's cost inside the pipeline) -
            # ~1.0 says the kernel is priced at its instruction stream, which is what "the chip is full" means for it
            tkf = tk.get("k_fast_cells") or {}
            if same_inputs and tkf.get("valu_per_launch"):
                roof["at_marginal_cost"]["valu_issue_frac"] = tkf["valu_per_launch"] * valu_cycles("k_fast_cells") / (SIMDS * CLOCK_HZ * ms / 1e3)
                roof["at_marginal_cost"]["valu_cycles_per_instruction"] = valu_cycles("k_fast_cells")
        also = sorted(legs[1:], key=lambda x: -x["total_ms"])
        # The ceiling this integer / bitwise path really works against: vector-instruction issue.  Wave-level vector
        # instructions of the timed region (per-launch counts of the committed SQ pass x this run's launches) x the mean issue
        # cycles of each kernel's opcode mix (valu_cycles: 3.9 - 4.3 for these kernels - they are built from the dear class)
        # / (1 024 SIMDs x 2.4 GHz x elapsed).  Reported when the run's inputs equal the profiled ones.
        valu_frac = None
        if same_inputs:
            pairs_ = (("kernel.fast_cells", "k_fast_cells"), ("kernel.orient_desc", "k_orient_desc"), ("kernel.pyr_down(all levels)", "k_pyr_rows"),
                      ("kernel.octree", "k_octree"), ("kernel.stereo_match", "k_stereo_match"))
            tot_valu = 0.0
            for stat, kname in pairs_:
                t = tk.get(kname)
                n = kern[stat]["launches"]
                if not t or not n or "valu_per_launch" not in t:
                    tot_valu = None
                    break
                tot_valu += t["valu_per_launch"] * n * (7 if kname == "k_pyr_rows" else 1) * valu_cycles(kname)  # the pyramid stat groups 7 launches
            if tot_valu:
                valu_frac = tot_valu / (SIMDS * CLOCK_HZ * elapsed_rank)
        R_pair = 2 * (3 * sumP - P[-1])  # read bytes per stereo pair, unfused accounting
        out = {
            "metric": "frames/sec extract+match", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": args.workload, "note": cfg_note, "frame": "one rectified stereo pair",
                       "image": [w, h], "nfeatures": nf, "nlevels": NLEVELS, "scale_factor": SCALE,
                       "fast_thresholds": [INI_TH, MIN_TH], "batch_pairs_per_gpu": B, "distinct_pairs": D,
                       "scenes": min(args.scenes, D), "scene_density": args.density,
                       "scene_kind": f"mosaic of {args.mosaic}-px tiles" if args.mosaic else "object scene on fronto-parallel planes" if args.scene == "planes" else "objects on a smooth background",
                       "inputs": "resident in HBM before the timed region (host_in: pinned host memory, uploaded inside it)",
                       "batches_in_flight": len(fes),
                       "parallelism": f"{world} independent stream(s), one per GPU, no collective",
                       "host_threads_per_gpu": ctx.host_threads, "device": ctx.device_name, "hw_queues": ctx.hw_queues,
                       "numa_node_of_rank0": numa["node"], "cpus_pinned_rank0": numa["cpus"]},
            "per_rank_frames_per_s": rank_fps,
            # beside `value` (the contract: all frames / the whole timed region, fill and drain of the pipeline included): the
            # rate between step completions in the middle of the region, summed over the ranks
            "steady_state_frames_per_s": sum(rank_steady) if all(rank_steady) else None,
            "keypoints_per_s": kps * args.steps / elapsed,
            "keypoints_per_frame": kps / (B * world),
            "stereo_matches_per_frame": matches / (B * world),
            "stereo_match_fraction": matches / max(kpsL, 1),
            "device_octree_fallbacks": fallbacks,
            "pipeline_hbm_read_frac": fps / world * R_pair / (HBM_PEAK_GBS * 1e9),
            "pipeline_valu_issue_frac": valu_frac,
            "kernels": kern, "host_ms_per_step": host,
            "roofline": roof,
            "roofline_other_kernels": also,
            "library": orb.version(),
            "profiles": {TRAFFIC_JSON: tj_state, MARGINAL_JSON: mc_state},
            "valu_issue_pricing": {"source": f"profiles/{VALU_MIX_JSON} (tools/valu_mix.py over profiles/r06_valu_rates.txt)",
                                   "cycles_per_instruction": {k: valu_cycles(k) for k in ("k_fast_cells", "k_orient_desc", "k_pyr_rows", "k_octree", "k_stereo_match")}},
            "host_in": host_in,
        }
        # The other workloads north_star names (N = 1 only: they are this box's numbers, not part of the scaling curve).  The
        # headline's buffers are released first; every leg builds and releases its own.
        out["workloads"] = None
        out["latency"] = None
        if world == 1 and not args.no_workloads:
            for f in fes:
                f.close()
            devL.free()
            devR.free()
            cpu = not args.no_cpu_baseline
            wl = {}
            WB = args.workload_batch
            wl["stereo_752x480_nf1200"] = stereo_leg(orb, ctx, "stereo_752x480_nf1200", 752, 480, 1200, WB, 48, 5, cpu=cpu)
            wl["tracking_512x512_nf2000"] = tracking_leg(orb, ctx, frames=args.workload_frames, cpu=cpu)
            thr = tracking_batch_leg(orb, ctx, B=args.tracking_batch, steps=16, warmup=4)
            thr = {k_: v_ for k_, v_ in thr.items() if not k_.startswith("_")}
            if "cpu_baseline" in wl["tracking_512x512_nf2000"]:
                thr["gpu_over_cpu"] = thr["value"] / wl["tracking_512x512_nf2000"]["cpu_baseline"]["value"]
            wl["tracking_512x512_nf2000"]["throughput"] = thr
            wl["dense_1280x720_nf2000"] = stereo_leg(orb, ctx, "dense", 1280, 720, 2000, WB, 16, 4, mosaic=10, cpu=cpu, cpu_budget_s=3.0)
            wl["planes_1280x720_nf2000"] = stereo_leg(orb, ctx, "planes", 1280, 720, 2000, WB, 16, 4, planes=True, cpu=cpu, cpu_budget_s=3.0)
            out["workloads"] = wl
            out["latency"] = latency_leg(orb, ctx)
        if not args.no_cpu_baseline:  # rank 0, after the last barrier: the other ranks are done and the host cores are free
            out["cpu_baseline"] = cpu_baseline(w, h, nf, pairs)
            out["gpu_over_cpu_per_gpu"] = fps / world / out["cpu_baseline"]["value"]
            if world == 1:
                out["gpu_over_cpu"] = fps / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    shard.finish(dist)


if __name__ == "__main__":
    main()
