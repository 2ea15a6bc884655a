"""Per-frame latency of the tracking searches on one MI355X: a 752x480 stereo frame, the motion-model
search over the last frame's points followed by the local-map step (isInFrustum + SearchByProjection).
  separate : ft_search_last_frame, ft_is_in_frustum, ft_search_local_points - every call marshals its arrays
             (what the reference does per kernel, CudaFrame::setMemory)
  resident : ft_tracked_frame_* - frame uploaded once, frustum fields stay on the device
  bound    : the same with the frame bound to the buffers the stereo front end left in HBM (no upload at all)
  oracle   : the CPU restatement on one host core
usage: python tests/tools/bench_tracking.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from fasttrack_amd import orb
from oracle import binding as ob
from tests import scenarios as sc

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w, h, nf = 752, 480, 1200
LOG_SF = float(np.float32(np.log(np.float32(1.2))))
fr = sc.oracle_stereo_frame(w, h, nf, 23)
sf, _ = ob.scale_factors(1.2, 8)
sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["intr"]["mbf"], fr["intr"]["mb"])
pts, Rcw, tcw = sc.map_points_scenario(fr["kL"], fr["dL"], sm["depth"], fr["intr"], 8, sf, 123, M=2500)
last, Tcw_last = sc.last_frame_scenario(fr["kL"], fr["dL"], sm["uright"], sm["depth"], fr["intr"], w, h, seed=4)
kw = dict(keys=fr["kL"], descriptors=fr["dL"], bounds=sc.frame_bounds(w, h), mbf=fr["intr"]["mbf"], mb=fr["intr"]["mb"],
          uright=sm["uright"], cam=[fr["intr"][k] for k in ("fx", "fy", "cx", "cy")])
ctx = orb.Context(0)


def timeit(fn, n):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


def separate():
    F = orb.FrameView(scale_factors=sf, **kw)
    orb.KernelController.launchPoseEstimationKernel(ctx, F, last, Tcw_last, 15.0, False, False, True)
    f = orb.is_in_frustum(ctx, F, orb.make_pose(Rcw, tcw), pts, 0.5, LOG_SF)
    return orb.KernelController.launchSearchLocalPointsKernel(ctx, F, sc.local_points_from_frustum(f, pts), 3.0)["n"]


tf = orb.TrackedFrame(ctx, 4096, 4096)
pose = orb.make_pose(Rcw, tcw)


def resident():
    F = orb.FrameView(scale_factors=sf, **kw)
    tf.upload(F)
    tf.search_last_frame(last, Tcw_last, 15.0)
    return tf.track_local_map(pose, pts, 0.5, LOG_SF, 3.0)["n"]


# the frame as the stereo front end leaves it in HBM: bound, not uploaded (the flow of a SLAM front end)
fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, 1, fr["intr"]["mbf"], fr["intr"]["mb"])
fe.process([fr["L"]], [fr["R"]])  # the same pair the oracle frame was extracted from: identical keypoints in HBM
tfb = orb.TrackedFrame(ctx, max_keypoints=fe.capacity, max_points=4096)


def bound():
    F = orb.FrameView(scale_factors=sf, **kw)
    tfb.bind_stereo(fe, 0, F)
    tfb.search_last_frame(last, Tcw_last, 15.0)
    return tfb.track_local_map(pose, pts, 0.5, LOG_SF, 3.0)["n"]


def oracle():
    F = ob.FrameView(scale_factors_=sf, **kw)
    ob.search_last_frame(F, last, Tcw_last, 15.0, False, False, True)
    f = ob.is_in_frustum(F, ob.make_pose(Rcw, tcw), pts, 0.5, LOG_SF)
    return ob.search_local_points(F, sc.local_points_from_frustum(f, pts), 3.0)["n"]


assert separate() == resident() == bound() == oracle()


def library_ms(fn, names, n):
    """time inside the C++ entry points only (the library's own per-call timers), without the Python marshalling"""
    fn()
    ctx.reset_stats()
    for _ in range(n):
        fn()
    tot = 0.0
    for name in names:
        ms, calls = ctx.get_stat(name)
        tot += ms / max(calls, 1)
    return tot

out = {"frame": [w, h], "keypoints": int(len(fr["kL"])), "last_frame_points": int(len(last["valid"])),
       "local_map_points": int(len(pts["world_pos"])),
       "ms_per_frame": {"separate_calls": timeit(separate, reps), "resident_frame": timeit(resident, reps), "bound_to_front_end": timeit(bound, reps),
                        "oracle_1_core": timeit(oracle, max(3, reps // 5))},
       "ms_per_frame_inside_the_library": {
           "resident_frame (search_last_frame + track_local_map)": library_ms(resident, ["tracked.search_last_frame.total", "tracked.track_local_map.total"], reps),
           "bound_to_front_end": library_ms(bound, ["tracked.search_last_frame.total", "tracked.track_local_map.total"], reps)},
       "frustum_only_ms": timeit(lambda: orb.is_in_frustum(ctx, orb.FrameView(scale_factors=sf, **kw), pose, pts, 0.5, LOG_SF), reps),
       "frustum_only_oracle_ms": timeit(lambda: ob.is_in_frustum(ob.FrameView(scale_factors_=sf, **kw), ob.make_pose(Rcw, tcw), pts, 0.5, LOG_SF), reps),
       "note": "median wall time per frame including the Python marshalling of the calls"}
print(json.dumps(out))
