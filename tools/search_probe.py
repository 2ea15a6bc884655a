"""First-pass cost of the local-map search against the window size (run under rocprofv3 --kernel-trace; tools/search_probe.sh):
a two-camera KB8 frame of 2 x 2000 keypoints resident on the device, 2000 local map points, th in TH; prints per th the
number of passes.  The kernel durations come from the trace."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fasttrack_amd import orb, scenarios as sc, synth
ctx = orb.Context(0)
w, h, nf = 512, 512, 2000
ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=2)
L, R = synth.make_planes_pair(w, h, seed=77)
(kL, dL, _), (kR, dR, _) = ex.extract_batch([L, R], (0, 511))
m = orb.KernelController.launchFisheyeStereoMatchKernel(ctx, dL, dR)["matches"].astype(np.int32)
r2l = np.full(len(kR), -1, np.int32); ok = m >= 0; r2l[m[ok]] = np.nonzero(ok)[0]
sf = np.asarray(ex.GetScaleFactors(), np.float32)
cam = list(sc.KB8_CAM); intr = dict(fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
Trl = np.concatenate([np.eye(3), [[-0.101], [0.0], [0.0]]], 1).astype(np.float32)
F = orb.FrameView(keys=kL, keys_right=kR, descriptors=np.concatenate([dL, dR]), scale_factors=sf, bounds=sc.frame_bounds(w, h),
                  left_to_right=m, right_to_left=r2l, cam_model=1, cam=cam, Trl=Trl)
depth = np.zeros(len(kL), np.float32)
pts, Rcw, tcw = sc.map_points_scenario(kL, dL, depth, intr, 8, sf, 4, M=2000)
tf = orb.TrackedFrame(ctx, 2 * ex.max_keypoints + 64, 4096)
LOG_SF = float(np.float32(np.log(np.float32(1.2))))
for th in [float(x) for x in os.environ.get("TH", "1 3 7 15 40").split()]:
    for rep in range(3):
        tf.upload(F)
        ctx.reset_stats()
        b = tf.track_local_map(orb.make_pose(Rcw, tcw, (0.101, 0.0, 0.0)), pts, 0.5, LOG_SF, th)
    print("th", th, "matches", b["n"], "passes", ctx.get_stat("tracked.track_local_map.passes")[0], flush=True)
