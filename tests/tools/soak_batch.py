#!/usr/bin/env python3
"""Randomized parity run of the batched projection searches (ft_tracked_batch_*) on an MI355X: every trial draws B frames of
random kinds - two-camera KannalaBrandt8 frames and rectified pinhole stereo frames of several sizes and feature counts, cut
down to random keypoint counts - with random last-frame points, local maps (0 .. 2000 points), poses, window factors and
occupancies, runs the reference's sequence on all of them through ONE batch (search last frame -> isInFrustum -> local map) and
compares every frame's assignments, frustum fields, nToMatch and final holder_obs with the oracle's sequence on that frame.
A trial runs under search_cache 3 (one-launch resolution, k_resolve_batch), 2 (the same for 24 frames and more) or 1 (claim passes)
and a random burst length; 5 % of the
trials use windows so wide that candidate lists outgrow the cache (resolution falls back to the passes for those frames).
usage: tests/tools/soak_batch.py [--trials N] [--seed S] [--frames B]      exit code 1 on any mismatch"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fasttrack_amd import orb, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402
import scenarios as sc  # noqa: E402

LOG_SF = float(np.float32(np.log(np.float32(1.2))))
TRL = np.concatenate([np.eye(3), [[-0.101], [0.0], [0.0]]], 1).astype(np.float32)
TLR = (0.101, 0.0, 0.0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--frames", type=int, default=16)
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    ctx = orb.Context(0)
    sf, _ = ob.scale_factors(1.2, 8)
    B = args.frames
    tb = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=4400, max_points=2304)
    fails = frames_n = matches = pinned_trials = 0
    modes = {1: 0, 2: 0, 3: 0}
    t0 = time.time()
    bases = {}

    def base(kind, w, h, nf, seed):
        key = (kind, w, h, nf, seed)
        if key not in bases:
            if len(bases) > 24:
                bases.pop(next(iter(bases)))
            fr = sc.fisheye_frame_scenario(w, h, nf, seed) if kind == 1 else sc.oracle_stereo_frame(w, h, nf, seed)
            if kind == 0:
                fr["sm"] = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["intr"]["mbf"], fr["intr"]["mb"])
            bases[key] = fr
        return bases[key]
    for t in range(args.trials):
        th = float(rng.choice([1.0, 3.0, 7.0, 15.0, 30.0, 150.0], p=[0.19, 0.19, 0.19, 0.19, 0.19, 0.05]))  # (150: lists beyond the cache)
        opts = dict(search_cache=int(rng.choice([3, 3, 2, 1])), pass_burst=int(rng.choice([12, 12, 5, 2])))
        far = bool(rng.random() < 0.3)
        th_far = float(rng.uniform(4, 12))
        nn = float(rng.choice([0.8, 0.6, 0.9]))
        views, lasts, Tcws, ptss, poses, orc = [], [], [], [], [], []
        for f in range(B):
            kind = int(rng.random() < 0.6)
            w, h = (512, 512) if kind == 1 else [(752, 480), (640, 480), (376, 240)][int(rng.integers(0, 3))]
            nf = int(rng.choice([500, 1000, 1500, 2000]))
            fr = base(kind, w, h, nf, int(rng.integers(0, 6)))
            nl = max(int(len(fr["kL"]) * rng.uniform(0.4, 1.0)), 1) if rng.random() < 0.3 else len(fr["kL"])
            kL, dL = fr["kL"][:nl], fr["dL"][:nl]
            holder = np.where(rng.random(nl) < 0.1, rng.integers(0, 3, nl), -1).astype(np.int32)
            if kind == 1:
                nr = len(fr["kR"])
                l2r = np.where(fr["l2r"][:nl] < nr, fr["l2r"][:nl], -1).astype(np.int32)
                r2l = np.where(fr["r2l"] < nl, fr["r2l"], -1).astype(np.int32)
                hold = np.concatenate([holder, np.full(nr, -1, np.int32)])
                kw = dict(keys=kL, keys_right=fr["kR"], descriptors=np.concatenate([dL, fr["dR"]]), bounds=sc.frame_bounds(w, h),
                          left_to_right=l2r, right_to_left=r2l, cam_model=1, cam=list(sc.KB8_CAM), Trl=TRL, holder_obs=hold)
                intr = dict(fx=sc.KB8_CAM[0], fy=sc.KB8_CAM[1], cx=sc.KB8_CAM[2], cy=sc.KB8_CAM[3])
                depth, uright, tlr = np.zeros(nl, np.float32), None, TLR
            else:
                uright, depth, intr, tlr = fr["sm"]["uright"][:nl], fr["sm"]["depth"][:nl], fr["intr"], (0, 0, 0)
                kw = dict(keys=kL, descriptors=dL, bounds=sc.frame_bounds(w, h), mbf=intr["mbf"], mb=intr["mb"], uright=uright,
                          holder_obs=holder, cam=[intr[k] for k in ("fx", "fy", "cx", "cy")])
            oF, gF = ob.FrameView(scale_factors_=sf, **kw), orb.FrameView(scale_factors=sf, **kw)
            seed = int(rng.integers(0, 1 << 30))
            last, Tcw = sc.last_frame_scenario(kL, dL, uright, depth, intr, w, h, seed=seed)
            M = int(rng.integers(0, 2001))
            pts, Rcw, tcw = sc.map_points_scenario(kL, dL, depth, intr, 8, sf, seed + 1, M=max(M, 1))
            if M == 0:
                pts = {k: v[:0] for k, v in pts.items()}
            if rng.random() < 0.05:
                last = {k: v[:0] for k, v in last.items()}
            o1 = ob.search_last_frame(oF, last, Tcw, th, False, False, True)
            ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, tlr), pts, 0.5, LOG_SF)
            o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts, far, th_far), th, nn)
            views.append(gF); lasts.append(last); Tcws.append(Tcw); ptss.append(pts); poses.append(orb.make_pose(Rcw, tcw, tlr))
            orc.append((o1, ofr, o2, oF))
        # round 6: a third of the trials hand the point arrays over in pinned memory (read in place: k_gather_batch, frustum fields
        # scattered into pinned arrays), half of the trials use the two halves of the calls (submit / wait)
        pinned = rng.random() < 0.33 and pinned_trials < 40   # (pinned blocks live as long as the context: bounded)
        pinned_trials += int(pinned)
        split = rng.random() < 0.5
        with ctx.options(**opts):  # one-launch resolution (3; 2 from 24 frames on) or the claim passes (1), long and short bursts
            tb.upload(views)
            pl1 = tb.prepare_last(lasts, Tcws, ctx=ctx if pinned else None)
            pl2 = tb.prepare_local(poses, ptss, ctx=ctx if pinned else None)
            if split:
                tb.search_last_frame(pl1, th=th, submit=True)
                g1 = tb.wait()
                tb.track_local_map(pl2, viewing_cos_limit=0.5, log_scale_factor=LOG_SF, th=th, nn_ratio=nn, far_points=far, th_far_points=th_far, submit=True)
                g2 = tb.wait()
            else:
                g1 = tb.search_last_frame(pl1, th=th)
                g2 = tb.track_local_map(pl2, viewing_cos_limit=0.5, log_scale_factor=LOG_SF, th=th, nn_ratio=nn, far_points=far, th_far_points=th_far)
        modes[opts["search_cache"]] += 1
        for f in range(B):
            o1, ofr, o2, oF = orc[f]
            what = None
            if g1[f]["n"] != o1["n"] or not np.array_equal(g1[f]["assign"], o1["assign"]):
                what = "last-frame search"
            for k, _ in ob.FRUSTUM_FIELDS:
                if not what and not np.array_equal(g2[f][k], ofr[k]):
                    what = "frustum " + k
            if not what and g2[f]["n_to_match"] != ofr["n"]:
                what = "nToMatch"
            if not what and (g2[f]["n"] != o2["n"] or not np.array_equal(g2[f]["assign"], o2["assign"])):
                what = "local-map search"
            if not what and not np.array_equal(tb.holder_obs(f), oF.holder_obs):
                what = "holder_obs"
            if what:
                fails += 1
                print(f"MISMATCH trial {t} frame {f} th {th} far {far} {opts}: {what}", flush=True)
            frames_n += 1
            matches += o1["n"] + o2["n"]
    tb.close()
    try:
        fallbacks = ctx.get_stat("tracked_batch.resolve_fallbacks")[1]
    except orb.FastTrackError:
        fallbacks = 0
    one = modes[3] + (modes[2] if B >= 24 else 0)
    print(f"{args.trials} batches of {B} frames ({one} resolved in one launch, {fallbacks} searches of those fell back to the passes; "
          f"{args.trials - one} by the passes; {pinned_trials} with the point arrays read in place from pinned memory): {frames_n} frames, {2 * frames_n} searches, {matches} matches, {fails} mismatches, {time.time() - t0:.0f} s")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
