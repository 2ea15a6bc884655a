#!/usr/bin/env python3
"""Long randomized parity run of the projection searches on an MI355X (not part of the test suite): both
ORBmatcher::SearchByProjection overloads, Frame::isInFrustum and the resident tracked frame against the oracle.
usage: tests/tools/soak_search.py [--trials N] [--seed S]

A trial draws a frame (size, feature count, scene kind; every fourth one a two-camera fisheye frame), extracts it with the
oracle, and then runs several searches on it, each with its own random point set, window factor, flags and occupancy of
mvpMapPoints: the one-shot kernel-controller calls and the sequence on a resident frame (search last frame -> frustum ->
local map, holder_obs carried over).  Assignments, every raw array, the frustum fields and the final occupancy must equal
the oracle's.  The two-camera trials add the KannalaBrandt8 model: a last-frame search and isInFrustum projecting through it
and the fisheye stereo triangulation on a random rig, every output compared for equality.  Prints one line per failure and a summary; exit code 1 on any mismatch."""
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
RAW1 = ("best_dist", "best_dist2", "best_level", "best_level2", "best_idx")
RAW2 = RAW1 + ("best_dist_r", "best_dist2_r", "best_level_r", "best_level2_r", "best_idx_r")


def frame_of(rng, w, h, nf, kind, seed):
    if kind == 0:
        L, R = synth.make_stereo_pair(w, h, seed)
    elif kind == 1:
        L, R = synth.make_planes_pair(w, h, seed=seed)
    else:
        L, R = synth.make_mosaic_pair(w, h, seed=seed, block=int(rng.integers(8, 20)))
    exL, exR = ob.Extractor(nf), ob.Extractor(nf)
    kL, dL, _ = exL.extract(L)
    kR, dR, _ = exR.extract(R)
    return dict(L=L, R=R, exL=exL, exR=exR, kL=kL, dL=dL, kR=kR, dR=dR, intr=synth.intrinsics(w, h))


def views(fr, sf, w, h, uright, holder):
    args = dict(keys=fr["kL"], descriptors=fr["dL"], bounds=sc.frame_bounds(w, h), mbf=fr["intr"]["mbf"], mb=fr["intr"]["mb"],
                uright=uright, holder_obs=holder, cam=[fr["intr"][k] for k in ("fx", "fy", "cx", "cy")])
    return ob.FrameView(scale_factors_=sf, **args), orb.FrameView(scale_factors=sf, **args)


def diff(g, o, keys):
    if g["n"] != o["n"]:
        return "n %d vs %d" % (g["n"], o["n"])
    for k in ("assign",) + tuple(keys):
        if not np.array_equal(g[k], o[k]):
            return k
    return None


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    ctx = orb.Context(0)
    sf, _ = ob.scale_factors(1.2, 8)
    sizes = [(752, 480), (640, 480), (512, 512), (1280, 720), (376, 240)]
    fails, searches, matched, t0 = 0, 0, 0, time.time()
    kb8_frustum_points = kb8_pairs = se3_searches = 0

    def check(tag, trial, what):
        nonlocal fails
        if what:
            fails += 1
            print(f"MISMATCH trial {trial} {tag}: {what}", flush=True)

    for t in range(args.trials):
        w, h = sizes[int(rng.integers(0, len(sizes)))]
        nf = int(rng.integers(300, 2001))
        seed = int(rng.integers(0, 1 << 30))
        two_cam = t % 4 == 3
        desc = f"{w}x{h} nf{nf} seed{seed}"
        if two_cam:
            fr = frame_of(rng, w, h, nf, 0, seed)
            m = ob.fisheye_match(fr["dL"], fr["dR"])
            l2r = m["matches"].astype(np.int32)
            r2l = np.full(len(fr["kR"]), -1, np.int32)
            r2l[l2r[l2r >= 0]] = np.nonzero(l2r >= 0)[0]
            fr["l2r"], fr["r2l"] = l2r, r2l
            if len(fr["kL"]) < 8 or len(fr["kR"]) < 8:
                continue
            kw = dict(keys=fr["kL"], keys_right=fr["kR"], descriptors=np.concatenate([fr["dL"], fr["dR"]]),
                      bounds=sc.frame_bounds(w, h), left_to_right=l2r, right_to_left=r2l)
            for _ in range(3):
                th = float(rng.choice([1.0, 3.0, 7.0, 15.0]))
                pts = sc.two_camera_points(fr, sf, int(rng.integers(0, 1 << 30)), M=int(rng.integers(50, 3000)))
                oF, gF = ob.FrameView(scale_factors_=sf, **kw), orb.FrameView(scale_factors=sf, **kw)
                o = ob.search_local_points(oF, pts, th)
                g = orb.KernelController.launchSearchLocalPointsKernel(ctx, gF, pts, th)
                check(f"two-camera local th{th} [{desc}]", t, diff(g, o, RAW2) or
                      (None if np.array_equal(gF.holder_obs, oF.holder_obs) else "holder_obs"))
                searches += 1
                matched += o["n"]
            # --- KannalaBrandt8 on the same frame: SearchByProjection(last frame) and isInFrustum project through atan2f /
            # cosf / sinf, the fisheye stereo triangulation unprojects through tanf - all equal to the oracle's bits ---
            cam = list(sc.KB8_CAM)
            cam[0] *= float(rng.uniform(0.8, 1.3)); cam[1] = cam[0] * float(rng.uniform(0.999, 1.001))
            cam[2], cam[3] = w / 2 + float(rng.uniform(-8, 8)), h / 2 + float(rng.uniform(-8, 8))
            Trl = np.concatenate([np.eye(3), [[-float(rng.uniform(0.05, 0.2))], [0.0], [0.0]]], 1).astype(np.float32)
            kwk = dict(kw, cam_model=1, cam=cam, Trl=Trl)
            N = len(fr["kL"])
            z = rng.uniform(1.5, 9, N).astype(np.float32)
            X = ((fr["kL"]["x"] - cam[2]) / cam[0] * z).astype(np.float32)
            Y = ((fr["kL"]["y"] - cam[3]) / cam[1] * z).astype(np.float32)
            last = dict(valid=(rng.random(N) < 0.8).astype(np.uint8), world_pos=np.stack([X, Y, z], 1),
                        descriptors=fr["dL"].copy(), observations=rng.integers(0, 4, N).astype(np.int32),
                        octave=fr["kL"]["octave"].astype(np.int32), angle=fr["kL"]["angle"].copy())
            Tcw = sc.random_pose(rng, 0.02, 0.005)
            th = float(rng.choice([7.0, 15.0]))
            oF, gF = ob.FrameView(scale_factors_=sf, **kwk), orb.FrameView(scale_factors=sf, **kwk)
            if t % 8 == 7:  # both poses in the Sophus form
                q, tt = sc.random_se3(rng, 0.02, 0.005)
                qr, _ = sc.random_se3(rng, 0.0, 0.01)
                o = ob.search_last_frame(oF, last, ob.SE3(q, tt), th, False, False, True, Trl=ob.SE3(qr, Trl[:, 3]))
                g = orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, orb.SE3(q, tt), th, False, False, True,
                                                                    Trl=orb.SE3(qr, Trl[:, 3]))
                se3_searches += 1
            else:
                o = ob.search_last_frame(oF, last, Tcw, th, False, False, True)
                g = orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, Tcw, th, False, False, True)
            check(f"KB8 last th{th} [{desc}]", t, diff(g, o, ("best_dist", "best_idx", "best_dist_r", "best_idx_r")))
            searches += 1
            matched += o["n"]
            intr = dict(fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
            pts, Rcw, tcw = sc.map_points_scenario(fr["kL"], fr["dL"], np.zeros(N, np.float32), intr, 8, sf, int(rng.integers(0, 1 << 30)))
            tlr = (-float(Trl[0, 3]), 0.0, 0.0)
            oF, gF = ob.FrameView(scale_factors_=sf, **kwk), orb.FrameView(scale_factors=sf, **kwk)
            ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, tlr), pts, 0.5, LOG_SF)
            gfr = orb.is_in_frustum(ctx, gF, orb.make_pose(Rcw, tcw, tlr), pts, 0.5, LOG_SF)
            bad = next((k for k, _ in ob.FRUSTUM_FIELDS if not np.array_equal(gfr[k], ofr[k])), None)
            check(f"KB8 frustum [{desc}]", t, bad or (None if gfr["n"] == ofr["n"] else "n"))
            kb8_frustum_points += len(pts["world_pos"])
            S = sc.fisheye_rig_scenario(int(rng.integers(0, 1 << 30)), n=int(rng.integers(200, 2500)), noise=float(rng.uniform(0.1, 0.8)))
            n = len(S["xy1"])
            dL = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            perm = rng.permutation(n)
            dR = dL[perm].copy()
            flips = rng.integers(0, 256, (n, 6))
            for k in range(6):
                dR[np.arange(n), flips[:, k] // 8] ^= (1 << (flips[:, k] % 8)).astype(np.uint8)
            kL = np.zeros(n, ob.KP_DTYPE); kR = np.zeros(n, ob.KP_DTYPE)
            kL["x"], kL["y"], kL["octave"] = S["xy1"][:, 0], S["xy1"][:, 1], S["octave1"]
            kR["x"], kR["y"], kR["octave"] = S["xy2"][perm, 0], S["xy2"][perm, 1], S["octave2"][perm]
            ls2 = (sf ** 2).astype(np.float32)
            o = ob.fisheye_stereo(ob.make_rig(sc.KB8_CAM, sc.KB8_CAM, S["Rlr"], S["tlr"]), dL, kL, dR, kR, ls2)
            g = orb.fisheye_stereo(ctx, sc.KB8_CAM, sc.KB8_CAM, S["Rlr"], S["tlr"], dL, kL, dR, kR, ls2)
            bad = next((k for k in ("matches", "depth", "p3d") if not np.array_equal(g[k], o[k])), None)
            check(f"KB8 triangulation n{n} [{desc}]", t, bad or (None if g["n"] == o["n"] else "n"))
            kb8_pairs += n
            continue
        kind = int(rng.integers(0, 3))
        fr = frame_of(rng, w, h, nf, kind, seed)
        N = len(fr["kL"])
        if N < 8:
            continue
        sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["intr"]["mbf"], fr["intr"]["mb"])
        mono = rng.random() < 0.25
        uright = None if mono else sm["uright"]
        # --- one-shot SearchByProjection(Frame, MapPoints) ---
        for _ in range(2):
            th = float(rng.choice([1.0, 3.0, 5.0, 7.0, 15.0, 30.0, 60.0]))
            holder = np.where(rng.random(N) < rng.uniform(0, 0.3), rng.integers(0, 3, N), -1).astype(np.int32)
            pts = sc.local_points_scenario(fr["kL"], fr["dL"], sf, w, h, seed=int(rng.integers(0, 1 << 30)), M=int(rng.integers(20, 4000)),
                                           uright=uright, dense=bool(rng.integers(0, 2)))
            oF, gF = views(fr, sf, w, h, uright, holder)
            o = ob.search_local_points(oF, pts, th)
            g = orb.KernelController.launchSearchLocalPointsKernel(ctx, gF, pts, th)
            check(f"local th{th} kind{kind} mono{int(mono)} [{desc}]", t, diff(g, o, RAW1) or
                  (None if np.array_equal(gF.holder_obs, oF.holder_obs) else "holder_obs"))
            searches += 1
            matched += o["n"]
        if mono or (sm["depth"] > 0).sum() < 8:
            continue
        # --- one-shot SearchByProjection(CurrentFrame, LastFrame) ---
        th = float(rng.choice([7.0, 15.0, 30.0, 90.0]))
        fwd = bool(rng.random() < 0.2)
        bwd = (not fwd) and bool(rng.random() < 0.2)
        ori = bool(rng.random() < 0.8)
        last, Tcw = sc.last_frame_scenario(fr["kL"], fr["dL"], sm["uright"], sm["depth"], fr["intr"], w, h, seed=int(rng.integers(0, 1 << 30)))
        oF, gF = views(fr, sf, w, h, sm["uright"], None)
        if t % 2:  # every other trial hands the pose over as Sophus::SE3f holds it (the CPU branch's quaternion arithmetic)
            q, tt = sc.random_se3(rng, 0.03, 0.006)
            oT, gT = ob.SE3(q, tt), orb.SE3(q, tt)
            se3_searches += 1
        else:
            oT = gT = Tcw
        o = ob.search_last_frame(oF, last, oT, th, fwd, bwd, ori)
        g = orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, gT, th, fwd, bwd, ori)
        check(f"last th{th} fwd{int(fwd)} bwd{int(bwd)} ori{int(ori)} se3{t % 2} [{desc}]", t, diff(g, o, ("best_dist", "best_idx")) or
              (None if np.array_equal(gF.holder_obs, oF.holder_obs) else "holder_obs"))
        searches += 1
        matched += o["n"]
        # --- the sequence on a resident frame ---
        pts, Rcw, tcw = sc.map_points_scenario(fr["kL"], fr["dL"], sm["depth"], fr["intr"], 8, sf, int(rng.integers(0, 1 << 30)),
                                               M=int(rng.integers(100, 4000)))
        far = bool(rng.random() < 0.3)
        th_far = float(np.percentile(sm["depth"][sm["depth"] > 0], 85)) if far else 0.0
        th_last, th_local = float(rng.choice([7.0, 15.0])), float(rng.choice([1.0, 3.0, 5.0, 15.0]))
        oF, gF = views(fr, sf, w, h, sm["uright"], None)
        o1 = ob.search_last_frame(oF, last, Tcw, th_last, False, False, True)
        ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw), pts, 0.5, LOG_SF)
        o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts, far, th_far), th_local)
        tf = orb.TrackedFrame(ctx, max_keypoints=4096, max_points=4096)
        tf.upload(gF)
        g1 = tf.search_last_frame(last, Tcw, th_last)
        g2 = tf.track_local_map(orb.make_pose(Rcw, tcw), pts, 0.5, LOG_SF, th_local, far_points=far, th_far_points=th_far)
        what = diff(g1, o1, ()) and "last: " + diff(g1, o1, ())
        if not what:
            for k, _ in ob.FRUSTUM_FIELDS:
                if not np.array_equal(g2[k], ofr[k]):
                    what = "frustum " + k
                    break
        if not what and g2["n_to_match"] != ofr["n"]:
            what = "n_to_match"
        if not what:
            what = diff(g2, o2, ()) and "local: " + diff(g2, o2, ())
        if not what and not np.array_equal(tf.holder_obs(), oF.holder_obs):
            what = "holder_obs"
        check(f"tracked th{th_last}/{th_local} far{int(far)} kind{kind} [{desc}]", t, what)
        tf.close()
        searches += 2
        matched += o1["n"] + o2["n"]
        if (t + 1) % 50 == 0:
            print(f"trial {t + 1}: {searches} searches, {matched} matches, {fails} mismatches, {time.time() - t0:.0f} s", flush=True)
    def calls(name):
        try:
            return ctx.get_stat(name)[1]
        except Exception:
            return 0
    print(f"KannalaBrandt8, all bit-exact comparisons: {kb8_frustum_points} frustum points, {kb8_pairs} triangulated pairs; "
          f"{se3_searches} last-frame searches with Sophus-form poses")
    print(f"{args.trials} trials, {searches} searches, {matched} matches, {fails} mismatches, {time.time() - t0:.0f} s")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
