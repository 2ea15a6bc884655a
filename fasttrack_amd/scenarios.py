"""Seeded synthetic tracking scenarios: last-frame map points, local map points for isInFrustum /
SearchByProjection, a fisheye rig - pure numpy on top of a frame's keypoints, shared by the tests, the latency tools and
bench.py's tracking leg.  Nothing here touches the oracle or the GPU."""
import numpy as np

KB8_CAM = [190.978, 190.973, 254.93, 256.90, 0.0034, 0.0007, -0.0020, 0.00020]  # TUM-VI-like


def frame_bounds(width, height):
    # rectified / undistorted pinhole: mnMinX = 0, mnMaxX = cols (Frame::ComputeImageBounds, no distortion)
    return (0.0, 0.0, float(width), float(height))



def random_pose(rng, trans=0.05, rot=0.01):
    """small SE(3) step as a row-major 3x4 float32 matrix"""
    w = rng.normal(0, rot, 3)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Rm = np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * K @ K if th > 1e-12 else np.eye(3)
    t = rng.normal(0, trans, 3)
    return np.concatenate([Rm, t[:, None]], 1).astype(np.float32)



def random_se3(rng, trans=0.05, rot=0.01):
    """the same kind of step as (q, t): unit quaternion x y z w (float32) and translation, as Sophus::SE3f holds a pose"""
    w = rng.normal(0, rot, 3)
    th = np.linalg.norm(w)
    axis = w / th if th > 1e-12 else np.array([1.0, 0.0, 0.0])
    q = np.concatenate([np.sin(th / 2) * axis, [np.cos(th / 2)]]).astype(np.float32)
    return q, rng.normal(0, trans, 3).astype(np.float32)


def local_points_scenario(keys, desc, sf, width, height, seed, M=1500, zero_obs_frac=0.15, uright=None,
                          mbf=40.0, dense=False):
    """Local map points built from a frame's own keypoints: projections jittered around keypoints,
    descriptors with a few flipped bits, a share of zero-observation points and skipped points.
    dense=True packs the projections so that windows overlap and in-call claiming matters."""
    rng = np.random.default_rng(seed)
    N = len(keys)
    src = rng.integers(0, N, M)
    if dense:
        src = rng.integers(0, max(N // 6, 1), M)
    jitter = rng.normal(0, 2.0, (M, 2)).astype(np.float32)
    px = (keys["x"][src] + jitter[:, 0]).astype(np.float32)
    py = (keys["y"][src] + jitter[:, 1]).astype(np.float32)
    d = desc[src].copy()
    flips = rng.integers(0, 256, (M, 12))
    for k in range(12):
        m = rng.random(M) < 0.5
        d[m, flips[m, k] // 8] ^= (1 << (flips[m, k] % 8)).astype(np.uint8)
    level = np.clip(keys["octave"][src] + rng.integers(0, 2, M), 0, len(sf) - 1).astype(np.int32)
    view_cos = np.where(rng.random(M) < 0.5, 0.9995, 0.9).astype(np.float32)
    obs = np.where(rng.random(M) < zero_obs_frac, 0, rng.integers(1, 6, M)).astype(np.int32)
    skip = (rng.random(M) < 0.05).astype(np.uint8)
    if uright is not None:
        ur = uright[src]
        pxr = np.where(ur > 0, ur + rng.normal(0, 1.0, M), px - 5.0).astype(np.float32)
    else:
        pxr = (px - 5.0).astype(np.float32)
    return dict(skip=skip, in_view=np.ones(M, np.uint8), in_view_r=np.zeros(M, np.uint8), level=level,
                level_r=np.full(M, -1, np.int32), view_cos=view_cos, view_cos_r=view_cos.copy(), proj_x=px,
                proj_y=py, proj_xr=pxr, proj_yr=py.copy(), descriptors=d, observations=obs)



def last_frame_scenario(keys, desc, uright, depth, intr, width, height, seed, zero_obs_frac=0.15, cam_model=0,
                        cam_extra=None):
    """Last-frame map points: back-project keypoints that have stereo depth with the frame's intrinsics,
    then move the camera by a small random SE(3) step."""
    rng = np.random.default_rng(seed)
    N = len(keys)
    z = np.where(depth > 0, depth, rng.uniform(2.0, 10.0, N)).astype(np.float32)
    fx, fy, cx, cy = [float(intr[k]) for k in ("fx", "fy", "cx", "cy")]
    X = (keys["x"] - cx) / fx * z
    Y = (keys["y"] - cy) / fy * z
    world = np.stack([X, Y, z], 1).astype(np.float32)
    valid = (rng.random(N) < 0.8).astype(np.uint8)
    d = desc.copy()
    flips = rng.integers(0, 256, (N, 8))
    for k in range(8):
        m = rng.random(N) < 0.5
        d[m, flips[m, k] // 8] ^= (1 << (flips[m, k] % 8)).astype(np.uint8)
    obs = np.where(rng.random(N) < zero_obs_frac, 0, rng.integers(1, 6, N)).astype(np.int32)
    Tcw = random_pose(rng)
    angle = (keys["angle"] + rng.normal(0, 3.0, N)).astype(np.float32) % np.float32(360.0)
    return dict(valid=valid, world_pos=world, descriptors=d, observations=obs, octave=keys["octave"].astype(np.int32),
                angle=angle.astype(np.float32)), Tcw



def map_points_scenario(keys, desc, depth, intr, nlevels, sf, seed, M=2500, tlr=None):
    """Local map points for Frame::isInFrustum: back-projected keypoints (stereo depth where available) seen from
    a camera that moved by a small SE(3) step, plus points behind the camera, outside the image, outside their
    scale-invariance range and seen at a grazing angle, so that every early return is taken."""
    rng = np.random.default_rng(seed)
    N = len(keys)
    src = rng.integers(0, N, M)
    z = np.where(depth[src] > 0, depth[src], rng.uniform(1.0, 12.0, M)).astype(np.float32)
    fx, fy, cx, cy = [float(intr[k]) for k in ("fx", "fy", "cx", "cy")]
    X = ((keys["x"][src] + rng.normal(0, 1.5, M)) - cx) / fx * z
    Y = ((keys["y"][src] + rng.normal(0, 1.5, M)) - cy) / fy * z
    world = np.stack([X, Y, z], 1).astype(np.float32)
    kind = rng.random(M)
    world[kind < 0.05, 2] *= -1.0                       # behind the camera
    world[(kind >= 0.05) & (kind < 0.12), 0] *= 6.0     # outside the image
    T = random_pose(rng, trans=0.1, rot=0.02)
    Rcw, tcw = T[:, :3].copy(), T[:, 3].copy()
    Ow = -(Rcw.T @ tcw)
    PO = world - Ow[None, :]
    dist = np.linalg.norm(PO, axis=1)
    # normals: mean viewing direction with noise; a share at grazing angles
    nrm = PO / dist[:, None] + rng.normal(0, 0.25, (M, 3))  # mNormalVector: mean unit vector camera -> point
    graze = rng.random(M) < 0.1
    nrm[graze] = np.cross(PO[graze], rng.normal(0, 1, (int(graze.sum()), 3)))
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1), 1e-9)[:, None]
    # scale invariance range: the point was observed at level `lv` from distance d0
    lv = np.clip(keys["octave"][src], 0, nlevels - 1)
    d0 = dist * rng.uniform(0.6, 1.6, M)
    max_d = (d0 * sf[lv]).astype(np.float32)
    min_d = (max_d / sf[nlevels - 1]).astype(np.float32)
    d = desc[src].copy()
    flips = rng.integers(0, 256, (M, 10))
    for k in range(10):
        m = rng.random(M) < 0.5
        d[m, flips[m, k] // 8] ^= (1 << (flips[m, k] % 8)).astype(np.uint8)
    obs = np.where(rng.random(M) < 0.15, 0, rng.integers(1, 6, M)).astype(np.int32)
    pts = dict(world_pos=world, normal=nrm.astype(np.float32), max_distance=max_d, min_distance=min_d,
               skip=(rng.random(M) < 0.04).astype(np.uint8), descriptors=d, observations=obs)
    return pts, Rcw.astype(np.float32), tcw.astype(np.float32)



def local_points_from_frustum(fr: dict, pts: dict, far_points=False, th_far=0.0):
    """what ORBmatcher::SearchByProjection reads from the MapPoints after isInFrustum (ORBmatcher.cc:66-74)"""
    skip = (~(fr["in_view"].astype(bool) | fr["in_view_r"].astype(bool))) | pts["skip"].astype(bool)
    if far_points:
        skip |= fr["depth"] > np.float32(th_far)
    return dict(skip=skip.astype(np.uint8), in_view=fr["in_view"], in_view_r=fr["in_view_r"], level=fr["level"],
                level_r=fr["level_r"], view_cos=fr["view_cos"], view_cos_r=fr["view_cos_r"], proj_x=fr["proj_x"],
                proj_y=fr["proj_y"], proj_xr=fr["proj_xr"], proj_yr=fr["proj_yr"], descriptors=pts["descriptors"],
                observations=pts["observations"])



def kb8_project64(cam, P):
    """KannalaBrandt8::project in float64 (independent statement for the tests)"""
    P = np.asarray(P, np.float64)
    r2 = P[:, 0] ** 2 + P[:, 1] ** 2
    theta = np.arctan2(np.sqrt(r2), P[:, 2])
    psi = np.arctan2(P[:, 1], P[:, 0])
    r = theta + cam[4] * theta ** 3 + cam[5] * theta ** 5 + cam[6] * theta ** 7 + cam[7] * theta ** 9
    return np.stack([cam[0] * r * np.cos(psi) + cam[2], cam[1] * r * np.sin(psi) + cam[3]], 1)



def fisheye_rig_scenario(seed, n=1500, noise=0.3):
    """A stereo fisheye rig (baseline 0.1 m, small relative rotation) and n point pairs: most are projections of one
    3-D point into both cameras (+ pixel noise), the rest are wrong associations, far points (no parallax) and
    points behind a camera, so that every return code of TriangulateMatches occurs."""
    rng = np.random.default_rng(seed)
    w = rng.normal(0, 0.01, 3)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Rlr = (np.eye(3) + K + 0.5 * K @ K).astype(np.float32)     # x_l = Rlr x_r + tlr
    tlr = np.array([0.1, 0.002, -0.001], np.float32)
    z = rng.uniform(0.3, 2.2, n)   # 0.1 m baseline: beyond ~3 m the parallax test (cos > 0.9998) rejects
    z[rng.random(n) < 0.08] = rng.uniform(80, 400, int((rng.random(n) < 0.08).sum()) or 1)[0]
    Xl = np.stack([rng.uniform(-1.2, 1.2, n) * z, rng.uniform(-1.0, 1.0, n) * z, z], 1)
    Xr = (Xl - tlr[None, :].astype(np.float64)) @ Rlr.astype(np.float64)   # Rlr^T (x_l - tlr)
    xy1 = kb8_project64(KB8_CAM, Xl) + rng.normal(0, noise, (n, 2))
    xy2 = kb8_project64(KB8_CAM, Xr) + rng.normal(0, noise, (n, 2))
    wrong = rng.random(n) < 0.12
    xy2[wrong] = xy2[rng.permutation(n)][wrong]
    octave1, octave2 = rng.integers(0, 8, n), rng.integers(0, 8, n)
    return dict(Rlr=Rlr, tlr=tlr, xy1=xy1.astype(np.float32), xy2=xy2.astype(np.float32), Xl=Xl, wrong=wrong,
                octave1=octave1, octave2=octave2)


def bow_match_scenario(voc, transform, n_kf, n_f, seed, two_cam=False, levelsup=4):
    """A keyframe and a frame for ORBmatcher::SearchByBoW.  voc: synth.make_vocabulary dict; transform(desc, levelsup) ->
    dict with fv_nodes / fv_offsets / fv_features (either implementation's Vocabulary.transform).  Keyframe descriptors sit
    near vocabulary centres; most frame descriptors are keyframe descriptors with 0 - 14 flipped bits (several copies of the
    same one: equal distances, failed ratio tests, claims), some are 40 - 60 bits away (around TH_LOW), the rest random.
    Angles follow one dominant rotation with outliers, so the rotation histogram removes some matches.
    -> dict(kf=side, f=side, has_point, nleft) with side = dict(fv_*, descriptors, angles)"""
    rng = np.random.default_rng(seed)
    centres = voc["descriptors"][1:]

    def near(base, lo, hi):
        out = base.copy()
        for i in range(len(out)):
            nb = int(rng.integers(lo, hi + 1))
            if nb:
                bits = np.unpackbits(out[i])
                bits[rng.choice(256, nb, replace=False)] ^= 1
                out[i] = np.packbits(bits)
        return out
    dK = near(centres[rng.integers(0, len(centres), n_kf)], 0, 20)
    src = rng.integers(0, max(n_kf, 1), n_f)
    kind = rng.random(n_f)
    if n_kf == 0:
        dK = np.zeros((0, 32), np.uint8)
    dF = np.zeros((n_f, 32), np.uint8)
    if n_kf:
        dF[kind < 0.7] = near(dK[src[kind < 0.7]], 0, 14)
    mid = (kind >= 0.7) & (kind < 0.85)
    if n_kf and mid.any():
        dF[mid] = near(dK[src[mid]], 40, 60)
    rnd = kind >= 0.85
    dF[rnd] = rng.integers(0, 256, (int(rnd.sum()), 32), dtype=np.uint8)
    dup = rng.random(n_f) < 0.1      # exact copies of another frame descriptor: ties in scan order
    if n_f:
        dF[dup] = dF[rng.integers(0, n_f, int(dup.sum()))]
    aK = rng.uniform(0, 360, n_kf).astype(np.float32)
    rot = np.float32(rng.uniform(0, 360))
    aF = (aK[src] - rot + rng.normal(0, 4, n_f)).astype(np.float32) if n_kf else np.zeros(n_f, np.float32)
    out_l = rng.random(n_f) < 0.2
    aF[out_l] = rng.uniform(0, 360, int(out_l.sum()))
    aF = np.mod(aF, np.float32(360)).astype(np.float32)
    fk, ff = transform(dK, levelsup), transform(dF, levelsup)
    side = lambda t, d, a: dict(fv_nodes=t["fv_nodes"], fv_offsets=t["fv_offsets"], fv_features=t["fv_features"], descriptors=d, angles=a)
    return dict(kf=side(fk, dK, aK), f=side(ff, dF, aF), has_point=(rng.random(n_kf) < 0.8).astype(np.uint8),
                nleft=(int(n_f * 0.55) if two_cam else -1))

