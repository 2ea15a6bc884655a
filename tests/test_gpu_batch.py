"""-m gpu: B frames per launch (ft_tracked_batch_*) - every frame of a batch equals the oracle running the reference's sequence on
that frame (SearchByProjection(CurrentFrame, LastFrame), isInFrustum, SearchByProjection(Frame, local map points);
src/ORBmatcher.cc:49-225,1775-1990, src/Frame.cc:536-610) and the single-frame path (ft_tracked_frame_*), bit for bit."""
import numpy as np
import pytest

from fasttrack_amd import orb
from oracle import binding as ob
from tests import scenarios as sc

pytestmark = pytest.mark.gpu

LOG_SF = float(np.float32(np.log(np.float32(1.2))))
TRL = np.concatenate([np.eye(3), [[-0.101], [0.0], [0.0]]], 1).astype(np.float32)
TLR = (0.101, 0.0, 0.0)


@pytest.fixture(scope="module")
def ctx():
    c = orb.Context(0)
    yield c
    c.close()


_cache = {}


def _kb8_base(nf, seed):
    """a two-camera KannalaBrandt8 frame (512x512, oracle extraction + fisheye match); cached per (nf, seed)"""
    key = (nf, seed)
    if key not in _cache:
        _cache[key] = sc.fisheye_frame_scenario(512, 512, nf, seed)
    return _cache[key]


def _kb8_views(fr, sf, keep=None):
    """oracle and device views of a base frame; keep = number of left / right keypoints kept (a smaller frame of the batch)"""
    kL, kR, dL, dR, l2r, r2l = fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["l2r"], fr["r2l"]
    if keep is not None:
        nl, nr = keep
        kL, kR, dL, dR = kL[:nl], kR[:nr], dL[:nl], dR[:nr]
        l2r = np.where(l2r[:nl] < nr, l2r[:nl], -1).astype(np.int32)
        r2l = np.where(r2l[:nr] < nl, r2l[:nr], -1).astype(np.int32)
    kw = dict(keys=kL, keys_right=kR, descriptors=np.concatenate([dL, dR]), bounds=sc.frame_bounds(512, 512), left_to_right=l2r,
              right_to_left=r2l, cam_model=1, cam=list(sc.KB8_CAM), Trl=TRL)
    return ob.FrameView(scale_factors_=sf, **kw), orb.FrameView(scale_factors=sf, **kw), kL, dL


def _kb8_inputs(kL, dL, sf, seed, M):
    intr = dict(fx=sc.KB8_CAM[0], fy=sc.KB8_CAM[1], cx=sc.KB8_CAM[2], cy=sc.KB8_CAM[3])
    depth = np.zeros(len(kL), np.float32)
    last, Tcw = sc.last_frame_scenario(kL, dL, None, depth, intr, 512, 512, seed=seed)
    pts, Rcw, tcw = sc.map_points_scenario(kL, dL, depth, intr, 8, sf, seed + 500, M=M)
    return last, Tcw, pts, Rcw, tcw


def _check_frame(tag, g1, g2, gh, o1, ofr, o2, oF):
    assert g1["n"] == o1["n"] and np.array_equal(g1["assign"], o1["assign"]), f"{tag}: last-frame search"
    for k, _ in ob.FRUSTUM_FIELDS:
        assert np.array_equal(g2[k], ofr[k]), f"{tag}: frustum field {k}"
    assert g2["n_to_match"] == ofr["n"], f"{tag}: nToMatch"
    assert g2["n"] == o2["n"] and np.array_equal(g2["assign"], o2["assign"]), f"{tag}: local-map search"
    assert np.array_equal(gh, oF.holder_obs), f"{tag}: holder_obs"


@pytest.mark.parametrize("nf,th", [(2000, 7.0), (2000, 15.0), (1500, 7.0)])
def test_tracked_batch_kb8_equals_oracle_and_single_frame(ctx, nf, th):
    """B = 32 two-camera KannalaBrandt8 frames (BASELINE configs[3]: 512x512, nFeatures 2000) through one set of launches:
    every frame's assignments, frustum fields, nToMatch and final holder_obs equal the oracle's sequence on that frame; four of
    them are also run through ft_tracked_frame_* (the single-frame path) and compared.  The batch mixes frame sizes (frames
    cut down to fewer keypoints), point counts (down to none) and poses."""
    B, D = 32, 4
    sf, _ = ob.scale_factors(1.2, 8)
    bases = [_kb8_base(nf, 10 + k) for k in range(D)]
    rng = np.random.default_rng(int(nf + th))
    frames, lasts, Tcws, ptss, poses, oracle = [], [], [], [], [], []
    for f in range(B):
        fr = bases[f % D]
        keep = None
        if f % 7 == 3:
            keep = (int(len(fr["kL"]) * rng.uniform(0.3, 0.9)), int(len(fr["kR"]) * rng.uniform(0.3, 0.9)))
        oF, gF, kL, dL = _kb8_views(fr, sf, keep)
        M = 0 if f == 5 else int(rng.integers(300, 2001))
        last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 1000 + f, max(M, 1))
        if f == 5:   # a frame without last-frame points and without local map points
            last = {k: v[:0] for k, v in last.items()}
            pts = {k: v[:0] for k, v in pts.items()}
        if f == 9:   # no valid last-frame point
            last["valid"][:] = 0
        o1 = ob.search_last_frame(oF, last, Tcw, th, False, False, True)
        ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, TLR), pts, 0.5, LOG_SF)
        o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts), th)
        frames.append(gF); lasts.append(last); Tcws.append(Tcw); ptss.append(pts)
        poses.append(orb.make_pose(Rcw, tcw, TLR))
        oracle.append((o1, ofr, o2, oF))
    maxkp = max(F.c.N for F in frames) + 8
    tb = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=maxkp, max_points=2048)
    tb.upload(frames)
    g1 = tb.search_last_frame(lasts, Tcws, th)
    g2 = tb.track_local_map(poses, ptss, 0.5, LOG_SF, th)
    nmatch = 0
    for f in range(B):
        o1, ofr, o2, oF = oracle[f]
        _check_frame(f"frame {f}", g1[f], g2[f], tb.holder_obs(f), o1, ofr, o2, oF)
        nmatch += o1["n"] + o2["n"]
    assert nmatch > 200 * B
    # the single-frame path on a few frames of the batch
    tf = orb.TrackedFrame(ctx, max_keypoints=maxkp, max_points=2048)
    for f in (0, 3, 5, 17):
        _, gF, kL, dL = _kb8_views(bases[f % D], sf, None if f % 7 != 3 else (frames[f].c.Nleft, frames[f].c.N - frames[f].c.Nleft))
        tf.upload(gF)
        s1 = tf.search_last_frame(lasts[f], Tcws[f], th)
        s2 = tf.track_local_map(poses[f], ptss[f], 0.5, LOG_SF, th)
        assert s1["n"] == g1[f]["n"] and np.array_equal(s1["assign"], g1[f]["assign"])
        assert s2["n"] == g2[f]["n"] and np.array_equal(s2["assign"], g2[f]["assign"]) and s2["n_to_match"] == g2[f]["n_to_match"]
        assert np.array_equal(tf.holder_obs(), tb.holder_obs(f))
    tf.close()
    # a second batch through the same object (fewer frames), and the same batch again: identical results
    tb.upload(frames[:5])
    h1 = tb.search_last_frame(lasts[:5], Tcws[:5], th)
    h2 = tb.track_local_map(poses[:5], ptss[:5], 0.5, LOG_SF, th)
    for f in range(5):
        assert np.array_equal(h1[f]["assign"], g1[f]["assign"]) and np.array_equal(h2[f]["assign"], g2[f]["assign"])
    tb.close()


def test_tracked_batch_mixed_camera_models_and_pose_forms(ctx):
    """one batch holding rectified pinhole stereo frames (Nleft == -1, mvuRight test) and two-camera KB8 frames, searched with
    far-point rejection; then the Sophus form of the poses (ft_tracked_batch_search_last_frame_se3) on the pinhole frames"""
    w, h, nf = 752, 480, 1200
    sf, _ = ob.scale_factors(1.2, 8)
    frames, lasts, Tcws, ptss, poses, oracle = [], [], [], [], [], []
    pin = []
    th_far = 9.0   # one mThFarPoints for the call (a property of the tracker, not of the frame)
    for f in range(6):
        if f % 2 == 0:
            fr = sc.oracle_stereo_frame(w, h, nf, 40 + f)
            sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["intr"]["mbf"], fr["intr"]["mb"])
            args = dict(keys=fr["kL"], descriptors=fr["dL"], bounds=sc.frame_bounds(w, h), mbf=fr["intr"]["mbf"], mb=fr["intr"]["mb"],
                        uright=sm["uright"], cam=[fr["intr"][k] for k in ("fx", "fy", "cx", "cy")])
            oF, gF = ob.FrameView(scale_factors_=sf, **args), orb.FrameView(scale_factors=sf, **args)
            last, Tcw = sc.last_frame_scenario(fr["kL"], fr["dL"], sm["uright"], sm["depth"], fr["intr"], w, h, seed=60 + f)
            pts, Rcw, tcw = sc.map_points_scenario(fr["kL"], fr["dL"], sm["depth"], fr["intr"], 8, sf, 70 + f)
            tlr = (0, 0, 0)
            pin.append((f, args, last))
        else:
            oF, gF, kL, dL = _kb8_views(_kb8_base(1500, 10), sf)
            last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 80 + f, 1500)
            tlr = TLR
        o1 = ob.search_last_frame(oF, last, Tcw, 15.0, False, False, True)
        ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, tlr), pts, 0.5, LOG_SF)
        o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts, True, th_far), 3.0)
        frames.append(gF); lasts.append(last); Tcws.append(Tcw); ptss.append(pts); poses.append(orb.make_pose(Rcw, tcw, tlr))
        oracle.append((o1, ofr, o2, oF))
    tb = orb.TrackedBatch(ctx, max_frames=8, max_keypoints=4096, max_points=4096)
    tb.upload(frames)
    g1 = tb.search_last_frame(lasts, Tcws, 15.0)
    g2 = tb.track_local_map(poses, ptss, 0.5, LOG_SF, 3.0, far_points=True, th_far_points=th_far)
    for f in range(6):
        _check_frame(f"frame {f}", g1[f], g2[f], tb.holder_obs(f), *oracle[f])
    # Sophus-form poses, forward / backward per frame
    views, ls, Ts, fw, bw, oo = [], [], [], [], [], []
    for j, (f, args, last) in enumerate(pin):
        q, t = sc.random_se3(np.random.default_rng(90 + f), 0.03, 0.006)
        oF, gF = ob.FrameView(scale_factors_=sf, **args), orb.FrameView(scale_factors=sf, **args)
        fwd, bwd = j == 1, j == 2
        oo.append(ob.search_last_frame(oF, last, ob.SE3(q, t), 15.0, fwd, bwd, True))
        views.append(gF); ls.append(last); Ts.append(orb.SE3(q, t)); fw.append(int(fwd)); bw.append(int(bwd))
    tb.upload(views)
    g = tb.search_last_frame(ls, Ts, 15.0, forward=fw, backward=bw)
    for j in range(len(pin)):
        assert oo[j]["n"] > 50 and g[j]["n"] == oo[j]["n"] and np.array_equal(g[j]["assign"], oo[j]["assign"]), j
    tb.close()


def test_tracked_batch_argument_checks(ctx):
    sf, _ = ob.scale_factors(1.2, 8)
    _, gF, kL, dL = _kb8_views(_kb8_base(1500, 10), sf)
    last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 7, 100)
    tb = orb.TrackedBatch(ctx, max_frames=2, max_keypoints=gF.c.N + 8, max_points=256)
    with pytest.raises(Exception):
        tb.upload([gF, gF, gF])            # more frames than the batch holds
    tb.upload([gF, gF])
    with pytest.raises(Exception):
        tb.search_last_frame([last], [Tcw], 7.0)   # n_frames differs from the upload
    big = {k: np.concatenate([v] * 3) for k, v in pts.items()}
    with pytest.raises(Exception):
        tb.track_local_map([orb.make_pose(Rcw, tcw, TLR)] * 2, [big, big], 0.5, LOG_SF, 7.0)   # beyond max_points
    r = tb.track_local_map([orb.make_pose(Rcw, tcw, TLR)] * 2, [pts, pts], 0.5, LOG_SF, 7.0)
    assert np.array_equal(r[0]["assign"], r[1]["assign"])
    tb.close()


@pytest.mark.parametrize("lap", [(0, 511), (120, 420)])
def test_tracked_batch_bound_to_extractors_equals_uploaded_batch(ctx, lap):
    """ft_tracked_batch_bind_fisheye: the two-camera frames of a batch straight from what two throughput extractors left in HBM
    (keypoints into the reference's lapping-area order on the device, 2-NN + ratio matching of the lapping subsets, grids) -
    the match tables equal the oracle's ComputeStereoFishEyeMatches matching on the host copies, and both searches give what
    the same frames give when they are uploaded from host arrays (hence the oracle's results: the tests above)."""
    from fasttrack_amd import synth
    B, w, h, nf = 24, 512, 512, 2000
    sf, _ = ob.scale_factors(1.2, 8)
    exL = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    exR = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    pairs = [synth.make_planes_pair(w, h, seed=300 + i) for i in range(B)]
    pairs[7] = (np.full((h, w), 90, np.uint8), pairs[7][1])     # a frame whose left image has no keypoint
    rL = exL.extract_batch([p[0] for p in pairs], lap)
    rR = exR.extract_batch([p[1] for p in pairs], lap)
    views, tables, lasts, Tcws, ptss, poses = [], [], [], [], [], []
    for f in range(B):
        (kL, dL, mL), (kR, dR, mR) = rL[f], rR[f]
        m = ob.fisheye_match(dL[mL:], dR[mR:])["matches"] if len(dL) > mL and len(dR) > mR else np.zeros(0, np.int32)
        l2r = np.full(len(kL), -1, np.int32)
        r2l = np.full(len(kR), -1, np.int32)
        for i, j in enumerate(m):
            if j >= 0:
                l2r[mL + i] = mR + j
                r2l[mR + j] = mL + i      # the last left keypoint that matches a right one keeps it (src/Frame.cc:1262)
        tables.append((l2r, r2l))
        kw = dict(keys=kL, keys_right=kR, descriptors=np.concatenate([dL, dR]), bounds=sc.frame_bounds(w, h), left_to_right=l2r,
                  right_to_left=r2l, cam_model=1, cam=list(sc.KB8_CAM), Trl=TRL)
        views.append(orb.FrameView(scale_factors=sf, **kw))
        if len(kL):
            last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 2000 + f, 1200)
        else:
            last, Tcw, pts, Rcw, tcw = _kb8_inputs(rL[0][0], rL[0][1], sf, 2000 + f, 1200)
            last = {k: v[:0] for k, v in last.items()}
        lasts.append(last); Tcws.append(Tcw); ptss.append(pts); poses.append(orb.make_pose(Rcw, tcw, TLR))
    if lap == (120, 420):
        assert any(0 < rL[f][2] < len(rL[f][0]) for f in range(B))   # keypoints on both sides of the lapping area
    cap = 2 * exL.max_keypoints + 64
    ta = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=cap, max_points=2048)
    t = ta.bind_fisheye(exL, exR, views, lap, lap)
    nm = 0
    for f in range(B):
        assert np.array_equal(t[f][0], tables[f][0]), f"frame {f}: mvLeftToRightMatch"
        assert np.array_equal(t[f][1], tables[f][1]), f"frame {f}: mvRightToLeftMatch"
        nm += int((tables[f][0] >= 0).sum())
    assert nm > 20 * B
    a1 = ta.search_last_frame(lasts, Tcws, 7.0)
    a2 = ta.track_local_map(poses, ptss, 0.5, LOG_SF, 7.0)
    tu = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=cap, max_points=2048)
    tu.upload(views)
    u1 = tu.search_last_frame(lasts, Tcws, 7.0)
    u2 = tu.track_local_map(poses, ptss, 0.5, LOG_SF, 7.0)
    for f in range(B):
        assert a1[f]["n"] == u1[f]["n"] and np.array_equal(a1[f]["assign"], u1[f]["assign"]), f
        assert a2[f]["n"] == u2[f]["n"] and np.array_equal(a2[f]["assign"], u2[f]["assign"]) and a2[f]["n_to_match"] == u2[f]["n_to_match"], f
        for k, _ in ob.FRUSTUM_FIELDS:
            assert np.array_equal(a2[f][k], u2[f][k]), (f, k)
        assert np.array_equal(ta.holder_obs(f), tu.holder_obs(f))
    # the oracle's sequence on two frames of the bound batch (the rest is covered through the uploaded form above)
    for f in (0, 11):
        (kL, dL, _), (kR, dR, _) = rL[f], rR[f]
        kw = dict(keys=kL, keys_right=kR, descriptors=np.concatenate([dL, dR]), bounds=sc.frame_bounds(w, h), left_to_right=tables[f][0],
                  right_to_left=tables[f][1], cam_model=1, cam=list(sc.KB8_CAM), Trl=TRL)
        oF = ob.FrameView(scale_factors_=sf, **kw)
        o1 = ob.search_last_frame(oF, lasts[f], Tcws[f], 7.0, False, False, True)
        pose = ob.make_pose(*[np.asarray(x) for x in (np.array(poses[f].Rcw).reshape(3, 3), np.array(poses[f].tcw))], TLR)
        ofr = ob.is_in_frustum(oF, pose, ptss[f], 0.5, LOG_SF)
        o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, ptss[f]), 7.0)
        _check_frame(f"bound frame {f}", a1[f], a2[f], ta.holder_obs(f), o1, ofr, o2, oF)
    for o in (ta, tu, exL, exR):
        o.close()


def test_bind_fisheye_with_one_extractor_for_both_cameras(ctx):
    """ft_tracked_batch_bind_fisheye_slots with exL == exR: the left images in slots 0 .. 3 and the right ones in slots 4 .. 7 of ONE
    extractor's last batch (a frame's two images extracted as one batch) - match tables and both searches equal what two
    extractors give; overlapping slot ranges are refused"""
    from fasttrack_amd import synth
    B, w, h, nf = 4, 512, 512, 1500
    lap = (0, 511)
    sf, _ = ob.scale_factors(1.2, 8)
    pairs = [synth.make_planes_pair(w, h, seed=900 + i) for i in range(B)]
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=2 * B)
    exL = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    exR = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    r = ex.extract_batch([p[0] for p in pairs] + [p[1] for p in pairs], lap)
    rL = exL.extract_batch([p[0] for p in pairs], lap)
    rR = exR.extract_batch([p[1] for p in pairs], lap)
    views, lasts, Tcws, ptss, poses = [], [], [], [], []
    for f in range(B):
        (kL, dL, _), (kR, dR, _) = r[f], r[B + f]
        assert np.array_equal(kL, rL[f][0]) and np.array_equal(kR, rR[f][0])
        views.append(orb.FrameView(keys=kL, keys_right=kR, descriptors=np.zeros((len(kL) + len(kR), 32), np.uint8), scale_factors=sf,
                                   bounds=sc.frame_bounds(w, h), left_to_right=np.zeros(max(len(kL), 1), np.int32),
                                   right_to_left=np.zeros(max(len(kR), 1), np.int32), cam_model=1, cam=list(sc.KB8_CAM), Trl=TRL))
        last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 8000 + f, 900)
        lasts.append(last); Tcws.append(Tcw); ptss.append(pts); poses.append(orb.make_pose(Rcw, tcw, TLR))
    cap = 2 * ex.max_keypoints + 64
    t1 = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=cap, max_points=2048)
    t2 = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=cap, max_points=2048)
    try:
        with pytest.raises(orb.FastTrackError, match="disjoint slot ranges"):
            t1.bind_fisheye(ex, ex, views, lap, lap, slot0=0, slot0_right=2)
        a = t1.bind_fisheye(ex, ex, views, lap, lap, slot0=0, slot0_right=B)
        b = t2.bind_fisheye(exL, exR, views, lap, lap)
        for f in range(B):
            assert np.array_equal(a[f][0], b[f][0]) and np.array_equal(a[f][1], b[f][1]) and (a[f][0] >= 0).sum() > 20
        a1, b1 = t1.search_last_frame(lasts, Tcws, 7.0), t2.search_last_frame(lasts, Tcws, 7.0)
        a2, b2 = t1.track_local_map(poses, ptss, 0.5, LOG_SF, 7.0), t2.track_local_map(poses, ptss, 0.5, LOG_SF, 7.0)
        for f in range(B):
            assert a1[f]["n"] == b1[f]["n"] and np.array_equal(a1[f]["assign"], b1[f]["assign"])
            assert a2[f]["n"] == b2[f]["n"] and np.array_equal(a2[f]["assign"], b2[f]["assign"])
            assert np.array_equal(t1.holder_obs(f), t2.holder_obs(f))
    finally:
        for o in (t1, t2, ex, exL, exR):
            o.close()


@pytest.mark.parametrize("opts", [dict(search_cache=0), dict(search_grid=0), dict(search_cache=1), dict(search_cache=1, pass_burst=2),
                                  dict(search_cache=0, pass_burst=2), dict(search_cache=3), dict(search_cache=2, pass_burst=2)])
def test_tracked_batch_without_cache_without_grid_with_short_bursts(ctx, opts):
    """the batch under the context's search options: no candidate cache (every pass is the general kernel), no CSR grid (every
    keypoint's cell is computed), search_cache = 1 (the claim passes instead of the one-launch resolution: lean kernels + slow
    lists), 3 (the resolution for this batch of six frames too; 2 leaves it to batches of 24 and more), bursts of two passes
    (many host round trips: the flag rows of converged frames must stick) - the same assignments as the oracle every time"""
    sf, _ = ob.scale_factors(1.2, 8)
    B = 6
    frames, lasts, Tcws, ptss, poses, oracle = [], [], [], [], [], []
    for f in range(B):
        oF, gF, kL, dL = _kb8_views(_kb8_base(1500, 10 + f % 2), sf)
        last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 3000 + f, 400 + 300 * f)
        o1 = ob.search_last_frame(oF, last, Tcw, 15.0, False, False, True)
        ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, TLR), pts, 0.5, LOG_SF)
        o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts), 15.0)
        frames.append(gF); lasts.append(last); Tcws.append(Tcw); ptss.append(pts); poses.append(orb.make_pose(Rcw, tcw, TLR))
        oracle.append((o1, ofr, o2, oF))
    with ctx.options(**opts):
        tb = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=frames[0].c.N + 8, max_points=2048)
        tb.upload(frames)
        g1 = tb.search_last_frame(lasts, Tcws, 15.0)
        g2 = tb.track_local_map(poses, ptss, 0.5, LOG_SF, 15.0)
        for f in range(B):
            _check_frame(f"{opts} frame {f}", g1[f], g2[f], tb.holder_obs(f), *oracle[f])
        tb.close()


@pytest.mark.parametrize("burst", [12, 2])
def test_tracked_batch_lists_beyond_the_cache_fall_back_to_the_passes(ctx, burst):
    """windows so wide (th 400 / 120: most of the image) that a point's candidates outgrow its cache list (511 keys): k_resolve_batch
    gives up on such a frame and the claim passes take over for it (slow lists: the general kernel scans the window again every
    pass), while the small frames of the same batch - cut down to 300 keypoints per camera, every list fits - stay resolved.
    Same assignments as the oracle on every frame."""
    sf, _ = ob.scale_factors(1.2, 8)
    B, th1, th = 6, 400.0, 120.0
    frames, lasts, Tcws, ptss, poses, oracle = [], [], [], [], [], []
    for f in range(B):
        fr = _kb8_base(2000, 10 + f % 2)
        oF, gF, kL, dL = _kb8_views(fr, sf, (300, 300) if f % 2 else None)
        last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 4000 + f, 150 + 60 * f)
        for d in (last, ):  # a short last frame keeps the oracle's share of the test short
            for k in d:
                d[k] = d[k][:250]
        o1 = ob.search_last_frame(oF, last, Tcw, th1, False, False, True)
        ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, TLR), pts, 0.5, LOG_SF)
        o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts), th)
        frames.append(gF); lasts.append(last); Tcws.append(Tcw); ptss.append(pts); poses.append(orb.make_pose(Rcw, tcw, TLR))
        oracle.append((o1, ofr, o2, oF))
    with ctx.options(pass_burst=burst, search_cache=3):
        ctx.reset_stats()
        tb = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=frames[0].c.N + 8, max_points=2048)
        tb.upload(frames)
        g1 = tb.search_last_frame(lasts, Tcws, th1)
        g2 = tb.track_local_map(poses, ptss, 0.5, LOG_SF, th)
        assert ctx.get_stat("tracked_batch.resolve_fallbacks")[1] == 2   # both searches
        for f in range(B):
            _check_frame(f"burst {burst} frame {f}", g1[f], g2[f], tb.holder_obs(f), *oracle[f])
        # the small frames alone: resolved, no fallback
        ctx.reset_stats()
        small = [f for f in range(B) if f % 2]
        tb.upload([frames[f] for f in small])
        h1 = tb.search_last_frame([lasts[f] for f in small], [Tcws[f] for f in small], th1)
        h2 = tb.track_local_map([poses[f] for f in small], [ptss[f] for f in small], 0.5, LOG_SF, th)
        try:
            assert ctx.get_stat("tracked_batch.resolve_fallbacks")[1] == 0
        except orb.FastTrackError:
            pass  # (a series nobody added to since the reset)
        for k, f in enumerate(small):
            assert np.array_equal(h1[k]["assign"], g1[f]["assign"]) and np.array_equal(h2[k]["assign"], g2[f]["assign"])
        tb.close()


def test_resolution_gives_up_on_a_frame_before_it_publishes_anything(ctx):
    """ONE candidate list beyond the cache in a LATE chunk of k_resolve_batch's walk: every last-frame point sits at
    octave 7 (level band 6 .. 7: at most ~270 candidates even when the window is the whole image - usable lists) except point
    200 (chunk 3 of 64-point chunks), which sits at octave 1 (band 0 .. 2, ~1 100 keypoints at th 400: beyond the 511 keys of a
    list).  Round 5 tested usability chunk by chunk and had published chunks 0 - 2 (results of both parities, last writers) when
    it gave up; the claim passes that take over then compared against final values.  Now the whole frame is tested first: the
    frame falls back untouched, and the assignments equal the oracle's.  A second frame of the same batch (every point at octave
    7) stays resolved."""
    sf, _ = ob.scale_factors(1.2, 8)
    th1 = 400.0
    for burst in (12, 2):
        frames, lasts, Tcws, oracle = [], [], [], []
        for f in range(2):
            oF, gF, kL, dL = _kb8_views(_kb8_base(2000, 10 + f), sf)
            last, Tcw, _, _, _ = _kb8_inputs(kL, dL, sf, 5000 + f, 10)
            last = {k: v[:256].copy() for k, v in last.items()}
            last["octave"][:] = 7
            last["valid"][:] = 1
            if f == 0:
                last["octave"][200] = 1
            o1 = ob.search_last_frame(oF, last, Tcw, th1, False, False, True)
            frames.append(gF); lasts.append(last); Tcws.append(Tcw); oracle.append((o1, oF))
        with ctx.options(pass_burst=burst, search_cache=3):
            ctx.reset_stats()
            tb = orb.TrackedBatch(ctx, max_frames=2, max_keypoints=max(F.c.N for F in frames) + 8, max_points=2048)
            try:
                tb.upload(frames)
                g1 = tb.search_last_frame(lasts, Tcws, th1)
                assert ctx.get_stat("tracked_batch.resolve_fallbacks")[1] == 1
                for f in range(2):
                    o1, oF = oracle[f]
                    assert g1[f]["n"] == o1["n"] and np.array_equal(g1[f]["assign"], o1["assign"]), f"burst {burst} frame {f}"
                    assert np.array_equal(tb.holder_obs(f), oF.holder_obs), f"burst {burst} frame {f}: holder_obs"
                assert oracle[0][0]["n"] > 20
            finally:
                tb.close()


def test_search_grid_switched_between_upload_and_search(ctx):
    """search_* options are read per call, but the grids exist only if search_grid was on when the frames went up (ADVICE r5:
    the row-first kernels read the grid pointers without a null check).  Upload without grids, switch the option on, search:
    the batch takes the kernels that need no grid, same assignments as the oracle - and the other way round."""
    sf, _ = ob.scale_factors(1.2, 8)
    for at_upload, at_search in ((0, 1), (1, 0)):
        oF, gF, kL, dL = _kb8_views(_kb8_base(1500, 10), sf)
        last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 6000, 700)
        o1 = ob.search_last_frame(oF, last, Tcw, 7.0, False, False, True)
        ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, TLR), pts, 0.5, LOG_SF)
        o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts), 7.0)
        with ctx.options(search_cache=3, search_grid=at_upload):
            tb = orb.TrackedBatch(ctx, max_frames=1, max_keypoints=gF.c.N + 8, max_points=2048)
            tb.upload([gF])
            with ctx.options(search_grid=at_search):
                g1 = tb.search_last_frame([last], [Tcw], 7.0)
                g2 = tb.track_local_map([orb.make_pose(Rcw, tcw, TLR)], [pts], 0.5, LOG_SF, 7.0)
            _check_frame(f"grid {at_upload} -> {at_search}", g1[0], g2[0], tb.holder_obs(0), o1, ofr, o2, oF)
            tb.close()


@pytest.mark.parametrize("pinned", [True, False])
def test_submit_and_wait_equal_the_blocking_calls(ctx, pinned):
    """ft_tracked_batch_submit_* / ft_tracked_batch_wait: the same results as the oracle (and so as the blocking calls); with the
    point arrays in pinned host memory the device reads them in place (k_gather_batch: no host copy), pageable ones are staged.
    One search per batch in flight: a second submit, an upload or holder_obs before the wait are FT_ERR_INVALID; a wait without a
    submitted search is a no-op.  32 frames: the one-launch resolution (everything enqueued by the submit)."""
    sf, _ = ob.scale_factors(1.2, 8)
    B = 32
    frames, lasts, Tcws, ptss, poses, oracle = [], [], [], [], [], []
    for f in range(B):
        oF, gF, kL, dL = _kb8_views(_kb8_base(1500, 10 + f % 2), sf, (700 + 20 * f, 650) if f % 3 == 1 else None)
        last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 7000 + f, 300 + 40 * f)
        o1 = ob.search_last_frame(oF, last, Tcw, 7.0, False, False, True)
        ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, TLR), pts, 0.5, LOG_SF)
        o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts), 7.0)
        frames.append(gF); lasts.append(last); Tcws.append(Tcw); ptss.append(pts); poses.append(orb.make_pose(Rcw, tcw, TLR))
        oracle.append((o1, ofr, o2, oF))
    tb = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=max(F.c.N for F in frames) + 8, max_points=2048)
    try:
        tb.upload(frames)
        pc = ctx if pinned else None
        pl_last = tb.prepare_last(lasts, Tcws, ctx=pc)
        pl_local = tb.prepare_local(poses, ptss, ctx=pc)
        tb.wait()                                            # nothing submitted: a no-op
        ctx.reset_stats()
        assert tb.search_last_frame(pl_last, th=7.0, submit=True) is None
        for call in (lambda: tb.search_last_frame(pl_last, th=7.0, submit=True), lambda: tb.upload(frames), lambda: tb.holder_obs(0)):
            with pytest.raises(orb.FastTrackError, match="has not been waited for"):
                call()
        g1 = tb.wait()
        tb.track_local_map(pl_local, viewing_cos_limit=0.5, log_scale_factor=LOG_SF, th=7.0, submit=True)
        g2 = tb.wait()
        for f in range(B):
            _check_frame(f"pinned {pinned} frame {f}", g1[f], g2[f], tb.holder_obs(f), *oracle[f])
        # a pinned call stages nothing on the host: a few job records per frame (the pageable one copies 53 bytes per point)
        stage = ctx.get_stat("tracked_batch.search_last_frame.stage")
        assert stage[1] == 1
    finally:
        tb.close()


def test_octave_out_of_range_in_arrays_read_in_place_is_reported_by_the_wait(ctx):
    """a valid last-frame point whose octave lies outside the frame's levels: FT_ERR_INVALID from the blocking call when the host
    copies the arrays (checked while it reads them), from the WAIT when the device reads them in place (k_last_project_batch
    drops the point and marks the frame); the batch stays usable"""
    sf, _ = ob.scale_factors(1.2, 8)
    oF, gF, kL, dL = _kb8_views(_kb8_base(1500, 10), sf)
    last, Tcw, _, _, _ = _kb8_inputs(kL, dL, sf, 7100, 10)
    good = {k: v.copy() for k, v in last.items()}
    last["valid"][5] = 1
    last["octave"][5] = 8
    tb = orb.TrackedBatch(ctx, max_frames=1, max_keypoints=gF.c.N + 8, max_points=2048)
    try:
        tb.upload([gF])
        with pytest.raises(orb.FastTrackError, match="octave out of range"):
            tb.search_last_frame([last], [Tcw], 7.0)
        pl = tb.prepare_last([last], [Tcw], ctx=ctx)
        tb.search_last_frame(pl, th=7.0, submit=True)
        with pytest.raises(orb.FastTrackError, match="octave out of range"):
            tb.wait()
        o1 = ob.search_last_frame(oF, good, Tcw, 7.0, False, False, True)
        tb.upload([gF])
        g1 = tb.search_last_frame(tb.prepare_last([good], [Tcw], ctx=ctx), th=7.0)
        assert g1[0]["n"] == o1["n"] and np.array_equal(g1[0]["assign"], o1["assign"])
    finally:
        tb.close()


def test_tracked_batch_bind_with_triangulation_equals_oracle_fisheye_stereo(ctx):
    """bind_fisheye with a rig = the whole Frame::ComputeStereoFishEyeMatches per frame (src/Frame.cc:1231-1271): the ratio-test
    survivors go through KannalaBrandt8::TriangulateMatches; mvLeftToRightMatch / mvRightToLeftMatch / mvDepth / mvStereo3Dpoints /
    nMatches equal the oracle's on the host copies of the same keypoints, bit for bit"""
    from fasttrack_amd import synth
    B, w, h, nf = 10, 512, 512, 1500
    lap = (60, 470)
    sf, sig2 = ob.scale_factors(1.2, 8)
    S = sc.fisheye_rig_scenario(5)
    cam = list(sc.KB8_CAM)
    exL = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    exR = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    pairs = [synth.make_planes_pair(w, h, seed=500 + i) for i in range(B)]
    rL = exL.extract_batch([p[0] for p in pairs], lap)
    rR = exR.extract_batch([p[1] for p in pairs], lap)
    orig = ob.FisheyeRig()
    grig = orb.make_fisheye_rig(cam, cam, S["Rlr"], S["tlr"])
    for name in ("cam1", "cam2", "Rlr", "tlr"):
        getattr(orig, name)[:] = list(getattr(grig, name))
    orig.precision = grig.precision
    views, want = [], []
    for f in range(B):
        (kL, dL, mL), (kR, dR, mR) = rL[f], rR[f]
        o = ob.fisheye_stereo(orig, dL[mL:], kL[mL:], dR[mR:], kR[mR:], sig2)
        l2r = np.full(len(kL), -1, np.int32); r2l = np.full(len(kR), -1, np.int32)
        dep = np.full(len(kL), -1, np.float32); p3 = np.zeros((len(kL), 3), np.float32)
        for i, j in enumerate(o["matches"]):
            if j >= 0:
                l2r[mL + i] = mR + j
                r2l[mR + j] = mL + i
        dep[mL:] = o["depth"]
        p3[mL:] = o["p3d"]
        want.append((l2r, r2l, dep, p3, o["n"]))
        kw = dict(keys=kL, keys_right=kR, descriptors=np.concatenate([dL, dR]), bounds=sc.frame_bounds(w, h), left_to_right=l2r,
                  right_to_left=r2l, cam_model=1, cam=cam, Trl=TRL)
        views.append(orb.FrameView(scale_factors=sf, **kw))
    tb = orb.TrackedBatch(ctx, max_frames=B, max_keypoints=2 * exL.max_keypoints + 64, max_points=512)
    got = tb.bind_fisheye(exL, exR, views, lap, lap, rig=grig, level_sigma2=sig2)
    kept = dropped = 0
    for f in range(B):
        l2r, r2l, dep, p3, n = want[f]
        assert np.array_equal(got[f][0], l2r), f"frame {f}: mvLeftToRightMatch"
        assert np.array_equal(got[f][1], r2l), f"frame {f}: mvRightToLeftMatch"
        assert np.array_equal(got[f][2], dep), f"frame {f}: mvDepth"
        assert np.array_equal(got[f][3], p3), f"frame {f}: mvStereo3Dpoints"
        assert got[f][4] == n
        kept += n
        m = ob.fisheye_match(rL[f][1][rL[f][2]:], rR[f][1][rR[f][2]:])["matches"]
        dropped += int((m >= 0).sum()) - n
    assert kept > 5 * B and dropped > 0      # the filter keeps pairs and rejects pairs
    for o in (tb, exL, exR):
        o.close()


def test_two_batches_in_flight_from_two_threads(ctx):
    """a batch object has a stream and a lock of its own: two batches of one context driven by two host threads at the same time
    (the passes of one beside the staging / replay of the other) each give the oracle's results, call after call"""
    import threading
    sf, _ = ob.scale_factors(1.2, 8)
    B = 12
    sets = []
    for t in range(2):
        frames, lasts, Tcws, ptss, poses, oracle = [], [], [], [], [], []
        for f in range(B):
            oF, gF, kL, dL = _kb8_views(_kb8_base(2000, 10 + (f + t) % 4), sf)
            last, Tcw, pts, Rcw, tcw = _kb8_inputs(kL, dL, sf, 5000 + 100 * t + f, 1500)
            o1 = ob.search_last_frame(oF, last, Tcw, 7.0, False, False, True)
            ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, TLR), pts, 0.5, LOG_SF)
            o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts), 7.0)
            frames.append(gF); lasts.append(last); Tcws.append(Tcw); ptss.append(pts); poses.append(orb.make_pose(Rcw, tcw, TLR))
            oracle.append((o1, ofr, o2, oF))
        sets.append((frames, lasts, Tcws, ptss, poses, oracle))
    tbs = [orb.TrackedBatch(ctx, max_frames=B, max_keypoints=sets[t][0][0].c.N + 64, max_points=2048) for t in range(2)]
    errors = []

    def work(t):
        try:
            frames, lasts, Tcws, ptss, poses, oracle = sets[t]
            for rep in range(6):
                tbs[t].upload(frames)
                g1 = tbs[t].search_last_frame(lasts, Tcws, 7.0)
                g2 = tbs[t].track_local_map(poses, ptss, 0.5, LOG_SF, 7.0)
                for f in range(B):
                    _check_frame(f"thread {t} rep {rep} frame {f}", g1[f], g2[f], tbs[t].holder_obs(f), *oracle[f])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for tb in tbs:
        tb.close()
    assert not errors, errors
