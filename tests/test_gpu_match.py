"""Parity of the HIP matchers against the oracle (needs an MI355X): stereo (row-band Hamming + SAD +
parabola + median cut), fisheye 2-NN, both projection searches, DescriptorDistance."""
import numpy as np
import pytest

from fasttrack_amd import orb, synth
from oracle import binding as ob
from tests import scenarios as sc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = orb.Context(0)
    yield c
    c.close()


def _calls(ctx, name):
    try:
        return ctx.get_stat(name)[1]
    except Exception:
        return 0


def test_descriptor_distance(ctx):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (5000, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (5000, 32), dtype=np.uint8)
    b[:100] = a[:100]
    b[100:200] = ~a[100:200]
    g = orb.KernelController.descriptor_distance(ctx, a, b)
    o = np.array([ob.descriptor_distance(a[i], b[i]) for i in range(len(a))])
    assert np.array_equal(g, o) and g[:100].max() == 0 and g[100:200].min() == 256


@pytest.mark.parametrize("w,h,nf,seed", [(752, 480, 1200, 3), (1280, 720, 2000, 4), (640, 480, 1000, 5)])
def test_stereo_match_bit_exact(ctx, w, h, nf, seed):
    fr = sc.oracle_stereo_frame(w, h, nf, seed)
    exL = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    exR = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    exL(fr["L"])
    exR(fr["R"])
    mbf, mb = fr["intr"]["mbf"], fr["intr"]["mb"]
    for cut in (False, True):
        o = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], mbf, mb, median_cut=cut)
        g = orb.KernelController.launchStereoMatchKernel(exL, exR, fr["kL"], fr["kR"], fr["dL"], fr["dR"], mbf, mb,
                                                         median_cut=cut)
        assert g["n"] == o["n"] and o["n"] > 20
        assert np.array_equal(g["uright"] >= 0, o["uright"] >= 0)
        assert np.allclose(g["uright"], o["uright"], rtol=0, atol=1e-4)
        assert np.allclose(g["depth"], o["depth"], rtol=1e-6, atol=1e-4)
        assert np.array_equal(g["uright"], o["uright"]) and np.array_equal(g["depth"], o["depth"])
        if not cut:
            assert np.array_equal(g["sad"], o["sad"])


def test_stereo_edge_cases(ctx):
    w, h, nf = 640, 480, 1000
    fr = sc.oracle_stereo_frame(w, h, nf, 8)
    exL = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    exR = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    exL(fr["L"])
    exR(fr["R"])
    mbf, mb = fr["intr"]["mbf"], fr["intr"]["mb"]
    # no right keypoints / no left keypoints
    g = orb.KernelController.launchStereoMatchKernel(exL, exR, fr["kL"], fr["kR"][:0], fr["dL"], fr["dR"][:0], mbf, mb)
    assert g["n"] == 0 and (g["uright"] == -1).all()
    g = orb.KernelController.launchStereoMatchKernel(exL, exR, fr["kL"][:0], fr["kR"], fr["dL"][:0], fr["dR"], mbf, mb)
    assert g["n"] == 0
    # identical images: zero disparity -> the reference clamps to 0.01 (Frame.cc:979-983)
    exR(fr["L"])
    fr["exR"].extract(fr["L"])
    # (with every SAD == 0 the median cut removes all matches: thDist = 0 and `dist < thDist` never holds)
    for cut, expect in ((True, 0), (False, 100)):
        o = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kL"], fr["dL"], fr["dL"], mbf, mb, median_cut=cut)
        g = orb.KernelController.launchStereoMatchKernel(exL, exR, fr["kL"], fr["kL"], fr["dL"], fr["dL"], mbf, mb,
                                                         median_cut=cut)
        assert g["n"] == o["n"] and np.array_equal(g["uright"], o["uright"]) and np.array_equal(g["depth"], o["depth"])
        assert (o["depth"] > 0).sum() >= expect
    assert np.isclose(o["uright"][o["depth"] > 0], fr["kL"]["x"][o["depth"] > 0], atol=1.0).all()


@pytest.mark.parametrize("w,h,nf,B", [(752, 480, 1200, 3), (1280, 720, 2000, 2), (640, 480, 1000, 9)])
def test_stereo_frontend_fused(ctx, w, h, nf, B):
    intr = synth.intrinsics(w, h)
    fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"])
    fe_pageable = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"], pinned_outputs=False)
    pairs = [synth.make_stereo_pair(w, h, 40 + b) for b in range(B)]
    outs = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
    outs_pg = fe_pageable.process([p[0] for p in pairs], [p[1] for p in pairs])
    devL = [ctx.to_device(p[0]) for p in pairs]
    devR = [ctx.to_device(p[1]) for p in pairs]
    outs_dev = fe.process(devL, devR, on_device=True, stride=w)
    for b in range(B):
        oL, oR = ob.Extractor(nf), ob.Extractor(nf)
        kL, dL, _ = oL.extract(pairs[b][0])
        kR, dR, _ = oR.extract(pairs[b][1])
        o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
        for out in (outs[b], outs_dev[b], outs_pg[b]):
            assert np.array_equal(out["keysL"], kL) and np.array_equal(out["keysR"], kR)
            assert np.array_equal(out["descL"], dL) and np.array_equal(out["descR"], dR)
            assert out["n"] == o["n"]
            assert np.array_equal(out["uright"], o["uright"]) and np.array_equal(out["depth"], o["depth"])


def test_stereo_frontend_submit_wait_two_in_flight(ctx):
    """bench.py's double buffering: two front ends, the next batch submitted before the previous is waited for"""
    import ctypes as C
    w, h, nf, B = 752, 480, 1200, 4
    intr = synth.intrinsics(w, h)
    fes = [orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"]) for _ in range(2)]
    batches = [[synth.make_stereo_pair(w, h, 70 + 10 * k + b) for b in range(B)] for k in range(3)]
    dev = [[(ctx.to_device(p[0]), ctx.to_device(p[1])) for p in bt] for bt in batches]
    ptrs = [((C.c_void_p * B)(*[d[0].ptr for d in dv]), (C.c_void_p * B)(*[d[1].ptr for d in dv])) for dv in dev]
    got = []
    for k in range(3):
        fes[k & 1].submit_raw(ptrs[k][0], ptrs[k][1], B, True, w)
        if k > 0:
            f = fes[(k - 1) & 1]
            f.wait()
            got.append([(f._kL[b, :f._nL[b]].copy(), f._dL[b, :f._nL[b]].copy(), f._ur[b, :f._nL[b]].copy(), int(f._nm[b])) for b in range(B)])
    f = fes[0]
    f.wait()
    got.append([(f._kL[b, :f._nL[b]].copy(), f._dL[b, :f._nL[b]].copy(), f._ur[b, :f._nL[b]].copy(), int(f._nm[b])) for b in range(B)])
    for k in range(3):
        for b in range(B):
            oL, oR = ob.Extractor(nf), ob.Extractor(nf)
            kL, dL, _ = oL.extract(batches[k][b][0])
            kR, dR, _ = oR.extract(batches[k][b][1])
            o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
            gk, gd, gu, gn = got[k][b]
            assert np.array_equal(gk, kL) and np.array_equal(gd, dL) and np.array_equal(gu, o["uright"]) and gn == o["n"]


@pytest.mark.parametrize("B", [4, 12])  # latency mode (captured graph, upload kernel) and one strided copy per batch
def test_stereo_frontend_submit_wait_pinned_ring(ctx, B):
    """The input-frame contract of submit / wait (include/fasttrack_amd.h): frames in pinned host memory are read in place
    after submit returns, so a camera ring buffer may recycle a slot once wait() has returned for the batch that used it -
    and not before.  Two front ends in flight over a ring of two slots, every slot overwritten right after its wait()."""
    import ctypes as C
    w, h, nf = 752, 480, 1200
    intr = synth.intrinsics(w, h)
    fes = [orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"]) for _ in range(2)]
    batches = [[synth.make_stereo_pair(w, h, 170 + 10 * k + b) for b in range(B)] for k in range(4)]
    ring = [(ctx.pinned_array((B, h, w), np.uint8), ctx.pinned_array((B, h, w), np.uint8)) for _ in range(2)]
    ptrs = [((C.c_void_p * B)(*[r[0].ctypes.data + b * w * h for b in range(B)]),
             (C.c_void_p * B)(*[r[1].ctypes.data + b * w * h for b in range(B)])) for r in ring]

    def fill(slot, k):
        for b in range(B):
            ring[slot][0][b], ring[slot][1][b] = batches[k][b]

    def results(f):
        return [(f._kL[b, :f._nL[b]].copy(), f._dL[b, :f._nL[b]].copy(), f._ur[b, :f._nL[b]].copy(), int(f._nm[b])) for b in range(B)]

    got = {}
    for k in range(4):
        s = k & 1
        if k >= 2:
            fes[s].wait()          # batch k - 2 is complete: its results are final ...
            got[k - 2] = results(fes[s])
        fill(s, k)                 # ... and its slot may be overwritten
        fes[s].submit_raw(ptrs[s][0], ptrs[s][1], B, False, w)
    for k in (2, 3):
        fes[k & 1].wait()
        got[k] = results(fes[k & 1])
        ring[k & 1][0][:] = 0      # overwriting after wait() changes nothing that was delivered
    for k in range(4):
        for b in range(B):
            oL, oR = ob.Extractor(nf), ob.Extractor(nf)
            kL, dL, _ = oL.extract(batches[k][b][0])
            kR, dR, _ = oR.extract(batches[k][b][1])
            o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
            gk, gd, gu, gn = got[k][b]
            assert np.array_equal(gk, kL) and np.array_equal(gd, dL) and np.array_equal(gu, o["uright"]) and gn == o["n"]


def test_stereo_frontend_device_octree_overflow(ctx):
    """a pair whose level-0 candidates exceed what k_octree sorts in LDS: wait() re-submits the batch with the
    host octree; the outputs still equal the oracle's, and the following batch runs on the device again"""
    w, h, nf, B = 1280, 720, 2000, 2
    intr = synth.intrinsics(w, h)
    fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"])

    def calls(name):
        try:
            return ctx.get_stat(name)[1]
        except Exception:
            return 0
    f0, b0 = calls("stereo.device_octree_fallbacks"), calls("stereo.device_octree_batches")
    noisy = (synth.make_noise(w, h, seed=8), synth.make_noise(w, h, seed=9))
    calm = synth.make_stereo_pair(w, h, 21)
    for k, pairs in enumerate(([noisy, calm], [calm, calm])):
        outs = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
        for (imL, imR), out in zip(pairs, outs):
            oL, oR = ob.Extractor(nf), ob.Extractor(nf)
            kL, dL, _ = oL.extract(imL)
            kR, dR, _ = oR.extract(imR)
            o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
            assert np.array_equal(out["keysL"], kL) and np.array_equal(out["keysR"], kR)
            assert np.array_equal(out["descL"], dL) and np.array_equal(out["descR"], dR)
            assert out["n"] == o["n"] and np.array_equal(out["uright"], o["uright"]) and np.array_equal(out["depth"], o["depth"])
        assert calls("stereo.device_octree_fallbacks") == f0 + 1, k
    assert calls("stereo.device_octree_batches") == b0 + 2


@pytest.mark.parametrize("hist", [True, False])
def test_dense_frames_stay_on_the_device_and_overflows_are_repaired_per_pair(ctx, hist, monkeypatch):
    """Throughput mode (more than 16 pairs per batch).  Levels with more than 4 096 candidates go to the histogram tier of the
    device octree (k_octree_hist, any count up to 65 535): dense and pure-noise frames stay on the device from the first batch
    on.  With that tier switched off (option oct_hist=0, taken when the front end is created) levels up to 16 384 candidates go to
    the sorted big tier, whose grid the host sizes from the previous batch: the first dense batch finds it absent and is
    repaired pair by pair with the host octree, the second one stays on the device, and a pair with a level beyond 16 384
    candidates is always repaired - that pair only.  Every output equals the oracle's."""
    w, h, nf, B = 752, 480, 1200, 18
    intr = synth.intrinsics(w, h)
    with ctx.options(oct_hist=int(hist)):
        fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"])

    def calls(name):
        try:
            return ctx.get_stat(name)[1]
        except Exception:
            return 0

    def check(pairs, outs, which):
        for b in which:
            (imL, imR), out = pairs[b], outs[b]
            oL, oR = ob.Extractor(nf), ob.Extractor(nf)
            kL, dL, _ = oL.extract(imL)
            kR, dR, _ = oR.extract(imR)
            o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
            assert np.array_equal(out["keysL"], kL) and np.array_equal(out["keysR"], kR), b
            assert np.array_equal(out["descL"], dL) and np.array_equal(out["descR"], dR), b
            assert out["n"] == o["n"] and np.array_equal(out["uright"], o["uright"]) and np.array_equal(out["depth"], o["depth"]), b
    dense = [synth.make_mosaic_pair(w, h, seed=60 + i, block=8) for i in range(3)]
    oex = ob.Extractor(nf)
    oex.extract(dense[0][0])
    top = max(len(oex.candidates(l)) for l in range(8))
    assert 4096 < top <= 16384, f"test input should need the second tier only ({top})"
    calm = [synth.make_stereo_pair(w, h, 70 + i) for i in range(3)]
    batch1 = [dense[i % 3] if i % 2 == 0 else calm[i % 3] for i in range(B)]  # 9 dense pairs among 18
    f0 = calls("stereo.device_octree_fallbacks")
    outs = fe.process([p[0] for p in batch1], [p[1] for p in batch1])
    r1 = 0 if hist else 9
    assert calls("stereo.device_octree_fallbacks") == f0 + r1, "first dense batch: on the device / the dense pairs, and only they, are repaired"
    check(batch1, outs, [0, 1, 2, 17])
    outs = fe.process([p[0] for p in batch1], [p[1] for p in batch1])
    assert calls("stereo.device_octree_fallbacks") == f0 + r1, "second dense batch stays on the device"
    check(batch1, outs, [0, 3, 4, 16])
    # a frame beyond the second tier (pure noise: tens of thousands of candidates at level 0) among dense and calm ones
    noisy = (synth.make_noise(w, h, seed=8), synth.make_noise(w, h, seed=9))
    batch2 = list(batch1)
    batch2[5] = noisy
    batch2[11] = (calm[0][0], noisy[1])  # only the right camera overflows
    outs = fe.process([p[0] for p in batch2], [p[1] for p in batch2])
    assert calls("stereo.device_octree_fallbacks") == f0 + (0 if hist else 11)
    check(batch2, outs, [4, 5, 6, 11])
    if hist:
        # the tier follows the demand: after a few calm batches it is not launched any more (a kernel less in the octree
        # lane), the dense batch that arrives then is repaired pair by pair - once - and the next one is back on the device
        calm_batch = [calm[i % 3] for i in range(B)]
        for _ in range(3):
            fe.process([p[0] for p in calm_batch], [p[1] for p in calm_batch])
        f1 = calls("stereo.device_octree_fallbacks")
        assert f1 == f0
        outs = fe.process([p[0] for p in batch1], [p[1] for p in batch1])
        assert calls("stereo.device_octree_fallbacks") == f1 + 9, "dense pairs of the first batch after the tier was retired: repaired"
        check(batch1, outs, [0, 1, 2])
        outs = fe.process([p[0] for p in batch1], [p[1] for p in batch1])
        assert calls("stereo.device_octree_fallbacks") == f1 + 9, "the tier is back"
        check(batch1, outs, [2, 16])
    fe.close()


@pytest.mark.parametrize("hist", [True, False])
def test_dense_frames_in_concurrent_sub_batches(ctx, hist, monkeypatch):
    """256 pairs per batch = two sub-batches whose octree kernels run side by side on the two octree streams, every pair with
    levels beyond 4 096 candidates: the lists of the tiers behind k_octree are per stream.  Both batches run on the device
    (histogram tier) and must agree slot for slot.  With that tier switched off the first batch finds the sorted big tier
    absent and goes through the host-octree pipeline as a whole; the second runs on the device and must reproduce it.  The four
    distinct pairs are checked against the oracle."""
    w, h, nf, B = 640, 480, 1000, 256
    intr = synth.intrinsics(w, h)
    with ctx.options(oct_hist=int(hist)):
        fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"])
    dense = [synth.make_mosaic_pair(w, h, seed=160 + i, block=6) for i in range(4)]
    oex = ob.Extractor(nf)
    oex.extract(dense[0][0])
    top = max(len(oex.candidates(l)) for l in range(8))
    assert 4096 < top <= 16384, top
    pairs = [dense[b % 4] for b in range(B)]

    def calls(name):
        try:
            return ctx.get_stat(name)[1]
        except Exception:
            return 0
    f0 = calls("stereo.device_octree_fallbacks")
    first = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
    r1 = 0 if hist else B
    assert calls("stereo.device_octree_fallbacks") == f0 + r1
    second = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
    assert calls("stereo.device_octree_fallbacks") == f0 + r1, "the second dense batch stays on the device"
    for b in range(B):
        for k in ("keysL", "keysR", "descL", "descR", "uright", "depth"):
            assert np.array_equal(first[b][k], second[b][k]), (b, k)
        assert first[b]["n"] == second[b]["n"]
    for b in range(4):
        oL, oR = ob.Extractor(nf), ob.Extractor(nf)
        kL, dL, _ = oL.extract(pairs[b][0])
        kR, dR, _ = oR.extract(pairs[b][1])
        o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
        out = second[b]
        assert np.array_equal(out["keysL"], kL) and np.array_equal(out["keysR"], kR)
        assert np.array_equal(out["descL"], dL) and np.array_equal(out["descR"], dR)
        assert out["n"] == o["n"] and np.array_equal(out["uright"], o["uright"]) and np.array_equal(out["depth"], o["depth"])


def test_fisheye_match(ctx):
    rng = np.random.default_rng(2)
    fr = sc.oracle_stereo_frame(512, 512, 2000, 6)
    for (dL, dR) in [(fr["dL"], fr["dR"]), (fr["dL"][:700], fr["dR"][:1]), (fr["dL"][:5], fr["dR"][:0]),
                     (rng.integers(0, 256, (300, 32), dtype=np.uint8), rng.integers(0, 256, (2500, 32), dtype=np.uint8))]:
        o = ob.fisheye_match(dL, dR)
        g = orb.KernelController.launchFisheyeStereoMatchKernel(ctx, dL, dR)
        assert np.array_equal(g["matches"], o["matches"])
        if len(dR) >= 2:
            assert np.array_equal(g["best"], o["best"]) and np.array_equal(g["second"], o["second"])
    # ties: duplicated train rows -> the earlier index wins and the ratio test fails (best == second)
    dR = np.concatenate([fr["dR"][:50], fr["dR"][:50]])
    o = ob.fisheye_match(fr["dR"][:50], dR)
    g = orb.KernelController.launchFisheyeStereoMatchKernel(ctx, fr["dR"][:50], dR)
    assert np.array_equal(g["matches"], o["matches"]) and (g["best"] == 0).all() and (g["second"] == 0).all()


def _frame_views(fr, sf, w, h, uright=None, holder=None):
    args = dict(keys=fr["kL"], descriptors=fr["dL"], bounds=sc.frame_bounds(w, h), mbf=fr["intr"]["mbf"],
                mb=fr["intr"]["mb"], uright=uright, holder_obs=holder,
                cam=[fr["intr"][k] for k in ("fx", "fy", "cx", "cy")])
    return ob.FrameView(scale_factors_=sf, **args), orb.FrameView(scale_factors=sf, **args)


@pytest.mark.parametrize("th,dense", [(1.0, False), (3.0, False), (7.0, True), (15.0, True), (60.0, True)])
def test_search_local_points_mono_stereo(ctx, th, dense):
    """(th 60: windows with more candidates than the claim iteration's candidate cache holds per point - those points scan
    their window again in every pass, the others walk their cached keys)"""
    w, h, nf = 752, 480, 1200
    fr = sc.oracle_stereo_frame(w, h, nf, 12)
    sf, _ = ob.scale_factors(1.2, 8)
    sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["intr"]["mbf"], fr["intr"]["mb"])
    rng = np.random.default_rng(5)
    holder = np.where(rng.random(len(fr["kL"])) < 0.1, rng.integers(0, 3, len(fr["kL"])), -1).astype(np.int32)
    pts = sc.local_points_scenario(fr["kL"], fr["dL"], sf, w, h, seed=int(th * 10), M=1800, uright=sm["uright"], dense=dense)
    oF, gF = _frame_views(fr, sf, w, h, uright=sm["uright"], holder=holder)
    o = ob.search_local_points(oF, pts, th)
    g = orb.KernelController.launchSearchLocalPointsKernel(ctx, gF, pts, th)
    assert o["n"] > 50
    assert g["n"] == o["n"] and np.array_equal(g["assign"], o["assign"])
    assert np.array_equal(gF.holder_obs, oF.holder_obs)
    for k in ("best_dist", "best_dist2", "best_level", "best_level2", "best_idx"):
        assert np.array_equal(g[k], o[k]), k


@pytest.mark.parametrize("nf", [1500, 2000])  # 2000 = BASELINE configs[3] (TUM-VI: nFeatures 2000)
def test_search_local_points_two_cameras(ctx, nf):
    w, h = 512, 512
    fr = sc.fisheye_frame_scenario(w, h, nf, 9)
    sf, _ = ob.scale_factors(1.2, 8)
    pts = sc.two_camera_points(fr, sf, 3)
    kw = dict(keys=fr["kL"], keys_right=fr["kR"], descriptors=np.concatenate([fr["dL"], fr["dR"]]),
              bounds=sc.frame_bounds(w, h), left_to_right=fr["l2r"], right_to_left=fr["r2l"])
    oF = ob.FrameView(scale_factors_=sf, **kw)
    gF = orb.FrameView(scale_factors=sf, **kw)
    o = ob.search_local_points(oF, pts, 1.0)
    g = orb.KernelController.launchSearchLocalPointsKernel(ctx, gF, pts, 1.0)
    assert o["n"] > 50 and (o["assign"][len(fr["kL"]):] >= 0).sum() > 10
    assert g["n"] == o["n"] and np.array_equal(g["assign"], o["assign"])
    assert np.array_equal(gF.holder_obs, oF.holder_obs)
    for k in ("best_dist", "best_dist2", "best_level", "best_level2", "best_idx", "best_dist_r", "best_dist2_r",
              "best_level_r", "best_level2_r", "best_idx_r"):
        assert np.array_equal(g[k], o[k]), k


@pytest.mark.parametrize("th,fwd,bwd,ori", [(7.0, False, False, True), (15.0, True, False, True), (15.0, False, True, False),
                                           (90.0, False, True, True)])
def test_search_last_frame_pinhole(ctx, th, fwd, bwd, ori):
    w, h, nf = 752, 480, 1200
    fr = sc.oracle_stereo_frame(w, h, nf, 14)
    sf, _ = ob.scale_factors(1.2, 8)
    sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["intr"]["mbf"], fr["intr"]["mb"])
    last, Tcw = sc.last_frame_scenario(fr["kL"], fr["dL"], sm["uright"], sm["depth"], fr["intr"], w, h, seed=2)
    oF, gF = _frame_views(fr, sf, w, h, uright=sm["uright"])
    o = ob.search_last_frame(oF, last, Tcw, th, fwd, bwd, ori)
    g = orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, Tcw, th, fwd, bwd, ori)
    assert o["n"] > 100
    assert g["n"] == o["n"] and np.array_equal(g["assign"], o["assign"])
    assert np.array_equal(g["best_dist"], o["best_dist"]) and np.array_equal(g["best_idx"], o["best_idx"])
    assert np.array_equal(gF.holder_obs, oF.holder_obs)


@pytest.mark.parametrize("nf", [1500, 2000])  # 2000 = BASELINE configs[3] (TUM-VI: nFeatures 2000)
def test_search_last_frame_two_cameras_kb8(ctx, nf):
    """fisheye stereo (config 4): KannalaBrandt8 projection (atan2f / cosf / sinf as glibc evaluates them, libm_f32.h),
    right-camera search through Trl: assignments, distances and indices equal the oracle's."""
    w, h = 512, 512
    fr = sc.fisheye_frame_scenario(w, h, nf, 10)
    sf, _ = ob.scale_factors(1.2, 8)
    cam = [190.978, 190.973, 254.93, 256.90, 0.0034, 0.0007, -0.0020, 0.00020]  # TUM-VI-like KB8
    rng = np.random.default_rng(1)
    N = len(fr["kL"])
    # back-project left keypoints through an approximate inverse (pinhole-ish near the centre is enough)
    z = rng.uniform(2, 8, N).astype(np.float32)
    X = ((fr["kL"]["x"] - cam[2]) / cam[0] * z).astype(np.float32)
    Y = ((fr["kL"]["y"] - cam[3]) / cam[1] * z).astype(np.float32)
    last = dict(valid=(rng.random(N) < 0.8).astype(np.uint8), world_pos=np.stack([X, Y, z], 1),
                descriptors=fr["dL"].copy(), observations=rng.integers(0, 4, N).astype(np.int32),
                octave=fr["kL"]["octave"].astype(np.int32), angle=fr["kL"]["angle"].copy())
    Tcw = sc.random_pose(rng, 0.02, 0.005)
    Trl = np.concatenate([np.eye(3), [[-0.1], [0.0], [0.0]]], 1).astype(np.float32)
    kw = dict(keys=fr["kL"], keys_right=fr["kR"], descriptors=np.concatenate([fr["dL"], fr["dR"]]),
              bounds=sc.frame_bounds(w, h), left_to_right=fr["l2r"], right_to_left=fr["r2l"], cam_model=1, cam=cam,
              Trl=Trl)
    oF = ob.FrameView(scale_factors_=sf, **kw)
    gF = orb.FrameView(scale_factors=sf, **kw)
    o = ob.search_last_frame(oF, last, Tcw, 15.0, False, False, True)
    g = orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, Tcw, 15.0, False, False, True)
    assert o["n"] > 30
    assert g["n"] == o["n"] and np.array_equal(g["assign"], o["assign"])
    for k in ("best_dist", "best_idx", "best_dist_r", "best_idx_r"):
        assert np.array_equal(g[k], o[k]), k


@pytest.mark.parametrize("th,fwd,bwd,ori", [(7.0, False, False, True), (15.0, True, False, False), (30.0, False, True, True)])
def test_search_last_frame_sophus_pose_form(ctx, th, fwd, bwd, ori):
    """the poses as Sophus::SE3f holds and applies them (ft_search_last_frame_se3): `Tcw * x3Dw` of the CPU branch
    (src/ORBmatcher.cc:1805) is a quaternion rotation, not a matrix product - device and oracle evaluate the same operations,
    one-shot and on a resident frame"""
    w, h, nf = 752, 480, 1200
    fr = sc.oracle_stereo_frame(w, h, nf, 15)
    sf, _ = ob.scale_factors(1.2, 8)
    sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["intr"]["mbf"], fr["intr"]["mb"])
    last, _ = sc.last_frame_scenario(fr["kL"], fr["dL"], sm["uright"], sm["depth"], fr["intr"], w, h, seed=3)
    q, t = sc.random_se3(np.random.default_rng(int(th)), 0.03, 0.006)
    oF, gF = _frame_views(fr, sf, w, h, uright=sm["uright"])
    o = ob.search_last_frame(oF, last, ob.SE3(q, t), th, fwd, bwd, ori)
    g = orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, orb.SE3(q, t), th, fwd, bwd, ori)
    assert o["n"] > 100
    assert g["n"] == o["n"] and np.array_equal(g["assign"], o["assign"])
    assert np.array_equal(g["best_dist"], o["best_dist"]) and np.array_equal(g["best_idx"], o["best_idx"])
    assert np.array_equal(gF.holder_obs, oF.holder_obs)
    _, gF2 = _frame_views(fr, sf, w, h, uright=sm["uright"])
    tf = orb.TrackedFrame(ctx, max_keypoints=4096, max_points=4096)
    tf.upload(gF2)
    g2 = tf.search_last_frame(last, orb.SE3(q, t), th, fwd, bwd, ori)
    assert g2["n"] == o["n"] and np.array_equal(g2["assign"], o["assign"])
    tf.close()
    with pytest.raises(Exception):
        orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, orb.SE3([0.5, 0, 0, 0.5], t), th)  # not a unit quaternion


@pytest.mark.parametrize("nf", [1500, 2000])  # 2000 = BASELINE configs[3] (TUM-VI: nFeatures 2000)
def test_search_last_frame_sophus_pose_form_two_cameras_kb8(ctx, nf):
    """two-camera KB8 frame: Tcw and GetRelativePoseTrl() both in the Sophus form"""
    w, h = 512, 512
    fr = sc.fisheye_frame_scenario(w, h, nf, 11)
    sf, _ = ob.scale_factors(1.2, 8)
    cam = list(sc.KB8_CAM)
    rng = np.random.default_rng(2)
    N = len(fr["kL"])
    z = rng.uniform(2, 8, N).astype(np.float32)
    X = ((fr["kL"]["x"] - cam[2]) / cam[0] * z).astype(np.float32)
    Y = ((fr["kL"]["y"] - cam[3]) / cam[1] * z).astype(np.float32)
    last = dict(valid=(rng.random(N) < 0.8).astype(np.uint8), world_pos=np.stack([X, Y, z], 1),
                descriptors=fr["dL"].copy(), observations=rng.integers(0, 4, N).astype(np.int32),
                octave=fr["kL"]["octave"].astype(np.int32), angle=fr["kL"]["angle"].copy())
    q, t = sc.random_se3(rng, 0.02, 0.005)
    qr, _ = sc.random_se3(rng, 0.0, 0.01)
    tr = np.array([-0.1, 0.001, 0.002], np.float32)
    kw = dict(keys=fr["kL"], keys_right=fr["kR"], descriptors=np.concatenate([fr["dL"], fr["dR"]]),
              bounds=sc.frame_bounds(w, h), left_to_right=fr["l2r"], right_to_left=fr["r2l"], cam_model=1, cam=cam)
    oF, gF = ob.FrameView(scale_factors_=sf, **kw), orb.FrameView(scale_factors=sf, **kw)
    o = ob.search_last_frame(oF, last, ob.SE3(q, t), 15.0, False, False, True, Trl=ob.SE3(qr, tr))
    g = orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, orb.SE3(q, t), 15.0, False, False, True, Trl=orb.SE3(qr, tr))
    assert o["n"] > 30 and (o["best_idx_r"] >= 0).sum() > 30
    assert g["n"] == o["n"] and np.array_equal(g["assign"], o["assign"])
    for k in ("best_dist", "best_idx", "best_dist_r", "best_idx_r"):
        assert np.array_equal(g[k], o[k]), k
    with pytest.raises(Exception):
        orb.KernelController.launchPoseEstimationKernel(ctx, gF, last, orb.SE3(q, t), 15.0)  # two cameras need Trl


LOG_SF = float(np.float32(np.log(np.float32(1.2))))  # Frame::mfLogScaleFactor = log(mfScaleFactor) stored as float


def _frustum_case(w, h, nf, seed):
    fr = sc.oracle_stereo_frame(w, h, nf, seed)
    sf, _ = ob.scale_factors(1.2, 8)
    sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], fr["intr"]["mbf"], fr["intr"]["mb"])
    pts, Rcw, tcw = sc.map_points_scenario(fr["kL"], fr["dL"], sm["depth"], fr["intr"], 8, sf, seed + 100)
    return fr, sf, sm, pts, Rcw, tcw


def test_is_in_frustum_pinhole_bit_exact(ctx):
    """Frame::isInFrustum + PredictScale for rectified stereo (Nleft == -1): every flag, level and float field equals
    the oracle's bit for bit (pinhole projection is plain float arithmetic)."""
    w, h = 752, 480
    fr, sf, sm, pts, Rcw, tcw = _frustum_case(w, h, 1200, 21)
    oF, gF = _frame_views(fr, sf, w, h, uright=sm["uright"])
    for limit in (0.5, 0.9):
        o = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw), pts, limit, LOG_SF)
        g = orb.is_in_frustum(ctx, gF, orb.make_pose(Rcw, tcw), pts, limit, LOG_SF)
        assert 200 < o["n"] < len(pts["world_pos"]) - 200
        assert g["n"] == o["n"]
        for k, _ in ob.FRUSTUM_FIELDS:
            assert np.array_equal(g[k], o[k]), k
    empty = {k: v[:0] for k, v in pts.items()}
    assert orb.is_in_frustum(ctx, gF, orb.make_pose(Rcw, tcw), empty, 0.5, LOG_SF)["n"] == 0


@pytest.mark.parametrize("nf", [1500, 2000])  # 2000 = BASELINE configs[3] (TUM-VI: nFeatures 2000)
def test_is_in_frustum_two_cameras_kb8(ctx, nf):
    """fisheye stereo (Nleft != -1): both cameras through isInFrustumChecks, right camera pose composed from Trl / Tlr.
    KannalaBrandt8::project uses atan2f / cosf / sinf of the host libm; the device evaluates glibc's algorithms for the
    three (libm_f32.h, checked against the host on every argument), so every flag, level and float equals the oracle's
    bit for bit - as for the pinhole model."""
    w, h = 512, 512
    fr = sc.fisheye_frame_scenario(w, h, nf, 10)
    sf, _ = ob.scale_factors(1.2, 8)
    cam = [190.978, 190.973, 254.93, 256.90, 0.0034, 0.0007, -0.0020, 0.00020]
    Trl = np.concatenate([np.eye(3), [[-0.1], [0.0], [0.0]]], 1).astype(np.float32)
    kw = dict(keys=fr["kL"], keys_right=fr["kR"], descriptors=np.concatenate([fr["dL"], fr["dR"]]),
              bounds=sc.frame_bounds(w, h), left_to_right=fr["l2r"], right_to_left=fr["r2l"], cam_model=1, cam=cam, Trl=Trl)
    oF, gF = ob.FrameView(scale_factors_=sf, **kw), orb.FrameView(scale_factors=sf, **kw)
    intr = dict(fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
    tlr = (0.1, 0.0, 0.0)
    differ = 0
    for seed in (77, 78, 79):
        pts, Rcw, tcw = sc.map_points_scenario(fr["kL"], fr["dL"], np.zeros(len(fr["kL"]), np.float32), intr, 8, sf, seed)
        o = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw, tlr), pts, 0.5, LOG_SF)
        g = orb.is_in_frustum(ctx, gF, orb.make_pose(Rcw, tcw, tlr), pts, 0.5, LOG_SF)
        assert o["in_view"].sum() > 200 and o["in_view_r"].sum() > 200
        differ += int((o["in_view"] != o["in_view_r"]).sum())
        assert g["n"] == o["n"]
        for k, _ in ob.FRUSTUM_FIELDS:
            assert np.array_equal(g[k], o[k]), (seed, k)
    assert differ > 5  # the two cameras do not see the same set


def _oracle_tracking_sequence(oF, last, Tcw_last, pts, Rcw, tcw, th_last, th_local, far, th_far):
    """TrackWithMotionModel's search followed by SearchLocalPoints on the same frame (holder_obs carries over)"""
    o1 = ob.search_last_frame(oF, last, Tcw_last, th_last, False, False, True)
    ofr = ob.is_in_frustum(oF, ob.make_pose(Rcw, tcw), pts, 0.5, LOG_SF)
    o2 = ob.search_local_points(oF, sc.local_points_from_frustum(ofr, pts, far, th_far), th_local)
    return o1, ofr, o2


@pytest.mark.parametrize("far", [False, True])
def test_tracked_frame_sequence_equals_oracle(ctx, far):
    """ft_tracked_frame: upload once, then SearchByProjection(last frame) and isInFrustum + SearchByProjection(local
    map) on the resident frame - assignments, frustum fields and the final mvpMapPoints occupancy equal the oracle
    running the same sequence; the frustum outputs never visit the host between the two kernels."""
    w, h = 752, 480
    fr, sf, sm, pts, Rcw, tcw = _frustum_case(w, h, 1200, 23)
    last, Tcw_last = sc.last_frame_scenario(fr["kL"], fr["dL"], sm["uright"], sm["depth"], fr["intr"], w, h, seed=4)
    oF, gF = _frame_views(fr, sf, w, h, uright=sm["uright"])
    th_far = float(np.percentile(sm["depth"][sm["depth"] > 0], 85)) if far else 0.0
    o1, ofr, o2 = _oracle_tracking_sequence(oF, last, Tcw_last, pts, Rcw, tcw, 15.0, 3.0, far, th_far)
    tf = orb.TrackedFrame(ctx, max_keypoints=4096, max_points=4096)
    tf.upload(gF)
    g1 = tf.search_last_frame(last, Tcw_last, 15.0)
    assert o1["n"] > 100 and g1["n"] == o1["n"] and np.array_equal(g1["assign"], o1["assign"])
    g2 = tf.track_local_map(orb.make_pose(Rcw, tcw), pts, 0.5, LOG_SF, 3.0, far_points=far, th_far_points=th_far)
    for k, _ in ob.FRUSTUM_FIELDS:
        assert np.array_equal(g2[k], ofr[k]), k
    assert g2["n_to_match"] == ofr["n"]
    assert o2["n"] > 30 and g2["n"] == o2["n"] and np.array_equal(g2["assign"], o2["assign"])
    assert np.array_equal(tf.holder_obs(), oF.holder_obs)
    # a second frame through the same object
    tf.upload(gF)
    assert np.array_equal(tf.holder_obs(), gF.holder_obs)
    tf.close()


def test_tracked_frame_bound_to_stereo_frontend(ctx):
    """the same sequence with the frame bound to what the stereo front end left in HBM (keypoints, descriptors,
    mvuRight of pair 1 of the batch): no upload of the frame arrays at all"""
    w, h, nf, B = 752, 480, 1200, 2
    intr = synth.intrinsics(w, h)
    fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"])
    pairs = [synth.make_stereo_pair(w, h, 31 + b) for b in range(B)]
    outs = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
    out = outs[1]
    sf, _ = ob.scale_factors(1.2, 8)
    fr = dict(kL=out["keysL"], dL=out["descL"], intr=intr)
    pts, Rcw, tcw = sc.map_points_scenario(out["keysL"], out["descL"], out["depth"], intr, 8, sf, 55)
    last, Tcw_last = sc.last_frame_scenario(out["keysL"], out["descL"], out["uright"], out["depth"], intr, w, h, seed=6)
    oF, gF = _frame_views(fr, sf, w, h, uright=out["uright"])
    o1, ofr, o2 = _oracle_tracking_sequence(oF, last, Tcw_last, pts, Rcw, tcw, 15.0, 3.0, False, 0.0)
    tf = orb.TrackedFrame(ctx, max_keypoints=fe.capacity, max_points=4096)
    tf.bind_stereo(fe, 1, gF)
    g1 = tf.search_last_frame(last, Tcw_last, 15.0)
    g2 = tf.track_local_map(orb.make_pose(Rcw, tcw), pts, 0.5, LOG_SF, 3.0)
    assert g1["n"] == o1["n"] and np.array_equal(g1["assign"], o1["assign"])
    assert g2["n"] == o2["n"] and np.array_equal(g2["assign"], o2["assign"]) and o2["n"] > 50
    assert np.array_equal(tf.holder_obs(), oF.holder_obs)
    with pytest.raises(Exception):
        tf.bind_stereo(fe, 5, gF)  # slot out of range
    tf.close()


def test_fisheye_stereo_with_triangulation(ctx):
    """complete ComputeStereoFishEyeMatches: 2-NN + ratio on the device, then KannalaBrandt8::TriangulateMatches per
    pair.  Every float step is reproducible: the Newton unprojection and its tanf, atan2f / cosf / sinf of the
    reprojection (glibc's algorithms, libm_f32.h) and the one-sided Jacobi SVD in double, which device and oracle evaluate
    with the same operations in the same order - so matches, depths and 3-D points equal the oracle's bit for bit.
    (The reference's own SVD is Eigen's JacobiSVD: that step of the ORACLE is unpinned)"""
    S = sc.fisheye_rig_scenario(9, n=2000)
    rng = np.random.default_rng(4)
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
    ls2 = (ob.scale_factors(1.2, 8)[0] ** 2).astype(np.float32)
    o = ob.fisheye_stereo(ob.make_rig(sc.KB8_CAM, sc.KB8_CAM, S["Rlr"], S["tlr"]), dL, kL, dR, kR, ls2)
    g = orb.fisheye_stereo(ctx, sc.KB8_CAM, sc.KB8_CAM, S["Rlr"], S["tlr"], dL, kL, dR, kR, ls2)
    assert o["n"] > 1000
    assert g["n"] == o["n"] and np.array_equal(g["matches"], o["matches"])
    assert np.array_equal(g["depth"], o["depth"]) and np.array_equal(g["p3d"], o["p3d"])
    assert (g["depth"][g["matches"] < 0] == -1).all()
    e = orb.fisheye_stereo(ctx, sc.KB8_CAM, sc.KB8_CAM, S["Rlr"], S["tlr"], dL[:0], kL[:0], dR, kR, ls2)
    assert e["n"] == 0


def test_full_size_batch_properties(ctx):
    """BASELINE.json's full configuration in bench.py's own launch shape (1280x720 stereo, nFeatures 2000, 512 pairs per
    batch = two sub-batches of 256 pairs of the device-octree pipeline, i.e. 256 images per kernel launch and camera, two
    batches in flight):
    too large for the oracle in a test, so it is checked through properties - repeated pairs give identical results
    wherever they sit in the batch and in whichever front end, a sample of pairs equals the oracle bit for bit,
    every depth is mbf / disparity, keypoints respect the border and the per-level quotas."""
    import ctypes as C
    w, h, nf, B, D = 1280, 720, 2000, 512, 8
    intr = synth.intrinsics(w, h)
    fes = [orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"]) for _ in range(2)]
    pairs = [synth.make_stereo_pair(w, h, seed=900 + i) for i in range(D)]
    dev = [(ctx.to_device(p[0]), ctx.to_device(p[1])) for p in pairs]
    pL = (C.c_void_p * B)(*[dev[b % D][0].ptr for b in range(B)])
    pR = (C.c_void_p * B)(*[dev[b % D][1].ptr for b in range(B)])
    fes[0].submit_raw(pL, pR, B, True, w)
    fes[1].submit_raw(pL, pR, B, True, w)
    fes[0].wait()
    fes[1].wait()
    quotas = fes[0].left.features_per_level()
    sf, _ = ob.scale_factors(1.2, 8)

    def res(fe, b):
        n = int(fe._nL[b])
        return fe._kL[b, :n], fe._dL[b, :n], fe._ur[b, :n], fe._dp[b, :n], int(fe._nm[b])
    for b in range(B):
        k0, d0, u0, z0, m0 = res(fes[0], b % D)
        for fe in fes:
            k, d, u, z, m = res(fe, b)
            assert np.array_equal(k, k0) and np.array_equal(d, d0) and np.array_equal(u, u0) and np.array_equal(z, z0) and m == m0
    for b in range(D):
        k, d, u, z, m = res(fes[0], b)
        assert 0.9 * nf <= len(k) <= nf + 24
        lv = k["octave"]
        assert (np.diff(lv) >= 0).all() and all((lv == l).sum() <= max(quotas[l] + 3, 8) for l in range(8))
        xl, yl = k["x"] / sf[lv], k["y"] / sf[lv]
        assert (xl >= 18.99).all() and (yl >= 18.99).all()
        ok = z > 0
        assert ok.sum() == m and m > 100
        assert np.allclose(z[ok], intr["mbf"] / (k["x"][ok] - u[ok]), rtol=1e-5)
        assert (u[~ok] == -1).all() and d.any(axis=1).all()
    for b in (0, 5):  # the oracle on two of the pairs
        oL, oR = ob.Extractor(nf), ob.Extractor(nf)
        kL, dL, _ = oL.extract(pairs[b][0])
        kR, dR, _ = oR.extract(pairs[b][1])
        o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
        k, d, u, z, m = res(fes[1], b + 2 * D)
        assert np.array_equal(k, kL) and np.array_equal(d, dL) and np.array_equal(u, o["uright"]) and np.array_equal(z, o["depth"])


def test_features_in_area(ctx):
    """Frame::GetFeaturesInArea with the grid evaluated on the device: same indices in the same order as the oracle's
    64x48 grid walk, for both cameras of a two-camera frame, with and without level limits, at the image borders."""
    w, h = 512, 512
    fr = sc.fisheye_frame_scenario(w, h, 1500, 10)
    sf, _ = ob.scale_factors(1.2, 8)
    kw = dict(keys=fr["kL"], keys_right=fr["kR"], descriptors=np.concatenate([fr["dL"], fr["dR"]]),
              bounds=sc.frame_bounds(w, h), left_to_right=fr["l2r"], right_to_left=fr["r2l"])
    oF, gF = ob.FrameView(scale_factors_=sf, **kw), orb.FrameView(scale_factors=sf, **kw)
    rng = np.random.default_rng(8)
    nq = 400
    x = rng.uniform(-30, w + 30, nq).astype(np.float32)
    y = rng.uniform(-30, h + 30, nq).astype(np.float32)
    r = rng.choice([1.0, 7.5, 15.0, 40.0, 120.0], nq).astype(np.float32)
    lo = rng.integers(-1, 5, nq).astype(np.int32)
    hi = np.where(rng.random(nq) < 0.3, -1, lo + rng.integers(0, 4, nq)).astype(np.int32)
    right = (rng.random(nq) < 0.4).astype(np.uint8)
    got, cnt = orb.features_in_area(ctx, gF, x, y, r, lo, hi, right, capacity=2048)
    total = 0
    for q in range(nq):
        o = ob.features_in_area(oF, float(x[q]), float(y[q]), float(r[q]), int(lo[q]), int(hi[q]), bool(right[q]))
        assert cnt[q] == len(o) and np.array_equal(got[q], o), q
        total += len(o)
    assert total > 5000
    # rectified frame (Nleft == -1) and a capacity smaller than the hit count
    fr2 = sc.oracle_stereo_frame(752, 480, 1200, 12)
    oF2, gF2 = _frame_views(fr2, sf, 752, 480)
    got, cnt = orb.features_in_area(ctx, gF2, [376.0], [240.0], [200.0], [-1], [-1], None, capacity=16)
    o = ob.features_in_area(oF2, 376.0, 240.0, 200.0, -1, -1, False)
    assert cnt[0] == len(o) > 16 and np.array_equal(got[0], o[:16])


def test_stereo_frontend_random_configurations(ctx):
    """seeded sweep of the fused front end over image sizes, feature counts, batch sizes and baselines: keypoints,
    descriptors, mvuRight, mvDepth and the match count equal extract(L), extract(R), ComputeStereoMatches of the oracle"""
    rng = np.random.default_rng(77)
    for trial in range(8):
        w, h = int(rng.integers(200, 900)), int(rng.integers(160, 620))
        nf = int(rng.choice([300, 1000, 2000]))
        B = int(rng.integers(1, 4))
        intr = synth.intrinsics(w, h)
        mbf = float(intr["mbf"] * rng.choice([0.5, 1.0, 2.0]))
        fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, mbf, intr["mb"])
        pairs = [synth.make_stereo_pair(w, h, 500 + 10 * trial + b) for b in range(B)]
        outs = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
        for (imL, imR), out in zip(pairs, outs):
            oL, oR = ob.Extractor(nf), ob.Extractor(nf)
            kL, dL, _ = oL.extract(imL)
            kR, dR, _ = oR.extract(imR)
            o = ob.stereo_match(oL, oR, kL, kR, dL, dR, mbf, intr["mb"])
            assert np.array_equal(out["keysL"], kL) and np.array_equal(out["keysR"], kR), (trial, w, h)
            assert np.array_equal(out["descL"], dL) and np.array_equal(out["descR"], dR)
            assert out["n"] == o["n"] and np.array_equal(out["uright"], o["uright"]) and np.array_equal(out["depth"], o["depth"])
        fe.close()


def test_stereo_frontend_graph_replay(ctx):
    """latency mode: a small batch with a fixed call shape is captured once as a HIP graph and replayed; frames that
    are resident in HBM change between replays through the level-0 pointer table, results always equal the oracle's"""
    import ctypes as C
    w, h, nf, B = 752, 480, 1200, 2
    intr = synth.intrinsics(w, h)
    fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"])

    def calls(name):
        try:
            return ctx.get_stat(name)[1]
        except Exception:
            return 0
    c0, l0 = calls("stereo.graph_captures"), calls("stereo.graph_launch")
    frames = [[synth.make_stereo_pair(w, h, 700 + 10 * k + b) for b in range(B)] for k in range(3)]
    dev = [[(ctx.to_device(p[0]), ctx.to_device(p[1])) for p in fr] for fr in frames]
    for k in range(3):
        pL = (C.c_void_p * B)(*[d[0].ptr for d in dev[k]])
        pR = (C.c_void_p * B)(*[d[1].ptr for d in dev[k]])
        fe.process_raw(pL, pR, B, True, w)
        for b in range(B):
            oL, oR = ob.Extractor(nf), ob.Extractor(nf)
            kL, dL, _ = oL.extract(frames[k][b][0])
            kR, dR, _ = oR.extract(frames[k][b][1])
            o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
            n = int(fe._nL[b])
            assert np.array_equal(fe._kL[b, :n], kL) and np.array_equal(fe._dL[b, :n], dL), (k, b)
            assert np.array_equal(fe._ur[b, :n], o["uright"]) and int(fe._nm[b]) == o["n"]
    if calls("stereo.graph_capture_failed") == 0:
        assert calls("stereo.graph_captures") == c0 + 1, "the call shape did not change: one capture"
        assert calls("stereo.graph_launch") == l0 + 2, "the second and third frame replay the graph"
    # host frames: every call brings new arrays (new pointers); they reach the captured uploads through pinned staging
    c1, l1 = calls("stereo.graph_captures"), calls("stereo.graph_launch")
    for k in range(3):
        outs = fe.process([p[0].copy() for p in frames[k]], [p[1].copy() for p in frames[k]])
        for b in range(B):
            oL = ob.Extractor(nf)
            kL, dL, _ = oL.extract(frames[k][b][0])
            assert np.array_equal(outs[b]["keysL"], kL) and np.array_equal(outs[b]["descL"], dL), (k, b)
    if calls("stereo.graph_capture_failed") == 0:
        assert calls("stereo.graph_captures") == c1 + 1 and calls("stereo.graph_launch") == l1 + 2
    # frames the caller keeps in pinned memory are read where they are (dense, and as views into a wider pinned buffer:
    # stride > width, rows starting at odd addresses), mixed with a pageable frame in the same call
    wide = ctx.pinned_array((h, w + 13), np.uint8)
    dense = ctx.pinned_array((h, w), np.uint8)
    for k in range(2):
        (L0, R0), (L1, R1) = frames[k][0], frames[k][1]
        wide[:, 5:5 + w] = L0
        dense[:] = R1
        # (a call has one stride: both frames are views of wider pinned buffers)
        fe1 = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, 1, intr["mbf"], intr["mb"])
        Rw = ctx.pinned_array((h, w + 13), np.uint8)
        Rw[:, 5:5 + w] = R1
        pl1, pr1 = (C.c_void_p * 1)(wide.ctypes.data + 5), (C.c_void_p * 1)(Rw.ctypes.data + 5)
        fe1.process_raw(pl1, pr1, 1, False, w + 13)
        oL, oR = ob.Extractor(nf), ob.Extractor(nf)
        kL, dL, _ = oL.extract(L0)
        kR, dR, _ = oR.extract(R1)
        o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
        n = int(fe1._nL[0])
        assert np.array_equal(fe1._kL[0, :n], kL) and np.array_equal(fe1._dL[0, :n], dL)
        assert np.array_equal(fe1._kR[0, :int(fe1._nR[0])], kR) and np.array_equal(fe1._ur[0, :n], o["uright"])
        # pinned and pageable frames in one call (dense rows)
        fe.process_raw((C.c_void_p * 2)(dense.ctypes.data, L1.ctypes.data), (C.c_void_p * 2)(R0.ctypes.data, dense.ctypes.data),
                       2, False, w)
        oA = ob.Extractor(nf)
        kA, dA, _ = oA.extract(R1)  # `dense` holds R1: left frame of pair 0 and right frame of pair 1
        assert np.array_equal(fe._kL[0, :int(fe._nL[0])], kA) and np.array_equal(fe._kR[1, :int(fe._nR[1])], kA)
        kB, dB, _ = ob.Extractor(nf).extract(L1)
        assert np.array_equal(fe._kL[1, :int(fe._nL[1])], kB) and np.array_equal(fe._dL[1, :int(fe._nL[1])], dB)
        fe1.close()
    # a different batch size is a different graph
    pL = (C.c_void_p * 1)(dev[0][0][0].ptr)
    pR = (C.c_void_p * 1)(dev[0][0][1].ptr)
    fe.process_raw(pL, pR, 1, True, w)
    oL = ob.Extractor(nf)
    kL, dL, _ = oL.extract(frames[0][0][0])
    assert np.array_equal(fe._kL[0, :int(fe._nL[0])], kL)


def test_candidate_cache_changes_nothing(ctx):
    """the claim iteration with and without its candidate cache (FT_SEARCH_CACHE=0, read once per process): the resident-frame
    tracking sequence on a two-camera KB8 frame at th 7 and 15 gives identical assignments and frustum fields"""
    import os
    import subprocess
    import sys
    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
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
last, Tcw = sc.last_frame_scenario(kL, dL, None, depth, intr, w, h, seed=3)
pts, Rcw, tcw = sc.map_points_scenario(kL, dL, depth, intr, 8, sf, 4, M=2000)
tf = orb.TrackedFrame(ctx, 2 * ex.max_keypoints + 64, 4096)
hs = hashlib.sha256()
for th in (7.0, 15.0):
    tf.upload(F)
    a = tf.search_last_frame(last, Tcw, th)
    b = tf.track_local_map(orb.make_pose(Rcw, tcw, (0.101, 0.0, 0.0)), pts, 0.5, float(np.float32(np.log(np.float32(1.2)))), th)
    assert a["n"] > 100 and b["n"] > 100
    for x in (a["assign"], b["assign"], b["level"], b["proj_x"], tf.holder_obs()):
        hs.update(np.ascontiguousarray(x).tobytes())
print("HASH", hs.hexdigest(), ctx.get_stat("tracked.track_local_map.passes")[0])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for cache in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, FT_SEARCH_CACHE=cache))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][-1])
    assert outs[0] == outs[1], outs


def test_randomized_soaks_replayed_in_the_suite():
    """50 trials of tests/tools/soak_search.py (random frame sizes, feature counts up to 2000, two-camera KB8 frames, both
    searches, the frustum, the resident frame, Sophus-form poses) and 4 batches of tests/tools/soak_batch.py (B frames per launch),
    all compared with the oracle for equality"""
    import importlib.util
    import os
    tools = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools")
    for name, argv in (("soak_search", ["--trials", "50", "--seed", "5"]), ("soak_batch", ["--trials", "4", "--seed", "5", "--frames", "12"])):
        spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert mod.main(argv) == 0, name
