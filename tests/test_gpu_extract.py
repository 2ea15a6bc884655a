"""Parity of the HIP extractor against the oracle, stage by stage and end to end (needs an MI355X).

Bar (BASELINE.json north_star): keypoint coordinates / scores / octaves, descriptor bytes and pyramid
pixels bit-exact; orientation floats within 1e-4 (they are in fact compared bit-exactly first).
"""
import numpy as np
import pytest

from fasttrack_amd import orb, synth
from oracle import binding as ob

pytestmark = pytest.mark.gpu

CONFIGS = [  # (width, height, nfeatures) of BASELINE.json configs 2, 3, 4
    (640, 480, 1000),
    (752, 480, 1200),
    (512, 512, 2000),
]


@pytest.fixture(scope="module")
def ctx():
    c = orb.Context(0)
    yield c
    c.close()


def _check_same(gk, gd, ok, od):
    assert len(gk) == len(ok)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(gk[f], ok[f]), f
    assert np.allclose(gk["angle"], ok["angle"], rtol=0, atol=1e-4)
    assert np.array_equal(gk["angle"], ok["angle"]), "angles agree to 1e-4 but not bit-exactly"
    assert np.array_equal(gd, od)


@pytest.mark.parametrize("w,h,nf", CONFIGS)
def test_pyramid_and_candidates_bit_exact(ctx, w, h, nf):
    img = synth.make_image(w, h, seed=11)
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    oex = ob.Extractor(nf)
    ex(img)
    oex.extract(img)
    for level in range(8):
        assert np.array_equal(ex.image_pyramid_level(level), oex.level(level)), f"pyramid level {level}"
        gc, oc = ex.candidates(level), oex.candidates(level)
        assert gc.shape == oc.shape and np.array_equal(gc, oc), f"FAST candidates level {level}"
        assert len(gc) > 0


@pytest.mark.parametrize("w,h,nf", CONFIGS + [(1280, 720, 2000)])
def test_extract_bit_exact(ctx, w, h, nf):
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    oex = ob.Extractor(nf)
    for seed in (1, 2):
        img = synth.make_image(w, h, seed=seed)
        gk, gd, gm = ex(img)
        ok, od, om = oex.extract(img)
        _check_same(gk, gd, ok, od)
        assert gm == om
        assert len(gk) >= nf * 0.9


@pytest.mark.parametrize("w,h,nf,lap", [
    (752, 480, 1000, (0, 1000)),   # BASELINE configs[0]: EuRoC monocular call shape, Frame::ExtractORB(0, im, 0, 1000) (src/Frame.cc:335)
    (752, 480, 5000, (0, 1000)),   # mpIniORBextractor = 5 * nFeatures (src/Tracking.cc:632)
    (1280, 720, 10000, (0, 0)),    # 5 * 2000 at the bench size
])
def test_extract_monocular_call_shape_and_ini_extractor(ctx, w, h, nf, lap):
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    oex = ob.Extractor(nf)
    for seed, dens in ((3, 1.0), (4, 3.0)):
        img = synth.make_image(w, h, seed=seed, density=dens)
        gk, gd, gm = ex(img, lapping_area=lap)
        ok, od, om = oex.extract(img, lap=lap)
        _check_same(gk, gd, ok, od)
        assert gm == om
        if lap == (0, 1000):
            assert gm == 0 and len(gk) > 0.6 * min(nf, 3000)  # every keypoint has x in [0, 1000]: all filled from the back


@pytest.mark.parametrize("w,h", [(640, 480), (752, 480), (161, 123)])
def test_gaussian_blur_of_whole_levels_bit_exact(ctx, w, h):
    """a7 directly: every pyramid level blurred by the descriptor kernel's own blur routines (od_hblur4 / od_vblur7,
    BORDER_REFLECT_101) equals the oracle's cv::GaussianBlur(7x7, sigma 2) restatement, pixel for pixel"""
    ex = orb.ORBextractor(ctx, 500, 1.2, 8, 20, 7, w, h)
    oex = ob.Extractor(500)
    for img in (synth.make_image(w, h, seed=21), synth.make_noise(w, h, seed=22)):
        ex(img)
        oex.extract(img)
        for level in range(8):
            src = ex.image_pyramid_level(level)
            assert np.array_equal(ex.blurred_level(level), ob.gaussian_blur7(src)), f"blur of level {level}"


def test_equally_spaced_host_frames_go_up_as_one_copy(ctx):
    """host frames at a constant distance (a ring buffer / clip in one allocation, stride == width == slot pitch): the batch is
    uploaded by one strided copy; same results as the oracle, also with a gap between the frames and for a 3-frame batch
    (below the one-copy threshold)"""
    w, h, nf = 640, 480, 800
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=6)
    oex = ob.Extractor(nf)
    clip = np.zeros((6, h + 7, w), np.uint8)  # 7 spare rows between the frames
    for b in range(6):
        clip[b, :h] = synth.make_image(w, h, seed=80 + b)
    for frames in ([clip[b, :h] for b in range(6)], [clip[b, :h] for b in range(3)]):
        assert all(f.flags["C_CONTIGUOUS"] for f in frames)
        res = ex.extract_batch(frames)
        for f, (gk, gd, gm) in zip(frames, res):
            ok, od, om = oex.extract(f)
            _check_same(gk, gd, ok, od)


@pytest.mark.parametrize("w,h,sf,levels", [(333, 257, 1.2, 8), (1000, 96, 1.2, 2), (641, 479, 1.5, 5), (512, 512, 2.0, 4),
                                            (97, 83, 1.1, 6), (1279, 717, 1.25, 8), (406, 302, 2.4, 3), (300, 200, 3.0, 2)])
@pytest.mark.parametrize("form", [1, 0], ids=["row-streaming", "tile-kernel"])
def test_row_streaming_pyramid_bit_exact(ctx, w, h, sf, levels, form):
    """launches of 8+ images build the pyramid with k_pyr_rows (option pyr_rows = 1, the default: a lane owns two output columns,
    rows streamed top to bottom) or with the tile kernel k_pyr_down (pyr_rows = 0): every level of the first and the last
    image equals the oracle's cv::resize chain - odd sizes, INTER_AREA at exactly 2.0, other scale factors (3.0 takes the
    tile kernel either way: its taps do not fit the 8-byte window)"""
    nf = 300
    with ctx.options(pyr_rows=form):
        ex = orb.ORBextractor(ctx, nf, sf, levels, 20, 7, w, h, max_batch=9)
    oex = ob.Extractor(nf, sf, levels)
    imgs = [synth.make_image(w, h, seed=90 + b) for b in range(8)] + [synth.make_noise(w, h, seed=99)]
    res = ex.extract_batch(imgs)
    for slot in (0, 8):
        ok, od, om = oex.extract(imgs[slot])
        for level in range(levels):
            assert np.array_equal(ex.image_pyramid_level(level, slot=slot), oex.level(level)), f"slot {slot} level {level}"
        _check_same(res[slot][0], res[slot][1], ok, od)


def test_lapping_area_partition(ctx):
    """fisheye / monocular callers pass a lapping area: keypoints inside fill from the back (ORBextractor.cc:1476-1485)"""
    w, h, nf = 512, 512, 2000
    img = synth.make_image(w, h, seed=5)
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    oex = ob.Extractor(nf)
    for lap in [(0, 511), (0, 1000), (200, 300), (0, 0)]:
        gk, gd, gm = ex(img, lap)
        ok, od, om = oex.extract(img, lap)
        _check_same(gk, gd, ok, od)
        assert gm == om
    assert ex(img, (0, 1000))[2] == 0  # monocular: everything is written from the back


def test_wide_batch_delivered_in_order_into_pinned_arrays(ctx):
    """A batch of more than eight images whose result arrays lie in pinned memory (ft_host_malloc): the device writes keypoints and
    descriptors there itself, in the order ORBextractor::operator() returns them (ORBextractor.cc:1466-1487) - equal to the
    oracle and to the staged path (pageable arrays, the host's pass) for several lapping areas, a repaired image included."""
    w, h, nf, B = 512, 512, 2000, 11
    imgs = [synth.make_image(w, h, seed=300 + b) for b in range(B)]
    imgs[4] = synth.make_noise(w, h, seed=9)  # more candidates in a level than the device octree's first tier sorts
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    oex = ob.Extractor(nf)
    for lap in [(0, 511), (120, 420), (0, 0)]:
        d0 = _calls(ctx, "extract.delivered_in_order_on_device")
        got = ex.extract_batch(imgs, lap, pinned=True)
        assert _calls(ctx, "extract.delivered_in_order_on_device") == d0 + 1
        staged = ex.extract_batch(imgs, lap)
        assert _calls(ctx, "extract.delivered_in_order_on_device") == d0 + 1
        for b in range(B):
            ok, od, om = oex.extract(imgs[b], lap)
            _check_same(got[b][0], got[b][1], ok, od)
            _check_same(staged[b][0], staged[b][1], ok, od)
            assert got[b][2] == om == staged[b][2]
    # rows too short for an image: the error of the staged path, nothing written beyond the rows
    import ctypes as C
    from fasttrack_amd import _capi
    ptrs, keep = orb.ORBextractor.image_ptrs([np.ascontiguousarray(im) for im in imgs], False)
    cap = 100
    kps, desc = ctx.pinned_array((B + 1, cap), orb.KP_DTYPE), ctx.pinned_array((B + 1, cap, 32), np.uint8)
    kps.view(np.uint8)[:] = 0xAB
    n, nm = np.zeros(B, np.int32), np.zeros(B, np.int32)
    rc = _capi.lib().ft_extract_batch(ex._h, ptrs, B, 0, w, h, w, 0, 511, orb.ptr(kps), orb.ptr(desc), cap, orb.ptr(n), orb.ptr(nm))
    assert rc != 0
    assert (kps.view(np.uint8) == 0xAB).all()


def test_edge_images(ctx):
    w, h, nf = 640, 480, 1000
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    oex = ob.Extractor(nf)
    # featureless: no keypoints at all
    flat = synth.make_flat(w, h)
    gk, gd, gm = ex(flat)
    assert len(gk) == 0 and gm == 0
    assert len(oex.extract(flat)[0]) == 0
    # uniform noise: dense candidates (capacity paths), low-contrast: the minThFAST tier
    for img in (synth.make_noise(w, h, 3), (synth.make_image(w, h, 9) // 4 + 90).astype(np.uint8)):
        gk, gd, gm = ex(img)
        ok, od, om = oex.extract(img)
        _check_same(gk, gd, ok, od)
    # empty input: ORBextractor::operator() returns -1
    assert ex(np.zeros((0, 0), np.uint8))[2] == -1
    # strided rows
    big = np.zeros((h, w + 24), np.uint8)
    img = synth.make_image(w, h, 4)
    big[:, :w] = img
    import ctypes as C
    from fasttrack_amd._capi import lib, ptr, check, KP_DTYPE
    cap = ex.max_keypoints
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n, nm = C.c_int(), C.c_int()
    check(lib().ft_extract(ex._h, ptr(big), w, h, big.strides[0], 0, 0, ptr(kps), ptr(desc), cap, C.byref(n), C.byref(nm)))
    ok, od, _ = oex.extract(img)
    _check_same(kps[:n.value], desc[:n.value], ok, od)


def test_small_and_odd_sizes(ctx):
    for (w, h, nf, levels) in [(97, 83, 200, 3), (160, 120, 300, 4), (333, 257, 500, 8), (1000, 96, 400, 2),
                               (1919, 1079, 1500, 8), (64, 400, 100, 1)]:
        img = synth.make_image(w, h, seed=w)
        ex = orb.ORBextractor(ctx, nf, 1.2, levels, 20, 7, w, h)
        oex = ob.Extractor(nf, 1.2, levels)
        gk, gd, gm = ex(img)
        ok, od, om = oex.extract(img)
        _check_same(gk, gd, ok, od)
        for level in range(levels):
            assert np.array_equal(ex.image_pyramid_level(level), oex.level(level)), (w, h, level)


def test_other_parameters(ctx):
    w, h = 640, 480
    img = synth.make_image(w, h, seed=21)
    for (nf, sf, levels, ini, mn) in [(500, 1.5, 5, 30, 10), (1500, 1.1, 10, 12, 5), (800, 2.0, 4, 20, 7)]:
        ex = orb.ORBextractor(ctx, nf, sf, levels, ini, mn, w, h)
        oex = ob.Extractor(nf, sf, levels, ini, mn)
        gk, gd, gm = ex(img)
        ok, od, om = oex.extract(img)
        _check_same(gk, gd, ok, od)


def test_batch_equals_single_and_device_resident(ctx):
    w, h, nf, B = 752, 480, 1200, 5
    imgs = [synth.make_image(w, h, seed=100 + b) for b in range(B)]
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B)
    oex = ob.Extractor(nf)
    res = ex.extract_batch(imgs)
    dev = [ctx.to_device(im) for im in imgs]
    res_dev = ex.extract_batch(dev, on_device=True, width=w, height=h, stride=w)
    for b in range(B):
        ok, od, om = oex.extract(imgs[b])
        _check_same(res[b][0], res[b][1], ok, od)
        _check_same(res_dev[b][0], res_dev[b][1], ok, od)
    # pyramids of every slot stay resident
    for b in (0, B - 1):
        oex.extract(imgs[b])
        assert np.array_equal(ex.image_pyramid_level(3, slot=b), oex.level(3))


def test_getters_match_oracle_tables(ctx):
    ex = orb.ORBextractor(ctx, 1200, 1.2, 8, 20, 7, 752, 480)
    sf, inv = ob.scale_factors(1.2, 8)
    assert np.array_equal(ex.GetScaleFactors(), sf) and np.array_equal(ex.GetInverseScaleFactors(), inv)
    assert np.array_equal(ex.GetScaleSigmaSquares(), sf * sf)
    assert np.array_equal(ex.features_per_level(), ob.features_per_level(1200, 1.2, 8))
    lw, lh = ob.level_sizes(752, 480, 1.2, 8)
    assert [ex.level_size(l) for l in range(8)] == list(zip(lw.tolist(), lh.tolist()))


def _calls(ctx, name):
    try:
        return ctx.get_stat(name)[1]
    except Exception:
        return 0


def test_device_octree_equals_host_octree(ctx, monkeypatch):
    """k_octree (default) and the host octree (option device_octree=0, taken when the extractor is created) select
    the same keypoints in the same order - and both equal the oracle."""
    for (w, h, nf) in [(752, 480, 1200), (1280, 720, 2000), (320, 240, 500)]:
        with ctx.options(device_octree=0):
            ex_host = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=3)
        ex_dev = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=3)
        imgs = [synth.make_image(w, h, seed=40 + i, density=d) for i, d in enumerate((1.0, 0.3, 2.0))]
        before = _calls(ctx, "extract.device_octree_batches"), _calls(ctx, "extract.device_octree_fallbacks")
        rh = ex_host.extract_batch(imgs)
        assert _calls(ctx, "extract.device_octree_batches") == before[0], "device_octree=0 must keep the octree on the host"
        rd = ex_dev.extract_batch(imgs)
        assert _calls(ctx, "extract.device_octree_batches") == before[0] + 1, "the device octree did not run"
        assert _calls(ctx, "extract.device_octree_fallbacks") == before[1], "unexpected fallback to the host octree"
        oex = ob.Extractor(nf)
        for img, (hk, hd, hm), (dk, dd, dm) in zip(imgs, rh, rd):
            ok, od, om = oex.extract(img)
            _check_same(dk, dd, ok, od)
            _check_same(hk, hd, ok, od)
            assert hm == om and dm == om


def test_device_octree_overflow_falls_back_to_host(ctx):
    """more candidates in one level than k_octree sorts in LDS (FT_OCT_MAXN = 4096): the kernel raises its
    overflow flag and the batch is redone with the host octree - same result as the oracle, and the next
    batch goes back to the device."""
    w, h, nf = 1280, 720, 2000
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=2)
    oex = ob.Extractor(nf)
    noisy, calm = synth.make_noise(w, h, seed=3), synth.make_image(w, h, seed=4)
    f0, b0 = _calls(ctx, "extract.device_octree_fallbacks"), _calls(ctx, "extract.device_octree_batches")
    res = ex.extract_batch([noisy, calm])
    oex.extract(noisy)
    assert max(len(oex.candidates(l)) for l in range(8)) > 4096, "test input no longer overflows"
    assert _calls(ctx, "extract.device_octree_fallbacks") == f0 + 1
    for img, (gk, gd, gm) in zip([noisy, calm], res):
        ok, od, om = oex.extract(img)
        _check_same(gk, gd, ok, od)
    res = ex.extract_batch([calm, calm])
    assert _calls(ctx, "extract.device_octree_fallbacks") == f0 + 1
    assert _calls(ctx, "extract.device_octree_batches") == b0 + 2
    ok, od, om = oex.extract(calm)
    _check_same(res[1][0], res[1][1], ok, od)


def test_device_octree_second_tier(ctx):
    """latency mode (small batches): one kernel (k_octree_auto) picks the formulation per level - levels with more than
    FT_OCT_MAXN = 4 096 candidates (and every level with plenty of candidates for its quota) take the histogram formulation,
    the others the sorted one: dense frames stay on the device from the first frame on; every result equals the oracle's"""
    w, h, nf = 1280, 720, 2000
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=2)
    oex = ob.Extractor(nf)
    dense = synth.make_mosaic_pair(w, h, seed=5, block=12)
    expect = []
    for img in dense:
        ok, od, om = oex.extract(img)
        counts = [len(oex.candidates(l)) for l in range(8)]
        assert max(counts) > 4096 and max(counts) <= 16384, counts
        expect.append((ok, od))
    f0 = _calls(ctx, "extract.device_octree_fallbacks")
    for k in range(3):
        res = ex.extract_batch([dense[0], dense[1]])
        for (gk, gd, gm), (ok, od) in zip(res, expect):
            _check_same(gk, gd, ok, od)
    assert _calls(ctx, "extract.device_octree_fallbacks") == f0, "a dense frame fell back to the host octree"


def _emission_order(pts, wCell, hCell, nCols, nRows):
    """cell row, cell column, then row-major inside the cell (ORBextractor.cc:1136-1199); pts relative to the border"""
    x, y = pts[:, 0] - 3, pts[:, 1] - 3
    cj, ci = np.minimum(x // wCell, nCols - 1), np.minimum(y // hCell, nRows - 1)
    return np.lexsort((x, y, cj, ci))


@pytest.mark.parametrize("w,h,nf", [(1280, 720, 2000), (752, 480, 1200), (640, 480, 1000), (1920, 400, 3000)])
def test_device_octree_tiers_on_directed_candidates(ctx, w, h, nf):
    """The three device formulations of DistributeOctTree (ORBextractor.cc:660-884) on candidate sets made for them, through
    the test tap: k_octree (sorted keys, <= 4 096), k_octree_hist (histogram over the tree nodes of depth D, any count), and
    k_octree_big (sorted, <= 16 384) for the levels whose tree outgrows the histogram - uniform and clustered sets, ties in
    the responses, counts beyond 16 384 - all equal to the oracle in content and order."""
    import ctypes as C
    from fasttrack_amd import _capi
    L = 8
    lw, lh, quota, nc, nr, wc, hc = [np.zeros(L, np.int32) for _ in range(7)]
    assert _capi.lib().ft_level_geometry(w, h, nf, 1.2, L, *[_capi.ptr(a) for a in (lw, lh, quota, nc, nr, wc, hc)]) == 0
    ex = orb.ORBextractor(ctx, nf, 1.2, L, 20, 7, w, h, max_batch=1)
    rng = np.random.default_rng(w + nf)
    tiers_seen = set()
    for trial in range(36):
        level = int(rng.integers(0, 3)) if trial % 4 else int(rng.integers(0, L))
        W, H = int(lw[level]) - 32, int(lh[level]) - 32  # maxBorder - minBorder (border 16)
        kind = trial % 6
        n = int(rng.integers(4200, 14000)) if kind < 4 else int(rng.integers(17000, 30000)) if kind == 4 else int(rng.integers(50, 4000))
        n = min(n, (W - 6) * (H - 6) // 5)
        if kind in (1, 3):  # a few dense clusters: the tree goes deep where the candidates are
            k = int(rng.integers(1, 5))
            cx, cy = rng.integers(3, W - 3, k), rng.integers(3, H - 3, k)
            sel = rng.integers(0, k, n)
            sd = (W / 40 if kind == 1 else W / 12)
            pts = np.stack([np.clip(rng.normal(cx[sel], sd), 3, W - 4), np.clip(rng.normal(cy[sel], sd * H / W), 3, H - 4)], 1).astype(np.int64)
            # + a sparse background: it keeps the node list growing pass after pass, so the tree really goes deep
            pts = np.concatenate([pts, np.stack([rng.integers(3, W - 3, 120), rng.integers(3, H - 3, 120)], 1)])
        else:
            pts = np.stack([rng.integers(3, W - 3, n), rng.integers(3, H - 3, n)], 1)
        pts = np.unique(pts, axis=0)
        pts = pts[_emission_order(pts, int(wc[level]), int(hc[level]), int(nc[level]), int(nr[level]))]
        resp = rng.integers(7, 12 if trial % 2 else 200, (len(pts), 1))  # narrow range: many ties for the first-maximum rule
        xys = np.concatenate([pts, resp], 1).astype(np.int32)
        want = xys[ob.distribute_octree(xys, 16, 16 + W, 16, 16 + H, int(quota[level]))]
        shuffled = xys[rng.permutation(len(xys))]  # the device ranks the candidates by their coordinates
        got, tier = ex.octree_on_device(level, shuffled, tiers=7)
        if tier == 0:  # the histogram gave up and the level has more keys than the sorted big tier holds: left to the host
            assert kind in (1, 3) and len(xys) > 4096 and len(got) == 0, (trial, level, len(xys), kind)
            continue
        assert tier in (1, 2, 3), (trial, level, len(xys), tier)
        assert np.array_equal(got, want), (trial, level, len(xys), tier)
        tiers_seen.add(tier)
        if tier == 2 and len(xys) <= 16384:  # the sorted big tier on the same level (it holds 8 192 keys when the quotas are large)
            got3, t3 = ex.octree_on_device(level, shuffled, tiers=5)
            assert t3 in (0, 3) and (t3 == 3 or len(xys) > 8192), (trial, level, len(xys), t3)
            if t3 == 3:
                assert np.array_equal(got3, want), (trial, level, len(xys))
                tiers_seen.add(3)
        # the kernel of latency-mode launches (k_octree_auto: formulation chosen per level, the sorted one taking over in the
        # same workgroup when the histogram gives up)
        gota, ta = ex.octree_on_device(level, shuffled, tiers=15)
        assert ta in (2, 3) and np.array_equal(gota, want), (trial, level, len(xys), ta)
        if tier != 1:  # and with neither: left to the host, nothing half-written
            got0, t0 = ex.octree_on_device(level, shuffled, tiers=1)
            assert t0 == 0 and len(got0) == 0
    assert {1, 2, 3} <= tiers_seen, tiers_seen
    ex.close()


def test_device_octree_histogram_gives_up_to_sorted_tier(ctx):
    """a level whose quota is spent inside one small cluster (a sparse background keeps the list growing pass after pass -
    a cluster alone ends the reference's loop early through `lNodes.size() == prevSize`, ORBextractor.cc:797): the tree grows
    deeper than the histogram's table, k_octree_hist gives up and hands the level to k_octree_big; with that tier not allowed
    the level is left to the host"""
    w, h, nf = 1280, 720, 2000
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=1)
    W, H = w - 32, h - 32
    rng = np.random.default_rng(11)
    ys, xs = np.mgrid[200:290, 300:390]  # 8 100 candidates in a 90 x 90 px block
    bg = np.stack([rng.integers(3, W - 3, 150), rng.integers(3, H - 3, 150)], 1)
    pts = np.unique(np.concatenate([np.stack([xs.ravel(), ys.ravel()], 1), bg]), axis=0)
    quota = ex.features_per_level()[0]
    lw, lh, q, nc, nr, wc, hc = [np.zeros(8, np.int32) for _ in range(7)]
    from fasttrack_amd import _capi
    assert _capi.lib().ft_level_geometry(w, h, nf, 1.2, 8, *[_capi.ptr(a) for a in (lw, lh, q, nc, nr, wc, hc)]) == 0
    pts = pts[_emission_order(pts, int(wc[0]), int(hc[0]), int(nc[0]), int(nr[0]))]
    xys = np.concatenate([pts, rng.integers(7, 40, (len(pts), 1))], 1).astype(np.int32)
    want = xys[ob.distribute_octree(xys, 16, 16 + W, 16, 16 + H, int(quota))]
    got, tier = ex.octree_on_device(0, xys, tiers=7)
    assert tier == 3 and np.array_equal(got, want)
    got, tier = ex.octree_on_device(0, xys, tiers=3)
    assert tier == 0 and len(got) == 0
    ex.close()


def test_two_host_threads_two_extractors(ctx):
    """Frame's constructor runs the left and the right extractor on two host threads (src/Frame.cc:127-130): two
    extractors of one context called concurrently (ctypes releases the GIL) give the single-threaded results, and a
    second context on the same device works next to the first."""
    import threading
    w, h, nf = 752, 480, 1200
    imgs = [synth.make_image(w, h, seed=60 + i) for i in range(6)]
    exL = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    ref = [exL(im) for im in imgs]
    ctx2 = orb.Context(0)
    for other in (ctx, ctx2):  # same context (the reference's layout), then a second context on the device
        exR = orb.ORBextractor(other, nf, 1.2, 8, 20, 7, w, h)
        got = {0: [], 1: []}
        err = []

        def work(k, ex):
            try:
                for _ in range(3):
                    got[k] = [ex(im) for im in imgs]
            except Exception as e:  # pragma: no cover
                err.append(e)
        ts = [threading.Thread(target=work, args=(0, exL)), threading.Thread(target=work, args=(1, exR))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not err, err
        for k in (0, 1):
            for (rk, rd, rm), (gk, gd, gm) in zip(ref, got[k]):
                assert np.array_equal(rk, gk) and np.array_equal(rd, gd) and rm == gm
        del exR
    ctx2.close()


def test_host_threads_on_shared_lanes(ctx):
    """Wide extractors (max_batch > 16) do not own their streams: they run on the context's lanes, which several extractors
    share (ft_host.h).  Four of them - as many as the lane table has rows - driven by four host threads at once, with host
    frames (the shared upload stream) and with resident ones, give the single-threaded results."""
    import threading
    w, h, nf, B = 640, 480, 1000, 24
    imgs = [synth.make_image(w, h, seed=300 + i) for i in range(B)]
    exs = [orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=32) for _ in range(4)]
    ref = exs[0].extract_batch(imgs)
    dev = [ctx.to_device(im) for im in imgs]
    got, err = {}, []

    def work(k):
        try:
            for it in range(3):
                if (k + it) & 1:
                    got[k] = exs[k].extract_batch(dev, on_device=True, width=w, height=h, stride=w)
                else:
                    got[k] = exs[k].extract_batch(imgs)
        except Exception as e:  # pragma: no cover
            err.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err
    for k in range(4):
        for (rk, rd, rm), (gk, gd, gm) in zip(ref, got[k]):
            assert np.array_equal(rk, gk) and np.array_equal(rd, gd) and rm == gm


def test_random_configurations_bit_exact(ctx):
    """seeded sweep over image sizes, feature counts, pyramid depths, scale factors and FAST thresholds (cells of every
    width class: fixed-pitch 48 / 64 and the any-size fallback; levels with a single cell row; quotas above and below the
    candidate count): keypoints and descriptors equal the oracle's"""
    rng = np.random.default_rng(2024)
    checked = 0
    for trial in range(24):
        w = int(rng.integers(70, 1000))
        h = int(rng.integers(70, 760))
        nf = int(rng.choice([150, 500, 1000, 2500, 4000]))
        nlevels = int(rng.integers(2, 11))
        sfac = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
        ini, mn = (20, 7) if trial % 3 else (int(rng.integers(12, 40)), int(rng.integers(3, 12)))
        if min(w, h) / (sfac ** (nlevels - 1)) < 40:  # the last level must still hold a 35-px cell inside its borders
            nlevels = max(2, int(np.log(min(w, h) / 40.0) / np.log(sfac)) + 1)
        img = synth.make_image(w, h, seed=300 + trial, density=float(rng.choice([0.3, 1.0, 2.5])))
        ex = orb.ORBextractor(ctx, nf, sfac, nlevels, ini, mn, w, h)
        oex = ob.Extractor(nf, sfac, nlevels, ini, mn)
        gk, gd, gm = ex(img)
        ok, od, om = oex.extract(img)
        _check_same(gk, gd, ok, od)
        assert gm == om
        checked += len(ok)
        del ex
    assert checked > 5000
