"""Oracle vs the committed golden vectors (tests/golden/*.npz, made by tests/tools/make_golden_vectors.py) - CPU only.
The vectors were produced by this oracle: they pin it against drift, they do not pin it to OpenCV
(parity unpinned, see oracle/orb_oracle.h)."""
import os

import numpy as np

from oracle import binding as ob
from tests import scenarios as sc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_extract_vectors(golden_dir):
    for name in ("extract_160x120_s1.npz", "extract_320x240_s2_lap.npz"):
        g = _load(golden_dir, name)
        levels = int(g["nlevels"])
        ex = ob.Extractor(int(g["nfeatures"]), 1.2, levels, 20, 7)
        k, d, nm = ex.extract(g["image"], tuple(int(v) for v in g["lap"]))
        assert np.array_equal(k, g["keypoints"]) and np.array_equal(d, g["descriptors"]) and nm == int(g["n_mono"])
        assert [len(ex.candidates(l)) for l in range(levels)] == list(g["cand_counts"])
        assert np.array_equal(ex.candidates(0), g["cand_level0"])
        assert np.array_equal(ex.candidates(levels - 1), g["cand_last"])
        assert np.array_equal(ex.level(1), g["level1"]) and np.array_equal(ex.level(levels - 1), g["level_last"])
        assert int(ex.blurred(1).astype(np.uint64).sum()) == int(g["blurred_level1_crc"][0])


def test_stereo_vectors(golden_dir):
    g = _load(golden_dir, "stereo_320x240_s3.npz")
    nf = int(g["nfeatures"])
    exL, exR = ob.Extractor(nf), ob.Extractor(nf)
    kL, dL, _ = exL.extract(g["left"])
    kR, dR, _ = exR.extract(g["right"])
    assert np.array_equal(kL, g["keysL"]) and np.array_equal(dR, g["descR"])
    sm = ob.stereo_match(exL, exR, kL, kR, dL, dR, float(g["mbf"]), float(g["mb"]))
    assert sm["n"] == int(g["n"]) and np.array_equal(sm["uright"], g["uright"]) and np.array_equal(sm["depth"], g["depth"])
    sm0 = ob.stereo_match(exL, exR, kL, kR, dL, dR, float(g["mbf"]), float(g["mb"]), median_cut=False)
    assert np.array_equal(sm0["sad"], g["sad_nocut"]) and np.array_equal(sm0["hamming_idx"], g["hamming_idx"])
    fm = ob.fisheye_match(dL, dR)
    assert np.array_equal(fm["matches"], g["fisheye_matches"]) and np.array_equal(fm["second"], g["fisheye_second"])


def test_search_vectors(golden_dir):
    g = _load(golden_dir, "search_320x240_s5.npz")
    w, h = int(g["width"]), int(g["height"])
    pts = {k[3:]: g[k] for k in g.files if k.startswith("lp_") and k not in ("lp_th", "lp_assign", "lp_n", "lp_best_dist", "lp_best_dist2", "lp_best_idx")}
    F = ob.FrameView(keys=g["keys"], descriptors=g["descriptors"], scale_factors_=g["sf"], bounds=sc.frame_bounds(w, h),
                     mbf=float(g["mbf"]), mb=float(g["mb"]), uright=g["uright"])
    lo = ob.search_local_points(F, pts, float(g["lp_th"]))
    assert lo["n"] == int(g["lp_n"]) and np.array_equal(lo["assign"], g["lp_assign"])
    assert np.array_equal(lo["best_dist"], g["lp_best_dist"]) and np.array_equal(lo["best_idx"], g["lp_best_idx"])
    last = {k[3:]: g[k] for k in ("lf_valid", "lf_world_pos", "lf_descriptors", "lf_observations", "lf_octave", "lf_angle")}
    F2 = ob.FrameView(keys=g["keys"], descriptors=g["descriptors"], scale_factors_=g["sf"], bounds=sc.frame_bounds(w, h),
                      mbf=float(g["mbf"]), mb=float(g["mb"]), uright=g["uright"], cam=g["cam"])
    la = ob.search_last_frame(F2, last, g["lf_Tcw"], float(g["lf_th"]), False, False, True)
    assert la["n"] == int(g["lf_n"]) and np.array_equal(la["assign"], g["lf_assign"])
    assert np.array_equal(la["best_dist"], g["lf_best_dist"]) and np.array_equal(la["best_idx"], g["lf_best_idx"])


def test_bow_match_vectors(golden_dir):
    g = _load(golden_dir, "bow_match_s6.npz")
    total = 0
    for tag in ("mono", "two"):
        kf = {k: g[f"{tag}_kf_{k}"] for k in ("fv_nodes", "fv_offsets", "fv_features", "descriptors", "angles")}
        f = {k: g[f"{tag}_f_{k}"] for k in ("fv_nodes", "fv_offsets", "fv_features", "descriptors", "angles")}
        for ori in (0, 1):
            r = ob.search_by_bow(kf, g[f"{tag}_has_point"], f, int(g[f"{tag}_nleft"]), 0.7, bool(ori))
            assert r["n"] == int(g[f"{tag}_n_ori{ori}"]) and np.array_equal(r["matches"], g[f"{tag}_matches_ori{ori}"])
            total += r["n"]
    assert total > 300

