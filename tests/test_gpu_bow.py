"""Parity of ft_bow_transform (Frame::ComputeBoW, DBoW2 TemplatedVocabulary::transform with the tree walk on the
device) against the oracle: word / node / weight of every feature, BowVector ids and double values bit for bit,
FeatureVector nodes and feature lists.  Needs an MI355X."""
import os

import numpy as np
import pytest

from fasttrack_amd import orb, synth
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = orb.Context(0)
    yield c
    c.close()


def _same(a, b):
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def _descriptors(voc, n, seed):
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    m = n // 4
    d[:m] = voc["descriptors"][rng.integers(1, len(voc["parent"]), m)]  # exact centres: distance 0 and ties between siblings
    flips = rng.integers(0, 256, (m, 3))
    for i in range(m):  # near-centre descriptors
        b = np.unpackbits(d[m + i].copy() if False else voc["descriptors"][rng.integers(1, len(voc["parent"]))])
        b[flips[i]] ^= 1
        d[m + i] = np.packbits(b)
    return d


@pytest.mark.parametrize("k,L,ragged", [(10, 4, False), (10, 3, True), (3, 6, True), (20, 2, False), (16, 3, False), (1, 3, False)])
def test_transform_equals_oracle(ctx, k, L, ragged):
    voc = synth.make_vocabulary(k, L, seed=100 + k + L, ragged=ragged)
    args = (voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    for scoring, weighting in [(0, 0), (1, 1), (5, 0), (5, 2), (3, 3)]:
        gv = orb.Vocabulary(ctx, k, L, scoring, weighting, *args)
        ov = ob.Vocabulary(k, L, scoring, weighting, *args)
        assert (gv.n_nodes, gv.n_words) == (ov.n_nodes, ov.n_words)
        for n in (1, 15, 17, 64, 1203):
            d = _descriptors(voc, n, seed=n + scoring)
            for levelsup in (0, 2, 4, L + 1):
                _same(ov.transform(d, levelsup), gv.transform(d, levelsup))
        gv.close()


def test_orbvoc_shape_text_file_and_extractor_descriptors(ctx, tmp_path):
    """a tree of ORBvoc.txt's shape (k = 10, L = 6 would be 1.1 M nodes; L = 5 keeps the test short) loaded from the
    text format, fed with the descriptors of a real extraction - from host memory and where the front end left them"""
    voc = synth.make_vocabulary(10, 5, seed=7)
    path = os.path.join(tmp_path, "voc.txt")
    synth.write_vocabulary_text(voc, path)
    gv, ov = orb.Vocabulary(ctx, path=path), ob.Vocabulary(path=path)
    assert gv.n_words == ov.n_words == 10 ** 5 and (gv.k, gv.L) == (10, 5)
    w, h, nf = 752, 480, 1200
    intr = synth.intrinsics(w, h)
    fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, 2, intr["mbf"], intr["mb"])
    pairs = [synth.make_stereo_pair(w, h, 40 + b) for b in range(2)]
    outs = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
    for b, out in enumerate(outs):
        want = ov.transform(out["descL"], 4)
        _same(want, gv.transform(out["descL"], 4))
        assert abs(want["bow_values"].sum() - 1.0) < 1e-12 and len(want["bow_ids"]) > 0.5 * len(out["descL"])
        dptr = fe.device_descriptors(b, right=False)
        _same(want, gv.transform(None, 4, device_ptr=dptr, n=len(out["descL"])))
    fe.close()
    gv.close()


def test_edge_cases(ctx, tmp_path):
    voc = synth.make_vocabulary(4, 2, seed=3)
    gv = orb.Vocabulary(ctx, 4, 2, 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    e = gv.transform(np.zeros((0, 32), np.uint8), 4)
    assert len(e["bow_ids"]) == 0 and len(e["fv_nodes"]) == 0 and len(e["word"]) == 0
    # every word stopped: nothing is added to either map (w > 0 fails, TemplatedVocabulary.h:1157)
    z = dict(voc)
    z["weights"] = np.zeros_like(voc["weights"])
    gz = orb.Vocabulary(ctx, 4, 2, 0, 0, z["parent"], z["is_leaf"], z["descriptors"], z["weights"])
    r = gz.transform(np.random.default_rng(0).integers(0, 256, (50, 32), dtype=np.uint8), 1)
    assert len(r["bow_ids"]) == 0 and len(r["fv_nodes"]) == 0 and (r["weight"] == 0).all()
    # a root without children is an empty vocabulary
    g0 = orb.Vocabulary(ctx, 4, 2, 0, 0, np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros((1, 32), np.uint8), np.zeros(1))
    r0 = g0.transform(np.zeros((5, 32), np.uint8), 1)
    assert len(r0["bow_ids"]) == 0 and len(r0["fv_nodes"]) == 0
    from fasttrack_amd._capi import FastTrackError
    with pytest.raises(FastTrackError):
        orb.Vocabulary(ctx, 25, 2, 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    bad = voc["parent"].copy()
    bad[3] = 999
    with pytest.raises(FastTrackError):
        orb.Vocabulary(ctx, 4, 2, 0, 0, bad, voc["is_leaf"], voc["descriptors"], voc["weights"])
    with pytest.raises(FastTrackError):
        orb.Vocabulary(ctx, path=os.path.join(tmp_path, "missing.txt"))
    for v in (gv, gz, g0):
        v.close()


def _bow_side(d):
    return orb.BowSide(d["fv_nodes"], d["fv_offsets"], d["fv_features"], d["descriptors"], d["angles"])


@pytest.mark.parametrize("k,L,levelsup", [(10, 5, 3), (10, 4, 4), (3, 6, 2), (20, 2, 1), (6, 4, 0)])
def test_search_by_bow_equals_oracle(ctx, k, L, levelsup):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&) on the FeatureVectors of the device transform: assignments and nmatches
    equal the oracle's - one and two cameras, with and without the rotation histogram, several ratios; nodes from a few to
    hundreds of features (levelsup 0 = one node per word, levelsup >= L = everything in the root's node)."""
    from fasttrack_amd import scenarios as sc
    voc = synth.make_vocabulary(k, L, seed=7 + k, ragged=(k == 3))
    args = (voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    gv, ov = orb.Vocabulary(ctx, k, L, 0, 0, *args), ob.Vocabulary(k, L, 0, 0, *args)
    total = 0
    for seed, (nk, nf) in enumerate([(1500, 2000), (2000, 700), (64, 65), (1, 1), (300, 3000)]):
        for two_cam in (False, True):
            S = sc.bow_match_scenario(voc, gv.transform, nk, nf, 10 * seed + two_cam, two_cam=two_cam, levelsup=levelsup)
            O = sc.bow_match_scenario(voc, ov.transform, nk, nf, 10 * seed + two_cam, two_cam=two_cam, levelsup=levelsup)
            assert np.array_equal(S["f"]["fv_features"], O["f"]["fv_features"])
            for ratio, ori in [(0.7, True), (0.7, False), (0.9, True), (0.6, False)]:
                o = ob.search_by_bow(O["kf"], O["has_point"], O["f"], O["nleft"], ratio, ori)
                g = orb.search_by_bow(ctx, _bow_side(S["kf"]), S["has_point"], _bow_side(S["f"]), S["nleft"], ratio, ori)
                assert g["n"] == o["n"] and np.array_equal(g["matches"], o["matches"]), (nk, nf, two_cam, ratio, ori)
                total += o["n"]
    assert total > 2000
    gv.close()


def test_search_by_bow_one_big_node_and_edge_cases(ctx):
    """levelsup >= L puts every feature into node 0: one wave walks thousands of frame features per keyframe feature and the
    claims of 2000 keyframe features in sequence; empty sides, no map points, bad arguments"""
    from fasttrack_amd import scenarios as sc
    voc = synth.make_vocabulary(4, 3, seed=2)
    args = (voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    gv, ov = orb.Vocabulary(ctx, 4, 3, 0, 0, *args), ob.Vocabulary(4, 3, 0, 0, *args)
    S = sc.bow_match_scenario(voc, ov.transform, 700, 2500, 3, two_cam=True, levelsup=5)
    assert len(S["f"]["fv_nodes"]) == 1
    for ori in (False, True):
        o = ob.search_by_bow(S["kf"], S["has_point"], S["f"], S["nleft"], 0.8, ori)
        g = orb.search_by_bow(ctx, _bow_side(S["kf"]), S["has_point"], _bow_side(S["f"]), S["nleft"], 0.8, ori)
        assert o["n"] > 300 and g["n"] == o["n"] and np.array_equal(g["matches"], o["matches"])
    g = orb.search_by_bow(ctx, _bow_side(S["kf"]), np.zeros(700, np.uint8), _bow_side(S["f"]), -1)
    assert g["n"] == 0 and (g["matches"] == -1).all()
    for nk, nf in ((0, 40), (40, 0)):
        E = sc.bow_match_scenario(voc, ov.transform, nk, nf, 1)
        g = orb.search_by_bow(ctx, _bow_side(E["kf"]), E["has_point"], _bow_side(E["f"]), -1)
        assert g["n"] == 0 and len(g["matches"]) == nf
    bad = dict(S["f"]); bad["fv_features"] = S["f"]["fv_features"].copy(); bad["fv_features"][3] = 1 << 21
    with pytest.raises(Exception):
        orb.search_by_bow(ctx, _bow_side(S["kf"]), S["has_point"], _bow_side(bad), -1)
    with pytest.raises(Exception):
        orb.search_by_bow(ctx, _bow_side(S["kf"]), S["has_point"], _bow_side(S["f"]), 5000)
    gv.close()

