"""Frame::ComputeBoW on the CPU side: the oracle's restatement of DBoW2's TemplatedVocabulary::transform
(Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1253) against an independent plain-Python walk, the text format
round trip (loadFromTextFile :1338-1423), weighting / scoring variants.  No GPU."""
import os

import numpy as np
import pytest

from fasttrack_amd import synth
from oracle import binding as ob

POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def py_transform(voc, scoring, weighting, desc, levelsup):
    parent = voc["parent"]
    children = {}
    for i in range(1, len(parent)):
        children.setdefault(int(parent[i]), []).append(i)
    words = {}
    for i in range(1, len(parent)):
        if voc["is_leaf"][i]:
            words[i] = len(words)
    L = voc["L"]
    bow, fv, per = {}, {}, []
    for fi, d in enumerate(desc):
        node, level, nid = 0, 0, 0
        while node in children:
            level += 1
            ch = children[node]
            dist = [int(POP[np.bitwise_xor(d, voc["descriptors"][c])].sum()) for c in ch]
            node = ch[int(np.argmin(dist))]  # first minimum
            if level == L - levelsup:
                nid = node
        if L - levelsup > level:
            nid = node
        w = float(voc["weights"][node])
        per.append((words[node], nid, w))
        if w > 0:
            if weighting in (0, 1):
                bow[words[node]] = bow.get(words[node], 0.0) + w if words[node] in bow else w
            else:
                bow.setdefault(words[node], w)
            fv.setdefault(nid, []).append(fi)
    ids = sorted(bow)
    vals = [bow[i] for i in ids]
    must = scoring != 5
    if weighting in (0, 1) and vals and not must:
        vals = [v / float(len(vals)) for v in vals]
    if must:
        norm = 0.0
        if scoring != 1:
            for v in vals:
                norm += abs(v)
        else:
            for v in vals:
                norm += v * v
            norm = float(np.sqrt(np.float64(norm)))
        if norm > 0:
            vals = [v / norm for v in vals]
    return per, ids, vals, fv


@pytest.mark.parametrize("k,L,ragged", [(4, 3, False), (10, 3, True), (3, 5, True)])
@pytest.mark.parametrize("scoring,weighting", [(0, 0), (1, 1), (5, 0), (5, 2), (2, 3)])
def test_oracle_transform_equals_plain_python(k, L, ragged, scoring, weighting):
    voc = synth.make_vocabulary(k, L, seed=11 * k + L, ragged=ragged)
    ov = ob.Vocabulary(k, L, scoring, weighting, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    rng = np.random.default_rng(k + L)
    desc = rng.integers(0, 256, (150, 32), dtype=np.uint8)
    desc[:40] = voc["descriptors"][rng.integers(1, len(voc["parent"]), 40)]  # exact node centres: distance-0 hits and ties
    for levelsup in (0, 1, L - 1, L, L + 2):
        r = ov.transform(desc, levelsup)
        per, ids, vals, fv = py_transform(voc, scoring, weighting, desc, levelsup)
        assert [tuple(x) for x in zip(r["word"].tolist(), r["node"].tolist(), r["weight"].tolist())] == per
        assert r["bow_ids"].tolist() == ids and r["bow_values"].tolist() == vals  # doubles bit for bit
        assert r["fv_nodes"].tolist() == sorted(fv)
        for j, nid in enumerate(sorted(fv)):
            assert r["fv_features"][r["fv_offsets"][j]:r["fv_offsets"][j + 1]].tolist() == fv[nid]


def test_text_format_round_trip(tmp_path):
    voc = synth.make_vocabulary(6, 3, seed=5, ragged=True)
    path = os.path.join(tmp_path, "voc.txt")
    synth.write_vocabulary_text(voc, path, scoring=0, weighting=0)
    a = ob.Vocabulary(6, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    b = ob.Vocabulary(path=path)
    assert (a.n_nodes, a.n_words) == (b.n_nodes, b.n_words) == (len(voc["parent"]), int(voc["is_leaf"].sum()))
    desc = np.random.default_rng(1).integers(0, 256, (300, 32), dtype=np.uint8)
    ra, rb = a.transform(desc, 2), b.transform(desc, 2)
    assert all(np.array_equal(ra[k], rb[k]) for k in ra)
    with open(os.path.join(tmp_path, "bad.txt"), "w") as f:
        f.write("25 3 0 0\n")
    with pytest.raises(ValueError):
        ob.Vocabulary(path=os.path.join(tmp_path, "bad.txt"))


def test_l1_bow_vector_sums_to_one_and_empty_input():
    voc = synth.make_vocabulary(10, 3, seed=2)
    ov = ob.Vocabulary(10, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    desc = np.random.default_rng(3).integers(0, 256, (1000, 32), dtype=np.uint8)
    r = ov.transform(desc, 4)
    assert abs(r["bow_values"].sum() - 1.0) < 1e-12 and (np.diff(r["bow_ids"].astype(np.int64)) > 0).all()
    assert (r["node"] == 0).all()  # levelsup >= L: every feature files under the root
    assert r["fv_offsets"][-1] == (r["weight"] > 0).sum()
    e = ov.transform(np.zeros((0, 32), np.uint8), 4)
    assert len(e["bow_ids"]) == 0 and len(e["fv_nodes"]) == 0


def py_search_by_bow(kf, has_point, f, nleft, nn_ratio, check_orientation):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (/root/reference/src/ORBmatcher.cc:322-524) with Python dicts for the
    two FeatureVectors, numpy for the Hamming distances: an independent statement for small cases"""
    fvK = {int(n): kf["fv_features"][kf["fv_offsets"][j]:kf["fv_offsets"][j + 1]] for j, n in enumerate(kf["fv_nodes"])}
    fvF = {int(n): f["fv_features"][f["fv_offsets"][j]:f["fv_offsets"][j + 1]] for j, n in enumerate(f["fv_nodes"])}
    N = len(f["descriptors"])
    matches = np.full(N, -1, np.int32)
    rot_hist = [[] for _ in range(30)]
    n = 0
    bits = lambda a, b: int(np.unpackbits(np.bitwise_xor(a, b)).sum())

    def push(k_idx, f_idx):
        rot = np.float32(kf["angles"][k_idx]) - np.float32(f["angles"][f_idx])
        if rot < 0:
            rot = np.float32(rot + np.float32(360.0))
        v = float(np.float32(rot * np.float32(1.0 / 30)))
        b = int(np.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))   # roundf: half away from zero
        rot_hist[0 if b == 30 else b].append(f_idx)
    for node in sorted(set(fvK) & set(fvF)):
        for k_idx in fvK[node]:
            if not has_point[k_idx]:
                continue
            b1, i1, b2, b1r, i1r, b2r = 256, -1, 256, 256, -1, 256
            for f_idx in fvF[node]:
                if matches[f_idx] >= 0:
                    continue
                d = bits(kf["descriptors"][k_idx], f["descriptors"][f_idx])
                if nleft == -1 or f_idx < nleft:
                    if d < b1:
                        b2, b1, i1 = b1, d, f_idx
                    elif d < b2:
                        b2 = d
                else:
                    if d < b1r:
                        b2r, b1r, i1r = b1r, d, f_idx
                    elif d < b2r:
                        b2r = d
            if b1 <= 50:
                if np.float32(b1) < np.float32(nn_ratio) * np.float32(b2):
                    matches[i1] = k_idx
                    if check_orientation:
                        push(k_idx, i1)
                    n += 1
                if b1r <= 50:
                    matches[i1r] = k_idx
                    if check_orientation:
                        push(k_idx, i1r)
                    n += 1
    if check_orientation:
        # ComputeThreeMaxima (/root/reference/src/ORBmatcher.cc:2210-2251)
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for b, sz in enumerate(len(h) for h in rot_hist):
            if sz > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, sz, i2, i1, b
            elif sz > m2:
                m3, m2, i3, i2 = m2, sz, i2, b
            elif sz > m3:
                m3, i3 = sz, b
        if m2 < np.float32(0.1) * np.float32(m1):
            i2 = i3 = -1
        elif m3 < np.float32(0.1) * np.float32(m1):
            i3 = -1
        keep = (i1, i2, i3)
        for b in range(30):
            if b not in keep:
                for f_idx in rot_hist[b]:
                    matches[f_idx] = -1
                    n -= 1
    return matches, n


@pytest.mark.parametrize("two_cam", [False, True])
@pytest.mark.parametrize("check_orientation", [False, True])
def test_oracle_search_by_bow_equals_plain_python(two_cam, check_orientation):
    from fasttrack_amd import scenarios as sc
    voc = synth.make_vocabulary(6, 4, seed=31, ragged=True)
    ov = ob.Vocabulary(6, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    total = 0
    for seed, (nk, nf, lu) in enumerate([(180, 200, 2), (250, 120, 3), (60, 300, 1), (1, 1, 2)]):
        S = sc.bow_match_scenario(voc, ov.transform, nk, nf, seed, two_cam=two_cam, levelsup=lu)
        o = ob.search_by_bow(S["kf"], S["has_point"], S["f"], S["nleft"], 0.7, check_orientation)
        m, n = py_search_by_bow(S["kf"], S["has_point"], S["f"], S["nleft"], 0.7, check_orientation)
        assert np.array_equal(o["matches"], m) and o["n"] == n == int((m >= 0).sum())
        total += n
    assert total > 60


def test_oracle_search_by_bow_edge_cases():
    from fasttrack_amd import scenarios as sc
    voc = synth.make_vocabulary(5, 3, seed=5)
    ov = ob.Vocabulary(5, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    S = sc.bow_match_scenario(voc, ov.transform, 120, 150, 9)
    # no map points at all / empty sides
    o = ob.search_by_bow(S["kf"], np.zeros(120, np.uint8), S["f"], -1, 0.7, True)
    assert o["n"] == 0 and (o["matches"] == -1).all()
    E = sc.bow_match_scenario(voc, ov.transform, 0, 50, 1)
    assert ob.search_by_bow(E["kf"], E["has_point"], E["f"], -1, 0.7, True)["n"] == 0
    E = sc.bow_match_scenario(voc, ov.transform, 50, 0, 1)
    r = ob.search_by_bow(E["kf"], E["has_point"], E["f"], -1, 0.7, True)
    assert r["n"] == 0 and len(r["matches"]) == 0
    # a frame identical to the keyframe, every point good, no orientation check: every feature finds itself unless an earlier
    # keyframe feature of its node took it (duplicates) or the ratio test fails
    d = S["kf"]["descriptors"]
    same = dict(S["kf"])
    o = ob.search_by_bow(S["kf"], np.ones(len(d), np.uint8), same, -1, 0.7, False)
    hit = o["matches"] >= 0
    assert hit.mean() > 0.5 and all((d[o["matches"][i]] == d[i]).all() for i in np.nonzero(hit)[0])

