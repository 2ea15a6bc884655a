#!/usr/bin/env python3
"""Latency of Frame::ComputeBoW (ft_bow_transform) on a vocabulary of ORBvoc.txt's shape (k = 10, L = 6: 1.1 M nodes,
10^6 words) for the descriptors of one frame, against the oracle (DBoW2's transform restated) on one host core - and of
ORBmatcher::SearchByBoW (ft_search_by_bow) between two such frames (left / right image of a pair as keyframe / frame)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fasttrack_amd import orb, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402


def med(f, reps):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


def main():
    ctx = orb.Context(0)
    voc = synth.make_vocabulary(10, 6, seed=1)
    args = (voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    gv, ov = orb.Vocabulary(ctx, 10, 6, 0, 0, *args), ob.Vocabulary(10, 6, 0, 0, *args)
    out = {"vocabulary": {"k": 10, "L": 6, "nodes": gv.n_nodes, "words": gv.n_words}}
    w, h = 752, 480
    intr = synth.intrinsics(w, h)
    for nf in (1200, 2000):
        fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, 1, intr["mbf"], intr["mb"])
        L, R = synth.make_stereo_pair(w, h, 3)[:2]
        o = fe.process([L], [R])[0]
        d = o["descL"]
        a, b = gv.transform(d, 4), ov.transform(d, 4)
        assert all(np.array_equal(a[k], b[k]) for k in a)
        dptr = fe.device_descriptors(0)
        out["n%d" % len(d)] = {
            "device_ms_host_descriptors": med(lambda: gv.transform(d, 4), 50),
            "device_ms_resident_descriptors": med(lambda: gv.transform(None, 4, device_ptr=dptr, n=len(d)), 50),
            "oracle_1_core_ms": med(lambda: ov.transform(d, 4), 10),
            "bow_entries": int(len(a["bow_ids"])), "feature_vector_nodes": int(len(a["fv_nodes"]))}
        # SearchByBoW: the right image's features as the keyframe, the left image's as the frame
        dR, kR, kL = o["descR"], o["keysR"], o["keysL"]
        tR = gv.transform(dR, 4)
        has = np.ones(len(dR), np.uint8)
        gK = orb.BowSide(tR["fv_nodes"], tR["fv_offsets"], tR["fv_features"], dR, kR["angle"])
        gF = orb.BowSide(a["fv_nodes"], a["fv_offsets"], a["fv_features"], d, kL["angle"])
        oK = dict(tR, descriptors=dR, angles=kR["angle"]); oF = dict(a, descriptors=d, angles=kL["angle"])
        g, oo = orb.search_by_bow(ctx, gK, has, gF), ob.search_by_bow(oK, has, oF)
        assert g["n"] == oo["n"] and np.array_equal(g["matches"], oo["matches"])
        out["n%d" % len(d)]["search_by_bow"] = {
            "device_ms": med(lambda: orb.search_by_bow(ctx, gK, has, gF), 50),
            "oracle_1_core_ms": med(lambda: ob.search_by_bow(oK, has, oF), 20), "matches": int(g["n"]),
            "common_nodes": int(len(np.intersect1d(tR["fv_nodes"], a["fv_nodes"])))}
        fe.close()
    out["note"] = "median wall time per call including the Python marshalling"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
