#!/usr/bin/env python3
"""Long randomized parity run on an MI355X (not part of the test suite): extractor and fused stereo front end against
the oracle over many random configurations (every fifth trial also Frame::ComputeBoW on a random vocabulary).
usage: tests/tools/soak_parity.py [--trials N] [--seed S]

Every trial draws image size, feature count, pyramid depth, scale factor, FAST thresholds and texture density,
runs the HIP path through the C ABI and compares keypoints (all fields, angles bit-exactly), descriptors and - for
the front end - mvuRight / mvDepth with the oracle.  Prints one line per failure and a summary; exit code 1 on any
mismatch."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fasttrack_amd import orb, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402


def same_keys(gk, gd, ok, od):
    if len(gk) != len(ok):
        return "count %d vs %d" % (len(gk), len(ok))
    for f in ("x", "y", "size", "response", "octave", "class_id", "angle"):
        if not np.array_equal(gk[f], ok[f]):
            return "field " + f
    if not np.array_equal(gd, od):
        return "descriptors"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--frontend-every", type=int, default=4, help="every n-th trial also runs the fused stereo front end")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = orb.Context(0)
    bad = 0
    kps = 0
    t0 = time.time()
    for trial in range(args.trials):
        w = int(rng.integers(70, 1400))
        h = int(rng.integers(70, 800))
        nf = int(rng.choice([100, 300, 500, 1000, 2000, 3000, 5000]))
        nlevels = int(rng.integers(1, 11))
        sfac = float(rng.choice([1.1, 1.15, 1.2, 1.25, 1.3, 1.5, 2.0]))
        ini, mn = (20, 7) if rng.random() < 0.5 else (int(rng.integers(10, 60)), int(rng.integers(2, 15)))
        mn = min(mn, ini)
        if min(w, h) / (sfac ** (nlevels - 1)) < 40:
            nlevels = max(1, int(np.log(min(w, h) / 40.0) / np.log(sfac)) + 1)
        dens = float(rng.choice([0.0, 0.1, 0.3, 1.0, 2.5, 6.0]))
        mosaic = int(rng.choice([0, 0, 0, 6, 9, 14]))  # dense mosaics: levels of 4 k .. 30 k candidates (second octree tier / repair)
        cfg = dict(w=w, h=h, nf=nf, nlevels=nlevels, sfac=sfac, ini=ini, mn=mn, dens=dens, trial=trial, mosaic=mosaic)
        try:
            img = synth.make_image(w, h, seed=int(rng.integers(1 << 30)), density=dens)
            if mosaic:
                img = synth.make_mosaic_pair(w, h, int(rng.integers(1 << 30)), block=mosaic)[0]
            if rng.random() < 0.15:  # noise: many weak corners, dense candidate lists
                img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
            ex = orb.ORBextractor(ctx, nf, sfac, nlevels, ini, mn, w, h)
            oex = ob.Extractor(nf, sfac, nlevels, ini, mn)
            gk, gd, gm = ex(img)
            ok, od, om = oex.extract(img)
            err = same_keys(gk, gd, ok, od) or (None if gm == om else "mono count")
            kps += len(ok)
            if err is None and (mosaic or trial % 7 == 0):
                # the same frame again: a frame that overflowed the first octree tier was repaired on the host the first time
                # and takes the histogram tier (k_octree_hist) now; any other frame replays the captured graph
                gk, gd, gm = ex(img)
                err = same_keys(gk, gd, ok, od) or (None if gm == om else "mono count (second call)")
                if err:
                    err += " (second call)"
            del ex
            if err is None and trial % 6 == 1 and w * h < 500 * 400:
                # a wide batch into pinned arrays: the device delivers the results itself in the reference's output order
                # (k_deliver_ordered), random lapping area; the same batch into pageable arrays (staging + the host's pass)
                Bw = int(rng.integers(9, 15))
                lap = (0, w) if rng.random() < 0.3 else tuple(sorted(int(v) for v in rng.integers(0, w, size=2)))
                imgs = [img] + [synth.make_image(w, h, seed=int(rng.integers(1 << 30)), density=dens) for _ in range(Bw - 1)]
                exw = orb.ORBextractor(ctx, nf, sfac, nlevels, ini, mn, w, h, max_batch=Bw)
                got, staged = exw.extract_batch(imgs, lap, pinned=True), exw.extract_batch(imgs, lap)
                for im, g, st in zip(imgs, got, staged):
                    k2, d2, m2 = oex.extract(im, lap)
                    err = err or same_keys(g[0], g[1], k2, d2) or same_keys(st[0], st[1], k2, d2) or (None if g[2] == m2 == st[2] else "mono count (wide batch)")
                    kps += len(k2)
                if err:
                    err += " (wide batch, lapping area %s)" % (lap,)
                del exw
            if err is None and trial % 5 == 0 and len(ok) > 0:  # Frame::ComputeBoW on this frame's descriptors
                k, Lv = int(rng.integers(1, 21)), int(rng.integers(1, 5))
                if k ** Lv > 20000:
                    Lv = max(1, int(np.log(20000) / np.log(max(k, 2))))
                voc = synth.make_vocabulary(k, Lv, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(2)),
                                            flip_bits=int(rng.choice([4, 40, 128])))
                sc, wt = int(rng.integers(0, 6)), int(rng.integers(0, 4))
                va = (voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
                gv, ov = orb.Vocabulary(ctx, k, Lv, sc, wt, *va), ob.Vocabulary(k, Lv, sc, wt, *va)
                lu = int(rng.integers(0, Lv + 2))
                a, b = gv.transform(od, lu), ov.transform(od, lu)
                if not all(np.array_equal(a[key], b[key]) for key in a):
                    err = "bow transform k=%d L=%d scoring=%d weighting=%d levelsup=%d" % (k, Lv, sc, wt, lu)
                gv.close()
            if err is None and trial % args.frontend_every == 0 and nlevels >= 2:
                # latency mode (1-3 pairs, captured graph, both cameras in one launch) or, for small frames now and then,
                # throughput mode (more than 16 pairs: per-pair repair, second-tier grid sized from the previous batch)
                big = w * h < 400 * 300 and rng.random() < 0.5
                B = int(rng.integers(17, 22)) if big else int(rng.integers(1, 4))
                intr = synth.intrinsics(w, h)
                fe = orb.StereoFrontend(ctx, nf, sfac, nlevels, ini, mn, w, h, B, intr["mbf"], intr["mb"])
                pairs = [synth.make_mosaic_pair(w, h, int(rng.integers(1 << 30)), block=mosaic) if mosaic and rng.random() < 0.5
                         else synth.make_planes_pair(w, h, int(rng.integers(1 << 30))) if rng.random() < 0.4  # ~60 % of the keypoints match
                         else synth.make_stereo_pair(w, h, int(rng.integers(1 << 30))) for _ in range(B)]
                outs = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
                if big:  # a second batch: the one that runs with the second tier switched on
                    outs = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
                for (imL, imR), out in zip(pairs, outs):
                    oL, oR = ob.Extractor(nf, sfac, nlevels, ini, mn), ob.Extractor(nf, sfac, nlevels, ini, mn)
                    kL, dL, _ = oL.extract(imL)
                    kR, dR, _ = oR.extract(imR)
                    o = ob.stereo_match(oL, oR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
                    err = err or same_keys(out["keysL"], out["descL"], kL, dL) or same_keys(out["keysR"], out["descR"], kR, dR)
                    if err is None and not (out["n"] == o["n"] and np.array_equal(out["uright"], o["uright"]) and
                                            np.array_equal(out["depth"], o["depth"])):
                        err = "stereo match"
                    kps += len(kL) + len(kR)
                fe.close()
        except Exception as e:  # noqa: BLE001
            err = "exception %r" % (e,)
        if err:
            bad += 1
            print("MISMATCH", err, cfg, flush=True)
    print("trials %d keypoints %d mismatches %d  (%.0f s)" % (args.trials, kps, bad, time.time() - t0))
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
