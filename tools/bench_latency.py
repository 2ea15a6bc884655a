"""Latency of one frame through the C ABI on one MI355X (the way a SLAM front end calls it: one stereo pair at a time):
ft_extract on one image, and the fused stereo front end on one pair, host images in / host results out.
usage: python tools/bench_latency.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from fasttrack_amd import orb, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ctx = orb.Context(0)
out = {}
for (w, h, nf) in [(752, 480, 1200), (1280, 720, 2000)]:
    intr = synth.intrinsics(w, h)
    L, R = synth.make_stereo_pair(w, h, 5)
    ex = orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h)
    fe = orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, 1, intr["mbf"], intr["mb"])

    def timeit(fn):
        for _ in range(5):
            fn()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(t))
    # the same frames in pinned host memory (ft_host_malloc): the upload kernel reads them where they are
    Lp, Rp = ctx.pinned_array(L.shape, np.uint8), ctx.pinned_array(R.shape, np.uint8)
    Lp[:], Rp[:] = L, R
    out[f"{w}x{h}_nf{nf}"] = {"extract_one_image_ms": timeit(lambda: ex(L)),
                              "stereo_pair_extract_and_match_ms": timeit(lambda: fe.process([L], [R])),
                              "extract_one_image_pinned_frame_ms": timeit(lambda: ex(Lp)),
                              "stereo_pair_pinned_frames_ms": timeit(lambda: fe.process([Lp], [Rp]))}
print(json.dumps(out))
