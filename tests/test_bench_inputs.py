"""The synthetic inputs of bench.py (CPU): every pair of a step is a distinct frame, shifted scenes stay rectified, the dense
mosaic scenes carry the candidate counts the second octree tier is for."""
import importlib.util
import os

import numpy as np

from fasttrack_amd import synth
from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod_inputs", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_stream_frames_are_distinct_and_rectified():
    bench = _bench()
    w, h, D, S = 160, 120, 12, 3
    hostL, hostR, base = bench.make_stream(lambda shape, dt: np.zeros(shape, dt), w, h, rank=1, D=D, scenes=S)
    assert hostL.shape == (D, h, w) and len(base) == S
    assert len({hostL[d].tobytes() for d in range(D)}) == D and len({hostR[d].tobytes() for d in range(D)}) == D
    for d in range(D):
        s, k = d % S, d // S
        dx, dy = (53 * k) % w, (29 * k) % h
        # the same cyclic shift for both images of a pair: rows stay aligned, disparities are kept
        assert np.array_equal(np.roll(hostL[d], (-dy, -dx), (0, 1)), base[s][0])
        assert np.array_equal(np.roll(hostR[d], (-dy, -dx), (0, 1)), base[s][1])
    other, _, _ = bench.make_stream(lambda shape, dt: np.zeros(shape, dt), w, h, rank=2, D=2, scenes=2)
    assert not np.array_equal(other[0], hostL[0])  # one stream per rank: different seeds


def test_mosaic_pair_is_dense_seeded_and_rectified():
    w, h = 320, 240
    L, R = synth.make_mosaic_pair(w, h, seed=4, block=8, disparity=12)
    L2, R2 = synth.make_mosaic_pair(w, h, seed=4, block=8, disparity=12)
    assert np.array_equal(L, L2) and np.array_equal(R, R2) and L.shape == (h, w) and L.dtype == np.uint8
    # right image = the same plane 12 px further left (up to the +-3 noise of each view)
    diff = np.abs(L[:, 12:].astype(int) - R[:, :-12].astype(int))
    assert diff.max() <= 6
    obj = synth.make_image(w, h, seed=4)
    exd, exo = ob.Extractor(500), ob.Extractor(500)
    exd.extract(L)
    exo.extract(obj)
    assert len(exd.candidates(0)) > 3 * len(exo.candidates(0))  # several times the FAST survivors of the object scenes
