"""The assignment of the wide extractors' streams to the context's lanes (FT_LANE_MAP, fasttrack_amd/csrc/ft_host.h) decides
which kernels may run side by side - never the results: the same batches through different tables (the default one, private
streams, everything on ONE lane, a hand-made one) and different numbers of hardware queues give identical outputs."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import sys, hashlib, numpy as np, ctypes as C
sys.path.insert(0, %r)
from fasttrack_amd import orb, synth
ctx = orb.Context(0)
w, h, nf, B = 640, 480, 1000, 24
intr = synth.intrinsics(w, h)
pairs = [synth.make_stereo_pair(w, h, 900 + b) for b in range(B)]
fes = [orb.StereoFrontend(ctx, nf, 1.2, 8, 20, 7, w, h, 32, intr["mbf"], intr["mb"]) for _ in range(2)]
hsh = hashlib.sha256()
for k in range(4):
    fe = fes[k & 1]
    out = fe.process([p[0] for p in pairs[k::2] + pairs[:k]][:B], [p[1] for p in pairs[k::2] + pairs[:k]][:B])
    for o in out:
        for key in ("keysL", "descL", "keysR", "descR", "uright", "depth"):
            hsh.update(np.ascontiguousarray(o[key]).tobytes())
        hsh.update(str(o["n"]).encode())
print("HASH", hsh.hexdigest())
""" % ROOT


def _run(lane_map, queues):
    env = dict(os.environ)
    env.pop("FT_LANE_MAP", None)
    env.pop("GPU_MAX_HW_QUEUES", None)
    if lane_map is not None:
        env["FT_LANE_MAP"] = lane_map
    if queues is not None:
        env["GPU_MAX_HW_QUEUES"] = str(queues)
    r = subprocess.run([sys.executable, "-c", SNIPPET], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][-1]


def test_results_do_not_depend_on_the_lane_table():
    ref = _run(None, None)
    for lane_map, queues in (("own", None), ("0 0 0 0", None), ("0 1 2 3 3 2 1 0 1 1 1 1", 4), (None, 2)):
        assert _run(lane_map, queues) == ref, (lane_map, queues)
