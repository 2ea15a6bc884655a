"""Extraction alone on configs[3]'s images (VERDICT r5 item 5's figure): 128 resident 512 x 512 KB8 pairs, nFeatures 2000, lapping area
(0, 511), both cameras (two extractors on two host threads as Frame's constructor has them), results delivered into pinned host
arrays; ms per step of 128 pairs, one step at a time and two steps in flight (two pairs of extractors).  Runs on the GPU box."""
import concurrent.futures, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fasttrack_amd import orb, synth
ctx = orb.Context(0)
w, h, nf, B = 512, 512, 2000, 128
pairs = [synth.make_planes_pair(w, h, seed=7000 + i) for i in range(B)]
devL, devR = ctx.to_device(np.stack([p[0] for p in pairs])), ctx.to_device(np.stack([p[1] for p in pairs]))
pL = (C.c_void_p * B)(*[devL.ptr.value + b * w * h for b in range(B)])
pR = (C.c_void_p * B)(*[devR.ptr.value + b * w * h for b in range(B)])
pool = concurrent.futures.ThreadPoolExecutor(4)


class Set:
    def __init__(self):
        self.ex = [orb.ORBextractor(ctx, nf, 1.2, 8, 20, 7, w, h, max_batch=B) for _ in range(2)]
        cap = self.ex[0].max_keypoints
        self.k = [ctx.pinned_array((B, cap), orb.KP_DTYPE) for _ in range(2)]
        self.d = [ctx.pinned_array((B, cap, 32), np.uint8) for _ in range(2)]
        self.n = [np.zeros(B, np.int32) for _ in range(4)]

    def step(self):
        a = pool.submit(self.ex[0].extract_batch_into, pL, B, True, w, h, w, (0, 511), self.k[0], self.d[0], self.n[0], self.n[1])
        b = pool.submit(self.ex[1].extract_batch_into, pR, B, True, w, h, w, (0, 511), self.k[1], self.d[1], self.n[2], self.n[3])
        a.result(); b.result()


sets = [Set(), Set()]
for s in sets:
    s.step(); s.step()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    sets[0].step()
one = (time.perf_counter() - t0) / N
two = concurrent.futures.ThreadPoolExecutor(2)
t0 = time.perf_counter()
fs = [two.submit(lambda s=s: [s.step() for _ in range(N)]) for s in sets]
[f.result() for f in fs]
both = (time.perf_counter() - t0) / (2 * N)
print("extraction of 128 512x512 pairs (%.0f keypoints per image): %.3f ms per step one step at a time, %.3f ms per step with two steps in flight"
      % (float(sets[0].n[0].mean()), 1e3 * one, 1e3 * both))
