import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fasttrack_amd import orb, synth
ctx = orb.Context(0)
voc = synth.make_vocabulary(10, 6, seed=1)
gv = orb.Vocabulary(ctx, 10, 6, 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
w, h = 752, 480
intr = synth.intrinsics(w, h)
fe = orb.StereoFrontend(ctx, 2000, 1.2, 8, 20, 7, w, h, 1, intr["mbf"], intr["mb"])
L, R = synth.make_stereo_pair(w, h, 3)[:2]
o = fe.process([L], [R])[0]
d, dR = o["descL"], o["descR"]
a, tR = gv.transform(d, 4), gv.transform(dR, 4)
print("node sizes F max", np.diff(a["fv_offsets"]).max(), "K max", np.diff(tR["fv_offsets"]).max())
has = np.ones(len(dR), np.uint8)
gK = orb.BowSide(tR["fv_nodes"], tR["fv_offsets"], tR["fv_features"], dR, o["keysR"]["angle"])
gF = orb.BowSide(a["fv_nodes"], a["fv_offsets"], a["fv_features"], d, o["keysL"]["angle"])
for _ in range(100):
    orb.search_by_bow(ctx, gK, has, gF)
t0 = time.perf_counter()
for _ in range(200):
    orb.search_by_bow(ctx, gK, has, gF)
print("ms per call", (time.perf_counter() - t0) / 200 * 1e3)
