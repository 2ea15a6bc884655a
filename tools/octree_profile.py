"""Phase times of the device octree kernel (FT_DEBUG_OCT_PROFILE=1): one 16-pair batch of the bench workload."""
import os, sys
os.environ["FT_DEBUG_OCT_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fasttrack_amd import orb, synth
ctx = orb.Context(0)
B = int(os.environ.get("OP_PAIRS", "16"))  # pairs per batch: 16 = a light launch, 256 = the bench's launch shape
w, h = 1280, 720
intr = synth.intrinsics(w, h)
fe = orb.StereoFrontend(ctx, 2000, 1.2, 8, 20, 7, w, h, B, intr["mbf"], intr["mb"])
base = [synth.make_stereo_pair(w, h, seed=100 + i) for i in range(min(B, 16))]
pairs = [base[i % len(base)] for i in range(B)]
out = fe.process([p[0] for p in pairs], [p[1] for p in pairs])
print("keypoints", [len(o["keysL"]) for o in out][:4])
fe.close()
