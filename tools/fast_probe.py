"""What the time of k_fast_cells is made of (runs on the GPU box): the extractor on 256 resident 1280x720 frames (two launches of
128) with the kernel's timing probe FT_DEBUG_FAST = 0 (the real kernel), 2 (no NMS / emission), 3 (also a three-pixel hash instead
of the score network), 4 (staging of the tile only).  Prints the HIP-event time per launch of every extraction kernel.
usage: for d in 0 2 3 4; do FT_DEBUG_FAST=$d python tools/fast_probe.py; done"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fasttrack_amd import orb, synth
ctx = orb.Context(0)
w, h, B = int(os.environ.get("FP_W", "1280")), int(os.environ.get("FP_H", "720")), int(os.environ.get("FP_B", "256"))  # FP_W=512 FP_H=512: configs[3]'s images
mosaic = int(os.environ.get("FP_MOSAIC", "0"))  # dense-corner frames instead (bench.py --mosaic)
base = [synth.make_mosaic_pair(w, h, seed=s, block=mosaic)[0] if mosaic else synth.make_image(w, h, seed=s) for s in range(8)]
arr = np.stack([np.roll(base[b % 8], (29 * (b // 8), 53 * (b // 8)), (0, 1)) for b in range(B)])
dev = ctx.to_device(arr)
ex = orb.ORBextractor(ctx, 2000, 1.2, 8, 20, 7, w, h, max_batch=B)
import ctypes as C
imgs = [dev.ptr.value + b * w * h for b in range(B)]
for _ in range(2): ex.extract_batch(imgs, on_device=True, width=w, height=h, stride=w)
ctx.reset_stats(); ctx.set_kernel_timing(True)
for _ in range(5): ex.extract_batch(imgs, on_device=True, width=w, height=h, stride=w)
out = {}
for k in ("kernel.pyr_down(all levels)", "kernel.fast_cells", "kernel.compact", "kernel.octree", "kernel.orient_desc"):
    ms, n = ctx.get_stat(k); out[k] = round(ms / max(n, 1), 4)
print(os.environ.get("FT_DEBUG_FAST", "0"), "mosaic", mosaic, out)
