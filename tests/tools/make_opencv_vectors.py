#!/usr/bin/env python3
"""Pinning recipe for the OpenCV-backed arithmetic of the hot path.

The oracle restates cv::resize (INTER_LINEAR, 8UC1), cv::GaussianBlur (7x7, sigma 2, BORDER_REFLECT_101),
cv::FAST (9/16, non-max suppression, with scores) and cv::fastAtan2 from the published OpenCV 4.x algorithms,
because OpenCV is not in the build image.  This script turns that into a data question: run it ONCE on any
machine that has OpenCV >= 4.5.2 with Python bindings (the version ORB-SLAM3 / FastTrack require is >= 4.4,
reference CMakeLists.txt:35; 4.5.2 is where the bit-exact Gaussian kernel generator settled),

    python3 tests/tools/make_opencv_vectors.py            # writes tests/golden/opencv_*.npz

and commit the files it writes.  tests/test_opencv_vectors.py then compares the oracle with OpenCV's own
outputs on every `pytest -m "not gpu"` run (it is skipped, saying so, while the files are absent).  Each file
holds data only: the seeded synthetic input, the parameters, OpenCV's outputs and cv2.__version__.

What is called, exactly as the reference calls it:
  resize       cv2.resize(prev_level, (w, h), 0, 0, cv2.INTER_LINEAR), level l from level l-1    ORBextractor.cc:1508
  blur         cv2.GaussianBlur(level, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)            ORBextractor.cc:1457
  FAST         cv2.FastFeatureDetector_create(t, True, cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)        ORBextractor.cc:1157,1176
               (= cv::FAST(img, kps, t, true)) on whole levels and on 41x43 cell-sized sub-images, t in {20, 7}
  fastAtan2    cv2.fastAtan2(y, x) on integer moment pairs and on a dense float sweep              ORBextractor.cc:65
  cvRound      cv2 has no binding; np.rint (half-to-even) is the documented behaviour - not pinned here.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fasttrack_amd import synth  # noqa: E402  (pure numpy: the same seeded frames the tests use)

G = os.path.join(ROOT, "tests", "golden")
SIZES = [(160, 120, 1), (320, 240, 2), (752, 480, 3)]  # (width, height, seed); the first two are the golden-vector sizes
NLEVELS, SCALE = 8, 1.2


def level_sizes(w, h):
    """ORBextractor.cc:398-414,1499-1500: float32 scale factors, cvRound(dim * 1/sf)"""
    out, sf = [], np.float32(1.0)
    for _ in range(NLEVELS):
        inv = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        sf = np.float32(sf * np.float32(SCALE))
    return out


class Cv2Backend:
    """the four OpenCV entry points, called as the reference calls them"""

    def __init__(self):
        import cv2
        self.cv2 = cv2
        self.version = cv2.__version__

    def resize(self, src, size):
        return self.cv2.resize(src, size, 0, 0, self.cv2.INTER_LINEAR)

    def blur(self, src):
        return self.cv2.GaussianBlur(src, (7, 7), 2, 2, borderType=self.cv2.BORDER_REFLECT_101)

    def fast(self, img, t):
        det = self.cv2.FastFeatureDetector_create(t, True, self.cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
        kps = det.detect(np.ascontiguousarray(img), None)
        return np.array([(int(k.pt[0]), int(k.pt[1]), int(k.response)) for k in kps], np.int32).reshape(-1, 3)

    def fast_atan2(self, y, x):
        return self.cv2.fastAtan2(float(y), float(x))


def generate(out_dir, be, sizes=SIZES):
    """writes opencv_<kind>_<w>x<h>.npz and opencv_fastatan2.npz into out_dir using backend `be`"""
    os.makedirs(out_dir, exist_ok=True)
    for (w, h, seed) in sizes:
        for kind, img in (("scene", synth.make_image(w, h, seed)), ("noise", synth.make_noise(w, h, seed + 100))):
            lsz = level_sizes(w, h)
            levels = [img]
            for l in range(1, NLEVELS):
                levels.append(be.resize(levels[l - 1], lsz[l]))
            data = dict(cv_version=np.array(be.version), image=img, nlevels=NLEVELS, scale=SCALE, sizes=np.array(lsz, np.int32))
            for l in range(1, NLEVELS):
                data[f"level{l}"] = levels[l]
            for l in range(NLEVELS):
                data[f"blur{l}"] = be.blur(levels[l])
            # FAST on whole levels 0, 1 and the last one, both thresholds, NMS on; (x, y, response) in detector order
            for l in (0, 1, NLEVELS - 1):
                for t in (20, 7):
                    data[f"fast_l{l}_t{t}"] = be.fast(levels[l], t)
            # FAST on cell-sized sub-images (the reference calls it per cell: ORBextractor.cc:1157), incl. NMS at the rim
            rng = np.random.default_rng(seed)
            rects = []
            for i in range(24):
                cw, ch = int(rng.integers(7, 48)), int(rng.integers(7, 48))
                x0, y0 = int(rng.integers(0, max(1, w - cw))), int(rng.integers(0, max(1, h - ch)))
                cw, ch = min(cw, w - x0), min(ch, h - y0)
                t = int(rng.choice([20, 7]))
                rects.append((x0, y0, cw, ch, t))
                data[f"cell{i}"] = be.fast(img[y0:y0 + ch, x0:x0 + cw], t)
            data["cell_rects"] = np.array(rects, np.int32)
            np.savez_compressed(os.path.join(out_dir, f"opencv_{kind}_{w}x{h}.npz"), **data)
    # fastAtan2: integer moment pairs as IC_Angle produces them, plus a float sweep over all octants and the axes
    rng = np.random.default_rng(7)
    m = rng.integers(-200000, 200001, (20000, 2)).astype(np.float32)
    a = np.linspace(0, 2 * np.pi, 7201)
    sweep = (np.stack([np.sin(a), np.cos(a)], 1) * 1000).astype(np.float32)
    special = np.array([[0, 0], [0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, 1], [1, -1], [-1, -1], [1e-30, 1], [1, 1e-30]], np.float32)
    yx = np.concatenate([m, sweep, special])
    ang = np.array([be.fast_atan2(y, x) for y, x in yx], np.float32)
    np.savez_compressed(os.path.join(out_dir, "opencv_fastatan2.npz"), cv_version=np.array(be.version), yx=yx, angle=ang)


def main():
    be = Cv2Backend()
    ver = tuple(int(x) for x in be.version.split(".")[:3])
    if ver < (4, 5, 2):
        sys.exit(f"OpenCV {be.version} is older than 4.5.2: the fixed-point Gaussian kernel differs ([18,34,49,55,...])")
    generate(G, be)
    print("wrote tests/golden/opencv_*.npz with OpenCV", be.version)


if __name__ == "__main__":
    main()
