"""Seeded synthetic camera frames for tests, smoke and bench.

uint8 grayscale, row-major, stride = width: low-frequency background + random filled rectangles and
discs of random intensity + uniform noise of +-3, so FAST fires across all cells and both threshold
tiers (iniThFAST 20 / minThFAST 7) occur.  A stereo pair is the same scene with every object shifted
left by its own disparity in [1, 0.25*fx] px plus independent noise.  Intrinsics follow
Examples/Stereo/EuRoC.yaml of the reference scaled to the image width.
"""
from __future__ import annotations

import numpy as np

EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, width=752, height=480, bf_over_fx=0.110074)


def intrinsics(width: int, height: int) -> dict:
    """EuRoC-like pinhole intrinsics scaled to (width, height); mbf = bf, mb = mbf / fx."""
    sx = width / EUROC["width"]
    sy = height / EUROC["height"]
    fx = np.float32(EUROC["fx"] * sx)
    fy = np.float32(EUROC["fy"] * sy)
    cx = np.float32(EUROC["cx"] * sx)
    cy = np.float32(EUROC["cy"] * sy)
    mbf = np.float32(np.float32(EUROC["bf_over_fx"]) * fx)
    mb = np.float32(mbf / fx)
    return dict(fx=fx, fy=fy, cx=cx, cy=cy, mbf=mbf, mb=mb)


def _background(rng: np.random.Generator, w: int, h: int) -> np.ndarray:
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    bg = np.full((h, w), 110.0, np.float32)
    for _ in range(4):
        fx_, fy_ = rng.uniform(0.5, 3.0, 2) * 2 * np.pi / np.array([w, h])
        ph = rng.uniform(0, 2 * np.pi)
        bg += rng.uniform(8, 25) * np.sin(fx_ * x + fy_ * y + ph)
    return bg


def _objects(rng: np.random.Generator, w: int, h: int, fx: float, density: float):
    n = max(8, int(density * w * h / 1500.0))
    objs = []
    for _ in range(n):
        kind = int(rng.integers(0, 3))  # 0 rect, 1 disc, 2 small speckle rect
        cx = float(rng.uniform(-10, w + 10))
        cy = float(rng.uniform(-10, h + 10))
        if kind == 2:
            sw, sh = rng.integers(2, 7, 2)
        else:
            sw, sh = rng.integers(6, max(8, w // 12), 2)
        val = float(rng.uniform(15, 240))
        disp = float(rng.uniform(1.0, 0.25 * fx))
        objs.append((kind, cx, cy, int(sw), int(sh), val, disp))
    # far objects first so near ones (large disparity) occlude them in both views
    objs.sort(key=lambda o: o[6])
    return objs


def _render(bg: np.ndarray, objs, shift_sign: float) -> np.ndarray:
    h, w = bg.shape
    img = bg.copy()
    for kind, cx, cy, sw, sh, val, disp in objs:
        ox = cx - shift_sign * disp
        x0, x1 = int(round(ox - sw / 2)), int(round(ox + sw / 2))
        y0, y1 = int(round(cy - sh / 2)), int(round(cy + sh / 2))
        xa, xb, ya, yb = max(x0, 0), min(x1, w), max(y0, 0), min(y1, h)
        if xa >= xb or ya >= yb:
            continue
        if kind == 1:
            yy, xx = np.mgrid[ya:yb, xa:xb]
            rx, ry = max(sw / 2.0, 1.0), max(sh / 2.0, 1.0)
            m = ((xx - ox) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
            img[ya:yb, xa:xb][m] = val
        else:
            img[ya:yb, xa:xb] = val
    return img


def _finish(rng: np.random.Generator, img: np.ndarray) -> np.ndarray:
    noise = rng.integers(-3, 4, img.shape).astype(np.float32)
    return np.clip(np.rint(img + noise), 0, 255).astype(np.uint8)


def make_image(width: int, height: int, seed: int, density: float = 1.0) -> np.ndarray:
    """One synthetic frame (height, width) uint8, C-contiguous."""
    rng = np.random.default_rng(seed)
    fx = float(intrinsics(width, height)["fx"])
    bg = _background(rng, width, height)
    objs = _objects(rng, width, height, fx, density)
    return np.ascontiguousarray(_finish(rng, _render(bg, objs, 0.0)))


def make_stereo_pair(width: int, height: int, seed: int, density: float = 1.0):
    """(left, right) synthetic rectified pair; right = objects shifted left by their disparity."""
    rng = np.random.default_rng(seed)
    fx = float(intrinsics(width, height)["fx"])
    bg = _background(rng, width, height)
    objs = _objects(rng, width, height, fx, density)
    left = _finish(rng, _render(bg, objs, 0.0))
    right = _finish(rng, _render(bg, objs, 1.0))
    return np.ascontiguousarray(left), np.ascontiguousarray(right)


def make_mosaic_pair(width: int, height: int, seed: int, block: int = 8, disparity: int = 12):
    """Dense-corner stereo pair: a mosaic of block x block tiles with random grey levels from four well separated values
    (every tile corner where the levels differ is a FAST corner: ~(w / block) * (h / block) of them at level 0 - the
    5-30 k candidates per image SURVEY a4 expects from real frames, where the object scenes give ~3 k) + noise of +-3;
    the right image is the same fronto-parallel plane `disparity` px closer to the left edge."""
    rng = np.random.default_rng(seed)
    levels = np.array([35.0, 95.0, 160.0, 220.0], np.float32)
    gw = (width + disparity) // block + 2
    grid = rng.choice(levels, size=(height // block + 2, gw))
    plane = np.kron(grid, np.ones((block, block), np.float32))
    oy, ox = int(rng.integers(0, block)), int(rng.integers(0, block))
    left = plane[oy:oy + height, ox:ox + width]
    right = plane[oy:oy + height, ox + disparity:ox + disparity + width]
    return np.ascontiguousarray(_finish(rng, left)), np.ascontiguousarray(_finish(rng, right))


def make_planes_pair(width: int, height: int, seed: int, density: float = 1.0, planes: int = 3):
    """Rectified pair with the stereo statistics of an indoor sequence: the object scene of make_image painted on `planes`
    fronto-parallel surfaces (horizontal bands of the image, each at one disparity in [4, 0.12 fx] px) instead of on
    occluding objects at a disparity each.  Inside a band both cameras see the same texture, so most left keypoints have
    their partner in the right image (make_stereo_pair: ~10 %, because its keypoints sit on occlusion edges whose two
    sides move differently) and the SAD / parabola half of Frame::ComputeStereoMatches (Frame.cc:921-987) is loaded as
    on EuRoC-like frames (30-45 % and more)."""
    rng = np.random.default_rng(seed)
    fx = float(intrinsics(width, height)["fx"])
    dmax = max(5, int(0.12 * fx))
    wide = width + dmax + 1
    bg = _background(rng, wide, height)
    objs = _objects(rng, wide, height, fx, density)
    base = _render(bg, objs, 0.0)
    disp = np.sort(rng.integers(4, dmax + 1, planes))[::-1]  # nearer planes (larger disparity) at the bottom, like a floor
    edges = np.linspace(0, height, planes + 1).astype(int)
    left = base[:, :width].copy()
    right = np.empty_like(left)
    for i in range(planes):
        d = int(disp[planes - 1 - i])
        right[edges[i]:edges[i + 1]] = base[edges[i]:edges[i + 1], d:d + width]
    return np.ascontiguousarray(_finish(rng, left)), np.ascontiguousarray(_finish(rng, right))


def make_flat(width: int, height: int, value: int = 128) -> np.ndarray:
    """Featureless frame: exercises the empty-cell / zero-keypoint paths."""
    return np.full((height, width), value, np.uint8)


def make_noise(width: int, height: int, seed: int) -> np.ndarray:
    """Uniform random bytes: worst case for candidate counts (dense FAST responses)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (height, width), dtype=np.uint8)


def make_vocabulary(k: int, L: int, seed: int, flip_bits: int = 40, stop_fraction: float = 0.02, ragged: bool = False):
    """Synthetic DBoW2-style vocabulary tree: node 0 = root, every inner node has k children whose 256-bit centres are
    the parent's with `flip_bits` random bits flipped (so walks are decided by real Hamming contests, ties included);
    leaves carry idf-like weights, a few of them 0 (stopped words).  ragged=True ends some branches one level early.
    Returns dict(k, L, parent, is_leaf, descriptors (n x 32 u8), weights) with nodes in breadth-first order - the
    order ORBvoc.txt lists them in."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight, depth = [0], [0], [np.zeros(32, np.uint8)], [0.0], [0]
    frontier = [0]
    root_centre = rng.integers(0, 256, 32, dtype=np.uint8)
    for level in range(1, L + 1):
        nxt = []
        for p in frontier:
            base = root_centre if p == 0 else desc[p]
            for _ in range(k):
                bits = np.unpackbits(base).copy()
                flip = rng.choice(256, size=max(1, flip_bits // level), replace=False)
                bits[flip] ^= 1
                nid = len(parent)
                parent.append(p); desc.append(np.packbits(bits)); depth.append(level)
                is_leaf = level == L or (ragged and level == L - 1 and rng.random() < 0.15)
                leaf.append(1 if is_leaf else 0)
                weight.append(0.0 if not is_leaf else (0.0 if rng.random() < stop_fraction else float(rng.uniform(0.5, 12.0))))
                if not is_leaf:
                    nxt.append(nid)
        frontier = nxt
    return dict(k=k, L=L, parent=np.array(parent, np.int32), is_leaf=np.array(leaf, np.uint8),
                descriptors=np.stack(desc).astype(np.uint8), weights=np.array(weight, np.float64))


def write_vocabulary_text(voc: dict, path: str, scoring: int = 0, weighting: int = 0) -> None:
    """The text format of ORB-SLAM3's vocabulary (TemplatedVocabulary::saveToTextFile): header "k L  scoring weighting",
    one line per node: parent is_leaf 32 descriptor bytes weight."""
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (voc["k"], voc["L"], scoring, weighting))
        for i in range(1, len(voc["parent"])):
            f.write("%d %d %s %r\n" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["descriptors"][i]),
                                        float(voc["weights"][i])))
