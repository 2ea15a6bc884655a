"""ctypes binding of oracle/liborb_oracle.so - TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package (fasttrack_amd/).  See oracle/orb_oracle.h for the role of the oracle and the
"parity unpinned" caveat.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborb_oracle.so")


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int), ("class_id", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28 == C.sizeof(KeyPoint)

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int)
f32p = C.POINTER(C.c_float)


class Frame(C.Structure):
    _fields_ = [("N", C.c_int), ("Nleft", C.c_int),
                ("mnMinX", C.c_float), ("mnMinY", C.c_float), ("mnMaxX", C.c_float), ("mnMaxY", C.c_float),
                ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("mbf", C.c_float), ("mb", C.c_float),
                ("keys", C.c_void_p), ("keys_right", C.c_void_p), ("descriptors", C.c_void_p),
                ("uright", C.c_void_p), ("holder_obs", C.c_void_p), ("left_to_right", C.c_void_p),
                ("right_to_left", C.c_void_p), ("cam_model", C.c_int), ("cam", C.c_float * 8),
                ("Trl", C.c_float * 12), ("scale_factors", C.c_void_p), ("nlevels", C.c_int)]


class LocalPoints(C.Structure):
    _fields_ = [("M", C.c_int), ("skip", C.c_void_p), ("in_view", C.c_void_p), ("in_view_r", C.c_void_p),
                ("level", C.c_void_p), ("level_r", C.c_void_p), ("view_cos", C.c_void_p),
                ("view_cos_r", C.c_void_p), ("proj_x", C.c_void_p), ("proj_y", C.c_void_p),
                ("proj_xr", C.c_void_p), ("proj_yr", C.c_void_p), ("descriptors", C.c_void_p),
                ("observations", C.c_void_p)]


class LastPoints(C.Structure):
    _fields_ = [("N", C.c_int), ("valid", C.c_void_p), ("world_pos", C.c_void_p), ("descriptors", C.c_void_p),
                ("observations", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p)]


class FisheyeRig(C.Structure):
    _fields_ = [("cam1", C.c_float * 8), ("cam2", C.c_float * 8), ("precision", C.c_float), ("Rlr", C.c_float * 9),
                ("tlr", C.c_float * 3)]


def make_rig(cam1, cam2, Rlr, tlr, precision=1e-6):
    r = FisheyeRig()
    r.cam1[:] = [float(v) for v in cam1]
    r.cam2[:] = [float(v) for v in cam2]
    r.precision = precision
    r.Rlr[:] = [float(v) for v in np.asarray(Rlr, np.float32).reshape(-1)]
    r.tlr[:] = [float(v) for v in np.asarray(tlr, np.float32).reshape(-1)]
    return r


class FramePose(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("tlr", C.c_float * 3)]


class MapPoints(C.Structure):
    _fields_ = [("M", C.c_int), ("skip", C.c_void_p), ("world_pos", C.c_void_p), ("normal", C.c_void_p),
                ("max_distance", C.c_void_p), ("min_distance", C.c_void_p)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile) if the .so is missing or stale."""
    src = [os.path.join(_HERE, f) for f in ("orb_oracle.cpp", "orb_oracle.h", "orb_pattern.inc", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_extractor_create.restype = C.c_void_p
        L.orc_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_extractor_destroy.argtypes = [C.c_void_p]
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, i32p]
        L.orc_compute_pyramid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        for f in (L.orc_get_level, L.orc_get_blurred):
            f.argtypes = [C.c_void_p, C.c_int, C.POINTER(u8p), i32p, i32p, i32p]
        L.orc_get_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_get_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.orc_brief_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_cv_round_f.argtypes = [C.c_float]
        L.orc_cv_round_d.argtypes = [C.c_double]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_gaussian_blur7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_gaussian_kernel7_fixed.argtypes = [C.c_void_p]
        L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_is_corner.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_fast_corner_score.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_fast_mask_has_arc9.argtypes = [C.c_uint]
        L.orc_scale_factors.argtypes = [C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_features_per_level.argtypes = [C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_umax.argtypes = [C.c_void_p]
        L.orc_level_sizes.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_pattern.restype = C.POINTER(C.c_int8)
        L.orc_distribute_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int]
        L.orc_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int]
        L.orc_fisheye_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_features_in_area.argtypes = [C.POINTER(Frame), C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_int]
        L.orc_search_local_points.argtypes = [C.POINTER(Frame), C.POINTER(LocalPoints), C.c_float, C.c_float] + [C.c_void_p] * 11
        L.orc_search_last_frame.argtypes = [C.POINTER(Frame), C.POINTER(LastPoints), C.c_void_p, C.c_float,
                                            C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.orc_three_maxima.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------------
# thin numpy-level helpers
# ---------------------------------------------------------------------------------------------
def scale_factors(sf: float, nlevels: int):
    a = np.zeros(nlevels, np.float32)
    b = np.zeros(nlevels, np.float32)
    lib().orc_scale_factors(sf, nlevels, _p(a), _p(b))
    return a, b


def features_per_level(nfeatures: int, sf: float, nlevels: int):
    a = np.zeros(nlevels, np.int32)
    lib().orc_features_per_level(nfeatures, sf, nlevels, _p(a))
    return a


def umax():
    a = np.zeros(16, np.int32)
    lib().orc_umax(_p(a))
    return a


def level_sizes(w, h, sf, nlevels):
    a = np.zeros(nlevels, np.int32)
    b = np.zeros(nlevels, np.int32)
    lib().orc_level_sizes(w, h, sf, nlevels, _p(a), _p(b))
    return a, b


def pattern():
    return np.ctypeslib.as_array(lib().orc_pattern(), shape=(1024,)).copy()


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gaussian_blur7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast_atan2(y: float, x: float) -> float:
    """cv::fastAtan2 (degrees)"""
    return float(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def gaussian_kernel7():
    a = np.zeros(7, np.int32)
    lib().orc_gaussian_kernel7_fixed(_p(a))
    return a


def fast9_16(img: np.ndarray, threshold: int, nonmax: bool = True) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast9_16(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nonmax), _p(out), cap)
    return out[:n].copy()


def distribute_octree(xys: np.ndarray, minX, maxX, minY, maxY, N) -> np.ndarray:
    xys = np.ascontiguousarray(xys, np.int32)
    n = xys.shape[0]
    out = np.zeros(max(n, 1), np.int32)
    k = lib().orc_distribute_octree(_p(xys), n, minX, maxX, minY, maxY, N, _p(out), n)
    return out[:k].copy()


def descriptor_distance(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


class Extractor:
    """Mirror of ORB_SLAM3::ORBextractor (CPU branch) over the oracle."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.nfeatures, self.scale_factor, self.nlevels = nfeatures, scale_factor, nlevels
        self.ini_th, self.min_th = ini_th, min_th
        self._h = lib().orc_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_extractor_destroy(self._h)
            self._h = None

    def extract(self, img: np.ndarray, lap=(0, 0)):
        """returns (keypoints[KP_DTYPE], descriptors[n,32], n_mono)"""
        img = np.ascontiguousarray(img, np.uint8)
        # DistributeOctTree overshoots its per-level quota: a breadth-first pass splits every node of the list before
        # the size is checked (ORBextractor.cc:719-797), so a level may return up to four times its quota (or its
        # initial nodes); orc_extract never writes past `cap` and returns the full count
        cap = 4 * self.nfeatures + 128 * self.nlevels
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        nm = C.c_int(0)
        h, w = (img.shape if img.size else (0, 0))
        n = lib().orc_extract(self._h, _p(img) if img.size else None, w, h, img.strides[0] if img.size else 0,
                              lap[0], lap[1], _p(kps), _p(desc), cap, C.byref(nm))
        if n < 0:
            return None, None, -1
        assert n <= cap
        return kps[:n].copy(), desc[:n].copy(), nm.value

    def pyramid(self, img: np.ndarray):
        img = np.ascontiguousarray(img, np.uint8)
        lib().orc_compute_pyramid(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0])
        return [self.level(l) for l in range(self.nlevels)]

    def _img(self, fn, level):
        d = u8p()
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        if fn(self._h, level, C.byref(d), C.byref(w), C.byref(h), C.byref(s)) != 0:
            return None
        return np.ctypeslib.as_array(d, shape=(h.value, s.value))[:, :w.value].copy()

    def level(self, level):
        return self._img(lib().orc_get_level, level)

    def blurred(self, level):
        return self._img(lib().orc_get_blurred, level)

    def candidates(self, level) -> np.ndarray:
        n = lib().orc_get_candidates(self._h, level, None, 0)
        out = np.zeros((max(n, 1), 3), np.int32)
        lib().orc_get_candidates(self._h, level, _p(out), n)
        return out[:n]

    def level_keypoints(self, level):
        n = lib().orc_get_level_keypoints(self._h, level, None, None, 0)
        kps = np.zeros(max(n, 1), KP_DTYPE)
        desc = np.zeros((max(n, 1), 32), np.uint8)
        lib().orc_get_level_keypoints(self._h, level, _p(kps), _p(desc), n)
        return kps[:n], desc[:n]


def stereo_match(exL: Extractor, exR: Extractor, keysL, keysR, descL, descR, mbf, mb, median_cut=True):
    """Frame::ComputeStereoMatches.  returns dict(uright, depth, sad, hamming_idx, n)"""
    nL, nR = len(keysL), len(keysR)
    keysL = np.ascontiguousarray(keysL)
    keysR = np.ascontiguousarray(keysR)
    descL = np.ascontiguousarray(descL, np.uint8)
    descR = np.ascontiguousarray(descR, np.uint8)
    ur = np.zeros(max(nL, 1), np.float32)
    dp = np.zeros(max(nL, 1), np.float32)
    sad = np.zeros(max(nL, 1), np.int32)
    hi = np.zeros(max(nL, 1), np.int32)
    n = lib().orc_stereo_match(exL._h, exR._h, _p(keysL), nL, _p(keysR), nR, _p(descL), _p(descR),
                               float(mbf), float(mb), _p(ur), _p(dp), _p(sad), _p(hi), int(median_cut))
    return dict(uright=ur[:nL], depth=dp[:nL], sad=sad[:nL], hamming_idx=hi[:nL], n=n)


def fisheye_match(descL, descR):
    descL = np.ascontiguousarray(descL, np.uint8)
    descR = np.ascontiguousarray(descR, np.uint8)
    nL, nR = len(descL), len(descR)
    m = np.zeros(max(nL, 1), np.int32)
    b = np.zeros(max(nL, 1), np.int32)
    s = np.zeros(max(nL, 1), np.int32)
    n = lib().orc_fisheye_match(_p(descL), nL, _p(descR), nR, _p(m), _p(b), _p(s))
    return dict(matches=m[:nL], best=b[:nL], second=s[:nL], n=n)


class FrameView:
    """Owns the numpy arrays behind an orc_frame."""

    def __init__(self, keys, descriptors, scale_factors_, bounds, mbf=0.0, mb=0.0, uright=None,
                 holder_obs=None, keys_right=None, left_to_right=None, right_to_left=None, cam_model=0,
                 cam=None, Trl=None):
        self.keys = np.ascontiguousarray(keys)
        self.keys_right = None if keys_right is None else np.ascontiguousarray(keys_right)
        nleft = -1 if keys_right is None else len(self.keys)
        n = len(self.keys) + (0 if keys_right is None else len(self.keys_right))
        self.descriptors = np.ascontiguousarray(descriptors, np.uint8)
        assert self.descriptors.shape == (n, 32)
        self.sf = np.ascontiguousarray(scale_factors_, np.float32)
        self.uright = None if uright is None else np.ascontiguousarray(uright, np.float32)
        self.holder_obs = (np.full(n, -1, np.int32) if holder_obs is None
                           else np.ascontiguousarray(holder_obs, np.int32).copy())
        self.l2r = None if left_to_right is None else np.ascontiguousarray(left_to_right, np.int32)
        self.r2l = None if right_to_left is None else np.ascontiguousarray(right_to_left, np.int32)
        minx, miny, maxx, maxy = [np.float32(v) for v in bounds]
        f = Frame()
        f.N, f.Nleft = n, nleft
        f.mnMinX, f.mnMinY, f.mnMaxX, f.mnMaxY = minx, miny, maxx, maxy
        f.grid_inv_w = np.float32(64) / np.float32(maxx - minx)
        f.grid_inv_h = np.float32(48) / np.float32(maxy - miny)
        f.mbf, f.mb = float(mbf), float(mb)
        f.keys = _p(self.keys)
        f.keys_right = _p(self.keys_right)
        f.descriptors = _p(self.descriptors)
        f.uright = _p(self.uright)
        f.holder_obs = _p(self.holder_obs)
        f.left_to_right = _p(self.l2r)
        f.right_to_left = _p(self.r2l)
        f.cam_model = cam_model
        cam = np.zeros(8, np.float32) if cam is None else np.asarray(cam, np.float32)
        for i in range(8):
            f.cam[i] = float(cam[i]) if i < len(cam) else 0.0
        Trl = np.eye(3, 4, dtype=np.float32) if Trl is None else np.asarray(Trl, np.float32).reshape(3, 4)
        for i in range(12):
            f.Trl[i] = float(Trl.flat[i])
        f.scale_factors = _p(self.sf)
        f.nlevels = len(self.sf)
        self.c = f
        self.N, self.Nleft = n, nleft


def features_in_area(F: FrameView, x, y, r, min_level=-1, max_level=-1, right=False):
    out = np.zeros(max(F.N, 1), np.int32)
    n = lib().orc_features_in_area(C.byref(F.c), x, y, r, min_level, max_level, int(right), _p(out), F.N)
    return out[:n].copy()


def search_local_points(F: FrameView, pts: dict, th: float, nn_ratio: float = 0.8):
    """pts: dict of arrays (skip,in_view,in_view_r,level,level_r,view_cos,view_cos_r,proj_x,proj_y,proj_xr,
    proj_yr,descriptors,observations).  Mutates F.holder_obs like the reference mutates mvpMapPoints."""
    M = len(pts["skip"])
    keep = {}

    def arr(k, dt):
        keep[k] = np.ascontiguousarray(pts[k], dt)
        return _p(keep[k])

    P = LocalPoints()
    P.M = M
    P.skip, P.in_view, P.in_view_r = arr("skip", np.uint8), arr("in_view", np.uint8), arr("in_view_r", np.uint8)
    P.level, P.level_r = arr("level", np.int32), arr("level_r", np.int32)
    P.view_cos, P.view_cos_r = arr("view_cos", np.float32), arr("view_cos_r", np.float32)
    P.proj_x, P.proj_y = arr("proj_x", np.float32), arr("proj_y", np.float32)
    P.proj_xr, P.proj_yr = arr("proj_xr", np.float32), arr("proj_yr", np.float32)
    P.descriptors, P.observations = arr("descriptors", np.uint8), arr("observations", np.int32)
    assign = np.zeros(max(F.N, 1), np.int32)
    outs = [np.zeros(max(M, 1), np.int32) for _ in range(10)]
    n = lib().orc_search_local_points(C.byref(F.c), C.byref(P), th, nn_ratio, _p(assign), *[_p(o) for o in outs])
    names = ["best_dist", "best_dist2", "best_level", "best_level2", "best_idx",
             "best_dist_r", "best_dist2_r", "best_level_r", "best_level2_r", "best_idx_r"]
    r = {k: o[:M] for k, o in zip(names, outs)}
    r["assign"] = assign[:F.N]
    r["n"] = n
    return r


class SE3:
    """A rigid transform as Sophus::SE3f holds it: unit quaternion (x, y, z, w) and translation, float32"""

    def __init__(self, q, t):
        self.q = np.ascontiguousarray(q, np.float32).reshape(4)
        self.t = np.ascontiguousarray(t, np.float32).reshape(3)

    def apply(self, p):
        """Sophus::SE3f * point through the oracle (orc_se3_transform)"""
        p = np.ascontiguousarray(p, np.float32).reshape(3)
        y = np.zeros(3, np.float32)
        lib().orc_se3_transform.restype = None
        lib().orc_se3_transform.argtypes = [C.c_void_p] * 4
        lib().orc_se3_transform(_p(self.q), _p(self.t), _p(p), _p(y))
        return y

    def matrix(self):
        """the 3x4 float32 matrix Eigen::Quaternionf::toRotationMatrix gives (for callers of the matrix form)"""
        x, y, z, w = [np.float32(v) for v in self.q]
        tx, ty, tz = np.float32(2) * x, np.float32(2) * y, np.float32(2) * z
        twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
        one = np.float32(1)
        R = np.array([[one - (tyy + tzz), txy - twz, txz + twy], [txy + twz, one - (txx + tzz), tyz - twx],
                      [txz - twy, tyz + twx, one - (txx + tyy)]], np.float32)
        return np.concatenate([R, self.t.reshape(3, 1)], 1).astype(np.float32)


def search_last_frame(Cur: FrameView, last: dict, Tcw, th, forward=False, backward=False, check_orientation=True, Trl=None):
    """last: dict(valid, world_pos[N,3], descriptors, observations, octave, angle).  Tcw: a 3x4 matrix (y = R x + t, the
    reference's GPU boundary) or an SE3 (quaternion form, what the CPU branch evaluates; then Trl - an SE3 - replaces Cur's
    matrix for the right camera)."""
    N = len(last["valid"])
    keep = {}

    def arr(k, dt):
        keep[k] = np.ascontiguousarray(last[k], dt)
        return _p(keep[k])

    Lp = LastPoints()
    Lp.N = N
    Lp.valid, Lp.world_pos = arr("valid", np.uint8), arr("world_pos", np.float32)
    Lp.descriptors, Lp.observations = arr("descriptors", np.uint8), arr("observations", np.int32)
    Lp.octave, Lp.angle = arr("octave", np.int32), arr("angle", np.float32)
    assign = np.zeros(max(Cur.N, 1), np.int32)
    outs = [np.zeros(max(N, 1), np.int32) for _ in range(4)]
    if isinstance(Tcw, SE3):  # the Sophus form of the CPU branch; Trl likewise
        lib().orc_search_last_frame_se3.restype = C.c_int
        lib().orc_search_last_frame_se3.argtypes = [C.POINTER(Frame), C.POINTER(LastPoints)] + [C.c_void_p] * 4 + \
            [C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        n = lib().orc_search_last_frame_se3(C.byref(Cur.c), C.byref(Lp), _p(Tcw.q), _p(Tcw.t), None if Trl is None else _p(Trl.q),
                                            None if Trl is None else _p(Trl.t), th, int(forward), int(backward),
                                            int(check_orientation), _p(assign), *[_p(o) for o in outs])
    else:
        T = np.ascontiguousarray(np.asarray(Tcw, np.float32).reshape(3, 4))
        n = lib().orc_search_last_frame(C.byref(Cur.c), C.byref(Lp), _p(T), th, int(forward), int(backward),
                                        int(check_orientation), _p(assign), *[_p(o) for o in outs])
    names = ["best_dist", "best_idx", "best_dist_r", "best_idx_r"]
    r = {k: o[:N] for k, o in zip(names, outs)}
    r["assign"] = assign[:Cur.N]
    r["n"] = n
    return r


FRUSTUM_FIELDS = [("in_view", np.uint8), ("in_view_r", np.uint8), ("level", np.int32), ("level_r", np.int32),
                  ("view_cos", np.float32), ("view_cos_r", np.float32), ("proj_x", np.float32), ("proj_y", np.float32),
                  ("proj_xr", np.float32), ("proj_yr", np.float32), ("depth", np.float32), ("depth_r", np.float32)]


def make_pose(Rcw, tcw, tlr=(0, 0, 0)):
    """-> FramePose with mOw = -Rcw^T tcw evaluated in float32 like Frame::UpdatePoseMatrices (Frame.cc:487-497)"""
    Rcw = np.asarray(Rcw, np.float32).reshape(3, 3)
    tcw = np.asarray(tcw, np.float32).reshape(3)
    Ow = (-(Rcw.T.astype(np.float32) @ tcw)).astype(np.float32)
    T = FramePose()
    T.Rcw[:] = [float(v) for v in Rcw.reshape(-1)]
    T.tcw[:] = [float(v) for v in tcw]
    T.Ow[:] = [float(v) for v in Ow]
    T.tlr[:] = [float(v) for v in np.asarray(tlr, np.float32)]
    return T


def is_in_frustum(F: FrameView, pose: FramePose, pts: dict, viewing_cos_limit: float, log_scale_factor: float):
    """pts: dict(world_pos[M,3], normal[M,3], max_distance[M], min_distance[M], skip[M] optional)."""
    M = len(pts["world_pos"])
    keep = {}

    def arr(k, dt):
        keep[k] = np.ascontiguousarray(pts[k], dt)
        return _p(keep[k])

    P = MapPoints()
    P.M = M
    P.skip = arr("skip", np.uint8) if pts.get("skip") is not None else None
    P.world_pos, P.normal = arr("world_pos", np.float32), arr("normal", np.float32)
    P.max_distance, P.min_distance = arr("max_distance", np.float32), arr("min_distance", np.float32)
    outs = [np.zeros(max(M, 1), dt) for _, dt in FRUSTUM_FIELDS]
    lib().orc_is_in_frustum.restype = C.c_int
    lib().orc_is_in_frustum.argtypes = [C.POINTER(Frame), C.POINTER(FramePose), C.POINTER(MapPoints), C.c_float,
                                        C.c_float] + [C.c_void_p] * 12
    n = lib().orc_is_in_frustum(C.byref(F.c), C.byref(pose), C.byref(P), viewing_cos_limit, log_scale_factor,
                                *[_p(o) for o in outs])
    r = {k: o[:M] for (k, _), o in zip(FRUSTUM_FIELDS, outs)}
    r["n"] = n
    return r


def kb8_triangulate(rig: FisheyeRig, xy1, xy2, sigma1, sigma2):
    """KannalaBrandt8::TriangulateMatches per pair -> (code[n], p3d[n,3])"""
    xy1 = np.ascontiguousarray(xy1, np.float32); xy2 = np.ascontiguousarray(xy2, np.float32)
    s1 = np.ascontiguousarray(sigma1, np.float32); s2 = np.ascontiguousarray(sigma2, np.float32)
    n = len(xy1)
    code = np.zeros(max(n, 1), np.float32); p3d = np.zeros((max(n, 1), 3), np.float32)
    lib().orc_kb8_triangulate.restype = None
    lib().orc_kb8_triangulate.argtypes = [C.POINTER(FisheyeRig), C.c_int] + [C.c_void_p] * 6
    lib().orc_kb8_triangulate(C.byref(rig), n, _p(xy1), _p(xy2), _p(s1), _p(s2), _p(code), _p(p3d))
    return code[:n], p3d[:n]


def fisheye_stereo(rig: FisheyeRig, descL, keysL, descR, keysR, level_sigma2):
    """Frame::ComputeStereoFishEyeMatches on the lapping-area subsets -> dict(matches, depth, p3d, n)"""
    descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
    keysL = np.ascontiguousarray(keysL); keysR = np.ascontiguousarray(keysR)
    ls2 = np.ascontiguousarray(level_sigma2, np.float32)
    nL, nR = len(descL), len(descR)
    m = np.full(max(nL, 1), -1, np.int32); d = np.zeros(max(nL, 1), np.float32); p = np.zeros((max(nL, 1), 3), np.float32)
    lib().orc_fisheye_stereo.restype = C.c_int
    lib().orc_fisheye_stereo.argtypes = [C.POINTER(FisheyeRig), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n = lib().orc_fisheye_stereo(C.byref(rig), _p(descL), _p(keysL), nL, _p(descR), _p(keysR), nR, _p(ls2), _p(m), _p(d), _p(p))
    return dict(matches=m[:nL], depth=d[:nL], p3d=p[:nL], n=n)


class Vocabulary:
    """DBoW2 ORBVocabulary (oracle): build from arrays (node 0 = root) or from the ORBvoc.txt text format."""

    def __init__(self, k=None, L=None, scoring=0, weighting=0, parent=None, is_leaf=None, descriptors=None, weights=None,
                 path=None):
        L_ = lib()
        L_.orc_vocabulary_create.restype = C.c_void_p
        L_.orc_vocabulary_create.argtypes = [C.c_int] * 5 + [C.c_void_p] * 4
        L_.orc_vocabulary_load_text.restype = C.c_void_p
        L_.orc_vocabulary_load_text.argtypes = [C.c_char_p]
        L_.orc_vocabulary_destroy.argtypes = [C.c_void_p]
        L_.orc_vocabulary_nodes.argtypes = [C.c_void_p]
        L_.orc_vocabulary_words.argtypes = [C.c_void_p]
        L_.orc_bow_transform.restype = None
        L_.orc_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 10
        if path is not None:
            self._h = L_.orc_vocabulary_load_text(str(path).encode())
            if not self._h:
                raise ValueError("not a vocabulary text file: %s" % path)
        else:
            parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            descriptors = np.ascontiguousarray(descriptors, np.uint8); weights = np.ascontiguousarray(weights, np.float64)
            self._h = L_.orc_vocabulary_create(k, L, scoring, weighting, len(parent), _p(parent), _p(is_leaf),
                                               _p(descriptors), _p(weights))
        self.n_nodes = L_.orc_vocabulary_nodes(self._h)
        self.n_words = L_.orc_vocabulary_words(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_vocabulary_destroy(self._h)
            self._h = None

    def transform(self, descriptors, levelsup=4):
        """-> dict(word, node, weight per feature; bow_ids, bow_values; fv_nodes, fv_offsets, fv_features)"""
        d = np.ascontiguousarray(descriptors, np.uint8)
        n = len(d)
        m = max(n, 1)
        word = np.zeros(m, np.uint32); node = np.zeros(m, np.uint32); w = np.zeros(m, np.float64)
        bi = np.zeros(m, np.uint32); bv = np.zeros(m, np.float64); nb = C.c_int(0)
        fn = np.zeros(m, np.uint32); fo = np.zeros(m + 1, np.int32); ff = np.zeros(m, np.uint32); nf = C.c_int(0)
        lib().orc_bow_transform(self._h, _p(d), n, levelsup, _p(word), _p(node), _p(w), _p(bi), _p(bv),
                                C.addressof(nb), _p(fn), _p(fo), _p(ff), C.addressof(nf))
        return dict(word=word[:n], node=node[:n], weight=w[:n], bow_ids=bi[:nb.value], bow_values=bv[:nb.value],
                    fv_nodes=fn[:nf.value], fv_offsets=fo[:nf.value + 1], fv_features=ff[:fo[nf.value]])


class BowSide(C.Structure):
    """orc_bow_side (same layout as the product's ft_bow_side)"""
    _fields_ = [("n", C.c_int), ("n_nodes", C.c_int), ("fv_nodes", C.c_void_p), ("fv_offsets", C.c_void_p),
                ("fv_features", C.c_void_p), ("descriptors", C.c_void_p), ("angles", C.c_void_p)]


def search_by_bow(kf, kf_has_point, frame, frame_nleft=-1, nn_ratio=0.7, check_orientation=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (orc_search_by_bow).  kf / frame: dicts with fv_nodes, fv_offsets,
    fv_features (Vocabulary.transform), descriptors, angles -> dict(matches, n)"""
    keep = []

    def side(d):
        s = BowSide()
        arrs = [np.ascontiguousarray(d["fv_nodes"], np.uint32), np.ascontiguousarray(d["fv_offsets"], np.int32),
                np.ascontiguousarray(d["fv_features"], np.uint32), np.ascontiguousarray(d["descriptors"], np.uint8),
                np.ascontiguousarray(d["angles"], np.float32)]
        if len(arrs[1]) == 0:
            arrs[1] = np.zeros(1, np.int32)
        keep.extend(arrs)
        s.n, s.n_nodes = len(arrs[3]), len(arrs[0])
        s.fv_nodes, s.fv_offsets, s.fv_features, s.descriptors, s.angles = [_p(a) for a in arrs]
        return s
    K, F = side(kf), side(frame)
    has = np.ascontiguousarray(kf_has_point, np.uint8)
    m = np.full(max(F.n, 1), -1, np.int32)
    lib().orc_search_by_bow.restype = C.c_int
    lib().orc_search_by_bow.argtypes = [C.POINTER(BowSide), C.c_void_p, C.POINTER(BowSide), C.c_int, C.c_float, C.c_int,
                                        C.c_void_p]
    n = lib().orc_search_by_bow(C.byref(K), _p(has), C.byref(F), int(frame_nleft), float(nn_ratio), int(bool(check_orientation)),
                                _p(m))
    return dict(matches=m[:F.n], n=n)

