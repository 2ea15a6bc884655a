"""Python driver over the C ABI, mirroring the reference's operator interface for the hot path:

  ORBextractor            <- ORB_SLAM3::ORBextractor        (reference include/ORBextractor.h:100-197)
  StereoFrontend          <- Frame::Frame(stereo) extraction threads + ComputeStereoMatches (src/Frame.cc:102-230)
  KernelController.*      <- KernelController::launch*      (reference include/Kernels/KernelController.h:31-46)

Python is only the test / bench driver here: every number comes out of the HIP kernels behind
libfasttrack_amd.so.  The C++ mirror for ORB-SLAM3 builds is include/fasttrack_amd.hpp.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import KP_DTYPE, FastTrackError, check, lib, ptr

__all__ = ["Context", "ORBextractor", "StereoFrontend", "KernelController", "FrameView", "KP_DTYPE", "FastTrackError"]


class Context:
    """One per GPU: KernelController::setCUDADevice + initializeKernels (KernelController.h:15-25)."""

    def __init__(self, device: int = 0, host_threads: int = 0):
        self._h = C.c_void_p()
        check(lib().ft_context_create(device, host_threads, C.byref(self._h)))
        self.device = device

    def close(self):
        """Destroys the context.  ft_context_destroy refuses (FT_ERR_INVALID) while extractors, front ends or tracked frames
        of the context are alive: then NOTHING is released here - the handle and the pinned arrays those objects may still
        write into stay valid - and the error is raised, so an early close() cannot leak the context silently."""
        if getattr(self, "_h", None):
            check(lib().ft_context_destroy(self._h))  # destroys only when nothing lives on the context any more ...
            self._h = None
            self._pinned = []  # ... and takes the pinned allocations of ft_host_malloc with it (context.cpp)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tuning options (ft_context_set_option; the table is FT_TUNING_OPTIONS in csrc/ft_host.h) ----
    def set_option(self, name: str, value: int):
        check(lib().ft_context_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int()
        check(lib().ft_context_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """context manager: `with ctx.options(device_octree=0): ex = ORBextractor(ctx, ...)` - objects created inside take the
        switches, the previous values come back afterwards"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = {k: self.get_option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return cm()

    @staticmethod
    def option_table():
        """[(name, environment variable, default, description)] of every tuning option"""
        out = []
        i = 0
        while True:
            n, e, d, doc = C.c_char_p(), C.c_char_p(), C.c_int(), C.c_char_p()
            if lib().ft_option_describe(i, C.byref(n), C.byref(e), C.byref(d), C.byref(doc)) != 0:
                return out
            out.append((n.value.decode(), e.value.decode(), d.value, doc.value.decode()))
            i += 1

    @staticmethod
    def option_range(name: str):
        lo, hi = C.c_int(), C.c_int()
        check(lib().ft_option_range(name.encode(), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def synchronize(self):
        check(lib().ft_context_synchronize(self._h))

    @property
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        check(lib().ft_context_device_name(self._h, buf, 256))
        return buf.value.decode()

    @property
    def host_threads(self) -> int:
        return lib().ft_context_host_threads(self._h)

    @property
    def hw_queues(self) -> int:
        """GPU_MAX_HW_QUEUES the context was created under (4 = unset): which lane table the wide extractors use"""
        return lib().ft_context_hw_queues(self._h)

    def set_kernel_timing(self, enabled: bool):
        check(lib().ft_context_set_kernel_timing(self._h, int(enabled)))

    def get_stat(self, name: str):
        """-> (total_ms, calls) of a stat / kernel-timing series"""
        ms, n = C.c_double(), C.c_long()
        check(lib().ft_context_get_stat(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_stats(self):
        check(lib().ft_context_reset_stats(self._h))

    def save_stats(self, path: str):
        check(lib().ft_context_save_stats(self._h, path.encode()))

    def pinned_array(self, shape, dtype) -> np.ndarray:
        """numpy array in pinned host memory (ft_host_malloc): device copies land in it directly.
        The memory lives as long as the context."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        check(lib().ft_host_malloc(self._h, max(n, 1), C.byref(p)))
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        arr[...] = np.zeros((), dtype)
        self._pinned = getattr(self, "_pinned", []) + [p]
        return arr

    # frames resident in HBM
    def to_device(self, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        d = C.c_void_p()
        check(lib().ft_device_malloc(self._h, arr.nbytes, C.byref(d)))
        check(lib().ft_memcpy_h2d(self._h, d, ptr(arr), arr.nbytes))
        return DeviceBuffer(self, d, arr.nbytes)


class DeviceBuffer:
    def __init__(self, ctx: Context, dptr, nbytes: int):
        self.ctx, self.ptr, self.nbytes = ctx, dptr, nbytes

    def free(self):
        if self.ptr:
            lib().ft_device_free(self.ctx._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self.ctx._h:
                self.free()
        except Exception:
            pass


def version() -> str:
    """ft_version(): library version and the hash of the csrc sources it was built from ('... csrc:<hash>')"""
    return lib().ft_version().decode()


def _image_ptrs(images, on_device):
    n = len(images)
    arr = (C.c_void_p * n)()
    keep = []
    for b, im in enumerate(images):
        if on_device:
            arr[b] = im.ptr if isinstance(im, DeviceBuffer) else im
        else:
            a = np.ascontiguousarray(im, np.uint8)
            keep.append(a)
            arr[b] = a.ctypes.data
    return arr, keep


class ORBextractor:
    """ORB_SLAM3::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, imageWidth, imageHeight)."""

    def __init__(self, ctx: Context, nfeatures: int, scale_factor: float, nlevels: int, ini_th_fast: int,
                 min_th_fast: int, image_width: int, image_height: int, max_batch: int = 1, _handle=None):
        self.ctx = ctx
        self.width, self.height = image_width, image_height
        self._owned = _handle is None
        if _handle is None:
            self._h = C.c_void_p()
            check(lib().ft_extractor_create(ctx._h, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast,
                                            image_width, image_height, max_batch, C.byref(self._h)))
        else:
            self._h = C.c_void_p(_handle)
        self.nlevels = lib().ft_extractor_levels(self._h)
        self.max_batch = lib().ft_extractor_max_batch(self._h)
        self.max_keypoints = lib().ft_extractor_max_keypoints(self._h)

    def close(self):
        if getattr(self, "_h", None) and self._owned and self.ctx._h:
            lib().ft_extractor_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # getters of include/ORBextractor.h:117-144
    def GetLevels(self):
        return self.nlevels

    def _sf(self, which):
        out = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        check(lib().ft_extractor_scale_factors(self._h, *[ptr(o) for o in out]))
        return out[which]

    def GetScaleFactors(self):
        return self._sf(0)

    def GetInverseScaleFactors(self):
        return self._sf(1)

    def GetScaleSigmaSquares(self):
        return self._sf(2)

    def GetInverseScaleSigmaSquares(self):
        return self._sf(3)

    def features_per_level(self):
        q = np.zeros(self.nlevels, np.int32)
        check(lib().ft_extractor_features_per_level(self._h, ptr(q)))
        return q

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        check(lib().ft_extractor_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def __call__(self, image: np.ndarray, lapping_area=(0, 0)):
        """operator(): returns (keypoints[KP_DTYPE], descriptors[n,32] uint8, n_mono); -1 for an empty image."""
        if image is None or image.size == 0:
            st = lib().ft_extract(self._h, None, 0, 0, 0, 0, 0, None, None, 0, None, None)
            if st == _capi.FT_ERR_EMPTY:
                return None, None, -1
            check(st)
        image = np.ascontiguousarray(image, np.uint8)
        cap = self.max_keypoints
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, nm = C.c_int(), C.c_int()
        check(lib().ft_extract(self._h, ptr(image), image.shape[1], image.shape[0], image.strides[0],
                               lapping_area[0], lapping_area[1], ptr(kps), ptr(desc), cap, C.byref(n), C.byref(nm)))
        return kps[:n.value].copy(), desc[:n.value].copy(), nm.value

    def extract_batch(self, images, lapping_area=(0, 0), on_device=False, width=None, height=None, stride=None, pinned=False):
        """images: list of host arrays or DeviceBuffers (frames already in HBM).  pinned: the result arrays live in pinned host
        memory of the context - a batch of more than eight images is then written there by the device itself, in the
        reference's output order (k_deliver_ordered)"""
        B = len(images)
        if not on_device:
            width, height = images[0].shape[1], images[0].shape[0]
            stride = width
            images = [np.ascontiguousarray(im, np.uint8) for im in images]
        ptrs, keep = _image_ptrs(images, on_device)
        cap = self.max_keypoints
        alloc = self.ctx.pinned_array if pinned else (lambda shape, dt: np.zeros(shape, dt))
        kps = alloc((B, cap), KP_DTYPE)
        desc = alloc((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        nm = np.zeros(B, np.int32)
        check(lib().ft_extract_batch(self._h, ptrs, B, int(on_device), width, height, stride, lapping_area[0],
                                     lapping_area[1], ptr(kps), ptr(desc), cap, ptr(n), ptr(nm)))
        return [(kps[b, :n[b]].copy(), desc[b, :n[b]].copy(), int(nm[b])) for b in range(B)]

    def extract_batch_into(self, ptrs, B, on_device, width, height, stride, lapping_area, kps, desc, n, nm):
        """ft_extract_batch into arrays the caller keeps (kps [B, cap] KP_DTYPE, desc [B, cap, 32], n / nm [B] int32): the lean form
        for loops that must not time Python; ptrs = the (pointer array, keep-alive) pair of image_ptrs()"""
        check(lib().ft_extract_batch(self._h, ptrs, B, int(on_device), width, height, stride, lapping_area[0], lapping_area[1],
                                     ptr(kps), ptr(desc), kps.shape[1], ptr(n), ptr(nm)))

    @staticmethod
    def image_ptrs(images, on_device):
        return _image_ptrs(images, on_device)

    # mvImagePyramid[level] (host copy) of image slot `slot`
    def image_pyramid_level(self, level, slot=0) -> np.ndarray:
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(lib().ft_extractor_download_level(self._h, slot, level, ptr(out), w))
        return out

    def blurred_level(self, level, slot=0) -> np.ndarray:
        """the level after GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101), through the descriptor kernel's blur routines"""
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(lib().ft_extractor_download_blurred_level(self._h, slot, level, ptr(out), w))
        return out

    def candidates(self, level, slot=0) -> np.ndarray:
        n = C.c_int()
        check(lib().ft_extractor_download_candidates(self._h, slot, level, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        check(lib().ft_extractor_download_candidates(self._h, slot, level, ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def octree_on_device(self, level, xys, tiers=7):
        """test tap: the device formulation of DistributeOctTree of one level on the given (x, y, score) candidates;
        returns (retained (x, y, score) rows in the reference's result order, tier that produced them; 0 = left to the host)"""
        xys = np.ascontiguousarray(xys, np.int32).reshape(-1, 3)
        out = np.zeros((len(xys) + 64, 3), np.int32)
        n, tier = C.c_int(), C.c_int()
        check(lib().ft_extractor_octree_on_device(self._h, level, ptr(xys), len(xys), tiers, ptr(out), len(out), C.byref(n), C.byref(tier)))
        return out[:n.value].copy(), tier.value


class FrameView:
    """Owns the arrays behind an ft_frame_view (the fields CudaFrame::setMemory marshals)."""

    def __init__(self, keys, descriptors, scale_factors, bounds, mbf=0.0, mb=0.0, uright=None, holder_obs=None,
                 keys_right=None, left_to_right=None, right_to_left=None, cam_model=0, cam=None, Trl=None):
        self.keys = np.ascontiguousarray(keys)
        self.keys_right = None if keys_right is None else np.ascontiguousarray(keys_right)
        nleft = -1 if keys_right is None else len(self.keys)
        n = len(self.keys) + (0 if keys_right is None else len(self.keys_right))
        self.descriptors = np.ascontiguousarray(descriptors, np.uint8).reshape(n, 32)
        self.sf = np.ascontiguousarray(scale_factors, np.float32)
        self.uright = None if uright is None else np.ascontiguousarray(uright, np.float32)
        self.holder_obs = (np.full(n, -1, np.int32) if holder_obs is None
                           else np.ascontiguousarray(holder_obs, np.int32).copy())
        self.l2r = None if left_to_right is None else np.ascontiguousarray(left_to_right, np.int32)
        self.r2l = None if right_to_left is None else np.ascontiguousarray(right_to_left, np.int32)
        minx, miny, maxx, maxy = [np.float32(v) for v in bounds]
        f = _capi.FrameView()
        f.N, f.Nleft = n, nleft
        f.mnMinX, f.mnMinY, f.mnMaxX, f.mnMaxY = minx, miny, maxx, maxy
        f.grid_inv_w = np.float32(64) / np.float32(maxx - minx)   # Frame.cc:184-185
        f.grid_inv_h = np.float32(48) / np.float32(maxy - miny)
        f.mbf, f.mb = float(mbf), float(mb)
        f.keys, f.keys_right, f.descriptors = ptr(self.keys), ptr(self.keys_right), ptr(self.descriptors)
        f.uright, f.holder_obs = ptr(self.uright), ptr(self.holder_obs)
        f.left_to_right, f.right_to_left = ptr(self.l2r), ptr(self.r2l)
        f.cam_model = cam_model
        cam = np.zeros(8, np.float32) if cam is None else np.asarray(cam, np.float32)
        for i in range(8):
            f.cam[i] = float(cam[i]) if i < len(cam) else 0.0
        Trl = np.eye(3, 4, dtype=np.float32) if Trl is None else np.asarray(Trl, np.float32).reshape(3, 4)
        for i in range(12):
            f.Trl[i] = float(Trl.flat[i])
        f.scale_factors, f.nlevels = ptr(self.sf), len(self.sf)
        self.c, self.N, self.Nleft = f, n, nleft


class KernelController:
    """Static facade of the reference (include/Kernels/KernelController.h), minus the run-mode flags:
    this build has no CPU twin to switch to."""

    @staticmethod
    def launchStereoMatchKernel(exL: ORBextractor, exR: ORBextractor, keysL, keysR, descL, descR, mbf, mb,
                                median_cut=True, slot=0):
        """-> dict(uright=mvuRight, depth=mvDepth, sad, n).  KernelController.h:31-36 / Frame.cc:1007-1063."""
        keysL, keysR = np.ascontiguousarray(keysL), np.ascontiguousarray(keysR)
        descL, descR = np.ascontiguousarray(descL, np.uint8), np.ascontiguousarray(descR, np.uint8)
        nL, nR = len(keysL), len(keysR)
        ur = np.zeros(max(nL, 1), np.float32)
        dp = np.zeros(max(nL, 1), np.float32)
        sad = np.zeros(max(nL, 1), np.int32)
        nm = C.c_int()
        check(lib().ft_stereo_match(exL._h, exR._h, slot, ptr(keysL), nL, ptr(keysR), nR, ptr(descL), ptr(descR),
                                    float(mbf), float(mb), int(median_cut), ptr(ur), ptr(dp), ptr(sad), C.byref(nm)))
        return dict(uright=ur[:nL], depth=dp[:nL], sad=sad[:nL], n=nm.value)

    @staticmethod
    def launchFisheyeStereoMatchKernel(ctx: Context, descL, descR):
        """-> dict(matches, best, second).  KernelController.h:38 / Frame.cc:1231-1255."""
        descL, descR = np.ascontiguousarray(descL, np.uint8), np.ascontiguousarray(descR, np.uint8)
        nL, nR = len(descL), len(descR)
        m = np.full(max(nL, 1), -1, np.int32)
        b = np.zeros(max(nL, 1), np.int32)
        s = np.zeros(max(nL, 1), np.int32)
        check(lib().ft_fisheye_match(ctx._h, ptr(descL), nL, ptr(descR), nR, ptr(m), ptr(b), ptr(s)))
        return dict(matches=m[:nL], best=b[:nL], second=s[:nL], n=int((m[:nL] >= 0).sum()))

    @staticmethod
    def launchSearchLocalPointsKernel(ctx: Context, F: FrameView, pts: dict, th: float, nn_ratio: float = 0.8):
        """KernelController.h:40-42 + the acceptance loop of ORBmatcher.cc:241-308."""
        M = len(pts["skip"])
        keep = {}

        def arr(k, dt):
            keep[k] = np.ascontiguousarray(pts[k], dt)
            return ptr(keep[k])

        P = _capi.LocalPoints()
        P.M = M
        P.skip, P.in_view, P.in_view_r = arr("skip", np.uint8), arr("in_view", np.uint8), arr("in_view_r", np.uint8)
        P.level, P.level_r = arr("level", np.int32), arr("level_r", np.int32)
        P.view_cos, P.view_cos_r = arr("view_cos", np.float32), arr("view_cos_r", np.float32)
        P.proj_x, P.proj_y = arr("proj_x", np.float32), arr("proj_y", np.float32)
        P.proj_xr, P.proj_yr = arr("proj_xr", np.float32), arr("proj_yr", np.float32)
        P.descriptors, P.observations = arr("descriptors", np.uint8), arr("observations", np.int32)
        assign = np.zeros(max(F.N, 1), np.int32)
        outs = [np.zeros(max(M, 1), np.int32) for _ in range(10)]
        nm = C.c_int()
        check(lib().ft_search_local_points(ctx._h, C.byref(F.c), C.byref(P), th, nn_ratio, ptr(assign), C.byref(nm),
                                           *[ptr(o) for o in outs]))
        names = ["best_dist", "best_dist2", "best_level", "best_level2", "best_idx",
                 "best_dist_r", "best_dist2_r", "best_level_r", "best_level2_r", "best_idx_r"]
        r = {k: o[:M] for k, o in zip(names, outs)}
        r["assign"], r["n"] = assign[:F.N], nm.value
        return r

    @staticmethod
    def launchPoseEstimationKernel(ctx: Context, Cur: FrameView, last: dict, Tcw, th, forward=False, backward=False,
                                   check_orientation=True, Trl=None):
        """KernelController.h:44-46 + the histogram loop of ORBmatcher.cc:2013-2081.  Tcw: 3x4 matrix, or an SE3 (then Trl, an
        SE3 as well, is the right camera's pose of a two-camera frame): the CPU branch's Sophus arithmetic."""
        N = len(last["valid"])
        keep = {}

        def arr(k, dt):
            keep[k] = np.ascontiguousarray(last[k], dt)
            return ptr(keep[k])

        Lp = _capi.LastPoints()
        Lp.N = N
        Lp.valid, Lp.world_pos = arr("valid", np.uint8), arr("world_pos", np.float32)
        Lp.descriptors, Lp.observations = arr("descriptors", np.uint8), arr("observations", np.int32)
        Lp.octave, Lp.angle = arr("octave", np.int32), arr("angle", np.float32)
        assign = np.zeros(max(Cur.N, 1), np.int32)
        outs = [np.zeros(max(N, 1), np.int32) for _ in range(4)]
        nm = C.c_int()
        if isinstance(Tcw, SE3):  # the Sophus form the reference's CPU branch multiplies with (ft_search_last_frame_se3)
            check(lib().ft_search_last_frame_se3(ctx._h, C.byref(Cur.c), C.byref(Lp), C.byref(Tcw.c), None if Trl is None else C.byref(Trl.c),
                                                 th, int(forward), int(backward), int(check_orientation), ptr(assign), C.byref(nm),
                                                 *[ptr(o) for o in outs]))
        else:
            T = np.ascontiguousarray(np.asarray(Tcw, np.float32).reshape(3, 4))
            check(lib().ft_search_last_frame(ctx._h, C.byref(Cur.c), C.byref(Lp), ptr(T), th, int(forward), int(backward),
                                             int(check_orientation), ptr(assign), C.byref(nm), *[ptr(o) for o in outs]))
        names = ["best_dist", "best_idx", "best_dist_r", "best_idx_r"]
        r = {k: o[:N] for k, o in zip(names, outs)}
        r["assign"], r["n"] = assign[:Cur.N], nm.value
        return r

    @staticmethod
    def descriptor_distance(ctx: Context, a, b):
        a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
        n = len(a)
        d = np.zeros(max(n, 1), np.int32)
        check(lib().ft_descriptor_distance(ctx._h, ptr(a), ptr(b), n, ptr(d)))
        return d[:n]


class Vocabulary:
    """ORBVocabulary (DBoW2 TemplatedVocabulary<FORB>) with the tree walk of Frame::ComputeBoW on the device.
    Build from arrays (node 0 = root, see ft_vocabulary_create) or from ORB-SLAM3's ORBvoc.txt text format."""

    def __init__(self, ctx: Context, k=None, L=None, scoring=0, weighting=0, parent=None, is_leaf=None, descriptors=None,
                 weights=None, path=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        if path is not None:
            check(lib().ft_vocabulary_load_text(ctx._h, str(path).encode(), C.byref(self._h)))
        else:
            parent = np.ascontiguousarray(parent, np.int32)
            is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            descriptors = np.ascontiguousarray(descriptors, np.uint8)
            weights = np.ascontiguousarray(weights, np.float64)
            check(lib().ft_vocabulary_create(ctx._h, k, L, scoring, weighting, len(parent), ptr(parent), ptr(is_leaf),
                                             ptr(descriptors), ptr(weights), C.byref(self._h)))
        k_, L_, nn, nw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().ft_vocabulary_info(self._h, C.byref(k_), C.byref(L_), C.byref(nn), C.byref(nw)))
        self.k, self.L, self.n_nodes, self.n_words = k_.value, L_.value, nn.value, nw.value

    def close(self):
        if self._h:
            lib().ft_vocabulary_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, descriptors, levelsup=4, device_ptr=None, n=None):
        """Frame::ComputeBoW: -> dict(word, node, weight per feature; bow_ids, bow_values (BowVector);
        fv_nodes, fv_offsets, fv_features (FeatureVector, CSR)).  device_ptr / n: descriptors already in HBM."""
        if device_ptr is None:
            d = np.ascontiguousarray(descriptors, np.uint8)
            n = len(d)
            src, on_dev = ptr(d), 0
        else:
            src, on_dev = C.c_void_p(device_ptr), 1
        m = max(n, 1)
        word = np.zeros(m, np.uint32); node = np.zeros(m, np.uint32); w = np.zeros(m, np.float64)
        bi = np.zeros(m, np.uint32); bv = np.zeros(m, np.float64); nb = C.c_int(0)
        fn = np.zeros(m, np.uint32); fo = np.zeros(m + 1, np.int32); ff = np.zeros(m, np.uint32); nf = C.c_int(0)
        check(lib().ft_bow_transform(self._h, src, n, on_dev, levelsup, ptr(word), ptr(node), ptr(w), ptr(bi), ptr(bv), m,
                                     C.byref(nb), ptr(fn), ptr(fo), ptr(ff), m, C.byref(nf)))
        return dict(word=word[:n], node=node[:n], weight=w[:n], bow_ids=bi[:nb.value], bow_values=bv[:nb.value],
                    fv_nodes=fn[:nf.value], fv_offsets=fo[:nf.value + 1], fv_features=ff[:fo[nf.value]])


class StereoFrontend:
    """Fused extract(left) + extract(right) + ComputeStereoMatches for batches of rectified pairs."""

    def __init__(self, ctx: Context, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, width, height,
                 max_batch, mbf, mb, pinned_outputs=True):
        self.ctx = ctx
        self._h = C.c_void_p()
        check(lib().ft_stereo_frontend_create(ctx._h, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast,
                                              width, height, max_batch, float(mbf), float(mb), C.byref(self._h)))
        self.width, self.height, self.max_batch = width, height, max_batch
        args = (ctx, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, width, height, max_batch)
        self.left = ORBextractor(*args, _handle=lib().ft_stereo_frontend_left(self._h))
        self.right = ORBextractor(*args, _handle=lib().ft_stereo_frontend_right(self._h))
        self.capacity = self.left.max_keypoints
        B, cap = max_batch, self.capacity
        alloc = ctx.pinned_array if pinned_outputs else (lambda shape, dt: np.zeros(shape, dt))
        self._kL = alloc((B, cap), KP_DTYPE)
        self._kR = alloc((B, cap), KP_DTYPE)
        self._dL = alloc((B, cap, 32), np.uint8)
        self._dR = alloc((B, cap, 32), np.uint8)
        self._nL = np.zeros(B, np.int32)
        self._nR = np.zeros(B, np.int32)
        self._ur = alloc((B, cap), np.float32)
        self._dp = alloc((B, cap), np.float32)
        self._nm = np.zeros(B, np.int32)

    def device_descriptors(self, slot, right=False):
        """device pointer of the descriptors of pair `slot` of the last batch (see ft_stereo_frontend_device_descriptors)"""
        p, n = C.c_void_p(), C.c_int()
        check(lib().ft_stereo_frontend_device_descriptors(self._h, slot, 1 if right else 0, C.byref(p), C.byref(n)))
        return p.value

    def close(self):
        if getattr(self, "_h", None) and self.ctx._h:
            lib().ft_stereo_frontend_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_raw(self, ptrsL, ptrsR, batch, on_device, stride):
        """No per-call allocation: results land in the preallocated arrays (used by bench.py)."""
        check(lib().ft_stereo_frontend_process(self._h, ptrsL, ptrsR, batch, int(on_device), self.width, self.height,
                                               stride, ptr(self._kL), ptr(self._dL), ptr(self._nL), ptr(self._kR),
                                               ptr(self._dR), ptr(self._nR), self.capacity, ptr(self._ur),
                                               ptr(self._dp), ptr(self._nm)))

    def submit_raw(self, ptrsL, ptrsR, batch, on_device, stride):
        """Asynchronous half: enqueue a batch; results are valid after wait()."""
        check(lib().ft_stereo_frontend_submit(self._h, ptrsL, ptrsR, batch, int(on_device), self.width, self.height,
                                              stride, ptr(self._kL), ptr(self._dL), ptr(self._nL), ptr(self._kR),
                                              ptr(self._dR), ptr(self._nR), self.capacity, ptr(self._ur),
                                              ptr(self._dp), ptr(self._nm)))

    def wait(self):
        check(lib().ft_stereo_frontend_wait(self._h))

    def process(self, imagesL, imagesR, on_device=False, stride=None):
        B = len(imagesL)
        pL, keepL = _image_ptrs(imagesL, on_device)
        pR, keepR = _image_ptrs(imagesR, on_device)
        # host frames were made dense by _image_ptrs: their row stride is the width whatever the caller's view had
        self.process_raw(pL, pR, B, on_device, (stride or self.width) if on_device else self.width)
        out = []
        for b in range(B):
            nl, nr = int(self._nL[b]), int(self._nR[b])
            out.append(dict(keysL=self._kL[b, :nl].copy(), descL=self._dL[b, :nl].copy(),
                            keysR=self._kR[b, :nr].copy(), descR=self._dR[b, :nr].copy(),
                            uright=self._ur[b, :nl].copy(), depth=self._dp[b, :nl].copy(), n=int(self._nm[b])))
        return out


FRUSTUM_FIELDS = [("in_view", np.uint8), ("in_view_r", np.uint8), ("level", np.int32), ("level_r", np.int32),
                  ("view_cos", np.float32), ("view_cos_r", np.float32), ("proj_x", np.float32), ("proj_y", np.float32),
                  ("proj_xr", np.float32), ("proj_yr", np.float32), ("depth", np.float32), ("depth_r", np.float32)]


def make_pose(Rcw, tcw, tlr=(0, 0, 0)):
    """ft_frame_pose from mRcw / mtcw; mOw = -Rcw^T tcw in float32 (Frame::UpdatePoseMatrices)."""
    Rcw = np.asarray(Rcw, np.float32).reshape(3, 3)
    tcw = np.asarray(tcw, np.float32).reshape(3)
    Ow = (-(Rcw.T.astype(np.float32) @ tcw)).astype(np.float32)
    T = _capi.FramePose()
    T.Rcw[:] = [float(v) for v in Rcw.reshape(-1)]
    T.tcw[:] = [float(v) for v in tcw]
    T.Ow[:] = [float(v) for v in Ow]
    T.tlr[:] = [float(v) for v in np.asarray(tlr, np.float32)]
    return T


def _map_points(pts: dict, keep: dict, ctx=None):
    M = len(pts["world_pos"])
    P = _capi.MapPoints()
    P.M = M

    def arr(k, dt):
        if pts.get(k) is None:
            return None
        keep[k] = _pinned_copy(ctx, pts[k], dt)
        return ptr(keep[k])
    P.skip = arr("skip", np.uint8)
    P.world_pos, P.normal = arr("world_pos", np.float32), arr("normal", np.float32)
    P.max_distance, P.min_distance = arr("max_distance", np.float32), arr("min_distance", np.float32)
    P.descriptors, P.observations = arr("descriptors", np.uint8), arr("observations", np.int32)
    return P, M


def _frustum_result(M, ctx=None):
    """ctx: the arrays live in pinned host memory of that context (the device fills them directly)"""
    outs = {k: (np.zeros(max(M, 1), dt) if ctx is None else ctx.pinned_array((max(M, 1),), dt)) for k, dt in FRUSTUM_FIELDS}
    R = _capi.FrustumResult()
    for k, _ in FRUSTUM_FIELDS:
        setattr(R, k, ptr(outs[k]))
    return R, outs


def is_in_frustum(ctx: Context, F: "FrameView", pose, pts: dict, viewing_cos_limit: float, log_scale_factor: float):
    """Frame::isInFrustum for all points (ft_is_in_frustum); pts: world_pos, normal, max_distance, min_distance[, skip]."""
    keep = {}
    P, M = _map_points(pts, keep)
    R, outs = _frustum_result(M)
    n = C.c_int()
    check(lib().ft_is_in_frustum(ctx._h, C.byref(F.c), C.byref(pose), C.byref(P), viewing_cos_limit, log_scale_factor,
                                 C.byref(R), C.byref(n)))
    r = {k: v[:M] for k, v in outs.items()}
    r["n"] = n.value
    return r


class TrackedFrame:
    """Device-resident frame for the projection searches (ft_tracked_frame_*)."""

    def __init__(self, ctx: Context, max_keypoints: int, max_points: int):
        self.ctx = ctx
        self._h = C.c_void_p()
        check(lib().ft_tracked_frame_create(ctx._h, max_keypoints, max_points, C.byref(self._h)))
        self.N = 0

    def close(self):
        if getattr(self, "_h", None) and self.ctx._h:
            lib().ft_tracked_frame_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, F: "FrameView"):
        check(lib().ft_tracked_frame_upload(self._h, C.byref(F.c)))
        self.N = F.c.N

    def bind_stereo(self, fe: "StereoFrontend", slot: int, meta: "FrameView"):
        check(lib().ft_tracked_frame_bind_stereo(self._h, fe._h, slot, C.byref(meta.c)))
        self.N = meta.c.N

    def holder_obs(self):
        out = np.zeros(max(self.N, 1), np.int32)
        check(lib().ft_tracked_frame_holder_obs(self._h, ptr(out)))
        return out[:self.N]

    def search_last_frame(self, last: dict, Tcw, th, forward=False, backward=False, check_orientation=True, Trl=None):
        N = len(last["valid"])
        keep = {k: np.ascontiguousarray(last[k], dt) for k, dt in
                (("valid", np.uint8), ("world_pos", np.float32), ("descriptors", np.uint8), ("observations", np.int32),
                 ("octave", np.int32), ("angle", np.float32))}
        Lp = _capi.LastPoints()
        Lp.N = N
        for k in keep:
            setattr(Lp, k, ptr(keep[k]))
        assign = np.zeros(max(self.N, 1), np.int32)
        n = C.c_int()
        if isinstance(Tcw, SE3):
            check(lib().ft_tracked_frame_search_last_frame_se3(self._h, C.byref(Lp), C.byref(Tcw.c), None if Trl is None else C.byref(Trl.c),
                                                               th, int(forward), int(backward), int(check_orientation), ptr(assign),
                                                               C.byref(n)))
        else:
            T = np.ascontiguousarray(np.asarray(Tcw, np.float32).reshape(3, 4))
            check(lib().ft_tracked_frame_search_last_frame(self._h, C.byref(Lp), ptr(T), th, int(forward), int(backward),
                                                           int(check_orientation), ptr(assign), C.byref(n)))
        return dict(assign=assign[:self.N], n=n.value)

    def track_local_map(self, pose, pts: dict, viewing_cos_limit, log_scale_factor, th, nn_ratio=0.8, far_points=False,
                        th_far_points=0.0):
        keep = {}
        P, M = _map_points(pts, keep)
        R, outs = _frustum_result(M)
        assign = np.zeros(max(self.N, 1), np.int32)
        n, nt = C.c_int(), C.c_int()
        check(lib().ft_tracked_frame_track_local_map(self._h, C.byref(pose), C.byref(P), viewing_cos_limit, log_scale_factor,
                                                     th, nn_ratio, int(far_points), th_far_points, C.byref(R), C.byref(nt),
                                                     ptr(assign), C.byref(n)))
        r = {k: v[:M] for k, v in outs.items()}
        r.update(assign=assign[:self.N], n=n.value, n_to_match=nt.value)
        return r


class TrackedBatch:
    """B device-resident frames searched through one set of launches (ft_tracked_batch_*): per frame the results of
    TrackedFrame's calls.  The marshalled inputs of a call (ctypes arrays of ft_last_points / ft_map_points) can be prepared
    once with prepare_last / prepare_local and handed in again, so that a benchmark loop does not time Python."""

    def __init__(self, ctx: Context, max_frames: int, max_keypoints: int, max_points: int, pinned=False):
        """pinned: the assignment arrays the searches fill live in pinned host memory (the device writes them directly) and
        prepare_last / prepare_local called on the object put the point arrays there too (the device reads them in place)"""
        self.ctx = ctx
        self._h = C.c_void_p()
        check(lib().ft_tracked_batch_create(ctx._h, max_frames, max_keypoints, max_points, C.byref(self._h)))
        self.N = []
        self._pin_ctx = ctx if pinned else None

    def close(self):
        if getattr(self, "_h", None) and self.ctx._h:
            lib().ft_tracked_batch_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, frames):
        """frames: list of FrameView, or the (array, views) pair prepare_frames returned"""
        arr, views = frames if isinstance(frames, tuple) else self.prepare_frames(frames)
        check(lib().ft_tracked_batch_upload(self._h, len(views), arr))
        self._after_load([F.c.N for F in views])

    @staticmethod
    def prepare_frames(views):
        arr = (_capi.FrameView * len(views))(*[F.c for F in views])
        return arr, list(views)

    def _after_load(self, counts):
        self.N = list(counts)
        if self._pin_ctx is not None:
            self._assign = [self._pin_ctx.pinned_array((max(n, 1),), np.int32) for n in self.N]
        else:
            self._assign = [np.zeros(max(n, 1), np.int32) for n in self.N]
        self._assign_ptrs = (C.c_void_p * len(self.N))(*[ptr(a) for a in self._assign])
        self._nm = np.zeros(len(self.N), np.int32)
        self._nt = np.zeros(len(self.N), np.int32)

    def bind_fisheye(self, exL, exR, views, lap_l, lap_r, slot0=0, want_tables=True, rig=None, level_sigma2=None, slot0_right=None):
        """ft_tracked_batch_bind_fisheye: views = FrameViews (or the pair prepare_frames returned) whose keys / keys_right are the
        host copies of what exL / exR extracted last (slots slot0 ...).  rig = make_fisheye_rig(...): with the triangulation filter
        of Frame::ComputeStereoFishEyeMatches.  -> [(left_to_right, right_to_left)] (with a rig: [(l2r, r2l, depth, p3d, n)]) or None"""
        arr, vs = views if isinstance(views, tuple) else self.prepare_frames(views)
        n = len(vs)
        l2r = r2l = dep = p3 = None
        pl = pr = pd = pp = None
        nst = None
        ls2 = None if level_sigma2 is None else np.ascontiguousarray(level_sigma2, np.float32)
        if want_tables:
            l2r = [np.full(max(F.c.Nleft, 1), -1, np.int32) for F in vs]
            r2l = [np.full(max(F.c.N - F.c.Nleft, 1), -1, np.int32) for F in vs]
            pl = (C.c_void_p * n)(*[ptr(a) for a in l2r])
            pr = (C.c_void_p * n)(*[ptr(a) for a in r2l])
            if rig is not None:
                dep = [np.zeros(max(F.c.Nleft, 1), np.float32) for F in vs]
                p3 = [np.zeros((max(F.c.Nleft, 1), 3), np.float32) for F in vs]
                pd = (C.c_void_p * n)(*[ptr(a) for a in dep])
                pp = (C.c_void_p * n)(*[ptr(a) for a in p3])
                nst = np.zeros(n, np.int32)
        # slot0_right: the right images' first slot (exL may be exR: one extractor, e.g. a frame's two images as a batch of two)
        check(lib().ft_tracked_batch_bind_fisheye_slots(self._h, exL._h, exR._h, slot0, slot0 if slot0_right is None else slot0_right, n,
                                                        lap_l[0], lap_l[1], lap_r[0], lap_r[1], arr,
                                                        None if rig is None else C.byref(rig), ptr(ls2), pl, pr, pd, pp, ptr(nst)))
        if [F.c.N for F in vs] != self.N:
            self._after_load([F.c.N for F in vs])
        if not want_tables:
            return None
        if rig is None:
            return [(l2r[f][:vs[f].c.Nleft], r2l[f][:vs[f].c.N - vs[f].c.Nleft]) for f in range(n)]
        return [(l2r[f][:vs[f].c.Nleft], r2l[f][:vs[f].c.N - vs[f].c.Nleft], dep[f][:vs[f].c.Nleft], p3[f][:vs[f].c.Nleft], int(nst[f]))
                for f in range(n)]

    def holder_obs(self, f):
        out = np.zeros(max(self.N[f], 1), np.int32)
        check(lib().ft_tracked_batch_holder_obs(self._h, f, ptr(out)))
        return out[:self.N[f]]

    @staticmethod
    def prepare_last(lasts, Tcws, forward=None, backward=None, ctx=None):
        """lasts: list of dicts (TrackedFrame.search_last_frame's), Tcws: list of 3x4 matrices, or of SE3 (then Trl = list of SE3 or None).
        ctx: the point arrays are copied into pinned host memory of that context (ft_host_malloc) - the device reads them in place"""
        n = len(lasts)
        keep = []
        arr = (_capi.LastPoints * n)()
        for f, last in enumerate(lasts):
            k = {key: _pinned_copy(ctx, last[key], dt) for key, dt in
                 (("valid", np.uint8), ("world_pos", np.float32), ("descriptors", np.uint8), ("observations", np.int32),
                  ("octave", np.int32), ("angle", np.float32))}
            keep.append(k)
            arr[f].N = len(k["valid"])
            for key in k:
                setattr(arr[f], key, ptr(k[key]))
        se3 = n > 0 and isinstance(Tcws[0], SE3)
        if se3:
            T = (_capi.SE3 * n)(*[t.c for t in Tcws])
        else:
            T = np.ascontiguousarray(np.stack([np.asarray(t, np.float32).reshape(3, 4) for t in Tcws]), np.float32)
        fw = None if forward is None else np.ascontiguousarray(forward, np.int32)
        bw = None if backward is None else np.ascontiguousarray(backward, np.int32)
        return dict(n=n, arr=arr, keep=keep, T=T, se3=se3, fw=fw, bw=bw)

    def search_last_frame(self, lasts, Tcws=None, th=7.0, forward=None, backward=None, check_orientation=True, Trl=None, copy=True,
                          submit=False):
        """submit=True: ft_tracked_batch_submit_search_last_frame - returns at once, wait() delivers the results"""
        pl = lasts if isinstance(lasts, dict) and "arr" in lasts else self.prepare_last(lasts, Tcws, forward, backward, ctx=getattr(self, "_pin_ctx", None))
        n = pl["n"]
        L = lib()
        if pl["se3"]:
            trl = None if Trl is None else (_capi.SE3 * n)(*[t.c for t in Trl])
            fn = L.ft_tracked_batch_submit_search_last_frame_se3 if submit else L.ft_tracked_batch_search_last_frame_se3
            check(fn(self._h, n, pl["arr"], pl["T"], trl, th, ptr(pl["fw"]), ptr(pl["bw"]), int(check_orientation), self._assign_ptrs, ptr(self._nm)))
        else:
            fn = L.ft_tracked_batch_submit_search_last_frame if submit else L.ft_tracked_batch_search_last_frame
            check(fn(self._h, n, pl["arr"], ptr(pl["T"]), th, ptr(pl["fw"]), ptr(pl["bw"]), int(check_orientation), self._assign_ptrs, ptr(self._nm)))
        self._pending = (pl, "last")   # (the arrays of a submitted call stay referenced until the wait)
        if submit or not copy:
            return None
        return self._results_last(n)

    def _results_last(self, n):
        return [dict(assign=self._assign[f][:self.N[f]].copy(), n=int(self._nm[f])) for f in range(n)]

    def wait(self, copy=True):
        """ft_tracked_batch_wait: the results of the submitted search (as the blocking call returns them), or None"""
        check(lib().ft_tracked_batch_wait(self._h))
        pend, self._pending = getattr(self, "_pending", None), None
        if pend is None or not copy:
            return None
        pl, kind = pend
        return self._results_last(pl["n"]) if kind == "last" else self._results_local(pl)

    @staticmethod
    def prepare_local(poses, pts_list, want_frustum=True, ctx=None):
        """ctx: the point arrays are copied into pinned host memory of that context - the device reads them in place"""
        n = len(pts_list)
        keep, outs = [], []
        P = (_capi.MapPoints * n)()
        R = (_capi.FrustumResult * n)() if want_frustum else None
        for f, pts in enumerate(pts_list):
            k = {}
            Pf, M = _map_points(pts, k, ctx)
            keep.append(k)
            P[f] = Pf
            if want_frustum:
                Rf, o = _frustum_result(M, ctx)
                R[f] = Rf
                outs.append((M, o))
        T = (_capi.FramePose * n)(*poses)
        return dict(n=n, P=P, R=R, T=T, keep=keep, outs=outs)

    def track_local_map(self, poses, pts_list=None, viewing_cos_limit=0.5, log_scale_factor=0.0, th=1.0, nn_ratio=0.8, far_points=False,
                        th_far_points=0.0, copy=True, submit=False):
        pl = poses if isinstance(poses, dict) and "P" in poses else self.prepare_local(poses, pts_list, ctx=getattr(self, "_pin_ctx", None))
        n = pl["n"]
        fn = lib().ft_tracked_batch_submit_track_local_map if submit else lib().ft_tracked_batch_track_local_map
        check(fn(self._h, n, pl["T"], pl["P"], viewing_cos_limit, log_scale_factor, th, nn_ratio, int(far_points), th_far_points, pl["R"],
                 ptr(self._nt), self._assign_ptrs, ptr(self._nm)))
        self._pending = (pl, "local")
        if submit or not copy:
            return None
        return self._results_local(pl)

    def _results_local(self, pl):
        n = pl["n"]
        res = []
        for f in range(n):
            r = {}
            if pl["R"] is not None:
                M, o = pl["outs"][f]
                r = {k: v[:M].copy() for k, v in o.items()}
            r.update(assign=self._assign[f][:self.N[f]].copy(), n=int(self._nm[f]), n_to_match=int(self._nt[f]))
            res.append(r)
        return res


def _pinned_copy(ctx, a, dtype):
    """a contiguous copy of `a`: in pinned host memory of ctx (Context.pinned_array) when ctx is given"""
    a = np.ascontiguousarray(a, dtype)
    if ctx is None:
        return a
    p = ctx.pinned_array(a.shape if a.size else (1,), a.dtype)
    if a.size:
        p[...] = a
        return p
    return p[:0]


def make_fisheye_rig(cam1, cam2, Rlr, tlr, precision=1e-6):
    """ft_fisheye_rig: mpCamera / mpCamera2 (fx fy cx cy k1..k4), mRlr, mtlr, KannalaBrandt8::precision"""
    rig = _capi.FisheyeRig()
    rig.cam1[:] = [float(v) for v in cam1]
    rig.cam2[:] = [float(v) for v in cam2]
    rig.precision = precision
    rig.Rlr[:] = [float(v) for v in np.asarray(Rlr, np.float32).reshape(-1)]
    rig.tlr[:] = [float(v) for v in np.asarray(tlr, np.float32).reshape(-1)]
    return rig


def fisheye_stereo(ctx: Context, cam1, cam2, Rlr, tlr, descL, keysL, descR, keysR, level_sigma2, precision=1e-6):
    """Frame::ComputeStereoFishEyeMatches on the lapping-area subsets (ft_fisheye_stereo)."""
    rig = make_fisheye_rig(cam1, cam2, Rlr, tlr, precision)
    descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
    keysL = np.ascontiguousarray(keysL); keysR = np.ascontiguousarray(keysR)
    ls2 = np.ascontiguousarray(level_sigma2, np.float32)
    nL, nR = len(descL), len(descR)
    m = np.full(max(nL, 1), -1, np.int32)
    d = np.zeros(max(nL, 1), np.float32)
    p = np.zeros((max(nL, 1), 3), np.float32)
    n = C.c_int()
    check(lib().ft_fisheye_stereo(ctx._h, C.byref(rig), ptr(descL), ptr(keysL), nL, ptr(descR), ptr(keysR), nR, ptr(ls2),
                                  len(ls2), ptr(m), ptr(d), ptr(p), C.byref(n)))
    return dict(matches=m[:nL], depth=d[:nL], p3d=p[:nL], n=n.value)


class SE3:
    """A pose as Sophus::SE3f holds it: unit quaternion (x, y, z, w = Eigen::Quaternionf::coeffs()) and translation"""

    def __init__(self, q, t):
        self.q = np.ascontiguousarray(q, np.float32).reshape(4)
        self.t = np.ascontiguousarray(t, np.float32).reshape(3)
        self.c = _capi.SE3()
        self.c.q[:] = [float(v) for v in self.q]
        self.c.t[:] = [float(v) for v in self.t]


class BowSide:
    """One side of ORBmatcher::SearchByBoW: the FeatureVector of Vocabulary.transform (fv_nodes, fv_offsets, fv_features),
    the descriptors it indexes and the keypoint angles (for mbCheckOrientation).  Owns the arrays behind an ft_bow_side."""

    def __init__(self, fv_nodes, fv_offsets, fv_features, descriptors, angles=None):
        self.fv_nodes = np.ascontiguousarray(fv_nodes, np.uint32)
        self.fv_offsets = np.ascontiguousarray(fv_offsets, np.int32)
        self.fv_features = np.ascontiguousarray(fv_features, np.uint32)
        self.descriptors = np.ascontiguousarray(descriptors, np.uint8)
        self.angles = None if angles is None else np.ascontiguousarray(angles, np.float32)
        assert len(self.fv_offsets) == len(self.fv_nodes) + 1 or (len(self.fv_nodes) == 0 and len(self.fv_offsets) <= 1)
        if len(self.fv_offsets) == 0:
            self.fv_offsets = np.zeros(1, np.int32)
        c = _capi.BowSide()
        c.n, c.n_nodes = len(self.descriptors), len(self.fv_nodes)
        c.fv_nodes, c.fv_offsets, c.fv_features = ptr(self.fv_nodes), ptr(self.fv_offsets), ptr(self.fv_features)
        c.descriptors = ptr(self.descriptors)
        c.angles = None if self.angles is None else ptr(self.angles)
        self.c = c


def search_by_bow(ctx: Context, kf: BowSide, kf_has_point, frame: BowSide, frame_nleft=-1, nn_ratio=0.7, check_orientation=True):
    """ORBmatcher(nn_ratio, check_orientation).SearchByBoW(pKF, F, vpMapPointMatches) (ft_search_by_bow) ->
    dict(matches[frame.n] = keyframe feature index or -1, n = nmatches)"""
    has = np.ascontiguousarray(kf_has_point, np.uint8)
    assert len(has) == kf.c.n
    m = np.full(max(frame.c.n, 1), -1, np.int32)
    n = C.c_int(0)
    check(lib().ft_search_by_bow(ctx._h, C.byref(kf.c), ptr(has), C.byref(frame.c), int(frame_nleft), float(nn_ratio),
                                 int(bool(check_orientation)), ptr(m), C.byref(n)))
    return dict(matches=m[:frame.c.n], n=n.value)


def features_in_area(ctx: Context, F: "FrameView", x, y, r, min_level, max_level, right=None, capacity=512):
    """Frame::GetFeaturesInArea for arrays of queries (ft_features_in_area) -> list of index arrays"""
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32); r = np.ascontiguousarray(r, np.float32)
    lo = np.ascontiguousarray(min_level, np.int32); hi = np.ascontiguousarray(max_level, np.int32)
    rt = None if right is None else np.ascontiguousarray(right, np.uint8)
    nq = len(x)
    idx = np.zeros((max(nq, 1), capacity), np.int32)
    cnt = np.zeros(max(nq, 1), np.int32)
    check(lib().ft_features_in_area(ctx._h, C.byref(F.c), nq, ptr(x), ptr(y), ptr(r), ptr(lo), ptr(hi), ptr(rt), ptr(idx),
                                    capacity, ptr(cnt)))
    return [idx[q, :min(cnt[q], capacity)].copy() for q in range(nq)], cnt[:nq]
