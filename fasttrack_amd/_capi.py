"""ctypes loader of libfasttrack_amd.so (the C ABI declared in include/fasttrack_amd.h).

The library is the product; there is no Python or CPU fallback.  Loading fails loudly when the shared
object is missing, and every compute call raises FastTrackError when no usable gfx950 device exists.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# FT_LIB: another build of the library (A/B measurements of compile-time variants, tools/ab_build.sh); never set in tests
LIB_PATH = os.environ.get("FT_LIB") or os.path.join(_PKG, "libfasttrack_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "fasttrack_amd.h")

FT_OK, FT_ERR_INVALID, FT_ERR_NO_DEVICE, FT_ERR_HIP, FT_ERR_CAPACITY, FT_ERR_EMPTY = 0, -1, -2, -3, -4, -5

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class FastTrackError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"fasttrack_amd status {status}: {message}")
        self.status = status


class FrameView(C.Structure):
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


class SE3(C.Structure):
    """ft_se3: unit quaternion (x, y, z, w) and translation, as Sophus::SE3f holds a pose"""
    _fields_ = [("q", C.c_float * 4), ("t", C.c_float * 3)]


class BowSide(C.Structure):
    """ft_bow_side: a DBoW2 FeatureVector in CSR form with the descriptors / angles it indexes"""
    _fields_ = [("n", C.c_int), ("n_nodes", C.c_int), ("fv_nodes", C.c_void_p), ("fv_offsets", C.c_void_p),
                ("fv_features", C.c_void_p), ("descriptors", C.c_void_p), ("angles", C.c_void_p)]


class FramePose(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("tlr", C.c_float * 3)]


class MapPoints(C.Structure):
    _fields_ = [("M", C.c_int), ("skip", C.c_void_p), ("world_pos", C.c_void_p), ("normal", C.c_void_p),
                ("max_distance", C.c_void_p), ("min_distance", C.c_void_p), ("descriptors", C.c_void_p),
                ("observations", C.c_void_p)]


class FrustumResult(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("in_view", "in_view_r", "level", "level_r", "view_cos", "view_cos_r",
                                          "proj_x", "proj_y", "proj_xr", "proj_yr", "depth", "depth_r")]


def declared_symbols() -> list[str]:
    """Every FT_API entry point declared in include/fasttrack_amd.h."""
    text = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"FT_API[^;(]*?\b(ft_[a-z0-9_]+)\s*\(", text)))


_lib = None


def _hip_runtime_mapped() -> bool:
    """Is the HIP runtime the library binds to (the system one next to hipcc, RTLD_DEEPBIND) already mapped into the process?
    A DIFFERENT copy of the runtime - PyTorch bundles its own under torch/lib - does not count: each copy reads the
    environment when IT initialises."""
    try:
        with open("/proc/self/maps") as f:
            mapped = {ln.split()[-1] for ln in f if "libamdhip64" in ln}
    except OSError:
        return False
    ours = {os.path.realpath(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", n))
            for n in ("libamdhip64.so", "libamdhip64.so.7")}
    return any(os.path.realpath(m) in ours for m in mapped)


def lib() -> C.CDLL:
    """Load the shared library (RTLD_LOCAL | RTLD_DEEPBIND so it binds to the system ROCm runtime even
    when another HIP runtime, e.g. the one bundled with PyTorch, is present in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C fasttrack_amd/csrc`).  fasttrack_amd has no CPU fallback.")
    # Application-side choice, made before the first HIP call of the process (the HIP runtime reads the variable once, when
    # it initialises): a hardware queue per lane of the context, which selects the library's eight-lane table
    # (include/fasttrack_amd.h, ft_context_create).  The library itself never touches the environment.
    # If the HIP runtime is already mapped into the process (torch, a device probe) it may have initialised with four queues:
    # exporting the variable now would make ft_context_hw_queues() report 10 while the runtime multiplexes onto 4 - the lanes
    # of the shipped table would then share queues in an order nobody chose.  In that case the variable is left alone (the
    # context takes private streams, the best four queues offer) and a warning says how to get the lanes.
    if "GPU_MAX_HW_QUEUES" not in os.environ:
        if _hip_runtime_mapped():
            import warnings
            warnings.warn("fasttrack_amd: a HIP runtime was loaded before the library and GPU_MAX_HW_QUEUES is unset - contexts "
                          "will use private streams (about 10 % below the lane table on wide batches); export "
                          "GPU_MAX_HW_QUEUES=10 before the first HIP call of the process", RuntimeWarning, stacklevel=2)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = "10"
    mode = os.RTLD_LOCAL | os.RTLD_NOW | getattr(os, "RTLD_DEEPBIND", 0)
    L = C.CDLL(LIB_PATH, mode=mode)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    ip = C.POINTER(C.c_int)
    L.ft_version.restype = C.c_char_p
    L.ft_last_error.restype = C.c_char_p
    L.ft_device_count.restype = i
    L.ft_device_pci_bus_id.argtypes = [i, C.c_char_p, i]
    L.ft_context_create.argtypes = [i, i, C.POINTER(vp)]
    L.ft_context_destroy.argtypes = [vp]
    L.ft_context_synchronize.argtypes = [vp]
    L.ft_context_device_name.argtypes = [vp, C.c_char_p, i]
    L.ft_context_host_threads.argtypes = [vp]
    L.ft_context_hw_queues.argtypes = [vp]
    L.ft_context_set_lane_map.argtypes = [vp, vp, i]
    L.ft_context_set_option.argtypes = [vp, C.c_char_p, i]
    L.ft_context_get_option.argtypes = [vp, C.c_char_p, ip]
    L.ft_option_describe.argtypes = [i, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), ip, C.POINTER(C.c_char_p)]
    L.ft_option_range.argtypes = [C.c_char_p, ip, ip]
    L.ft_context_save_stats.argtypes = [vp, C.c_char_p]
    L.ft_context_set_kernel_timing.argtypes = [vp, i]
    L.ft_context_get_stat.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.ft_context_reset_stats.argtypes = [vp]
    L.ft_device_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.ft_device_free.argtypes = [vp, vp]
    L.ft_host_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.ft_host_free.argtypes = [vp, vp]
    L.ft_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.ft_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.ft_extractor_create.argtypes = [vp, i, f, i, i, i, i, i, i, C.POINTER(vp)]
    L.ft_extractor_destroy.argtypes = [vp]
    L.ft_extractor_levels.argtypes = [vp]
    L.ft_extractor_max_batch.argtypes = [vp]
    L.ft_extractor_max_keypoints.argtypes = [vp]
    L.ft_extractor_scale_factors.argtypes = [vp, vp, vp, vp, vp]
    L.ft_extractor_features_per_level.argtypes = [vp, vp]
    L.ft_extractor_level_size.argtypes = [vp, i, ip, ip]
    L.ft_extract.argtypes = [vp, vp, i, i, i, i, i, vp, vp, i, ip, ip]
    L.ft_extract_batch.argtypes = [vp, vp, i, i, i, i, i, i, i, vp, vp, i, vp, vp]
    L.ft_extractor_download_level.argtypes = [vp, i, i, vp, i]
    L.ft_extractor_device_level.argtypes = [vp, i, i, C.POINTER(vp), ip]
    L.ft_extractor_download_candidates.argtypes = [vp, i, i, vp, i, ip]
    L.ft_extractor_octree_on_device.argtypes = [vp, i, vp, i, i, vp, i, ip, ip]
    L.ft_stereo_match.argtypes = [vp, vp, i, vp, i, vp, i, vp, vp, f, f, i, vp, vp, vp, ip]
    L.ft_stereo_frontend_create.argtypes = [vp, i, f, i, i, i, i, i, i, f, f, C.POINTER(vp)]
    L.ft_stereo_frontend_destroy.argtypes = [vp]
    L.ft_stereo_frontend_left.argtypes = [vp]
    L.ft_stereo_frontend_left.restype = vp
    L.ft_stereo_frontend_right.argtypes = [vp]
    L.ft_stereo_frontend_right.restype = vp
    L.ft_stereo_frontend_process.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp, vp, vp, vp, i, vp, vp, vp]
    L.ft_stereo_frontend_submit.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp, vp, vp, vp, i, vp, vp, vp]
    L.ft_stereo_frontend_wait.argtypes = [vp]
    L.ft_fisheye_match.argtypes = [vp, vp, i, vp, i, vp, vp, vp]
    L.ft_search_local_points.argtypes = [vp, C.POINTER(FrameView), C.POINTER(LocalPoints), f, f, vp, ip] + [vp] * 10
    L.ft_search_last_frame.argtypes = [vp, C.POINTER(FrameView), C.POINTER(LastPoints), vp, f, i, i, i, vp, ip] + [vp] * 4
    L.ft_features_in_area.argtypes = [vp, C.POINTER(FrameView), i, vp, vp, vp, vp, vp, vp, vp, i, vp]
    L.ft_fisheye_stereo.argtypes = [vp, C.POINTER(FisheyeRig), vp, vp, i, vp, vp, i, vp, i, vp, vp, vp, ip]
    L.ft_is_in_frustum.argtypes = [vp, C.POINTER(FrameView), C.POINTER(FramePose), C.POINTER(MapPoints), f, f,
                                   C.POINTER(FrustumResult), ip]
    L.ft_tracked_frame_create.argtypes = [vp, i, i, C.POINTER(vp)]
    L.ft_tracked_frame_destroy.argtypes = [vp]
    L.ft_tracked_frame_upload.argtypes = [vp, C.POINTER(FrameView)]
    L.ft_tracked_frame_bind_stereo.argtypes = [vp, vp, i, C.POINTER(FrameView)]
    L.ft_tracked_frame_search_last_frame.argtypes = [vp, C.POINTER(LastPoints), vp, f, i, i, i, vp, ip]
    L.ft_tracked_frame_search_last_frame_se3.argtypes = [vp, C.POINTER(LastPoints), C.POINTER(SE3), C.POINTER(SE3), f, i, i, i, vp, ip]
    L.ft_search_last_frame_se3.argtypes = [vp, C.POINTER(FrameView), C.POINTER(LastPoints), C.POINTER(SE3), C.POINTER(SE3), f, i, i,
                                           i, vp, ip] + [vp] * 4
    L.ft_tracked_frame_track_local_map.argtypes = [vp, C.POINTER(FramePose), C.POINTER(MapPoints), f, f, f, f, i, f,
                                                   C.POINTER(FrustumResult), ip, vp, ip]
    L.ft_tracked_frame_holder_obs.argtypes = [vp, vp]
    L.ft_tracked_batch_create.argtypes = [vp, i, i, i, C.POINTER(vp)]
    L.ft_tracked_batch_destroy.argtypes = [vp]
    L.ft_tracked_batch_upload.argtypes = [vp, i, C.POINTER(FrameView)]
    L.ft_tracked_batch_search_last_frame.argtypes = [vp, i, C.POINTER(LastPoints), vp, f, vp, vp, i, C.POINTER(vp), vp]
    L.ft_tracked_batch_search_last_frame_se3.argtypes = [vp, i, C.POINTER(LastPoints), C.POINTER(SE3), C.POINTER(SE3), f, vp, vp, i,
                                                         C.POINTER(vp), vp]
    L.ft_tracked_batch_track_local_map.argtypes = [vp, i, C.POINTER(FramePose), C.POINTER(MapPoints), f, f, f, f, i, f,
                                                   C.POINTER(FrustumResult), vp, C.POINTER(vp), vp]
    L.ft_tracked_batch_submit_search_last_frame.argtypes = L.ft_tracked_batch_search_last_frame.argtypes
    L.ft_tracked_batch_submit_search_last_frame_se3.argtypes = L.ft_tracked_batch_search_last_frame_se3.argtypes
    L.ft_tracked_batch_submit_track_local_map.argtypes = L.ft_tracked_batch_track_local_map.argtypes
    L.ft_tracked_batch_wait.argtypes = [vp]
    L.ft_tracked_batch_holder_obs.argtypes = [vp, i, vp]
    L.ft_tracked_batch_bind_fisheye.argtypes = [vp, vp, vp, i, i, i, i, i, i, C.POINTER(FrameView), C.POINTER(FisheyeRig), vp, C.POINTER(vp),
                                                C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]
    L.ft_tracked_batch_bind_fisheye_slots.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, C.POINTER(FrameView), C.POINTER(FisheyeRig), vp,
                                                      C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]
    L.ft_descriptor_distance.argtypes = [vp, vp, vp, i, vp]
    L.ft_stereo_frontend_device_descriptors.argtypes = [vp, i, i, C.POINTER(vp), ip]
    L.ft_vocabulary_create.argtypes = [vp, i, i, i, i, i, vp, vp, vp, vp, C.POINTER(vp)]
    L.ft_vocabulary_load_text.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.ft_vocabulary_destroy.argtypes = [vp]
    L.ft_vocabulary_info.argtypes = [vp, ip, ip, ip, ip]
    L.ft_bow_transform.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, vp, i, ip, vp, vp, vp, i, ip]
    L.ft_search_by_bow.argtypes = [vp, C.POINTER(BowSide), vp, C.POINTER(BowSide), i, f, i, vp, ip]
    L.ft_selftest_libm.argtypes = [vp, i, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong),
                                   C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint32)]
    L.ft_octree_distribute.argtypes = [vp, i, i, i, i, i, i, vp, i, ip]
    L.ft_level_geometry.argtypes = [i, i, i, f, i, vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(status: int) -> None:
    if status != FT_OK:
        raise FastTrackError(status, lib().ft_last_error().decode(errors="replace"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
