"""Seeded synthetic scenarios shared by the oracle tests (CPU) and the parity tests (GPU)."""
import numpy as np

from fasttrack_amd import synth
from fasttrack_amd.scenarios import (KB8_CAM, bow_match_scenario, fisheye_rig_scenario, frame_bounds, kb8_project64, last_frame_scenario,  # noqa: F401
                                     local_points_from_frustum, local_points_scenario, map_points_scenario, random_pose,
                                     random_se3)
from oracle import binding as ob

KP = ob.KP_DTYPE


def oracle_stereo_frame(width, height, nfeatures, seed):
    """Extract a synthetic pair with the oracle; returns dict with images, extractors, keys, descriptors."""
    L, R = synth.make_stereo_pair(width, height, seed)
    exL, exR = ob.Extractor(nfeatures), ob.Extractor(nfeatures)
    kL, dL, _ = exL.extract(L)
    kR, dR, _ = exR.extract(R)
    intr = synth.intrinsics(width, height)
    return dict(L=L, R=R, exL=exL, exR=exR, kL=kL, dL=dL, kR=kR, dR=dR, intr=intr)


def fisheye_frame_scenario(width, height, nfeatures, seed):
    """Two-camera (Nleft != -1) frame: left/right keypoints from the oracle on a synthetic pair, a
    brute-force left<->right match table, and local points visible in both cameras."""
    fr = oracle_stereo_frame(width, height, nfeatures, seed)
    m = ob.fisheye_match(fr["dL"], fr["dR"])
    l2r = m["matches"].astype(np.int32)
    r2l = np.full(len(fr["kR"]), -1, np.int32)
    for i, j in enumerate(l2r):
        if j >= 0:
            r2l[j] = i
    fr["l2r"], fr["r2l"] = l2r, r2l
    return fr



def two_camera_points(fr, sf, seed, M=1200, zero_obs_frac=0.2):
    rng = np.random.default_rng(seed)
    kL, kR, dL = fr["kL"], fr["kR"], fr["dL"]
    NL, NR = len(kL), len(kR)
    srcL = rng.integers(0, max(NL // 4, 1), M)
    srcR = rng.integers(0, max(NR // 4, 1), M)
    d = dL[srcL].copy()
    flips = rng.integers(0, 256, (M, 10))
    for k in range(10):
        mm = rng.random(M) < 0.5
        d[mm, flips[mm, k] // 8] ^= (1 << (flips[mm, k] % 8)).astype(np.uint8)
    both = rng.random(M)
    return dict(skip=(rng.random(M) < 0.04).astype(np.uint8), in_view=(both < 0.8).astype(np.uint8),
                in_view_r=(both > 0.3).astype(np.uint8),
                level=np.clip(kL["octave"][srcL] + rng.integers(0, 2, M), 0, len(sf) - 1).astype(np.int32),
                level_r=np.where(rng.random(M) < 0.1, -1,
                                 np.clip(kR["octave"][srcR] + rng.integers(0, 2, M), 0, len(sf) - 1)).astype(np.int32),
                view_cos=np.where(rng.random(M) < 0.5, 0.9995, 0.9).astype(np.float32),
                view_cos_r=np.where(rng.random(M) < 0.5, 0.9995, 0.9).astype(np.float32),
                proj_x=(kL["x"][srcL] + rng.normal(0, 2, M)).astype(np.float32),
                proj_y=(kL["y"][srcL] + rng.normal(0, 2, M)).astype(np.float32),
                proj_xr=(kR["x"][srcR] + rng.normal(0, 2, M)).astype(np.float32),
                proj_yr=(kR["y"][srcR] + rng.normal(0, 2, M)).astype(np.float32),
                descriptors=d,
                observations=np.where(rng.random(M) < zero_obs_frac, 0, rng.integers(1, 6, M)).astype(np.int32))
