#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small input/expected-output vectors for the hot path.

The reference (C++/OpenCV/CUDA) cannot be built or imported here, so these vectors are
produced by the repo's own oracle (oracle/) from seeded synthetic frames; they pin the oracle against
silent drift and give the GPU tests a fixed target that does not depend on the oracle being rebuilt.
Each file holds only data: inputs (images, parameters) and expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fasttrack_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402
from tests import scenarios as sc  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def extract_case(name, w, h, nf, levels, seed, lap=(0, 0)):
    img = synth.make_image(w, h, seed)
    ex = ob.Extractor(nf, 1.2, levels, 20, 7)
    k, d, nm = ex.extract(img, lap)
    cands = [ex.candidates(l) for l in range(levels)]
    lv = [ex.level(l) for l in range(levels)]
    np.savez_compressed(os.path.join(G, name), image=img, nfeatures=nf, nlevels=levels, lap=np.array(lap),
                        keypoints=k, descriptors=d, n_mono=nm,
                        cand_counts=np.array([len(c) for c in cands]),
                        cand_level0=cands[0], cand_last=cands[-1],
                        level1=lv[1], level_last=lv[-1],
                        blurred_level1_crc=np.array([int(ex.blurred(1).astype(np.uint64).sum())]))


def stereo_case(name, w, h, nf, seed):
    fr = sc.oracle_stereo_frame(w, h, nf, seed)
    mbf, mb = fr["intr"]["mbf"], fr["intr"]["mb"]
    sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], mbf, mb)
    sm0 = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], mbf, mb, median_cut=False)
    fm = ob.fisheye_match(fr["dL"], fr["dR"])
    np.savez_compressed(os.path.join(G, name), left=fr["L"], right=fr["R"], nfeatures=nf, mbf=mbf, mb=mb,
                        keysL=fr["kL"], keysR=fr["kR"], descL=fr["dL"], descR=fr["dR"],
                        uright=sm["uright"], depth=sm["depth"], n=sm["n"], sad_nocut=sm0["sad"],
                        hamming_idx=sm0["hamming_idx"], fisheye_matches=fm["matches"], fisheye_best=fm["best"],
                        fisheye_second=fm["second"])


def search_case(name, w, h, nf, seed):
    fr = sc.oracle_stereo_frame(w, h, nf, seed)
    sf, _ = ob.scale_factors(1.2, 8)
    mbf, mb = fr["intr"]["mbf"], fr["intr"]["mb"]
    sm = ob.stereo_match(fr["exL"], fr["exR"], fr["kL"], fr["kR"], fr["dL"], fr["dR"], mbf, mb)
    pts = sc.local_points_scenario(fr["kL"], fr["dL"], sf, w, h, seed=3, M=600, uright=sm["uright"], dense=True)
    F = ob.FrameView(keys=fr["kL"], descriptors=fr["dL"], scale_factors_=sf, bounds=sc.frame_bounds(w, h), mbf=mbf,
                     mb=mb, uright=sm["uright"])
    lo = ob.search_local_points(F, pts, 3.0)
    last, Tcw = sc.last_frame_scenario(fr["kL"], fr["dL"], sm["uright"], sm["depth"], fr["intr"], w, h, seed=4)
    F2 = ob.FrameView(keys=fr["kL"], descriptors=fr["dL"], scale_factors_=sf, bounds=sc.frame_bounds(w, h), mbf=mbf,
                      mb=mb, uright=sm["uright"], cam=[fr["intr"][k] for k in ("fx", "fy", "cx", "cy")])
    la = ob.search_last_frame(F2, last, Tcw, 7.0, False, False, True)
    np.savez_compressed(os.path.join(G, name), width=w, height=h, keys=fr["kL"], descriptors=fr["dL"], sf=sf,
                        uright=sm["uright"], mbf=mbf, mb=mb, cam=np.array([fr["intr"][k] for k in ("fx", "fy", "cx", "cy")], np.float32),
                        **{"lp_" + k: v for k, v in pts.items()}, lp_th=3.0, lp_assign=lo["assign"], lp_n=lo["n"],
                        lp_best_dist=lo["best_dist"], lp_best_dist2=lo["best_dist2"], lp_best_idx=lo["best_idx"],
                        **{"lf_" + k: v for k, v in last.items()}, lf_Tcw=Tcw, lf_th=7.0, lf_assign=la["assign"],
                        lf_n=la["n"], lf_best_dist=la["best_dist"], lf_best_idx=la["best_idx"])


def bow_match_case(name, seed=6):
    """ORBmatcher::SearchByBoW on a seeded keyframe / frame pair (both FeatureVectors in CSR form, one- and two-camera
    frames, with and without the rotation histogram): the oracle's assignments"""
    voc = synth.make_vocabulary(8, 4, seed=seed)
    ov = ob.Vocabulary(8, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    out = {}
    for tag, two in (("mono", False), ("two", True)):
        S = sc.bow_match_scenario(voc, ov.transform, 400, 500, seed, two_cam=two, levelsup=2)
        for side in ("kf", "f"):
            for k, v in S[side].items():
                out[f"{tag}_{side}_{k}"] = v
        out[f"{tag}_has_point"], out[f"{tag}_nleft"] = S["has_point"], S["nleft"]
        for ori in (0, 1):
            r = ob.search_by_bow(S["kf"], S["has_point"], S["f"], S["nleft"], 0.7, bool(ori))
            out[f"{tag}_matches_ori{ori}"], out[f"{tag}_n_ori{ori}"] = r["matches"], r["n"]
    np.savez_compressed(os.path.join(G, name), **out)


def libm_case(name, seed=1):
    """The rBRIEF rotation binds to libm's cosf / sinf, PredictScale to logf (glibc 2.35 in this image).  48 angles at
    which cosf / sinf (not correctly rounded) make computeOrbDescriptor sample a different pixel than the narrowed double
    cos / sin would (found by tests/tools/find_libm_angles.cpp), with the descriptors the oracle computes there on a
    seeded image, the cosf / sinf bits themselves, and logf on a sweep of PredictScale ratios."""
    import ctypes
    import subprocess
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(), "fla")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "tools", "find_libm_angles.cpp"), "-o", exe])
    bits = [int(x, 16) for x in subprocess.run([exe, "48", str(seed)], capture_output=True, text=True, check=True).stdout.split()]
    deg = np.array(bits, np.uint32).view(np.float32)
    m = ctypes.CDLL("libm.so.6")
    for f in (m.cosf, m.sinf, m.logf):
        f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
    factor = np.float32(np.pi / np.float32(180.0))  # (float)(CV_PI / 180.f), ORBextractor.cc:68
    rad = (deg * factor).astype(np.float32)
    cs = np.array([m.cosf(float(r)) for r in rad], np.float32)
    sn = np.array([m.sinf(float(r)) for r in rad], np.float32)
    img = synth.make_image(96, 96, 11)
    blurred = ob.gaussian_blur7(img)
    L = ob.lib()
    desc = np.zeros((len(deg), 32), np.uint8)
    for i, d in enumerate(deg):
        L.orc_brief_descriptor(blurred.ctypes.data, blurred.strides[0], ctypes.c_float(48.0), ctypes.c_float(47.0),
                               ctypes.c_float(float(d)), desc[i].ctypes.data)
    rng = np.random.default_rng(seed)
    ratio = np.concatenate([np.float32(1.2) ** np.arange(-3, 12, dtype=np.float32),
                            np.exp(rng.uniform(np.log(1e-3), np.log(1e4), 4000)).astype(np.float32)]).astype(np.float32)
    lg = np.array([m.logf(float(r)) for r in ratio], np.float32)
    np.savez_compressed(os.path.join(G, name), image=img, x=48.0, y=47.0, angle_deg=deg, cosf_bits=cs.view(np.uint32),
                        sinf_bits=sn.view(np.uint32), descriptors=desc, ratio=ratio, logf_bits=lg.view(np.uint32))


def pair_x(ybits):
    """kernels_selftest.hip pair_x: the x that ft_selftest_libm(func 4) pairs with the y of these bits"""
    h = (ybits.astype(np.uint64) * 0x9E3779B1) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x85EBCA77) & 0xFFFFFFFF
    h ^= h >> 13
    bits = (h & 0x807FFFFF) | ((118 + ((h >> 24) & 15)) << 23)
    return bits.astype(np.uint32).view(np.float32)


def kb8_libm_case(name, seed=2, n=256):
    """KannalaBrandt8::project / unproject bind to libm's atan2f and tanf (fdlibm's float routines in glibc 2.35, within
    1 ulp but not correctly rounded): n arguments each at which glibc's result is NOT the narrowed double value - the
    arguments where evaluating in double on the device (rounds 2-3) gave a different float than the host."""
    import ctypes
    m = ctypes.CDLL("libm.so.6")
    m.atan2f.restype, m.atan2f.argtypes = ctypes.c_float, [ctypes.c_float, ctypes.c_float]
    m.tanf.restype, m.tanf.argtypes = ctypes.c_float, [ctypes.c_float]
    m.atanf.restype, m.atanf.argtypes = ctypes.c_float, [ctypes.c_float]
    rng = np.random.default_rng(seed)
    ys, a2 = [], []
    while len(ys) < n:
        y = np.float32(rng.uniform(-4, 4))
        x = pair_x(np.array([y], np.float32).view(np.uint32))[0]
        r = np.float32(m.atan2f(float(y), float(x)))
        if r != np.float32(np.arctan2(np.float64(y), np.float64(x))):
            ys.append(y); a2.append(r)
    ts, tn = [], []
    while len(ts) < n:
        t = np.float32(rng.uniform(0, np.pi / 2))
        r = np.float32(m.tanf(float(t)))
        if r != np.float32(np.tan(np.float64(t))):
            ts.append(t); tn.append(r)
    xs, at = [], []
    while len(xs) < n:
        x = np.float32(np.exp(rng.uniform(np.log(1e-3), np.log(1e3)))) * np.float32(rng.choice([-1, 1]))
        r = np.float32(m.atanf(float(x)))
        if r != np.float32(np.arctan(np.float64(x))):
            xs.append(x); at.append(r)
    ys = np.array(ys, np.float32)
    np.savez_compressed(os.path.join(G, name), atan2_y=ys, atan2_x=pair_x(ys.view(np.uint32)),
                        atan2f_bits=np.array(a2, np.float32).view(np.uint32), tan_x=np.array(ts, np.float32),
                        tanf_bits=np.array(tn, np.float32).view(np.uint32), atan_x=np.array(xs, np.float32),
                        atanf_bits=np.array(at, np.float32).view(np.uint32))


if __name__ == "__main__":
    libm_case("libm_rotation_glibc235.npz")
    kb8_libm_case("libm_kb8_glibc235.npz")
    bow_match_case("bow_match_s6.npz")
    extract_case("extract_160x120_s1.npz", 160, 120, 300, 4, 1)
    extract_case("extract_320x240_s2_lap.npz", 320, 240, 500, 8, 2, lap=(100, 200))
    stereo_case("stereo_320x240_s3.npz", 320, 240, 500, 3)
    search_case("search_320x240_s5.npz", 320, 240, 500, 5)
    print({f: os.path.getsize(os.path.join(G, f)) for f in sorted(os.listdir(G))})
