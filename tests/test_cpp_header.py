"""The C++ mirror (include/fasttrack_amd.hpp, the adapter INTEGRATION.md tells a maintainer to use) must compile with plain
g++ against the C ABI, link against the shared library, fail loudly through fasttrack::Error without a GPU - and, on the GPU,
give the oracle's results through the reference's own call shapes (ORBextractor::operator(), KernelController::launch*,
reference include/ORBextractor.h:105-146, include/Kernels/KernelController.h:31-46)."""
import os
import subprocess

import numpy as np
import pytest

from fasttrack_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "demo")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cpp_mirror_demo.cpp"), "-L", os.path.join(ROOT, "fasttrack_amd"),
                           "-lfasttrack_amd", "-Wl,-rpath," + os.path.join(ROOT, "fasttrack_amd"), "-o", exe])
    return exe


def test_mirror_header_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if _capi.lib().ft_device_count() == 0:
        assert r.returncode == 1 and "no CPU fallback" in r.stderr
    else:
        assert r.returncode == 0 and "keypoints" in r.stdout


def _read(buf, off, dtype):
    n = int(np.frombuffer(buf, np.int32, 1, off)[0])
    a = np.frombuffer(buf, dtype, n, off + 4)
    return a, off + 4 + n * np.dtype(dtype).itemsize


@pytest.mark.gpu
def test_mirror_header_runs_the_kernels_and_equals_the_oracle(tmp_path):
    """examples/cpp_mirror_demo.cpp on a seeded stereo pair written to files: keypoints, descriptors, mvuRight, mvDepth, the
    (SAD, index) pairs of launchStereoMatchKernel and the assignment of launchSearchLocalPointsKernel, read back from the file
    the program writes, equal the oracle's bit for bit"""
    from fasttrack_amd import scenarios as scn, synth
    from oracle import binding as ob
    exe = _build(tmp_path)
    w, h, nf = 752, 480, 1200
    L, R = synth.make_planes_pair(w, h, seed=31)
    intr = synth.intrinsics(w, h)
    exL, exR = ob.Extractor(nf), ob.Extractor(nf)
    kL, dL, monoL = exL.extract(L)
    kR, dR, _ = exR.extract(R)
    sm = ob.stereo_match(exL, exR, kL, kR, dL, dR, intr["mbf"], intr["mb"])
    sf, _ = ob.scale_factors(1.2, 8)
    pts = scn.local_points_scenario(kL, dL, sf, w, h, seed=9, M=1500, uright=sm["uright"], mbf=float(intr["mbf"]))
    oF = ob.FrameView(keys=kL, descriptors=dL, scale_factors_=sf, bounds=scn.frame_bounds(w, h), mbf=intr["mbf"], mb=intr["mb"],
                      uright=sm["uright"])
    o = ob.search_local_points(oF, pts, 3.0)
    L.tofile(tmp_path / "l.raw")
    R.tofile(tmp_path / "r.raw")
    with open(tmp_path / "pts.bin", "wb") as f:
        f.write(np.int32(len(pts["skip"])).tobytes())
        for k, dt in (("skip", np.uint8), ("in_view", np.uint8), ("in_view_r", np.uint8), ("level", np.int32), ("level_r", np.int32),
                      ("view_cos", np.float32), ("view_cos_r", np.float32), ("proj_x", np.float32), ("proj_y", np.float32),
                      ("proj_xr", np.float32), ("proj_yr", np.float32), ("descriptors", np.uint8), ("observations", np.int32)):
            f.write(np.ascontiguousarray(pts[k], dt).tobytes())
    r = subprocess.run([exe, str(w), str(h), str(nf), str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), repr(float(intr["mbf"])),
                        repr(float(intr["mb"])), str(tmp_path / "out.bin"), str(tmp_path / "pts.bin"), "3.0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    buf = open(tmp_path / "out.bin", "rb").read()
    off = 0
    gkL, off = _read(buf, off, ob.KP_DTYPE)
    gdL, off = _read(buf, off, np.uint8)
    gkR, off = _read(buf, off, ob.KP_DTYPE)
    gdR, off = _read(buf, off, np.uint8)
    gur, off = _read(buf, off, np.float32)
    gdp, off = _read(buf, off, np.float32)
    gsi, off = _read(buf, off, np.int32)
    nm, off = _read(buf, off, np.int32)
    assign, off = _read(buf, off, np.int32)
    holder, off = _read(buf, off, np.int32)
    bd, off = _read(buf, off, np.int32)
    bi, off = _read(buf, off, np.int32)
    assert off == len(buf)
    assert np.array_equal(gkL, kL) and np.array_equal(gkR, kR)
    assert np.array_equal(gdL.reshape(-1, 32), dL) and np.array_equal(gdR.reshape(-1, 32), dR)
    assert np.array_equal(gur, sm["uright"]) and np.array_equal(gdp, sm["depth"]) and sm["n"] > 0.3 * len(kL)
    idx = gsi.reshape(-1, 2)[:, 1]
    assert set(np.nonzero(sm["uright"] >= 0)[0]) <= set(idx.tolist()) and len(idx) == len(set(idx.tolist())), "vDistIdx lists the matched left keypoints"
    assert int(nm[0]) == o["n"] > 100 and np.array_equal(assign, o["assign"])
    assert np.array_equal(bd, o["best_dist"]) and np.array_equal(bi, o["best_idx"])
    assert np.array_equal(holder, oF.holder_obs)
    assert f"{len(kL)} + {len(kR)} keypoints (mono index {monoL}" in r.stdout
