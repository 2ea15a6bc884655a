"""The host-libm seam of the path (CPU only): the reference evaluates the rBRIEF rotation with std::cos(float) /
std::sin(float) (/root/reference/src/ORBextractor.cc:34,73-74) and PredictScale with std::log(float)
(/root/reference/src/MapPoint.cc:539), i.e. glibc's cosf / sinf / logf, which are not correctly rounded; KannalaBrandt8::
project / unproject add atan2f and tanf (/root/reference/src/CameraModels/KannalaBrandt8.cpp:67-84,114-143; atanf checked on
all 2^32 floats, atan2f on 2^31 pairs, tanf on every float of (-120, 120)).
  * fasttrack_amd/csrc/libm_f32.h (what the kernels evaluate) equals this host's libm on EVERY float the path can
    produce (tests/cpp/test_libm_f32.cpp, exhaustive, both the FMA-contracted and the plain build of glibc's source);
  * this host's libm and the oracle reproduce tests/golden/libm_rotation_glibc235.npz: 48 angles at which cosf / sinf
    make computeOrbDescriptor sample a different pixel than a correctly rounded cos / sin would."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.parametrize("contract", [1, 0])
def test_device_restatement_equals_host_libm_on_every_float(tmp_path, contract):
    src = os.path.join(ROOT, "tests", "cpp", "test_libm_f32.cpp")
    exe = str(tmp_path / f"tl{contract}")
    flags = ["-mfma"] if contract and _has_fma() else []
    subprocess.check_call(["g++", "-O2", "-std=c++20", "-ffp-contract=off", f"-DFT_LIBM_CONTRACT={contract}", *flags, src,
                           "-o", exe, "-lpthread"])
    # exhaustive with a hardware FMA (about 8 s on 8 threads); software fma is ~50x slower: strided then
    # the plain build is not the variant an FMA host selects: every 7th float there
    stride = "1" if flags else ("7" if not contract else "101")
    out = subprocess.run([exe, stride, str(os.cpu_count() or 1)], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    for fn in ("cosf", "sinf", "cosf(neg)", "sinf(neg)", "logf(0,1e4]", "logf(1e4,max]", "atanf(all)", "atan2f(sample)", "tanf(+120)",
               "tanf(-120)"):
        assert f"{fn}: checked" in out.stdout
    assert out.stdout.count("mismatches 0") == 10, out.stdout


def test_host_libm_and_oracle_reproduce_the_glibc_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "libm_rotation_glibc235.npz"))
    m = ctypes.CDLL("libm.so.6")
    for f in (m.cosf, m.sinf, m.logf):
        f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
    deg = g["angle_deg"]
    rad = (deg * np.float32(np.pi / np.float32(180.0))).astype(np.float32)
    cs = np.array([m.cosf(float(r)) for r in rad], np.float32).view(np.uint32)
    sn = np.array([m.sinf(float(r)) for r in rad], np.float32).view(np.uint32)
    assert np.array_equal(cs, g["cosf_bits"]) and np.array_equal(sn, g["sinf_bits"])
    # every one of these angles is a case where the correctly rounded value differs
    cr = np.cos(rad.astype(np.float64)).astype(np.float32).view(np.uint32)
    sr = np.sin(rad.astype(np.float64)).astype(np.float32).view(np.uint32)
    assert np.all((cr != cs) | (sr != sn))
    lg = np.array([m.logf(float(r)) for r in g["ratio"]], np.float32).view(np.uint32)
    assert np.array_equal(lg, g["logf_bits"])
    blurred = ob.gaussian_blur7(g["image"])
    L = ob.lib()
    for i, d in enumerate(deg):
        desc = np.zeros(32, np.uint8)
        L.orc_brief_descriptor(blurred.ctypes.data, blurred.strides[0], ctypes.c_float(float(g["x"])),
                               ctypes.c_float(float(g["y"])), ctypes.c_float(float(d)), desc.ctypes.data)
        assert np.array_equal(desc, g["descriptors"][i]), i


def test_host_libm_reproduces_the_kb8_vectors(golden_dir):
    """tests/golden/libm_kb8_glibc235.npz: arguments at which glibc's atan2f / atanf / tanf are not the correctly rounded
    values (what KannalaBrandt8::project / unproject evaluate on the host)"""
    g = np.load(os.path.join(golden_dir, "libm_kb8_glibc235.npz"))
    m = ctypes.CDLL("libm.so.6")
    m.atan2f.restype, m.atan2f.argtypes = ctypes.c_float, [ctypes.c_float, ctypes.c_float]
    for f in (m.tanf, m.atanf):
        f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
    a2 = np.array([m.atan2f(float(y), float(x)) for y, x in zip(g["atan2_y"], g["atan2_x"])], np.float32)
    tn = np.array([m.tanf(float(x)) for x in g["tan_x"]], np.float32)
    at = np.array([m.atanf(float(x)) for x in g["atan_x"]], np.float32)
    assert np.array_equal(a2.view(np.uint32), g["atan2f_bits"])
    assert np.array_equal(tn.view(np.uint32), g["tanf_bits"])
    assert np.array_equal(at.view(np.uint32), g["atanf_bits"])
    # and every one of them differs from the narrowed double
    assert np.all(a2 != np.arctan2(g["atan2_y"].astype(np.float64), g["atan2_x"].astype(np.float64)).astype(np.float32))
    assert np.all(tn != np.tan(g["tan_x"].astype(np.float64)).astype(np.float32))
    assert np.all(at != np.arctan(g["atan_x"].astype(np.float64)).astype(np.float32))

