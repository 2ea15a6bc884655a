"""-m gpu: the DEVICE evaluation of libm_f32.h (what k_orient_desc and k_frustum call) equals the GPU box's host libm
on every float the path can produce - through the C ABI (ft_selftest_libm)."""
import ctypes as C

import numpy as np
import pytest

from fasttrack_amd import _capi, orb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("func,first,last", [(0, 0, 0x40C90FDC), (1, 0, 0x40C90FDC), (0, 0x80000000, 0xC0C90FDC),
                                             (1, 0x80000000, 0xC0C90FDC), (2, 1, 0x461C4000), (3, 0, 0xFFFFFFFF),
                                             (4, 0x30000000, 0x4FFFFFFF), (4, 0xB0000000, 0xCFFFFFFF),
                                             (5, 0, 0x42EFFFFF), (5, 0x80000000, 0xC2EFFFFF)],
                         ids=["cosf[0,2pi]", "sinf[0,2pi]", "cosf[-2pi,0]", "sinf[-2pi,0]", "logf(0,1e4]", "atanf(all)",
                              "atan2f(y>0)", "atan2f(y<0)", "tanf[0,120)", "tanf(-120,0]"])
def test_device_libm_equals_host_libm_exhaustively(func, first, last):
    ctx = orb.Context(0)
    L = _capi.lib()
    checked, bad, first_bad = C.c_ulonglong(0), C.c_ulonglong(0), C.c_uint32(0)
    _capi.check(L.ft_selftest_libm(ctx._h, func, first, last, 1, C.byref(checked), C.byref(bad), C.byref(first_bad)))
    assert checked.value == last - first + 1
    assert bad.value == 0, f"{bad.value} mismatches, first at bits 0x{first_bad.value:08x}"


def test_device_libm_reproduces_the_glibc_vectors(golden_dir):
    """the committed glibc 2.35 vectors (angles where cosf / sinf are not the correctly rounded values), one argument
    at a time through the same entry point: a mismatch there names the GPU box's libm as a different one"""
    import os
    g = np.load(os.path.join(golden_dir, "libm_rotation_glibc235.npz"))
    ctx = orb.Context(0)
    L = _capi.lib()
    rad = (g["angle_deg"] * np.float32(np.pi / np.float32(180.0))).astype(np.float32).view(np.uint32)
    for func, args in ((0, rad), (1, rad), (2, g["ratio"].view(np.uint32)[:200])):
        for b in args:
            checked, bad = C.c_ulonglong(0), C.c_ulonglong(0)
            _capi.check(L.ft_selftest_libm(ctx._h, func, int(b), int(b), 1, C.byref(checked), C.byref(bad), None))
            assert checked.value == 1 and bad.value == 0, (func, hex(int(b)))


def test_device_libm_reproduces_the_kb8_vectors(golden_dir):
    """atan2f / atanf / tanf at arguments where glibc is not correctly rounded (tests/golden/libm_kb8_glibc235.npz; the x of
    an atan2f vector is the one ft_selftest_libm pairs with its y)"""
    import os
    g = np.load(os.path.join(golden_dir, "libm_kb8_glibc235.npz"))
    ctx = orb.Context(0)
    L = _capi.lib()
    for func, args in ((4, g["atan2_y"]), (5, g["tan_x"]), (3, g["atan_x"])):
        for b in args.view(np.uint32):
            checked, bad = C.c_ulonglong(0), C.c_ulonglong(0)
            _capi.check(L.ft_selftest_libm(ctx._h, func, int(b), int(b), 1, C.byref(checked), C.byref(bad), None))
            assert checked.value == 1 and bad.value == 0, (func, hex(int(b)))

