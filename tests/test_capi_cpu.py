"""C-ABI library checks that need no GPU: it loads, exports every declared symbol, fails loudly without
a device, and its host-only stages (octree, level geometry) equal the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from fasttrack_amd import _capi
from oracle import binding as ob


def test_library_exports_every_declared_symbol():
    L = _capi.lib()
    declared = _capi.declared_symbols()
    assert len(declared) >= 38
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    # and nothing internal leaks: exported ft_* symbols are exactly the declared ones
    out = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T ft_" in l)
    assert exported == declared
    assert L.ft_version().decode().startswith("fasttrack_amd")


def test_keypoint_layout_is_cv_keypoint():
    assert _capi.KP_DTYPE.itemsize == 28
    assert [_capi.KP_DTYPE.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_no_device_fails_loudly():
    """No CPU fallback: on a box without a GPU every compute entry point is unreachable because no
    context can be created."""
    L = _capi.lib()
    if L.ft_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    st = L.ft_context_create(0, 0, C.byref(h))
    assert st == _capi.FT_ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in L.ft_last_error()
    from fasttrack_amd import orb
    with pytest.raises(_capi.FastTrackError):
        orb.Context(0)


def test_missing_library_is_an_import_error(tmp_path, monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _capi.lib()


def _octree(xys, minX, maxX, minY, maxY, N):
    L = _capi.lib()
    xys = np.ascontiguousarray(xys, np.int32)
    out = np.zeros(len(xys) + 8, np.int32)
    n = C.c_int()
    st = L.ft_octree_distribute(_capi.ptr(xys), len(xys), minX, maxX, minY, maxY, N, _capi.ptr(out), len(out), C.byref(n))
    assert st == 0, L.ft_last_error()
    return out[:n.value].copy()


def test_host_octree_equals_oracle():
    rng = np.random.default_rng(11)
    for trial in range(150):
        W, H = int(rng.integers(40, 1300)), int(rng.integers(40, 720))
        n, N = int(rng.integers(1, 6000)), int(rng.integers(1, 500))
        pts = np.unique(np.stack([rng.integers(3, W - 3, n), rng.integers(3, H - 3, n)], 1), axis=0)
        rng.shuffle(pts)
        # few distinct scores and sizes -> many (size, UL.x) ties inside std::sort
        xys = np.concatenate([pts, rng.integers(7, 12, (len(pts), 1))], 1).astype(np.int32)
        a = ob.distribute_octree(xys, 16, 16 + W, 16, 16 + H, N)
        b = _octree(xys, 16, 16 + W, 16, 16 + H, N)
        assert np.array_equal(a, b), (trial, W, H, n, N)
    # clustered candidates (deep subdivision) and an emission-ordered (cell-major) input
    cl = (rng.normal(0, 6, (3000, 2)) + [200, 100]).astype(int).clip(3, 396)
    cl = np.unique(cl, axis=0)
    xys = np.concatenate([cl, rng.integers(7, 200, (len(cl), 1))], 1).astype(np.int32)
    assert np.array_equal(ob.distribute_octree(xys, 16, 416, 16, 316, 200), _octree(xys, 16, 416, 16, 316, 200))
    ex = ob.Extractor(1000)
    from fasttrack_amd import synth
    ex.extract(synth.make_image(640, 480, 3))
    for level in (0, 3, 7):
        c = ex.candidates(level)
        lw, lh = ob.level_sizes(640, 480, 1.2, 8)
        q = ob.features_per_level(1000, 1.2, 8)[level]
        assert np.array_equal(ob.distribute_octree(c, 16, int(lw[level]) - 16, 16, int(lh[level]) - 16, int(q)),
                              _octree(c, 16, int(lw[level]) - 16, 16, int(lh[level]) - 16, int(q)))


def test_level_geometry_equals_oracle_and_survey_tables():
    L = _capi.lib()
    for (w, h, nf) in [(752, 480, 1200), (640, 480, 1000), (512, 512, 2000), (1280, 720, 2000), (333, 257, 500)]:
        arrs = [np.zeros(8, np.int32) for _ in range(7)]
        assert L.ft_level_geometry(w, h, nf, 1.2, 8, *[_capi.ptr(a) for a in arrs]) == 0
        lw, lh, quota, ncols, nrows, wcell, hcell = arrs
        ow, oh = ob.level_sizes(w, h, 1.2, 8)
        assert np.array_equal(lw, ow) and np.array_equal(lh, oh)
        assert np.array_equal(quota, ob.features_per_level(nf, 1.2, 8))
    # FAST cell grid at level 0
    for (w, h, grid) in [(752, 480, (20, 12, 36, 38)), (640, 480, (17, 12, 36, 38)), (512, 512, (13, 13, 37, 37)),
                         (1280, 720, (35, 19, 36, 37))]:
        arrs = [np.zeros(8, np.int32) for _ in range(7)]
        L.ft_level_geometry(w, h, 1000, 1.2, 8, *[_capi.ptr(a) for a in arrs])
        assert (arrs[3][0], arrs[4][0], arrs[5][0], arrs[6][0]) == grid


def test_bad_arguments_are_rejected():
    L = _capi.lib()
    n = C.c_int()
    xy = np.array([[5000, 1, 9]], np.int32)
    assert L.ft_octree_distribute(_capi.ptr(xy), 1, 16, 100, 16, 100, 5, None, 0, C.byref(n)) == _capi.FT_ERR_INVALID
    assert L.ft_level_geometry(640, 480, 1000, 1.0, 8, *[None] * 7) == _capi.FT_ERR_INVALID
    assert L.ft_extractor_create(None, 1000, 1.2, 8, 20, 7, 640, 480, 1, None) == _capi.FT_ERR_INVALID


def test_std_sort_replay_matches_libstdcxx(tmp_path):
    """octree_paths.h replays libstdc++'s introsort move by move (the device octree cannot call std::sort)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tsr")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(root, "tests", "cpp", "test_sort_replay.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("mode", ["1", "2", "3"])
def test_path_code_octree_equals_oracle(mode):
    """The path-code formulations the device kernel is built from (octree_paths.h; 1 = node-list replay over
    sorted path codes, 2 = the round formulation k_octree executes: array list + prefix sums, 3 = the same rounds over a
    histogram of the candidates per tree node instead of sorted keys, as k_octree_hist runs them - with up to 20 000
    candidates and clustered sets; a level that formulation gives up on goes to the sorted rounds, as on the device), executed
    single-threaded on the host through ft_octree_distribute with FT_DEBUG_OCTREE_PATHS=<mode>, equal the oracle -
    in a fresh process because the switch is read once."""
    code = r'''
import numpy as np, ctypes as C, sys, os
sys.path.insert(0, %r)
from fasttrack_amd import _capi
from oracle import binding as ob
L = _capi.lib()
def octree(xys, a, b, c, d, N):
    xys = np.ascontiguousarray(xys, np.int32); out = np.zeros(len(xys) + 64, np.int32); n = C.c_int()
    assert L.ft_octree_distribute(_capi.ptr(xys), len(xys), a, b, c, d, N, _capi.ptr(out), len(out), C.byref(n)) == 0
    return out[:n.value].copy()
rng = np.random.default_rng(5)
MODE = os.environ["FT_DEBUG_OCTREE_PATHS"]
for trial in range(300):
    W, H = int(rng.integers(40, 1300)), int(rng.integers(40, 720))
    if trial %% 7 == 0: W, H = int(rng.integers(300, 2000)), int(rng.integers(20, 60))  # many root nodes
    n, N = int(rng.integers(1, 20000 if MODE == "3" else 5000)), int(rng.integers(1, 500))
    if MODE == "3" and trial %% 3 == 0:
        cx, cy = rng.integers(3, W - 3), rng.integers(3, H - 3)
        pts = np.stack([np.clip(rng.normal(cx, W / 16, n), 3, W - 4).astype(int), np.clip(rng.normal(cy, H / 16, n), 3, H - 4).astype(int)], 1)
    else:
        pts = np.stack([rng.integers(3, W - 3, n), rng.integers(3, H - 3, n)], 1)
    pts = np.unique(pts, axis=0); rng.shuffle(pts)
    xys = np.concatenate([pts, rng.integers(7, 12, (len(pts), 1))], 1).astype(np.int32)
    assert np.array_equal(ob.distribute_octree(xys, 16, 16 + W, 16, 16 + H, N), octree(xys, 16, 16 + W, 16, 16 + H, N)), trial
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FT_DEBUG_OCTREE_PATHS=mode)
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_malformed_lane_map_is_an_error_not_a_different_table():
    """FT_LANE_MAP is checked before anything else in ft_context_create: a typo, an out-of-range lane or an incomplete set is
    FT_ERR_INVALID with a message (round 2 dropped the offending entries silently and ran on a shifted table); a well-formed
    value gets as far as the device probe."""
    code = r'''
import os, sys, ctypes as C
sys.path.insert(0, %r)
from fasttrack_amd import _capi
L = _capi.lib()
h = C.c_void_p()
for bad in ("1 2 x 4", "1 2 3", "1 2 3 99", "own 1"):
    os.environ["FT_LANE_MAP"] = bad
    assert L.ft_context_create(0, 0, C.byref(h)) == _capi.FT_ERR_INVALID, bad
    assert b"FT_LANE_MAP" in L.ft_last_error(), bad
for good in ("own", "0 1 2 3", "1,2,3,4, 5 1 3 1", "", "  "):  # exported but empty = unset (how a shell neutralises a variable)
    os.environ["FT_LANE_MAP"] = good
    rc = L.ft_context_create(0, 0, C.byref(h))
    assert rc in (0, _capi.FT_ERR_NO_DEVICE), (good, rc)
    if rc == 0: L.ft_context_destroy(h)
assert os.environ.get("GPU_MAX_HW_QUEUES") == "10"  # exported by the Python driver, not by the library
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_option_out_of_range_in_the_environment_is_an_error():
    """FT_<NAME> outside the option's range: ft_context_create reports FT_ERR_INVALID (before it looks for a device)"""
    code = ("from fasttrack_amd import _capi; import ctypes as C\n"
            "h = C.c_void_p(); rc = _capi.lib().ft_context_create(0, 1, C.byref(h))\n"
            "print('rc', rc, _capi.lib().ft_last_error().decode())")
    env = dict(os.environ, FT_PASS_BURST="99")
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert f"rc {_capi_const('FT_ERR_INVALID')}" in out.stdout and "FT_PASS_BURST=99 is outside [2, 14]" in out.stdout, out.stdout + out.stderr[-2000:]


@pytest.mark.parametrize("bad", ["abc", "1x", "", " 7 7"])
def test_option_text_in_the_environment_must_be_an_integer(bad):
    """FT_GRAPH=abc / =1x: FT_ERR_INVALID, not atoi's 0 or 1; an empty value is "unset" """
    code = ("from fasttrack_amd import _capi; import ctypes as C\n"
            "h = C.c_void_p(); rc = _capi.lib().ft_context_create(0, 1, C.byref(h))\n"
            "print('rc', rc, _capi.lib().ft_last_error().decode())")
    env = dict(os.environ, FT_GRAPH=bad)
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if bad == "":
        assert "is not an integer" not in out.stdout
    else:
        assert f"rc {_capi_const('FT_ERR_INVALID')}" in out.stdout and "is not an integer" in out.stdout, out.stdout + out.stderr[-2000:]


def test_library_is_not_built_with_threadgroup_split():
    """k_resolve_batch orders a chunk's atomics before the next chunk's loads through ONE CU's in-order vector-memory path
    (kernels_search.hip): the kernel descriptors the Makefile's flags produce must say tg_split 0, and the Makefile refuses
    -mtgsplit"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "fasttrack_amd", "csrc")
    out = subprocess.run(["make", "-C", csrc, "check-tgsplit"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "tg_split 0" in out.stdout, out.stdout + out.stderr[-2000:]
    bad = subprocess.run(["make", "-C", csrc, "-n", "CXXFLAGS=-O3 -mtgsplit"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "tgsplit" in bad.stderr


def _capi_const(name):
    from fasttrack_amd import _capi
    return getattr(_capi, name)


def test_option_table_matches_the_header_documentation():
    """every tuning option of csrc/ft_host.h (FT_TUNING_OPTIONS, enumerated through ft_option_describe) is documented in
    include/fasttrack_amd.h with its default; its environment spelling is FT_ + upper-case name; and the library has ONE
    getenv (context.cpp)"""
    import re
    from fasttrack_amd import _capi, orb
    table = orb.Context.option_table()
    assert 8 <= len(table) <= 13 and len({t[0] for t in table}) == len(table)  # (variants that lost twice are deleted, not switched off; round 6 added blocking_sync: a host policy, not a variant)
    header = open(_capi.HEADER_PATH).read()
    for name, env, default, doc in table:
        assert env == "FT_" + name.upper() and doc
        m = re.search(r"^ \*   %s\s+(-?\d+)\s+(-?\d+) \.\. (-?\d+)\s" % re.escape(name), header, re.M)
        assert m, f"option {name} is not documented in fasttrack_amd.h"
        lo, hi = orb.Context.option_range(name)
        assert (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (default, lo, hi), (name, default, lo, hi)
        assert lo <= default <= hi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "fasttrack_amd", "csrc")
    n = sum(len(re.findall(r"\bgetenv\s*\(", open(os.path.join(csrc, f)).read())) for f in os.listdir(csrc)
            if f.endswith((".cpp", ".hip", ".h")))
    assert n == 1, n
