"""The C++ mirror (include/fasttrack_amd.hpp) must compile with plain g++ against the C ABI, link against the
shared library, and - without a GPU - fail loudly through fasttrack::Error."""
import os
import subprocess

from fasttrack_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mirror_header_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    exe = str(tmp_path / "demo")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cpp_mirror_demo.cpp"), "-L", os.path.join(ROOT, "fasttrack_amd"),
                           "-lfasttrack_amd", "-Wl,-rpath," + os.path.join(ROOT, "fasttrack_amd"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if _capi.lib().ft_device_count() == 0:
        assert r.returncode == 1 and "no CPU fallback" in r.stderr
    else:
        assert r.returncode == 0 and "keypoints" in r.stdout
