"""fasttrack_amd - MI355X-native ORB tracking front end (extract + stereo / projection matchers).

The product is libfasttrack_amd.so (hand-written HIP for gfx950 behind the C ABI of
include/fasttrack_amd.h); this package is the thin ctypes driver used by tests and bench.py.
Importing the package does not load the library; the first call does, and fails loudly if it is missing.
"""
from ._capi import FastTrackError, KP_DTYPE  # noqa: F401
