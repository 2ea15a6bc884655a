"""Oracle vs the constant tables the reference pins - CPU only."""
import os

import numpy as np

from oracle import binding as ob


def test_pattern_matches_reference_table(golden_dir):
    ref = np.fromfile(os.path.join(golden_dir, "bit_pattern_31.i8"), dtype=np.int8)
    assert ref.size == 1024
    assert np.array_equal(ob.pattern(), ref)
    # every rotated sample must stay inside the 19-px border: max radius < 19
    pts = ref.reshape(512, 2).astype(np.float64)
    assert np.hypot(pts[:, 0], pts[:, 1]).max() < 18.5


def test_fast_arc9_matches_reference_ctable(golden_dir):
    """bit (m&7) of byte (m>>3)-63 of the reference's c_table (src/fast.cu:24) <=> mask m has an arc of >= 9."""
    tab = np.fromfile(os.path.join(golden_dir, "fast9_ctable.u8"), dtype=np.uint8)
    assert tab.size == 8129
    L = ob.lib()
    checked = 0
    for m in range(1 << 16):
        byte = (m >> 3) - 63
        mine = L.orc_fast_mask_has_arc9(m)
        if byte < 0:
            # masks below 504 have < 9 bits: cannot hold an arc
            assert mine == 0
            continue
        ref = (int(tab[byte]) >> (m & 7)) & 1
        assert mine == ref, hex(m)
        checked += 1
    assert checked == 65536 - 504


def test_fast_ring_matches_reference(golden_dir):
    ring = np.fromfile(os.path.join(golden_dir, "fast_circle.i8"), dtype=np.int8).reshape(16, 2)
    # a single bright pixel at ring position k (and 8 neighbours on the arc) must make a corner;
    # use the ring table to place 9 contiguous bright pixels and check detection both polarities
    for start in range(16):
        img = np.full((7, 7), 100, np.uint8)
        for j in range(9):
            dx, dy = ring[(start + j) % 16]
            img[3 + dy, 3 + dx] = 200
        pts = ob.fast9_16(img, 20, True)
        assert len(pts) == 1 and tuple(pts[0][:2]) == (3, 3)
        assert pts[0][2] == 99  # score = min |v - p| - 1 over the arc
        img2 = np.full((7, 7), 100, np.uint8)
        for j in range(8):  # only 8 contiguous: not a corner
            dx, dy = ring[(start + j) % 16]
            img2[3 + dy, 3 + dx] = 200
        assert len(ob.fast9_16(img2, 20, True)) == 0


def test_umax_and_quotas_and_levels():
    assert list(ob.umax()) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert 2 * sum(2 * u + 1 for u in ob.umax()[1:]) + 31 == 749
    # SURVEY section 8 tables (computed there from the reference formulas)
    assert list(ob.features_per_level(1000, 1.2, 8)) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(ob.features_per_level(1200, 1.2, 8)) == [261, 217, 181, 151, 126, 105, 87, 72]
    assert list(ob.features_per_level(2000, 1.2, 8)) == [434, 362, 302, 251, 209, 175, 145, 122]
    lw, lh = ob.level_sizes(752, 480, 1.2, 8)
    assert list(zip(lw, lh)) == [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193),
                                 (252, 161), (210, 134)]
    for (w, h), total in [((640, 480), 950532), ((512, 512), 811960), ((1280, 720), 2853088)]:
        lw, lh = ob.level_sizes(w, h, 1.2, 8)
        assert int((lw.astype(np.int64) * lh).sum()) == total


def test_gaussian_kernel_fixed_point():
    k = ob.gaussian_kernel7()
    assert list(k) == [18, 34, 48, 56, 48, 34, 18] and k.sum() == 256


def test_cv_round_half_even():
    L = ob.lib()
    assert [L.orc_cv_round_f(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    assert L.orc_cv_round_d(2.5) == 2 and L.orc_cv_round_d(3.5) == 4


def test_matcher_constants():
    # TH_HIGH=100, TH_LOW=50 -> thOrbDist 75: a pair at Hamming 74 is refined, 75 is not (Frame.cc:840,919)
    a = np.zeros(32, np.uint8)
    b = np.zeros(32, np.uint8)
    b[:9] = 0xFF
    b[9] = 0x03
    assert ob.descriptor_distance(a, b) == 74
    assert ob.descriptor_distance(a, np.full(32, 0xFF, np.uint8)) == 256
