"""Pins the one piece of the G-PT path the reference lets anyone pin: its random number generator
and the order in which a 1-core run consumes it.

* `Random` = SFMT-19937 (src/libcore/random.cpp), restated in oracle/sfmt_random.hpp, must reproduce the `reference[]` table of the
  reference's OWN test (src/tests/test_random.cpp:436-501, Random(4321)) bit for bit -- tests/golden/sfmt_reference.json holds that
  table (data only, extracted by tests/golden/make_sfmt_golden.py).
* the other behaviours that test file checks (set / seed-from-generator, test_random.cpp:790-816; the mean, :513-523) on the restatement;
* the work order: spiral blocks (imageproc.cpp:28-78), Hilbert pixels (sfcurve.h:34-107), and the serial render mode built on them.
"""
import json
import os

import numpy as np

from gradientdomain_mitsuba_amd import scenes
from oracle import gpt_oracle as go

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sfmt_reference.json")


def test_sfmt19937_reproduces_the_reference_test_table_bit_for_bit():
    g = json.load(open(GOLD))
    ref = np.array([int(x, 16) for x in g["values_hex"]], dtype=np.uint64)
    assert g["seed"] == 4321 and ref.size == g["count"] == 192
    got = go.Random(g["seed"]).ulongs(ref.size)
    assert np.array_equal(got, ref)                                    # TestRandom::test00_validate


def test_next_float_is_the_top_52_bits_in_unit_interval():
    a, b = go.Random(4321), go.Random(4321)
    u = a.ulongs(1000)
    f = b.floats(1000)
    bits = (u >> np.uint64(12)) | np.uint64(0x3FF0000000000000)
    assert np.array_equal(f, bits.view(np.float64) - 1.0)               # random.cpp:616-626 (DOUBLE_PRECISION)
    assert (f >= 0).all() and (f < 1).all()
    c, d = go.Random(4321), go.Random(4321)
    u = c.ulongs(1000)
    s = d.floats_single(1000)
    bits32 = (((u & np.uint64(0xFFFFFFFF)) >> np.uint64(9)).astype(np.uint32)) | np.uint32(0x3F800000)
    assert np.array_equal(s, bits32.view(np.float32) - np.float32(1.0))  # random.cpp:630-640
    assert abs(go.Random().floats(100000).mean() - 0.5) < 1e-3 * 5     # test01_mean's expectation (default seed 5489)


def test_set_copies_the_state_and_seeding_from_a_generator_does_not():
    r1, r2 = go.Random(1234), go.Random(5678)                           # TestRandom::test09_set
    assert (r1.ulongs(20000) != r2.ulongs(20000)).all()
    r1.set(r2)
    assert np.array_equal(r1.ulongs(20000), r2.ulongs(20000))
    # Random::seed(Random *) == init_by_array over 312 draws of the source (random.cpp:519-524): same key by hand -> same stream
    src1, src2 = go.Random(99), go.Random(99)
    child = src1.clone()
    manual = go.Random(0)
    manual.seed_array(src2.ulongs(312))
    assert np.array_equal(child.ulongs(5000), manual.ulongs(5000))
    assert np.array_equal(src1.ulongs(100), src2.ulongs(100))           # both sources advanced by the same 312 draws
    assert (go.Random(99).clone().ulongs(2000) != go.Random(99).ulongs(2000)).all()


def test_next_uint_is_rejection_under_the_power_of_two_mask():
    r, raw = go.Random(7), go.Random(7)
    n = 10
    got = [r.uint(n) for _ in range(200)]
    stream = [int(v) & 15 for v in raw.ulongs(2000)]
    want = [v for v in stream if v < n][:200]
    assert got == want and max(got) < n


def test_spiral_blocks_cover_the_image_once_from_the_centre_outwards():
    for (w, h, bs) in ((1280, 720, 32), (100, 70, 32), (32, 32, 32), (33, 1, 32), (5, 200, 16)):
        b = go.spiral_blocks(w, h, bs)
        nbx, nby = -(-w // bs), -(-h // bs)
        assert len(b) == nbx * nby
        cover = np.zeros((h, w), np.int32)
        for (x, y, bw, bh) in b:
            assert x % bs == 0 and y % bs == 0 and 0 < bw <= bs and 0 < bh <= bs
            cover[y:y + bh, x:x + bw] += 1
        assert (cover == 1).all()
        assert (b[0][0], b[0][1]) == ((nbx // 2) * bs, (nby // 2) * bs)          # m_curBlock = numBlocks / 2
        if len(b) > 1 and nbx > nbx // 2 + 1:
            assert (b[1][0], b[1][1]) == ((nbx // 2 + 1) * bs, (nby // 2) * bs)  # first step: right
    # the 3 x 3 spiral written out: centre, right, down, left, left, up, up, right, right
    b = go.spiral_blocks(96, 96, 32)
    assert [(int(x) // 32, int(y) // 32) for (x, y, _, _) in b] == [(1, 1), (2, 1), (2, 2), (1, 2), (0, 2), (0, 1), (0, 0), (1, 0), (2, 0)]


def test_hilbert_points_visit_every_pixel_once_and_neighbours_in_turn():
    assert go.hilbert_points(2, 2).tolist() == [[0, 0], [1, 0], [1, 1], [0, 1]]          # order 1 by hand from sfcurve.h:92-103
    for (w, h) in ((32, 32), (16, 16), (32, 20), (7, 32), (1, 1), (3, 5)):
        p = go.hilbert_points(w, h).astype(np.int32)
        assert len(p) == w * h and len({(int(x), int(y)) for x, y in p}) == w * h
        assert (p[:, 0] < w).all() and (p[:, 1] < h).all() and tuple(p[0]) == (0, 0)
        if w == h and w & (w - 1) == 0 and w > 1:
            assert (np.abs(np.diff(p, axis=0)).sum(1) == 1).all()                        # a true Hilbert curve: unit steps


def test_serial_render_is_the_same_estimator_fed_by_the_reference_stream():
    """The serial mode draws from ONE SFMT stream in the reference's order; the counter mode from per-(pixel, sample) streams.  Same
    estimator: equal weights everywhere, equal expectations (checked on image means within Monte Carlo noise), deterministic, and a
    different parent seed gives a different film."""
    sc = scenes.cornell_box(48, 40, "diffuse")
    O = go.Scene(sc)
    cfg = go.config(maxDepth=6, spp=8)
    a, rays_a = O.render_serial(cfg, block_size=32)
    b, rays_b = O.render_serial(cfg, block_size=32)
    assert np.array_equal(a, b) and rays_a == rays_b
    c, _ = O.render(cfg)
    for k in range(5):
        # the weight channel does not depend on the random numbers -- except for the rare sample within 1e-5 of a pixel edge, which the
        # box filter (radius 0.5 + 1e-5, box.cpp:38) splats onto two pixels (4e-5 of all samples)
        assert (np.abs(a[k][..., 3] - c[k][..., 3]) > 1e-9).sum() <= 12
    for k, tol in ((1, 0.05), (4, 0.25)):          # (the directly visible light covers ~20 pixels of this film: its mean is noisy at 8 spp)
        ma, mc = a[k][..., :3].sum() / a[k][..., 3].sum(), c[k][..., :3].sum() / c[k][..., 3].sum()
        assert abs(ma - mc) <= tol * abs(mc), (k, ma, mc)
    d, _ = O.render_serial(cfg, block_size=32, parent_seed=1)
    assert not np.array_equal(a[1], d[1])
    e, _ = O.render_serial(cfg, block_size=16)                                    # another block size = another pixel order = other numbers per pixel
    assert not np.array_equal(a[1], e[1])
    # the first sample of the run: the clone's first two draws place it in the centre block's first Hilbert pixel
    parent = go.Random(5489)
    first = parent.clone().floats(2)
    blocks = go.spiral_blocks(48, 40, 32)
    px, py = int(blocks[0][0]), int(blocks[0][1])
    one = go.config(maxDepth=1, spp=1)                                            # depth 1: only very-direct light, one put per sample on buffer 4
    f, _ = O.render_serial(one, block_size=32)
    assert f[4][py, px, 3] > 0 and 0 <= first[0] < 1 and 0 <= first[1] < 1
