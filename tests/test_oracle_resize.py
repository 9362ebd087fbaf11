"""CPU: the oracle's restatement of Pillow's 8-bit bilinear resize vs the PIL-generated fixture and, when Pillow is
importable, vs Pillow itself at the TTA sizes (shortest edge 450..1200 of a 1024 tile, scaled down to keep it quick)."""
import os

import numpy as np
import pytest

from oracle import resize as orz

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_pil.npz"))


def test_against_pil_fixture():
    i = 0
    while "in_%d" % i in G:
        img, want = G["in_%d" % i], G["out_%d" % i]
        got = orz.resize_bilinear_u8(img.transpose(2, 0, 1), want.shape[0], want.shape[1]).transpose(1, 2, 0)
        assert np.array_equal(got, want), i
        i += 1
    assert i >= 7


def test_against_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for h, w, nh, nw in [(256, 256, 113, 113), (256, 256, 300, 300), (160, 200, 90, 113), (33, 517, 100, 60), (300, 40, 13, 77)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        got = orz.resize_bilinear_u8(img.transpose(2, 0, 1), nh, nw).transpose(1, 2, 0)
        assert np.array_equal(got, want), (h, w, nh, nw)


def test_flips_are_index_reversals():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (3, 40, 56), dtype=np.uint8)
    base = orz.resize_bilinear_u8(img, 25, 70)
    assert np.array_equal(orz.resize_bilinear_u8(img, 25, 70, hflip=True), base[:, :, ::-1])
    assert np.array_equal(orz.resize_bilinear_u8(img, 25, 70, vflip=True), base[:, ::-1, :])
