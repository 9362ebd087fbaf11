"""CPU: the C restatement (oracle/poly_oracle.c) against the reference's known
answers, the committed golden vectors (made from the reference's own compiled
polyiou.cpp) and, when present, oracle/_ref itself."""
import numpy as np
import pytest

import oracle
from conftest import rrects


def test_known_answers():
    sq = [0, 0, 1, 0, 1, 1, 0, 1]
    # polyiou.cpp:149-150 (commented main): unit square vs +0.5 shift -> 1/7
    assert oracle.iou_poly(sq, [0.5, 0.5, 1.5, 0.5, 1.5, 1.5, 0.5, 1.5]) == pytest.approx(1 / 7, abs=1e-15)
    # polyiou.cpp:137-146: identical degenerate quads -> union == 0 branch -> 1.0
    d = [686, 2976, 709, 2976, 724, 2976, 701, 2976]
    assert oracle.iou_poly(d, d) == 1.0
    assert oracle.iou_poly(sq, sq) == 1.0
    assert oracle.iou_poly(sq, [5, 5, 6, 5, 6, 6, 5, 6]) == 0.0
    # poly_overlaps_test.py:7-24: rbox (1,1,2,10,0) vs (2,1,2,10,0) -> 10/30
    a = [0, -4, 2, -4, 2, 6, 0, 6]
    b = [1, -4, 3, -4, 3, 6, 1, 6]
    assert oracle.iou_poly(a, b) == pytest.approx(1 / 3, abs=1e-15)
    # clockwise input is re-oriented (polyiou.cpp:96-97)
    assert oracle.iou_poly(sq[::-1][1::2] + sq[::-1][0::2], sq) >= 0


def test_golden_pairs_bit_exact(golden):
    g = golden("iou_pairs")
    got = oracle.iou_poly_pairs(g["p"], g["q"])
    assert np.array_equal(got, g["iou"]), "fp64 IoU must match the reference bit for bit"


def test_against_compiled_reference_if_present():
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(5)
    p = rrects(20000, rng, extent=300.0).astype(np.float64)
    q = rrects(20000, rng, extent=300.0).astype(np.float64)
    assert np.array_equal(oracle.iou_poly_pairs(p, q), oracle.ref_iou_poly_pairs(p, q))
    p = rng.normal(0, 5, (20000, 8))
    q = rng.normal(0, 5, (20000, 8))          # self-intersecting / arbitrary quads
    assert np.array_equal(oracle.iou_poly_pairs(p, q), oracle.ref_iou_poly_pairs(p, q))


def test_f32_entry_equals_widened():
    rng = np.random.default_rng(6)
    p = rrects(500, rng, extent=100.0)
    q = rrects(500, rng, extent=100.0)
    assert np.array_equal(oracle.iou_poly_pairs(p, q),
                          oracle.iou_poly_pairs(p.astype(np.float64), q.astype(np.float64)))
