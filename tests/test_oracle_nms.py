"""CPU: greedy NMS oracle against the golden keep lists produced by the
reference's batched_nms_poly (nms.py) and its known answer."""
import numpy as np
import pytest

import oracle
from oracle import postprocess as pp
from conftest import rrects

CASES = ["rand1", "rand2", "rand63", "rand64", "rand65", "rand300", "rand1000", "ties45",
         "degenerate", "kat_resultmerge", "thr05_oneclass", "negcoords"]


@pytest.mark.parametrize("name", CASES)
def test_golden_keep_lists(golden, name):
    g = golden("nms_cases")
    keep = pp.batched_nms_poly(g[name + "_boxes"], g[name + "_scores"], g[name + "_classes"],
                               float(g[name + "_thr"]))
    assert keep.tolist() == g[name + "_keep"].tolist()
    fast = pp.batched_nms_poly(g[name + "_boxes"], g[name + "_scores"], g[name + "_classes"],
                               float(g[name + "_thr"]), fast=True)
    assert fast.tolist() == keep.tolist()


def test_resultmerge_known_answer():
    # tools/prepare_dota/ResultMerge.py:54-63
    d = np.array([[6.86e2, 2.976e3, 7.09e2, 2.976e3, 7.24e2, 2.976e3, 7.01e2, 2.976e3, 2.7137e-3],
                  [6.86e2, 2.976e3, 7.09e2, 2.976e3, 7.24e2, 2.976e3, 7.01e2, 2.976e3, 2.7097e-3]], np.float32)
    assert oracle.poly_nms(d, 0.1) == [0]


def test_empty_and_order():
    assert oracle.poly_nms(np.zeros((0, 9), np.float32), 0.1) == []
    d = np.zeros((4, 9), np.float32)
    d[:, :8] = np.array([0, 0, 1, 0, 1, 1, 0, 1], np.float32) + 10 * np.arange(4)[:, None]
    d[:, 8] = [0.5, 0.9, 0.5, 0.7]
    # disjoint boxes: all kept, score descending, equal scores -> larger index first
    assert oracle.poly_nms(d, 0.1) == [1, 3, 2, 0]
    assert oracle.score_order(d).tolist() == np.argsort(d[:, 8], kind="stable")[::-1].tolist()


def test_class_offsets_float32():
    rng = np.random.default_rng(3)
    b = rrects(50, rng, extent=100.0)
    s = rng.uniform(0, 1, 50).astype(np.float32)
    c = rng.integers(0, 16, 50)
    d = oracle.build_dets9(b, s, c)
    cc = np.where(c == 5, 4, c).astype(np.float32)
    span = np.float32(np.float32(b.max() - b.min()) + np.float32(1))
    exp = (b + (cc * span)[:, None]).astype(np.float32)
    assert np.array_equal(d[:, :8], exp) and np.array_equal(d[:, 8], s)


def test_fast_equals_plain_on_dense_random():
    rng = np.random.default_rng(11)
    for m, ext in ((500, 100.0), (1500, 1024.0)):
        b = rrects(m, rng, extent=ext)
        s = rng.uniform(0.05, 1, m).astype(np.float32)
        c = rng.integers(0, 15, m)
        d = oracle.build_dets9(b, s, c)
        assert oracle.poly_nms(d, 0.1, fast=True) == oracle.poly_nms(d, 0.1)


def test_idempotent():
    rng = np.random.default_rng(12)
    b = rrects(800, rng, extent=300.0)
    s = rng.uniform(0.05, 1, 800).astype(np.float32)
    d = oracle.build_dets9(b, s, np.zeros(800, np.int64))
    k = oracle.poly_nms(d, 0.1)
    assert oracle.poly_nms(d[k], 0.1) == list(range(len(k)))


def test_fuzz_pair_generator_hugs_the_threshold():
    """tests/nms_fuzz.py (the GPU decision fuzz's input): nine of ten families put the reference IoU within 5.5e-3 of the
    threshold, on both sides of it."""
    import nms_fuzz
    for thr in (0.05, 0.5):
        dets, fam = nms_fuzz.make_pairs(20000, thr, seed=3)
        want, iou = nms_fuzz.expected_keep_counts(dets, thr)
        assert dets.dtype == np.float32 and dets.shape == (20000, 2, 9)
        for f in range(8):
            k = fam == f
            assert (np.abs(iou[k] - thr) <= 5.5e-3).mean() > 0.95, (thr, f)
            assert 0.3 < (want[k] == 1).mean() < 0.7, (thr, f)
        assert (np.abs(iou[fam == 8] - thr) <= 5.5e-3).mean() > 0.4      # the axis-aligned half


def test_general_quad_generator_is_nonconvex_and_hugs_the_threshold():
    """tests/nms_fuzz.py make_general_pairs (input of the GPU fuzz of the winding-number fast path): mostly non-convex pairs,
    the shifted-copy types within 5.5e-2 of the threshold on both sides, bow-ties with NEGATIVE reference IoU included."""
    import nms_fuzz
    dets, typ = nms_fuzz.make_general_pairs(10000, 0.1, seed=5)
    want, iou = nms_fuzz.expected_keep_counts(dets, 0.1)
    nonconvex = ~(nms_fuzz.is_convex(dets[:, 0, :8]) & nms_fuzz.is_convex(dets[:, 1, :8]))
    assert nonconvex.mean() > 0.8
    for t in (0, 1, 4):
        k = typ == t
        assert (np.abs(iou[k] - 0.1) <= 5.5e-2).mean() > 0.9, t
        assert 0.3 < (want[k] == 1).mean() < 0.7, t
    assert (iou < 0).any()                                # winding-number products of bow-ties can be negative (polyiou.cpp:69-93)
    assert (np.abs(dets[typ == 4, 0, 0:2]) <= 0.5).all()  # vertex 0 within 0.5 px of the origin
