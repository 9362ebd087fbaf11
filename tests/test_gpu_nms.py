"""GPU parity: polygon IoU and rotated NMS through the C-ABI vs the CPU oracle
(bit-exact: fp64 IoU values, kept indices)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import postprocess as pp
from conftest import NMS_SET_KINDS, nms_candidate_set, rrects

NMS_EXACT_ONLY = 1          # include/dafne_amd.h DAFNE_NMS_EXACT_ONLY

pytestmark = pytest.mark.gpu

CASES = ["rand1", "rand2", "rand63", "rand64", "rand65", "rand300", "rand1000", "ties45",
         "degenerate", "kat_resultmerge", "thr05_oneclass", "negcoords"]


def dev():
    return torch.device("cuda", 0)


def test_iou_pairs_golden_bit_exact(golden):
    from dafne_amd.modeling.nms import poly_iou_pairs
    g = golden("iou_pairs")
    got = poly_iou_pairs(torch.from_numpy(g["p"]).to(dev()), torch.from_numpy(g["q"]).to(dev())).cpu().numpy()
    assert np.array_equal(got, g["iou"])


def test_iou_pairs_random_bit_exact():
    from dafne_amd.modeling.nms import poly_iou_pairs
    rng = np.random.default_rng(77)
    n = 100000
    p = rrects(n, rng, extent=300.0).astype(np.float64)
    q = rrects(n, rng, extent=300.0).astype(np.float64)
    p[n // 2:] = rng.normal(0, 8, (n - n // 2, 8))      # arbitrary / self-intersecting quads
    q[n // 2:] = rng.normal(0, 8, (n - n // 2, 8))
    q[:1000] = p[:1000]                                   # identical
    q[1000:2000] = p[1000:2000] + 1e-7                    # nearly identical
    p[2000:3000] += 20000.0                               # class-offset magnitudes
    q[2000:3000] += 20000.0
    got = poly_iou_pairs(torch.from_numpy(p).to(dev()), torch.from_numpy(q).to(dev())).cpu().numpy()
    exp = oracle.iou_poly_pairs(p, q)
    assert np.array_equal(got, exp), int((got != exp).sum())


@pytest.mark.parametrize("name", CASES)
def test_batched_nms_poly_golden(golden, name):
    from dafne_amd.modeling.nms import batched_nms_poly
    g = golden("nms_cases")
    keep = batched_nms_poly(torch.from_numpy(g[name + "_boxes"]).to(dev()),
                            torch.from_numpy(g[name + "_scores"]).to(dev()),
                            torch.from_numpy(g[name + "_classes"]).to(dev()), float(g[name + "_thr"]))
    assert keep.dtype == torch.int64
    assert keep.cpu().tolist() == g[name + "_keep"].tolist()


def test_poly_gpu_nms_matches_oracle_and_known_answer():
    from dafne_amd.modeling.nms import poly_gpu_nms
    d = np.array([[6.86e2, 2.976e3, 7.09e2, 2.976e3, 7.24e2, 2.976e3, 7.01e2, 2.976e3, 2.7137e-3],
                  [6.86e2, 2.976e3, 7.09e2, 2.976e3, 7.24e2, 2.976e3, 7.01e2, 2.976e3, 2.7097e-3]], np.float32)
    assert poly_gpu_nms(d, 0.1, 0) == [0]          # ResultMerge.py:54-63
    assert poly_gpu_nms(np.zeros((0, 9), np.float32), 0.1, 0) == []
    rng = np.random.default_rng(8)
    for m, ext, thr in ((777, 200.0, 0.1), (2000, 1024.0, 0.1), (1500, 150.0, 0.3), (3000, 400.0, 0.1)):
        b = rrects(m, rng, extent=ext)
        s = rng.uniform(0.05, 1, m).astype(np.float32)
        s[::5] = s[1::5][: len(s[::5])]                       # score ties
        c = rng.integers(0, 16, m)
        d9 = oracle.build_dets9(b, s, c)
        assert poly_gpu_nms(d9, thr, 0) == oracle.poly_nms(d9, thr, fast=True)


def test_tiny_and_degenerate_boxes_bypass_prefilter_correctly():
    """Boxes with ~0 area: the union==0 branch makes disjoint zero-area boxes
    suppress each other -- the hull pre-filter must not hide it."""
    from dafne_amd.modeling.nms import poly_gpu_nms
    rng = np.random.default_rng(9)
    m = 300
    b = rrects(m, rng, extent=100.0)
    b[:100, 2:] = np.tile(b[:100, :2], (1, 3))              # points
    b[100:150] = rrects(50, rng, extent=100.0, lo=1e-4, hi=1e-3, jitter=0.0)   # microscopic
    s = rng.uniform(0.05, 1, m).astype(np.float32)
    d9 = np.concatenate([b, s[:, None]], 1).astype(np.float32)
    assert poly_gpu_nms(d9, 0.1, 0) == oracle.poly_nms(d9, 0.1)


def test_batched_with_device_counts_and_cap():
    from dafne_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(10)
    N, cap = 3, 1200
    counts = [1200, 0, 517]
    boxes = np.zeros((N, cap, 8), np.float32)
    scores = np.zeros((N, cap), np.float32)
    classes = np.zeros((N, cap), np.int32)
    for i, m in enumerate(counts):
        boxes[i, :m] = rrects(m, rng, extent=600.0)
        scores[i, :m] = rng.uniform(0.05, 1, m)
        classes[i, :m] = rng.integers(0, 15, m)
    d = dev()
    tb, ts, tc = (torch.from_numpy(a).to(d) for a in (boxes, scores, classes))
    tn = torch.tensor(counts, dtype=torch.int32, device=d)
    keep = torch.full((N, cap), -1, dtype=torch.int64, device=d)
    nk = torch.zeros(N, dtype=torch.int32, device=d)
    nbytes = L.dafne_poly_nms_workspace_bytes(N, cap)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    post = 400
    _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), _lib.ptr(tn), N, cap,
                                                  0.1, post, _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes,
                                                  0, _lib.current_stream()))
    torch.cuda.synchronize()
    for i, m in enumerate(counts):
        det = {"pred_corners": boxes[i, :m], "scores": scores[i, :m], "pred_classes": classes[i, :m].astype(np.int64)}
        exp_keep = pp.batched_nms_poly(det["pred_corners"], det["scores"], det["pred_classes"], 0.1, fast=True)
        if len(exp_keep) > post:
            kth = np.sort(scores[i][exp_keep])[len(exp_keep) - post]
            exp_keep = exp_keep[scores[i][exp_keep] >= kth]
        n = int(nk[i])
        assert keep[i, :n].cpu().tolist() == exp_keep.tolist()


def test_full_size_properties():
    """M = 10 000 (5 levels x PRE_NMS_TOPK) and 27 000 (TTA merge): idempotence and
    mutual non-overlap of the kept set, checked with the oracle on a sample."""
    from dafne_amd.modeling.nms import batched_nms_poly, poly_gpu_nms, poly_iou_pairs
    rng = np.random.default_rng(1234)
    for m in (10000, 27000):
        b = rrects(m, rng, extent=1024.0)
        s = rng.uniform(0.05, 1, m).astype(np.float32)
        c = rng.integers(0, 16, m)
        d9 = oracle.build_dets9(b, s, c)          # nms.py:74-90 on the host
        k = np.asarray(poly_gpu_nms(d9, 0.1, 0))
        assert k.tolist() == oracle.poly_nms(d9, 0.1, fast=True)            # the full-size list itself, bit-exact
        # the fused device path (offsets computed on the GPU) gives the same list
        fused = batched_nms_poly(torch.from_numpy(b).to(dev()), torch.from_numpy(s).to(dev()),
                                 torch.from_numpy(c).to(dev()), 0.1)
        assert fused.cpu().tolist() == k.tolist()
        assert len(np.unique(k)) == len(k) and np.all(np.diff(s[k]) <= 0)     # unique, sorted by score
        # idempotent (equal scores may swap: ties go to the larger row index)
        assert sorted(poly_gpu_nms(d9[k], 0.1, 0)) == list(range(len(k)))
        # kept boxes never overlap above the threshold (sampled pairs, device IoU)
        i = rng.integers(0, len(k), 300000)
        j = rng.integers(0, len(k), 300000)
        sel = i != j
        tk = torch.from_numpy(d9[k, :8]).to(dev()).double()
        iou = poly_iou_pairs(tk[torch.from_numpy(i[sel]).to(dev())], tk[torch.from_numpy(j[sel]).to(dev())])
        assert float(iou.max()) <= 0.1
        # every suppressed box has an earlier kept box with IoU > thr (oracle, sample)
        sup = np.setdiff1d(np.arange(m), k)[:50]
        for t in sup:
            cand = k[(s[k] > s[t]) | ((s[k] == s[t]) & (k > t))]
            iou_t = oracle.iou_poly_pairs(np.repeat(d9[t:t + 1, :8], len(cand), 0), d9[cand, :8])
            assert (iou_t > 0.1).any()


def test_fast_path_decisions_around_the_threshold():
    """The convex fast path of nms_iou decides `IoU > thresh` from a geometric clip and hands anything within
    1e-3 of the threshold (or non-convex / tiny / degenerate) to the reference-order clip: 6000 two-box images
    whose IoU is spread tightly around 0.1, plus concave, self-intersecting, sub-pixel and huge-offset pairs,
    must give the oracle's keep lists."""
    from dafne_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(2024)
    n = 6000
    dets = np.zeros((n, 2, 9), dtype=np.float32)
    for i in range(n):
        w, h = rng.uniform(4, 200), rng.uniform(4, 60)
        ang = rng.uniform(0, np.pi)
        ca, sa = np.cos(ang), np.sin(ang)
        base = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
        rot = np.array([[ca, -sa], [sa, ca]])
        a = base @ rot.T + rng.uniform(0, 900, 2)
        # shift along the long side: IoU of two equal rectangles shifted by s*w is (1-s)/(1+s); target ~0.1 -> s ~ 9/11
        target = 0.1 + rng.normal(0, 2e-3) if i % 3 else rng.uniform(0.0, 0.3)
        s = (1 - target) / (1 + target)
        b = a + s * w * np.array([ca, sa])
        kind = i % 10
        if kind == 7:       # concave arrowhead
            b[2] = b.mean(0)
        elif kind == 8:     # self-intersecting bow tie
            b[[2, 3]] = b[[3, 2]]
        elif kind == 9:     # sub-pixel pair
            a, b = a * 1e-2, b * 1e-2
        if i % 50 == 0:     # class-offset magnitude
            a, b = a + 30000.0, b + 30000.0
        dets[i, 0, :8], dets[i, 1, :8] = a.reshape(-1), b.reshape(-1)
        dets[i, 0, 8], dets[i, 1, 8] = 0.9, 0.8
    d = torch.from_numpy(dets).to(dev())
    keep = torch.empty((n, 2), dtype=torch.int64, device=dev())
    nk = torch.zeros(n, dtype=torch.int32, device=dev())
    nbytes = L.dafne_poly_nms_workspace_bytes(n, 2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    _lib.check(L.dafne_poly_nms_batched_hip(_lib.ptr(d), None, n, 2, 0.1, 0, _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes,
                                            0, _lib.current_stream()), "nms")
    got = nk.cpu().numpy()
    want = np.array([len(oracle.poly_nms(dets[i], 0.1)) for i in range(n)])
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert 0.2 < (want == 1).mean() < 0.8          # the set really straddles the threshold


@pytest.mark.parametrize("n_zero", [0, 1, 2, 5])
def test_class_major_tile_order_matches_global_greedy(n_zero):
    """select_over_all_levels lays the rows out class by class (tiles of blocks without a common class are
    skipped).  That is only equivalent to the global greedy pass while at most ONE box has exactly zero area
    (two zero-area boxes have IoU 1 whatever their class: polyiou.cpp's union == 0 branch); with more the
    kernels fall back to score order.  Checked against the oracle, incl. quantised scores (ties across
    classes), class 5 -> 4 merging, a post-NMS cap with ties, and an image with a single class."""
    from dafne_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(100 + n_zero)
    N, cap = 4, 3000
    counts = [3000, 1500, 700, 64]
    boxes = np.zeros((N, cap, 8), np.float32)
    scores = np.zeros((N, cap), np.float32)
    classes = np.zeros((N, cap), np.int32)
    for i, m in enumerate(counts):
        boxes[i, :m] = rrects(m, rng, extent=500.0)
        scores[i, :m] = np.round(rng.uniform(0.05, 1, m), 2)          # many exact ties
        classes[i, :m] = 7 if i == 2 else rng.integers(0, 16, m)
        zi = rng.choice(m, n_zero, replace=False)
        for k, z in enumerate(zi):                                     # exact zero area, different classes
            boxes[i, z] = np.tile(boxes[i, z, :2], 4)
            classes[i, z] = (3 * k + i) % 16
    d = dev()
    tb, ts, tc = (torch.from_numpy(a).to(d) for a in (boxes, scores, classes))
    tn = torch.tensor(counts, dtype=torch.int32, device=d)
    keep = torch.full((N, cap), -1, dtype=torch.int64, device=d)
    nk = torch.zeros(N, dtype=torch.int32, device=d)
    nbytes = L.dafne_poly_nms_workspace_bytes(N, cap)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    post = 300
    _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), _lib.ptr(tn), N, cap, 0.1, post,
                                                  _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    for i, m in enumerate(counts):
        exp_keep = pp.batched_nms_poly(boxes[i, :m], scores[i, :m], classes[i, :m].astype(np.int64), 0.1, fast=True)
        if len(exp_keep) > post:
            kth = np.sort(scores[i][exp_keep])[len(exp_keep) - post]
            exp_keep = exp_keep[scores[i][exp_keep] >= kth]
        assert keep[i, :int(nk[i])].cpu().tolist() == exp_keep.tolist(), (i, n_zero)


@pytest.mark.parametrize("n_zero", [0, 3])
def test_class_major_order_on_the_counting_path(n_zero):
    """Same equivalence for sets above 16384 rows (the TTA merge), where the layout comes from
    nms_cls_layout + the rank-by-counting kernel instead of the in-LDS sort."""
    from dafne_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(300 + n_zero)
    m = 17000
    boxes = rrects(m, rng, extent=1500.0)[None].copy()
    scores = np.round(rng.uniform(0.05, 1, (1, m)), 3).astype(np.float32)
    classes = rng.integers(0, 16, (1, m)).astype(np.int32)
    for k, z in enumerate(rng.choice(m, n_zero, replace=False)):
        boxes[0, z] = np.tile(boxes[0, z, :2], 4)
        classes[0, z] = (5 * k + 1) % 16
    d = dev()
    tb, ts, tc = (torch.from_numpy(a).to(d) for a in (boxes, scores, classes))
    keep = torch.full((1, m), -1, dtype=torch.int64, device=d)
    nk = torch.zeros(1, dtype=torch.int32, device=d)
    nbytes = L.dafne_poly_nms_workspace_bytes(1, m)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), None, 1, m, 0.1, 1000,
                                                  _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    exp_keep = pp.batched_nms_poly(boxes[0], scores[0], classes[0].astype(np.int64), 0.1, fast=True)
    if len(exp_keep) > 1000:
        kth = np.sort(scores[0][exp_keep])[len(exp_keep) - 1000]
        exp_keep = exp_keep[scores[0][exp_keep] >= kth]
    assert keep[0, :int(nk[0])].cpu().tolist() == exp_keep.tolist()


def test_chunked_sort_path_with_ragged_counts():
    """m_cap above 16384 with per-image counts on both sides of the chunk size: image 0 is sorted in two chunks and
    merged by binary search, image 1 (9 000 rows) in one; many equal scores (ties go to the larger row index)."""
    from dafne_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(77)
    m_cap, counts = 20000, [20000, 9000]
    boxes = np.stack([rrects(m_cap, rng, extent=1800.0) for _ in counts])
    scores = np.round(rng.uniform(0.05, 1, (2, m_cap)), 2).astype(np.float32)        # ~95 distinct values: ties everywhere
    classes = rng.integers(0, 15, (2, m_cap)).astype(np.int32)
    d = dev()
    tb, ts, tc = (torch.from_numpy(a).to(d) for a in (boxes, scores, classes))
    tn = torch.tensor(counts, dtype=torch.int32, device=d)
    keep = torch.full((2, m_cap), -1, dtype=torch.int64, device=d)
    nk = torch.zeros(2, dtype=torch.int32, device=d)
    nbytes = L.dafne_poly_nms_workspace_bytes(2, m_cap)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), _lib.ptr(tn), 2, m_cap, 0.1, 1000,
                                                  _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    for i, m in enumerate(counts):
        exp_keep = pp.batched_nms_poly(boxes[i, :m], scores[i, :m], classes[i, :m].astype(np.int64), 0.1, fast=True)
        if len(exp_keep) > 1000:
            kth = np.sort(scores[i][exp_keep])[len(exp_keep) - 1000]
            exp_keep = exp_keep[scores[i][exp_keep] >= kth]
        assert keep[i, :int(nk[i])].cpu().tolist() == exp_keep.tolist(), i


def test_coincident_edges_duplicates_and_nesting():
    """Axis-parallel and 45-degree boxes on an integer grid: exact duplicates, boxes sharing whole edges or single
    vertices, nested boxes, vertices lying on another box's edge.  These are the cases where a sign test of the
    register-only fast path (boundary integral of clipped edge intervals) could go either way; they must be detected
    (|signed distance| <= 1e-6 |edge|) and take the exact reference-order path.  Keep lists vs the oracle."""
    from dafne_amd.modeling.nms import batched_nms_poly, poly_gpu_nms
    rng = np.random.default_rng(99)
    rows = []
    for _ in range(1500):                       # axis-parallel, integer corners: edges coincide / touch all the time
        x0, y0 = rng.integers(0, 40, 2) * 4.0
        w, h = rng.integers(1, 6, 2) * 4.0
        rows.append([x0, y0, x0 + w, y0, x0 + w, y0 + h, x0, y0 + h])
    for _ in range(700):                        # diamonds on the same grid (vertices land on the rectangles' edges)
        cx, cy = rng.integers(2, 40, 2) * 4.0
        r = rng.integers(1, 4) * 4.0
        rows.append([cx - r, cy, cx, cy - r, cx + r, cy, cx, cy + r])
    b = np.asarray(rows, np.float32)
    b = np.concatenate([b, b[rng.integers(0, len(b), 400)]])          # exact duplicates
    cw = b[rng.integers(0, len(b), 200)].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)      # clockwise twins
    b = np.concatenate([b, cw])
    for thr in (0.1, 0.3, 1.0 / 3.0, 0.5):
        s = np.round(rng.uniform(0.05, 1, len(b)), 2).astype(np.float32)        # many equal scores
        d9 = np.concatenate([b, s[:, None]], 1).astype(np.float32)
        assert poly_gpu_nms(d9, thr, 0) == oracle.poly_nms(d9, thr)
        # class-aware select path (class offsets, class-major tile order, IoU upper bound in the tile pre-filter)
        c = rng.integers(0, 6, len(b))
        got = batched_nms_poly(torch.from_numpy(b).to(dev()), torch.from_numpy(s).to(dev()), torch.from_numpy(c).to(dev()), thr)
        assert got.cpu().tolist() == pp.batched_nms_poly(b, s, c.astype(np.int64), thr).tolist()


def _select(boxes, scores, classes, thr, post, counts=None, flags=0):
    """dafne_select_over_all_levels_hip on [N,M,*] arrays -> (list of keep lists, stats [N,4])."""
    from dafne_amd import _lib
    L = _lib.load()
    N, m = scores.shape
    d = dev()
    tb, ts, tc = (torch.from_numpy(np.ascontiguousarray(a)).to(d) for a in (boxes, scores, classes.astype(np.int32)))
    tn = None if counts is None else torch.tensor(counts, dtype=torch.int32, device=d)
    keep = torch.full((N, m), -1, dtype=torch.int64, device=d)
    nk = torch.zeros(N, dtype=torch.int32, device=d)
    nbytes = L.dafne_poly_nms_workspace_bytes(N, m)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), _lib.ptr(tn), N, m, thr, post,
                                                  _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, int(flags), _lib.current_stream()))
    torch.cuda.synchronize()
    off = L.dafne_poly_nms_stats_offset(N, m, 0)
    stats = ws[off:off + 16 * N].view(torch.int32).reshape(N, 4).cpu().numpy()
    return [keep[i, :int(nk[i])].cpu().tolist() for i in range(N)], stats


def _oracle_select(b, s, c, thr, post):
    k = pp.batched_nms_poly(b, s, c, thr, fast=True)
    if post > 0 and len(k) > post:
        kth = np.sort(s[k])[len(k) - post]
        k = k[s[k] >= kth]
    return k.tolist()


@pytest.mark.parametrize("m", [10000, 27000])
@pytest.mark.parametrize("kind", NMS_SET_KINDS)
def test_full_size_sets_exact_vs_oracle(kind, m):
    """SURVEY 8(d) candidate sets at the two full sizes -- M = 10 000 (5 levels x PRE_NMS_TOPK: configs 2/3/5) and
    27 000 (the 27-view TTA merge: config 4) -- uniform, DENSE (70 % of the boxes in one 256^2 window) and the DOTA-1.5
    SKEWED class histogram (60 % in {4,5,6}, 16 classes, incl. the 5 -> 4 merge): the keep lists of the class-aware
    path (with and without the post-NMS cap) and of the plain [M,9] entry point equal the oracle's, bit for bit."""
    from dafne_amd.modeling.nms import poly_gpu_nms
    rng = np.random.default_rng(1234)
    b, s, c = nms_candidate_set(kind, m, rng)
    for post in (0, 1000):
        got, stats = _select(b[None], s[None], c[None], 0.1, post)
        assert got[0] == _oracle_select(b, s, c, 0.1, post), (kind, m, post)
    assert stats[0].sum() > 0
    d9 = oracle.build_dets9(b, s, c)
    assert poly_gpu_nms(d9, 0.1, 0) == oracle.poly_nms(d9, 0.1, fast=True)
    print("NMS_PATHS %s M=%d fast-suppress %d fast-keep %d exact %d exact(overflow tiles) %d" % ((kind, m) + tuple(stats[0])))


def test_exact_only_mode_changes_nothing():
    """The per-call flag DAFNE_NMS_EXACT_ONLY switches off the hull pre-filter, the IoU upper bound and the convex decision
    fast path: every pair of every live tile goes through polyiou.cpp's operation order.  Same keep lists as the
    default mode and as the UNFILTERED oracle, on dense / skewed sets and a batch with ragged counts."""
    from dafne_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(4321)
    sets = [nms_candidate_set(kind, 3000, rng, extent=400.0) for kind in NMS_SET_KINDS]
    B = np.stack([x[0] for x in sets]); S = np.stack([x[1] for x in sets]); C = np.stack([x[2] for x in sets])
    counts = [3000, 2500, 1777]
    fast, st_fast = _select(B, S, C, 0.1, 300, counts)
    exact, st_exact = _select(B, S, C, 0.1, 300, counts, flags=NMS_EXACT_ONLY)
    assert st_exact[:, :2].sum() == 0 and st_exact[:, 2:].sum() > st_fast[:, 2:].sum()
    assert st_fast[:, :2].sum() > 0
    for i, n in enumerate(counts):
        k = pp.batched_nms_poly(B[i, :n], S[i, :n], C[i, :n], 0.1, fast=False)          # unfiltered oracle
        if len(k) > 300:
            k = k[S[i][k] >= np.sort(S[i][k])[len(k) - 300]]
        assert exact[i] == k.tolist() and fast[i] == k.tolist(), i


@pytest.mark.parametrize("thr", [0.05, 0.1, 0.3, 0.5])
def test_decision_fuzz_one_million_pairs(thr):
    """>= 10^6 two-box images in total (250 000 per threshold) whose reference IoU lies within +-5e-3 of the threshold:
    shifted / rotated copies with sides from 1 to 2000 px and aspect up to 1:50, fp32 class-offset magnitudes up to
    16 x (span + 1), near-convex quads with a 1e-3 px reflex vertex, pairs sitting on the decision edge of the hull
    IoU upper bound (tests/nms_fuzz.py).  The keep count of every image must equal the oracle's decision
    iou_poly(fp64) > thr; the path counters say how many pairs each path decided."""
    import nms_fuzz
    from dafne_amd import _lib
    L = _lib.load()
    n, chunk = 250000, 50000
    paths = np.zeros(4, np.int64)
    near = 0
    for c0 in range(0, n, chunk):
        dets, fam = nms_fuzz.make_pairs(chunk, thr, seed=int(thr * 1000) * 100 + c0 // chunk)
        want, iou = nms_fuzz.expected_keep_counts(dets, thr)
        near += int((np.abs(iou - thr) <= 5.5e-3).sum())
        d = torch.from_numpy(dets).to(dev())
        keep = torch.empty((chunk, 2), dtype=torch.int64, device=dev())
        nk = torch.zeros(chunk, dtype=torch.int32, device=dev())
        nbytes = L.dafne_poly_nms_workspace_bytes(chunk, 2)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
        _lib.check(L.dafne_poly_nms_batched_hip(_lib.ptr(d), None, chunk, 2, thr, 0, _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws),
                                                nbytes, 0, _lib.current_stream()), "nms")
        got = nk.cpu().numpy()
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (thr, c0, bad[:10], fam[bad[:10]], iou[bad[:10]])
        off = L.dafne_poly_nms_stats_offset(chunk, 2, 0)
        paths += ws[off:off + 16 * chunk].view(torch.int32).reshape(chunk, 4).sum(0).cpu().numpy()
        del ws, d, keep, nk
    listed = int(paths.sum())
    print("NMS_FUZZ thr=%.2f pairs=%d within5e-3=%d listed=%d fast-suppress=%d fast-keep=%d exact=%d dropped-by-scan=%d"
          % (thr, n, near, listed, paths[0], paths[1], paths[2] + paths[3], n - listed))
    assert near >= 0.75 * n                      # the set really hugs the threshold
    assert paths[2] + paths[3] >= 0.3 * n        # ... so most pairs need the reference-order path,
    assert paths[0] + paths[1] > 0               # and the fast path still decided some


@pytest.mark.parametrize("thr", [0.05, 0.1, 0.3, 0.5])
def test_decision_fuzz_general_quads(thr):
    """100 000 two-box images per threshold of NON-convex quads (random four points, darts, bow-ties, against each other and
    against rectangles; a vertex within 0.5 px of the coordinate origin; fp32 class-offset magnitudes), half of them with the
    reference IoU within 5e-3 of the threshold, half within 5e-2 (tests/nms_fuzz.py make_general_pairs).  These are the
    quads an untrained head emits (79 % of the bench pipeline's listed pairs): the winding-number form of the fast path
    decides the ones away from the threshold, the reference-order clip the rest.  Every keep count must equal the
    oracle's decision iou_poly(fp64) > thr -- with the fast paths on and with them switched off."""
    import nms_fuzz
    from dafne_amd import _lib
    L = _lib.load()
    n = 100000
    dets, typ = nms_fuzz.make_general_pairs(n, thr, seed=int(thr * 1000) + 7)
    want, iou = nms_fuzz.expected_keep_counts(dets, thr)
    nonconvex = ~(nms_fuzz.is_convex(dets[:, 0, :8]) & nms_fuzz.is_convex(dets[:, 1, :8]))
    assert nonconvex.mean() > 0.8
    d = torch.from_numpy(dets).to(dev())
    nbytes = L.dafne_poly_nms_workspace_bytes(n, 2)
    for exact_only in (0, 1):
        keep = torch.empty((n, 2), dtype=torch.int64, device=dev())
        nk = torch.zeros(n, dtype=torch.int32, device=dev())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
        _lib.check(L.dafne_poly_nms_batched_hip(_lib.ptr(d), None, n, 2, thr, 0, _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws),
                                                nbytes, NMS_EXACT_ONLY if exact_only else 0, _lib.current_stream()), "nms")
        got = nk.cpu().numpy()
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (thr, exact_only, bad[:10], typ[bad[:10]], iou[bad[:10]])
        off = L.dafne_poly_nms_stats_offset(n, 2, 0)
        paths = ws[off:off + 16 * n].view(torch.int32).reshape(n, 4).sum(0).cpu().numpy()
        if not exact_only:
            listed = int(paths.sum())
            print("NMS_FUZZ_GENERAL thr=%.2f pairs=%d non-convex=%d listed=%d fast-suppress=%d fast-keep=%d exact=%d"
                  % (thr, n, int(nonconvex.sum()), listed, paths[0], paths[1], paths[2] + paths[3]))
            assert paths[0] + paths[1] >= 0.25 * listed       # the general fast path carries the pairs away from the threshold
            assert paths[2] + paths[3] >= 0.1 * listed        # ... and the reference-order path the ones at it
        del ws, keep, nk


def test_shims_import_by_the_reference_names(golden):
    """shims/poly_nms.py and shims/polyiou.py are what a maintainer puts on PYTHONPATH in place of the external
    `poly_nms` CUDA extension (nms.py:6,91) and the SWIG `polyiou` module (voc_eval.py:184, ResultMerge:38-43): import
    them by those names, call them with the reference's argument forms, compare with the golden fixtures."""
    import importlib
    import os
    import sys
    shim_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shims")
    sys.path.insert(0, shim_dir)
    try:
        for name in ("poly_nms", "polyiou", "_dafne_amd_lib"):
            sys.modules.pop(name, None)
        poly_nms = importlib.import_module("poly_nms")
        polyiou = importlib.import_module("polyiou")
        assert os.path.dirname(poly_nms.__file__) == shim_dir and os.path.dirname(polyiou.__file__) == shim_dir
        g = golden("nms_cases")
        for name in ("rand300", "ties45", "degenerate", "kat_resultmerge"):
            d9 = oracle.build_dets9(g[name + "_boxes"], g[name + "_scores"], g[name + "_classes"])     # nms.py:74-90
            keep = poly_nms.poly_gpu_nms(d9, float(g[name + "_thr"]), 0)
            assert isinstance(keep, list) and keep == g[name + "_keep"].tolist()
        assert poly_nms.poly_gpu_nms(np.zeros((0, 9), np.float32), 0.1, 0) == []
        gi = golden("iou_pairs")
        for k in (0, 7, 1500, 3005, len(gi["iou"]) - 1):
            v = polyiou.iou_poly(polyiou.VectorDouble(gi["p"][k]), polyiou.VectorDouble(gi["q"][k].tolist()))
            assert isinstance(v, float) and v == gi["iou"][k]
        assert np.array_equal(polyiou.iou_poly_pairs(gi["p"], gi["q"]), gi["iou"])
        v = polyiou.VectorDouble()
        for x in (0, 0, 1, 0, 1, 1, 0, 1):
            v.push_back(x)
        assert v.size() == 8 and polyiou.iou_poly(v, polyiou.VectorDouble([0.5, 0, 1.5, 0, 1.5, 1, 0.5, 1])) == 1.0 / 3.0
    finally:
        sys.path.remove(shim_dir)
        for name in ("poly_nms", "polyiou", "_dafne_amd_lib"):
            sys.modules.pop(name, None)
