"""Pair generator for the rotated-NMS decision fuzz (tests/test_gpu_nms.py, tests/test_oracle_nms.py).

Every pair is a two-box image whose fp64 reference IoU (polyiou.cpp:112-133 on the float32 rows) lies within a few
1e-3 of the NMS threshold, i.e. exactly where the three analytic shortcuts of the HIP path (guarded hull pre-filter,
IoU upper bound, convex decision fast path) would have to be wrong to change a keep list.

Families (index % 10):
  0-3  rectangle + copy shifted along its long side; IoU of two equal rectangles shifted by s*w is (1-s)/(1+s), the
       shift is chosen for a target IoU of thr + U(-5e-3, 5e-3); side lengths 1 .. 2000 px, aspect 1 .. 50
  4    the same, thin (aspect 20 .. 50) and long
  5    the same at fp32 class-offset magnitudes: both boxes + k * (span + 1), k < 16, span up to 3000 (nms.py:81-83)
  6    copy rotated by a small angle as well; shift found by bisection on the oracle IoU
  7    near-convex quads: a triangle-like quad whose 4th vertex sits 1e-3 px INSIDE the line through its neighbours
       (one reflex vertex) -- or 1e-3 px outside (barely convex); shift by bisection
  8    overlapping pairs whose HULL-based IoU upper bound is within 5e-3 of the threshold (the nms_scan bound's decision
       edge); half of them axis-aligned, where the bound EQUALS the true IoU
  9    random overlapping pairs, IoU anywhere (controls)
"""
import numpy as np

import oracle


def _rect(w, h, ang, cx, cy):
    ca, sa = np.cos(ang), np.sin(ang)
    dx = np.stack([-w / 2, w / 2, w / 2, -w / 2], 1)
    dy = np.stack([-h / 2, -h / 2, h / 2, h / 2], 1)
    x = cx[:, None] + dx * ca[:, None] - dy * sa[:, None]
    y = cy[:, None] + dx * sa[:, None] + dy * ca[:, None]
    out = np.empty((len(w), 8))
    out[:, 0::2], out[:, 1::2] = x, y
    return out


def _shift(q, dx, dy):
    r = q.copy()
    r[:, 0::2] += dx[:, None]
    r[:, 1::2] += dy[:, None]
    return r


def _bisect_shift(a, b0, ux, uy, lo, hi, target, iters=16):
    """shift t in [lo, hi] along (ux, uy) such that IoU(a, b0 + t*u) ~ target (IoU decreases with t)."""
    lo, hi = lo.copy(), hi.copy()
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        iou = oracle.iou_poly_pairs(a.astype(np.float32), _shift(b0, mid * ux, mid * uy).astype(np.float32))
        big = iou > target
        lo = np.where(big, mid, lo)
        hi = np.where(big, hi, mid)
    return 0.5 * (lo + hi)


def make_pairs(n, thr, seed):
    """-> (dets [n,2,9] float32 with scores 0.9 / 0.8, family [n] int)."""
    rng = np.random.default_rng(seed)
    fam = np.arange(n) % 10
    w = np.exp(rng.uniform(np.log(1.0), np.log(2000.0), n))
    asp = np.exp(rng.uniform(0.0, np.log(50.0), n))
    thin = fam == 4
    asp[thin] = rng.uniform(20.0, 50.0, thin.sum())
    w[thin] = np.exp(rng.uniform(np.log(60.0), np.log(2000.0), thin.sum()))
    h = w / asp
    ang = rng.uniform(0, np.pi, n)
    cx, cy = rng.uniform(0, 2000, n), rng.uniform(0, 2000, n)
    a = _rect(w, h, ang, cx, cy)
    target = thr + rng.uniform(-5e-3, 5e-3, n)
    ux, uy = np.cos(ang), np.sin(ang)
    s = (1 - target) / (1 + target) * w
    b = _shift(a, s * ux, s * uy)

    k6 = np.nonzero(fam == 6)[0]
    if len(k6):
        d = rng.uniform(-0.03, 0.03, len(k6))
        b0 = _rect(w[k6], h[k6], ang[k6] + d, cx[k6], cy[k6])
        t = _bisect_shift(a[k6], b0, ux[k6], uy[k6], np.zeros(len(k6)), 1.5 * w[k6], target[k6])
        b[k6] = _shift(b0, t * ux[k6], t * uy[k6])

    k7 = np.nonzero(fam == 7)[0]
    if len(k7):
        m = len(k7)
        ww, hh = np.maximum(w[k7], 4.0), np.maximum(h[k7], 2.0)
        eps = np.where(rng.random(m) < 0.5, 1e-3, -1e-3)           # +: inside the line (reflex), -: barely convex
        # local frame: P0 (-w/2,-h/2), P1 (w/2,-h/2), P2 (0,h/2), P3 = midpoint(P2,P0) moved eps along the inward normal
        p = np.zeros((m, 4, 2))
        p[:, 0] = np.stack([-ww / 2, -hh / 2], 1)
        p[:, 1] = np.stack([ww / 2, -hh / 2], 1)
        p[:, 2] = np.stack([np.zeros(m), hh / 2], 1)
        e = p[:, 0] - p[:, 2]
        nrm = np.stack([-e[:, 1], e[:, 0]], 1)          # left normal of P2->P0 = inward for a CCW triangle
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        p[:, 3] = 0.5 * (p[:, 0] + p[:, 2]) + eps[:, None] * nrm
        ca, sa = np.cos(ang[k7]), np.sin(ang[k7])
        q = np.empty((m, 8))
        q[:, 0::2] = cx[k7, None] + p[:, :, 0] * ca[:, None] - p[:, :, 1] * sa[:, None]
        q[:, 1::2] = cy[k7, None] + p[:, :, 0] * sa[:, None] + p[:, :, 1] * ca[:, None]
        a[k7] = q
        t = _bisect_shift(q, q, ux[k7], uy[k7], np.zeros(m), ww, target[k7])
        b[k7] = _shift(q, t * ux[k7], t * uy[k7])

    for f in (8, 9):
        k = np.nonzero(fam == f)[0]
        if not len(k):
            continue
        m = len(k)
        w2 = w[k] * np.exp(rng.uniform(-0.7, 0.7, m))
        h2 = w2 / np.exp(rng.uniform(0.0, np.log(6.0), m))
        ang_a, ang_b = ang[k].copy(), rng.uniform(0, np.pi, m)
        if f == 8:
            # half of the family axis-aligned: the hull IS the box, so the upper bound equals the true IoU and the pair
            # sits exactly on the bound's decision edge
            ax = rng.random(m) < 0.5
            ang_a[ax] = 0.0
            ang_b[ax] = 0.0
            # similar sizes, so that the IoU of the centred pair exceeds any of the tested thresholds
            w2[ax] = w[k][ax] * np.exp(rng.uniform(-0.15, 0.15, int(ax.sum())))
            h2[ax] = np.maximum(h[k], w[k] / 6.0)[ax] * np.exp(rng.uniform(-0.15, 0.15, int(ax.sum())))
        a[k] = _rect(w[k], np.maximum(h[k], w[k] / 6.0), ang_a, cx[k], cy[k])
        r = rng.uniform(0.0, 1.0, m) * 0.5 * (w[k] + w2)
        phi = rng.uniform(0, 2 * np.pi, m)
        b0 = _rect(w2, h2, ang_b, cx[k], cy[k])
        if f == 9:
            b[k] = _shift(b0, r * np.cos(phi), r * np.sin(phi))
        else:
            # slide b outwards until the hull-overlap bound ub / (A + B - ub) is near the threshold
            A, B = w[k] * np.maximum(h[k], w[k] / 6.0), w2 * h2
            lo, hi = np.zeros(m), 1.2 * (w[k] + w2)
            uxx, uyy = np.cos(phi), np.sin(phi)
            tgt = thr + rng.uniform(-5e-3, 5e-3, m)
            for _ in range(18):
                mid = 0.5 * (lo + hi)
                bb = _shift(b0, mid * uxx, mid * uyy)
                ow = np.minimum(a[k][:, 0::2].max(1), bb[:, 0::2].max(1)) - np.maximum(a[k][:, 0::2].min(1), bb[:, 0::2].min(1))
                oh = np.minimum(a[k][:, 1::2].max(1), bb[:, 1::2].max(1)) - np.maximum(a[k][:, 1::2].min(1), bb[:, 1::2].min(1))
                ub = np.minimum(np.maximum(ow, 0) * np.maximum(oh, 0), np.minimum(A, B))
                big = ub / (A + B - ub) > tgt
                lo, hi = np.where(big, mid, lo), np.where(big, hi, mid)
            b[k] = _shift(b0, lo * uxx, lo * uyy)

    k5 = np.nonzero(fam == 5)[0]
    if len(k5):
        off = rng.integers(0, 16, len(k5)) * (rng.uniform(100.0, 3000.0, len(k5)) + 1.0)
        a[k5] += off[:, None]
        b[k5] += off[:, None]

    dets = np.zeros((n, 2, 9), np.float32)
    dets[:, 0, :8], dets[:, 1, :8] = a.astype(np.float32), b.astype(np.float32)
    dets[:, 0, 8], dets[:, 1, 8] = 0.9, 0.8
    return dets, fam


def expected_keep_counts(dets, thr):
    """2-box images: the second (lower score) box survives iff iou_poly(box0, box1) <= thr (fp64, on the float32 rows)."""
    iou = oracle.iou_poly_pairs(np.ascontiguousarray(dets[:, 0, :8]), np.ascontiguousarray(dets[:, 1, :8]))
    return np.where(iou > thr, 1, 2).astype(np.int32), iou


# ----------------------------------------------------------------------------------------------------------------------
# General quads: reflex vertices and self-intersecting "bow-ties" (what a head with untrained weights emits).  The HIP
# path decides these with the winding-number form of the fast path (two fan triangles per quad, four convex triangle
# pairs, poly_nms.hip `fast_decision`) and everything within 1e-3 of the threshold with the reference-order clip.
#
# Types (index % 5):
#   0  four random points in a rotated w x h box (convex, dart or bow-tie, whatever comes out) + shifted copy
#   1  dart: a triangle whose fourth vertex is pushed 20-80 % of the way towards the opposite vertex + shifted copy
#   2  bow-tie: a rectangle with two vertices swapped (0,1,3,2) and jittered + shifted copy
#   3  type 0 / 1 / 2 shape against a plain rotated rectangle
#   4  type 0 with one vertex within 0.5 px of the coordinate ORIGIN (the reference fans from the origin: short cut lines)
# Half of the pairs aim at thr +- 5e-3, the other half at thr +- 5e-2 (so that the fast path gets to decide).
def is_convex(q):
    """[n,8] -> bool [n]: strictly convex (all corner cross products of one sign)."""
    x, y = q[:, 0::2].astype(np.float64), q[:, 1::2].astype(np.float64)
    ex, ey = np.roll(x, -1, 1) - x, np.roll(y, -1, 1) - y
    cr = ex * np.roll(ey, -1, 1) - ey * np.roll(ex, -1, 1)
    return (cr > 0).all(1) | (cr < 0).all(1)


def make_general_pairs(n, thr, seed):
    """-> (dets [n,2,9] float32 with scores 0.9 / 0.8, type [n] int)."""
    rng = np.random.default_rng(seed)
    typ = np.arange(n) % 5
    w = np.exp(rng.uniform(np.log(4.0), np.log(1500.0), n))
    h = w / np.exp(rng.uniform(0.0, np.log(8.0), n))
    ang = rng.uniform(0, 2 * np.pi, n)
    cx, cy = rng.uniform(0, 2000, n), rng.uniform(0, 2000, n)
    ca, sa = np.cos(ang), np.sin(ang)

    def place(p):                              # local [n,4,2] -> rotated + translated [n,8]
        q = np.empty((len(p), 8))
        q[:, 0::2] = cx[:, None] + p[:, :, 0] * ca[:, None] - p[:, :, 1] * sa[:, None]
        q[:, 1::2] = cy[:, None] + p[:, :, 0] * sa[:, None] + p[:, :, 1] * ca[:, None]
        return q

    shape = np.where(typ == 3, rng.integers(0, 3, n), np.where(typ == 4, 0, typ))
    p = np.zeros((n, 4, 2))
    r4 = rng.uniform(-0.5, 0.5, (n, 4, 2)) * np.stack([w, h], 1)[:, None, :]
    p[shape == 0] = r4[shape == 0]
    k = shape == 1                             # dart
    tri = np.stack([np.stack([-w / 2, -h / 2], 1), np.stack([w / 2, -h / 2], 1), np.stack([0 * w, h / 2], 1)], 1)
    depth = rng.uniform(0.2, 0.8, n)
    mid = 0.5 * (tri[:, 0] + tri[:, 1])
    inner = mid + depth[:, None] * (tri[:, 2] - mid)
    dart = np.stack([tri[:, 0], inner, tri[:, 1], tri[:, 2]], 1)       # reflex vertex between the base corners
    p[k] = dart[k]
    k = shape == 2                             # bow-tie
    rect = np.stack([np.stack([-w / 2, -h / 2], 1), np.stack([w / 2, -h / 2], 1), np.stack([w / 2, h / 2], 1), np.stack([-w / 2, h / 2], 1)], 1)
    bow = rect[:, [0, 1, 3, 2]] + rng.uniform(-0.15, 0.15, (n, 4, 2)) * np.stack([w, h], 1)[:, None, :]
    p[k] = bow[k]
    a = place(p)
    b0 = a.copy()
    k3 = typ == 3
    ang2 = rng.uniform(0, np.pi, n)
    w2, h2 = w * np.exp(rng.uniform(-0.4, 0.4, n)), h * np.exp(rng.uniform(-0.4, 0.4, n))
    b0[k3] = _rect(w2, h2, ang2, cx, cy)[k3]
    k4 = np.nonzero(typ == 4)[0]
    if len(k4):
        off = a[k4, 0:2] - rng.uniform(-0.5, 0.5, (len(k4), 2))
        a[k4, 0::2] -= off[:, 0:1]; a[k4, 1::2] -= off[:, 1:2]
        b0[k4, 0::2] -= off[:, 0:1]; b0[k4, 1::2] -= off[:, 1:2]
    wide = (np.arange(n) // 5) % 2 == 1
    target = thr + np.where(wide, rng.uniform(-5e-2, 5e-2, n), rng.uniform(-5e-3, 5e-3, n))
    phi = rng.uniform(0, 2 * np.pi, n)
    ux, uy = np.cos(phi), np.sin(phi)
    t = _bisect_shift(a, b0, ux, uy, np.zeros(n), 1.5 * (w + w2), target, iters=22)
    b = _shift(b0, t * ux, t * uy)
    koff = np.nonzero((np.arange(n) // 10) % 4 == 3)[0]            # a quarter at fp32 class-offset magnitudes (not type 4)
    koff = koff[typ[koff] != 4]
    off = rng.integers(1, 16, len(koff)) * (rng.uniform(100.0, 3000.0, len(koff)) + 1.0)
    a[koff] += off[:, None]
    b[koff] += off[:, None]
    dets = np.zeros((n, 2, 9), np.float32)
    dets[:, 0, :8], dets[:, 1, :8] = a.astype(np.float32), b.astype(np.float32)
    dets[:, 0, 8], dets[:, 1, 8] = 0.9, 0.8
    return dets, typ
