"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's algorithms for the hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``dafne_amd/`` does (tests/test_abi.py greps for it).

  poly_oracle.c        fp64 quad IoU + greedy polygon NMS      (C, liboracle.so)
  ref_shim.cpp + _ref  the reference's own polyiou.cpp, compiled here
  postprocess.py       decode / corner sort / class offsets / cap / rescale (numpy)
  evaluation.py        Task1 writer, tile ResultMerge, voc_eval mAP (numpy; next rows of SURVEY 8f)
  model.py             ResNet-FPN + DAFNe head in plain torch fp32
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(force=False):
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "poly_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    need_ref = os.path.isdir("/root/reference") and not os.path.exists(
        os.path.join(_HERE, "_ref", "libpolyiou_ref.so"))
    if force or stale or need_ref:
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        dp = ctypes.POINTER(ctypes.c_double)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int64)
        L.orc_iou_poly.restype = ctypes.c_double
        L.orc_iou_poly.argtypes = [dp, dp]
        L.orc_iou_poly_pairs.restype = None
        L.orc_iou_poly_pairs.argtypes = [dp, dp, ctypes.c_int64, dp]
        L.orc_iou_poly_pairs_f32.restype = None
        L.orc_iou_poly_pairs_f32.argtypes = [fp, fp, ctypes.c_int64, dp]
        L.orc_score_order.restype = None
        L.orc_score_order.argtypes = [fp, ctypes.c_int64, ip]
        for name in ("orc_poly_nms", "orc_poly_nms_fast"):
            f = getattr(L, name)
            f.restype = ctypes.c_int64
            f.argtypes = [fp, ctypes.c_int64, ctypes.c_double, ip]
        L.orc_poly_nms_f64.restype = ctypes.c_int64
        L.orc_poly_nms_f64.argtypes = [dp, ctypes.c_int64, ctypes.c_double, ctypes.c_int, ip]
        L.orc_build_dets9.restype = None
        L.orc_build_dets9.argtypes = [fp, fp, ip, ctypes.c_int64, fp]
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own polyiou.cpp (oracle/_ref), or None if never built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libpolyiou_ref.so")
        if not os.path.exists(path):
            try:
                build()
            except Exception:
                pass
        if not os.path.exists(path):
            return None
        R = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        R.ref_iou_poly.restype = ctypes.c_double
        R.ref_iou_poly.argtypes = [dp, dp]
        R.ref_iou_poly_pairs.restype = None
        R.ref_iou_poly_pairs.argtypes = [dp, dp, ctypes.c_long, dp]
        _REF = R
    return _REF


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def iou_poly(p, q):
    """fp64 IoU of two quads given as 8 numbers each (polyiou.cpp:112)."""
    p = np.ascontiguousarray(p, dtype=np.float64).reshape(8)
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(8)
    return float(lib().orc_iou_poly(_dp(p), _dp(q)))


def iou_poly_pairs(p, q):
    """Row-wise IoU of [n,8] vs [n,8]; float32 input is widened exactly."""
    p = np.ascontiguousarray(p)
    q = np.ascontiguousarray(q)
    n = p.shape[0]
    out = np.empty(n, dtype=np.float64)
    if n == 0:
        return out
    if p.dtype == np.float32 and q.dtype == np.float32:
        lib().orc_iou_poly_pairs_f32(_fp(p), _fp(q), n, _dp(out))
    else:
        p = np.ascontiguousarray(p, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        lib().orc_iou_poly_pairs(_dp(p), _dp(q), n, _dp(out))
    return out


def ref_iou_poly_pairs(p, q):
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libpolyiou_ref.so is not built")
    p = np.ascontiguousarray(p, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    out = np.empty(p.shape[0], dtype=np.float64)
    if p.shape[0]:
        R.ref_iou_poly_pairs(_dp(p), _dp(q), p.shape[0], _dp(out))
    return out


def poly_nms_f64(dets9, thresh, strict_hbb=True):
    """ResultMerge NMS on float64 [M,9] rows (py_cpu_nms_poly_fast / py_cpu_nms_poly); kept row
    indices in descending-score order."""
    dets9 = np.ascontiguousarray(dets9, dtype=np.float64).reshape(-1, 9)
    keep = np.empty(dets9.shape[0], dtype=np.int64)
    n = lib().orc_poly_nms_f64(_dp(dets9), dets9.shape[0], float(thresh), int(bool(strict_hbb)), _ip(keep)) \
        if dets9.shape[0] else 0
    return keep[:n].tolist()


def score_order(dets9):
    dets9 = np.ascontiguousarray(dets9, dtype=np.float32).reshape(-1, 9)
    order = np.empty(dets9.shape[0], dtype=np.int64)
    if dets9.shape[0]:
        lib().orc_score_order(_fp(dets9), dets9.shape[0], _ip(order))
    return order


def poly_nms(dets9, thresh, fast=False):
    """Greedy polygon NMS on float32 [M,9]; returns kept row indices (list[int]),
    descending score.  ``fast`` uses the guarded hull pre-filter."""
    dets9 = np.ascontiguousarray(dets9, dtype=np.float32).reshape(-1, 9)
    m = dets9.shape[0]
    keep = np.empty(max(m, 1), dtype=np.int64)
    f = lib().orc_poly_nms_fast if fast else lib().orc_poly_nms
    n = f(_fp(dets9), m, float(thresh), _ip(keep))
    return [int(v) for v in keep[:n]]


def build_dets9(boxes8, scores, classes):
    """nms.py:74-90 in float32: class 5->4, class offsets, hstack with scores."""
    boxes8 = np.ascontiguousarray(boxes8, dtype=np.float32).reshape(-1, 8)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    classes = np.ascontiguousarray(classes, dtype=np.int64).reshape(-1)
    m = boxes8.shape[0]
    out = np.empty((m, 9), dtype=np.float32)
    if m:
        lib().orc_build_dets9(_fp(boxes8), _fp(scores), _ip(classes), m, _fp(out))
    return out
