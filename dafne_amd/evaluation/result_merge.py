"""Tile ResultMerge on the device NMS.

Counterpart of dafne/utils/ResultMerge_multi_process.py (tools/prepare_dota holds the same file):
Task1_<class>.txt files with one detection per line on 1024x1024 TILES

    <orig>__<rate>__<x>___<y> <score> <x1> <y1> ... <x4> <y4>

are mapped back to the original image ((tile coordinate + offset) / rate, :170-176), grouped per
original image, and merged by greedy polygon NMS at 0.1 (:61-121, `py_cpu_nms_poly_fast`) -- the
reference walks every (kept, remaining) pair through SWIG `polyiou.iou_poly` in 16 worker processes.
Here every original image of a file is one row of ONE batched launch of the rotated-NMS kernels of
the hot path (`dafne_poly_nms_f64_batched_hip`: fp64 rows, the reference's hull pre-test as part of
the predicate), so the keep lists are the reference's.

Same entry points and argument meaning: mergebypoly(src, dst), mergebase(src, dst, nms),
mergesingle(dst, nms, fullname), nmsbynamedict(nameboxdict, nms, thresh), poly2origpoly(...).
`nms` is a callable (float64 [M,9], thresh) -> kept indices; the default is the device one.
There is no CPU fallback: without libdafne_amd.so / a GPU the calls raise.
"""
import os
import re

import numpy as np
import torch

from .. import _lib

## the thresh for nms when merge image (ResultMerge_multi_process.py:22)
nms_thresh = 0.1

_XY = re.compile(r"__\d+___\d+")
_RATE = re.compile(r"__([\d+\.]+)__\d+___")
_INT = re.compile(r"\d+")


def _merge_nms_batched(arrays, thresh, strict_hbb=True, device=None):
    """arrays: list of float64 [M_i, 9].  One device launch per size bucket; returns list of keep lists."""
    L = _lib.load()
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    out = [None] * len(arrays)
    todo = [i for i, a in enumerate(arrays) if len(a) > 0]
    for i, a in enumerate(arrays):
        if len(a) == 0:
            out[i] = []
    # buckets of similar size: rows are padded to the bucket's largest image
    todo.sort(key=lambda i: -len(arrays[i]))
    pos = 0
    with torch.cuda.device(dev):
        while pos < len(todo):
            m_cap = len(arrays[todo[pos]])
            n = 1
            while pos + n < len(todo) and n < 256 and (n + 1) * m_cap <= (1 << 18) and \
                    2 * len(arrays[todo[pos + n]]) >= m_cap:
                n += 1
            ids = todo[pos:pos + n]
            pos += n
            host = np.zeros((n, m_cap, 9), dtype=np.float64)
            counts = np.zeros(n, dtype=np.int32)
            for k, i in enumerate(ids):
                a = np.ascontiguousarray(arrays[i], dtype=np.float64).reshape(-1, 9)
                host[k, :a.shape[0]] = a
                counts[k] = a.shape[0]
            d = torch.from_numpy(host).to(dev)
            c = torch.from_numpy(counts).to(dev)
            keep = torch.empty((n, m_cap), dtype=torch.int64, device=dev)
            nk = torch.zeros(n, dtype=torch.int32, device=dev)
            nbytes = L.dafne_poly_nms_f64_workspace_bytes(n, m_cap)
            if nbytes == 0:
                raise _lib.DafneHipError("poly_nms_f64: bad size %d x %d" % (n, m_cap))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.dafne_poly_nms_f64_batched_hip(_lib.ptr(d), _lib.ptr(c), n, m_cap, float(thresh),
                                                        1 if strict_hbb else 0, _lib.ptr(keep), _lib.ptr(nk),
                                                        _lib.ptr(ws), nbytes, 0, _lib.current_stream()),
                       "dafne_poly_nms_f64_batched_hip")
            kh, nh = keep.cpu().numpy(), nk.cpu().numpy()
            for k, i in enumerate(ids):
                out[i] = kh[k, :nh[k]].tolist()
    return out


def py_cpu_nms_poly_fast(dets, thresh):
    """Name kept from the reference (:61); runs on the GPU.  dets: [M,9] float64 (8 coords + score)."""
    return _merge_nms_batched([np.asarray(dets, dtype=np.float64).reshape(-1, 9)], thresh, True)[0]


def py_cpu_nms_poly(dets, thresh):
    """Reference :24-58 (no hull pre-test), on the GPU."""
    return _merge_nms_batched([np.asarray(dets, dtype=np.float64).reshape(-1, 9)], thresh, False)[0]


def poly2origpoly(poly, x, y, rate):
    """Tile coordinates -> original image coordinates (:163-169), same float64 arithmetic."""
    r = float(rate)
    out = []
    for k in range(len(poly) // 2):
        out.append(float(poly[2 * k] + x) / r)
        out.append(float(poly[2 * k + 1] + y) / r)
    return out


def parse_task1_lines(lines):
    """Lines of a Task1_<class>.txt on tiles -> {orig image: [[x1..y4, score], ...]} in file order."""
    boxes = {}
    for raw in lines:
        tok = raw.strip().split(" ")
        sub = tok[0]
        orig = sub.split("__")[0]
        xy = _INT.findall(_XY.findall(sub)[0])
        x, y = int(xy[0]), int(xy[1])
        rate = _RATE.findall(sub)[0]
        det = poly2origpoly([float(v) for v in tok[2:]], x, y, rate)
        det.append(float(tok[1]))
        boxes.setdefault(orig, []).append(det)
    return boxes


def nmsbynamedict(nameboxdict, nms, thresh):
    """Per original image: keep the rows the NMS keeps, in its (descending-score) order (:141-158).
    With the default device NMS all images of the dict go out as batched launches."""
    names = list(nameboxdict)
    if nms in (py_cpu_nms_poly_fast, py_cpu_nms_poly):
        keeps = _merge_nms_batched([np.array(nameboxdict[n], dtype=np.float64) for n in names], thresh,
                                   nms is py_cpu_nms_poly_fast)
    else:
        keeps = [nms(np.array(nameboxdict[n]), thresh) for n in names]
    return {n: [nameboxdict[n][i] for i in k] for n, k in zip(names, keeps)}


def mergesingle(dstpath, nms, fullname):
    """One Task1_<class>.txt: parse, shift/scale, NMS per original image, write (:171-205)."""
    name = os.path.basename(os.path.splitext(fullname)[0])
    with open(fullname, "r") as f:
        boxes = parse_task1_lines(f.readlines())
    merged = nmsbynamedict(boxes, nms, nms_thresh)
    with open(os.path.join(dstpath, name + ".txt"), "w") as f:
        for img, dets in merged.items():
            for det in dets:
                f.write(img + " " + str(det[-1]) + " " + " ".join(map(str, det[0:-1])) + "\n")


def _files_under(root):
    out = []
    for r, _, files in os.walk(root):
        out.extend(os.path.join(r, f) for f in files)
    return out


def mergebase(srcpath, dstpath, nms):
    for fn in _files_under(srcpath):
        mergesingle(dstpath, nms, fn)


def mergebase_parallel(srcpath, dstpath, nms):
    """The reference fans the files out over Pool(16) (:207-214); the device batches need no pool."""
    mergebase(srcpath, dstpath, nms)


def mergebypoly(srcpath, dstpath):
    """srcpath: Task1 files on tiles; dstpath: merged files on original images (:229-243)."""
    mergebase_parallel(srcpath, dstpath, py_cpu_nms_poly_fast)
