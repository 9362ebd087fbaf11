"""oracle/evaluation.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the steps right after the hot path, pinned by
tests/golden/eval_merge.npz (outputs of the reference's own functions run under stubs,
tests/golden/make_golden_eval.py):

  merge_file_lines    dafne/utils/ResultMerge_multi_process.py:171-205 (mergesingle) with
                      py_cpu_nms_poly_fast (:61-121) as the NMS
  voc_eval_lines      dafne/evaluation/voc_eval.py:41-224
  task1_line          dafne/evaluation/dota_evaluation.py:157-158 (number formatting)

Pure-Python loops + the C oracle (oracle.poly_nms_f64, oracle.iou_poly): small cases only.
"""
import re

import numpy as np

from . import iou_poly, poly_nms_f64


def tile_to_orig(tile_name, coords):
    """'<orig>__<rate>__<x>___<y>' + 8 tile coordinates -> (orig, 8 original-image coordinates):
    (c + offset) / rate in float64 (ResultMerge_multi_process.py:163-169,175-187)."""
    orig = tile_name.split("__")[0]
    m = re.search(r"__(\d+)___(\d+)", tile_name)
    x, y = int(m.group(1)), int(m.group(2))
    rate = float(re.findall(r"__([\d+\.]+)__\d+___", tile_name)[0])
    out = []
    for k in range(4):
        out.append(float(coords[2 * k] + x) / rate)
        out.append(float(coords[2 * k + 1] + y) / rate)
    return orig, out


def merge_file_lines(lines, thresh=0.1, strict_hbb=True):
    """Task1 lines on tiles -> merged Task1 lines on original images (file order of first
    appearance for the images, descending score inside an image)."""
    per_img = {}
    for raw in lines:
        tok = raw.strip().split(" ")
        orig, poly = tile_to_orig(tok[0], [float(v) for v in tok[2:10]])
        per_img.setdefault(orig, []).append(poly + [float(tok[1])])
    out = []
    for img, dets in per_img.items():
        keep = poly_nms_f64(np.array(dets, dtype=np.float64), thresh, strict_hbb)
        for i in keep:
            d = dets[i]
            out.append(img + " " + str(d[8]) + " " + " ".join(str(v) for v in d[:8]))
    return out


def voc_eval_lines(det_lines, gt_by_image, classname, ovthresh=0.5, use_07_metric=True):
    """det_lines: Task1 lines of one class; gt_by_image: {image: [(8 floats, name, difficult)]} for
    every image of the image set.  Returns (rec, prec, ap)."""
    recs, npos = {}, 0
    for img, objs in gt_by_image.items():
        sel = [o for o in objs if o[1] == classname]
        recs[img] = {"bbox": [list(map(float, o[0])) for o in sel], "difficult": [bool(o[2]) for o in sel],
                     "det": [False] * len(sel)}
        npos += sum(1 for o in sel if not o[2])
    rows = [ln.strip().split(" ") for ln in det_lines]
    conf = np.array([float(r[1]) for r in rows])
    order = np.argsort(-conf)
    tp, fp = np.zeros(len(rows)), np.zeros(len(rows))
    for d, ri in enumerate(order):
        r = rows[ri]
        bb = [float(v) for v in r[2:10]]
        rec = recs[r[0]]
        best, jbest = -np.inf, -1
        bx0, by0, bx1, by1 = min(bb[0::2]), min(bb[1::2]), max(bb[0::2]), max(bb[1::2])
        for j, g in enumerate(rec["bbox"]):
            gx0, gy0, gx1, gy1 = min(g[0::2]), min(g[1::2]), max(g[0::2]), max(g[1::2])
            iw = max(min(gx1, bx1) - max(gx0, bx0) + 1.0, 0.0)
            ih = max(min(gy1, by1) - max(gy0, by0) + 1.0, 0.0)
            inter = iw * ih
            uni = (bx1 - bx0 + 1.0) * (by1 - by0 + 1.0) + (gx1 - gx0 + 1.0) * (gy1 - gy0 + 1.0) - inter
            if not inter / uni > 0:
                continue
            ov = iou_poly(g, bb)
            if ov > best:            # first maximum wins, like np.argmax
                best, jbest = ov, j
        if best > ovthresh:
            if not rec["difficult"][jbest]:
                if not rec["det"][jbest]:
                    tp[d] = 1.0
                    rec["det"][jbest] = True
                else:
                    fp[d] = 1.0
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            ap += (np.max(prec[rec >= t]) if np.sum(rec >= t) else 0) / 11.0
    else:
        mrec = np.concatenate(([0.0], rec, [1.0]))
        mpre = np.concatenate(([0.0], prec, [0.0]))
        for i in range(mpre.size - 1, 0, -1):
            mpre[i - 1] = max(mpre[i - 1], mpre[i])
        i = np.where(mrec[1:] != mrec[:-1])[0]
        ap = float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))
    return rec, prec, ap


def task1_line(fname, score, corners):
    return "%s %.4f " % (fname, score) + " ".join("%.2f" % float(c) for c in corners)
