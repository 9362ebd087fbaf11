"""VOC-style AP for oriented boxes with the polygon IoU on the GPU.

Counterpart of dafne/evaluation/voc_eval.py: `voc_ap(rec, prec, use_07_metric)` (:7-38) and
`voc_eval(detpath, annopath, imagesetfile, classname, ovthresh, use_07_metric, parse_gt)` (:41-224)
with the same arguments and the same return (rec, prec, ap, data_scores_overlap).

The reference matches detections to ground truth one at a time and calls SWIG `polyiou.iou_poly` for
every ground-truth box whose axis-aligned hull overlaps the detection's (:150-189).  Which pairs are
clipped does not depend on the matching state, so here ALL (ground truth, detection) pairs of the class
that pass the hull test are collected first and go through `dafne_poly_iou_pairs_hip` in one launch
(bit-identical to polyiou.cpp); the greedy TP/FP marking then runs on the host over the cached values.
"""
import numpy as np
import torch

from .. import _lib


def poly_iou_pairs_device(p, q, device=None):
    """fp64 IoU of rows p[i] vs q[i] ([n,8] each) on the GPU; returns a float64 numpy array."""
    p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 8)
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 8)
    n = p.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.float64)
    L = _lib.load()
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.device(dev):
        dp, dq = torch.from_numpy(p).to(dev), torch.from_numpy(q).to(dev)
        out = torch.empty(n, dtype=torch.float64, device=dev)
        _lib.check(L.dafne_poly_iou_pairs_hip(_lib.ptr(dp), _lib.ptr(dq), n, _lib.ptr(out), _lib.current_stream()),
                   "dafne_poly_iou_pairs_hip")
        return out.cpu().numpy()


def voc_ap(rec, prec, use_07_metric=False):
    """AP from a precision/recall curve: VOC07 11-point interpolation or the exact area."""
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            sel = rec >= t
            p = np.max(prec[sel]) if np.sum(sel) != 0 else 0
            ap = ap + p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def _hull(b):
    return b[..., 0::2].min(-1), b[..., 1::2].min(-1), b[..., 0::2].max(-1), b[..., 1::2].max(-1)


def hull_candidates(bbgt, bb):
    """Indices of the ground-truth rows whose hull IoU (with the +1 pixel convention) with the
    detection is > 0 (voc_eval.py:150-178)."""
    gx0, gy0, gx1, gy1 = _hull(bbgt)
    bx0, by0, bx1, by1 = _hull(bb)
    iw = np.maximum(np.minimum(gx1, bx1) - np.maximum(gx0, bx0) + 1.0, 0.0)
    ih = np.maximum(np.minimum(gy1, by1) - np.maximum(gy0, by0) + 1.0, 0.0)
    inters = iw * ih
    uni = (bx1 - bx0 + 1.0) * (by1 - by0 + 1.0) + (gx1 - gx0 + 1.0) * (gy1 - gy0 + 1.0) - inters
    return np.where(inters / uni > 0)[0]


def voc_eval(detpath, annopath, imagesetfile, classname, ovthresh=0.5, use_07_metric=False, parse_gt=None,
             iou_pairs=poly_iou_pairs_device):
    with open(imagesetfile, "r") as f:
        imagenames = [x.strip() for x in f.readlines()]
    class_recs = {}
    npos = 0
    for name in imagenames:
        objs = [o for o in parse_gt(annopath.format(name)) if o["name"] == classname]
        bbox = np.array([o["bbox"] for o in objs])
        difficult = np.array([o["difficult"] for o in objs]).astype(bool)
        npos += int(sum(~difficult))
        class_recs[name] = {"bbox": bbox, "difficult": difficult, "det": [False] * len(objs)}

    with open(detpath.format(classname), "r") as f:
        rows = [x.strip().split(" ") for x in f.readlines()]
    image_ids = [r[0] for r in rows]
    confidence = np.array([float(r[1]) for r in rows])
    BB = np.array([[float(z) for z in r[2:]] for r in rows])
    sorted_ind = np.argsort(-confidence)
    if BB.shape[0] > 0:
        BB = BB[sorted_ind, :]
    image_ids = [image_ids[x] for x in sorted_ind]
    nd = len(image_ids)

    # ---- pass 1: every (ground truth, detection) pair that the matching loop will clip
    cand = []
    pair_gt, pair_bb = [], []
    for d in range(nd):
        bbgt = class_recs[image_ids[d]]["bbox"].astype(float)
        idx = hull_candidates(bbgt, BB[d, :].astype(float)) if bbgt.size > 0 else np.zeros(0, dtype=np.int64)
        cand.append(idx)
        if idx.size:
            pair_gt.append(bbgt[idx])
            pair_bb.append(np.repeat(BB[d:d + 1].astype(float), idx.size, 0))
    if pair_gt:
        ious = iou_pairs(np.concatenate(pair_gt, 0), np.concatenate(pair_bb, 0))
    else:
        ious = np.zeros(0, dtype=np.float64)

    # ---- pass 2: greedy marking in descending confidence (voc_eval.py:132-205)
    tp, fp = np.zeros(nd), np.zeros(nd)
    data_scores_overlap = []
    off = 0
    for d in range(nd):
        rec_d = class_recs[image_ids[d]]
        conf = confidence[d]          # as in the reference: the UNSORTED array is indexed here (:135)
        ovmax, jmax = -np.inf, -1
        k = cand[d].size
        if k:
            ov = ious[off:off + k]
            off += k
            ovmax = np.max(ov)
            jmax = cand[d][np.argmax(ov)]
        if ovmax > ovthresh:
            if not rec_d["difficult"][jmax]:
                if not rec_d["det"][jmax]:
                    tp[d] = 1.0
                    rec_d["det"][jmax] = 1
                    data_scores_overlap.append([conf, ovmax, 1, classname])
                else:
                    fp[d] = 1.0
                    data_scores_overlap.append([conf, ovmax, 0, classname])
        else:
            fp[d] = 1.0

    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric), data_scores_overlap
