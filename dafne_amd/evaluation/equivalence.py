""""Detections equivalent to the reference" in the reference's own metric: VOC AP of one detector's output against ANOTHER
detector's output taken as ground truth (here: the engine -- or the oracle's bf16 emulation -- against the fp32 oracle run from
the same uint8 images).

The scoring is exactly the evaluator's: Task1-style rows `image score x1 y1 .. x4 y4` per class through `voc_eval`
(dafne/evaluation/voc_eval.py:41-224: descending confidence, hull pre-filter, polygon IoU > threshold, first match of a
ground-truth box is the true positive, VOC07 11-point AP as dafne/evaluation/dota_evaluation.py:385-395 calls it), polygon IoU on
the GPU (`dafne_poly_iou_pairs_hip`).  Every reference detection is a non-difficult ground-truth box of its class; classes
without reference detections are skipped (their AP is undefined in voc_eval: 0 / 0)."""
import os
import tempfile

import numpy as np

from .voc_eval import voc_eval, poly_iou_pairs_device


def equivalence_ap(dets, refs, thresholds=(0.5, 0.75), use_07_metric=True, iou_pairs=poly_iou_pairs_device):
    """dets, refs: per image a dict with `pred_corners` [n,8], `scores` [n], `pred_classes` [n] (numpy).  Returns
    {"iou_0.50": {"mean": mAP over the classes, "weighted_mean": the same weighted by the classes' reference boxes (a class with a
    handful of boxes moves `mean` by 1/11 AP steps per box), "per_class": {class id: AP}, "classes": k}, ..., "images": N,
    "reference_boxes": M, "reference_boxes_per_class": {class id: m}, "detections": D}."""
    assert len(dets) == len(refs)
    names = ["img%04d" % i for i in range(len(refs))]
    gt = {}
    classes = set()
    for nme, r in zip(names, refs):
        cls = np.asarray(r["pred_classes"]).astype(np.int64)
        gt[nme] = [{"name": "c%d" % c, "difficult": 0, "bbox": [float(v) for v in b]}
                   for c, b in zip(cls, np.asarray(r["pred_corners"], dtype=np.float64).reshape(-1, 8))]
        classes.update(int(c) for c in cls)
    per_cls_n = {}
    for v in gt.values():
        for o in v:
            c = int(o["name"][1:])
            per_cls_n[c] = per_cls_n.get(c, 0) + 1
    out = {"images": len(refs), "reference_boxes": int(sum(len(v) for v in gt.values())), "reference_boxes_per_class": per_cls_n,
           "detections": int(sum(len(d["scores"]) for d in dets)), "metric": "VOC07 11-point AP" if use_07_metric else "VOC area AP"}
    with tempfile.TemporaryDirectory(prefix="dafne_equiv_") as tmp:
        with open(os.path.join(tmp, "images.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
        for c in sorted(classes):
            with open(os.path.join(tmp, "Task1_c%d.txt" % c), "w") as f:
                for nme, d in zip(names, dets):
                    cls = np.asarray(d["pred_classes"]).astype(np.int64)
                    sel = np.nonzero(cls == c)[0]
                    cor = np.asarray(d["pred_corners"], dtype=np.float64).reshape(-1, 8)
                    sc = np.asarray(d["scores"], dtype=np.float64)
                    for j in sel:
                        f.write("%s %r %s\n" % (nme, float(sc[j]), " ".join(repr(float(v)) for v in cor[j])))
        for thr in thresholds:
            per = {}
            for c in sorted(classes):
                _, _, ap, _ = voc_eval(os.path.join(tmp, "Task1_{:s}.txt"), "{:s}", os.path.join(tmp, "images.txt"), "c%d" % c,
                                       ovthresh=thr, use_07_metric=use_07_metric, parse_gt=lambda nme: gt[nme], iou_pairs=iou_pairs)
                per[c] = float(ap)
            wsum = float(sum(per_cls_n[c] for c in per))
            out["iou_%.2f" % thr] = {"mean": float(np.mean(list(per.values()))) if per else None,
                                     "weighted_mean": float(sum(per[c] * per_cls_n[c] for c in per) / wsum) if per else None,
                                     "per_class": per, "classes": len(per)}
    return out
