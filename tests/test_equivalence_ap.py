"""CPU: `equivalence_ap` (AP of one detector's output against another's as ground truth, dafne/evaluation/voc_eval.py:41-224's
scoring) on hand-made sets, polygon IoU from the oracle instead of the GPU."""
import numpy as np

import oracle
from dafne_amd.evaluation.equivalence import equivalence_ap


def _iou_pairs(p, q):
    return np.asarray(oracle.iou_poly_pairs(np.asarray(p, dtype=np.float64).reshape(-1, 8), np.asarray(q, dtype=np.float64).reshape(-1, 8)))


def _boxes(rng, n):
    c = rng.uniform(50, 950, (n, 2))
    w, h = rng.uniform(10, 60, n), rng.uniform(10, 60, n)
    a = rng.uniform(0, np.pi, n)
    dx = np.stack([-w, w, w, -w], 1) / 2
    dy = np.stack([-h, -h, h, h], 1) / 2
    x = c[:, :1] + dx * np.cos(a)[:, None] - dy * np.sin(a)[:, None]
    y = c[:, 1:] + dx * np.sin(a)[:, None] + dy * np.cos(a)[:, None]
    return np.stack([x, y], 2).reshape(n, 8)


def _set(rng, n, ncls=3):
    return {"pred_corners": _boxes(rng, n), "scores": rng.uniform(0.05, 1, n), "pred_classes": rng.integers(0, ncls, n)}


def test_identical_sets_score_one():
    rng = np.random.default_rng(0)
    refs = [_set(rng, 40) for _ in range(3)]
    r = equivalence_ap(refs, refs, iou_pairs=_iou_pairs)
    assert abs(r["iou_0.50"]["mean"] - 1.0) < 1e-12 and abs(r["iou_0.75"]["mean"] - 1.0) < 1e-12 and r["iou_0.50"]["classes"] == 3
    assert r["reference_boxes"] == 120 and r["detections"] == 120 and r["images"] == 3


def test_missing_shifted_and_spurious_detections_cost_ap():
    rng = np.random.default_rng(1)
    refs = [_set(rng, 60) for _ in range(2)]
    # (a) a third of the reference boxes missing: recall stops at 2/3 -> VOC07 AP = 7/11 (recall points 0 .. 0.6 at precision 1)
    dets = [{k: v[:40] for k, v in r.items()} for r in refs]
    a = equivalence_ap(dets, refs, iou_pairs=_iou_pairs)
    assert 0.5 < a["iou_0.50"]["mean"] < 0.8
    # (b) every box shifted by 1.5 px: still a match at IoU 0.5 (boxes are >= 10 px wide), mostly lost at 0.75 for small ones
    dets = [dict(r, pred_corners=r["pred_corners"] + 1.5) for r in refs]
    b = equivalence_ap(dets, refs, iou_pairs=_iou_pairs)
    assert b["iou_0.50"]["mean"] > 0.95 and b["iou_0.75"]["mean"] < b["iou_0.50"]["mean"]
    # (c) spurious high-score detections in front of the true ones: precision drops at every recall level
    sp = _set(np.random.default_rng(2), 30)
    dets = [{"pred_corners": np.concatenate([r["pred_corners"], sp["pred_corners"]]), "scores": np.concatenate([r["scores"], sp["scores"] + 1.0]),
             "pred_classes": np.concatenate([r["pred_classes"], sp["pred_classes"]])} for r in refs]
    c = equivalence_ap(dets, refs, iou_pairs=_iou_pairs)
    assert c["iou_0.50"]["mean"] < 0.9
    # a class present only among the detections is skipped (no ground truth: AP undefined)
    dets = [dict(r, pred_classes=np.where(np.arange(len(r["scores"])) == 0, 7, r["pred_classes"])) for r in refs]
    e = equivalence_ap(dets, refs, iou_pairs=_iou_pairs)
    assert 7 not in e["iou_0.50"]["per_class"]
