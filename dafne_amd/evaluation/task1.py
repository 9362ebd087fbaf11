"""What the reference's four dataset evaluations share (dafne/evaluation/{dota,hrsc,ucas_aod,icdar15}_evaluation.py are one
file edited four times): the Task1 result files and the per-class voc_eval loop.

  write_task1_files   `_generate_task_1_files` (dota_evaluation.py:110-164, hrsc_evaluation.py:104-152): one
                      `Task1_<class>.txt` per class with `<image> <score> x1 y1 .. x4 y4` lines + imageset.txt
  score_task1         the loop of `do_*_evaluation` (dota_evaluation.py:370-414, hrsc_evaluation.py:293-347): voc_eval per
                      class (VOC07 11-point, TEST.IOU_TH) on the device IoU kernel, scores_overlap.csv, results.txt,
                      results["task1"] = {class: ap, ..., "map": mean}
The plotting / visualisation halves (PR-curve PNGs, sample images) are outside the path.
"""
import os
from collections import OrderedDict

import numpy as np

from .voc_eval import voc_eval


def task1_scores(scores, centerness, cfg):
    """Class confidence written to the Task1 files: the reported score is sqrt(cls * ctr) unless
    CENTERNESS_USE_IN_SCORE, so score^2 / ctr recovers cls (dota_evaluation.py:136-140); fp32 like the reference."""
    d = cfg.MODEL.DAFNE
    if d.CENTERNESS != "none" and not d.CENTERNESS_USE_IN_SCORE:
        s = np.asarray(scores, dtype=np.float32)
        return (s ** 2) / np.asarray(centerness, dtype=np.float32)
    return np.asarray(scores, dtype=np.float32)


def _host(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def write_task1_files(predictions, output_folder, task1_dir, classnames, cfg, require_square=False, skip_labels=()):
    """predictions: per-image dicts {"file_name", "height", "width", "corners" [K,8], "labels" [K], "scores" [K],
    "centerness" [K]} (DafneEvaluator's, or evaluation.gather.to_predictions + file names).  The image key of a line is
    the file's base name without its 4-character extension, as in the reference.  require_square: DOTA tiles only
    (dota_evaluation.py:128); skip_labels: labels left out (DOTA-1.5's container-crane, :149-151)."""
    files = {i: open(os.path.join(task1_dir, "Task1_%s.txt" % c), "w") for i, c in enumerate(classnames)}
    names = set()
    try:
        for p in predictions:
            fname = p["file_name"].split("/")[-1][:-4]
            names.add(fname)
            if require_square:
                assert p["height"] == p["width"]
            corners = _host(p["corners"]).astype(np.float32).reshape(-1, 8)
            labels = _host(p["labels"]).reshape(-1)
            scores = task1_scores(_host(p["scores"]), _host(p["centerness"]), cfg)
            for i in range(corners.shape[0]):
                label = int(labels[i])
                if label in skip_labels:
                    continue
                c = corners[i]
                files[label].write("%s %.4f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n"
                                   % (fname, scores[i], c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]))
    finally:
        for f in files.values():
            f.close()
    with open(os.path.join(output_folder, "imageset.txt"), "w") as f:
        f.write("\n".join(list(names)))


def score_task1(classnames, task1_dir, annopath, output_folder, parse_gt, cfg, results):
    detpath = os.path.join(task1_dir, "Task1_{:s}.txt")
    imagesetfile = os.path.join(output_folder, "imageset.txt")
    task_results = OrderedDict()
    mean_ap = 0.0
    rows = []
    for c in classnames:
        rec, prec, ap, so = voc_eval(detpath, annopath, imagesetfile, c, ovthresh=cfg.TEST.IOU_TH,
                                     use_07_metric=True, parse_gt=parse_gt)
        mean_ap += ap
        task_results[c] = ap
        rows += so
    np.savetxt(fname=os.path.join(output_folder, "scores_overlap.csv"), X=rows, delimiter=",", fmt="%s")
    task_results["map"] = mean_ap / len(classnames)
    results["task1"] = task_results
    with open(os.path.join(output_folder, "results.txt"), "w") as f:
        for k, v in task_results.items():
            f.write(f"{k: <18}: {v:2.4f}\n")
    return task_results
