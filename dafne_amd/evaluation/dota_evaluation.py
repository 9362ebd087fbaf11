"""DOTA Task1 files, tile merge and mAP on the MI355X kernels.

Counterpart of the result-writing half of dafne/evaluation/dota_evaluation.py:
  parse_gt                 (:60-107)   DOTA labelTxt parser
  _generate_task_1_files   (:110-164)  per-class `Task1_<class>.txt` + imageset.txt
  run_merge                (:181-184)  tile ResultMerge  -> result_merge.mergebypoly (device NMS)
  do_dota_evaluation       (:308-414)  voc_eval per class (VOC07 11-point) -> {"task1": {class: ap, "map": ...}}
The plotting / zip / visualisation helpers of the reference (:166-178,187-305) are outside the path.
"""
import os
from collections import OrderedDict

import numpy as np

from .result_merge import mergebypoly
from .voc_eval import voc_eval

CLASSNAMES_DOTA_1_0 = ["plane", "baseball-diamond", "bridge", "ground-track-field", "small-vehicle", "large-vehicle",
                       "ship", "tennis-court", "basketball-court", "storage-tank", "soccer-ball-field", "roundabout",
                       "harbor", "swimming-pool", "helicopter"]


def parse_gt(filename):
    """One labelTxt file -> [{"name", "difficult", "bbox": 8 floats}]; lines with fewer than 9 fields
    (the imagesource/gsd header) are skipped; 9 fields mean difficult = 0."""
    objects = []
    with open(filename, "r") as f:
        for line in f:
            tok = line.strip().split(" ")
            if len(tok) < 9:
                continue
            obj = {"name": tok[8]}
            if len(tok) == 9:
                obj["difficult"] = 0
            elif len(tok) == 10:
                obj["difficult"] = int(tok[9])
            obj["bbox"] = [float(v) for v in tok[:8]]
            objects.append(obj)
    return objects


def task1_scores(scores, centerness, cfg):
    """Class confidence written to the Task1 files: the reported score is sqrt(cls * ctr) unless
    CENTERNESS_USE_IN_SCORE, so score^2 / ctr recovers cls (:136-140); fp32 like the reference."""
    d = cfg.MODEL.DAFNE
    if d.CENTERNESS != "none" and not d.CENTERNESS_USE_IN_SCORE:
        s = np.asarray(scores, dtype=np.float32)
        return (s ** 2) / np.asarray(centerness, dtype=np.float32)
    return np.asarray(scores, dtype=np.float32)


def _generate_task_1_files(metadata, predictions, output_folder, task1_dir, classnames, cfg):
    """predictions: per-image dicts {"file_name", "height", "width", "corners" [K,8], "labels" [K],
    "scores" [K], "centerness" [K]} (evaluation.gather.to_predictions + file names)."""
    files = {i: open(os.path.join(task1_dir, "Task1_%s.txt" % c), "w") for i, c in enumerate(classnames)}
    names = set()
    skip_crane = bool(cfg.DATASETS.DOTA_REMOVE_CONTAINER_CRANE)
    try:
        for p in predictions:
            fname = p["file_name"].split("/")[-1][:-4]
            names.add(fname)
            assert p["height"] == p["width"]
            corners = np.asarray(p["corners"], dtype=np.float32).reshape(-1, 8)
            labels = np.asarray(p["labels"]).reshape(-1)
            scores = task1_scores(np.asarray(p["scores"]), np.asarray(p["centerness"]), cfg)
            for i in range(corners.shape[0]):
                label = int(labels[i])
                if label == 15 and skip_crane:      # 'container-crane'
                    continue
                c = corners[i]
                files[label].write("%s %.4f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n"
                                   % (fname, scores[i], c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]))
    finally:
        for f in files.values():
            f.close()
    with open(os.path.join(output_folder, "imageset.txt"), "w") as f:
        f.write("\n".join(list(names)))


def run_merge(src, dst):
    mergebypoly(src, dst)


def do_dota_evaluation(dataset_name, metadata, predictions, output_folder, logger, results, cfg):
    classnames = list(CLASSNAMES_DOTA_1_0)
    if "1_5" in dataset_name and not cfg.DATASETS.DOTA_REMOVE_CONTAINER_CRANE:
        classnames.append("container-crane")
    task1_dir = os.path.join(output_folder, "Task1")
    os.makedirs(task1_dir, exist_ok=True)
    _generate_task_1_files(metadata, predictions, output_folder, task1_dir, classnames, cfg)
    if metadata.is_test:
        merged = os.path.join(output_folder, "Task1_merged")
        os.makedirs(merged, exist_ok=True)
        run_merge(src=task1_dir, dst=merged)
        return
    detpath = os.path.join(task1_dir, "Task1_{:s}.txt")
    annopath = os.path.join(metadata.root_dir, "labelTxt", "{:s}.txt")
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
