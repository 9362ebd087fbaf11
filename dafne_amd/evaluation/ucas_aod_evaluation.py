"""UCAS-AOD scoring (BASELINE config 4's dataset): Task1 files + VOC07 AP of "car" and "airplane" on the device IoU kernel.

Counterpart of dafne/evaluation/ucas_aod_evaluation.py (parse_gt :91-103, _generate_task_1_files :106-150,
do_ucas_aod_evaluation :280-347) and of the annotation reader it goes through, dafne/data/datasets/ucas_aod.py:
  load_annotation   (:39-54)   Annotations/<image>.txt, one object per line: class name + 8 corner coordinates (+ fields that are
                               not read); the coordinates land in an int32 array, i.e. they are TRUNCATED toward zero
  parse_annotation  (:75-128)  drops boxes whose axis-aligned hull is <= 2 px wide or high or has an aspect ratio >= 30
                               (the reference also opens the image for its size; the ground truth does not need it)
Every kept object: {"name", "difficult": 0, "bbox": the 8 truncated coordinates}.
"""
import os

import numpy as np

from .inference import DafneEvaluator
from .task1 import score_task1, write_task1_files

classnames = ["car", "airplane"]


def load_annotation(root_dir, img_id):
    boxes, names = [], []
    with open(os.path.join(root_dir, "Annotations", img_id + ".txt"), "r", encoding="utf-8-sig") as f:
        for line in f.read().split("\n"):
            if len(line) == 0:
                continue
            tok = line.split()
            names.append(tok[0])
            boxes.append([float(v) for v in tok[1:9]])
    return {"boxes": np.array(boxes, dtype=np.float64).astype(np.int32).reshape(-1, 8), "names": names}


def parse_gt(annopath):
    anno_dir, fname = os.path.split(annopath)
    root_dir = os.path.split(anno_dir)[0]
    anno = load_annotation(root_dir, fname[:-4])
    objs = []
    for box, name in zip(anno["boxes"], anno["names"]):
        if name not in classnames:                 # "__background__" (label -1 in the reference): skipped
            continue
        w = np.abs(box[0::2].max() - box[0::2].min())
        h = np.abs(box[1::2].max() - box[1::2].min())
        ar = np.maximum(w / (h + 1e-16), h / (w + 1e-16))
        if not ((w > 2) & (h > 2) & (ar < 30)):
            continue
        objs.append({"name": name, "difficult": 0, "bbox": box.tolist()})
    return objs


def _generate_task_1_files(metadata, predictions, output_folder, task1_dir, cfg):
    write_task1_files(predictions, output_folder, task1_dir, classnames, cfg)


def do_ucas_aod_evaluation(dataset_name, metadata, predictions, output_folder, logger, results, cfg):
    task1_dir = os.path.join(output_folder, "Task1")
    os.makedirs(task1_dir, exist_ok=True)
    _generate_task_1_files(metadata, predictions, output_folder, task1_dir, cfg)
    annopath = os.path.join(metadata.root_dir, "Annotations", "{:s}.txt")
    score_task1(classnames, task1_dir, annopath, output_folder, parse_gt, cfg, results)


class UcasAodEvaluator(DafneEvaluator):
    def _eval_predictions(self, predictions):
        do_ucas_aod_evaluation(dataset_name=self._dataset_name, metadata=self._metadata, predictions=predictions,
                               output_folder=self._output_dir, logger=self._logger, results=self._results, cfg=self._cfg)
