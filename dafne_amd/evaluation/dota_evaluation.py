"""DOTA Task1 files, tile merge and mAP on the MI355X kernels.

Counterpart of the result-writing half of dafne/evaluation/dota_evaluation.py:
  parse_gt                 (:60-107)   DOTA labelTxt parser
  _generate_task_1_files   (:110-164)  per-class `Task1_<class>.txt` + imageset.txt
  run_merge                (:181-184)  tile ResultMerge  -> result_merge.mergebypoly (device NMS)
  do_dota_evaluation       (:308-414)  voc_eval per class (VOC07 11-point) -> {"task1": {class: ap, "map": ...}}
The plotting / zip / visualisation helpers of the reference (:166-178,187-305) are outside the path.
"""
import os

from .result_merge import mergebypoly
from .inference import DafneEvaluator
from .task1 import score_task1, task1_scores, write_task1_files  # noqa: F401  (task1_scores: the name callers import from here)

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


def _generate_task_1_files(metadata, predictions, output_folder, task1_dir, classnames, cfg):
    """predictions: per-image dicts {"file_name", "height", "width", "corners" [K,8], "labels" [K],
    "scores" [K], "centerness" [K]} (evaluation.gather.to_predictions + file names)."""
    skip = (15,) if bool(cfg.DATASETS.DOTA_REMOVE_CONTAINER_CRANE) else ()        # 'container-crane'
    write_task1_files(predictions, output_folder, task1_dir, classnames, cfg, require_square=True, skip_labels=skip)


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
    annopath = os.path.join(metadata.root_dir, "labelTxt", "{:s}.txt")
    score_task1(classnames, task1_dir, annopath, output_folder, parse_gt, cfg, results)


class DotaEvaluator(DafneEvaluator):
    def _eval_predictions(self, predictions):
        do_dota_evaluation(dataset_name=self._dataset_name, metadata=self._metadata, predictions=predictions,
                           output_folder=self._output_dir, logger=self._logger, results=self._results, cfg=self._cfg)
