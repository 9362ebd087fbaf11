"""HRSC2016 scoring (BASELINE config 0's dataset): Task1 file + VOC07 AP of the one class on the device IoU kernel.

Counterpart of dafne/evaluation/hrsc_evaluation.py:
  xywha2xy4            (:77-82)    (cx, cy, w, h, angle in radians) -> the 4 corners
  parse_gt             (:85-103)   labelXml/<image>.xml: every HRSC_Object is a "ship" with its `difficult` flag and the
                                   rotated box mbox_cx / mbox_cy / mbox_w / mbox_h / mbox_ang
  _generate_task_1_files (:104-152), do_hrsc_evaluation (:278-347) -> results["task1"] = {"ship": ap, "map": ap}
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from .inference import DafneEvaluator
from .task1 import score_task1, write_task1_files

classnames = ["ship"]


def xywha2xy4(xywha):
    x, y, w, h, a = xywha
    corner = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
    rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    return rot.dot(corner.T).T + [x, y]


def parse_gt(filename):
    objects = []
    root = ET.parse(filename).getroot()
    for obj in root.findall("HRSC_Objects")[0].findall("HRSC_Object"):
        box = [float(obj.find(k).text) for k in ("mbox_cx", "mbox_cy", "mbox_w", "mbox_h", "mbox_ang")]
        objects.append({"name": "ship", "difficult": int(obj.find("difficult").text),
                        "bbox": xywha2xy4(box).reshape(-1).tolist()})
    return objects


def _generate_task_1_files(metadata, predictions, output_folder, task1_dir, cfg):
    write_task1_files(predictions, output_folder, task1_dir, classnames, cfg)


def do_hrsc_evaluation(dataset_name, metadata, predictions, output_folder, logger, results, cfg):
    task1_dir = os.path.join(output_folder, "Task1")
    os.makedirs(task1_dir, exist_ok=True)
    _generate_task_1_files(metadata, predictions, output_folder, task1_dir, cfg)
    annopath = os.path.join(metadata.root_dir, "labelXml", "{:s}.xml")
    score_task1(classnames, task1_dir, annopath, output_folder, parse_gt, cfg, results)


class HrscEvaluator(DafneEvaluator):
    def _eval_predictions(self, predictions):
        do_hrsc_evaluation(dataset_name=self._dataset_name, metadata=self._metadata, predictions=predictions,
                           output_folder=self._output_dir, logger=self._logger, results=self._results, cfg=self._cfg)
