#!/usr/bin/env python3
"""Counterpart of the reference's `tools/plain_train_net.py --eval-only` for the MI355X engine
(plain_train_net.py:521-544 setup, :573-589 build/load/test): config YAML + optional weights ->
run the detector over images and dump per-image predictions (the evaluator's dict layout,
dafne/evaluation/dafne_evaluator.py:44-58).  Without a dataset (none ships here) it runs on
synthetic uint8 BGR images with seeded random weights -- config #1 of BASELINE.json is
`--config-file configs/hrsc_r50.yaml --height 800 --width 1216 --num-images 1`.

    python tools/eval_net.py --config-file configs/hrsc_r50.yaml [--weights model.pth] [KEY VALUE ...]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", required=True)
    ap.add_argument("--weights", default="")
    ap.add_argument("--num-images", type=int, default=1)
    ap.add_argument("--height", type=int, default=0, help="synthetic image height (default INPUT.MIN_SIZE_TEST)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tta", action="store_true", help="TEST.AUG multi-scale / flip inference")
    ap.add_argument("--output", default="")
    ap.add_argument("--task1-dir", default="", help="DOTA configs: write Task1_<class>.txt files here and merge the tiles "
                                                    "(Task1_merged/) with the device NMS (dota_evaluation.py:110-184)")
    ap.add_argument("opts", nargs=argparse.REMAINDER, help="KEY VALUE config overrides")
    args = ap.parse_args()

    import dafne_amd.modeling  # noqa: F401  (registers the classes)
    from dafne_amd.checkpoint import load_weights
    from dafne_amd.config import load_cfg
    from dafne_amd.evaluation.gather import to_predictions
    from dafne_amd.modeling.tta import OneStageRCNNWithTTA
    from dafne_amd.registry import build_model

    cfg = load_cfg(args.config_file, args.opts)
    if not torch.cuda.is_available():
        raise SystemExit("eval_net.py needs an MI355X: the HIP path has no CPU fallback")
    dev = torch.device("cuda", 0)
    model = build_model(cfg)
    if args.weights or cfg.MODEL.WEIGHTS:
        missing, unexpected = load_weights(model, args.weights or cfg.MODEL.WEIGHTS)
        print("loaded weights: %d missing, %d unexpected keys" % (len(missing), len(unexpected)))
    else:
        import bench
        model.load_state_dict(bench.seeded_state_dict(model, args.seed))
    model.to(dev)
    model.invalidate()
    h = args.height or cfg.INPUT.MIN_SIZE_TEST
    w = args.width or cfg.INPUT.MIN_SIZE_TEST
    g = torch.Generator().manual_seed(args.seed)
    inputs = [{"image": torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8), "height": h, "width": w,
               "image_id": i, "file_name": "P%04d__1__0___%d.png" % (i // 4, 824 * (i % 4))} for i in range(args.num_images)]
    runner = OneStageRCNNWithTTA(cfg, model) if args.tta else model
    outputs = []
    for inp in inputs:                       # batch size 1 like detectron2's inference_on_dataset
        outputs += runner([inp])
    preds = []
    for inp, out in zip(inputs, outputs):
        inst = out["instances"].to("cpu")
        preds.append({"image_id": inp["image_id"], "file_name": inp["file_name"], "height": inp["height"],
                      "width": inp["width"], "labels": inst.pred_classes, "scores": inst.scores,
                      "corners": inst.pred_corners, "centerness": inst.centerness})
        print("image %d: %d detections, best score %.4f" % (inp["image_id"], len(inst),
                                                             float(inst.scores.max()) if len(inst) else 0.0))
    if args.output:
        torch.save(preds, args.output)
    if args.task1_dir:
        from dafne_amd.evaluation import dota_evaluation as de
        names = list(de.CLASSNAMES_DOTA_1_0) + (["container-crane"] if cfg.MODEL.DAFNE.NUM_CLASSES == 16 else [])
        t1 = os.path.join(args.task1_dir, "Task1")
        merged = os.path.join(args.task1_dir, "Task1_merged")
        os.makedirs(t1, exist_ok=True)
        os.makedirs(merged, exist_ok=True)
        de._generate_task_1_files(None, preds, args.task1_dir, t1, names[:cfg.MODEL.DAFNE.NUM_CLASSES], cfg)
        de.run_merge(t1, merged)
        n_in = sum(len(open(os.path.join(t1, f)).readlines()) for f in os.listdir(t1))
        n_out = sum(len(open(os.path.join(merged, f)).readlines()) for f in os.listdir(merged))
        print("Task1: %d tile detections -> %d after the tile merge (%s)" % (n_in, n_out, merged))
    return preds


if __name__ == "__main__":
    main()
