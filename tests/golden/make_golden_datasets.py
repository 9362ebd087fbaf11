#!/usr/bin/env python3
"""Generate tests/golden/eval_datasets.npz: the HRSC2016 and UCAS-AOD halves of the reference's evaluation, RUN HERE.

The reference's dafne/evaluation/hrsc_evaluation.py and ucas_aod_evaluation.py import half of detectron2, cv2, bs4, seaborn and
matplotlib at module level; the functions on the path need none of that.  So the function DEFINITIONS are taken out of the files
where they lie (ast: nothing is copied into this repo) and executed with the few globals they read:
  hrsc_evaluation.py      xywha2xy4, parse_gt (labelXml/*.xml), _generate_task_1_files
  data/datasets/ucas_aod.py   load_annotation, parse_annotation   (cv2.imread -> a stand-in that returns the image's shape)
  ucas_aod_evaluation.py  parse_gt (Annotations/*.txt through parse_annotation), _generate_task_1_files
and voc_eval (dafne/evaluation/voc_eval.py, loaded as in make_golden_eval.py) scores the written Task1 files with those parsers.
Inputs are synthetic annotation texts + predictions; the fixture stores the texts, the parsed objects, the Task1 lines and rec / prec / ap.

    python tests/golden/make_golden_datasets.py          (build container only)
"""
import ast
import logging
import os
import sys
import tempfile
import types
import xml.etree.ElementTree as ET

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_eval as mge  # noqa: E402


def ref_functions(relpath, names, ns):
    """exec the named top-level function definitions of a reference file in namespace `ns`."""
    path = os.path.join(mg.REF, relpath)
    tree = ast.parse(open(path).read())
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), (relpath, [n.name for n in picked])
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


def hrsc_xml(rng, n):
    rows = []
    for _ in range(n):
        cx, cy = rng.uniform(60, 900, 2)
        w, h = rng.uniform(40, 300), rng.uniform(10, 60)
        a = rng.uniform(-np.pi / 2, np.pi / 2)
        rows.append((cx, cy, w, h, a, int(rng.uniform() < 0.2)))
    body = "".join("<HRSC_Object><Object_ID>%d</Object_ID><Class_ID>100000001</Class_ID><difficult>%d</difficult>"
                   "<mbox_cx>%.4f</mbox_cx><mbox_cy>%.4f</mbox_cy><mbox_w>%.4f</mbox_w><mbox_h>%.4f</mbox_h><mbox_ang>%.6f</mbox_ang>"
                   "</HRSC_Object>" % (k, d, cx, cy, w, h, a) for k, (cx, cy, w, h, a, d) in enumerate(rows))
    return ("<HRSC_Image><Img_ID>1</Img_ID><Img_SizeWidth>%d</Img_SizeWidth><Img_SizeHeight>%d</Img_SizeHeight>"
            "<HRSC_Objects>%s</HRSC_Objects></HRSC_Image>" % (int(rng.integers(900, 1300)), int(rng.integers(600, 900)), body))


def ucas_txt(rng, n):
    """UCAS-AOD annotation lines: class name, 8 corner coordinates (floats in the files), then theta x y w h.  Some boxes
    are degenerate (thinner than 3 px / aspect ratio >= 30): the reference's parser drops them."""
    lines = []
    for k in range(n):
        q = mg.rrects(1, rng, extent=1000.0, lo=12.0, hi=120.0)[0].astype(np.float64)
        if k % 7 == 3:
            q = np.array([100, 100, 101.5, 100, 101.5, 160, 100, 160], np.float64) + k          # 1.5 px wide
        if k % 11 == 5:
            q = np.array([50, 50, 400, 50, 400, 58, 50, 58], np.float64) + k                     # aspect ratio > 30
        name = "car" if rng.uniform() < 0.6 else "airplane"
        lines.append(name + " " + " ".join("%.4f" % v for v in q) + " 0.0 1 2 3 4")
    return "\n".join(lines) + "\n"


def predictions_for(gts, rng, images, ext):
    """Per image: jittered ground truth + duplicates + false positives; unique scores at 4 decimals overall."""
    preds = []
    total = 0
    for img in images:
        g = gts[img]
        rows, labels = [], []
        for lab, q in g:
            if rng.uniform() < 0.85:
                rows.append(q + rng.normal(0, 1.5, 8)); labels.append(lab)
            if rng.uniform() < 0.25:
                rows.append(q + rng.normal(0, 5.0, 8)); labels.append(lab)
        nfp = max(2, len(g) // 3)
        for q in mg.rrects(nfp, rng, extent=ext, lo=12.0, hi=120.0).astype(np.float64):
            rows.append(q); labels.append(int(rng.integers(0, max(l for l, _ in g) + 1)) if g else 0)
        total += len(rows)
        preds.append((img, np.array(rows, np.float32).reshape(-1, 8), np.array(labels, np.int64)))
    sc = mge.unique_scores(total, rng)
    out, o = [], 0
    for img, rows, labels in preds:
        k = len(rows)
        ctr = rng.uniform(0.3, 1.0, k).astype(np.float32)
        out.append({"file_name": "/data/x/%s.png" % img, "image_id": img, "height": 800, "width": 1216,
                    "corners": torch.from_numpy(rows), "labels": torch.from_numpy(labels),
                    "scores": torch.from_numpy(sc[o:o + k].astype(np.float32)), "centerness": torch.from_numpy(ctr)})
        o += k
    return out


def main():
    assert os.path.isdir(mg.REF), "reference tree not present: fixtures can only be made in the build container"
    mg.install_stubs()
    mg._mod("polyiou", VectorDouble=mge.VectorDouble, iou_poly=mge.iou_poly)
    mg._mod("shapely")
    mg._mod("shapely.geometry")
    m = mg._mod("dafne.evaluation")
    m.__path__ = [os.path.join(mg.REF, "dafne", "evaluation")]
    if not hasattr(np, "bool"):
        np.bool = bool
    ve = mg.load_ref("dafne.evaluation.voc_eval")
    log = logging.getLogger("golden")
    cfg = mg.AttrDict({"MODEL": {"DAFNE": {"CENTERNESS": "oriented", "CENTERNESS_USE_IN_SCORE": True}}})
    cfg_noctr = mg.AttrDict({"MODEL": {"DAFNE": {"CENTERNESS": "oriented", "CENTERNESS_USE_IN_SCORE": False}}})
    rng = np.random.default_rng(20260930)
    fx = {}

    with tempfile.TemporaryDirectory() as tmp:
        # ------------------------------------------------------------------ HRSC2016
        H = ref_functions("dafne/evaluation/hrsc_evaluation.py", ["xywha2xy4", "parse_gt", "_generate_task_1_files"],
                          {"np": np, "ET": ET, "os": os, "logger": log, "classnames": ["ship"], "torch": torch})
        root = os.path.join(tmp, "hrsc")
        os.makedirs(os.path.join(root, "labelXml"))
        images = ["100000%03d" % k for k in (1, 2, 7)]
        gts = {}
        for img, n in zip(images, (12, 5, 20)):
            txt = hrsc_xml(rng, n)
            with open(os.path.join(root, "labelXml", img + ".xml"), "w") as f:
                f.write(txt)
            objs = H["parse_gt"](os.path.join(root, "labelXml", img + ".xml"))
            fx["hrsc_xml_" + img] = np.array(txt)
            fx["hrsc_gt_bbox_" + img] = np.array([o["bbox"] for o in objs], np.float64)
            fx["hrsc_gt_difficult_" + img] = np.array([o["difficult"] for o in objs], np.int64)
            assert all(o["name"] == "ship" for o in objs)
            gts[img] = [(0, np.array(o["bbox"])) for o in objs]
        fx["hrsc_images"] = np.array(images)
        # the records of load_hrsc (hrsc2016.py:54-128) for an ImageSets/test.txt naming the three images
        os.makedirs(os.path.join(root, "ImageSets"))
        with open(os.path.join(root, "ImageSets", "test.txt"), "w") as f:
            f.write("\n".join(images) + "\n")
        HD = ref_functions("dafne/data/datasets/hrsc2016.py", ["load_hrsc", "xywha2xy4"],
                           {"np": np, "ET": ET, "os": os, "BoxMode": types.SimpleNamespace(XYWH_ABS=1), "name2label": {"ship": 0}})
        nodebug = mg.AttrDict({"DEBUG": {"OVERFIT_NUM_IMAGES": -1}})
        recs = HD["load_hrsc"](root, "test", nodebug)
        fx["hrsc_rec_file"] = np.array([os.path.relpath(r["file_name"], root) for r in recs])
        fx["hrsc_rec_id"] = np.array([r["image_id"] for r in recs], np.int64)
        fx["hrsc_rec_wh"] = np.array([[r["width"], r["height"]] for r in recs], np.int64)
        fx["hrsc_rec_nobj"] = np.array([len(r["annotations"]) for r in recs], np.int64)
        fx["hrsc_xywha"] = rng.uniform(-3, 300, (6, 5))
        fx["hrsc_xywha_out"] = np.array([H["xywha2xy4"](r) for r in fx["hrsc_xywha"]])
        preds = predictions_for(gts, rng, images, 1000.0)
        for tag, c in (("", cfg), ("_noctr", cfg_noctr)):
            out = os.path.join(tmp, "hrsc_out" + tag)
            os.makedirs(os.path.join(out, "Task1"))
            H["_generate_task_1_files"](None, preds, out, os.path.join(out, "Task1"), c)
            fx["hrsc_task1_ship" + tag] = np.array(open(os.path.join(out, "Task1", "Task1_ship.txt")).read().splitlines())
            fx["hrsc_imageset" + tag] = np.array(sorted(open(os.path.join(out, "imageset.txt")).read().split("\n")))
        for k, p in enumerate(preds):
            for key in ("corners", "labels", "scores", "centerness"):
                fx["hrsc_pred%d_%s" % (k, key)] = p[key].numpy()
        out = os.path.join(tmp, "hrsc_out")
        for th in (0.5, 0.75):
            rec, prec, ap, _ = ve.voc_eval(os.path.join(out, "Task1", "Task1_{:s}.txt"), os.path.join(root, "labelXml", "{:s}.xml"),
                                           os.path.join(out, "imageset.txt"), "ship", ovthresh=th, use_07_metric=True, parse_gt=H["parse_gt"])
            fx["hrsc_rec_%d" % int(th * 100)], fx["hrsc_prec_%d" % int(th * 100)], fx["hrsc_ap_%d" % int(th * 100)] = rec, prec, np.float64(ap)
            print("hrsc thr %.2f: %d dets, ap %.6f" % (th, len(fx["hrsc_task1_ship"]), ap))

        # ------------------------------------------------------------------ UCAS-AOD
        sizes = {}
        cv2 = types.SimpleNamespace(imread=lambda p: np.zeros(sizes[os.path.basename(p)[:-4]] + (3,), np.uint8))
        boxmode = types.SimpleNamespace(XYWH_ABS=1, XYXY_ABS=0)
        names = ["__background__", "car", "airplane"]
        D = ref_functions("dafne/data/datasets/ucas_aod.py", ["load_annotation", "parse_annotation"],
                          {"np": np, "os": os, "cv2": cv2, "BoxMode": boxmode,
                           "name2label": {n: k for k, n in enumerate(names)}, "label2name": dict(enumerate(names))})
        classnames = ["car", "airplane"]
        U = ref_functions("dafne/evaluation/ucas_aod_evaluation.py", ["parse_gt", "_generate_task_1_files"],
                          {"np": np, "os": os, "logger": log, "classnames": classnames, "torch": torch,
                           "parse_annotation": D["parse_annotation"], "label2name": dict(enumerate(classnames))})
        root = os.path.join(tmp, "UCAS-AOD")
        os.makedirs(os.path.join(root, "Annotations"))
        images = ["P0001", "P0002", "P0611"]
        gts = {}
        for img, n in zip(images, (24, 9, 40)):
            sizes[img] = (659, 1280)
            txt = ucas_txt(rng, n)
            with open(os.path.join(root, "Annotations", img + ".txt"), "w") as f:
                f.write(txt)
            objs = U["parse_gt"](os.path.join(root, "Annotations", img + ".txt"))
            fx["ucas_txt_" + img] = np.array(txt)
            fx["ucas_gt_bbox_" + img] = np.array([o["bbox"] for o in objs], np.float64).reshape(-1, 8)
            fx["ucas_gt_name_" + img] = np.array([o["name"] for o in objs])
            assert all(o["difficult"] == 0 for o in objs)
            print("ucas", img, n, "lines ->", len(objs), "objects")
            gts[img] = [(classnames.index(o["name"]), np.array(o["bbox"], np.float64)) for o in objs]
        fx["ucas_images"] = np.array(images)
        os.makedirs(os.path.join(root, "ImageSets"))
        with open(os.path.join(root, "ImageSets", "test.txt"), "w") as f:
            f.write("\n".join(images) + "\n")
        UD = ref_functions("dafne/data/datasets/ucas_aod.py", ["load_ucas_aod"], dict(D))
        recs = UD["load_ucas_aod"](root, "test", mg.AttrDict({"DEBUG": {"OVERFIT_NUM_IMAGES": 2}}))          # the first two only
        fx["ucas_rec_file"] = np.array([os.path.relpath(r["file_name"], root) for r in recs])
        fx["ucas_rec_id"] = np.array([r["image_id"] for r in recs])
        preds = predictions_for(gts, rng, images, 1000.0)
        out = os.path.join(tmp, "ucas_out")
        os.makedirs(os.path.join(out, "Task1"))
        U["_generate_task_1_files"](None, preds, out, os.path.join(out, "Task1"), cfg)
        for c in classnames:
            fx["ucas_task1_" + c] = np.array(open(os.path.join(out, "Task1", "Task1_%s.txt" % c)).read().splitlines())
        for k, p in enumerate(preds):
            for key in ("corners", "labels", "scores", "centerness"):
                fx["ucas_pred%d_%s" % (k, key)] = p[key].numpy()
        for c in classnames:
            rec, prec, ap, _ = ve.voc_eval(os.path.join(out, "Task1", "Task1_{:s}.txt"), os.path.join(root, "Annotations", "{:s}.txt"),
                                           os.path.join(out, "imageset.txt"), c, ovthresh=0.5, use_07_metric=True, parse_gt=U["parse_gt"])
            fx["ucas_rec_" + c], fx["ucas_prec_" + c], fx["ucas_ap_" + c] = rec, prec, np.float64(ap)
            print("ucas %-8s: %d dets, ap %.6f" % (c, len(fx["ucas_task1_" + c]), ap))
    np.savez_compressed(os.path.join(HERE, "eval_datasets.npz"), **fx)
    print("wrote", os.path.join(HERE, "eval_datasets.npz"))


if __name__ == "__main__":
    main()
