#!/usr/bin/env python3
"""Generate tests/golden/eval_merge.npz by RUNNING THE REFERENCE's ResultMerge and voc_eval here.

Only runs where /root/reference exists.  The reference's
dafne/utils/ResultMerge_multi_process.py (mergesingle + py_cpu_nms_poly_fast) and
dafne/evaluation/voc_eval.py (voc_eval) are imported from where they lie with stand-ins for what
the image lacks:
  polyiou            SWIG module -> VectorDouble / iou_poly over the reference's own polyiou.cpp,
                     compiled into oracle/_ref
  shapely, detectron2 (pulled in by dota_utils / the package __init__) -> empty modules
  numpy.bool         removed in numpy 2 -> alias of bool for the duration of the run
Inputs are synthetic Task1 / labelTxt texts written to a temp dir; the fixture stores those texts
and the reference's outputs (merged file lines; rec / prec / ap per class).  Scores are unique at
the 4 printed decimals, so numpy's unstable argsort cannot reorder anything.

Usage:  python tests/golden/make_golden_eval.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs, load_ref, rrects)

import oracle  # noqa: E402


class VectorDouble(list):
    pass


def iou_poly(p, q):
    return float(oracle.ref_iou_poly_pairs(np.asarray(p, float).reshape(1, 8), np.asarray(q, float).reshape(1, 8))[0])


def unique_scores(n, rng, lo=500, hi=9999):
    return rng.choice(np.arange(lo, hi), size=n, replace=False) / 10000.0


def synth_tiles(rng):
    """Three original images cut into 1024 tiles (stride 824) at rate 1 and 0.5; objects that fall
    into several tiles are reported by each of them with sub-pixel jitter."""
    lines = {"plane": [], "ship": [], "small-vehicle": []}
    gts = {}
    for img, size in (("P0001", 2400), ("P0002", 1800), ("P0706", 1024)):
        for cls, n in (("plane", 60), ("ship", 150), ("small-vehicle", 400)):
            ext = float(size)
            objs = mg.rrects(n, rng, extent=ext, lo=10.0 if cls != "plane" else 40.0,
                             hi=60.0 if cls == "small-vehicle" else 200.0).astype(np.float64)
            if cls == "small-vehicle":      # parking-lot cluster
                objs[: n // 2] = mg.rrects(n // 2, rng, extent=220.0, lo=10, hi=40).astype(np.float64) + 300.0
            for rate in (1.0, 0.5):
                scaled = size * rate
                starts = list(range(0, max(int(scaled) - 1024, 0) + 1, 824))
                if starts[-1] + 1024 < scaled:
                    starts.append(int(scaled) - 1024)
                for x in starts:
                    for y in starts:
                        t = objs * rate - np.array([x, y] * 4)
                        cx, cy = t[:, 0::2].mean(1), t[:, 1::2].mean(1)
                        inside = (cx > 0) & (cx < 1024) & (cy > 0) & (cy < 1024)
                        sel = np.nonzero(inside & (rng.uniform(size=n) < 0.8))[0]
                        for k in sel:
                            q = t[k] + rng.normal(0, 0.7, 8)
                            rate_s = "1" if rate == 1.0 else "0.5"
                            lines[cls].append(("%s__%s__%d___%d" % (img, rate_s, x, y), " ".join("%.2f" % v for v in q)))
        gts[img] = None
    for cls in lines:       # scores unique per FILE at the printed precision (see module docstring)
        sc = unique_scores(len(lines[cls]), rng)
        lines[cls] = ["%s %.4f %s" % (nm, s, q) for (nm, q), s in zip(lines[cls], sc)]
    # degenerate rows: zero-area quads (identical points / collinear), exact duplicates of a tile box
    lines["plane"].append("P0001__1__0___0 0.0411 100.00 100.00 100.00 100.00 100.00 100.00 100.00 100.00")
    lines["plane"].append("P0001__1__0___0 0.0412 100.00 100.00 100.00 100.00 100.00 100.00 100.00 100.00")
    lines["plane"].append("P0001__1__0___0 0.0413 10.00 10.00 20.00 10.00 30.00 10.00 40.00 10.00")
    dup = lines["ship"][0].split(" ")
    lines["ship"].append(" ".join([dup[0], "0.0414"] + dup[2:]))
    return lines


def synth_val(rng):
    """Validation-style set on tiles: ground truth per tile + detections (jittered GT, duplicates,
    false positives, some GT marked difficult)."""
    classes = ["plane", "ship"]
    images = ["P0003__1__0___0", "P0003__1__824___0", "P0007__1__0___824"]
    gt_txt, det_lines = {}, {c: [] for c in classes}
    for img in images:
        rows = ["imagesource:GoogleEarth", "gsd:0.146"]
        for c in classes:
            n = 25 if c == "plane" else 60
            g = mg.rrects(n, rng, extent=1024.0, lo=15.0, hi=120.0).astype(np.float64)
            diff = (rng.uniform(size=n) < 0.15).astype(int)
            for k in range(n):
                rows.append(" ".join("%.1f" % v for v in g[k]) + " %s %d" % (c, diff[k]))
            dets = []
            for k in range(n):
                if rng.uniform() < 0.85:
                    dets.append(g[k] + rng.normal(0, 1.5, 8))
                if rng.uniform() < 0.25:
                    dets.append(g[k] + rng.normal(0, 4.0, 8))        # duplicate / poorly localised
            dets += list(mg.rrects(n // 2, rng, extent=1024.0, lo=15.0, hi=120.0).astype(np.float64))
            for q in dets:
                det_lines[c].append((img, q))
        gt_txt[img] = "\n".join(rows) + "\n"
    out = {}
    for c in classes:
        sc = unique_scores(len(det_lines[c]), rng)
        out[c] = ["%s %.4f " % (img, s) + " ".join("%.2f" % v for v in q) for (img, q), s in zip(det_lines[c], sc)]
    return images, gt_txt, out


def main():
    mg.install_stubs()
    mg._mod("polyiou", VectorDouble=VectorDouble, iou_poly=iou_poly)
    mg._mod("shapely")
    mg._mod("shapely.geometry")
    for pkg in ("dafne.evaluation",):
        m = mg._mod(pkg)
        m.__path__ = [os.path.join(mg.REF, *pkg.split("."))]
    if not hasattr(np, "bool"):
        np.bool = bool          # voc_eval.py:98 uses the alias numpy 2 removed
    mg.load_ref("dafne.utils.dota_utils")
    rm = mg.load_ref("dafne.utils.ResultMerge_multi_process")
    mg.load_ref("dafne.utils.sort_corners")
    ve = mg.load_ref("dafne.evaluation.voc_eval")

    rng = np.random.default_rng(20260928)
    fx = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- ResultMerge
        src, dst = os.path.join(tmp, "Task1"), os.path.join(tmp, "Task1_merged")
        os.makedirs(src)
        os.makedirs(dst)
        lines = synth_tiles(rng)
        for c, ls in lines.items():
            with open(os.path.join(src, "Task1_%s.txt" % c), "w") as f:
                f.write("\n".join(ls) + "\n")
            rm.mergesingle(dst, rm.py_cpu_nms_poly_fast, os.path.join(src, "Task1_%s.txt" % c))
            with open(os.path.join(dst, "Task1_%s.txt" % c)) as f:
                merged = [x.rstrip("\n") for x in f.readlines()]
            fx["merge_in_" + c] = np.array(ls)
            fx["merge_out_" + c] = np.array(merged)
            print("merge %-14s %5d tile rows -> %5d merged rows" % (c, len(ls), len(merged)))
        # ---- voc_eval
        images, gt_txt, dets = synth_val(rng)
        lab = os.path.join(tmp, "labelTxt")
        os.makedirs(lab)
        for img, txt in gt_txt.items():
            with open(os.path.join(lab, img + ".txt"), "w") as f:
                f.write(txt)
        with open(os.path.join(tmp, "imageset.txt"), "w") as f:
            f.write("\n".join(images))
        parse_gt = _parse_gt_plain
        fx["val_images"] = np.array(images)
        fx["val_gt"] = np.array([gt_txt[i] for i in images])
        for c, ls in dets.items():
            with open(os.path.join(tmp, "Task1_%s.txt" % c), "w") as f:
                f.write("\n".join(ls) + "\n")
            for th in (0.5, 0.75):
                rec, prec, ap, _ = ve.voc_eval(os.path.join(tmp, "Task1_{:s}.txt"), os.path.join(lab, "{:s}.txt"),
                                               os.path.join(tmp, "imageset.txt"), c, ovthresh=th,
                                               use_07_metric=True, parse_gt=parse_gt)
                tag = "%s_%d" % (c, int(th * 100))
                fx["val_rec_" + tag], fx["val_prec_" + tag], fx["val_ap_" + tag] = rec, prec, np.float64(ap)
                print("voc_eval %-6s thr %.2f: %4d dets, ap %.6f" % (c, th, len(ls), ap))
            rec, prec, ap, _ = ve.voc_eval(os.path.join(tmp, "Task1_{:s}.txt"), os.path.join(lab, "{:s}.txt"),
                                           os.path.join(tmp, "imageset.txt"), c, ovthresh=0.5,
                                           use_07_metric=False, parse_gt=parse_gt)
            fx["val_ap_area_" + c] = np.float64(ap)
            fx["val_det_" + c] = np.array(ls)
    np.savez_compressed(os.path.join(HERE, "eval_merge.npz"), **fx)
    print("wrote", os.path.join(HERE, "eval_merge.npz"))


def _parse_gt_plain(filename):
    """The generator's own labelTxt reader handed to the reference's voc_eval as its `parse_gt`
    argument (the reference's parser lives in a module that needs detectron2 at import time)."""
    objs = []
    with open(filename) as f:
        for line in f:
            t = line.strip().split(" ")
            if len(t) < 9:
                continue
            objs.append({"name": t[8], "difficult": int(t[9]) if len(t) == 10 else 0, "bbox": [float(v) for v in t[:8]]})
    return objs


if __name__ == "__main__":
    main()
