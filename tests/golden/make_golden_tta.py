#!/usr/bin/env python3
"""tests/golden/tta_merge.npz: RUNS the reference's dafne/modeling/tta.py in the build container.

What runs from /root/reference, unmodified, imported from where it lies:
  DotaDatasetMapperTTA.__call__                  (tta.py:45-135)   view list + per-view transform lists
  OneStageRCNNWithTTA._inference_one_image       (tta.py:218-232)
      ._get_augmented_inputs / ._batch_inference / ._get_augmented_corners / ._merge_detections  (:170-268)
  DAFNeOutputs.select_over_all_levels -> ml_nms -> batched_nms_poly   (dafne_outputs.py:907-925, nms.py:10-92)

What is a stand-in (packages the image lacks; detectron2 v0.5 / fvcore semantics are [recalled]):
  fvcore.transforms.{HFlipTransform, VFlipTransform, NoOpTransform, TransformList}, detectron2's ResizeTransform /
  ResizeShortestEdge / RandomFlip / apply_augmentations -- hand-written below as numpy code that, like fvcore, works
  IN PLACE on the float32 array tta.py hands over (`pred_corners.cpu().numpy()`, :247-249), so the dtype of the inverse
  maps is numpy's: float32 throughout (python scalars are weak).  ResizeTransform.apply_image = PIL bilinear.
  The detector (`model.inference`) is a stand-in that returns canned per-view detections: the fixture pins the view
  order, the transform lists, the inverse coordinate maps and the merged keep set, not the network.
  poly_nms.poly_gpu_nms := greedy loop over the reference's compiled polyiou.cpp (as in make_golden.py).

The fixture holds numbers only.  Usage: python tests/golden/make_golden_tta.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


# ------------------------------------------------------------------ transform stand-ins (fvcore / d2 [recalled])
class Transform:
    def apply_image(self, img):
        return img


class NoOpTransform(Transform):
    def apply_coords(self, coords):
        return coords

    def inverse(self):
        return self


class HFlipTransform(Transform):
    def __init__(self, width):
        self.width = width

    def apply_image(self, img):
        return np.flip(img, axis=1)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords

    def inverse(self):
        return self


class VFlipTransform(Transform):
    def __init__(self, height):
        self.height = height

    def apply_image(self, img):
        return np.flip(img, axis=0)

    def apply_coords(self, coords):
        coords[:, 1] = self.height - coords[:, 1]
        return coords

    def inverse(self):
        return self


class ResizeTransform(Transform):
    def __init__(self, h, w, new_h, new_w, interp=None):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w

    def apply_image(self, img):
        from PIL import Image
        assert img.shape[:2] == (self.h, self.w) and img.dtype == np.uint8
        return np.asarray(Image.fromarray(img).resize((self.new_w, self.new_h), Image.BILINEAR))

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords

    def inverse(self):
        return ResizeTransform(self.new_h, self.new_w, self.h, self.w)


class TransformList(Transform):
    def __init__(self, transforms):
        flat = []
        for t in transforms:
            flat.extend(t.transforms if isinstance(t, TransformList) else [t])
        self.transforms = flat

    def apply_coords(self, coords):
        for t in self.transforms:
            coords = t.apply_coords(coords)
        return coords

    def apply_image(self, img):
        for t in self.transforms:
            img = t.apply_image(img)
        return img

    def __add__(self, other):
        return TransformList(self.transforms + (other.transforms if isinstance(other, TransformList) else [other]))

    def __radd__(self, other):
        return TransformList((other.transforms if isinstance(other, TransformList) else [other]) + self.transforms)

    def inverse(self):
        return TransformList([t.inverse() for t in self.transforms[::-1]])


class ResizeShortestEdge:
    def __init__(self, short_edge_length, max_size=sys.maxsize):
        self.size, self.max_size = short_edge_length, max_size

    def get_transform(self, image):
        h, w = image.shape[:2]
        size = self.size
        scale = size * 1.0 / min(h, w)
        if h < w:
            newh, neww = size, scale * w
        else:
            newh, neww = scale * h, size
        if max(newh, neww) > self.max_size:
            scale = self.max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return ResizeTransform(h, w, int(newh + 0.5), int(neww + 0.5))


class RandomFlip:
    def __init__(self, prob=0.5, *, horizontal=True, vertical=False):
        assert prob == 1.0
        self.horizontal, self.vertical = horizontal, vertical

    def get_transform(self, image):
        h, w = image.shape[:2]
        return HFlipTransform(w) if self.horizontal else VFlipTransform(h)


def apply_augmentations(augs, image):
    tfms = []
    for a in augs:
        t = a.get_transform(image)
        image = t.apply_image(image)
        tfms.append(t)
    return image, TransformList(tfms)


class _Unused:
    def __init__(self, *a, **k):
        raise NotImplementedError


class OneStageDetector(torch.nn.Module):
    """Stand-in detector: returns canned detections per view, in call order."""

    def __init__(self, outputs, canned):
        super().__init__()
        self.proposal_generator = type("PG", (), {})()
        self.proposal_generator.dafne_outputs = outputs
        self.canned, self.cursor, self.seen = canned, 0, []

    def inference(self, inputs, detected=None, do_postprocess=True):
        assert detected is None and do_postprocess is False
        out = []
        for x in inputs:
            self.seen.append(tuple(x["image"].shape))
            # fresh tensors per call, as a real forward pass yields: tta.py maps `pred_corners.cpu().numpy()` IN PLACE
            # (:247-249), which on these CPU stand-ins would otherwise write through to the canned arrays
            src = self.canned[self.cursor]
            inst = mg.Instances(src.image_size)
            for k, v in src.get_fields().items():
                inst._fields[k] = v.clone()
            out.append({"instances": inst})
            self.cursor += 1
        return out


class Cfg(mg.AttrDict):
    def clone(self):
        return self


def canned_views(rng, view_hw, classes, per_view, dense):
    """Detections in each view's own frame: the same underlying objects seen through every view (so that the merged
    NMS has real work), plus per-view jitter and view-only extras."""
    n_obj = per_view
    base = mg.rrects(n_obj, rng, extent=1.0, lo=0.02, hi=0.22, jitter=0.0).astype(np.float64)   # unit-square frame
    if dense:
        base[: n_obj * 7 // 10] = base[: n_obj * 7 // 10] * 0.25 + 0.3
    cls = rng.integers(0, classes, n_obj)
    out = []
    for (h, w, hf, vf) in view_hw:
        c = base.copy() + rng.normal(0, 0.004, base.shape)
        c[:, 0::2] *= w
        c[:, 1::2] *= h
        if hf:
            c[:, 0::2] = w - c[:, 0::2]
        if vf:
            c[:, 1::2] = h - c[:, 1::2]
        sel = rng.random(n_obj) < 0.8
        inst = mg.Instances((h, w))
        inst.pred_corners = torch.from_numpy(c[sel].astype(np.float32))
        inst.scores = torch.from_numpy(np.round(rng.uniform(0.05, 1, int(sel.sum())), 3).astype(np.float32))   # ties
        inst.centerness = torch.from_numpy(rng.uniform(0.1, 1, int(sel.sum())).astype(np.float32))
        inst.pred_classes = torch.from_numpy(cls[sel].astype(np.int64))
        out.append(inst)
    return out


def main():
    assert os.path.isdir(mg.REF)
    mg.install_stubs()
    mg._mod("fvcore.transforms", HFlipTransform=HFlipTransform, NoOpTransform=NoOpTransform)
    mg._mod("detectron2.data")
    mg._mod("detectron2.data.detection_utils", read_image=None)
    mg._mod("detectron2.data.transforms", RandomFlip=RandomFlip, ResizeShortestEdge=ResizeShortestEdge, Resize=_Unused,
            ResizeTransform=ResizeTransform, apply_augmentations=apply_augmentations)
    mg._mod("detectron2.data.transforms.augmentation_impl", RandomRotation=_Unused)
    mg._mod("dafne.modeling.one_stage_detector", OneStageDetector=OneStageDetector)
    mg.load_ref("dafne.utils.sort_corners")
    mg.load_ref("dafne.layers.deform_conv")
    mg.load_ref("dafne.modeling.losses.utils")
    mg.load_ref("dafne.modeling.losses.smooth_l1")
    mg.load_ref("dafne.modeling.nms.nms")
    outputs_mod = mg.load_ref("dafne.modeling.dafne.dafne_outputs")
    tta_mod = mg.load_ref("dafne.modeling.tta")

    res = {}
    cases = [
        # name, config dump, image (h, w), dataset (height, width), MIN_SIZES, MAX_SIZE, detections per view, dense
        ("d15", "dota-1.5_r101_ms.yaml", (128, 160), (128, 160), [96, 128, 160, 224], 256, 260, False),
        ("d10_pre", "dota-1.0_r101_ms.yaml", (120, 90), (240, 180), [64, 100, 150], 160, 200, True),   # pre_tfm != NoOp
        ("d15_cap", "dota-1.5_r101_ms.yaml", (96, 96), (96, 96), [80, 96, 128], 200, 700, False),      # > POST_NMS_TOPK
    ]
    rng = np.random.default_rng(606)
    for name, cfgfile, (h, w), (oh, ow), sizes, max_size, per_view, dense in cases:
        cfg = Cfg(mg.load_cfg(cfgfile))
        cfg["TEST"]["AUG"]["MIN_SIZES"] = sizes
        cfg["TEST"]["AUG"]["MAX_SIZE"] = max_size
        if name == "d15_cap":
            cfg["MODEL"]["DAFNE"]["POST_NMS_TOPK_TEST"] = 150
        outs = outputs_mod.DAFNeOutputs(cfg)
        outs.eval()
        # predict_proposals sets these as a side effect of every real inference call (dafne_outputs.py:747-749); the
        # stand-in detector never reaches it
        outs.pre_nms_thresh, outs.pre_nms_topk = outs.pre_nms_thresh_test, outs.pre_nms_topk_test
        outs.post_nms_topk = outs.post_nms_topk_test
        img = torch.from_numpy(rng.integers(0, 256, (3, h, w), dtype=np.uint8))
        mapper = tta_mod.DotaDatasetMapperTTA(cfg)
        views = mapper({"image": img, "height": oh, "width": ow})
        desc = []
        for v in views:
            t = v["transforms"].transforms
            desc.append((v["image"].shape[1], v["image"].shape[2], int(any(isinstance(x, HFlipTransform) for x in t)),
                         int(any(isinstance(x, VFlipTransform) for x in t))))
        canned = canned_views(rng, desc, cfg.MODEL.DAFNE.NUM_CLASSES, per_view, dense)
        model = OneStageDetector(outs, [c for c in canned])
        tta = tta_mod.OneStageRCNNWithTTA(cfg, model, tta_mapper=mapper)
        with torch.no_grad():
            aug, tfms = tta._get_augmented_inputs({"image": img, "height": oh, "width": ow})
            inst = tta._get_augmented_corners(aug, tfms)
            model.cursor = 0
            merged = tta([{"image": img, "height": oh, "width": ow}])[0]["instances"]
        assert model.seen[: len(views)] == [tuple(v["image"].shape) for v in views]
        res[name + "_image"] = img.numpy()
        res[name + "_orig_hw"] = np.array([oh, ow], np.int64)
        res[name + "_min_sizes"] = np.array(sizes, np.int64)
        res[name + "_max_size"] = np.int64(max_size)
        res[name + "_cfg"] = np.array([cfg.MODEL.DAFNE.NUM_CLASSES, cfg.MODEL.DAFNE.POST_NMS_TOPK_TEST], np.int64)
        res[name + "_nms_th"] = np.float64(cfg.MODEL.DAFNE.NMS_TH)
        res[name + "_views"] = np.array(desc, np.int64)                      # (h, w, hflip, vflip) in mapper order
        for k, v in enumerate(views):
            if name == "d10_pre":                 # view pixels (PIL resize + flips) for one case: keeps the fixture small
                res["%s_view%d_image" % (name, k)] = v["image"].numpy()
            f = canned[k].get_fields()
            for key in ("pred_corners", "scores", "centerness", "pred_classes"):
                res["%s_view%d_%s" % (name, k, key)] = f[key].numpy()
        res[name + "_inv_corners"] = inst.pred_corners.numpy()               # every view mapped back, concatenated
        mf = merged.get_fields()
        for key in ("pred_corners", "scores", "centerness", "pred_classes"):
            res["%s_merged_%s" % (name, key)] = mf[key].numpy()
        print("tta", name, "views", len(views), "dets in", len(inst), "merged", len(merged))
    np.savez_compressed(os.path.join(HERE, "tta_merge.npz"), **res)


if __name__ == "__main__":
    main()
