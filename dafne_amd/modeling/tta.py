"""Test-time augmentation for the MI355X engine: same class names and call
contract as the reference's dafne/modeling/tta.py (`DotaDatasetMapperTTA` :29-135,
`OneStageRCNNWithTTA` :138-268).

For every TEST.AUG.MIN_SIZES entry the image is resized (shortest edge, capped by
MAX_SIZE) and additionally flipped horizontally / vertically; the detector runs on
chunks of 3 views with do_postprocess=False; corners are mapped back through the
inverse transforms (un-flip, then un-resize; float32 like fvcore / detectron2's
in-place apply_coords on the float32 numpy copy, :244-259 -- python-scalar operands are weak, so the
scale is rounded to float32 and `width - x` stays float32) and all views are merged by ONE rotated
NMS + cap (`select_over_all_levels`, :264-268) -- up to 27 x 1000 quads, on the GPU.

The pixel resampling (detectron2's ResizeTransform is PIL bilinear on uint8 [recalled]) and the flips run on
the GPU in `dafne_resize_bilinear_u8_hip`, bit-exact to Pillow's 8-bit resampler (pinned
against Pillow itself in tests/test_oracle_resize.py).  Rotation TTA (ROTATION_ANGLES) is empty in every
released config and not built.
"""
import copy
import os
from itertools import count

import torch
from torch import nn

from .. import _lib
from .. import postprocess as pp
from ..structures import Instances
from .one_stage_detector import OneStageDetector

__all__ = ["DotaDatasetMapperTTA", "OneStageRCNNWithTTA"]


class ResizeT:
    def __init__(self, h, w, new_h, new_w):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w

    def apply_image(self, img, hflip=False, vflip=False):            # uint8 CHW on the GPU
        return resize_u8(img, self.new_h, self.new_w, hflip, vflip)

    def apply_coords(self, c):             # [n,2] float32: x * float32(python double ratio), one fp32 multiply
        c = c.clone()
        c[:, 0] = c[:, 0] * torch.tensor(self.new_w * 1.0 / self.w, dtype=c.dtype)
        c[:, 1] = c[:, 1] * torch.tensor(self.new_h * 1.0 / self.h, dtype=c.dtype)
        return c

    def inverse(self):
        return ResizeT(self.new_h, self.new_w, self.h, self.w)


def resize_u8(img, new_h, new_w, hflip=False, vflip=False):
    """uint8 [C,H,W] CUDA tensor -> uint8 [C,new_h,new_w]: Pillow-exact bilinear resize + optional flips in
    one pass pair on the device.  No CPU path."""
    if img.dtype != torch.uint8 or img.dim() != 3:
        raise ValueError("resize_u8 takes a uint8 CHW tensor")
    if not img.is_cuda:
        raise _lib.DafneHipError("resize_u8: the MI355X engine has no CPU path (got a CPU tensor)")
    c, h, w = (int(v) for v in img.shape)
    if (h, w) == (new_h, new_w) and not hflip and not vflip:
        return img
    L = _lib.load()
    with torch.cuda.device(img.device):
        src = img.contiguous()
        out = torch.empty((c, new_h, new_w), dtype=torch.uint8, device=img.device)
        nbytes = L.dafne_resize_workspace_bytes(c, h, new_w)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
        _lib.check(L.dafne_resize_bilinear_u8_hip(_lib.ptr(src), 0, c, h, w, new_h, new_w, int(hflip), int(vflip),
                                                  _lib.ptr(out), _lib.ptr(ws), nbytes, _lib.current_stream()),
                   "dafne_resize_bilinear_u8_hip")
    return out


class HFlipT:
    def __init__(self, width):
        self.width = width

    def apply_image(self, img):
        return torch.flip(img, dims=[2])

    def apply_coords(self, c):
        c = c.clone()
        c[:, 0] = self.width - c[:, 0]
        return c

    def inverse(self):
        return self


class VFlipT:
    def __init__(self, height):
        self.height = height

    def apply_image(self, img):
        return torch.flip(img, dims=[1])

    def apply_coords(self, c):
        c = c.clone()
        c[:, 1] = self.height - c[:, 1]
        return c

    def inverse(self):
        return self


class TransformList:
    def __init__(self, tfms):
        self.tfms = list(tfms)

    def __add__(self, other):
        return TransformList(self.tfms + (other.tfms if isinstance(other, TransformList) else list(other)))

    def apply_coords(self, c):
        for t in self.tfms:
            c = t.apply_coords(c)
        return c

    def inverse(self):
        return TransformList([t.inverse() for t in reversed(self.tfms)])


def shortest_edge_size(h, w, size, max_size):
    """detectron2 ResizeShortestEdge.get_output_shape [recalled]."""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


class DotaDatasetMapperTTA:
    def __init__(self, cfg):
        self.min_sizes = cfg.TEST.AUG.MIN_SIZES
        self.max_size = cfg.TEST.AUG.MAX_SIZE
        self.resize_type = cfg.INPUT.RESIZE_TYPE
        self.vflip = cfg.TEST.AUG.VFLIP
        self.hflip = cfg.TEST.AUG.HFLIP
        self.rotation_angles = cfg.TEST.AUG.ROTATION_ANGLES
        self.image_format = cfg.INPUT.FORMAT
        if len(self.rotation_angles):
            raise NotImplementedError("rotation TTA is not used by any released config")
        if self.resize_type != "shortest-edge":
            raise NotImplementedError("INPUT.RESIZE_TYPE='both' is not used by any released config")

    def view_specs(self, h, w, orig_hw):
        """The views of an (h, w) image in the reference's order (tta.py:71-99: per size: plain, hflip, vflip):
        [(new_h, new_w, TransformList)] -- no pixels (what a rank that only merges needs)."""
        pre = TransformList([ResizeT(orig_hw[0], orig_hw[1], h, w)] if (h, w) != tuple(orig_hw) else [])
        specs = []
        for s in self.min_sizes:
            nh, nw = shortest_edge_size(h, w, s, self.max_size)
            rs = ResizeT(h, w, nh, nw)
            cands = [[rs]]
            if self.hflip:
                cands.append([rs, HFlipT(nw)])
            if self.vflip:
                cands.append([rs, VFlipT(nh)])
            for tf in cands:
                specs.append((nh, nw, pre + TransformList(tf)))
        return specs

    def __call__(self, dataset_dict, views=None):
        """views: optional iterable of view indices to build (a rank's shard of the views); default all."""
        image = dataset_dict["image"]
        h, w = int(image.shape[1]), int(image.shape[2])
        orig = (int(dataset_dict["height"]), int(dataset_dict["width"]))
        specs = self.view_specs(h, w, orig)
        ret = []
        for k in (range(len(specs)) if views is None else views):
            nh, nw, tfl = specs[k]
            tf = tfl.tfms[-2:] if isinstance(tfl.tfms[-1], (HFlipT, VFlipT)) else tfl.tfms[-1:]
            rs = tf[0]
            # resize and flip of a view in one device call (flips are exact index reversals)
            im = rs.apply_image(image, hflip=any(isinstance(t, HFlipT) for t in tf[1:]),
                                vflip=any(isinstance(t, VFlipT) for t in tf[1:]))
            dic = {kk: v for kk, v in dataset_dict.items() if kk != "image"}
            dic = copy.deepcopy(dic)
            dic["transforms"] = tfl
            dic["image"] = im.contiguous()
            ret.append(dic)
        return ret


class OneStageRCNNWithTTA(nn.Module):
    def __init__(self, cfg, model, tta_mapper=None, batch_size=3, images_per_group=3):
        """batch_size: views per detector call when ONE image is augmented (tta.py:173-197 runs them in chunks of 3).
        images_per_group: __call__ with several images of one size runs the same-shape views of up to this many images as ONE
        chunk (3 views x 3 images = 9 per call): the reference augments image by image, which is a scheduling choice -- every
        image's views, transforms and merged NMS are the same; 1 restores the per-image loop."""
        super().__init__()
        assert isinstance(model, OneStageDetector), \
            "TTA is only supported on OneStageDetector. Got a model of type {}".format(type(model))
        self.cfg = cfg.clone() if hasattr(cfg, "clone") else cfg
        self.model = model
        self.tta_mapper = tta_mapper if tta_mapper is not None else DotaDatasetMapperTTA(cfg)
        self.batch_size = batch_size
        self.images_per_group = max(1, int(images_per_group))

    def _batch_inference(self, batched_inputs, detected_instances=None):
        outputs, inputs = [], []
        for idx, inp in zip(count(), batched_inputs):
            inputs.append(inp)
            if len(inputs) == self.batch_size or idx == len(batched_inputs) - 1:
                outputs.extend(self.model.inference(inputs, None, do_postprocess=False))
                inputs = []
        return outputs

    def _batch_inference_packed(self, batched_inputs):
        """Same result as _batch_inference, without a host round trip per chunk: every chunk of `batch_size` views
        goes through the detector's pipelined path (three chunks in flight on the three compute streams, decode + NMS
        of a chunk on the side stream under the other chunks' convolutions; detections stay packed on the device) and
        the host waits once, after the last chunk."""
        outputs = []
        for rows, counts, out_hw in self._views_packed(batched_inputs):
            outputs.extend({"instances": r} for r in pp.rows_to_instances(rows, counts, out_hw))
        return outputs

    def _views_packed(self, batched_inputs, chunk_sizes=None, sync=True):
        """-> [(rows [b, k_cap, 18], counts [b], out_hw)] per chunk of `batch_size` views (or of chunk_sizes[j] views),
        packed on the device (host already synchronised with the side stream).  sync=False: -> (that list, event): the host
        has NOT waited; `event.synchronize()` before the rows are read."""
        m = self.model
        pending = []
        if chunk_sizes is None:
            chunk_sizes = [min(self.batch_size, len(batched_inputs) - i) for i in range(0, len(batched_inputs), self.batch_size)]
        assert sum(chunk_sizes) == len(batched_inputs)
        # chunks rotate over the detector's compute streams: three in flight for chunks of 3 views (one image), two for the
        # grouped chunks of 6-9 (measured per image: 22.8 / 23.1 ms with 3 / 2 streams for one image per group, 21.5 / 20.6 ms
        # for three; one stream: 31.9 / 23.3 ms)
        nstreams = 2 if max(chunk_sizes, default=0) >= 6 else 3
        i = 0
        for cs in chunk_sizes:
            chunk = batched_inputs[i:i + cs]
            i += cs
            imgs = [x["image"] for x in chunk]
            hs = [int(im.shape[1]) for im in imgs]
            ws = [int(im.shape[2]) for im in imgs]
            if any(im.dtype != torch.uint8 or not im.is_cuda for im in imgs):
                raise NotImplementedError("TTA views are uint8 CUDA tensors (DotaDatasetMapperTTA builds them on the GPU)")
            if len(set(zip(hs, ws))) == 1:
                batch = torch.stack(imgs)
            else:
                batch = torch.zeros(len(imgs), 3, max(hs), max(ws), dtype=torch.uint8, device=imgs[0].device)
                for k, im in enumerate(imgs):
                    batch[k, :, : hs[k], : ws[k]] = im
            out_hw = [(int(x.get("height", hs[k])), int(x.get("width", ws[k]))) for k, x in enumerate(chunk)]
            rows, counts = m.detect_packed(batch, valid_hw=list(zip(hs, ws)), out_hw=out_hw, do_postprocess=False, graphs=False,
                                           pipelined=True, splits=1, stream_offset=len(pending) % nstreams)
            pending.append((rows, counts, out_hw))
        if not sync:
            ev = torch.cuda.Event()
            ev.record(m.side_stream if m.side_stream is not None else torch.cuda.current_stream(m.device))
            return pending, ev
        if m.side_stream is not None:
            m.side_stream.synchronize()
        return pending

    # ---- the views of ONE image split over the GPUs of a node, merge NMS on rank 0 -----------------
    def _detect_view_range(self, input, lo, hi):
        """Packed detections (view coordinates, do_postprocess=False) of views [lo, hi) of `input`."""
        views = self.tta_mapper(input, views=range(lo, hi))
        for v in views:
            v.pop("transforms")
        chunks = self._views_packed(views)
        return torch.cat([c[0] for c in chunks]), torch.cat([c[1] for c in chunks])

    def inference_view_sharded(self, input, rank=0, world=1, group=None, device=None):
        """One image, its TTA views sharded contiguously over the ranks (tta.py:173-197 runs them in chunks of 3 on one
        GPU): every rank builds and runs ITS views, one gather_detections lands the packed per-view detections on rank
        0 in view order, rank 0 inverts the transforms, concatenates and runs the merged rotated NMS + cap
        (tta.py:237-268).  Every rank calls this (a collective); returns {"instances": ...} on rank 0, None elsewhere.
        Same result as __call__ on one GPU (same kernels on the same views; only where they run differs)."""
        from ..evaluation.driver import inference_on_images
        input = dict(input)
        if "height" not in input and "width" not in input:
            input["height"], input["width"] = int(input["image"].shape[1]), int(input["image"].shape[2])
        device = device if device is not None else self.model.device
        if not input["image"].is_cuda and torch.device(device).type == "cuda":
            input["image"] = input["image"].to(device)
        h, w = int(input["image"].shape[1]), int(input["image"].shape[2])
        specs = self.tta_mapper.view_specs(h, w, (int(input["height"]), int(input["width"])))
        n = len(specs)
        k_cap = self._view_k_cap()
        out = inference_on_images(lambda lo, hi: self._detect_view_range(input, lo, hi), n, k_cap, batch_size=max(n, 1),
                                  rank=rank, world=world, device=device, group=group)
        if out is None:
            return None
        rows_all, counts_all = out
        # image_size of every view's Instances = the ORIGINAL image's (height, width), as _views_packed builds them: the merged
        # result of the sharded path then carries the same image_size as __call__'s
        outputs = [{"instances": r} for r in pp.rows_to_instances(rows_all, counts_all, [(int(input["height"]), int(input["width"]))] * n)]
        instances = self._invert_and_concat(outputs, [s[2] for s in specs])
        return {"instances": self._merge_detections(instances)}

    def _view_k_cap(self):
        return self.model.proposal_generator.dafne_outputs.packed_k_cap()

    def __call__(self, batched_inputs):
        def _fill(d):
            ret = copy.copy(d)
            if "height" not in ret and "width" not in ret:
                ret["height"], ret["width"] = int(ret["image"].shape[1]), int(ret["image"].shape[2])
            return ret
        filled = [_fill(x) for x in batched_inputs]
        groups, i = [], 0
        while i < len(filled):
            j = i + 1
            key = self._group_key(filled[i])
            while j < len(filled) and j - i < self.images_per_group and self._group_key(filled[j]) == key:
                j += 1
            groups.append(filled[i:j])
            i = j
        if len(groups) == 1 and len(groups[0]) == 1:
            return [self._inference_one_image(groups[0][0])]
        # group g + 1's views are built and enqueued BEFORE the host waits for group g: its convolutions run under group g's
        # row read-back, inverse transforms and merged NMS (4-5 ms per image that the per-image loop leaves the GPU idle for)
        out, pend = [], None
        for grp in groups:
            cur = self._enqueue_images(grp)
            if pend is not None:
                out.extend(self._finish_images(pend))
            pend = cur
        out.extend(self._finish_images(pend))
        return out

    @staticmethod
    def _group_key(x):
        return (int(x["image"].shape[1]), int(x["image"].shape[2]), int(x["height"]), int(x["width"]))

    def _enqueue_images(self, inputs):
        """One or several images of ONE size: view k of every image has the same shape, so the runs of consecutive same-shape
        views (a size's plain / hflip / vflip) of all the images go through the detector as one chunk.  Enqueues everything;
        the host does not wait."""
        per = [self._get_augmented_inputs(x) for x in inputs]            # [(views, tfms)] per image
        nv = len(per[0][0])
        shape = lambda v: (int(v["image"].shape[1]), int(v["image"].shape[2]))
        runs, k = [], 0
        while k < nv:                                                    # runs of consecutive equal-shape views (of image 0 = of all)
            e = k + 1
            while e < nv and e - k < self.batch_size and shape(per[0][0][e]) == shape(per[0][0][k]):
                e += 1
            runs.append((k, e))
            k = e
        flat, where, sizes = [], [], []
        for (a, b) in runs:
            for i, (views, _) in enumerate(per):
                flat.extend(views[a:b])
                where.extend((i, v) for v in range(a, b))
            sizes.append((b - a) * len(per))
        pending, ev = self._views_packed(flat, sizes, sync=False)
        return per, nv, where, pending, ev

    def _finish_images(self, state):
        """Per image the views, their order, the inverse transforms and the merged NMS are those of _inference_one_image."""
        per, nv, where, pending, ev = state
        ev.synchronize()
        by_img = [[None] * nv for _ in per]
        pos = 0
        for rows, counts, out_hw in pending:
            for r in pp.rows_to_instances(rows, counts, out_hw):
                i, v = where[pos]
                by_img[i][v] = {"instances": r}
                pos += 1
        return [{"instances": self._merge_detections(self._invert_and_concat(by_img[i], per[i][1]))} for i in range(len(per))]

    def _inference_one_image(self, input):
        augmented_inputs, tfms = self._get_augmented_inputs(input)
        instances = self._get_augmented_corners(augmented_inputs, tfms)
        return {"instances": self._merge_detections(instances)}

    def _get_augmented_inputs(self, input):
        if not input["image"].is_cuda:          # dataset tensors arrive on the host: one upload, views are built on the GPU
            input = dict(input)
            input["image"] = input["image"].to(self.model.device)
        augmented_inputs = self.tta_mapper(input)
        tfms = [x.pop("transforms") for x in augmented_inputs]
        return augmented_inputs, tfms

    def _get_augmented_corners(self, augmented_inputs, tfms):
        return self._invert_and_concat(self._batch_inference_packed(augmented_inputs), tfms)

    def _invert_and_concat(self, outputs, tfms):
        """tta.py:237-262: every view's corners back through the inverse of its transform list, then one Instances.
        The lists this mapper builds are [pre-resize,] resize [, one flip]: their inverses -- un-flip, x * ratio, [x * ratio] -- are
        applied to ALL views' corners at once with per-row parameters (the same float32 operations in the same order as the
        per-view loop: equal bits, tests/test_gpu_model.py); anything else takes the per-view loop."""
        fast = OneStageRCNNWithTTA._invert_and_concat_fast(outputs, tfms)          # (neither form needs `self`)
        if fast is not None:
            return fast
        return OneStageRCNNWithTTA._invert_and_concat_loop(self, outputs, tfms)

    def _invert_and_concat_loop(self, outputs, tfms):
        lst = []
        for output, tfm in zip(outputs, tfms):
            inst = output["instances"]
            pc = inst.pred_corners
            n = pc.shape[0]
            orig = tfm.inverse().apply_coords(pc.reshape(-1, 2)).reshape(n, 8)      # float32 throughout (:247-259)
            r = Instances(inst.image_size)
            r.scores = inst.scores
            r.centerness = inst.centerness
            r.pred_corners = orig
            r.pred_classes = inst.pred_classes
            lst.append(r)
        return Instances.cat(lst)

    _PIN = {}

    @staticmethod
    def _pinned_table(n):
        ev = OneStageRCNNWithTTA._PIN.pop("ev", None)
        if ev is not None:
            ev.synchronize()
        t = OneStageRCNNWithTTA._PIN.get("t")
        if t is None or t.shape[0] < n:
            t = torch.empty(max(64, n), 9, dtype=torch.float64).pin_memory()
            OneStageRCNNWithTTA._PIN["t"] = t
        return t

    @staticmethod
    def _invert_and_concat_fast(outputs, tfms):
        insts = [o["instances"] for o in outputs]
        if not insts or any(i.pred_corners.dtype != torch.float32 for i in insts):
            return None
        rows = []                      # per view: (flip x?, width, flip y?, height, ratio x 1, ratio y 1, ratio x 2, ratio y 2)
        for tfm in tfms:
            ops = list(tfm.tfms)
            fx = fy = False
            wv = hv = 0.0
            if ops and isinstance(ops[-1], HFlipT):
                fx, wv = True, float(ops.pop().width)
            elif ops and isinstance(ops[-1], VFlipT):
                fy, hv = True, float(ops.pop().height)
            if not ops or len(ops) > 2 or not all(isinstance(t, ResizeT) for t in ops):
                return None
            inv = [t.inverse() for t in reversed(ops)]           # un-resize of the view first, then of the pre-resize
            rx = [t.new_w * 1.0 / t.w for t in inv] + [1.0]
            ry = [t.new_h * 1.0 / t.h for t in inv] + [1.0]
            rows.append((fx, wv, fy, hv, rx[0], ry[0], rx[1], ry[1]))
        dev = insts[0].pred_corners.device
        counts = [len(i) for i in insts]
        total = sum(counts)
        table = torch.tensor([list(map(float, r)) + [float(n)] for r, n in zip(rows, counts)], dtype=torch.float64)     # [views, 9]
        if dev.type == "cuda":
            # ONE small upload from a pinned buffer, asynchronous on the caller's stream (a pageable host-to-device copy makes
            # the host wait for the device: measured 33 instead of 19 ms per image, the next group's convolutions drained first)
            pin = OneStageRCNNWithTTA._pinned_table(table.shape[0])
            pin[: table.shape[0]].copy_(table)
            table = pin[: table.shape[0]].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            OneStageRCNNWithTTA._PIN["ev"] = ev                     # the buffer is rewritten only after this copy has read it
        expand = torch.repeat_interleave(table[:, :8], table[:, 8].to(torch.int64), dim=0, output_size=total)          # [rows, 8]
        per_row = lambda col, dt: expand[:, col].to(dt).unsqueeze(1)
        c = torch.cat([i.pred_corners for i in insts], dim=0)                       # [n, 8] = 4 x (x, y)
        x, y = c[:, 0::2], c[:, 1::2]
        x = torch.where(per_row(0, torch.bool), per_row(1, torch.float32) - x, x)
        y = torch.where(per_row(2, torch.bool), per_row(3, torch.float32) - y, y)
        x = x * per_row(4, torch.float32) * per_row(6, torch.float32)               # (x * r1) * r2, as two apply_coords calls
        y = y * per_row(5, torch.float32) * per_row(7, torch.float32)
        r = Instances(insts[0].image_size)
        r.scores = torch.cat([i.scores for i in insts], dim=0)
        r.centerness = torch.cat([i.centerness for i in insts], dim=0)
        r.pred_corners = torch.stack([x, y], dim=2).reshape(-1, 8)
        r.pred_classes = torch.cat([i.pred_classes for i in insts], dim=0)
        return r

    def _merge_detections(self, instances):
        return self.model.proposal_generator.dafne_outputs.select_over_all_levels([instances])[0]
