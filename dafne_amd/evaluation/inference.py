"""`inference_on_dataset(model, data_loader, evaluator)` for the MI355X engine.

Reference: `do_test` hands every test set to detectron2's `inference_on_dataset` (tools/plain_train_net.py:316-336), whose
loop is [recalled]  `evaluator.reset(); for inputs in data_loader: outputs = model(inputs); evaluator.process(inputs,
outputs)`, then `evaluator.evaluate()`; `DafneEvaluator.process` moves every image's fields to the host
(dafne/evaluation/dafne_evaluator.py:44-58) and `evaluate()` gathers the per-rank lists on rank 0 (:60-64).

Here the loop is STREAMED: `model.forward_streamed(inputs)` enqueues batch i on the layout bench.py times (sub-batches on
concurrent streams, the previous batch's post-process on a side stream under this batch's head towers) and hands back the
outputs of the OLDEST batch in flight (batch i - 2 for the detector; None while the pipeline fills), which go to
`evaluator.process` while the later batches run; `model.flush()` drains the rest, oldest first, one batch per call.  The
evaluator sees exactly the (inputs, outputs) pairs of the synchronous loop, in the same order.  A model without `forward_streamed` (the TTA
wrapper) is called synchronously.
"""
import copy
import logging
import os
import time
from collections import OrderedDict

import torch

from ..utils.host import capped_torch_threads
from .gather import gather_detections, to_predictions
from .driver import instances_to_rows


class DatasetEvaluator:
    """detectron2's evaluator protocol [recalled]: reset() / process(inputs, outputs) / evaluate()."""

    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        return {}


class DafneEvaluator(DatasetEvaluator):
    """The collecting part of the reference's DafneEvaluator (dafne_evaluator.py:18-69): per image {image_id, file_name,
    height, width, labels, scores, corners, centerness} on the host; `evaluate()` lands every rank's predictions on rank 0 in
    global image order.  distributed: ONE collective pair over fixed-layout device buffers (gather.gather_detections: RCCL
    on the MI355X, gloo in the CPU tests) instead of the reference's pickled lists -- every rank must have processed the same
    number of images (the driver pads shards; `pad_to`).  The dataset-specific scoring (`_eval_predictions`: Task1 files, tile
    merge, voc_eval) lives in the subclasses -- DotaEvaluator (dota_evaluation.py), HrscEvaluator (hrsc_evaluation.py),
    UcasAodEvaluator (ucas_aod_evaluation.py); `get_evaluator` picks one by dataset name as tools/plain_train_net.py:171-214
    does -- which return the reference's {"task1": {class: ap, ..., "map": ...}} from evaluate()."""

    def __init__(self, dataset_name, cfg, distributed, output_dir=None, k_cap=None, device=None, pad_to=None, metadata=None):
        self._dataset_name, self._cfg, self._distributed, self._output_dir = dataset_name, cfg, distributed, output_dir
        self._metadata = metadata        # the reference reads MetadataCatalog.get(dataset_name): .root_dir, .is_test
        self._logger = logging.getLogger(__name__)
        self._k_cap = k_cap
        self._device = device
        self._pad_to = pad_to            # images every rank contributes to the gather (ceil(n / world): shards may be ragged)
        self._predictions = []
        self._meta = []
        self._insts = []

    def reset(self):
        self._predictions, self._meta, self._insts = [], [], []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            meta = {"image_id": inp["image_id"], "file_name": inp.get("file_name", ""), "height": inp["height"], "width": inp["width"]}
            self._meta.append(meta)
            pred = dict(meta)
            inst = out.get("instances")
            if self._gathers():
                # stays on the device: gathered as packed rows in evaluate() (no per-image copy / host sync here: the local
                # predictions would be discarded in favour of the gathered rows).  One entry per image, an output without
                # "instances" included (None -> an empty image), so that rows and self._meta stay aligned
                self._insts.append(inst)
            elif inst is not None:
                inst = inst.to(torch.device("cpu"))
                pred["labels"], pred["scores"] = inst.pred_classes, inst.scores
                pred["corners"], pred["centerness"] = inst.pred_corners, inst.centerness
            self._predictions.append(pred)

    def _gathers(self):
        import torch.distributed as dist
        return bool(self._distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def evaluate(self):
        pad_to = self._pad_to
        import torch.distributed as dist
        if self._gathers():
            if self._k_cap is None:
                raise RuntimeError("DafneEvaluator(distributed=True) needs k_cap (DAFNeOutputs.packed_k_cap())")
            real = [i for i in self._insts if i is not None]
            dev = self._device if self._device is not None else (real[0].scores.device if real else None)
            if dev is None:
                # an empty shard (or outputs without "instances"): the collectives below must still run on the backend's device --
                # under RCCL a CPU tensor here errors or hangs against the other ranks' CUDA tensors
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            rows, counts = instances_to_rows(self._insts, self._k_cap, dev)
            n = pad_to
            if n is None:
                # contiguous shards are ragged whenever N % world != 0: every rank contributes the LARGEST local count
                # (a MAX all-reduce; it used to be the local count, i.e. mismatched shapes in the gather)
                t = torch.tensor([len(self._insts)], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                n = int(t.item())
            if n < len(self._insts):
                raise RuntimeError("DafneEvaluator: pad_to=%d but this rank processed %d images" % (n, len(self._insts)))
            if n > rows.shape[0]:                     # equal shapes on every rank: pad with empty images
                pad = n - rows.shape[0]
                rows = torch.cat([rows, rows.new_zeros((pad,) + tuple(rows.shape[1:]))])
                counts = torch.cat([counts, counts.new_zeros(pad)])
            out = gather_detections(rows, counts, dst=0)
            # the images' own fields (image_id -- an int for DOTA / HRSC, a string for UCAS-AOD --, file_name, height, width)
            # travel as the small pickled lists the reference gathers everything in (dafne_evaluator.py:62); a rank's list is in
            # the order of its rows, its length tells the padding apart
            metas = [None] * dist.get_world_size()
            dist.all_gather_object(metas, self._meta)
            if out is None:
                return {}
            rows_all, counts_all = out
            preds = to_predictions(rows_all, counts_all, image_ids=list(range(rows_all.shape[0])))
            predictions = []
            for r, ms in enumerate(metas):
                for j, m in enumerate(ms):
                    p = preds[r * n + j]
                    p.update(m)
                    predictions.append(p)
        else:
            predictions = self._predictions
        if not hasattr(self, "_eval_predictions"):
            self._results = {"predictions": predictions, "num_images": len(predictions)}
            return self._results
        # a dataset evaluator (DotaEvaluator / HrscEvaluator / UcasAodEvaluator): dafne_evaluator.py:69-84
        if len(predictions) == 0:
            self._logger.warning("[DafneEvaluator] Did not receive valid predictions.")
            return {}
        if self._output_dir:
            os.makedirs(self._output_dir, exist_ok=True)
            torch.save(predictions, os.path.join(self._output_dir, "instances_predictions.pth"))
        self._results = OrderedDict()
        self._eval_predictions(predictions)
        return copy.deepcopy(self._results)


def inference_on_dataset(model, data_loader, evaluator=None, stats=None):
    """Run `model` over `data_loader` (an iterable of batches: list[dict] with "image" uint8 CHW BGR, "height", "width", ...)
    and feed `evaluator` (reset / process / evaluate).  Returns evaluator.evaluate() (or the list of (inputs, outputs) pairs'
    outputs when there is no evaluator).  stats (dict, optional) receives images, seconds and images_per_sec of the loop
    (wall clock around the loop + the final drain, device synchronised on both sides)."""
    if evaluator is not None:
        evaluator.reset()
    streamed = hasattr(model, "forward_streamed")
    collected = []

    def deliver(inputs, outputs):
        if evaluator is not None:
            evaluator.process(inputs, outputs)
        else:
            collected.extend(outputs)
    cuda = torch.cuda.is_available()
    if cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    in_flight = []                       # inputs of the batches the streamed model has not handed back yet, oldest first
    was_training = getattr(model, "training", False)
    if hasattr(model, "eval"):
        model.eval()
    # the loop owns the cap on torch's intra-op pool (restored on exit): its host work is latency work
    with torch.no_grad(), capped_torch_threads(8):
        for inputs in data_loader:
            n += len(inputs)
            if streamed:
                in_flight.append(inputs)
                out_old = model.forward_streamed(inputs)         # the outputs of the OLDEST batch in flight, or None
                if out_old is not None:
                    deliver(in_flight.pop(0), out_old)
            else:
                deliver(inputs, model(inputs))
        if streamed:
            while in_flight:
                out_old = model.flush()
                if out_old is None:
                    raise RuntimeError("inference_on_dataset: the model dropped %d batch(es) in flight" % len(in_flight))
                deliver(in_flight.pop(0), out_old)
    if cuda:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if was_training and hasattr(model, "train"):
        model.train()
    if stats is not None:
        stats.update(images=n, seconds=dt, images_per_sec=n / dt if dt > 0 else float("inf"))
    return evaluator.evaluate() if evaluator is not None else collected


def get_evaluator(cfg, dataset_name, output_folder=None, metadata=None, distributed=True, **kw):
    """tools/plain_train_net.py:171-214: the evaluator of a dataset, by its name ("dota", "hrsc", "ucas"; the reference's
    fourth, ICDAR15, has no released config and no counterpart here).  metadata: an object with .root_dir (and .is_test for
    DOTA) -- what the reference takes from MetadataCatalog.get(dataset_name).  kw: k_cap / device / pad_to of DafneEvaluator."""
    if output_folder is None:
        output_folder = os.path.join(cfg.OUTPUT_DIR, "inference", dataset_name)
    if metadata is None:
        from ..data.datasets import MetadataCatalog          # filled by data.datasets.register_dota / _hrsc / _ucas_aod
        if dataset_name in MetadataCatalog:
            metadata = MetadataCatalog.get(dataset_name)
    name = dataset_name.lower()
    if "dota" in name:
        from .dota_evaluation import DotaEvaluator as cls
    elif "hrsc" in name:
        from .hrsc_evaluation import HrscEvaluator as cls
    elif "ucas" in name:
        from .ucas_aod_evaluation import UcasAodEvaluator as cls
    else:
        raise NotImplementedError("no Evaluator for the dataset %r (dota / hrsc / ucas)" % (dataset_name,))
    return cls(dataset_name=dataset_name, cfg=cfg, distributed=distributed, output_dir=output_folder, metadata=metadata, **kw)
