"""One-process-per-GPU inference over a list of images (the `--eval-only` loop of the reference).

Reference flow: `launch(main, num_gpus, ...)` starts one process per GPU (tools/plain_train_net.py:660-671); each
process runs detectron2's `inference_on_dataset` over ITS contiguous slice of the dataset (InferenceSampler
[recalled]), `DafneEvaluator.process` moves every image's fields to the host (dafne_evaluator.py:44-58) and
`evaluate()` does `comm.synchronize(); comm.gather(predictions, dst=0)` + chain (:60-64) -- pickled lists over gloo.

Here: rank r owns global images [r*per, (r+1)*per) with per = ceil(n / world) (`gather.shard_range`), runs them in
batches through the detector, keeps the detections packed on the device ([b, k_cap, 18] rows + counts), pads its
shard to `per` images (counts 0) so that every rank contributes equal shapes, and ONE collective pair
(`gather.gather_detections`: RCCL on MI355X, gloo in the CPU tests) lands everything on rank 0 in global image order.
No data-path collective besides that gather; images and weights never cross ranks.
"""
import torch

from .. import _lib
from .gather import gather_detections, shard_range


def instances_to_rows(instances, k_cap, device=None):
    """list[Instances] (the reference-signature output) -> packed ([n, k_cap, 18] float32, [n] int32): the inverse of
    postprocess.rows_to_instances, used to put TTA results (one merged Instances per image) on the gather path."""
    n = len(instances)
    real = [i for i in instances if i is not None]           # None = an image without detections output (an empty row)
    dev = device if device is not None else (real[0].scores.device if real else torch.device("cpu"))
    rows = torch.zeros(n, k_cap, _lib.DET_ROW, dtype=torch.float32, device=dev)
    counts = torch.zeros(n, dtype=torch.int32, device=dev)
    for i, inst in enumerate(instances):
        k = 0 if inst is None else len(inst)
        if k > k_cap:
            raise _lib.DafneHipError("image %d has %d detections, more than the gather capacity %d" % (i, k, k_cap))
        counts[i] = k
        if k == 0:
            continue
        r = rows[i, :k]
        r[:, 0:8] = inst.pred_corners
        r[:, 8] = inst.scores
        r[:, 9] = inst.centerness
        r[:, 10] = inst.pred_classes.to(torch.float32)
        if inst.has("fpn_levels"):
            r[:, 11] = inst.fpn_levels.to(torch.float32)
        if inst.has("pred_boxes"):
            r[:, 12:16] = inst.pred_boxes.tensor
        if inst.has("locations"):
            r[:, 16:18] = inst.locations
    return rows, counts


def inference_on_images(detect_batch, n_total, k_cap, batch_size=1, rank=0, world=1, device="cpu", group=None):
    """detect_batch(lo, hi) -> (rows [hi-lo, k_cap, 18] float32, counts [hi-lo] int32) on `device` for the GLOBAL
    images [lo, hi).  Every rank calls this; rank 0 gets (rows_all [n_total, k_cap, 18], counts_all [n_total]) in
    global image order, the others None."""
    lo, hi = shard_range(n_total, rank, world)
    per = (n_total + world - 1) // world if world > 0 else n_total
    rows = torch.zeros(max(per, 0), k_cap, _lib.DET_ROW, dtype=torch.float32, device=device)
    counts = torch.zeros(max(per, 0), dtype=torch.int32, device=device)
    for b0 in range(lo, hi, max(int(batch_size), 1)):
        b1 = min(b0 + max(int(batch_size), 1), hi)
        r, c = detect_batch(b0, b1)
        if tuple(r.shape) != (b1 - b0, k_cap, _lib.DET_ROW) or tuple(c.shape) != (b1 - b0,):
            raise _lib.DafneHipError("detect_batch returned %s / %s for images [%d, %d) with k_cap %d"
                                     % (tuple(r.shape), tuple(c.shape), b0, b1, k_cap))
        rows[b0 - lo:b1 - lo] = r
        counts[b0 - lo:b1 - lo] = c.to(torch.int32)
    out = gather_detections(rows, counts, dst=0, group=group)      # padded shards: equal shapes on every rank
    if out is None:
        return None
    rows_all, counts_all = out
    return rows_all[:n_total], counts_all[:n_total]                 # the padding sits behind the last real image
