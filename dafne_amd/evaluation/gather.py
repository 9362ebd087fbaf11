"""Final detection gather (the path's only collective).

Reference: DafneEvaluator.evaluate -> comm.synchronize(); comm.gather(predictions,
dst=0) (dafne/evaluation/dafne_evaluator.py:60-64): pickled Python lists over a
gloo side group.  Here: fixed-layout device buffers -- rows [B_local, k_cap, 18]
float32 + counts [B_local] int32 -- moved by ONE collective per tensor on the
default process group (RCCL on MI355X: every rank's ~0.7 MB goes out over all
xGMI links at once; gloo in the CPU tests).  Images are sharded contiguously:
rank r owns global images [r*B_local, (r+1)*B_local).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous shard like detectron2's InferenceSampler: rank r gets
    [r*ceil(n/world), ...) clipped to n."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def gather_detections(rows, counts, dst=0, group=None):
    """All ranks call this with equal-shape tensors.  Returns on rank `dst`
    (rows_all [world*B_local, k_cap, 18], counts_all [world*B_local]); None elsewhere.
    Single-process (no initialised group): returns the inputs."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rows, counts
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = rows.contiguous()
    counts = counts.contiguous()
    if rank == dst:
        rows_all = rows.new_empty((world,) + tuple(rows.shape))
        counts_all = counts.new_empty((world,) + tuple(counts.shape))
        rl = list(rows_all.unbind(0))
        cl = list(counts_all.unbind(0))
    else:
        rows_all = counts_all = rl = cl = None
    dist.gather(rows, rl, dst=dst, group=group)
    dist.gather(counts, cl, dst=dst, group=group)
    if rank != dst:
        return None
    return rows_all.reshape((-1,) + tuple(rows.shape[1:])), counts_all.reshape(-1)


def to_predictions(rows_all, counts_all, image_ids=None):
    """Gathered buffers -> the evaluator's per-image dicts
    (dafne_evaluator.py:44-58: labels, scores, corners, centerness)."""
    counts = counts_all.cpu().tolist()
    r = rows_all.cpu()
    preds = []
    for i, k in enumerate(counts):
        d = r[i, :k]
        preds.append({"image_id": image_ids[i] if image_ids is not None else i,
                      "labels": d[:, 10].to(torch.int64), "scores": d[:, 8],
                      "corners": d[:, 0:8], "centerness": d[:, 9]})
    return preds
