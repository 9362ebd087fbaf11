"""Rotated NMS front end: same names and argument meaning as the reference's
dafne/modeling/nms/nms.py (ml_nms :10-33, batched_nms_poly :37-92) and the
external poly_nms.poly_gpu_nms it calls (:91), backed by libdafne_amd.so.

No CPU path: tensors must live on the GPU (poly_gpu_nms takes a host numpy
array like the reference's extension and does the copies itself).
"""
import ctypes

import numpy as np
import torch

from ... import _lib


def _ws(n_images, m_cap, device):
    L = _lib.load()
    nbytes = L.dafne_poly_nms_workspace_bytes(n_images, m_cap)
    return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device), nbytes


def poly_gpu_nms(dets, thresh, device_id=0):
    """poly_nms.poly_gpu_nms(dets[M,9] float32 host array, thresh, device_id) ->
    list of kept row indices, descending score."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.ndim != 2 or dets.shape[1] != 9:
        raise ValueError("dets must be [M,9] (8 corner coordinates + score)")
    m = dets.shape[0]
    if m == 0:
        return []
    L = _lib.load()
    dev = torch.device("cuda", device_id)
    with torch.cuda.device(dev):
        d = torch.from_numpy(dets).to(dev)
        keep = torch.empty(m, dtype=torch.int64, device=dev)
        nk = torch.zeros(1, dtype=torch.int32, device=dev)
        ws, nbytes = _ws(1, m, dev)
        _lib.check(L.dafne_poly_nms_hip(_lib.ptr(d), m, float(thresh), _lib.ptr(keep), _lib.ptr(nk),
                                        _lib.ptr(ws), nbytes, 0, _lib.current_stream()), "dafne_poly_nms_hip")
        n = int(nk.item())
        return keep[:n].cpu().tolist()


def batched_nms_poly(boxes, scores, idxs, iou_threshold):
    """Class-aware polygon NMS (nms.py:37-92): class 5 is merged into 4, every
    class is shifted by float(class)*(max-min+1) in fp32, then greedy NMS.
    Returns an int64 tensor of kept indices, descending score."""
    assert boxes.shape[-1] == 8
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if not boxes.is_cuda:
        raise _lib.DafneHipError("batched_nms_poly: the MI355X engine has no CPU path (got a CPU tensor)")
    L = _lib.load()
    m = boxes.shape[0]
    dev = boxes.device
    with torch.cuda.device(dev):
        b = boxes.detach().to(torch.float32).contiguous()
        s = scores.detach().to(torch.float32).contiguous()
        c = idxs.detach().to(torch.int32).contiguous()
        keep = torch.empty(m, dtype=torch.int64, device=dev)
        nk = torch.zeros(1, dtype=torch.int32, device=dev)
        ws, nbytes = _ws(1, m, dev)
        _lib.check(L.dafne_select_over_all_levels_hip(
            _lib.ptr(b), _lib.ptr(s), _lib.ptr(c), None, 1, m, float(iou_threshold), 0,
            _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, 0, _lib.current_stream()),
            "dafne_select_over_all_levels_hip")
        return keep[: int(nk.item())]


def ml_nms(boxlist, nms_thresh, max_proposals=-1):
    """nms.py:10-33 on an Instances with pred_corners / scores / pred_classes."""
    if nms_thresh <= 0:
        return boxlist
    if boxlist.scores.shape[0] == 0:
        return boxlist
    keep = batched_nms_poly(boxlist.pred_corners, boxlist.scores, boxlist.pred_classes, nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep]


def poly_iou_pairs(p, q):
    """fp64 IoU of quad pairs ([n,8] each) on the GPU; the device twin of the
    reference's polyiou.iou_poly (tools/prepare_dota/polyiou.cpp:112)."""
    L = _lib.load()
    p = p.to(torch.float64).contiguous()
    q = q.to(torch.float64).contiguous()
    out = torch.empty(p.shape[0], dtype=torch.float64, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(L.dafne_poly_iou_pairs_hip(_lib.ptr(p), _lib.ptr(q), p.shape[0], _lib.ptr(out),
                                              _lib.current_stream()), "dafne_poly_iou_pairs_hip")
    return out
