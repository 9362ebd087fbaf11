"""Device-side post-process driver: decode -> rotated NMS + cap -> gather.

Thin host glue over the C-ABI entry points (include/dafne_amd.h); every tensor
stays on the GPU and no step synchronises with the host.  Mirrors, in one fused
pipeline, DAFNeOutputs.predict_proposals (dafne/modeling/dafne/dafne_outputs.py:
733-790) and OneStageDetector._postprocess (dafne/modeling/one_stage_detector.py:
79-98).
"""
import ctypes

import torch

from . import _lib
from .structures import Boxes, Instances


class LevelInput:
    """One FPN level's raw head outputs in NHWC fp32 (device)."""

    def __init__(self, logits, delta, center, ctrness, stride, scale=1.0,
                 delta_ps=None, center_ps=None, ctrness_ps=None, logits_ps=None):
        self.logits, self.delta, self.center, self.ctrness = logits, delta, center, ctrness
        self.stride, self.scale = int(stride), float(scale)
        n, h, w = logits.shape[0], logits.shape[1], logits.shape[2]
        self.N, self.H, self.W = n, h, w
        self.logits_ps = logits_ps or logits.shape[3]
        self.delta_ps = delta_ps or 8
        self.center_ps = center_ps or 2
        self.ctrness_ps = ctrness_ps or 1


class Candidates:
    """Per-image candidate rows after decode (device, fixed capacity)."""

    def __init__(self, n, m_cap, device):
        f32, i32 = torch.float32, torch.int32
        self.n, self.m_cap = n, m_cap
        self.corners = torch.empty(n, m_cap, 8, dtype=f32, device=device)
        self.scores = torch.empty(n, m_cap, dtype=f32, device=device)
        self.ctr = torch.empty(n, m_cap, dtype=f32, device=device)
        self.classes = torch.empty(n, m_cap, dtype=i32, device=device)
        self.locs = torch.empty(n, m_cap, 2, dtype=f32, device=device)
        self.levels = torch.empty(n, m_cap, dtype=i32, device=device)
        self.hbox = torch.empty(n, m_cap, 4, dtype=f32, device=device)
        self.counts = torch.zeros(n, dtype=i32, device=device)


def decode_levels(levels, *, num_classes, pre_nms_thresh, pre_nms_topk, thresh_with_ctr,
                  sort_corners, out=None):
    """forward_for_single_feature_map over all levels/images -> Candidates."""
    L = _lib.load()
    dev = levels[0].logits.device
    n = levels[0].N
    prm = _lib.DecodeParams(n, len(levels), num_classes, pre_nms_topk, float(pre_nms_thresh),
                            int(bool(thresh_with_ctr)), int(bool(sort_corners)),
                            len(levels) * pre_nms_topk)
    descs = (_lib.LevelDesc * len(levels))()
    for i, lv in enumerate(levels):
        for t in (lv.logits, lv.delta, lv.center, lv.ctrness):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        descs[i] = _lib.LevelDesc(lv.logits.data_ptr(), lv.delta.data_ptr(), lv.center.data_ptr(),
                                  lv.ctrness.data_ptr(), lv.logits_ps, lv.delta_ps, lv.center_ps,
                                  lv.ctrness_ps, lv.H, lv.W, lv.stride, lv.scale)
    with torch.cuda.device(dev):
        cand = out if out is not None else Candidates(n, prm.m_cap, dev)
        assert cand.m_cap == prm.m_cap and cand.n == n
        nbytes = L.dafne_decode_workspace_bytes(ctypes.byref(prm), descs)
        if nbytes == 0:
            _lib.check(1, "dafne_decode_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.dafne_decode_levels_hip(
            ctypes.byref(prm), descs, _lib.ptr(cand.corners), _lib.ptr(cand.scores), _lib.ptr(cand.ctr),
            _lib.ptr(cand.classes), _lib.ptr(cand.locs), _lib.ptr(cand.levels), _lib.ptr(cand.hbox),
            _lib.ptr(cand.counts), _lib.ptr(ws), nbytes, _lib.current_stream()), "dafne_decode_levels_hip")
    return cand


def select(cand, nms_thresh, post_topk, nms_flags=0):
    """ml_nms + cap for every image of the batch.  Returns (keep[N,m_cap] int64,
    num_keep[N] int32), device tensors.  nms_flags: per-call DAFNE_NMS_* bits (parity runs)."""
    L = _lib.load()
    dev = cand.corners.device
    n, m = cand.n, cand.m_cap
    with torch.cuda.device(dev):
        keep = torch.empty(n, m, dtype=torch.int64, device=dev)
        nk = torch.zeros(n, dtype=torch.int32, device=dev)
        nbytes = L.dafne_poly_nms_workspace_bytes(n, m)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        _lib.check(L.dafne_select_over_all_levels_hip(
            _lib.ptr(cand.corners), _lib.ptr(cand.scores), _lib.ptr(cand.classes), _lib.ptr(cand.counts),
            n, m, float(nms_thresh), int(post_topk), _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes,
            int(nms_flags), _lib.current_stream()), "dafne_select_over_all_levels_hip")
    return keep, nk


def gather(cand, keep, num_keep, sizes=None, k_cap=None, scale_corners=True):
    """Kept rows -> [N, k_cap, DET_ROW] float32 (+ detector_postprocess when
    ``sizes`` [N,6] = (net_h, net_w, out_h, out_w, orig_h, orig_w) is given)."""
    L = _lib.load()
    dev = cand.corners.device
    n, m = cand.n, cand.m_cap
    k_cap = k_cap or m
    with torch.cuda.device(dev):
        out = torch.empty(n, k_cap, _lib.DET_ROW, dtype=torch.float32, device=dev)
        cnt = torch.zeros(n, dtype=torch.int32, device=dev)
        if sizes is not None and not (isinstance(sizes, torch.Tensor) and sizes.is_cuda):
            sizes = torch.as_tensor(sizes, dtype=torch.float32).reshape(n, 6).to(dev, non_blocking=True)
        _lib.check(L.dafne_gather_detections_hip(
            _lib.ptr(cand.corners), _lib.ptr(cand.scores), _lib.ptr(cand.ctr), _lib.ptr(cand.classes),
            _lib.ptr(cand.locs), _lib.ptr(cand.levels), _lib.ptr(cand.hbox), _lib.ptr(keep),
            _lib.ptr(num_keep), _lib.ptr(sizes), (2 if scale_corners else 1) if sizes is not None else 0, n, m, k_cap, _lib.ptr(out),
            _lib.ptr(cnt), _lib.current_stream()), "dafne_gather_detections_hip")
    return out, cnt


def _instances_of_rows(r, image_size):
    inst = Instances(tuple(image_size))
    inst.pred_boxes = Boxes(r[:, 12:16])
    inst.pred_corners = r[:, 0:8]
    inst.scores = r[:, 8]
    inst.centerness = r[:, 9]
    inst.pred_classes = r[:, 10].to(torch.int64)
    inst.locations = r[:, 16:18]
    inst.fpn_levels = r[:, 11].to(torch.int64)
    return inst


def rows_to_instances(rows, counts, image_sizes, host_rows=None):
    """[N,k_cap,DET_ROW] + counts -> list[Instances] with the reference's fields
    (dafne_outputs.py:879-903, :777-780).  One host sync (counts).
    host_rows: the same rows already on the host (a landed copy, e.g. forward_streamed's pinned staging buffer): every
    Instances gets a host twin built from its slice (cloned: the staging buffer is reused), which `Instances.to("cpu")` hands
    out instead of copying seven fields per image from the device -- the reference's evaluators call .to(cpu) on every image
    (dafne_evaluator.py:48-55), 56 synchronous copies per batch of 8 = 5 ms of host time against 5.8 ms of GPU time per batch."""
    counts = counts.cpu().tolist()
    res = []
    for i, k in enumerate(counts):
        if k > rows.shape[1]:
            raise _lib.DafneHipError("detections of image %d (%d) exceed the output capacity %d"
                                     % (i, k, rows.shape[1]))
        inst = _instances_of_rows(rows[i, :k], image_sizes[i])
        if host_rows is not None:
            hr, size = host_rows[i, :k].clone(), image_sizes[i]
            if hasattr(inst, "attach_cpu_twin"):          # (detectron2's own Instances, when importable, copies field by field)
                inst.attach_cpu_twin(lambda hr=hr, size=size: _instances_of_rows(hr, size))
        res.append(inst)
    return res


def sort_quadrilateral(bboxes):
    """dafne.utils.sort_corners.sort_quadrilateral on the GPU ([n,8] float32)."""
    assert bboxes.dim() == 2 and bboxes.shape[1] == 8
    if bboxes.shape[0] == 0:
        return bboxes
    if not bboxes.is_cuda:
        raise _lib.DafneHipError("sort_quadrilateral: the MI355X engine has no CPU path")
    L = _lib.load()
    b = bboxes.to(torch.float32).contiguous()
    out = torch.empty_like(b)
    with torch.cuda.device(b.device):
        _lib.check(L.dafne_sort_quadrilateral_hip(_lib.ptr(b), _lib.ptr(out), b.shape[0], _lib.current_stream()),
                   "dafne_sort_quadrilateral_hip")
    return out
