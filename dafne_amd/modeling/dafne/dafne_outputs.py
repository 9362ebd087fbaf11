"""Inference half of the reference's DAFNeOutputs
(dafne/modeling/dafne/dafne_outputs.py:123-190 config, :733-925 inference), with
the same method names and argument meaning, backed by the HIP post-process.

Training (target assignment, losses; :44-731) is out of scope for this engine.
"""
import torch
from torch import nn

from ... import _lib
from ... import postprocess as pp
from ...structures import Instances
from ..nms.nms import ml_nms  # noqa: F401  (re-exported: the reference module imports it here)


class DAFNeOutputs(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d = cfg.MODEL.DAFNE
        self.pre_nms_thresh_test = d.INFERENCE_TH_TEST
        self.pre_nms_topk_test = d.PRE_NMS_TOPK_TEST
        self.post_nms_topk_test = d.POST_NMS_TOPK_TEST
        self.pre_nms_thresh = self.pre_nms_thresh_test
        self.pre_nms_topk = self.pre_nms_topk_test
        self.post_nms_topk = self.post_nms_topk_test
        self.nms_thresh = d.NMS_TH
        self.thresh_with_ctr = d.THRESH_WITH_CTR
        self.sort_corners = d.SORT_CORNERS
        self.centerness_mode = d.CENTERNESS
        self.has_centerness = self.centerness_mode != "none"
        assert self.centerness_mode in ["none", "plain", "oriented"]
        if not self.has_centerness:
            raise NotImplementedError("CENTERNESS='none' is not used by any released config")
        self.corner_prediction_strategy = d.CORNER_PREDICTION
        self.num_classes = d.NUM_CLASSES
        self.strides = d.FPN_STRIDES
        self.stride_norm = d.ENABLE_FPN_STRIDE_NORM

    def losses(self, *a, **k):
        raise NotImplementedError("training is outside the scope of the MI355X inference engine")

    # ---- fused device path used by the engine ---------------------------------
    def predict_packed(self, levels, sizes=None, k_cap=None, scale_corners=True):
        """levels: list[postprocess.LevelInput] (NHWC fp32).  Returns (rows, counts):
        [N,k_cap,18] float32 detections and their per-image counts, on the GPU."""
        cand = self.decode_packed(levels)
        return self.select_packed(cand, sizes=sizes, k_cap=k_cap, scale_corners=scale_corners)

    def decode_packed(self, levels, out=None):
        if not self.stride_norm:
            # ENABLE_FPN_STRIDE_NORM false (dafne_outputs.py:771-774): the regression is already in pixels.  The decode kernel
            # computes (reg * scale) * stride; with scale / stride for a power-of-two stride both products only shift the
            # exponent, so the result has the bits of reg * scale (no other rounding: tests/test_gpu_decode.py)
            adj = []
            for lv in levels:
                if lv.stride <= 0 or lv.stride & (lv.stride - 1):
                    raise NotImplementedError("ENABLE_FPN_STRIDE_NORM=False with an FPN stride that is not a power of two (%d)" % lv.stride)
                adj.append(pp.LevelInput(lv.logits, lv.delta, lv.center, lv.ctrness, lv.stride, lv.scale / lv.stride,
                                         delta_ps=lv.delta_ps, center_ps=lv.center_ps, ctrness_ps=lv.ctrness_ps,
                                         logits_ps=lv.logits_ps))
            levels = adj
        return pp.decode_levels(levels, num_classes=self.num_classes, pre_nms_thresh=self.pre_nms_thresh_test,
                                pre_nms_topk=self.pre_nms_topk_test, thresh_with_ctr=self.thresh_with_ctr,
                                sort_corners=self.sort_corners, out=out)

    def packed_k_cap(self, n_levels=5):
        """Row capacity of the packed detections select_packed returns by default -- what a caller sizing a gather
        buffer must use (tools/eval_net.py, evaluation/driver.py) instead of restating the rule."""
        m_cap = n_levels * self.pre_nms_topk_test
        return min(m_cap, max(self.post_nms_topk_test, 1) + 256) if self.post_nms_topk_test > 0 else m_cap

    def select_packed(self, cand, sizes=None, k_cap=None, scale_corners=True):
        if self.nms_thresh > 0:
            keep, nk = pp.select(cand, self.nms_thresh, self.post_nms_topk_test)
        else:   # ml_nms returns its input unchanged (nms.py:22-23); only the cap applies
            keep, nk = _identity_keep_with_cap(cand, self.post_nms_topk_test)
        if k_cap is None:
            k_cap = self.packed_k_cap(cand.m_cap // max(self.pre_nms_topk_test, 1))
        return pp.gather(cand, keep, nk, sizes=sizes, k_cap=k_cap, scale_corners=scale_corners)

    # ---- reference-signature path ----------------------------------------------
    def predict_proposals(self, logits_pred, corners_reg_pred, ctrness_pred, locations, image_sizes,
                          top_feats=None):
        """Same arguments as the reference (:733-741): per-level NCHW tensors, with
        corners_reg_pred already (center.repeat + delta) * scale.  ``locations`` is
        accepted for signature parity; the kernel regenerates them (dafne.py:37-44)."""
        levels = []
        for lg, rc, ct, s in zip(logits_pred, corners_reg_pred, ctrness_pred, self.strides):
            n, _, h, w = lg.shape
            lg_n = lg.detach().float().permute(0, 2, 3, 1).contiguous()
            rc_n = rc.detach().float().permute(0, 2, 3, 1).contiguous()
            ct_n = ct.detach().float().permute(0, 2, 3, 1).contiguous()
            zero = torch.zeros(n, h, w, 2, dtype=torch.float32, device=lg.device)
            levels.append(pp.LevelInput(lg_n, rc_n, zero, ct_n, s, 1.0))
        rows, counts = self.predict_packed(levels)
        sizes = [tuple(int(v) for v in (s.tolist() if isinstance(s, torch.Tensor) else s)) for s in image_sizes]
        return pp.rows_to_instances(rows, counts, sizes)

    def select_over_all_levels(self, boxlists):
        """:907-925, Instances in / Instances out (used by the TTA merge, tta.py:265): ml_nms, then the kthvalue cap
        with ties.  Both steps are one device call (dafne_select_over_all_levels_hip) and one host wait per image."""
        results = []
        for bl in boxlists:
            n = len(bl)
            if self.nms_thresh <= 0 or n == 0:          # ml_nms returns its input unchanged (nms.py:22-25)
                result = bl
                if n > self.post_nms_topk > 0:
                    s = result.scores
                    thr = torch.kthvalue(s, n - self.post_nms_topk + 1).values
                    result = result[torch.nonzero(s >= thr).squeeze(1)]
                results.append(result)
                continue
            if not bl.pred_corners.is_cuda:
                raise _lib.DafneHipError("select_over_all_levels: the MI355X engine has no CPU path (got CPU tensors)")
            L = _lib.load()
            dev = bl.pred_corners.device
            with torch.cuda.device(dev):
                b = bl.pred_corners.detach().to(torch.float32).contiguous()
                sc = bl.scores.detach().to(torch.float32).contiguous()
                c = bl.pred_classes.detach().to(torch.int32).contiguous()
                keep = torch.empty(n, dtype=torch.int64, device=dev)
                nk = torch.zeros(1, dtype=torch.int32, device=dev)
                nbytes = L.dafne_poly_nms_workspace_bytes(1, n)
                ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
                _lib.check(L.dafne_select_over_all_levels_hip(
                    _lib.ptr(b), _lib.ptr(sc), _lib.ptr(c), None, 1, n, float(self.nms_thresh),
                    int(max(self.post_nms_topk, 0)), _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, 0,
                    _lib.current_stream()), "dafne_select_over_all_levels_hip")
                results.append(bl[keep[: int(nk.item())]])
        return results


def _identity_keep_with_cap(cand, post_topk):
    n, m = cand.n, cand.m_cap
    dev = cand.scores.device
    keep = torch.arange(m, device=dev, dtype=torch.int64).repeat(n, 1)
    nk = cand.counts.clone()
    if post_topk > 0:
        raise NotImplementedError("NMS_TH <= 0 with a post-NMS cap is not used by any released config")
    return keep, nk
