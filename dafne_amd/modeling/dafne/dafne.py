"""`DAFNe` proposal generator and `DAFNeHead` for the MI355X engine.

Same registry name, attributes and call contract as the reference
(dafne/modeling/dafne/dafne.py:69-164 DAFNe, :167-494 DAFNeHead): the default
center-to-corner branch with CORNER_TOWER_ON_CENTER_TOWER, CTR_ON_REG, GN towers
and per-level Scale.  The ablation branches (direct / iterative / offset / angle,
deformable convs, BN towers) are not built.
"""
import torch
from torch import nn

from ... import engine
from ... import postprocess as pp
from ...registry import PROPOSAL_GENERATOR_REGISTRY
from ..params import ConvParams, ScaleParams, cls_prior_bias, make_tower
from .dafne_outputs import DAFNeOutputs


def compute_locations(h, w, stride, device):
    """dafne.py:37-44 (kept for API parity; the decode kernel regenerates them)."""
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    return torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + stride // 2


class DAFNeHead(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        d = cfg.MODEL.DAFNE
        if d.CORNER_PREDICTION != "center-to-corner" or d.MERGE_CORNER_CENTER_PRED \
                or not d.CORNER_TOWER_ON_CENTER_TOWER or not d.CTR_ON_REG or d.NORM != "GN" \
                or d.USE_DEFORMABLE or d.NUM_SHARE_CONVS != 0 or not d.USE_SCALE \
                or d.NUM_CLS_CONVS != 4 or d.NUM_BOX_CONVS != 4 or d.CENTERNESS == "none":
            raise NotImplementedError("engine builds the released head: center-to-corner, stacked corner tower, "
                                      "CTR_ON_REG, GN, 4+4 convs, USE_SCALE")
        chans = set(s.channels for s in input_shape)
        assert len(chans) == 1, "Each level must have the same channel!"
        c = chans.pop()
        self.num_classes = d.NUM_CLASSES
        self.weight_dtype = cfg.ENGINE.WEIGHT_DTYPE       # bf16 | fp8_e4m3
        self.fpn_strides = d.FPN_STRIDES
        self.num_levels = len(input_shape)
        self.in_channels_to_top_module = c
        self.cls_tower = make_tower(c)
        self.corners_tower = make_tower(c)
        self.share_tower = nn.Sequential()
        self.center_tower = make_tower(c)
        self.cls_logits = ConvParams(self.num_classes, c, 3)
        self.ctrness = ConvParams(1, c, 3)
        self.corners_pred = ConvParams(8, c, 3)
        self.center_pred = ConvParams(2, c, 3)
        self.scales = nn.ModuleList([ScaleParams(1.0) for _ in range(self.num_levels)])
        for m in (self.cls_logits, self.ctrness, self.corners_pred, self.center_pred):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_logits.bias, cls_prior_bias(d.PRIOR_PROB))
        self._packed = None
        self._plans = {}

    def invalidate(self):
        self._packed = None
        self._plans = {}

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def _weights(self, device):
        if self._packed is None:
            self._packed = engine.pack_head_weights(self.state_dict(), device, prefix="",
                                                           fp8=self.weight_dtype == "fp8_e4m3")
        return self._packed

    def run_raw(self, feats):
        """feats: list of NCHW float tensors -> HeadPlan (after running it)."""
        dev = feats[0].device
        key = tuple(tuple(f.shape) for f in feats)
        if key not in self._plans:
            acts = [engine.Act(f.shape[0], f.shape[2], f.shape[3], f.shape[1], dev) for f in feats]
            cl = engine.CallList()
            hp = engine.HeadPlan(self._weights(dev), acts, self.num_classes, dev, engine.Pool(dev), cl)
            self._plans[key] = (acts, cl, hp)
        acts, cl, hp = self._plans[key]
        for a, f in zip(acts, feats):
            a.t[:, 1:-1, 1:-1, :] = f.permute(0, 2, 3, 1).to(torch.bfloat16)
        cl.run()
        return hp

    def forward(self, images, x, top_module=None, yield_corners_towers=False):
        """Reference return tuple (dafne.py:481-494): per-level NCHW lists
        (logits, corners_reg, center_reg, ltrb_reg, ctrness, top_feats, towers)."""
        with torch.cuda.device(x[0].device):
            hp = self.run_raw(list(x))
        logits, regs, centers, ctrs = [], [], [], []
        for l in range(len(x)):
            sc = hp.scales[l]
            dc = hp.delta_ctr[l]
            center = hp.center[l]
            reg = (center.repeat(1, 1, 1, 4) + dc[..., :8]) * sc
            logits.append(hp.logits[l].permute(0, 3, 1, 2).contiguous())
            regs.append(reg.permute(0, 3, 1, 2).contiguous())
            centers.append((center * sc).permute(0, 3, 1, 2).contiguous())
            ctrs.append(dc[..., 8:9].permute(0, 3, 1, 2).contiguous())
        return logits, regs, centers, [], ctrs, [], {"corners_towers": [], "center_towers": [], "cls_towers": []}


def head_levels(hp, strides):
    """HeadPlan outputs -> decode inputs (no copies: strided views into the fused
    [delta8|ctrness] buffer)."""
    levels = []
    for l, s in enumerate(strides):
        dc = hp.delta_ctr[l]
        levels.append(pp.LevelInput(hp.logits[l], dc, hp.center[l], dc.view(-1)[8:], s, hp.scales[l],
                                    delta_ps=9, center_ps=2, ctrness_ps=9))
    return levels


@PROPOSAL_GENERATOR_REGISTRY.register()
class DAFNe(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.in_features = cfg.MODEL.DAFNE.IN_FEATURES
        self.fpn_strides = cfg.MODEL.DAFNE.FPN_STRIDES
        self.yield_proposal = cfg.MODEL.DAFNE.YIELD_PROPOSAL
        if list(self.in_features) != ["p3", "p4", "p5", "p6", "p7"] or list(self.fpn_strides) != [8, 16, 32, 64, 128]:
            # the fused detector plan (engine.DensePlan) always builds P3..P7 of the ResNet-FPN and decodes them with these
            # strides; a subset would otherwise be decoded with the wrong levels' strides without a word
            raise NotImplementedError("engine builds the released pyramid: MODEL.DAFNE.IN_FEATURES p3..p7 with FPN_STRIDES 8..128 "
                                      "(got %s / %s)" % (list(self.in_features), list(self.fpn_strides)))
        self.dafne_head = DAFNeHead(cfg, [input_shape[f] for f in self.in_features])
        self.in_channels_to_top_module = self.dafne_head.in_channels_to_top_module
        self.dafne_outputs = DAFNeOutputs(cfg)

    def compute_locations(self, features):
        return [compute_locations(f.shape[-2], f.shape[-1], s, f.device)
                for f, s in zip(features, self.fpn_strides)]

    def forward(self, images, features, gt_instances=None, top_module=None):
        """(list[Instances], {}) like the reference's eval branch (dafne.py:145-156)."""
        if self.training:
            raise NotImplementedError("training is outside the scope of the MI355X inference engine")
        feats = [features[f] for f in self.in_features]
        with torch.cuda.device(feats[0].device):
            hp = self.dafne_head.run_raw(feats)
            rows, counts = self.dafne_outputs.predict_packed(head_levels(hp, self.fpn_strides))
        sizes = [tuple(int(v) for v in s) for s in images.image_sizes]
        return pp.rows_to_instances(rows, counts, sizes), {}
