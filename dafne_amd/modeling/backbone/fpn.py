"""`build_dafne_resnet_fpn_backbone` for the MI355X engine.

Same registry name, constructor arguments and call contract as the reference
(dafne/modeling/backbone/fpn.py:58-91): a module mapping a normalised NCHW image
batch to {"p3".."p7"}, with `.size_divisibility == 32` and `.output_shape()`.
ResNet body / FPN follow detectron2 v0.5 [recalled, SURVEY appendix B];
LastLevelP6P7 consumes the P5 OUTPUT (fpn.py:25,78).
"""
import torch
from torch import nn

from ... import engine
from ...registry import BACKBONE_REGISTRY
from ...structures import ShapeSpec
from ..params import ConvParams, ResNetParams, TopBlockParams


class ResNetFPNBackbone(nn.Module):
    def __init__(self, depth, out_channels=256, weight_dtype="bf16"):
        super().__init__()
        self.depth = depth
        self.bottom_up = ResNetParams(depth)
        for lvl, c in ((3, 512), (4, 1024), (5, 2048)):
            setattr(self, "fpn_lateral%d" % lvl, ConvParams(out_channels, c, 1))
            setattr(self, "fpn_output%d" % lvl, ConvParams(out_channels, out_channels, 3))
        self.top_block = TopBlockParams(out_channels)
        self._out_channels = out_channels
        self.weight_dtype = weight_dtype          # cfg.ENGINE.WEIGHT_DTYPE: bf16 | fp8_e4m3
        self._packed = None
        self._plans = {}

    @property
    def size_divisibility(self):
        return 32

    def output_shape(self):
        return {"p%d" % l: ShapeSpec(channels=self._out_channels, stride=2 ** l) for l in range(3, 8)}

    def _weights(self, device):
        if self._packed is None:
            self._packed = engine.pack_backbone_weights(self.state_dict(), self.depth, device, prefix="",
                                                               fp8=self.weight_dtype == "fp8_e4m3")
        return self._packed

    def invalidate(self):
        self._packed = None
        self._plans = {}

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def forward(self, x):
        """x: [N,3,H,W] float, already normalised and padded to a multiple of 32."""
        if not x.is_cuda:
            raise RuntimeError("the MI355X engine has no CPU path")
        n, _, h, w = x.shape
        key = (n, h, w)
        with torch.cuda.device(x.device):
            if key not in self._plans:
                self._plans[key] = engine.DensePlan(self._weights(x.device), n, h, w, self.depth, 0, x.device,
                                                    with_head=False)
            plan = self._plans[key]
            plan.stem_in[:, 3:-3, 3:-3, :3] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
            plan.run()
            return {"p%d" % (i + 3): f.nchw_float() for i, f in enumerate(plan.features)}


@BACKBONE_REGISTRY.register()
def build_dafne_resnet_fpn_backbone(cfg, input_shape=None):
    if cfg.MODEL.BACKBONE.ANTI_ALIAS or cfg.MODEL.RESNETS.DEFORM_INTERVAL > 1:
        raise NotImplementedError("anti-aliased / deformable ResNets are not used by any released config")
    r = cfg.MODEL.RESNETS
    if r.DEPTH not in (50, 101) or r.NORM != "FrozenBN" or not r.STRIDE_IN_1X1 or r.NUM_GROUPS != 1 \
            or r.RES5_DILATION != 1:
        raise NotImplementedError("engine supports ResNet-50/101, FrozenBN, STRIDE_IN_1X1, no groups/dilation")
    if cfg.MODEL.FPN.NORM not in ("", None) or cfg.MODEL.FPN.FUSE_TYPE != "sum":
        raise NotImplementedError("engine supports FPN without norm, fuse type 'sum'")
    if cfg.MODEL.DAFNE.TOP_LEVELS != 2:
        raise NotImplementedError("engine supports TOP_LEVELS == 2 (P6 and P7)")
    if cfg.ENGINE.WEIGHT_DTYPE not in ("bf16", "fp8_e4m3"):
        raise NotImplementedError("ENGINE.WEIGHT_DTYPE %r (bf16 or fp8_e4m3)" % (cfg.ENGINE.WEIGHT_DTYPE,))
    return ResNetFPNBackbone(r.DEPTH, cfg.MODEL.FPN.OUT_CHANNELS, weight_dtype=cfg.ENGINE.WEIGHT_DTYPE)
