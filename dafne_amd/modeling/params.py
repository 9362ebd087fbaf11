"""Parameter containers whose state_dict keys equal the reference checkpoint's
(SURVEY 3.3): backbone.bottom_up.{stem,res2..res5}.*, backbone.fpn_lateral{3,4,5},
backbone.fpn_output{3,4,5}, backbone.top_block.{p6,p7},
proposal_generator.dafne_head.{cls,center,corners}_tower.{0,1,3,4,6,7,9,10},
...{cls_logits,ctrness,corners_pred,center_pred}, ...scales.{0..4}.scale.

The modules only HOLD fp32 parameters (for load_state_dict); the arithmetic runs
in the HIP engine on packed bf16 copies (engine.pack_model_weights).
"""
import math

import torch
from torch import nn


class ConvParams(nn.Module):
    """weight (+bias) of a conv; optional FrozenBN buffers under `.norm`."""

    def __init__(self, cout, cin, k, bias=True, frozen_bn=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        if frozen_bn:
            self.norm = FrozenBNParams(cout)
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")


class FrozenBNParams(nn.Module):
    """detectron2 FrozenBatchNorm2d buffers [recalled, SURVEY App. B]: eps 1e-5 and running_var initialised to
    1 - eps, so that a trunk that ships affine-only BN (MSRA R-50.pkl / R-101.pkl: `*_bn_s`, `*_bn_b`, no running
    statistics) gets scale = weight * rsqrt(running_var + eps) = weight exactly."""
    EPS = 1e-5

    def __init__(self, c):
        super().__init__()
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c) - self.EPS)


class AffineParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class ScaleParams(nn.Module):
    def __init__(self, v=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([float(v)]))


class BottleneckParams(nn.Module):
    def __init__(self, cin, cout, mid, first):
        super().__init__()
        if first:
            self.shortcut = ConvParams(cout, cin, 1, bias=False, frozen_bn=True)
        self.conv1 = ConvParams(mid, cin, 1, bias=False, frozen_bn=True)
        self.conv2 = ConvParams(mid, mid, 3, bias=False, frozen_bn=True)
        self.conv3 = ConvParams(cout, mid, 1, bias=False, frozen_bn=True)


class StemParams(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = ConvParams(64, 3, 7, bias=False, frozen_bn=True)


class ResNetParams(nn.Module):
    BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth):
        super().__init__()
        self.stem = StemParams()
        cin = 64
        for si, (nb, mid) in enumerate(zip(self.BLOCKS[depth], (64, 128, 256, 512))):
            blocks = []
            for b in range(nb):
                blocks.append(BottleneckParams(cin, mid * 4, mid, b == 0))
                cin = mid * 4
            setattr(self, "res%d" % (si + 2), nn.Sequential(*blocks))


class TopBlockParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.p6 = ConvParams(c, c, 3)
        self.p7 = ConvParams(c, c, 3)
        self.num_levels = 2
        self.in_feature = "p5"


def make_tower(c, n_convs=4):
    """Sequential with the reference's indices: conv at 0/3/6/9, GN at 1/4/7/10,
    ReLU (parameter-free placeholder) at 2/5/8/11 (dafne.py:318-348)."""
    mods = []
    for _ in range(n_convs):
        conv = ConvParams(c, c, 3)
        nn.init.normal_(conv.weight, std=0.01)          # dafne.py:273
        mods += [conv, AffineParams(c), nn.Identity()]
    return nn.Sequential(*mods)


def cls_prior_bias(prior_prob):
    return -math.log((1 - prior_prob) / prior_prob)     # dafne.py:283-285
