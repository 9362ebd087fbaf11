"""oracle/model.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain PyTorch (CPU, fp32) restatement of the dense half of the path:
ResNet-50/101 (detectron2 v0.5 semantics, SURVEY appendix B [recalled]) + FPN +
LastLevelP6P7 (dafne/modeling/backbone/fpn.py:16-37,58-91) + DAFNeHead's
center-to-corner branch (dafne/modeling/dafne/dafne.py:167-494).  Parameters
live in a flat dict keyed by the reference checkpoint's names.

Parity status: the HEAD is pinned by tests/golden/head_forward.npz (generated
by running the reference's DAFNeHead under stubs).  The BACKBONE is "parity
unpinned": detectron2 is not in /root/reference, so this restatement of its
ResNet/FPN is the definition the HIP engine is checked against.

``emulate_bf16=True`` rounds weights and activations to bf16 at exactly the
points where the HIP engine stores bf16 (after every fused conv epilogue), so
the engine can be compared layer-for-layer with a tight tolerance.
"""
import zlib

import numpy as np
import torch
import torch.nn.functional as F

STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


# ------------------------------------------------------------------ parameters
def param_shapes(depth=50, num_classes=15, fpn_ch=256):
    """name -> shape for every tensor of the inference graph."""
    s = {}

    def conv_bn(prefix, cout, cin, k):
        s[prefix + ".weight"] = (cout, cin, k, k)
        for b in ("weight", "bias", "running_mean", "running_var"):
            s[prefix + ".norm." + b] = (cout,)

    bu = "backbone.bottom_up."
    conv_bn(bu + "stem.conv1", 64, 3, 7)
    cin = 64
    for si, (nb, mid) in enumerate(zip(STAGE_BLOCKS[depth], (64, 128, 256, 512))):
        cout = mid * 4
        for b in range(nb):
            p = "%sres%d.%d." % (bu, si + 2, b)
            if b == 0:
                conv_bn(p + "shortcut", cout, cin, 1)
            conv_bn(p + "conv1", mid, cin, 1)
            conv_bn(p + "conv2", mid, mid, 3)
            conv_bn(p + "conv3", cout, mid, 1)
            cin = cout
    for lvl, c in ((3, 512), (4, 1024), (5, 2048)):
        s["backbone.fpn_lateral%d.weight" % lvl] = (fpn_ch, c, 1, 1)
        s["backbone.fpn_lateral%d.bias" % lvl] = (fpn_ch,)
        s["backbone.fpn_output%d.weight" % lvl] = (fpn_ch, fpn_ch, 3, 3)
        s["backbone.fpn_output%d.bias" % lvl] = (fpn_ch,)
    for n in ("p6", "p7"):
        s["backbone.top_block.%s.weight" % n] = (fpn_ch, fpn_ch, 3, 3)
        s["backbone.top_block.%s.bias" % n] = (fpn_ch,)
    s.update(head_param_shapes(num_classes, fpn_ch, "proposal_generator.dafne_head."))
    return s


def head_param_shapes(num_classes=15, ch=256, prefix=""):
    s = {}
    for tower in ("cls_tower", "center_tower", "corners_tower"):
        for i in range(4):
            s["%s%s.%d.weight" % (prefix, tower, 3 * i)] = (ch, ch, 3, 3)
            s["%s%s.%d.bias" % (prefix, tower, 3 * i)] = (ch,)
            s["%s%s.%d.weight" % (prefix, tower, 3 * i + 1)] = (ch,)
            s["%s%s.%d.bias" % (prefix, tower, 3 * i + 1)] = (ch,)
    for n, c in (("cls_logits", num_classes), ("ctrness", 1), ("corners_pred", 8), ("center_pred", 2)):
        s["%s%s.weight" % (prefix, n)] = (c, ch, 3, 3)
        s["%s%s.bias" % (prefix, n)] = (c,)
    for l in range(5):
        s["%sscales.%d.scale" % (prefix, l)] = (1,)
    return s


def _canon(name):
    return name.split("dafne_head.")[-1]


def _gen(name, shape, seed):
    """Deterministic, name-keyed synthetic value for one tensor."""
    key = _canon(name)
    rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
    leaf = key.split(".")[-1]
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        gain = 1.0 if any(t in key for t in ("cls_logits", "ctrness", "corners_pred", "center_pred")) else 2.0
        return rng.normal(0, np.sqrt(gain / fan_in), shape).astype(np.float32)
    if leaf == "scale":
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if leaf == "running_var":
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "running_mean":
        return rng.normal(0, 0.1, shape).astype(np.float32)
    if leaf == "weight":            # GN / FrozenBN gamma
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "bias":
        b = rng.normal(0, 0.1, shape).astype(np.float32)
        if key.startswith("cls_logits"):
            b -= 2.0
        return b
    raise KeyError(name)


def make_params(depth=50, num_classes=15, seed=0):
    return {k: torch.from_numpy(_gen(k, v, seed)) for k, v in param_shapes(depth, num_classes).items()}


def make_head_params(num_classes=15, seed=0, prefix=""):
    return {k: torch.from_numpy(_gen(k, v, seed)) for k, v in head_param_shapes(num_classes, 256, prefix).items()}


def fill_params(module, seed=0):
    """Fill a torch module's parameters/buffers in place with the same
    name-keyed values (used on the reference's DAFNeHead by make_golden.py)."""
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            t.copy_(torch.from_numpy(_gen(name, tuple(t.shape), seed)))


# --------------------------------------------------------------------- forward
def _bf(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


E4M3_MAX = 448.0


def quantize_weight_e4m3(w):
    """BASELINE config 5 ("fp8 weights"): the reference has no fp8 path, so this restatement DEFINES it.
    One power-of-two scale per output channel (smallest 2^k with amax / 2^k <= 448), values rounded to OCP e4m3
    (round to nearest even).  Returns the dequantised weight (exact in fp32 and in bf16)."""
    amax = w.reshape(w.shape[0], -1).abs().amax(dim=1)
    scale = torch.ones_like(amax)
    nz = amax > 0
    scale[nz] = torch.exp2(torch.ceil(torch.log2(amax[nz] / E4M3_MAX)))
    scale = torch.where(amax / scale > E4M3_MAX, scale * 2, scale)
    sh = (-1,) + (1,) * (w.dim() - 1)
    return (w / scale.reshape(sh)).to(torch.float8_e4m3fn).to(torch.float32) * scale.reshape(sh)


def quantize_act_e4m3(x):
    """Activation rounding of the fp8 MFMA layers: fp32 value clamped to +-448, rounded to e4m3 (in_qscale 1)."""
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def _wt(w, E, fp8):
    return _bf(quantize_weight_e4m3(w) if fp8 else w, E)


def _q8in(x, q):
    """Input of a layer that takes its plain (not GroupNorm-fed) activation in e4m3 with the calibrated power-of-two scale q
    (config 5, engine `act_q8`): the bf16 activation is multiplied by q, clamped to +-448, rounded to e4m3 and divided by q
    again (exact)."""
    return quantize_act_e4m3(x * q) / q


def fold_bn(P, prefix, eps=1e-5):
    """FrozenBatchNorm2d folded into the conv: y = conv(x, w*s) + (b - mean*s)."""
    s = P[prefix + ".norm.weight"] * torch.rsqrt(P[prefix + ".norm.running_var"] + eps)
    b = P[prefix + ".norm.bias"] - P[prefix + ".norm.running_mean"] * s
    return P[prefix + ".weight"] * s[:, None, None, None], b


def backbone_forward(P, x, depth=50, emulate_bf16=False, taps=None, fp8=False, act_q8=None):
    """x: [N,3,H,W] fp32, already normalised (and padded to /32).
    Returns {"p3".."p7"}.  ``taps`` (dict) collects intermediate tensors.
    fp8: every (BN-folded) weight is the dequantised e4m3 weight (quantize_weight_e4m3).
    act_q8 (with fp8): {engine weight key: in_qscale} -- the res4 / res5 3x3 layers ("res4.3.conv2") and the FPN output
    convolutions ("fpn_output3") listed there see their input rounded to e4m3 at that scale (the engine's calibrated
    fp8 MFMA layers, OneStageDetector.fp8_act_scales())."""
    E = emulate_bf16
    bu = "backbone.bottom_up."
    aq = act_q8 or {}

    def cbr(x, prefix, stride, pad, relu=True, res=None):
        w, b = fold_bn(P, prefix)
        key = prefix[len(bu):]
        if fp8 and key in aq:
            x = _q8in(x, aq[key])
        y = F.conv2d(x, _wt(w, E, fp8), b, stride=stride, padding=pad)
        if res is not None:
            y = y + res
        if relu:
            y = F.relu(y)
        return _bf(y, E)

    x = _bf(x, E)
    x = cbr(x, bu + "stem.conv1", 2, 3)
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["stem"] = x
    feats = {}
    for si, nb in enumerate(STAGE_BLOCKS[depth]):
        for b in range(nb):
            p = "%sres%d.%d." % (bu, si + 2, b)
            stride = 2 if (b == 0 and si > 0) else 1
            sc = cbr(x, p + "shortcut", stride, 0, relu=False) if b == 0 else x
            y = cbr(x, p + "conv1", stride, 0)          # STRIDE_IN_1X1
            y = cbr(y, p + "conv2", 1, 1)
            x = cbr(y, p + "conv3", 1, 0, relu=True, res=sc)
        feats["res%d" % (si + 2)] = x
        if taps is not None:
            taps["res%d" % (si + 2)] = x
    out = {}
    prev = None
    for lvl in (5, 4, 3):
        lat = F.conv2d(feats["res%d" % lvl], _wt(P["backbone.fpn_lateral%d.weight" % lvl], E, fp8),
                       P["backbone.fpn_lateral%d.bias" % lvl])
        if prev is not None:
            lat = lat + F.interpolate(prev, scale_factor=2, mode="nearest")
        prev = _bf(lat, E)
        pin = _q8in(prev, aq["fpn_output%d" % lvl]) if (fp8 and "fpn_output%d" % lvl in aq) else prev
        out["p%d" % lvl] = _bf(F.conv2d(pin, _wt(P["backbone.fpn_output%d.weight" % lvl], E, fp8),
                                        P["backbone.fpn_output%d.bias" % lvl], padding=1), E)
    p6 = _bf(F.conv2d(out["p5"], _wt(P["backbone.top_block.p6.weight"], E, fp8),
                      P["backbone.top_block.p6.bias"], stride=2, padding=1), E)
    p7 = _bf(F.conv2d(F.relu(p6), _wt(P["backbone.top_block.p7.weight"], E, fp8),
                      P["backbone.top_block.p7.bias"], stride=2, padding=1), E)
    out["p6"], out["p7"] = p6, p7
    return {k: out[k] for k in ("p3", "p4", "p5", "p6", "p7")}


def head_forward(P, feats, prefix="proposal_generator.dafne_head.", emulate_bf16=False, fp8=False, act_q8=None):
    """DAFNeHead.forward, center-to-corner branch with CORNER_TOWER_ON_CENTER_TOWER,
    CTR_ON_REG, USE_SCALE (dafne.py:350-370,388-414,459-494).  feats: list of 5
    [N,256,H,W].  Returns per-level lists (logits, reg, center, ctrness).
    fp8 (needs emulate_bf16): config 5 -- every weight is the dequantised e4m3 weight, and the tower layers whose
    input is a GroupNorm + ReLU output (layers 1..3 of each tower, layer 0 of the corners tower) see that input rounded
    to e4m3 from its fp32 value (the engine's fp8 MFMA layers quantise on load); all other activations are bf16."""
    E = emulate_bf16
    assert E or not fp8
    aq = act_q8 or {}

    def tower(x, name, first_q8=False):
        """x: fp32 activation (rounded at the consumer); returns the last layer's fp32 GroupNorm + ReLU output.
        act_q8["cls_tower.0"] / ["center_tower.0"]: the FPN-fed first layer takes its bf16 input in e4m3 at that scale."""
        for i in range(4):
            q8 = fp8 and (i > 0 or first_q8)
            if fp8 and i == 0 and not first_q8 and ("%s.0" % name) in aq:
                xin = _q8in(_bf(x, E), aq["%s.0" % name])
            else:
                xin = quantize_act_e4m3(x) if q8 else _bf(x, E)
            y = F.conv2d(xin, _wt(P["%s%s.%d.weight" % (prefix, name, 3 * i)], E, fp8),
                         P["%s%s.%d.bias" % (prefix, name, 3 * i)], padding=1)
            if E:
                # engine: statistics from the fp32 accumulator, value stored as bf16
                n, c, h, w = y.shape
                g = y.reshape(n, c // 8, -1)
                mean = g.mean(-1, keepdim=True)
                var = (g * g).mean(-1, keepdim=True) - mean * mean
                rstd = torch.rsqrt(var.clamp_min(0) + 1e-5)
                y16 = _bf(y, True).reshape(n, c // 8, -1)
                yn = ((y16 - mean) * rstd).reshape(n, c, h, w)
                yn = yn * P["%s%s.%d.weight" % (prefix, name, 3 * i + 1)][None, :, None, None] \
                    + P["%s%s.%d.bias" % (prefix, name, 3 * i + 1)][None, :, None, None]
                x = F.relu(yn)
            else:
                y = F.group_norm(y, y.shape[1] // 8, P["%s%s.%d.weight" % (prefix, name, 3 * i + 1)],
                                 P["%s%s.%d.bias" % (prefix, name, 3 * i + 1)], eps=1e-5)
                x = F.relu(y)
        return x

    def pred(x, name):
        return F.conv2d(_bf(x, E), _wt(P[prefix + name + ".weight"], E, fp8), P[prefix + name + ".bias"], padding=1)

    logits, regs, centers, ctrs = [], [], [], []
    for l, f in enumerate(feats):
        f = _bf(f, E)
        cls_t = tower(f, "cls_tower")
        ctr_t = tower(f, "center_tower")
        cor_t = tower(ctr_t, "corners_tower", first_q8=True)
        center = pred(ctr_t, "center_pred")
        delta = pred(cor_t, "corners_pred")
        reg = center.repeat(1, 4, 1, 1) + delta
        sc = P["%sscales.%d.scale" % (prefix, l)]
        regs.append(reg * sc)
        centers.append(center * sc)
        logits.append(pred(cls_t, "cls_logits"))
        ctrs.append(pred(cor_t, "ctrness"))
    return logits, regs, centers, ctrs


def preprocess(images_u8, pixel_mean, pixel_std, divis=32):
    """one_stage_detector.py:100-107 + d2 ImageList.from_tensors [recalled]:
    (x - mean)/std per image (CHW uint8, BGR), zero pad bottom/right to the batch
    max rounded up to a multiple of 32."""
    mean = torch.tensor(pixel_mean, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(pixel_std, dtype=torch.float32).view(3, 1, 1)
    ims = [(im.to(torch.float32) - mean) / std for im in images_u8]
    H = max(i.shape[1] for i in ims)
    W = max(i.shape[2] for i in ims)
    H = (H + divis - 1) // divis * divis
    W = (W + divis - 1) // divis * divis
    out = torch.zeros(len(ims), 3, H, W)
    for k, im in enumerate(ims):
        out[k, :, : im.shape[1], : im.shape[2]] = im
    return out, [tuple(i.shape[1:]) for i in ims]
