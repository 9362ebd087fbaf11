"""GPU parity of the fp8 (OCP e4m3) convolution path -- BASELINE config 5, dafne_conv2d_nhwc_fp8w_hip.

The reference has no fp8 path, so the definition lives here and in oracle/model.py (fp8=True): weights are e4m3 with
one power-of-two scale per output channel, the activation is multiplied by in_qscale, clamped to +-448 and rounded to
e4m3 (round to nearest even = torch.float8_e4m3fn), the products are accumulated in fp32.  An e4m3 x e4m3 product has
8 significant bits, so products are exact in fp32 and a torch fp32 convolution of the dequantised operands differs
from the kernel only by fp32 summation order and the final bf16 rounding: tolerance 2 bf16 ulps, as for the bf16 kernels.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv import bfr, close_bf16, dev

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


def q8(x, qscale=1.0):
    return (x * qscale).clamp(-448.0, 448.0).to(F8).float()


def run_fp8(xs, w, b, in_qscale, flags=0, gn_in=None, gn_stats=False):
    """xs: list of [N,C,H,W] float maps (levels sharing the weights).  Returns (outputs NCHW float on the CPU, partial, call)."""
    from dafne_amd import engine, _lib
    d = dev()
    n, cin = xs[0].shape[:2]
    cout = w.shape[0]
    ins = [engine.Act.from_nchw(x.to(d)) for x in xs]
    outs = [engine.Act(n, x.shape[2], x.shape[3], cout, d) for x in xs]
    wq, wscale = engine.pack_conv_fp8(w, d)
    bias = b.float().to(d).contiguous()
    segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, outs)]
    oscale = (wscale / in_qscale).contiguous()
    if gn_in is not None:
        flags |= engine.F_GNIN
    probe = engine.ConvCall(wq, bias, cin, cout, 3, 1, 1, flags, segs, n, gn_in=gn_in, fp8=(oscale, in_qscale))
    partial = None
    if gn_stats:
        partial = torch.zeros(4096, cout // 8, 2, dtype=torch.float32, device=d)
        probe = engine.ConvCall(wq, bias, cin, cout, 3, 1, 1, flags | engine.F_GN, segs, n, gn_partial=partial,
                                gn_in=gn_in, fp8=(oscale, in_qscale))
    assert probe.kernel_name() == "conv3x3_patch_fp8"
    probe(_lib.current_stream())
    torch.cuda.synchronize()
    for o in outs:      # the halo must still be zero
        assert float(o.t[:, 0].abs().max()) == 0 and float(o.t[:, -1].abs().max()) == 0
        assert float(o.t[:, :, 0].abs().max()) == 0 and float(o.t[:, :, -1].abs().max()) == 0
    return [o.nchw_float().cpu() for o in outs], partial, probe


def test_e4m3_weight_quantiser_is_exact_in_bf16():
    from dafne_amd import engine
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 64, 3, 3, generator=g) * torch.logspace(-4, 1, 64)[:, None, None, None]
    w[7] = 0
    q, s, deq = engine.quantize_weight_e4m3(w)
    assert q.dtype == F8 and torch.equal(deq.to(torch.bfloat16).float(), deq)
    assert torch.equal(torch.exp2(torch.round(torch.log2(s))), s)               # powers of two
    assert float((q.float().abs().amax(dim=(1, 2, 3)))[torch.arange(64) != 7].min()) >= 224.0     # the range is used
    assert float((deq - w).abs().max() / w.abs().max()) < 2.0 ** -4


@pytest.mark.parametrize("cin,cout,H,W,N,relu,qs", [
    (256, 256, 64, 64, 5, False, 1.0),       # FPN-output-like
    (64, 256, 40, 100, 3, True, 4.0),        # one slab, ragged in both directions
    (128, 512, 33, 47, 3, True, 0.5),        # two channel tiles, two slabs, ragged rows and columns
    (320, 256, 24, 64, 6, False, 16.0),      # five slabs
    (512, 256, 9, 31, 2, True, 2.0),         # eight slabs, a single ragged tile per image
])
def test_fp8_patch_kernel_vs_torch(cin, cout, H, W, N, relu, qs):
    from dafne_amd import engine
    g = torch.Generator().manual_seed(cin + cout + H + W)
    x = bfr(torch.randn(N, cin, H, W, generator=g) * 3.0)
    x[0, :, 0, 0] = 1000.0                    # saturates at 448 / qs
    x[0, :, -1, -1] = -1000.0
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    _, wscale, wdeq = engine.quantize_weight_e4m3(w)
    ref = F.conv2d(q8(x, qs), wdeq / qs, b, padding=1)
    if relu:
        ref = F.relu(ref)
    (got,), _, _ = run_fp8([x], w, b, qs, flags=engine.F_RELU if relu else 0)
    close_bf16(got, bfr(ref))


def test_fp8_gn_input_chain_over_levels():
    """Tower layer pair over three levels: layer 1 (bf16 patch kernel) emits the raw map + GroupNorm partial sums, the
    statistics are finalised, layer 2 is the fp8 kernel with GN_INPUT: GroupNorm + ReLU + e4m3 rounding while the patch
    is loaded, GroupNorm statistics of its own output.  The reference applies the same expression to the device's raw
    map and statistics."""
    from dafne_amd import engine, _lib
    d = dev()
    L = _lib.load()
    g = torch.Generator().manual_seed(78)
    C, N = 256, 12
    sizes = [(40, 72), (16, 32), (8, 8)]        # 15 + 2 + 1 tiles per image (the library wants >= 200 for layer 1)
    xs = [bfr(torch.randn(N, C, h, w, generator=g)) for h, w in sizes]
    w1 = bfr(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    w2 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.3 * torch.randn(C, generator=g)).to(d)
    ins = [engine.Act.from_nchw(x.to(d)) for x in xs]
    wp1, bp1 = engine.pack_conv(w1, b1, d)
    st = _lib.current_stream()
    raw = [engine.Act(N, h, w, C, d) for h, w in sizes]
    segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, raw)]
    partial = torch.zeros(4096, C // 8, 2, dtype=torch.float32, device=d)
    c1 = engine.ConvCall(wp1, bp1, C, C, 3, 1, 1, engine.F_GN, segs, N, gn_partial=partial)
    assert c1.kernel_name() == "conv3x3_patch"
    c1(st)
    stats = torch.zeros(len(raw), N, C // 8, 2, dtype=torch.float32, device=d)
    gsegs = (_lib.GnSeg * len(raw))()
    t0 = 0
    for k, (o, tpi) in enumerate(zip(raw, c1.tiles_per_image())):
        gsegs[k] = _lib.GnSeg(o.t.data_ptr(), o.h, o.w, t0, tpi)
        t0 += tpi * N
    _lib.check(L.dafne_groupnorm_finalize_hip(gsegs, len(raw), N, C, _lib.ptr(partial), _lib.ptr(stats),
                                              ctypes.c_float(1e-5), st), "finalize")
    torch.cuda.synchronize()
    raw_maps = [r.nchw_float() for r in raw]           # device tensors, fp32 view of the bf16 maps
    outs, partial2, c2 = run_fp8(raw_maps, w2, b2, 1.0, gn_in=(stats, gamma, beta), gn_stats=True)
    _, _, w2deq = engine.quantize_weight_e4m3(w2)
    t0 = 0
    for k, (rm, got) in enumerate(zip(raw_maps, outs)):
        mean = stats[k, :, :, 0].cpu()[:, :, None, None, None]
        rstd = stats[k, :, :, 1].cpu()[:, :, None, None, None]
        n, c, h, w = rm.shape
        xg = rm.cpu().reshape(n, c // 8, 8, h, w)
        y = (xg - mean) * rstd * gamma.cpu().reshape(1, c // 8, 8, 1, 1) + beta.cpu().reshape(1, c // 8, 8, 1, 1)
        y = F.relu(y).reshape(n, c, h, w)
        ref = F.conv2d(q8(y), w2deq, b2, padding=1)
        close_bf16(got, bfr(ref))
        # GroupNorm partial sums of the fp8 layer's own output: per (tile, group) sums over the valid pixels
        tpi = c2.tiles_per_image()[k]
        ps = partial2[t0:t0 + tpi * n].reshape(n, tpi, c // 8, 2).sum(1).cpu()
        t0 += tpi * n
        rg = ref.reshape(n, c // 8, -1)
        assert torch.allclose(ps[..., 0], rg.sum(-1), rtol=2e-3, atol=0.5)
        assert torch.allclose(ps[..., 1], (rg * rg).sum(-1), rtol=2e-3, atol=0.5)


def test_fp8w_rejects_unsupported_shapes():
    from dafne_amd import engine, _lib
    d = dev()
    x = engine.Act(1, 16, 16, 64, d)
    o = engine.Act(1, 16, 16, 128, d)
    w = torch.zeros(128, 64 * 9, dtype=torch.uint8, device=d)
    b = torch.zeros(128, device=d)
    sc = torch.ones(128, device=d)
    c = engine.ConvCall(w, b, 64, 128, 3, 1, 1, 0, [(x.t, o.t, None, 16, 16, 16, 16)], 1, fp8=(sc, 1.0))
    with pytest.raises(_lib.DafneHipError, match="fp8w"):
        c(_lib.current_stream())
