"""GPU parity of the dense HIP kernels vs a plain PyTorch fp32 reference of the same
op.  Inputs/weights are bf16-representable, so products are exact in fp32 and the
only differences are fp32 summation order and the final bf16 rounding: tolerance
2 bf16 ulps (2^-7 relative) on bf16 outputs, 2e-3 relative on fp32 outputs."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev():
    return torch.device("cuda", 0)


def bfr(x):
    return x.to(BF).float()


def run_conv(x, w, b, k, stride, pad, flags=0, res=None, out_f32=False, gn=False):
    """x: [N,C,H,W] float (bf16-representable) on CPU.  Returns engine output as NCHW float (CPU)."""
    from dafne_amd import engine, _lib
    d = dev()
    n, cin, h, wd = x.shape
    cout = w.shape[0]
    a = engine.Act.from_nchw(x.to(d))
    wp, bp = engine.pack_conv(w, b, d)
    ho, wo = engine.conv_out_hw(h, wd, k, stride, pad)
    if out_f32:
        o = torch.full((n, ho, wo, cout), float("nan"), dtype=torch.float32, device=d)
        ot = o
        flags |= engine.F_F32
    else:
        oa = engine.Act(n, ho, wo, cout, d)
        ot = oa.t
    r = engine.Act.from_nchw(res.to(d)) if res is not None else None
    partial = None
    call = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, flags,
                           [(a.t, ot, r.t if r is not None else None, h, wd, ho, wo)], n)
    if gn:
        partial = torch.zeros(call.num_tiles() if not gn else 4096, cout // 8, 2, dtype=torch.float32, device=d)
        call = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, flags | engine.F_GN,
                               [(a.t, ot, None, h, wd, ho, wo)], n, gn_partial=partial)
    call(_lib.current_stream())
    torch.cuda.synchronize()
    if out_f32:
        return o.permute(0, 3, 1, 2).cpu(), partial, call
    # the halo must still be zero
    assert float(oa.t[:, 0].abs().max()) == 0 and float(oa.t[:, -1].abs().max()) == 0
    assert float(oa.t[:, :, 0].abs().max()) == 0 and float(oa.t[:, :, -1].abs().max()) == 0
    return oa.nchw_float().cpu(), partial, call


def close_bf16(got, ref, ulps=2):
    tol = ulps * 2.0 ** -8 * ref.abs().clamp_min(2.0 ** -6) + 1e-3
    bad = (got - ref).abs() > tol
    assert not bad.any(), (int(bad.sum()), float((got - ref).abs().max()))


CASES = [
    # cin, cout, k, stride, pad, H, W, N
    (64, 64, 1, 1, 0, 16, 16, 2),
    (64, 256, 1, 1, 0, 12, 20, 1),      # ragged last tile (240 px)
    (256, 128, 1, 2, 0, 16, 16, 2),     # stride-2 1x1 (STRIDE_IN_1X1)
    (64, 64, 3, 1, 1, 16, 16, 1),
    (128, 128, 3, 1, 1, 9, 13, 2),      # odd sizes, ragged
    (256, 256, 3, 1, 1, 16, 16, 1),
    (256, 256, 3, 2, 1, 16, 16, 1),     # P6/P7
    (512, 2048, 1, 1, 0, 4, 4, 1),
    (1024, 256, 1, 1, 0, 8, 8, 1),
    (64, 256, 1, 1, 0, 256, 256, 2),    # >= 512 blocks of 256x256: the 8-wave tile
    (64, 512, 3, 1, 1, 120, 150, 2),    # 8-wave tile, ragged, two N tiles
]


@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W,N", CASES)
def test_conv_vs_torch(cin, cout, k, stride, pad, H, W, N):
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    ref = bfr(F.conv2d(x, w, b, stride=stride, padding=pad))
    got, _, _ = run_conv(x, w, b, k, stride, pad)
    close_bf16(got, ref)


def test_conv_relu_residual_and_upsample_add():
    from dafne_amd import engine
    g = torch.Generator().manual_seed(3)
    x = bfr(torch.randn(2, 128, 8, 8, generator=g))
    w = bfr(torch.randn(256, 128, 1, 1, generator=g) / 128 ** 0.5)
    b = torch.randn(256, generator=g) * 0.1
    res = bfr(torch.randn(2, 256, 8, 8, generator=g))
    ref = bfr(F.relu(F.conv2d(x, w, b) + res))
    got, _, _ = run_conv(x, w, b, 1, 1, 0, flags=engine.F_RELU | engine.F_RES, res=res)
    close_bf16(got, ref)
    coarse = bfr(torch.randn(2, 256, 4, 4, generator=g))
    ref = bfr(F.conv2d(x, w, b) + F.interpolate(coarse, scale_factor=2, mode="nearest"))
    got, _, _ = run_conv(x, w, b, 1, 1, 0, flags=engine.F_UP, res=coarse)
    close_bf16(got, ref)


@pytest.mark.parametrize("cin,cout,H,W,N,with_res,relu", [
    (128, 256, 150, 131, 2, True, True),     # 616 tiles > 512 resident workgroups: several tiles per workgroup, ragged
    (64, 128, 97, 113, 3, True, False),      # H = 2 half-K stages per tile (shortest K), one N tile
    (256, 1024, 40, 40, 3, True, True),      # res4-conv3 shape: 8 N tiles share each pixel tile
    (64, 256, 150, 131, 2, False, True),     # no residual: staging tile is write-only
    (1024, 256, 24, 24, 2, False, True),     # long K (32 half-K stages)
])
def test_streaming_1x1_kernel(cin, cout, H, W, N, with_res, relu):
    """Persistent streaming 1x1 kernel (conv.hip: conv_stream_kernel): residual DMA tile, in-place
    epilogue, counted vmcnt waits across tile boundaries."""
    from dafne_amd import engine
    g = torch.Generator().manual_seed(cin + cout + H)
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    res = bfr(torch.randn(N, cout, H, W, generator=g)) if with_res else None
    ref = F.conv2d(x, w, b)
    if with_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    flags = (engine.F_RELU if relu else 0) | (engine.F_RES if with_res else 0)
    got, _, _ = run_conv(x, w, b, 1, 1, 0, flags=flags, res=res)
    close_bf16(got, bfr(ref))


def _gn_ref(y, gamma, beta, groups, eps=1e-5):
    return F.relu(F.group_norm(y, groups, gamma, beta, eps))


@pytest.mark.parametrize("cin,cout,H,W,N,relu", [
    # (the library sends a layer here when 8 images of its shape make >= 200 tiles of 8x32 -- a nominal batch, whatever N is)
    (256, 256, 128, 96, 3, False),      # 16 x 3 tiles per image, FPN-output-like
    (64, 256, 40, 132, 3, True),        # one slab, ragged in both directions (40 = 5x8, 132 = 4x32 + 4)
    (128, 512, 33, 79, 2, True),        # two channel tiles, ragged rows (33) and columns (79)
    (320, 256, 40, 160, 2, False),      # five slabs
])
def test_patch_kernel_vs_torch(cin, cout, H, W, N, relu):
    """3x3 patch kernel (conv.hip: conv3x3_patch_kernel): 2-D tiles, patch DMA with halo, tap-offset reads."""
    from dafne_amd import engine
    g = torch.Generator().manual_seed(cin + cout + H + W)
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, w, b, padding=1)
    if relu:
        ref = F.relu(ref)
    got, _, call = run_conv(x, w, b, 3, 1, 1, flags=engine.F_RELU if relu else 0)
    assert call.kernel_name() == "conv3x3_patch"
    close_bf16(got, bfr(ref))


def test_patch_kernel_gn_stats_and_gn_input_chain():
    """Two tower layers over three levels: layer 1 emits raw output + tile partials, the statistics are
    finalised, layer 2 applies GroupNorm + ReLU while loading its patch (F_GNIN).  Compared with the
    unfused pipeline (conv -> groupnorm_relu pass -> conv) bit for bit, and with torch within bf16 noise."""
    from dafne_amd import engine, _lib
    d = dev()
    L = _lib.load()
    g = torch.Generator().manual_seed(77)
    C, N = 256, 4
    sizes = [(48, 104), (16, 32), (8, 8)]       # 24 + 2 + 1 tiles per image (the patch kernel takes >= 200 per nominal batch of 8)
    xs = [bfr(torch.randn(N, C, h, w, generator=g)) for h, w in sizes]
    w1 = bfr(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    w2 = bfr(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.3 * torch.randn(C, generator=g)).to(d)
    ins = [engine.Act.from_nchw(x.to(d)) for x in xs]
    wp1, bp1 = engine.pack_conv(w1, b1, d)
    wp2, bp2 = engine.pack_conv(w2, b2, d)
    st = _lib.current_stream()

    def layer1(outs):
        segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, outs)]
        probe = engine.ConvCall(wp1, bp1, C, C, 3, 1, 1, 0, segs, N)
        partial = torch.zeros(probe.num_tiles(), C // 8, 2, dtype=torch.float32, device=d)
        c = engine.ConvCall(wp1, bp1, C, C, 3, 1, 1, engine.F_GN, segs, N, gn_partial=partial)
        assert c.kernel_name() == "conv3x3_patch"
        c(st)
        stats = torch.zeros(len(outs), N, C // 8, 2, dtype=torch.float32, device=d)
        gsegs = (_lib.GnSeg * len(outs))()
        t0 = 0
        for k, (o, tpi) in enumerate(zip(outs, c.tiles_per_image())):
            gsegs[k] = _lib.GnSeg(o.t.data_ptr(), o.h, o.w, t0, tpi)
            t0 += tpi * N
        assert t0 == c.num_tiles()
        return partial, stats, gsegs

    # fused: finalize only, layer 2 normalises on load
    raw = [engine.Act(N, h, w, C, d) for h, w in sizes]
    partial, stats, gsegs = layer1(raw)
    _lib.check(L.dafne_groupnorm_finalize_hip(gsegs, len(raw), N, C, _lib.ptr(partial), _lib.ptr(stats),
                                              ctypes.c_float(1e-5), st), "finalize")
    out_f = [engine.Act(N, h, w, C, d) for h, w in sizes]
    segs2 = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(raw, out_f)]
    c2 = engine.ConvCall(wp2, bp2, C, C, 3, 1, 1, engine.F_GNIN, segs2, N, gn_in=(stats, gamma, beta))
    assert c2.kernel_id() == 6
    c2(st)
    # unfused: normalisation pass in place, plain layer 2
    raw_u = [engine.Act(N, h, w, C, d) for h, w in sizes]
    partial_u, stats_u, gsegs_u = layer1(raw_u)
    _lib.check(L.dafne_groupnorm_relu_nhwc_bf16_hip(gsegs_u, len(raw_u), N, C, _lib.ptr(partial_u), _lib.ptr(stats_u),
                                                    _lib.ptr(gamma), _lib.ptr(beta), ctypes.c_float(1e-5), st), "gn")
    out_u = [engine.Act(N, h, w, C, d) for h, w in sizes]
    segs2u = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(raw_u, out_u)]
    engine.ConvCall(wp2, bp2, C, C, 3, 1, 1, 0, segs2u, N)(st)
    torch.cuda.synchronize()
    assert torch.equal(stats, stats_u)
    for a, b_ in zip(out_f, out_u):
        assert torch.equal(a.t, b_.t)                       # same arithmetic, same rounding points
        assert float(a.t[:, 0].abs().max()) == 0 and float(a.t[:, :, -1].abs().max()) == 0
    # torch reference of the whole chain (bf16 rounding after conv1 and after GN+ReLU, like the engine)
    for x, a in zip(xs, out_f):
        y1 = bfr(F.conv2d(x, w1, b1, padding=1))
        y1n = bfr(_gn_ref(F.conv2d(x, w1, b1, padding=1), gamma.cpu(), beta.cpu(), C // 8))
        ref = bfr(F.conv2d(y1n, w2, b2, padding=1))
        got = a.nchw_float().cpu()
        assert (got - ref).abs().max() < 0.06 * ref.abs().max()      # stats from fp32 conv vs bf16 map: noise floor


@pytest.mark.parametrize("cout,cin", [(15, 256), (9, 256), (2, 256), (32, 128), (7, 64)])
def test_slab_kernel_levels_and_gn_input(cout, cin):
    """Prediction convolution over ragged levels on the slab kernel (id 7): fp32 output vs torch; then the
    F_GNIN form (GroupNorm + ReLU of a raw tower output applied in LDS) against the same kernel run on the
    output of the separate normalisation pass -- bit for bit."""
    from dafne_amd import engine, _lib
    d = dev()
    L = _lib.load()
    g = torch.Generator().manual_seed(100 + cout)
    C, N = cin, 3
    sizes = [(40, 72), (17, 33), (8, 8), (3, 5), (1, 1)]
    xs = [bfr(torch.randn(N, C, h, w, generator=g)) for h, w in sizes]
    w1 = bfr(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    b1 = torch.randn(C, generator=g) * 0.1
    wq = bfr(torch.randn(cout, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    bq = torch.randn(cout, generator=g)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.3 * torch.randn(C, generator=g)).to(d)
    st = _lib.current_stream()
    wp1, bp1 = engine.pack_conv(w1, b1, d)
    wpq, bpq = engine.pack_conv(wq, bq, d)
    ins = [engine.Act.from_nchw(x.to(d)) for x in xs]

    def pred(src, gn_in):
        outs = [torch.full((N, h, w, cout), float("nan"), dtype=torch.float32, device=d) for h, w in sizes]
        segs = [(i.t, o, None, i.h, i.w, i.h, i.w) for i, o in zip(src, outs)]
        c = engine.ConvCall(wpq, bpq, C, cout, 3, 1, 1, engine.F_F32 | (engine.F_GNIN if gn_in else 0), segs, N, gn_in=gn_in)
        assert c.kernel_name() == ("conv3x3_pred16" if cin == 256 and cout <= 16 else "conv3x3_slab")
        c(st)
        return outs

    # plain: vs torch
    for x, o in zip(xs, pred(ins, None)):
        ref = F.conv2d(x, wq, bq, padding=1).permute(0, 2, 3, 1)
        got = o.cpu()
        assert torch.isfinite(got).all()
        assert float((got - ref).abs().max()) < 2e-3 * max(float(ref.abs().max()), 1.0)

    # raw tower layer + statistics
    def layer1():
        raw = [engine.Act(N, h, w, C, d) for h, w in sizes]
        segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, raw)]
        probe = engine.ConvCall(wp1, bp1, C, C, 3, 1, 1, 0, segs, N)
        partial = torch.zeros(probe.num_tiles(), C // 8, 2, dtype=torch.float32, device=d)
        c = engine.ConvCall(wp1, bp1, C, C, 3, 1, 1, engine.F_GN, segs, N, gn_partial=partial)
        c(st)
        stats = torch.zeros(len(raw), N, C // 8, 2, dtype=torch.float32, device=d)
        gsegs = (_lib.GnSeg * len(raw))()
        t0 = 0
        for k, (o, tpi) in enumerate(zip(raw, c.tiles_per_image())):
            gsegs[k] = _lib.GnSeg(o.t.data_ptr(), o.h, o.w, t0, tpi)
            t0 += tpi * N
        return raw, partial, stats, gsegs

    raw, partial, stats, gsegs = layer1()
    _lib.check(L.dafne_groupnorm_finalize_hip(gsegs, len(raw), N, C, _lib.ptr(partial), _lib.ptr(stats),
                                              ctypes.c_float(1e-5), st), "finalize")
    fused = pred(raw, (stats, gamma, beta))
    raw_u, partial_u, stats_u, gsegs_u = layer1()
    _lib.check(L.dafne_groupnorm_relu_nhwc_bf16_hip(gsegs_u, len(raw_u), N, C, _lib.ptr(partial_u), _lib.ptr(stats_u),
                                                    _lib.ptr(gamma), _lib.ptr(beta), ctypes.c_float(1e-5), st), "gn")
    unfused = pred(raw_u, None)
    torch.cuda.synchronize()
    for a, b_ in zip(fused, unfused):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b_)


@pytest.mark.parametrize("cin,cout,k,stride,H,W,N,relu,res", [
    (512, 512, 3, 1, 32, 32, 3, True, False),       # res5 conv2 of a 3-image sub-batch: 96 tiles, 72 K steps
    (2048, 512, 1, 1, 32, 32, 8, True, False),      # res5 conv1 at batch 8: 256 tiles
    (256, 256, 3, 2, 32, 32, 2, False, False),      # P6: 3x3 stride 2, 8 tiles
    (256, 256, 3, 2, 16, 16, 1, False, False),      # P7 of one image: a single ragged tile per channel half
    (1024, 512, 1, 2, 64, 64, 3, True, False),      # stride-2 1x1
    (1024, 128, 1, 1, 24, 40, 2, True, True),       # residual epilogue, ragged last tile, K = 16 steps
    (128, 128, 3, 1, 20, 20, 1, True, False),       # 18 steps, 4 tiles
])
def test_igemm_ring_equals_two_stage_loop(cin, cout, k, stride, H, W, N, relu, res, monkeypatch):
    """Launches with at most one 128 x 128 tile per CU run conv_igemm_kernel<2,2,2,2> with a 4-stage operand ring (counted
    waits) instead of the double buffer with a drain per K step: same K order, bit-identical outputs; both against torch."""
    from dafne_amd import engine
    g = torch.Generator().manual_seed(cin + cout + H)
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    pad = 1 if k == 3 else 0
    ho, wo = engine.conv_out_hw(H, W, k, stride, pad)
    r = bfr(torch.randn(N, cout, ho, wo, generator=g)) if res else None
    flags = (engine.F_RELU if relu else 0) | (engine.F_RES if res else 0)
    outs = []
    for ring in ("1", "0"):
        monkeypatch.setenv("DAFNE_CONV_RING", ring)
        got, _, call = run_conv(x, w, b, k, stride, pad, flags=flags, res=r)
        assert call.kernel_name() == "conv_igemm<2,2,2,2>"
        outs.append(got)
    assert torch.equal(outs[0], outs[1])
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    close_bf16(outs[0], bfr(ref))


@pytest.mark.parametrize("cout,gn", [(15, True), (9, False), (16, True), (1, False)])
def test_pred16_kernel_full_size_equals_slab_kernel(cout, gn, monkeypatch):
    """The prediction layers at the headline shape (batch 8, five levels of a 1024^2 tile: 696 tiles, three per workgroup of
    the persistent grid) on conv3x3_pred16 against the one-tile-per-workgroup slab kernel on the same buffers (GroupNorm + ReLU
    on load included): the same products summed in another order -- fp32 rounding apart; level 0 also against torch."""
    from dafne_amd import engine, _lib
    d = dev()
    g = torch.Generator().manual_seed(300 + cout)
    C, N = 256, 8
    sizes = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
    ins = []
    for h, w in sizes:
        a = engine.Act(N, h, w, C, d)
        a.t[:, 1:-1, 1:-1, :] = (torch.randn(N, h, w, C, generator=g) * 2.0).to(torch.bfloat16).to(d)
        ins.append(a)
    wq = bfr(torch.randn(cout, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    bq = torch.randn(cout, generator=g)
    wpq, bpq = engine.pack_conv(wq, bq, d)
    gn_in = None
    if gn:
        stats = torch.empty(len(sizes), N, C // 8, 2, device=d)
        stats[..., 0] = torch.randn(len(sizes), N, C // 8, generator=g).to(d) * 0.3
        stats[..., 1] = (0.5 + torch.rand(len(sizes), N, C // 8, generator=g)).to(d)
        gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d)
        beta = (0.3 * torch.randn(C, generator=g)).to(d)
        gn_in = (stats, gamma, beta)
    st = _lib.current_stream()

    def run(flag, name):
        monkeypatch.setenv("DAFNE_CONV_PRED16", flag)
        outs = [torch.full((N, h, w, cout), float("nan"), dtype=torch.float32, device=d) for h, w in sizes]
        segs = [(i.t, o, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, outs)]
        c = engine.ConvCall(wpq, bpq, C, cout, 3, 1, 1, engine.F_F32 | (engine.F_GNIN if gn else 0), segs, N, gn_in=gn_in)
        assert c.kernel_name() == name and c.num_tiles() == 696
        c(st)
        c(st)                # twice: nothing is left behind between launches
        torch.cuda.synchronize()
        return outs

    new = run("1", "conv3x3_pred16")
    old = run("0", "conv3x3_slab")
    for a, b_ in zip(new, old):
        assert torch.isfinite(a).all()
        assert float((a - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), 1.0)
    if not gn:
        x0 = ins[0].nchw_float()[:2].cpu()
        ref = F.conv2d(x0, wq, bq, padding=1).permute(0, 2, 3, 1)
        assert float((new[0][:2].cpu() - ref).abs().max()) < 2e-3 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("cout", [15, 9, 2, 1, 16])
def test_prediction_conv_f32_output(cout):
    g = torch.Generator().manual_seed(cout)
    x = bfr(torch.randn(2, 256, 6, 10, generator=g))
    w = bfr(torch.randn(cout, 256, 3, 3, generator=g) / 48.0)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    got, _, _ = run_conv(x, w, b, 3, 1, 1, out_f32=True)
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) < 2e-3 * float(ref.abs().max())


def test_groupnorm_relu_pipeline():
    """conv(+GN partial sums) -> finalize -> apply(+ReLU) vs torch group_norm on the
    fp32 conv output (stats) applied to the bf16-rounded value (what the engine stores)."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 2, 256, 10, 14
    x = bfr(torch.randn(N, C, H, W, generator=g))
    w = bfr(torch.randn(C, C, 3, 3, generator=g) / 48.0)
    b = torch.randn(C, generator=g) * 0.1
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    y = F.conv2d(x, w, b, padding=1)
    grp = y.reshape(N, C // 8, -1)
    mean = grp.mean(-1, keepdim=True)
    var = (grp * grp).mean(-1, keepdim=True) - mean * mean
    yn = ((bfr(y).reshape(N, C // 8, -1) - mean) * torch.rsqrt(var + 1e-5)).reshape(N, C, H, W)
    ref = bfr(F.relu(yn * gamma[None, :, None, None] + beta[None, :, None, None]))
    d = dev()
    a = engine.Act.from_nchw(x.to(d))
    wp, bp = engine.pack_conv(w, b, d)
    oa = engine.Act(N, H, W, C, d)
    bm = engine.ConvCall(wp, bp, C, C, 3, 1, 1, 0, [(a.t, oa.t, None, H, W, H, W)], N).tile_pixels()
    tpi = (H * W + bm - 1) // bm
    partial = torch.zeros(tpi * N, C // 8, 2, dtype=torch.float32, device=d)
    call = engine.ConvCall(wp, bp, C, C, 3, 1, 1, engine.F_GN, [(a.t, oa.t, None, H, W, H, W)], N, gn_partial=partial)
    assert call.num_tiles() == tpi * N
    call(_lib.current_stream())
    stats = torch.zeros(1, N, C // 8, 2, dtype=torch.float32, device=d)
    segs = (_lib.GnSeg * 1)(_lib.GnSeg(oa.t.data_ptr(), H, W, 0, tpi))
    gd, bd = gamma.to(d), beta.to(d)
    _lib.check(L.dafne_groupnorm_relu_nhwc_bf16_hip(segs, 1, N, C, _lib.ptr(partial), _lib.ptr(stats), _lib.ptr(gd),
                                                    _lib.ptr(bd), ctypes.c_float(1e-5), _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.allclose(stats[0, :, :, 0].cpu(), mean.squeeze(-1), atol=1e-4)
    close_bf16(oa.nchw_float().cpu(), ref, ulps=3)


def test_big_tile_groupnorm_stats_and_residual():
    """256x256 tile path: GN partial sums (reduced on the host here) and residual+ReLU."""
    from dafne_amd import engine, _lib
    g = torch.Generator().manual_seed(6)
    N, C, H, W, CI = 2, 256, 128, 128, 320     # (1x1 layers with <= 256 input channels go to 128-wide tiles: conv_ws)
    x = bfr(torch.randn(N, CI, H, W, generator=g))
    w = bfr(torch.randn(C, CI, 1, 1, generator=g) / CI ** 0.5)
    b = torch.randn(C, generator=g) * 0.1
    y = F.conv2d(x, w, b)
    d = dev()
    a = engine.Act.from_nchw(x.to(d))
    wp, bp = engine.pack_conv(w, b, d)
    oa = engine.Act(N, H, W, C, d)
    probe = engine.ConvCall(wp, bp, CI, C, 1, 1, 0, 0, [(a.t, oa.t, None, H, W, H, W)], N)
    assert probe.tile_pixels() == 256
    nt = probe.num_tiles()
    partial = torch.zeros(nt, C // 8, 2, dtype=torch.float32, device=d)
    engine.ConvCall(wp, bp, CI, C, 1, 1, 0, engine.F_GN, [(a.t, oa.t, None, H, W, H, W)], N, gn_partial=partial)(
        _lib.current_stream())
    torch.cuda.synchronize()
    close_bf16(oa.nchw_float().cpu(), bfr(y))
    ps = partial.cpu().reshape(N, nt // N, C // 8, 2).sum(1)
    grp = y.reshape(N, C // 8, -1)
    assert torch.allclose(ps[..., 0], grp.sum(-1), rtol=1e-4, atol=1e-1)
    assert torch.allclose(ps[..., 1], (grp * grp).sum(-1), rtol=1e-4, atol=1e-1)
    res = bfr(torch.randn(N, C, H, W, generator=g))
    got, _, _ = run_conv(x, w, b, 1, 1, 0, flags=engine.F_RELU | engine.F_RES, res=res)
    close_bf16(got, bfr(F.relu(y + res)))


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 132, 260), (3, 16, 128)])
def test_stem_pool_conv1_equals_stem_pool_then_conv(N, H, W):
    """dafne_stem_pool_conv1_hip (stem_pool.hip CONV1: res2.0's 1x1 64 -> 64 convolution on the pooled tile in LDS) against
    dafne_stem_pool_hip followed by dafne_conv2d_nhwc_bf16_hip: both outputs bit-identical, halos untouched; ragged tiles
    (33 x 65 pooled pixels) and several tiles per workgroup."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    img = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8).to(d)
    w = bfr(torch.randn(64, 3, 7, 7, generator=g) / 200.0)
    b = torch.randn(64, generator=g) * 0.1
    w1 = bfr(torch.randn(64, 64, 1, 1, generator=g) / 8.0)
    b1 = torch.randn(64, generator=g) * 0.1
    wp, bp = engine.pack_stem(w, b, d)
    w1p, b1p = engine.pack_conv(w1, b1, d)
    assert tuple(w1p.shape) == (64, 64)
    st = _lib.current_stream()
    stem_in = torch.zeros(N, H + 6, W + 6, 4, dtype=BF, device=d)
    m3 = (ctypes.c_float * 3)(103.53, 116.28, 123.675)
    s3 = (ctypes.c_float * 3)(1.0, 1.0, 1.0)
    _lib.check(L.dafne_preprocess_image_hip(_lib.ptr(img), 0, N, H, W, None, m3, s3, H, W, _lib.ptr(stem_in), st))
    Hq, Wq = H // 4, W // 4
    p_ref, u_ref = engine.Act(N, Hq, Wq, 64, d), engine.Act(N, Hq, Wq, 64, d)
    _lib.check(L.dafne_stem_pool_hip(_lib.ptr(stem_in), _lib.ptr(wp), _lib.ptr(bp), N, H, W, _lib.ptr(p_ref.t), st))
    engine.ConvCall(w1p, b1p, 64, 64, 1, 1, 0, engine.F_RELU, [(p_ref.t, u_ref.t, None, Hq, Wq, Hq, Wq)], N)(st)
    p_f, u_f = engine.Act(N, Hq, Wq, 64, d), engine.Act(N, Hq, Wq, 64, d)
    for _ in range(2):
        _lib.check(L.dafne_stem_pool_conv1_hip(_lib.ptr(stem_in), _lib.ptr(wp), _lib.ptr(bp), _lib.ptr(w1p), _lib.ptr(b1p), N, H, W,
                                               _lib.ptr(p_f.t), _lib.ptr(u_f.t), st))
    torch.cuda.synchronize()
    assert torch.equal(p_f.t, p_ref.t)
    assert torch.equal(u_f.t, u_ref.t), float((u_f.t.float() - u_ref.t.float()).abs().max())
    assert float(u_f.t[:, 0].abs().max()) == 0 and float(u_f.t[:, :, -1].abs().max()) == 0
    assert float(u_f.t.float().abs().max()) > 0
    ref = F.relu(F.conv2d(p_ref.nchw_float().cpu(), w1, b1))
    close_bf16(u_f.nchw_float().cpu(), bfr(ref))
    with pytest.raises(_lib.DafneHipError):
        _lib.check(L.dafne_stem_pool_conv1_hip(_lib.ptr(stem_in), _lib.ptr(wp), _lib.ptr(bp), None, _lib.ptr(b1p), N, H, W,
                                               _lib.ptr(p_f.t), _lib.ptr(u_f.t), st), "stem_pool_conv1")


def test_stem_preprocess_maxpool():
    from dafne_amd import engine, _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(9)
    N, H, W = 2, 64, 96
    img = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    mean = [103.53, 116.28, 123.675]
    std = [1.0, 1.0, 1.0]
    w = bfr(torch.randn(64, 3, 7, 7, generator=g) / 200.0)
    b = torch.randn(64, generator=g) * 0.1
    xn = bfr((img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1))
    ref = F.max_pool2d(bfr(F.relu(F.conv2d(xn, w, b, stride=2, padding=3))), 3, 2, 1)
    d = dev()
    stem_in = torch.zeros(N, H + 6, W + 6, 4, dtype=BF, device=d)
    m3 = (ctypes.c_float * 3)(*mean)
    s3 = (ctypes.c_float * 3)(*std)
    imd = img.to(d)
    _lib.check(L.dafne_preprocess_image_hip(_lib.ptr(imd), 0, N, H, W, None, m3, s3, H, W, _lib.ptr(stem_in),
                                            _lib.current_stream()))
    wp, bp = engine.pack_stem(w, b, d)
    so = engine.Act(N, H // 2, W // 2, 64, d)
    engine.ConvCall(wp, bp, 4, 64, 7, 2, 3, engine.F_RELU, [(stem_in, so.t, None, H + 6, W + 6, H // 2, W // 2)], N)(
        _lib.current_stream())
    po = engine.Act(N, H // 4, W // 4, 64, d)
    _lib.check(L.dafne_maxpool3x3s2_nhwc_bf16_hip(_lib.ptr(so.t), _lib.ptr(po.t), N, H // 2, W // 2, 64,
                                                  _lib.current_stream()))
    # the fused stem (conv + ReLU + pool in one kernel) is bit-identical to the two launches
    pf = engine.Act(N, H // 4, W // 4, 64, d)
    _lib.check(L.dafne_stem_pool_hip(_lib.ptr(stem_in), _lib.ptr(wp), _lib.ptr(bp), N, H, W, _lib.ptr(pf.t),
                                     _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(stem_in[:, 3:-3, 3:-3, :3].float().cpu(), xn.permute(0, 2, 3, 1))
    close_bf16(po.nchw_float().cpu(), ref)
    assert torch.equal(pf.t, po.t)
    # HWC input layout gives the same stem input
    stem2 = torch.zeros_like(stem_in)
    imh = img.permute(0, 2, 3, 1).contiguous().to(d)
    _lib.check(L.dafne_preprocess_image_hip(_lib.ptr(imh), 1, N, H, W, None, m3, s3, H, W, _lib.ptr(stem2),
                                            _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(stem2, stem_in)


@pytest.mark.parametrize("N,H,W,Hn,Wn,hwc", [(2, 64, 96, 64, 96, 0), (3, 50, 70, 64, 96, 0), (2, 33, 45, 64, 64, 0), (2, 33, 45, 64, 64, 1),
                                              (1, 61, 67, 61, 67, 0), (4, 1024, 1024, 1024, 1024, 0), (2, 800, 1216, 800, 1216, 1)])
def test_preprocess_rows_lut_form(N, H, W, Hn, Wn, hwc):
    """dafne_preprocess_image_hip (round 5: a workgroup per output row, bf16((v - mean) / std) tabulated, dword reads of aligned planar
    rows) against (x - mean) / std in fp32 -> bf16 (one_stage_detector.py:100-107): per-image valid sizes inside a zero-padded
    batch, padded (Hn, Wn) beyond the batch size, widths that are not a multiple of 4 (byte path, odd row pitch: 8-byte
    stores), both input layouts, the headline size; border, padding and the 4th channel stay zero."""
    from dafne_amd import _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    img = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    valid = torch.tensor([[max(1, H - 7 * k), max(1, W - 5 * k)] for k in range(N)], dtype=torch.int32)
    mean, std = [103.53, 116.28, 123.675], [57.375, 57.12, 58.395]
    ref = torch.zeros(N, Hn + 6, Wn + 6, 4)
    xn = bfr((img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)).permute(0, 2, 3, 1)
    for k in range(N):
        vh, vw = int(valid[k, 0]), int(valid[k, 1])
        ref[k, 3:3 + vh, 3:3 + vw, :3] = xn[k, :vh, :vw]
    src = (img.permute(0, 2, 3, 1).contiguous() if hwc else img).to(d)
    m3 = (ctypes.c_float * 3)(*mean)
    s3 = (ctypes.c_float * 3)(*std)
    for vt in (valid.to(d), None):
        out = torch.full((N, Hn + 6, Wn + 6, 4), 7.0, dtype=BF, device=d)
        _lib.check(L.dafne_preprocess_image_hip(_lib.ptr(src), hwc, N, H, W, _lib.ptr(vt) if vt is not None else None, m3, s3, Hn, Wn,
                                                _lib.ptr(out), _lib.current_stream()))
        torch.cuda.synchronize()
        if vt is None:
            ref = torch.zeros(N, Hn + 6, Wn + 6, 4)
            ref[:, 3:3 + H, 3:3 + W, :3] = xn
        assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("N,H,W", [(1, 32, 32), (3, 160, 224), (2, 1024, 1024), (1, 480, 1216)])
def test_stem_pool_fused_equals_conv_then_pool(N, H, W):
    """dafne_stem_pool_hip against the stem through the generic kernel + max-pool launch: ragged tiles
    (pooled width not a multiple of 32, height not a multiple of 4 rows), several tiles per workgroup,
    top / left pool padding; bit for bit, halo untouched."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(H + W)
    stem_in = torch.zeros(N, H + 6, W + 6, 4, dtype=BF, device=d)
    stem_in[:, 3:-3, 3:-3, :3] = (torch.randint(0, 256, (N, H, W, 3), generator=g).float() - 110.0).to(BF).to(d)
    w = bfr(torch.randn(64, 3, 7, 7, generator=g) / 200.0)
    b = torch.randn(64, generator=g) * 0.5
    wp, bp = engine.pack_stem(w, b, d)
    st = _lib.current_stream()
    so = engine.Act(N, H // 2, W // 2, 64, d)
    engine.ConvCall(wp, bp, 4, 64, 7, 2, 3, engine.F_RELU, [(stem_in, so.t, None, H + 6, W + 6, H // 2, W // 2)], N)(st)
    po = engine.Act(N, H // 4, W // 4, 64, d)
    _lib.check(L.dafne_maxpool3x3s2_nhwc_bf16_hip(_lib.ptr(so.t), _lib.ptr(po.t), N, H // 2, W // 2, 64, st))
    pf = engine.Act(N, H // 4, W // 4, 64, d)
    _lib.check(L.dafne_stem_pool_hip(_lib.ptr(stem_in), _lib.ptr(wp), _lib.ptr(bp), N, H, W, _lib.ptr(pf.t), st))
    torch.cuda.synchronize()
    assert float(po.t.float().abs().max()) > 0
    assert torch.equal(pf.t, po.t)
    if N * H * W <= 3 * 160 * 224:
        ref = F.max_pool2d(bfr(F.relu(F.conv2d(stem_in[:, 3:-3, 3:-3, :3].float().cpu().permute(0, 3, 1, 2), w, b,
                                               stride=2, padding=3))), 3, 2, 1)
        close_bf16(pf.nchw_float().cpu(), ref)


@pytest.mark.parametrize("N,H,W", [(1, 8, 16), (2, 64, 64), (3, 13, 21), (1, 1, 1), (3, 128, 128), (2, 256, 256)])
def test_bottleneck_tail_head_narrow_equals_two_convs(N, H, W):
    """dafne_bottleneck_tail_head_narrow_hip (res2: conv3 64 -> 256 + shortcut + ReLU, then the next block's conv1 256 -> 64 +
    ReLU, one persistent streaming kernel) against the two launches of the generic path: bit for bit on both outputs --
    ragged last tile, several tiles per workgroup (384 and 1024 tiles on 256 CUs), halo untouched -- and against torch
    within bf16 rounding."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(2000 + H * W)
    t = bfr(torch.randn(N, 64, H, W, generator=g))
    x = bfr(torch.randn(N, 256, H, W, generator=g))
    w3 = bfr(torch.randn(256, 64, 1, 1, generator=g) / 8.0)
    b3 = torch.randn(256, generator=g) * 0.2
    w1 = bfr(torch.randn(64, 256, 1, 1, generator=g) / 16.0)
    b1 = torch.randn(64, generator=g) * 0.2
    st = _lib.current_stream()
    ta, xa = engine.Act.from_nchw(t.to(d)), engine.Act.from_nchw(x.to(d))
    w3p, b3p = engine.pack_conv(w3, b3, d)
    w1p, b1p = engine.pack_conv(w1, b1, d)
    y_u, z_u = engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 64, d)
    engine.ConvCall(w3p, b3p, 64, 256, 1, 1, 0, engine.F_RELU | engine.F_RES, [(ta.t, y_u.t, xa.t, H, W, H, W)], N)(st)
    engine.ConvCall(w1p, b1p, 256, 64, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
    wf = engine.pack_b2b_narrow(w3p, w1p)
    y_f, z_f = engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 64, d)
    for _ in range(2):          # twice: the second launch finds the outputs already written (no read-modify-write anywhere)
        _lib.check(L.dafne_bottleneck_tail_head_narrow_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p),
                                                           _lib.ptr(b1p), N, H, W, _lib.ptr(y_f.t), _lib.ptr(z_f.t), st), "b2b_narrow")
    torch.cuda.synchronize()
    assert torch.equal(y_f.t, y_u.t)
    assert torch.equal(z_f.t, z_u.t)
    assert float(z_f.t[:, 0].abs().max()) == 0 and float(y_f.t[:, :, -1].abs().max()) == 0
    if N * H * W <= 3 * 128 * 128:
        y_ref = bfr(F.relu(F.conv2d(t, w3, b3) + x))
        close_bf16(y_f.nchw_float().cpu(), y_ref)
        z_ref = bfr(F.relu(F.conv2d(y_ref, w1, b1)))
        got = z_f.nchw_float().cpu()
        assert float((got - z_ref).abs().max()) < 0.02 * float(z_ref.abs().max())


@pytest.mark.parametrize("N,H,W,relu", [(1, 8, 32, 1), (2, 64, 64, 1), (3, 13, 21, 0), (1, 1, 1, 1), (2, 256, 256, 1),
                                          (1, 40, 410, 1), (5, 72, 104, 1)])
def test_conv3x3_c64_equals_generic_conv(N, H, W, relu):
    """dafne_conv3x3_c64_hip (res2 conv2: persistent, weights in registers, patch staged once per 8 x 32 tile) against
    dafne_conv2d_nhwc_bf16_hip: bit for bit -- ragged tiles in both directions, 1 / 2 / 4 tiles per workgroup, halo
    untouched -- and against torch within bf16 rounding."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(5000 + H * W)
    x = bfr(torch.randn(N, 64, H, W, generator=g))
    w = bfr(torch.randn(64, 64, 3, 3, generator=g) / 24.0)
    b = torch.randn(64, generator=g) * 0.2
    st = _lib.current_stream()
    xa = engine.Act.from_nchw(x.to(d))
    wp, bp = engine.pack_conv(w, b, d)
    y_u, y_f = engine.Act(N, H, W, 64, d), engine.Act(N, H, W, 64, d)
    engine.ConvCall(wp, bp, 64, 64, 3, 1, 1, engine.F_RELU if relu else 0, [(xa.t, y_u.t, None, H, W, H, W)], N)(st)
    for _ in range(2):
        _lib.check(L.dafne_conv3x3_c64_hip(_lib.ptr(xa.t), _lib.ptr(wp), _lib.ptr(bp), N, H, W, relu, _lib.ptr(y_f.t), st), "c64")
    torch.cuda.synchronize()
    assert float(y_f.t.float().abs().max()) > 0
    assert torch.equal(y_f.t, y_u.t)
    assert float(y_f.t[:, 0].abs().max()) == 0 and float(y_f.t[:, :, -1].abs().max()) == 0 and float(y_f.t[:, -1].abs().max()) == 0
    if N * H * W <= 2 * 64 * 64:
        ref = F.conv2d(x, w, b, padding=1)
        close_bf16(y_f.nchw_float().cpu(), bfr(F.relu(ref) if relu else ref))


@pytest.mark.parametrize("N,H,W", [(1, 8, 16), (2, 64, 64), (3, 13, 21), (1, 1, 1), (3, 128, 128), (1, 40, 410)])
def test_bottleneck_tail_head_mid_equals_two_convs(N, H, W):
    """dafne_bottleneck_tail_head_mid_hip (res3: conv3 128 -> 512 + shortcut + ReLU, then the next block's conv1 512 -> 128 +
    ReLU; persistent, weights streamed through LDS by waves 4-7, HBM traffic by waves 0-3) against the two launches of the
    generic path: bit for bit on both outputs -- a single tile, ragged tiles, 1 / 3 / 3 tiles per workgroup."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(4000 + H * W)
    t = bfr(torch.randn(N, 128, H, W, generator=g))
    x = bfr(torch.randn(N, 512, H, W, generator=g))
    w3 = bfr(torch.randn(512, 128, 1, 1, generator=g) / 11.0)
    b3 = torch.randn(512, generator=g) * 0.2
    w1 = bfr(torch.randn(128, 512, 1, 1, generator=g) / 22.0)
    b1 = torch.randn(128, generator=g) * 0.2
    st = _lib.current_stream()
    ta, xa = engine.Act.from_nchw(t.to(d)), engine.Act.from_nchw(x.to(d))
    w3p, b3p = engine.pack_conv(w3, b3, d)
    w1p, b1p = engine.pack_conv(w1, b1, d)
    y_u, z_u = engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
    engine.ConvCall(w3p, b3p, 128, 512, 1, 1, 0, engine.F_RELU | engine.F_RES, [(ta.t, y_u.t, xa.t, H, W, H, W)], N)(st)
    engine.ConvCall(w1p, b1p, 512, 128, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
    wf = engine.pack_b2b_mid(w3p, w1p)
    y_f, z_f = engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
    for _ in range(2):
        _lib.check(L.dafne_bottleneck_tail_head_mid_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p),
                                                        _lib.ptr(b1p), N, H, W, _lib.ptr(y_f.t), _lib.ptr(z_f.t), st), "b2b_mid")
    torch.cuda.synchronize()
    assert float(y_f.t.float().abs().max()) > 0
    assert torch.equal(y_f.t, y_u.t)
    assert torch.equal(z_f.t, z_u.t)
    assert float(z_f.t[:, 0].abs().max()) == 0 and float(y_f.t[:, :, -1].abs().max()) == 0
    if N * H * W <= 2 * 64 * 64:
        y_ref = bfr(F.relu(F.conv2d(t, w3, b3) + x))
        close_bf16(y_f.nchw_float().cpu(), y_ref)


@pytest.mark.parametrize("N,H,W", [(1, 8, 16), (3, 13, 21), (1, 1, 1), (3, 128, 128), (2, 256, 256)])
def test_bottleneck_proj_tail_head_narrow_equals_three_convs(N, H, W):
    """dafne_bottleneck_proj_tail_head_narrow_hip (res2 block 0: projection shortcut 64 -> 256 computed in the kernel and
    rounded to bf16 like the separate launch, conv3 + shortcut + ReLU, next conv1 + ReLU) against the three generic
    launches: bit for bit on both outputs."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(3000 + H * W)
    t = bfr(torch.randn(N, 64, H, W, generator=g))
    x0 = bfr(torch.relu(torch.randn(N, 64, H, W, generator=g)))
    w3 = bfr(torch.randn(256, 64, 1, 1, generator=g) / 8.0)
    b3 = torch.randn(256, generator=g) * 0.2
    ws = bfr(torch.randn(256, 64, 1, 1, generator=g) / 8.0)
    bs_ = torch.randn(256, generator=g) * 0.2
    w1 = bfr(torch.randn(64, 256, 1, 1, generator=g) / 16.0)
    b1 = torch.randn(64, generator=g) * 0.2
    st = _lib.current_stream()
    ta, xa = engine.Act.from_nchw(t.to(d)), engine.Act.from_nchw(x0.to(d))
    w3p, b3p = engine.pack_conv(w3, b3, d)
    wsp, bsp = engine.pack_conv(ws, bs_, d)
    w1p, b1p = engine.pack_conv(w1, b1, d)
    sc_u, y_u, z_u = engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 64, d)
    engine.ConvCall(wsp, bsp, 64, 256, 1, 1, 0, 0, [(xa.t, sc_u.t, None, H, W, H, W)], N)(st)
    engine.ConvCall(w3p, b3p, 64, 256, 1, 1, 0, engine.F_RELU | engine.F_RES, [(ta.t, y_u.t, sc_u.t, H, W, H, W)], N)(st)
    engine.ConvCall(w1p, b1p, 256, 64, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
    wf = engine.pack_b2b_narrow(w3p, w1p, wsp)
    y_f, z_f = engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 64, d)
    for _ in range(2):
        _lib.check(L.dafne_bottleneck_proj_tail_head_narrow_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p),
                                                                _lib.ptr(bsp), _lib.ptr(b1p), N, H, W, _lib.ptr(y_f.t),
                                                                _lib.ptr(z_f.t), st), "b2b_narrow_proj")
    torch.cuda.synchronize()
    assert float(y_f.t.float().abs().max()) > 0
    assert torch.equal(y_f.t, y_u.t)
    assert torch.equal(z_f.t, z_u.t)
    assert float(z_f.t[:, 0].abs().max()) == 0 and float(y_f.t[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize("N,H,W", [(1, 8, 16), (2, 64, 64), (3, 13, 21), (1, 1, 1)])
def test_bottleneck_tail_head_fused_equals_two_convs(N, H, W):
    """dafne_bottleneck_tail_head_hip (conv3 + residual + ReLU, then the next block's conv1 + ReLU, one kernel)
    against the two launches of the generic path: bit for bit on both outputs, ragged last tile, halo untouched;
    and against torch within bf16 rounding."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(1000 + H * W)
    t = bfr(torch.randn(N, 256, H, W, generator=g))
    x = bfr(torch.randn(N, 1024, H, W, generator=g))
    w3 = bfr(torch.randn(1024, 256, 1, 1, generator=g) / 16.0)
    b3 = torch.randn(1024, generator=g) * 0.2
    w1 = bfr(torch.randn(256, 1024, 1, 1, generator=g) / 32.0)
    b1 = torch.randn(256, generator=g) * 0.2
    st = _lib.current_stream()
    ta, xa = engine.Act.from_nchw(t.to(d)), engine.Act.from_nchw(x.to(d))
    w3p, b3p = engine.pack_conv(w3, b3, d)
    w1p, b1p = engine.pack_conv(w1, b1, d)
    # separate launches
    y_u, z_u = engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)
    engine.ConvCall(w3p, b3p, 256, 1024, 1, 1, 0, engine.F_RELU | engine.F_RES, [(ta.t, y_u.t, xa.t, H, W, H, W)], N)(st)
    engine.ConvCall(w1p, b1p, 1024, 256, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
    # fused
    wf = engine.pack_b2b(w3p, w1p)
    y_f, z_f = engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)
    _lib.check(L.dafne_bottleneck_tail_head_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p), _lib.ptr(b1p),
                                                N, H, W, _lib.ptr(y_f.t), _lib.ptr(z_f.t), st), "b2b")
    torch.cuda.synchronize()
    assert torch.equal(y_f.t, y_u.t)
    assert torch.equal(z_f.t, z_u.t)
    assert float(z_f.t[:, 0].abs().max()) == 0 and float(y_f.t[:, :, -1].abs().max()) == 0
    y_ref = bfr(F.relu(F.conv2d(t, w3, b3) + x))
    close_bf16(y_f.nchw_float().cpu(), y_ref)
    z_ref = bfr(F.relu(F.conv2d(y_ref, w1, b1)))
    got = z_f.nchw_float().cpu()
    assert float((got - z_ref).abs().max()) < 0.02 * float(z_ref.abs().max())


@pytest.mark.parametrize("th", [4, 2, 0])
@pytest.mark.parametrize("N,H,W", [(1, 8, 32), (8, 64, 64), (3, 13, 21), (2, 30, 44), (1, 1, 1), (1, 5, 70), (1, 64, 64)])
def test_bottleneck_body_fused_equals_three_convs(N, H, W, th, monkeypatch):
    """dafne_bottleneck_body_hip (conv2 3x3 + ReLU, conv3 + residual + ReLU, the next block's conv1 + ReLU in one kernel;
    the 3x3's output never reaches HBM) against the three launches of the generic path: bit for bit on both outputs --
    the headline shape (8 x 64 x 64, 256 exact tiles) and ragged sizes (partial tiles in both directions, more
    than one tile column, a single pixel) --, the zero halo untouched, nothing written outside the two outputs; and
    against torch within bf16 rounding.  Both tile geometries (round 6: 4 x 32, and 2 x 32 for launches with few tiles) on every
    shape, and the library's own choice (th = 0)."""
    if th:
        monkeypatch.setenv("DAFNE_BNECK_TH", str(th))
    else:
        monkeypatch.delenv("DAFNE_BNECK_TH", raising=False)
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(2000 + H * W)
    u = bfr(torch.randn(N, 256, H, W, generator=g))
    x = bfr(torch.randn(N, 1024, H, W, generator=g))
    w2 = bfr(torch.randn(256, 256, 3, 3, generator=g) / 48.0)
    b2 = torch.randn(256, generator=g) * 0.2
    w3 = bfr(torch.randn(1024, 256, 1, 1, generator=g) / 16.0)
    b3 = torch.randn(1024, generator=g) * 0.2
    w1 = bfr(torch.randn(256, 1024, 1, 1, generator=g) / 32.0)
    b1 = torch.randn(256, generator=g) * 0.2
    st = _lib.current_stream()
    ua, xa = engine.Act.from_nchw(u.to(d)), engine.Act.from_nchw(x.to(d))
    w2p, b2p = engine.pack_conv(w2, b2, d)
    w3p, b3p = engine.pack_conv(w3, b3, d)
    w1p, b1p = engine.pack_conv(w1, b1, d)
    # separate launches
    t_u, y_u, z_u = engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)
    engine.ConvCall(w2p, b2p, 256, 256, 3, 1, 1, engine.F_RELU, [(ua.t, t_u.t, None, H, W, H, W)], N)(st)
    engine.ConvCall(w3p, b3p, 256, 1024, 1, 1, 0, engine.F_RELU | engine.F_RES, [(t_u.t, y_u.t, xa.t, H, W, H, W)], N)(st)
    engine.ConvCall(w1p, b1p, 1024, 256, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
    # fused: outputs are the interiors of two views inside ONE guarded allocation (a stray row would land in the guards)
    wf = engine.pack_bneck(w2p, w3p, w1p)
    ny, nz = y_u.t.numel(), z_u.t.numel()
    guard = 4096
    slab = torch.full((guard + ny + guard + nz + guard,), 7.0, dtype=torch.bfloat16, device=d)
    y_t = slab[guard:guard + ny].view_as(y_u.t)
    z_t = slab[2 * guard + ny:2 * guard + ny + nz].view_as(z_u.t)
    y_t.zero_()
    z_t.zero_()
    nscr = L.dafne_bottleneck_body_scratch_bytes()
    scr = torch.empty(nscr, dtype=torch.uint8, device=d)
    for _ in range(2):        # twice: the second launch finds the patch region of LDS / the dump area dirty
        _lib.check(L.dafne_bottleneck_body_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p),
                                               _lib.ptr(b1p), N, H, W, _lib.ptr(y_t), _lib.ptr(z_t), _lib.ptr(scr), nscr, st), "bneck")
    torch.cuda.synchronize()
    assert torch.equal(y_t, y_u.t)
    assert torch.equal(z_t, z_u.t)
    assert float(z_t[:, 0].abs().max()) == 0 and float(y_t[:, :, -1].abs().max()) == 0
    for lo, hi in ((0, guard), (guard + ny, 2 * guard + ny), (2 * guard + ny + nz, 3 * guard + ny + nz)):
        assert bool((slab[lo:hi] == 7.0).all())
    # the no-head form (the stage's last block: d_next NULL, zero conv1' section): d_out alone, bit for bit, nothing else written
    wf0 = engine.pack_bneck(w2p, w3p, torch.zeros_like(w1p))
    y_t.zero_()
    z_t.fill_(5.0)
    _lib.check(L.dafne_bottleneck_body_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf0), _lib.ptr(b2p), _lib.ptr(b3p), None, N, H, W,
                                           _lib.ptr(y_t), None, _lib.ptr(scr), nscr, st), "bneck (no head)")
    torch.cuda.synchronize()
    assert torch.equal(y_t, y_u.t) and bool((z_t == 5.0).all())
    for lo, hi in ((0, guard), (guard + ny, 2 * guard + ny), (2 * guard + ny + nz, 3 * guard + ny + nz)):
        assert bool((slab[lo:hi] == 7.0).all())
    # in place (the engine's form for a block whose shortcut operand is dead afterwards): d_out == d_res -- a tile reads its own X
    # rows chunk by chunk before it stores the chunk's Y rows over them; same bits, zero halo kept
    x_ip = xa.t.clone()
    z_t.zero_()
    _lib.check(L.dafne_bottleneck_body_hip(_lib.ptr(ua.t), _lib.ptr(x_ip), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p), _lib.ptr(b1p),
                                           N, H, W, _lib.ptr(x_ip), _lib.ptr(z_t), _lib.ptr(scr), nscr, st), "bneck (in place)")
    torch.cuda.synchronize()
    assert torch.equal(x_ip, y_u.t) and torch.equal(z_t, z_u.t)
    t_ref = bfr(F.relu(F.conv2d(u, w2, b2, padding=1)))
    close_bf16(t_u.nchw_float().cpu(), t_ref)
    y_ref = bfr(F.relu(F.conv2d(t_u.nchw_float().cpu(), w3, b3) + x))
    y_got = engine.Act(N, H, W, 1024, d)
    y_got.t.copy_(y_t)
    close_bf16(y_got.nchw_float().cpu(), y_ref)
    with pytest.raises(_lib.DafneHipError):
        _lib.check(L.dafne_bottleneck_body_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p),
                                               _lib.ptr(b1p), N, H, W, _lib.ptr(y_t), _lib.ptr(z_t), _lib.ptr(scr), 1024, st), "bneck")


_RP16 = False      # which form of the resident-patch kernel _rp_call builds: set by the rp16 fixture


@pytest.fixture(params=[False, True], ids=["m32", "m16"])
def rp16(request):
    """Both forms of conv3x3_rp_kernel: v_mfma_f32_32x32x16_bf16 (same K order as the generic kernels: bit-identical to them) and
    v_mfma_f32_16x16x32_bf16 (flag F_FRAG16, round 6: the instruction sums 32 k at once -- an output may land one bf16 ulp away)."""
    global _RP16
    _RP16 = request.param
    yield request.param
    _RP16 = False


def _rp_call(wp, bp, cout, flags, segs, N, d, **kw):
    from dafne_amd import engine
    if _RP16:
        return engine.ConvCall(wp, bp, 256, cout, 3, 1, 1, flags, segs, N, wfrag=engine.pack_conv3x3_frag16(wp), frag16=True, **kw)
    return engine.ConvCall(wp, bp, 256, cout, 3, 1, 1, flags, segs, N, wfrag=engine.pack_conv3x3_frag(wp), **kw)


def _same_conv_output(a, b, m16):
    """bit-identical (32x32x16 form: the generic kernel's K order) / at most one bf16 ulp apart in a few per mille of the outputs
    (16x16x32 form: another summation order inside the instruction)"""
    if not m16:
        assert torch.equal(a, b)
        return
    dlt = (a.float() - b.float()).abs()
    assert float((dlt > 0).float().mean()) < 2e-3 and float(dlt.max()) <= 2.0 ** -7 * max(float(b.float().abs().max()), 1e-30)


@pytest.mark.parametrize("cout,sizes,N,relu", [
    (256, [(64, 64)], 8, True),                                   # res4-like: exact 8 x 32 tiles
    # (the generic entry point must stay on conv_igemm here: < 200 patch-kernel tiles per nominal batch of 8)
    (256, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)], 2, False),          # five levels in one launch
    (512, [(25, 47), (5, 70)], 3, True),                          # two channel tiles; ragged rows and columns, 3 tile columns
    (256, [(1, 1), (3, 2)], 1, False),                            # smaller than a tile
])
def test_resident_patch_kernel_equals_generic_conv(cout, sizes, N, relu, rp16):
    """dafne_conv3x3_c256_hip (conv3x3_rp_kernel: whole 256-channel patch resident in LDS, weights streamed L2 -> registers,
    8 x 32 tiles) against dafne_conv2d_nhwc_bf16_hip on the same operands: same K order and epilogue expressions ->
    bit-identical outputs on every level, halo untouched; and against torch within bf16 rounding."""
    from dafne_amd import engine, _lib
    d = dev()
    g = torch.Generator().manual_seed(cout + sum(h * w for h, w in sizes))
    xs = [bfr(torch.randn(N, 256, h, w, generator=g)) for h, w in sizes]
    w = bfr(torch.randn(cout, 256, 3, 3, generator=g) / 48.0)
    b = torch.randn(cout, generator=g) * 0.1
    wp, bp = engine.pack_conv(w, b, d)
    ins = [engine.Act.from_nchw(x.to(d)) for x in xs]
    out_g = [engine.Act(N, h, wd, cout, d) for h, wd in sizes]
    out_r = [engine.Act(N, h, wd, cout, d) for h, wd in sizes]
    fl = engine.F_RELU if relu else 0
    st = _lib.current_stream()
    engine.ConvCall(wp, bp, 256, cout, 3, 1, 1, fl, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, out_g)], N)(st)
    c = _rp_call(wp, bp, cout, fl, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, out_r)], N, d)
    assert c.rp_ok() and c.kernel_name() == "conv3x3_rp"
    assert c.num_tiles() == sum(N * ((h + 7) // 8) * ((wd + 31) // 32) for h, wd in sizes)
    c(st)
    c(st)
    torch.cuda.synchronize()
    first = [r.t.clone() for r in out_r]
    c(st)
    torch.cuda.synchronize()
    for a, r, x, f in zip(out_g, out_r, xs, first):
        _same_conv_output(r.t, a.t, rp16)
        assert torch.equal(r.t, f)                        # run to run identical
        assert float(r.t[:, 0].abs().max()) == 0 and float(r.t[:, -1].abs().max()) == 0
        assert float(r.t[:, :, 0].abs().max()) == 0 and float(r.t[:, :, -1].abs().max()) == 0
        ref = F.conv2d(x, w, b, padding=1)
        close_bf16(r.nchw_float().cpu(), bfr(F.relu(ref) if relu else ref))


@pytest.mark.parametrize("gnin", [False, True])
def test_resident_patch_kernel_many_tiles_per_workgroup(gnin, rp16):
    """The persistent loop of conv3x3_rp_kernel over FIVE / FOUR tiles per workgroup (round 6: 8 x 32 tiles on a ring of three slab
    buffers -- a tile's slab 0 sits in buffer 0, 1, 2, 0, .. from tile to tile; slabs 0 / 1 of the next tile are requested during slabs
    2 / 3 of the current one): plain, ragged 150 x 150 maps (1140 tiles on 228 workgroups) bit-identical to the generic kernel;
    GN_INPUT, 16 x 128 x 128 (1024 tiles on 256 workgroups) against the patch kernel's GN_INPUT form (another K order and three
    roundings instead of one in the normalisation: one bf16 ulp in < 4e-3 of the outputs, as in the chain test)."""
    from dafne_amd import engine, _lib
    d = dev()
    g = torch.Generator().manual_seed(606 + gnin)
    C = 256
    N, H, W = (16, 128, 128) if gnin else (12, 150, 150)
    x = bfr(torch.randn(N, C, H, W, generator=g))
    w = bfr(torch.randn(C, C, 3, 3, generator=g) / 48.0)
    b = torch.randn(C, generator=g) * 0.1
    wp, bp = engine.pack_conv(w, b, d)
    xi = engine.Act.from_nchw(x.to(d))
    o_ref, o_rp = engine.Act(N, H, W, C, d), engine.Act(N, H, W, C, d)
    st = _lib.current_stream()
    kw, fl = {}, engine.F_RELU
    if gnin:
        stats = torch.zeros(1, N, C // 8, 2, dtype=torch.float32, device=d)
        stats[..., 0] = (torch.randn(N, C // 8, generator=g) * 0.2).to(d)
        stats[..., 1] = (0.5 + torch.rand(N, C // 8, generator=g)).to(d)
        gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d)
        beta = (0.3 * torch.randn(C, generator=g)).to(d)
        kw, fl = {"gn_in": (stats, gamma, beta)}, engine.F_GNIN
    ref = engine.ConvCall(wp, bp, C, C, 3, 1, 1, fl, [(xi.t, o_ref.t, None, H, W, H, W)], N, **kw)
    c = _rp_call(wp, bp, C, fl, [(xi.t, o_rp.t, None, H, W, H, W)], N, d, **kw)
    assert c.kernel_name() == "conv3x3_rp" and c.num_tiles() == N * ((H + 7) // 8) * ((W + 31) // 32) >= 1024
    ref(st)
    for _ in range(2):
        c(st)
    torch.cuda.synchronize()
    if gnin:
        assert ref.kernel_id() == 6
        dlt = (o_rp.t.float() - o_ref.t.float()).abs()
        assert float((dlt > 0).float().mean()) < 4e-3 and float(dlt.max()) <= 2.0 ** -7 * float(o_ref.t.float().abs().max())
    else:
        # (a map of this size goes to the patch kernel through the generic entry: its K order is (slab, kw, K half, kh), this kernel's
        # (slab, kh, kw, k16) like conv_igemm's -- one bf16 ulp in ~2e-4 of the outputs; the bit-identical comparison against
        # conv_igemm is test_resident_patch_kernel_equals_generic_conv at sizes that stay on it)
        assert ref.kernel_id() == 6
        dlt = (o_rp.t.float() - o_ref.t.float()).abs()
        assert float((dlt > 0).float().mean()) < 1e-3 and float(dlt.max()) <= 2.0 ** -7 * float(o_ref.t.float().abs().max())
    first = o_rp.t.clone()
    c(st)
    torch.cuda.synchronize()
    assert torch.equal(o_rp.t, first)                     # run to run identical
    close_bf16(o_rp.nchw_float().cpu()[:2], bfr(F.relu(F.conv2d(x[:2], w, b, padding=1))) if not gnin else o_rp.nchw_float().cpu()[:2])
    assert float(o_rp.t[:, 0].abs().max()) == 0 and float(o_rp.t[:, :, -1].abs().max()) == 0


def test_resident_patch_kernel_shape_rules():
    from dafne_amd import engine, _lib
    d = dev()
    wp, bp = engine.pack_conv(torch.zeros(256, 128, 3, 3), torch.zeros(256), d)
    a, o = engine.Act(1, 8, 8, 128, d), engine.Act(1, 8, 8, 256, d)
    c = engine.ConvCall(wp, bp, 128, 256, 3, 1, 1, 0, [(a.t, o.t, None, 8, 8, 8, 8)], 1)
    assert not c.rp_ok()                                  # Cin != 256
    wp, bp = engine.pack_conv(torch.zeros(256, 256, 3, 3), torch.zeros(256), d)
    a = engine.Act(1, 8, 8, 256, d)
    c = engine.ConvCall(wp, bp, 256, 256, 3, 1, 1, engine.F_RES, [(a.t, o.t, o.t, 8, 8, 8, 8)], 1)
    assert not c.rp_ok()                                  # residual epilogue
    with pytest.raises(_lib.DafneHipError):
        scr = engine.rp_scratch(d)
        _lib.check(_lib.load().dafne_conv3x3_c256_hip(ctypes.byref(c.prm), c.segs, _lib.ptr(wp), _lib.ptr(scr), scr.numel(), _lib.current_stream()), "c256")


def _gsegs(outs, tpis, N):
    from dafne_amd import _lib
    gsegs = (_lib.GnSeg * len(outs))()
    t0 = 0
    for k, (o, tpi) in enumerate(zip(outs, tpis)):
        gsegs[k] = _lib.GnSeg(o.t.data_ptr(), o.h, o.w, t0, tpi)
        t0 += tpi * N
    return gsegs


@pytest.mark.parametrize("fused_finalize", [False, True])
def test_resident_patch_kernel_gn_chain(fused_finalize, rp16):
    """Two tower layers over three ragged levels on the resident-patch kernel: layer 1 emits its raw output + per-tile
    GroupNorm partial sums (finalised by a separate launch, or by the last tile of every image: GN_FINALIZE), layer 2
    applies GroupNorm + ReLU to its patch in LDS (GN_INPUT).  Layer 1's output is bit-identical to the patch kernel's;
    the statistics agree with it to fp32 summation order and with torch; given THE SAME statistics layer 2 agrees with
    the same kernel on a separately normalised map and with the patch kernel's GN_INPUT form up to one bf16 ulp in < 4e-3 of the
    outputs (round 5: scale / shift normalisation, one rounding instead of three); and the whole chain matches torch within bf16 noise."""
    from dafne_amd import engine, _lib
    d = dev()
    L = _lib.load()
    g = torch.Generator().manual_seed(91)
    C, N = 256, 5
    sizes = [(40, 72), (17, 33), (8, 8)]
    xs = [bfr(torch.randn(N, C, h, w, generator=g)) for h, w in sizes]
    w1 = bfr(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    w2 = bfr(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.3 * torch.randn(C, generator=g)).to(d)
    ins = [engine.Act.from_nchw(x.to(d)) for x in xs]
    wp1, bp1 = engine.pack_conv(w1, b1, d)
    wp2, bp2 = engine.pack_conv(w2, b2, d)
    st = _lib.current_stream()

    def layer1(outs, rp):
        segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, outs)]
        mk = (lambda fl, **kw: _rp_call(wp1, bp1, C, fl, segs, N, d, **kw)) if rp else \
             (lambda fl, **kw: engine.ConvCall(wp1, bp1, C, C, 3, 1, 1, fl, segs, N, **kw))
        probe = mk(0)
        partial = torch.zeros(probe.num_tiles(), C // 8, 2, dtype=torch.float32, device=d)
        stats = torch.zeros(len(outs), N, C // 8, 2, dtype=torch.float32, device=d)
        fin = rp and fused_finalize
        counters = torch.zeros(len(outs), N, dtype=torch.int32, device=d)
        c = mk(engine.F_GN | (engine.F_GNFIN if fin else 0), gn_partial=partial, gn_fin=(stats, counters, 1e-5) if fin else None)
        c(st)
        tpis = c.tiles_per_image()
        assert sum(t * N for t in tpis) == c.num_tiles()
        if fin:
            c(st)                                        # the counters were reset by the first launch
        else:
            _lib.check(L.dafne_groupnorm_finalize_hip(_gsegs(outs, tpis, N), len(outs), N, C, _lib.ptr(partial), _lib.ptr(stats),
                                                      ctypes.c_float(1e-5), st), "finalize")
        return stats, partial, tpis

    raw_r = [engine.Act(N, h, w, C, d) for h, w in sizes]
    raw_p = [engine.Act(N, h, w, C, d) for h, w in sizes]
    stats_r, partial_r, tpis_r = layer1(raw_r, True)
    stats_p, _, _ = layer1(raw_p, False)
    torch.cuda.synchronize()
    for a, b_ in zip(raw_r, raw_p):
        _same_conv_output(a.t, b_.t, rp16)
    assert torch.allclose(stats_r, stats_p, rtol=2e-5, atol=2e-6)          # other tile grouping of the fp32 partial sums
    for k, x in enumerate(xs):
        y = F.conv2d(x, w1, b1, padding=1).reshape(N, C // 8, -1)
        assert torch.allclose(stats_r[k, :, :, 0].cpu(), y.mean(-1), atol=2e-3)
        assert torch.allclose(stats_r[k, :, :, 1].cpu(), torch.rsqrt(y.var(-1, unbiased=False) + 1e-5), rtol=5e-3)
    # layer 2 with GN_INPUT (normalisation applied to the patch in LDS) against the SAME kernel on a map normalised by the
    # separate pass (dafne_groupnorm_relu_nhwc_bf16_hip from the same partial sums): the same statistics; the on-load form
    # rounds once (scale / shift), the pass three times: operands one bf16 ulp apart in ~1e-3 of the elements.
    # (Against the patch kernel's GN_INPUT form the outputs differ in ~2e-4 of the elements by one bf16 ulp: that kernel
    # walks K as (slab, kw, K half, kh), this one as (slab, kh, kw, k16) like conv_igemm_kernel.)
    out_r = [engine.Act(N, h, w, C, d) for h, w in sizes]
    out_u = [engine.Act(N, h, w, C, d) for h, w in sizes]
    _rp_call(wp2, bp2, C, engine.F_GNIN, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(raw_r, out_r)], N, d,
             gn_in=(stats_r, gamma, beta))(st)
    norm = [engine.Act(N, h, w, C, d) for h, w in sizes]
    for a, b_ in zip(norm, raw_r):
        a.t.copy_(b_.t)
    stats_u = torch.zeros_like(stats_r)
    _lib.check(L.dafne_groupnorm_relu_nhwc_bf16_hip(_gsegs(norm, tpis_r, N), len(norm), N, C, _lib.ptr(partial_r), _lib.ptr(stats_u),
                                                    _lib.ptr(gamma), _lib.ptr(beta), ctypes.c_float(1e-5), st), "gn")
    _rp_call(wp2, bp2, C, 0, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(norm, out_u)], N, d)(st)
    out_p = [engine.Act(N, h, w, C, d) for h, w in sizes]
    cp = engine.ConvCall(wp2, bp2, C, C, 3, 1, 1, engine.F_GNIN, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(raw_r, out_p)],
                         N, gn_in=(stats_r, gamma, beta))
    assert cp.kernel_id() == 6
    cp(st)
    torch.cuda.synchronize()
    assert torch.equal(stats_u, stats_r)              # the fused finalisation reduces in the separate kernel's order
    for a, b_, c_ in zip(out_r, out_u, out_p):
        # round 5: the on-load form is y = max(a x + b, 0) with a = rstd gamma, b = beta - mean a (one fused rounding) against
        # ((x - mean) rstd) gamma + beta of the separate pass: a normalised bf16 OPERAND differs by one ulp now and then, which
        # moves ~1e-3 of the outputs by one bf16 ulp (measured 4e-5 .. 1e-3; scratch/rp_gnab_check.py)
        du = (a.t.float() - b_.t.float()).abs()
        assert float((du > 0).float().mean()) < 4e-3 and float(du.max()) <= 2.0 ** -7 * float(b_.t.float().abs().max())
        assert float(a.t[:, 0].abs().max()) == 0 and float(a.t[:, :, -1].abs().max()) == 0
        dlt = (a.t.float() - c_.t.float()).abs()
        assert float((dlt > 0).float().mean()) < 4e-3 and float(dlt.max()) <= 2.0 ** -7 * float(c_.t.float().abs().max())
    for x, a in zip(xs, out_r):
        y1n = bfr(_gn_ref(F.conv2d(x, w1, b1, padding=1), gamma.cpu(), beta.cpu(), C // 8))
        ref = bfr(F.conv2d(y1n, w2, b2, padding=1))
        got = a.nchw_float().cpu()
        assert (got - ref).abs().max() < 0.06 * ref.abs().max()


@pytest.mark.parametrize("gnin", [False, True])
def test_resident_patch_pair_launch_equals_two_launches(gnin, rp16):
    """dafne_conv3x3_c256_pair_hip (two layers of identical shape -- cls_tower.i / center_tower.i -- as ONE launch of the
    persistent kernel, each with its own tensors, weights and GroupNorm state) against two dafne_conv3x3_c256_hip launches:
    outputs and finalised statistics bit for bit, with plain and GroupNorm-on-load inputs, ragged levels."""
    from dafne_amd import engine, _lib
    d = dev()
    g = torch.Generator().manual_seed(123 + gnin)
    C, N = 256, 3
    sizes = [(40, 72), (17, 33), (8, 8)]
    st = _lib.current_stream()
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d)
    beta = (0.3 * torch.randn(C, generator=g)).to(d)

    def layer():
        xs = [engine.Act.from_nchw(bfr(torch.randn(N, C, h, w, generator=g)).to(d)) for h, w in sizes]
        wp, bp = engine.pack_conv(bfr(torch.randn(C, C, 3, 3, generator=g) / 48.0), torch.randn(C, generator=g) * 0.1, d)
        in_stats = torch.zeros(len(sizes), N, C // 8, 2, device=d)
        in_stats[..., 0] = (torch.randn(len(sizes), N, C // 8, generator=g) * 0.2).to(d)
        in_stats[..., 1] = (1 + 0.3 * torch.rand(len(sizes), N, C // 8, generator=g)).to(d)
        return xs, wp, bp, in_stats

    def call(L, outs):
        xs, wp, bp, in_stats = L
        segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(xs, outs)]
        kw = {"gn_in": (in_stats, gamma, beta)} if gnin else {}
        probe = _rp_call(wp, bp, C, engine.F_GNIN if gnin else 0, segs, N, d, **kw)
        partial = torch.zeros(probe.num_tiles(), C // 8, 2, dtype=torch.float32, device=d)
        stats = torch.zeros(len(outs), N, C // 8, 2, dtype=torch.float32, device=d)
        counters = torch.zeros(len(outs), N, dtype=torch.int32, device=d)
        c = _rp_call(wp, bp, C, engine.F_GN | engine.F_GNFIN | (engine.F_GNIN if gnin else 0), segs, N, d, gn_partial=partial,
                     gn_fin=(stats, counters, 1e-5), **kw)
        return c, stats

    LA, LB = layer(), layer()
    single = [[engine.Act(N, h, w, C, d) for h, w in sizes] for _ in range(2)]
    paired = [[engine.Act(N, h, w, C, d) for h, w in sizes] for _ in range(2)]
    (ca, sa), (cb, sb) = call(LA, single[0]), call(LB, single[1])
    ca(st)
    cb(st)
    (pa, psa), (pb_, psb) = call(LA, paired[0]), call(LB, paired[1])
    pair = engine.ConvPairCall(pa, pb_)
    assert pair.kernel_name() == "conv3x3_rp" and pair.flops == ca.flops + cb.flops
    pair(st)
    pair(st)                                          # the arrival counters were reset by the first launch
    torch.cuda.synchronize()
    for s_, p_ in zip(single[0] + single[1], paired[0] + paired[1]):
        assert torch.equal(s_.t, p_.t)
        assert float(p_.t[:, 0].abs().max()) == 0 and float(p_.t[:, :, -1].abs().max()) == 0
    assert torch.equal(sa, psa) and torch.equal(sb, psb)
    assert float(sa.abs().sum()) > 0


# ------------------------------------------------------------------------------------------------------------------
# conv_wr.hip: small-M layers (res5, FPN laterals / top outputs, P6 / P7) -- 128 px x 256 ch tiles, weights -> registers,
# deterministic split-K.  Reference blocks: detectron2 BottleneckBlock / FPN [recalled], backbone/fpn.py:16-37,58-91.
def _run_wr(x, w, b, k, stride, pad, flags=0, res=None, excl=True, reps=1):
    """-> (conv_wr output NCHW float CPU, conv_igemm output of the same call, slices)"""
    from dafne_amd import engine, _lib
    d = dev()
    L = _lib.load()
    n, cin, h, wd = x.shape
    cout = w.shape[0]
    a = engine.Act.from_nchw(x.to(d))
    wp, bp = engine.pack_conv(w, b, d)
    ho, wo = engine.conv_out_hw(h, wd, k, stride, pad)
    r = engine.Act.from_nchw(res.to(d)) if res is not None else None
    o_ref = engine.Act(n, ho, wo, cout, d)
    ref_call = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, flags, [(a.t, o_ref.t, r.t if r is not None else None, h, wd, ho, wo)], n,
                               shared_gpu=not excl)
    ref_call(_lib.current_stream())
    outs = []
    ws = engine.WrWorkspace(d)
    for _ in range(reps):
        o = engine.Act(n, ho, wo, cout, d)
        c = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, flags, [(a.t, o.t, r.t if r is not None else None, h, wd, ho, wo)], n,
                            shared_gpu=not excl)
        assert L.dafne_conv2d_wr_ok(ctypes.byref(c.prm), c.segs) == 1
        call = engine.WrCall(c, engine.pack_conv_frag(wp), ws)
        outs.append((o, call))
    for o, call in outs:
        call(_lib.current_stream())
    torch.cuda.synchronize()
    for o, _ in outs:
        assert float(o.t[:, 0].abs().max()) == 0 and float(o.t[:, -1].abs().max()) == 0
        assert float(o.t[:, :, 0].abs().max()) == 0 and float(o.t[:, :, -1].abs().max()) == 0
        assert torch.equal(o.t, outs[0][0].t)                           # the shared workspace leaves its tickets zero; run-to-run identical
    return outs[0][0].nchw_float().cpu(), o_ref.nchw_float().cpu(), outs[0][1].splits


WR_CASES = [
    # cin, cout, k, stride, H, W, N, flags ("r" relu, "s" residual, "u" top-down add), exclusive
    (512, 512, 3, 1, 32, 32, 2, "r", True),         # res5 conv2: 16 pixel tiles x 2 channel tiles -> 8 slices
    (512, 512, 3, 1, 32, 32, 8, "r", True),         # ... at batch 8: 2 slices of 36 steps
    (512, 2048, 1, 1, 32, 32, 8, "rs", True),       # res5 conv3: 512 tiles, one slice, residual
    (2048, 512, 1, 1, 32, 32, 3, "r", False),       # res5 conv1 in a sub-batch plan
    (1024, 2048, 1, 2, 64, 64, 2, "", True),        # res5 projection shortcut (stride 2)
    (1024, 512, 1, 2, 64, 64, 2, "r", True),        # res5.0 conv1 (STRIDE_IN_1X1)
    (2048, 256, 1, 1, 32, 32, 8, "", True),         # FPN lateral 5: 64 tiles -> 4 slices
    (1024, 256, 1, 1, 64, 64, 3, "u", True),        # FPN lateral 4 + nearest-2x top-down add
    (256, 256, 3, 1, 32, 32, 8, "", True),          # FPN output 5
    (256, 256, 3, 2, 32, 32, 8, "", True),          # P6
    (256, 256, 3, 2, 16, 16, 8, "", True),          # P7 (input = relu(P6)): 4 tiles x 8 slices
    (256, 256, 3, 2, 16, 16, 1, "", False),         # ... one image: a single half-empty tile
    (128, 256, 3, 1, 25, 38, 1, "r", True),         # ragged: 950 pixels (7.4 tiles), rows cross tile boundaries
    (64, 256, 1, 1, 7, 9, 3, "rs", True),           # one K64 step, 189 pixels
    (320, 768, 3, 2, 21, 27, 2, "u", False),        # odd input, 3 channel tiles, stride 2 + top-down add (output 11 x 14 is even in W only -> skipped below)
]


@pytest.mark.parametrize("cin,cout,k,stride,H,W,N,fl,excl", WR_CASES)
def test_conv_wr_vs_igemm_and_torch(cin, cout, k, stride, H, W, N, fl, excl):
    """dafne_conv2d_wr_hip vs torch (2 bf16 ulps, the conv tests' tolerance) and vs dafne_conv2d_nhwc_bf16_hip on the same
    operands: BIT-IDENTICAL with one slice (same K order, same epilogue expressions); with several slices the fp32 sum is
    grouped by slices -- a different rounding of the same sum, so the two kernels may differ by one bf16 ulp where the fp32
    value sits at a rounding boundary (checked: <= 1 ulp everywhere, equal for > 99 %)."""
    from dafne_amd import engine
    pad = 1 if k == 3 else 0
    ho, wo = engine.conv_out_hw(H, W, k, stride, pad)
    if "u" in fl and (ho % 2 or wo % 2):
        pytest.skip("top-down add needs an even map")
    g = torch.Generator().manual_seed(cin + cout + H + W + N)
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    flags = (engine.F_RELU if "r" in fl else 0) | (engine.F_RES if "s" in fl else 0) | (engine.F_UP if "u" in fl else 0)
    res = None
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    if "s" in fl:
        res = bfr(torch.randn(N, cout, ho, wo, generator=g))
        ref = ref + res
    if "u" in fl:
        res = bfr(torch.randn(N, cout, ho // 2, wo // 2, generator=g))
        ref = ref + F.interpolate(res, scale_factor=2, mode="nearest")
    if "r" in fl:
        ref = F.relu(ref)
    got, ig, splits = _run_wr(x, w, b, k, stride, pad, flags=flags, res=res, excl=excl, reps=3)
    close_bf16(got, bfr(ref))
    if splits == 1:
        assert torch.equal(got, ig), float((got - ig).abs().max())
    else:
        d = (got - ig).abs()
        ulp = 2.0 ** -7 * ig.abs().clamp_min(2.0 ** -10)        # one bf16 ulp is <= 2^-7 relative (absolute floor: outputs that
        assert bool((d <= ulp).all()), float((d / ulp).max())    # cancel to ~0 or sit at the ReLU edge differ by the fp32 noise)
        assert float((d == 0).float().mean()) > 0.99


def test_conv_wr_slices_follow_the_shape():
    """The slice count is a function of the PER-IMAGE shape and the CU count only -- NOT of the EXCLUSIVE hint (a batch gives the
    same bits alone on the GPU and beside other streams) and NOT of the batch size (an image gives the same bits whatever shares
    its batch); it is sized for the headline batch of 8: res5 conv2 -> 128 tiles x 2, lateral5 -> 64 tiles x 4, P7 -> 8
    (at most 8 slabs for the reducer, at least 4 K64 steps per slice)."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()

    def splits(cin, cout, k, stride, H, N, excl=True):
        pad = 1 if k == 3 else 0
        ho, wo = engine.conv_out_hw(H, H, k, stride, pad)
        a, o = engine.Act(N, H, H, cin, d), engine.Act(N, ho, wo, cout, d)
        wp, bp = engine.pack_conv(torch.zeros(cout, cin, k, k), torch.zeros(cout), d)
        c = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, 0, [(a.t, o.t, None, H, H, ho, wo)], N, shared_gpu=not excl)
        return L.dafne_conv2d_wr_splits(ctypes.byref(c.prm), c.segs)
    assert splits(512, 512, 3, 1, 32, 8) == splits(512, 512, 3, 1, 32, 1) == splits(512, 512, 3, 1, 32, 16) == 2
    assert splits(2048, 256, 1, 1, 32, 8) == 4
    assert splits(256, 256, 3, 2, 16, 8) == 8
    assert splits(512, 2048, 1, 1, 32, 8) == 1
    assert splits(512, 512, 3, 1, 32, 3, excl=False) == splits(512, 512, 3, 1, 32, 3, excl=True) == 2


# ------------------------------------------------------------------------------------------------------------------
# conv_blk_narrow.hip: a WHOLE res2 bottleneck body (3x3 + conv3 + shortcut + ReLU [+ next conv1 + ReLU]) in one kernel
@pytest.mark.parametrize("proj", [False, True])
@pytest.mark.parametrize("head", [True, False])
@pytest.mark.parametrize("N,H,W", [(1, 8, 32), (2, 64, 64), (3, 13, 21), (1, 1, 1), (1, 40, 410), (2, 256, 256), (5, 72, 104)])
def test_bottleneck_block_narrow_equals_the_separate_launches(N, H, W, head, proj):
    """dafne_bottleneck_block_narrow_hip (res2: conv2 3x3 64 -> 64 + ReLU, conv3 64 -> 256 + shortcut + ReLU, optionally the
    next block's conv1 256 -> 64 + ReLU; shortcut = identity rows or the projection of the block input, rounded to bf16 like
    the separate launch) against the generic launches: BIT FOR BIT on both outputs -- ragged tiles in both directions
    (out-of-image rows go to the dump area), 1 .. 16 tiles per workgroup (4096 tiles at 2 x 256 x 256), halo untouched,
    launched twice -- and against torch within bf16 rounding."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(7000 + H * W + 2 * head + proj)
    u = bfr(torch.relu(torch.randn(N, 64, H, W, generator=g)))
    w2 = bfr(torch.randn(64, 64, 3, 3, generator=g) / 24.0)
    b2 = torch.randn(64, generator=g) * 0.2
    w3 = bfr(torch.randn(256, 64, 1, 1, generator=g) / 8.0)
    b3 = torch.randn(256, generator=g) * 0.2
    w1 = bfr(torch.randn(64, 256, 1, 1, generator=g) / 16.0)
    b1 = torch.randn(64, generator=g) * 0.2
    ws = bfr(torch.randn(256, 64, 1, 1, generator=g) / 8.0)
    bs_ = torch.randn(256, generator=g) * 0.2
    x = bfr(torch.relu(torch.randn(N, 64 if proj else 256, H, W, generator=g)))
    st = _lib.current_stream()
    ua, xa = engine.Act.from_nchw(u.to(d)), engine.Act.from_nchw(x.to(d))
    w2p, b2p = engine.pack_conv(w2, b2, d)
    w3p, b3p = engine.pack_conv(w3, b3, d)
    w1p, b1p = engine.pack_conv(w1, b1, d)
    wsp, bsp = engine.pack_conv(ws, bs_, d)
    # the separate launches
    t_u, y_u, z_u = engine.Act(N, H, W, 64, d), engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 64, d)
    engine.ConvCall(w2p, b2p, 64, 64, 3, 1, 1, engine.F_RELU, [(ua.t, t_u.t, None, H, W, H, W)], N)(st)
    if proj:
        sc_u = engine.Act(N, H, W, 256, d)
        engine.ConvCall(wsp, bsp, 64, 256, 1, 1, 0, 0, [(xa.t, sc_u.t, None, H, W, H, W)], N)(st)
    else:
        sc_u = xa
    engine.ConvCall(w3p, b3p, 64, 256, 1, 1, 0, engine.F_RELU | engine.F_RES, [(t_u.t, y_u.t, sc_u.t, H, W, H, W)], N)(st)
    if head:
        engine.ConvCall(w1p, b1p, 256, 64, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
    # one kernel
    wf = engine.pack_blk_narrow(w2p, w3p, w1p if head else None, wsp if proj else None)
    scratch = torch.empty(L.dafne_bottleneck_block_narrow_scratch_bytes(), dtype=torch.uint8, device=d)
    y_f, z_f = engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 64, d)
    for _ in range(2):
        _lib.check(L.dafne_bottleneck_block_narrow_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p),
                                                       _lib.ptr(bsp) if proj else None, _lib.ptr(b1p) if head else None, N, H, W,
                                                       _lib.ptr(y_f.t), _lib.ptr(z_f.t) if head else None, _lib.ptr(scratch),
                                                       scratch.numel(), st), "blk_narrow")
    torch.cuda.synchronize()
    assert float(y_f.t.float().abs().max()) > 0
    assert torch.equal(y_f.t, y_u.t)
    if head:
        assert torch.equal(z_f.t, z_u.t)
    else:
        assert float(z_f.t.float().abs().max()) == 0           # untouched
    assert float(y_f.t[:, 0].abs().max()) == 0 and float(y_f.t[:, :, -1].abs().max()) == 0 and float(y_f.t[:, -1].abs().max()) == 0
    if not proj:
        # in place (identity block whose input is dead afterwards): d_out == d_res, same bits
        x_ip = xa.t.clone()
        z_f.t.zero_()
        _lib.check(L.dafne_bottleneck_block_narrow_hip(_lib.ptr(ua.t), _lib.ptr(x_ip), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p), None,
                                                       _lib.ptr(b1p) if head else None, N, H, W, _lib.ptr(x_ip),
                                                       _lib.ptr(z_f.t) if head else None, _lib.ptr(scratch), scratch.numel(), st),
                   "blk_narrow (in place)")
        torch.cuda.synchronize()
        assert torch.equal(x_ip, y_u.t)
        if head:
            assert torch.equal(z_f.t, z_u.t)
    if N * H * W <= 3 * 64 * 64:
        t_ref = bfr(F.relu(F.conv2d(u, w2, b2, padding=1)))
        sc = bfr(F.conv2d(x, ws, bs_)) if proj else x
        y_ref = bfr(F.relu(F.conv2d(t_ref, w3, b3) + sc))
        got = y_f.nchw_float().cpu()
        assert float((got - y_ref).abs().max()) < 0.03 * float(y_ref.abs().max())


# ------------------------------------------------------------------------------------------------------------------
# conv_blk_mid.hip: a WHOLE res3 bottleneck body (3x3 + conv3 + shortcut + ReLU [+ next conv1 + ReLU]) in one kernel
@pytest.mark.parametrize("head", [True, False])
@pytest.mark.parametrize("N,H,W", [(1, 8, 32), (2, 32, 64), (3, 13, 21), (1, 1, 1), (1, 20, 210), (8, 128, 128), (5, 36, 52)])
def test_bottleneck_block_mid_equals_the_separate_launches(N, H, W, head):
    """dafne_bottleneck_block_mid_hip (res3: conv2 3x3 128 -> 128 + ReLU, conv3 128 -> 512 + shortcut + ReLU, optionally the
    next block's conv1 512 -> 128 + ReLU) against the generic launches: BIT FOR BIT on both outputs -- the headline shape
    (8 x 128 x 128: 1024 tiles, four per workgroup), ragged tiles in both directions (out-of-image rows go to the dump area),
    one tile per workgroup and fewer tiles than CUs, halo untouched, launched twice -- and against torch within bf16 rounding."""
    from dafne_amd import engine, _lib
    L = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(9000 + H * W + head)
    u = bfr(torch.relu(torch.randn(N, 128, H, W, generator=g)))
    x = bfr(torch.relu(torch.randn(N, 512, H, W, generator=g)))
    w2 = bfr(torch.randn(128, 128, 3, 3, generator=g) / 34.0)
    b2 = torch.randn(128, generator=g) * 0.2
    w3 = bfr(torch.randn(512, 128, 1, 1, generator=g) / 11.0)
    b3 = torch.randn(512, generator=g) * 0.2
    w1 = bfr(torch.randn(128, 512, 1, 1, generator=g) / 22.0)
    b1 = torch.randn(128, generator=g) * 0.2
    st = _lib.current_stream()
    ua, xa = engine.Act.from_nchw(u.to(d)), engine.Act.from_nchw(x.to(d))
    w2p, b2p = engine.pack_conv(w2, b2, d)
    w3p, b3p = engine.pack_conv(w3, b3, d)
    w1p, b1p = engine.pack_conv(w1, b1, d)
    t_u, y_u, z_u = engine.Act(N, H, W, 128, d), engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
    engine.ConvCall(w2p, b2p, 128, 128, 3, 1, 1, engine.F_RELU, [(ua.t, t_u.t, None, H, W, H, W)], N)(st)
    engine.ConvCall(w3p, b3p, 128, 512, 1, 1, 0, engine.F_RELU | engine.F_RES, [(t_u.t, y_u.t, xa.t, H, W, H, W)], N)(st)
    if head:
        engine.ConvCall(w1p, b1p, 512, 128, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
    wf = engine.pack_blk_mid(w2p, w3p, w1p if head else None)
    scratch = torch.empty(L.dafne_bottleneck_block_mid_scratch_bytes(), dtype=torch.uint8, device=d)
    y_f, z_f = engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
    for _ in range(2):
        _lib.check(L.dafne_bottleneck_block_mid_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p),
                                                    _lib.ptr(b1p) if head else None, N, H, W, _lib.ptr(y_f.t),
                                                    _lib.ptr(z_f.t) if head else None, _lib.ptr(scratch), scratch.numel(), st), "blk_mid")
    torch.cuda.synchronize()
    assert float(y_f.t.float().abs().max()) > 0
    assert torch.equal(y_f.t, y_u.t)
    if head:
        assert torch.equal(z_f.t, z_u.t)
    else:
        assert float(z_f.t.float().abs().max()) == 0           # untouched
    assert float(y_f.t[:, 0].abs().max()) == 0 and float(y_f.t[:, :, -1].abs().max()) == 0 and float(y_f.t[:, -1].abs().max()) == 0
    # in place (the shortcut operand is dead afterwards): d_out == d_res, same bits
    x_ip = xa.t.clone()
    z_f.t.zero_()
    _lib.check(L.dafne_bottleneck_block_mid_hip(_lib.ptr(ua.t), _lib.ptr(x_ip), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p),
                                                _lib.ptr(b1p) if head else None, N, H, W, _lib.ptr(x_ip),
                                                _lib.ptr(z_f.t) if head else None, _lib.ptr(scratch), scratch.numel(), st), "blk_mid (in place)")
    torch.cuda.synchronize()
    assert torch.equal(x_ip, y_u.t)
    if head:
        assert torch.equal(z_f.t, z_u.t)
    if N * H * W <= 3 * 64 * 64:
        t_ref = bfr(F.relu(F.conv2d(u, w2, b2, padding=1)))
        y_ref = bfr(F.relu(F.conv2d(t_ref, w3, b3) + x))
        got = y_f.nchw_float().cpu()
        assert float((got - y_ref).abs().max()) < 0.03 * float(y_ref.abs().max())
