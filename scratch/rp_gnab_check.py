"""conv3x3_rp<GN on load>: the scale/shift variant (-DDAFNE_RP_GNAB library) against the separate normalisation pass + plain
kernel and against torch.  usage: DAFNE_AMD_LIB=... rp_gnab_check.py  (prints difference statistics; multi-tile workgroups,
ragged levels, five images so that tiles of a workgroup change image, paired launch)"""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch.nn.functional as F
from dafne_amd import engine, _lib
from test_gpu_conv import _rp_call, _gsegs, bfr, _gn_ref
d = torch.device("cuda", 0); L = _lib.load(); st = _lib.current_stream()
g = torch.Generator().manual_seed(91)
C = 256
for N, sizes in ((5, [(40, 72), (17, 33), (8, 8)]), (8, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)])):
    xs = [bfr(torch.randn(N, C, h, w, generator=g)) for h, w in sizes]
    w1 = bfr(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5); w2 = bfr(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d); beta = (0.3 * torch.randn(C, generator=g)).to(d)
    ins = [engine.Act.from_nchw(x.to(d)) for x in xs]
    wp1, bp1 = engine.pack_conv(w1, b1, d); wp2, bp2 = engine.pack_conv(w2, b2, d)
    raw = [engine.Act(N, h, w, C, d) for h, w in sizes]
    segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, raw)]
    probe = _rp_call(wp1, bp1, C, 0, segs, N, d)
    partial = torch.zeros(probe.num_tiles(), C // 8, 2, dtype=torch.float32, device=d)
    stats = torch.zeros(len(raw), N, C // 8, 2, dtype=torch.float32, device=d)
    c1 = _rp_call(wp1, bp1, C, engine.F_GN, segs, N, d, gn_partial=partial)
    c1(st)
    tpis = c1.tiles_per_image()
    _lib.check(L.dafne_groupnorm_finalize_hip(_gsegs(raw, tpis, N), len(raw), N, C, _lib.ptr(partial), _lib.ptr(stats), ctypes.c_float(1e-5), st), "fin")
    out_r = [engine.Act(N, h, w, C, d) for h, w in sizes]
    out_u = [engine.Act(N, h, w, C, d) for h, w in sizes]
    for rep in range(2):
        _rp_call(wp2, bp2, C, engine.F_GNIN, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(raw, out_r)], N, d, gn_in=(stats, gamma, beta))(st)
    norm = [engine.Act(N, h, w, C, d) for h, w in sizes]
    for a, b_ in zip(norm, raw): a.t.copy_(b_.t)
    stats_u = torch.zeros_like(stats)
    _lib.check(L.dafne_groupnorm_relu_nhwc_bf16_hip(_gsegs(norm, tpis, N), len(norm), N, C, _lib.ptr(partial), _lib.ptr(stats_u), _lib.ptr(gamma), _lib.ptr(beta), ctypes.c_float(1e-5), st), "gn")
    _rp_call(wp2, bp2, C, 0, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(norm, out_u)], N, d)(st)
    torch.cuda.synchronize()
    for k, (a, b_, x) in enumerate(zip(out_r, out_u, xs)):
        dl = (a.t.float() - b_.t.float()).abs()
        halo = float(a.t[:, 0].abs().max()) + float(a.t[:, :, -1].abs().max()) + float(a.t[:, -1].abs().max()) + float(a.t[:, :, 0].abs().max())
        y1n = bfr(_gn_ref(F.conv2d(x, w1, b1, padding=1), gamma.cpu(), beta.cpu(), C // 8))
        ref = bfr(F.conv2d(y1n, w2, b2, padding=1))
        got, sep = a.nchw_float().cpu(), b_.nchw_float().cpu()
        print("N=%d level %d %s: vs separate pass: differing %.2e of elements, max |d| %.4f (max |out| %.2f); vs torch: on-load max %.4f, separate max %.4f; halo %g"
              % (N, k, sizes[k], float((dl > 0).float().mean()), float(dl.max()), float(b_.t.float().abs().max()),
                 float((got - ref).abs().max()), float((sep - ref).abs().max()), halo))
