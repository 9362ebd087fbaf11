"""Head tower layer (256 -> 256, 3x3, five levels of a 1024^2 batch) on the bf16 patch kernel vs the fp8 patch kernel,
both with GroupNorm-on-load + GroupNorm statistics (the form the head uses).  usage: fp8_micro.py [batch]"""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = torch.device("cuda", 0)
C = 256
sizes = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
g = torch.Generator().manual_seed(0)
w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
b = torch.randn(C, generator=g) * 0.1
wp, bp = engine.pack_conv(w, b, d)
wq, ws = engine.pack_conv_fp8(w, d)
gamma = torch.ones(C, device=d); beta = torch.zeros(C, device=d)
NI = 3          # cycled instances (no MALL reuse between launches)
sets = []
for k in range(NI):
    ins = [engine.Act(B, h, ww, C, d) for h, ww in sizes]
    for a in ins:
        a.t[:, 1:-1, 1:-1, :] = torch.randn(B, a.h, a.w, C, device=d).to(torch.bfloat16)
    outs = [engine.Act(B, h, ww, C, d) for h, ww in sizes]
    stats = torch.zeros(5, B, C // 8, 2, device=d); stats[..., 1] = 1.0
    segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, outs)]
    partial = torch.zeros(4096, C // 8, 2, device=d)
    fl = engine.F_GN | engine.F_GNIN
    cb = engine.ConvCall(wp, bp, C, C, 3, 1, 1, fl, segs, B, gn_partial=partial, gn_in=(stats, gamma, beta))
    f8 = engine.pack_conv3x3_frag8(wq)
    cq = engine.ConvCall(wq, bp, C, C, 3, 1, 1, fl, segs, B, gn_partial=partial, gn_in=(stats, gamma, beta), fp8=(ws, 1.0))
    cr = engine.ConvCall(wq, bp, C, C, 3, 1, 1, fl, segs, B, gn_partial=partial, gn_in=(stats, gamma, beta), fp8=(ws, 1.0), frag8=f8)
    crn = engine.ConvCall(wq, bp, C, C, 3, 1, 1, engine.F_GN, segs, B, gn_partial=partial, fp8=(ws, 1.0), frag8=f8)
    cbr = engine.ConvCall(wp, bp, C, C, 3, 1, 1, fl, segs, B, gn_partial=partial, gn_in=(stats, gamma, beta), wfrag=engine.pack_conv3x3_frag(wp))
    assert cq.kernel_name() == "conv3x3_patch_fp8" and cr.kernel_name() == "conv3x3_rp8" and cbr.kernel_name() == "conv3x3_rp"
    cn = engine.ConvCall(wp, bp, C, C, 3, 1, 1, engine.F_GN, segs, B, gn_partial=partial)
    sets.append((cb, cq, cn, cr, crn, cbr, ins, outs))
st = _lib.current_stream()
flops = sets[0][0].flops
for name, idx in (("bf16 conv3x3_patch<GNIN>", 0), ("fp8 conv3x3_patch_fp8<GNIN>", 1), ("bf16 conv3x3_patch (no GN input)", 2),
                  ("fp8 conv3x3_rp8<GNIN>", 3), ("fp8 conv3x3_rp8 (no GN input)", 4), ("bf16 conv3x3_rp<GNIN>", 5)):
    for s in sets: s[idx](st)
    torch.cuda.synchronize()
    reps = 12
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(reps): sets[r % NI][idx](st)
    e.record(); torch.cuda.synchronize()
    us = 1e3 * a.elapsed_time(e) / reps
    print("%-30s batch %d: %.1f us  %.0f TFLOP/s  (%d tiles)" % (name, B, us, flops / us / 1e6, sets[0][idx].num_tiles()))
