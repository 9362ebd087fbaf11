"""Head tower layer (256 -> 256, 3x3, five levels of a 1024^2 batch) on the patch kernel vs the resident-patch kernel
(conv3x3_rp_kernel), with / without GroupNorm-on-load, both with GroupNorm statistics.  usage: rp_micro.py [batch]"""
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
wf, _f16 = engine.pack_rp(wp)       # DAFNE_RP_MFMA16=0: the 32x32x16 form
f16 = bool(_f16)
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
    partial = torch.zeros(8192, C // 8, 2, device=d)
    fl = engine.F_GN | engine.F_GNIN
    calls = [engine.ConvCall(wp, bp, C, C, 3, 1, 1, fl, segs, B, gn_partial=partial, gn_in=(stats, gamma, beta)),
             engine.ConvCall(wp, bp, C, C, 3, 1, 1, fl, segs, B, gn_partial=partial, gn_in=(stats, gamma, beta), wfrag=wf, frag16=f16),
             engine.ConvCall(wp, bp, C, C, 3, 1, 1, engine.F_GN, segs, B, gn_partial=partial),
             engine.ConvCall(wp, bp, C, C, 3, 1, 1, engine.F_GN, segs, B, gn_partial=partial, wfrag=wf, frag16=f16)]
    sets.append(calls)
st = _lib.current_stream()
flops = sets[0][0].flops
names = ["patch<GNIN>", "rp<GNIN>", "patch plain", "rp plain"]
res = {}
for rnd in range(3):
    for idx, name in enumerate(names):
        for s in sets: s[idx](st)
        torch.cuda.synchronize()
        reps = 12
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for r in range(reps): sets[r % NI][idx](st)
        e.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(a.elapsed_time(e) / reps)
for name in names:
    v = res[name]
    print("%-14s %.1f us (median %.1f)  %.0f TFLOP/s   tiles %d" % (name, min(v) * 1e3, sorted(v)[1] * 1e3, flops / (min(v) * 1e-3) / 1e12,
                                                                   sets[0][names.index(name)].num_tiles()))
if os.environ.get("DAFNE_RP_STAMPS"):
    # a -DDAFNE_RP_TIMING library: stamps of the THIRD tile of every workgroup (steady state) land in rows 4096.. of the partial buffer
    part = sets[(12 - 1) % NI][3].keep[2]
    torch.cuda.synchronize()
    part.zero_(); sets[(12 - 1) % NI][3](st); torch.cuda.synchronize()
    s = part.view(-1)[4096 * 64:].view(torch.int64)[:256 * 16].reshape(256, 16).cpu()
    d = (s[:, 1:8] - s[:, 0:7]).float().median(dim=0).values.tolist()
    tot = (s[:, 7] - s[:, 0]).float().median().item()
    rt = (s[:, 9] - s[:, 8]).float().median().item()          # 100 MHz ticks
    print("cycles, median over 256 workgroups (third tile): decode..step0 barrier %d | steps 0-35 %d | 36-71 %d | 72-107 %d | 108-143 %d | end barrier %d | slab3 DMA + epilogue %d | total %d"
          % (d[0], d[1], d[2], d[3], d[4], d[5], d[6], tot))
    print("tile wall time %.2f us -> shader clock %.2f GHz" % (rt / 100.0, tot / (rt * 10.0)))
