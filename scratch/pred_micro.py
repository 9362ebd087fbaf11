"""Prediction layer (256 -> cout, 3x3, five levels of a 1024^2 batch, GroupNorm + ReLU on load, fp32 out): conv3x3_slab vs
conv3x3_pred16.  usage: pred_micro.py [batch]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = torch.device("cuda", 0)
C = 256
sizes = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
g = torch.Generator().manual_seed(0)
st = _lib.current_stream()
for cout in (15, 9, 2):
    w = torch.randn(cout, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    wp, bp = engine.pack_conv(w, b, d)
    gamma = torch.ones(C, device=d); beta = torch.zeros(C, device=d)
    for gn in (True, False):
        res = []
        for env in ("0", "1"):
            os.environ["DAFNE_CONV_PRED16"] = env
            sets = []
            for k in range(3):
                ins = [engine.Act(B, h, ww, C, d) for h, ww in sizes]
                for a in ins:
                    a.t[:, 1:-1, 1:-1, :] = torch.randn(B, a.h, a.w, C, device=d).to(torch.bfloat16)
                outs = [torch.empty(B, h, ww, cout, dtype=torch.float32, device=d) for h, ww in sizes]
                stats = torch.zeros(5, B, C // 8, 2, device=d); stats[..., 1] = 1.0
                segs = [(i.t, o, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, outs)]
                c = engine.ConvCall(wp, bp, C, cout, 3, 1, 1, engine.F_F32 | (engine.F_GNIN if gn else 0), segs, B,
                                    gn_in=(stats, gamma, beta) if gn else None)
                sets.append((c, ins, outs, stats))
            for s in sets: s[0](st)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for r in range(30): sets[r % 3][0](st)
            e.record(); torch.cuda.synchronize()
            us = 1e3 * a.elapsed_time(e) / 30
            res.append("%s %.1f us (%.2f TB/s in)" % (sets[0][0].kernel_name(), us, sets[0][0].bytes / us / 1e6))
        print("cout %2d gn %d batch %d: " % (cout, gn, B) + "   ".join(res), flush=True)
