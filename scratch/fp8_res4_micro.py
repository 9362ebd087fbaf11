"""One-level fp8 3x3 layers (res4 conv2: 64x64, FPN outputs: 128/64/32) on conv3x3_patch_fp8 vs conv3x3_rp8.  usage: fp8_res4_micro.py"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
d = torch.device("cuda", 0)
C = 256
g = torch.Generator().manual_seed(0)
w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
b = torch.randn(C, generator=g) * 0.1
wp, bp = engine.pack_conv(w, b, d)
wq, ws = engine.pack_conv_fp8(w, d)
st = _lib.current_stream()
for (h, B) in ((64, 16), (64, 6), (64, 5), (128, 16), (128, 5), (32, 16), (32, 5)):
    res = []
    for env in ("0", "1"):
        os.environ["DAFNE_CONV_RP8"] = env
        sets = []
        for k in range(3):
            i = engine.Act(B, h, h, C, d); o = engine.Act(B, h, h, C, d)
            i.t[:, 1:-1, 1:-1, :] = torch.randn(B, h, h, C, device=d).to(torch.bfloat16)
            sets.append((engine.ConvCall(wq, bp, C, C, 3, 1, 1, engine.F_RELU, [(i.t, o.t, None, h, h, h, h)], B, fp8=(ws, 1.0)), i, o))
        for s in sets: s[0](st)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for r in range(30): sets[r % 3][0](st)
        e.record(); torch.cuda.synchronize()
        res.append("%s %.1f us (%d tiles)" % (sets[0][0].kernel_name(), 1e3 * a.elapsed_time(e) / 30, sets[0][0].num_tiles()))
    print("%dx%d batch %d: " % (h, h, B) + "   ".join(res), flush=True)
