# prediction convolution (3x3, 256 -> 15, fp32 out) over the five head levels at batch 8: slab kernel without GN_INPUT
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
N = 8
lv = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
for cout in (15, 9, 2):
    w = torch.randn(cout, 256, 3, 3) / 48
    wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
    ins = [engine.Act(N, h, w_, 256, dev) for h, w_ in lv]
    for a in ins: a.t[:, 1:-1, 1:-1].normal_()
    outs = [torch.zeros(N, h, w_, 32 if False else cout, dtype=torch.float32, device=dev) for h, w_ in lv]
    segs = [(a.t, o, None, h, w_, h, w_) for a, o, (h, w_) in zip(ins, outs, lv)]
    c = engine.ConvCall(wp, bp, 256, cout, 3, 1, 1, engine.F_F32, segs, N)
    st = _lib.current_stream()
    for _ in range(3): c(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c(st)
    e1.record(); torch.cuda.synchronize()
    print("cout %2d: %s  %.1f us" % (cout, c.kernel_name(), e0.elapsed_time(e1) * 1e3 / 20))
