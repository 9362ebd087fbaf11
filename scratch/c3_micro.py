import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
def mk(cin, cout, H, W, N=8):
    a = engine.Act(N, H, W, cin, dev); a.t[:, 1:-1, 1:-1].normal_()
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
    o = engine.Act(N, H, W, cout, dev)
    c = engine.ConvCall(wp, bp, cin, cout, 3, 1, 1, engine.F_RELU, [(a.t, o.t, None, H, W, H, W)], N)
    return c, (a, o, wp, bp)
st = _lib.current_stream()
for label, cin, cout, hw in [("P3-like 256->256 128x128", 256, 256, 128), ("res4 conv2 256->256 64x64", 256, 256, 64), ("res3 conv2 128->128 128x128", 128, 128, 128)]:
    inst = [mk(cin, cout, hw, hw) for _ in range(3)]
    for c, _ in inst: c(st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for c, _ in inst: c(st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 15)
    c = inst[0][0]
    print("%-30s tile=%d : %7.1f us  %7.1f TF" % (label, c.tile_pixels(), best, c.flops / best / 1e6))
