import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
def bench(cin, cout, k, stride, H, W, N=8, flags=0, res=False, reps=20, label=""):
    pad = 1 if k == 3 else 0
    a = engine.Act(N, H, W, cin, dev); a.t[:, 1:-1, 1:-1].normal_()
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
    ho, wo = engine.conv_out_hw(H, W, k, stride, pad)
    o = engine.Act(N, ho, wo, cout, dev)
    r = engine.Act(N, ho, wo, cout, dev) if res else None
    if r is not None: r.t[:, 1:-1, 1:-1].normal_()
    f = flags | (engine.F_RES | engine.F_RELU if res else 0)
    c = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, f, [(a.t, o.t, r.t if r is not None else None, H, W, ho, wo)], N)
    st = _lib.current_stream()
    for _ in range(3): c(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c(st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byt = (a.t.numel() + o.t.numel() * (2 if res else 1)) * 2
    print("%-28s cin%-4d cout%-4d k%d %3dx%-3d res=%d tile=%d : %7.1f us  %7.1f TF  %5.2f TB/s" % (label, cin, cout, k, H, W, res, c.tile_pixels(), us, c.flops / us / 1e6, byt / us / 1e6))
for rep in range(3):
    bench(256, 256, 3, 1, 128, 128, label="head-like p3 (256 tile)")
    bench(256, 256, 3, 1, 64, 64, label="res4 conv2 (128 tile)")
    bench(256, 1024, 1, 1, 64, 64, res=True, label="res4 conv3")
