import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
def mk(cin, cout, k, stride, H, W, N=8, res=False, relu=True):
    pad = 1 if k == 3 else 0
    a = engine.Act(N, H, W, cin, dev); a.t[:, 1:-1, 1:-1].normal_()
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
    ho, wo = engine.conv_out_hw(H, W, k, stride, pad)
    o = engine.Act(N, ho, wo, cout, dev)
    r = engine.Act(N, ho, wo, cout, dev) if res else None
    if r is not None: r.t[:, 1:-1, 1:-1].normal_()
    f = (engine.F_RES if res else 0) | (engine.F_RELU if relu else 0)
    c = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, f, [(a.t, o.t, r.t if r is not None else None, H, W, ho, wo)], N)
    byt = (a.t.numel() + o.t.numel() * (2 if res else 1)) * 2
    return c, byt, (a, o, r, wp, bp)
# several independent instances per shape, cycled, so that back-to-back launches do not find their operands in MALL
SHAPES = [("res4 conv3 256->1024 +res", 256, 1024, 1, 64, True),
          ("res4 conv1 1024->256", 1024, 256, 1, 64, False),
          ("res2 conv3 64->256 +res", 64, 256, 1, 256, True),
          ("res3 conv3 128->512 +res", 128, 512, 1, 128, True),
          ("res2 conv1 256->64", 256, 64, 1, 256, False),
          ("res4 conv3 no res", 256, 1024, 1, 64, False),
          ("res3 conv1 512->128", 512, 128, 1, 128, False),
          ("res5 conv3 512->2048 +res", 512, 2048, 1, 32, True),
          ("res5 conv1 2048->512", 2048, 512, 1, 32, False),
          ("lateral4 1024->256", 1024, 256, 1, 64, False)]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in sys.argv[1:])]
st = _lib.current_stream()
for label, cin, cout, k, hw, res in SHAPES:
    inst = [mk(cin, cout, k, 1, hw, hw, res=res) for _ in range(4)]
    for c, _, _ in inst: c(st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for c, _, _ in inst: c(st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    c, byt, _ = inst[0]
    print("%-30s tile=%d : %7.1f us  %7.1f TF  %5.2f TB/s" % (label, c.tile_pixels(), best, c.flops / best / 1e6, byt / best / 1e6))
