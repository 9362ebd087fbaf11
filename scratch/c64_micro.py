# res2 conv2 (3x3, 64 -> 64): dedicated kernel vs the generic launch, batch 8 at 256 x 256
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = 8, 256, 256
g = torch.Generator().manual_seed(5)
wp, bp = engine.pack_conv(torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g) * 0.1, d)
st = _lib.current_stream()
xa = engine.Act(N, H, W, 64, d); xa.t[:, 1:-1, 1:-1].normal_(); xa.t.relu_()
y = engine.Act(N, H, W, 64, d)
c = engine.ConvCall(wp, bp, 64, 64, 3, 1, 1, engine.F_RELU, [(xa.t, y.t, None, H, W, H, W)], N)
def fused():
    _lib.check(L.dafne_conv3x3_c64_hip(_lib.ptr(xa.t), _lib.ptr(wp), _lib.ptr(bp), N, H, W, 1, _lib.ptr(y.t), st), "c64")
for name, fn in (("generic", lambda: c(st)), ("c64", fused)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    fl = 2 * N * H * W * 64 * 576
    print("%-8s %.1f us   %.0f TFLOP/s" % (name, best * 1e3, fl / best * 1e-9))
