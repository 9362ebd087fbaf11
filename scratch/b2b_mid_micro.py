# res3 tail+head pair: fused kernel vs the two generic launches, batch 8 at 128 x 128
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = 8, 128, 128
g = torch.Generator().manual_seed(5)
w3p, b3p = engine.pack_conv(torch.randn(512, 128, 1, 1, generator=g) / 11, torch.randn(512, generator=g) * 0.1, d)
w1p, b1p = engine.pack_conv(torch.randn(128, 512, 1, 1, generator=g) / 22, torch.randn(128, generator=g) * 0.1, d)
wf = engine.pack_b2b_mid(w3p, w1p)
st = _lib.current_stream()
ta = engine.Act(N, H, W, 128, d); ta.t.normal_(); ta.t.relu_()
xa = engine.Act(N, H, W, 512, d); xa.t.normal_(); xa.t.relu_()
y, z = engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
c3 = engine.ConvCall(w3p, b3p, 128, 512, 1, 1, 0, engine.F_RELU | engine.F_RES, [(ta.t, y.t, xa.t, H, W, H, W)], N)
c1 = engine.ConvCall(w1p, b1p, 512, 128, 1, 1, 0, engine.F_RELU, [(y.t, z.t, None, H, W, H, W)], N)
def fused():
    _lib.check(L.dafne_bottleneck_tail_head_mid_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p), _lib.ptr(b1p), N, H, W, _lib.ptr(y.t), _lib.ptr(z.t), st), "n")
def two():
    c3(st); c1(st)
for name, fn in (("two launches", two), ("fused", fused)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    mb = N * H * W * (128 + 512 + 512 + 128) * 2 / 1e6
    print("%-13s %.1f us   %.2f TB/s counting the fused kernel's %.0f MB" % (name, best * 1e3, mb / best * 1e-6 * 1e3, mb))
