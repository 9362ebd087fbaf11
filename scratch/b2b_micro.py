# time dafne_bottleneck_tail_head_hip alone (res4 shape: batch 8, 64x64) with the library in DAFNE_AMD_LIB
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = 8, 64, 64
g = torch.Generator().manual_seed(5)
bf = torch.bfloat16
ta = engine.Act.from_nchw(torch.randn(N, 256, H, W, generator=g).to(d))
xa = engine.Act.from_nchw(torch.randn(N, 1024, H, W, generator=g).to(d))
w3p, b3p = engine.pack_conv(torch.randn(1024, 256, 1, 1, generator=g) / 16, torch.randn(1024, generator=g), d)
w1p, b1p = engine.pack_conv(torch.randn(256, 1024, 1, 1, generator=g) / 32, torch.randn(256, generator=g), d)
wf = engine.pack_b2b(w3p, w1p)
y, z = engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)
st = _lib.current_stream()
def run():
    _lib.check(L.dafne_bottleneck_tail_head_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p), _lib.ptr(b1p),
                                                N, H, W, _lib.ptr(y.t), _lib.ptr(z.t), st), "b2b")
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / 20)
fl = 2 * N * H * W * 2 * 256 * 1024
print("%s: %.1f us  %.0f TF" % (os.environ.get("DAFNE_AMD_LIB", "default"), best * 1e3, fl / (best * 1e-3) / 1e12))
