# dafne_bottleneck_tail_head_hip at the res4 shape with K cycled buffer sets (K * 168 MB: beyond the 256 MB MALL for K >= 2), post-ReLU inputs
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = 8, 64, 64
g = torch.Generator().manual_seed(5)
w3p, b3p = engine.pack_conv(torch.randn(1024, 256, 1, 1, generator=g) / 16, torch.randn(1024, generator=g), d)
w1p, b1p = engine.pack_conv(torch.randn(256, 1024, 1, 1, generator=g) / 32, torch.randn(256, generator=g), d)
wf = engine.pack_b2b(w3p, w1p)
st = _lib.current_stream()
for K in (1, 2, 4, 8):
    sets = []
    for k in range(K):
        ta = engine.Act.from_nchw(torch.relu(torch.randn(N, 256, H, W, generator=g)).to(d))
        xa = engine.Act.from_nchw(torch.relu(torch.randn(N, 1024, H, W, generator=g)).to(d))
        sets.append((ta, xa, engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)))
    def run(k):
        ta, xa, y, z = sets[k % K]
        _lib.check(L.dafne_bottleneck_tail_head_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p), _lib.ptr(b1p), N, H, W, _lib.ptr(y.t), _lib.ptr(z.t), st), "b2b")
    for k in range(2 * K): run(k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        a.record()
        for k in range(24): run(k)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 24)
    fl = 2 * N * H * W * 2 * 256 * 1024
    print("K=%d buffer sets: %.1f us  %.0f TF  %.2f TB/s algorithmic" % (K, best * 1e3, fl / (best * 1e-3) / 1e12, 168e6 / (best * 1e-3) / 1e12))
    del sets
