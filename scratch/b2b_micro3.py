# b2b in the network's own buffer pattern: Y ping-pong (X of block k+1 = Y of block k), conv2 (3x3 256->256) in between
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
from dafne_amd.engine import F_RELU
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = 8, 64, 64
g = torch.Generator().manual_seed(5)
w3p, b3p = engine.pack_conv(torch.randn(1024, 256, 1, 1, generator=g) / 16, torch.randn(1024, generator=g) * 0.1, d)
w1p, b1p = engine.pack_conv(torch.randn(256, 1024, 1, 1, generator=g) / 32, torch.randn(256, generator=g) * 0.1, d)
w2p, b2p = engine.pack_conv(torch.randn(256, 256, 3, 3, generator=g) / 48, torch.randn(256, generator=g) * 0.1, d)
wf = engine.pack_b2b(w3p, w1p)
st = _lib.current_stream()
Y = [engine.Act.from_nchw(torch.relu(torch.randn(N, 1024, H, W, generator=g)).to(d)), engine.Act(N, H, W, 1024, d)]
T = [engine.Act.from_nchw(torch.relu(torch.randn(N, 256, H, W, generator=g)).to(d)) for _ in range(3)]
def b2b(k):
    x, y = Y[k & 1], Y[(k + 1) & 1]
    t, z = T[k % 3], T[(k + 1) % 3]
    _lib.check(L.dafne_bottleneck_tail_head_hip(_lib.ptr(t.t), _lib.ptr(x.t), _lib.ptr(wf), _lib.ptr(b3p), _lib.ptr(b1p), N, H, W, _lib.ptr(y.t), _lib.ptr(z.t), st), "b2b")
convs = [engine.ConvCall(w2p, b2p, 256, 256, 3, 1, 1, F_RELU, [(T[(k + 1) % 3].t, T[(k + 2) % 3].t, None, H, W, H, W)], N) for k in range(3)]
def timed(with_conv, reps=24):
    evs = []
    for k in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b2b(k); b.record()
        if with_conv: convs[k % 3](st)
        evs.append((a, b))
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in evs[4:]) * 1e3, sum(a.elapsed_time(b) for a, b in evs[4:]) / (reps - 4) * 1e3
for wc in (False, True):
    timed(wc); r = timed(wc)
    print("network pattern (Y ping-pong%s): b2b min %.1f us, mean %.1f us" % (" + conv2 between" if wc else "", *r))
