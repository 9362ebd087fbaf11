import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, numpy as np
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
cin, cout, hw, res = [(256, 1024, 64, True), (64, 256, 256, True), (128, 512, 128, True)][int(sys.argv[1]) if len(sys.argv) > 1 else 0]
N = 8
a = engine.Act(N, hw, hw, cin, dev); a.t[:, 1:-1, 1:-1].normal_()
w = torch.randn(cout, cin, 1, 1) / cin ** 0.5
wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
o = engine.Act(N, hw, hw, cout, dev)
r = engine.Act(N, hw, hw, cout, dev); r.t[:, 1:-1, 1:-1].normal_()
log = torch.zeros(512 * 64, dtype=torch.int64, device=dev)
f = engine.F_RES | engine.F_RELU
c = engine.ConvCall(wp, bp, cin, cout, 1, 1, 0, f, [(a.t, o.t, r.t, hw, hw, hw, hw)], N, gn_partial=log)
st = _lib.current_stream()
junk = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
for _ in range(3):
    junk.fill_(1); c(st)
torch.cuda.synchronize()
L = log.cpu().numpy().reshape(512, 64).astype(np.int64)
rt0 = L[:, 0]; c0 = L[:, 1]
print("start skew (100MHz ticks): min %d max %d" % (rt0.min() - rt0.min(), rt0.max() - rt0.min()))
ntile = 0
st_ = L[:, 2:]
nz = (st_ != 0).sum(1)
print("stamps per block: min %d max %d" % (nz.min(), nz.max()))
nt = nz.min() // 4
S = st_[:, :nt * 4].reshape(512, nt, 4)
print("first tile start - kernel start: mean %.0f cyc" % (S[:, 0, 0] - c0).mean())
for t in range(nt):
    kloop = (S[:, t, 1] - S[:, t, 0]).mean(); rwait = (S[:, t, 2] - S[:, t, 1]).mean(); accp = (S[:, t, 3] - S[:, t, 2]).mean()
    nxt = (S[:, t + 1, 0] - S[:, t, 3]).mean() if t + 1 < nt else float("nan")
    print("tile %2d: K loop %6.0f  residual->LDS %6.0f  acc pass %6.0f  stores+next setup %6.0f" % (t, kloop, rwait, accp, nxt))
print("total cycles per block: mean %.0f  max %.0f" % ((S[:, -1, 3] - c0).mean(), (S[:, -1, 3] - c0).max()))
