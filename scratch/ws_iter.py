import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, numpy as np
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
cin, cout, hw = 256, 1024, 64
N = 8
a = engine.Act(N, hw, hw, cin, dev); a.t[:, 1:-1, 1:-1].normal_()
w = torch.randn(cout, cin, 1, 1) / cin ** 0.5
wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
o = engine.Act(N, hw, hw, cout, dev)
r = engine.Act(N, hw, hw, cout, dev); r.t[:, 1:-1, 1:-1].normal_()
log = torch.zeros(512 * 64, dtype=torch.int64, device=dev)
c = engine.ConvCall(wp, bp, cin, cout, 1, 1, 0, engine.F_RES | engine.F_RELU, [(a.t, o.t, r.t, hw, hw, hw, hw)], N, gn_partial=log)
st = _lib.current_stream()
junk = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
for _ in range(3):
    junk.fill_(1); c(st)
torch.cuda.synchronize()
L = log.cpu().numpy().reshape(512, 64).astype(np.int64)
c0 = L[:, 1]
S = L[:, 2:2 + 4 * 12].reshape(512, 4, 12)     # per tile: start, 8 iteration stamps (after barrier), kloop done, res in LDS, acc pass done
names = ["tile start"] + ["it%d barrier" % i for i in range(8)] + ["K done", "res->LDS", "acc pass"]
for t in range(4):
    d = np.diff(S[:, t, :], axis=1).mean(0)
    print("tile %d: first stamp +%6.0f | " % (t, (S[:, t, 0] - (S[:, t - 1, 11] if t else c0)).mean()) + " ".join("%5.0f" % x for x in d))
print("total", (S[:, 3, 11] - c0).mean())
