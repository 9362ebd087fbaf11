import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
N, H, W, cin, cout, k = 8, 128, 128, 256, 256, 3
a = engine.Act(N, H, W, cin, dev); a.t[:, 1:-1, 1:-1].normal_()
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
o = engine.Act(N, H, W, cout, dev)
c = engine.ConvCall(wp, bp, cin, cout, k, 1, 1, 0, [(a.t, o.t, None, H, W, H, W)], N)
st = _lib.current_stream()
for _ in range(5): c(st)
torch.cuda.synchronize()
