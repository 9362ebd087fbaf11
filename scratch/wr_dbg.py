"""debug: one layer, one kernel.  python scratch/wr_dbg.py cin cout k stride H N excl which(base|wr) [flags]"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
cin, cout, k, stride, H, N, excl = [int(v) for v in sys.argv[1:8]]
which = sys.argv[8]
d = torch.device("cuda", 0)
L = _lib.load()
pad = 1 if k == 3 else 0
ho, wo = engine.conv_out_hw(H, H, k, stride, pad)
g = torch.Generator().manual_seed(0)
a = engine.Act.from_nchw(torch.randn(N, cin, H, H, generator=g).to(d))
o = engine.Act(N, ho, wo, cout, d)
wp, bp = engine.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.randn(cout, generator=g), d)
c = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, 0, [(a.t, o.t, None, H, H, ho, wo)], N, shared_gpu=not excl)
print("kernel", c.kernel_name(), "splits", L.dafne_conv2d_wr_splits(ctypes.byref(c.prm), c.segs), "ws", L.dafne_conv2d_wr_workspace_bytes(ctypes.byref(c.prm), c.segs), flush=True)
if which == "wr":
    ws = engine.WrWorkspace(d)
    c = engine.WrCall(c, engine.pack_conv_frag(wp), ws)
for i in range(3):
    c(_lib.current_stream())
    torch.cuda.synchronize()
    print("ok", i, float(o.t.float().abs().sum()), flush=True)
