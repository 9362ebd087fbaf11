# conv_bneck vs the three separate launches: where do Y / Z differ?  usage: bneck_dbg.py [N H W]
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 8, 32)
g = torch.Generator().manual_seed(5)
w2p, b2p = engine.pack_conv(torch.randn(256, 256, 3, 3, generator=g) / 48, torch.randn(256, generator=g), d)
w3p, b3p = engine.pack_conv(torch.randn(1024, 256, 1, 1, generator=g) / 16, torch.randn(1024, generator=g), d)
w1p, b1p = engine.pack_conv(torch.randn(256, 1024, 1, 1, generator=g) / 32, torch.randn(256, generator=g), d)
wf = engine.pack_bneck(w2p, w3p, w1p)
nscr = L.dafne_bottleneck_body_scratch_bytes()
scr = torch.zeros(nscr, dtype=torch.uint8, device=d)
st = _lib.current_stream()
ua = engine.Act.from_nchw(torch.relu(torch.randn(N, 256, H, W, generator=g)).to(d))
xa = engine.Act.from_nchw(torch.relu(torch.randn(N, 1024, H, W, generator=g)).to(d))
t, y, z = engine.Act(N, H, W, 256, d), engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)
yf, zf = engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)
engine.ConvCall(w2p, b2p, 256, 256, 3, 1, 1, engine.F_RELU, [(ua.t, t.t, None, H, W, H, W)], N)(st)
engine.ConvCall(w3p, b3p, 256, 1024, 1, 1, 0, engine.F_RELU | engine.F_RES, [(t.t, y.t, xa.t, H, W, H, W)], N)(st)
engine.ConvCall(w1p, b1p, 1024, 256, 1, 1, 0, engine.F_RELU, [(y.t, z.t, None, H, W, H, W)], N)(st)
for rep in range(3):
    yf.t.zero_(); zf.t.zero_()
    _lib.check(L.dafne_bottleneck_body_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p), _lib.ptr(b1p), N, H, W, _lib.ptr(yf.t), _lib.ptr(zf.t), _lib.ptr(scr), nscr, st), "bneck")
    torch.cuda.synchronize()
    for name, a, b in (("Y", y.t, yf.t), ("Z", z.t, zf.t)):
        a = a.float().cpu(); b = b.float().cpu()          # [N, H+2, W+2, C]
        bad = (a != b)
        print("rep %d %s: %d of %d differ, max |d| %.4g" % (rep, name, int(bad.sum()), bad.numel(), float((a - b).abs().max())))
        if bad.any():
            ch = bad.any(dim=0).any(dim=0).any(dim=0)
            print("   channels:", [i for i in range(ch.numel()) if ch[i]][:64], "... (%d)" % int(ch.sum()))
            px = bad.any(dim=3)[0]
            rows = [i for i in range(px.shape[0]) if px[i].any()]
            cols = [i for i in range(px.shape[1]) if px[:, i].any()]
            print("   rows:", rows[:40], " cols:", cols[:40])
            idx = bad.nonzero()[:6]
            for i in idx:
                i = tuple(int(v) for v in i)
                print("   ", i, "want", float(a[i]), "got", float(b[i]))
