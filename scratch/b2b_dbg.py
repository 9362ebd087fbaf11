import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, torch.nn.functional as F
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bfr = lambda t: t.to(torch.bfloat16).float()
g = torch.Generator().manual_seed(5)
t = bfr(torch.randn(N, 256, H, W, generator=g)); x = bfr(torch.randn(N, 1024, H, W, generator=g))
w3 = bfr(torch.randn(1024, 256, 1, 1, generator=g) / 16.0); b3 = torch.randn(1024, generator=g) * 0.2
w1 = bfr(torch.randn(256, 1024, 1, 1, generator=g) / 32.0); b1 = torch.randn(256, generator=g) * 0.2
st = _lib.current_stream()
ta, xa = engine.Act.from_nchw(t.to(d)), engine.Act.from_nchw(x.to(d))
w3p, b3p = engine.pack_conv(w3, b3, d); w1p, b1p = engine.pack_conv(w1, b1, d)
y_u, z_u = engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)
engine.ConvCall(w3p, b3p, 256, 1024, 1, 1, 0, engine.F_RELU | engine.F_RES, [(ta.t, y_u.t, xa.t, H, W, H, W)], N)(st)
engine.ConvCall(w1p, b1p, 1024, 256, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
wf = engine.pack_b2b(w3p, w1p)
y_f, z_f = engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)
_lib.check(L.dafne_bottleneck_tail_head_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p), _lib.ptr(b1p),
                                            N, H, W, _lib.ptr(y_f.t), _lib.ptr(z_f.t), st), "b2b")
torch.cuda.synchronize()
for name, a, b in (("Y", y_f, y_u), ("Z", z_f, z_u)):
    da = (a.t.float() - b.t.float())[:, 1:-1, 1:-1]          # [N,H,W,C]
    bad = da != 0
    print(name, "mismatch frac", bad.float().mean().item(), "max abs", da.abs().max().item(), "ref max", b.t.float().abs().max().item())
    if bad.any():
        C = da.shape[-1]
        print("  by 32-channel group:", [round(v, 2) for v in bad.reshape(-1, C // 32, 32).float().mean((0, 2)).tolist()])
        pxbad = bad.reshape(N, H * W, C).float().mean((0, 2))
        print("  by pixel (first 32):", [round(v, 2) for v in pxbad[:32].tolist()])
        print("  by pixel mod 32 :", [round(pxbad[k::32].mean().item(), 2) for k in range(min(32, H * W))])
        nz = bad.nonzero()[:5]
        for i in nz:
            n_, h_, w_, c_ = i.tolist()
            print("   at", i.tolist(), "fused", a.t[n_, h_ + 1, w_ + 1, c_].item(), "ref", b.t[n_, h_ + 1, w_ + 1, c_].item())
