import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
N, H, W, head = [int(v) for v in sys.argv[1:5]]
d = torch.device("cuda", 0); L = _lib.load()
g = torch.Generator().manual_seed(1)
bf = lambda x: x.to(torch.bfloat16).float()
u = bf(torch.relu(torch.randn(N, 128, H, W, generator=g))); x = bf(torch.relu(torch.randn(N, 512, H, W, generator=g)))
w2p, b2p = engine.pack_conv(bf(torch.randn(128, 128, 3, 3, generator=g) / 34), torch.randn(128, generator=g) * .2, d)
w3p, b3p = engine.pack_conv(bf(torch.randn(512, 128, 1, 1, generator=g) / 11), torch.randn(512, generator=g) * .2, d)
w1p, b1p = engine.pack_conv(bf(torch.randn(128, 512, 1, 1, generator=g) / 22), torch.randn(128, generator=g) * .2, d)
ua, xa = engine.Act.from_nchw(u.to(d)), engine.Act.from_nchw(x.to(d))
st = _lib.current_stream()
t_u, y_u, z_u = engine.Act(N, H, W, 128, d), engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
engine.ConvCall(w2p, b2p, 128, 128, 3, 1, 1, engine.F_RELU, [(ua.t, t_u.t, None, H, W, H, W)], N)(st)
engine.ConvCall(w3p, b3p, 128, 512, 1, 1, 0, engine.F_RELU | engine.F_RES, [(t_u.t, y_u.t, xa.t, H, W, H, W)], N)(st)
engine.ConvCall(w1p, b1p, 512, 128, 1, 1, 0, engine.F_RELU, [(y_u.t, z_u.t, None, H, W, H, W)], N)(st)
torch.cuda.synchronize(); print("reference launches ok", flush=True)
wf = engine.pack_blk_mid(w2p, w3p, w1p if head else None)
scr = torch.empty(L.dafne_bottleneck_block_mid_scratch_bytes(), dtype=torch.uint8, device=d)
y_f, z_f = engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
rc = L.dafne_bottleneck_block_mid_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p), _lib.ptr(b1p) if head else None,
                                      N, H, W, _lib.ptr(y_f.t), _lib.ptr(z_f.t) if head else None, _lib.ptr(scr), scr.numel(), st)
print("rc", rc, L.dafne_last_error(), flush=True)
torch.cuda.synchronize(); print("kernel done", flush=True)
dy = (y_f.t.float() - y_u.t.float()).abs(); dz = (z_f.t.float() - z_u.t.float()).abs()
print("Y equal", bool(torch.equal(y_f.t, y_u.t)), "max diff", float(dy.max()), "n diff", int((dy > 0).sum()), "| Z equal", bool(torch.equal(z_f.t, z_u.t)), float(dz.max()), int((dz > 0).sum()))
if not torch.equal(y_f.t, y_u.t):
    idx = (dy > 0).nonzero()[:10]; print(idx.tolist())
    ch = (dy > 0).any(dim=0).any(dim=0).any(dim=0).nonzero().flatten(); print("channels differing:", ch.tolist()[:40], len(ch))
    rows = (dy > 0).any(dim=3).any(dim=0).any(dim=1).nonzero().flatten(); print("rows (haloed):", rows.tolist()[:40])
