import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (3, 128, 128)
g = torch.Generator().manual_seed(5)
BF = torch.bfloat16
w3p, b3p = engine.pack_conv((torch.randn(512, 128, 1, 1, generator=g) / 11).to(BF).float(), torch.randn(512, generator=g) * 0.1, d)
w1p, b1p = engine.pack_conv((torch.randn(128, 512, 1, 1, generator=g) / 22).to(BF).float(), torch.randn(128, generator=g) * 0.1, d)
wf = engine.pack_b2b_mid(w3p, w1p)
st = _lib.current_stream()
ta = engine.Act.from_nchw(torch.randn(N, 128, H, W, generator=g).to(BF).float().to(d))
xa = engine.Act.from_nchw(torch.randn(N, 512, H, W, generator=g).to(BF).float().to(d))
yu, zu = engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
yf, zf = engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)
engine.ConvCall(w3p, b3p, 128, 512, 1, 1, 0, engine.F_RELU | engine.F_RES, [(ta.t, yu.t, xa.t, H, W, H, W)], N)(st)
engine.ConvCall(w1p, b1p, 512, 128, 1, 1, 0, engine.F_RELU, [(yu.t, zu.t, None, H, W, H, W)], N)(st)
_lib.check(L.dafne_bottleneck_tail_head_mid_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p), _lib.ptr(b1p), N, H, W, _lib.ptr(yf.t), _lib.ptr(zf.t), st), "n")
torch.cuda.synchronize()
for name, a, b, C in (("Y", yf, yu, 512), ("Z", zf, zu, 128)):
    A = a.t[:, 1:-1, 1:-1].reshape(N, H * W, C); B = b.t[:, 1:-1, 1:-1].reshape(N, H * W, C)
    bad = (A != B)
    print(name, "mismatch fraction", float(bad.float().mean()))
    if bad.any():
        tiles = bad.reshape(N, -1, 64, C) if (H * W) % 64 == 0 else None
        if tiles is not None:
            per_tile = tiles.float().mean((2, 3))          # [N, tiles_per_img]
            print("  per image:", per_tile.mean(1).tolist())
            print("  bad tiles (img, tile):", [(int(i), int(j)) for i, j in torch.nonzero(per_tile > 0)[:12]])
            i, j = [int(v) for v in torch.nonzero(per_tile > 0)[0]]
            bt = tiles[i, j]
            print("  first bad tile: bad per 64-ch slab", bt.reshape(64, C // 64, 64).float().mean((0, 2)).tolist(), "bad per px quarter", bt.reshape(4, 16, C).float().mean((1, 2)).tolist())
