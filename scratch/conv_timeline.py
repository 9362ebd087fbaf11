import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ["DAFNE_AMD_LIB"] = os.path.join(R, "scratch", "libdafne_timing.so")
import torch, numpy as np
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
def run(cin, cout, k, H, W, N=8, res=False, label=""):
    pad = 1 if k == 3 else 0
    a = engine.Act(N, H, W, cin, dev); a.t[:, 1:-1, 1:-1].normal_()
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
    o = engine.Act(N, H, W, cout, dev)
    r = engine.Act(N, H, W, cout, dev) if res else None
    f = (engine.F_RES | engine.F_RELU) if res else 0
    dbg = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
    c = engine.ConvCall(wp, bp, cin, cout, k, 1, pad, f, [(a.t, o.t, r.t if r is not None else None, H, W, H, W)], N, gn_partial=dbg.view(torch.float32))
    st = _lib.current_stream()
    for _ in range(3): c(st)
    torch.cuda.synchronize()
    nb = c.num_tiles() * max(1, cout // (256 if c.tile_pixels() == 256 and cout % 256 == 0 else 128))
    nw = 8 if c.tile_pixels() == 256 and cout >= 128 else 4
    print("lib", _lib.LIB_PATH, "nonzero", int((dbg != 0).sum()), "nb", nb, "nw", nw)
    t = dbg.cpu().numpy().reshape(-1, 8)[: nb * nw]
    t0 = t[:, 0].min()
    d = t[:, :5] - t0
    print(label, "blocks", nb, "waves/blk", nw, "(clock ticks = 100MHz? units raw)")
    print("  block start   : min %d  median %d  max %d" % (d[:, 0].min(), np.median(d[:, 0]), d[:, 0].max()))
    seg = np.diff(d, axis=1)
    names = ["setup+prologue loads", "K loop", "epilogue pass 0", "epilogue pass 1"]
    for i, nme in enumerate(names):
        print("  %-22s median %8.0f  p90 %8.0f" % (nme, np.median(seg[:, i]), np.percentile(seg[:, i], 90)))
    print("  block total median", np.median(d[:, 4] - d[:, 0]))
    rt0, rt1 = t[:, 5].astype(np.float64), t[:, 6].astype(np.float64)
    base = rt0.min()
    s_us, e_us = (rt0 - base) / 100.0, (rt1 - base) / 100.0
    print("  realtime: kernel span %.1f us; block dur median %.1f us; first-round starts <= %.1f us (p99 of first %d), second-round start median %.1f us"
          % (e_us.max(), np.median(e_us - s_us), np.percentile(np.sort(s_us)[:256 * nw], 99), 256, np.median(np.sort(s_us)[256 * nw:]) if len(s_us) > 256 * nw else -1))
    hist = np.histogram(s_us, bins=8)[0]
    print("  start-time histogram (8 bins over span):", hist.tolist())
run(256, 1024, 1, 64, 64, res=True, label="res4 conv3 (+res)")
run(256, 256, 3, 64, 64, label="res4 conv2 3x3 (128 tile)")
run(1024, 256, 1, 64, 64, label="res4 conv1 1x1 K=1024 (128 tile)")
run(256, 256, 3, 128, 128, label="head-like 3x3")
