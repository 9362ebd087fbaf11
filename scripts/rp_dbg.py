# conv3x3_rp (resident-patch kernel) vs the generic kernel on the same operands: where do they differ?  usage: rp_dbg.py N H W [cout]
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
d = torch.device("cuda", 0)
N, H, W = (int(a) for a in sys.argv[1:4])
cout = int(sys.argv[4]) if len(sys.argv) > 4 else 256
g = torch.Generator().manual_seed(1)
x = torch.randn(N, 256, H, W, generator=g).to(torch.bfloat16).float()
w = (torch.randn(cout, 256, 3, 3, generator=g) / 48.0).to(torch.bfloat16).float()
b = torch.randn(cout, generator=g) * 0.1
if os.environ.get("RP_DBG_IDENT"):                      # output = the input tap (kh, kw): garbage maps straight to patch (pixel, channel)
    kh, kw = (int(v) for v in os.environ["RP_DBG_IDENT"].split(","))
    w = torch.zeros(cout, 256, 3, 3); b = torch.zeros(cout)
    for c in range(min(cout, 256)): w[c, c, kh, kw] = 1.0
wp, bp = engine.pack_conv(w, b, d)
xi = engine.Act.from_nchw(x.to(d))
og, orp = engine.Act(N, H, W, cout, d), engine.Act(N, H, W, cout, d)
st = _lib.current_stream()
engine.ConvCall(wp, bp, 256, cout, 3, 1, 1, 0, [(xi.t, og.t, None, H, W, H, W)], N)(st)
c = engine.ConvCall(wp, bp, 256, cout, 3, 1, 1, 0, [(xi.t, orp.t, None, H, W, H, W)], N, wfrag=engine.pack_conv3x3_frag(wp))
print("kernel", c.kernel_name(), "tiles", c.num_tiles())
for rep in range(2):
    orp.t.zero_(); c(st); torch.cuda.synchronize()
    a, r = og.t.float().cpu(), orp.t.float().cpu()
    bad = a != r
    print("rep %d: %d of %d differ, max |d| %.4g" % (rep, int(bad.sum()), bad.numel(), float((a - r).abs().max())))
    if bad.any():
        for n in range(N):
            bn = bad[n]
            rows = [i for i in range(bn.shape[0]) if bn[i].any()]
            cols = [i for i in range(bn.shape[1]) if bn[:, i].any()]
            chs = [i for i in range(bn.shape[2]) if bn[:, :, i].any()]
            if rows:
                print("  img %d rows %s cols %s..%s (%d) channels %d (first %s)" % (n, rows[:24], cols[:3], cols[-3:], len(cols), len(chs), chs[:8]))
        i = tuple(int(v) for v in bad.nonzero()[0])
        print("  first", i, "want", float(a[i]), "got", float(r[i]))
        if os.environ.get("RP_DBG_IDENT"):
            bn = bad[0]
            pix = bn.any(dim=2).nonzero().tolist()
            print("  bad (row, col) haloed:", pix[:40])
            r0, c0 = pix[0]
            print("  channels bad at first pixel:", bn[r0, c0].nonzero().flatten().tolist())
