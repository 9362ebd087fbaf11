"""conv3x3_rp_kernel<GNIN> vs the patch kernel's GN_INPUT form on the same raw map + statistics: mismatch pattern by pixel / channel."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
d = torch.device("cuda", 0)
C, N = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 2
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(40, 72)]
g = torch.Generator().manual_seed(3)
w = (torch.randn(C, C, 3, 3, generator=g) / 48).to(torch.bfloat16).float()
b = torch.randn(C, generator=g) * 0.1
wp, bp = engine.pack_conv(w, b, d)
wf = engine.pack_conv3x3_frag(wp)
gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(d); beta = (0.3 * torch.randn(C, generator=g)).to(d)
ins = [engine.Act.from_nchw(torch.randn(N, C, h, ww, generator=g).to(torch.bfloat16).float().to(d)) for h, ww in sizes]
stats = torch.zeros(len(sizes), N, C // 8, 2, device=d)
stats[..., 0] = torch.randn(len(sizes), N, C // 8, generator=g).to(d) * 0.2
stats[..., 1] = 1 + 0.3 * torch.rand(len(sizes), N, C // 8, generator=g).to(d)
o1 = [engine.Act(N, h, ww, C, d) for h, ww in sizes]; o2 = [engine.Act(N, h, ww, C, d) for h, ww in sizes]
st = _lib.current_stream()
engine.ConvCall(wp, bp, C, C, 3, 1, 1, engine.F_GNIN, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, o1)], N, gn_in=(stats, gamma, beta))(st)
engine.ConvCall(wp, bp, C, C, 3, 1, 1, engine.F_GNIN, [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, o2)], N, gn_in=(stats, gamma, beta), wfrag=wf)(st)
torch.cuda.synchronize()
for k, (a, r) in enumerate(zip(o1, o2)):
    A, Rr = a.t[:, 1:-1, 1:-1].float(), r.t[:, 1:-1, 1:-1].float()
    bad = (A != Rr)
    print("level", k, sizes[k], "mismatching elements", int(bad.sum()), "of", bad.numel(), "max diff", float((A - Rr).abs().max()))
    if bad.any():
        px = bad.any(-1)                          # [N, H, W]
        print("  images:", px.flatten(1).any(1).tolist())
        print("  rows with mismatches:", torch.nonzero(px.any(0).any(1)).flatten().tolist()[:40])
        print("  cols with mismatches:", torch.nonzero(px.any(0).any(0)).flatten().tolist()[:80])
        print("  channels with mismatches (count):", int(bad.any(0).any(0).any(0).sum()))
        print("  fraction of channels per bad pixel:", float(bad[px].float().mean()))
