import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
st = _lib.current_stream()
def run(cin, cout, hw, N, label):
    insts = []
    for _ in range(3):
        a = engine.Act(N, hw, hw, cin, dev); a.t[:, 1:-1, 1:-1].normal_()
        w = torch.randn(cout, cin, 1, 1) / cin ** 0.5
        wp, bp = engine.pack_conv(w, torch.zeros(cout), dev)
        o = engine.Act(N, hw, hw, cout, dev); r = engine.Act(N, hw, hw, cout, dev); r.t[:, 1:-1, 1:-1].normal_()
        c = engine.ConvCall(wp, bp, cin, cout, 1, 1, 0, engine.F_RES | engine.F_RELU, [(a.t, o.t, r.t, hw, hw, hw, hw)], N)
        insts.append((c, a, o, r, wp, bp))
    for c, *_ in insts: c(st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            for c, *_ in insts: c(st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 12)
    byt = (insts[0][1].t.numel() + 2 * insts[0][2].t.numel()) * 2
    print("%-34s %s : %7.1f us  %5.2f TB/s" % (label, insts[0][0].kernel_name(), best, byt / best / 1e6))
run(256, 1024, 64, 8, "256->1024 64x64 N=8 (2 KB rows)")
run(256, 512, 64, 16, "256->512 64x64 N=16 (1 KB rows)")
run(256, 256, 64, 32, "256->256 64x64 N=32 (512 B rows)")
run(256, 128, 64, 64, "256->128 64x64 N=64 (256 B rows)")
