import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd import engine, _lib
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(depth, dev)
batch = torch.randint(0, 256, (B, 3, 1024, 1024), dtype=torch.uint8).to(dev)
model.detect_packed(batch); torch.cuda.synchronize()
plan = model.plan(B, 1024, 1024)
stream = _lib.current_stream()
acc = {}
for rep in range(3):
    evs = []
    for i, c in enumerate(plan.calls):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); c(stream); b.record(); evs.append((i, c, a, b))
    torch.cuda.synchronize()
    for i, c, a, b in evs:
        acc.setdefault(i, []).append(a.elapsed_time(b))
tot = 0
groups = {}
for i, c in enumerate(plan.calls):
    ms = min(acc[i]); tot += ms
    if isinstance(c, engine.ConvCall):
        kn = c.kernel_name()
        p = c.prm
        hw = [(c.segs[s].Hout, c.segs[s].Wout) for s in range(p.n_segs)]
        key = "conv %dx%d s%d cin%-4d cout%-4d %s %s" % (p.KH, p.KW, p.stride, p.Cin, p.Cout, hw[0] if len(hw) == 1 else "5lvl", kn)
        g = groups.setdefault(key, [0, 0.0, 0.0]); g[0] += 1; g[1] += ms; g[2] += c.flops
    else:
        g = groups.setdefault(getattr(c, "name", None) or c.kernel_name() + " (pair)", [0, 0.0, 0.0]); g[0] += 1; g[1] += ms; g[2] += getattr(c, "flops", 0)
print("batch", B, "total ms (event sum)", tot, "per image", tot / B)
for k, (n, ms, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print("%-80s n=%3d  %.3f ms  %7.1f TF  %5.1f%%" % (k, n, ms, fl / (ms * 1e-3) / 1e12 if fl else 0, 100 * ms / tot) + "  %.4f ms/img" % (ms / B))
