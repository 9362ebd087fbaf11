"""cProfile of the host side of the timed step (graphs on): where the 0.36 ms per step goes."""
import os, sys, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
b = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
f = lambda: m.detect_packed(b, pipelined=True, splits=2)
for _ in range(8): f()
torch.cuda.synchronize()
pr = cProfile.Profile()
N = 200
pr.enable()
for _ in range(N): f()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(22)
out = s.getvalue()
print("\n".join(l[:150] for l in out.splitlines()[:45]))
