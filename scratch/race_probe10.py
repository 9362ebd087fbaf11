"""sort_quad_kernel (pure elementwise) on a fixed input beside the HEADLINE layout's load (batch 8, 1024^2, three sub-batch streams,
deferred post-process): K launches per step."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd import postprocess as pp
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(os.environ.get("K", "20"))
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
quads = (torch.rand(7500, 8, generator=g) * 600).cuda()
side = torch.cuda.Stream()
ref = pp.sort_quadrilateral(quads).clone(); torch.cuda.synchronize()
bad = 0; n = 0; pend = []
for it in range(steps):
    m.detect_packed(batch, pipelined=True, splits=3, defer=True)
    with torch.cuda.stream(side):
        for _ in range(K):
            pend.append(pp.sort_quadrilateral(quads))
    if len(pend) >= 16 * K or it == steps - 1:
        torch.cuda.synchronize()
        for a in pend:
            n += 1
            if not torch.equal(a, ref): bad += 1
        pend = []
print("%d sort_quad launches beside %d headline steps: %d differ" % (n, steps, bad))
