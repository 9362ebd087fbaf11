"""Control: sort_quad_kernel (pure elementwise) on a fixed input beside VENDOR kernels only (torch.matmul / elementwise ops on three
streams -- none of this repo's kernels): does the corruption need our convolutions at all?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import postprocess as pp
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
K = int(os.environ.get("K", "20"))
kind = os.environ.get("BG", "matmul")
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
quads = (torch.rand(7500, 8, generator=g) * 600).cuda()
side = torch.cuda.Stream()
cs = [torch.cuda.Stream(priority=-1) for _ in range(3)]
A = [torch.randn(2048, 2048, device=d).bfloat16() for _ in range(3)]
B = [torch.randn(2048, 2048, device=d).bfloat16() for _ in range(3)]
X = [torch.randn(1 << 22, device=d) for _ in range(3)]
ref = pp.sort_quadrilateral(quads).clone(); torch.cuda.synchronize()
bad = 0; n = 0; pend = []
for it in range(steps):
    for k in range(3):
        with torch.cuda.stream(cs[k]):
            if kind == "matmul":
                for _ in range(6): A[k] @ B[k]
            elif kind == "small":
                for _ in range(40): torch.relu(X[k][: 1 << 16] * 1.01)
            else:
                for _ in range(6): torch.relu(X[k] * 1.01)
    with torch.cuda.stream(side):
        for _ in range(K):
            pend.append(pp.sort_quadrilateral(quads))
    if len(pend) >= 16 * K or it == steps - 1:
        torch.cuda.synchronize()
        for a in pend:
            n += 1
            if not torch.equal(a, ref): bad += 1
        pend = []
print("background %s: %d sort_quad launches, %d differ" % (kind, n, bad))
