"""Beside torch.matmul on three streams: which side-stream kernels lose reproducibility?  sort_quad_kernel (this repo), and torch's own
compare / select / sort kernels on the same data."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import postprocess as pp
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 800
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
quads = (torch.rand(7500, 8, generator=g) * 600).cuda()
big = torch.rand(1 << 20, generator=g).cuda(); big2 = torch.rand(1 << 20, generator=g).cuda()
side = torch.cuda.Stream()
cs = [torch.cuda.Stream(priority=-1) for _ in range(3)]
A = [torch.randn(2048, 2048, device=d).bfloat16() for _ in range(3)]
B = [torch.randn(2048, 2048, device=d).bfloat16() for _ in range(3)]
ops = {
    "sort_quad_kernel": lambda: pp.sort_quadrilateral(quads),
    "torch.where(a > b, a, b)": lambda: torch.where(big > big2, big, big2),
    "torch cross-product select": lambda: torch.where((quads[:, 0] * quads[:, 3] - quads[:, 1] * quads[:, 2]) * (quads[:, 4] * quads[:, 7] - quads[:, 5] * quads[:, 6]) < 0, quads[:, 0], quads[:, 4]),
    "torch.sort": lambda: torch.sort(big[: 1 << 16]).values,
    "torch.min(dim=1).indices": lambda: quads.view(-1, 4, 2)[:, :, 0].min(dim=1).indices,
}
refs = {k: f().clone() for k, f in ops.items()}; torch.cuda.synchronize()
bad = {k: 0 for k in ops}; n = 0
for it in range(steps):
    for k in range(3):
        with torch.cuda.stream(cs[k]):
            for _ in range(6): A[k] @ B[k]
    outs = []
    with torch.cuda.stream(side):
        for _ in range(5):
            outs.append({k: f() for k, f in ops.items()})
    if it % 8 == 7 or it == steps - 1:
        torch.cuda.synchronize()
    # compare lazily at sync points
    if not hasattr(sys.modules[__name__], "_pend"): _pend = []
    _pend.extend(outs)
    if it % 8 == 7 or it == steps - 1:
        for o in _pend:
            n += 1
            for k, v in o.items():
                if not torch.equal(v, refs[k]): bad[k] += 1
        _pend = []
print("%d rounds beside matmul on three streams; launches that differ from the idle result:" % n)
for k, v in bad.items(): print("   %-32s %d" % (k, v))
