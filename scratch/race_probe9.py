"""sort_quad_kernel alone (an elementwise kernel: 8 floats in, sort_quadrilateral, 8 floats out) on a fixed input while the three compute
streams run single-image detector calls: is a PURE function of its input reproducible under that load?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_inference_loop as T
from dafne_amd import postprocess as pp
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
load = int(os.environ.get("LOAD", "3"))
cfg, m = T._gpu_model()
g = torch.Generator().manual_seed(5)
bg = torch.randint(0, 256, (1, 3, 448, 576), generator=g, dtype=torch.uint8).cuda()
quads = (torch.rand(7500, 8, generator=g) * 600).cuda()
side = torch.cuda.Stream(priority=int(os.environ.get("SIDE_PRIO", "0")))
ref = pp.sort_quadrilateral(quads).clone(); torch.cuda.synchronize()
# an even simpler pure kernel: torch's own elementwise op
ref2 = (quads * 1.5 + 2.0).clone()
bad = bad2 = 0; rot = 0; pend = []
for it in range(iters):
    for _ in range(load):
        m.detect_packed(bg, pipelined=True, splits=1, defer=True, stream_offset=rot); rot = (rot + 1) % int(os.environ.get("NSTREAMS", "3"))
    with torch.cuda.stream(side):
        pend.append((pp.sort_quadrilateral(quads), quads * 1.5 + 2.0))
    if len(pend) == 16 or it == iters - 1:
        torch.cuda.synchronize()
        for a, b in pend:
            if not torch.equal(a, ref):
                bad += 1
                if bad <= 3:
                    rows = (a != ref).any(1).nonzero().flatten().tolist()
                    print("  sort_quad rows differing: %s" % rows[:40])
            if not torch.equal(b, ref2): bad2 += 1
        pend = []
print("%d runs beside %d background calls each: sort_quad_kernel differs %d times, torch elementwise %d times" % (iters, load, bad, bad2))
