"""Batch invariance at full size: image k's packed detections at batch 8 / 4 / 3 / 1 (1024^2 and 512x640), R101."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(3)
for (h, w) in ((1024, 1024), (512, 640)):
    b8 = torch.randint(0, 256, (8, 3, h, w), generator=g, dtype=torch.uint8).to(d)
    r8, c8 = m.detect_packed(b8); torch.cuda.synchronize(); r8, c8 = r8.clone(), c8.clone()
    for lo, hi in ((0, 4), (4, 8), (2, 5), (0, 1), (7, 8), (0, 2)):
        r, c = m.detect_packed(b8[lo:hi].contiguous()); torch.cuda.synchronize()
        same = all(int(c[i]) == int(c8[lo + i]) and torch.equal(r[i, :int(c[i])], r8[lo + i, :int(c[i])]) for i in range(hi - lo))
        print("%dx%d images [%d,%d) of the batch of 8: identical %s  (counts %s vs %s)" % (h, w, lo, hi, same, c.tolist()[:3], c8[lo:hi].tolist()[:3]), flush=True)
        if not same:
            for name, p8, pk in (("b8", m.plan(8, h, w), m.plan(hi - lo, h, w)),):
                k8 = [x.kernel_name() for x in p8.calls]; kk = [x.kernel_name() for x in pk.calls]
                diff = [(i, a, b) for i, (a, b) in enumerate(zip(k8, kk)) if a != b]
                print("   launch lists differ at", diff[:8], len(k8), len(kk))
