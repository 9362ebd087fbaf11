"""The headline layout (batch 8, 1024^2, three unequal sub-batches, deferred post-process, graph replay), the same batch every step:
every step's detections against the first step's.  usage: race_probe8.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
ref = None; bad = []; pend = []
for i in range(steps):
    res = m.detect_packed(batch, pipelined=True, splits=3, defer=True)
    if res is not None: pend.append((i - 1, res))
    if len(pend) >= 16 or i == steps - 1:
        torch.cuda.synchronize()
        for j, (rows, counts) in pend:
            if ref is None:
                ref = (rows.clone(), counts.clone()); continue
            if not torch.equal(counts, ref[1]):
                bad.append((j, "counts")); continue
            for im in range(8):
                k = int(counts[im])
                if not torch.equal(rows[im, :k], ref[0][im, :k]):
                    bad.append((j, "image %d: %d elements differ" % (im, int((rows[im, :k] != ref[0][im, :k]).sum())))); break
        pend = []
print("%d steps of the headline layout: %d differ from the first" % (steps, len(bad)), bad[:8])
