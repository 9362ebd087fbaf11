import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
from test_gpu_model import build, dev
from dafne_amd.modeling.tta import OneStageRCNNWithTTA
cfg, m, P = build("dota-1.5_r101.yaml", seed=21)
cfg.TEST.AUG.MIN_SIZES = [96, 128, 160]; cfg.TEST.AUG.MAX_SIZE = 256
g = torch.Generator().manual_seed(12)
imgs = [torch.randint(0, 256, (3, 128, 160), generator=g, dtype=torch.uint8).to(dev()) for _ in range(3)]
inputs = [{"image": im, "height": 133, "width": 163} for im in imgs]
t = OneStageRCNNWithTTA(cfg, m)
per = [t._get_augmented_inputs(x) for x in inputs]
# per image, chunks of 3
alone = [t._views_packed(per[i][0]) for i in range(3)]
torch.cuda.synchronize()
alone = [[(r.clone(), c.clone()) for r, c, _ in a] for a in alone]
# grouped: run k of every image
flat = []
for k in range(3):
    for i in range(3):
        flat.extend(per[i][0][3 * k:3 * k + 3])
grp = t._views_packed(flat, [9, 9, 9])
torch.cuda.synchronize()
for k in range(3):
    rows, counts, _ = grp[k]
    for i in range(3):
        for v in range(3):
            ra, ca = alone[i][k]
            n1, n2 = int(ca[v]), int(counts[3 * i + v])
            a, b = ra[v, :n1].cpu().numpy(), rows[3 * i + v, :n2].cpu().numpy()
            same = n1 == n2 and np.array_equal(a, b)
            md = np.abs(a[:min(n1, n2)] - b[:min(n1, n2)]).max() if min(n1, n2) else 0
            print("size %d img %d view %d: n %d / %d identical %s maxdiff(first rows) %.4g" % (k, i, v, n1, n2, same, md))
# features: compare the head inputs? use detect on dense only
