"""Grouped TTA (3 images per group) vs one image per call at the released config's sizes (1024^2 tiles, 27 views): equal?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.modeling.tta import OneStageRCNNWithTTA
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0, cfgname="dota-1.5_r101.yaml", cls_prior=-1.5)
g = torch.Generator().manual_seed(0)
imgs = torch.randint(0, 256, (3, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
inp = [{"image": imgs[k], "height": 1024, "width": 1024} for k in range(3)]
one = OneStageRCNNWithTTA(cfg, m, images_per_group=1)
grp = OneStageRCNNWithTTA(cfg, m, images_per_group=3)
a = one(inp); b = grp(inp)
torch.cuda.synchronize()
for k in range(3):
    x, y = a[k]["instances"], b[k]["instances"]
    same = len(x) == len(y) and torch.equal(x.pred_corners, y.pred_corners) and torch.equal(x.scores, y.scores)
    print("image %d: merged result identical %s (%d / %d detections)" % (k, same, len(x), len(y)))
# per view: chunk of 3 vs chunk of 9, which sizes differ and which launches
per = [grp._get_augmented_inputs(x) for x in inp]
for s in range(9):
    v3 = per[0][0][3 * s:3 * s + 3]
    v9 = [v for i in range(3) for v in per[i][0][3 * s:3 * s + 3]]
    (r3, c3, _), = grp._views_packed(v3)
    (r9, c9, _), = grp._views_packed(v9, [9])
    same = all(int(c3[i]) == int(c9[i]) and torch.equal(r3[i, :int(c3[i])], r9[i, :int(c3[i])]) for i in range(3))
    h, w = int(v3[0]["image"].shape[1]), int(v3[0]["image"].shape[2])
    hn, wn = (h + 31) // 32 * 32, (w + 31) // 32 * 32
    k3 = [x.kernel_name() for x in m._pipe[(3, hn, wn, 1)]["plans"][0][0].calls] if (3, hn, wn, 1) in m._pipe else []
    k9 = [x.kernel_name() for x in m._pipe[(9, hn, wn, 1)]["plans"][0][0].calls] if (9, hn, wn, 1) in m._pipe else []
    from collections import Counter
    dd = (Counter(k3) - Counter(k9), Counter(k9) - Counter(k3))
    print("view size %dx%d: chunk 3 vs chunk 9 identical %s   kernels only in chunk-3 plan %s, only in chunk-9 plan %s" % (h, w, same, dict(dd[0]), dict(dd[1])))
