"""Config 4 (27 TTA views per image): one image per call vs groups of 2 / 3 images whose same-size views share a detector call."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.modeling.tta import OneStageRCNNWithTTA
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0, cfgname="dota-1.5_r101.yaml", cls_prior=-1.5)
g = torch.Generator().manual_seed(0)
imgs = torch.randint(0, 256, (12, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
inp = [{"image": imgs[k], "height": 1024, "width": 1024} for k in range(12)]
for G in (1, 3):
    tta = OneStageRCNNWithTTA(cfg, m, images_per_group=G)
    for _ in range(3):
        tta(inp[:G])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        out = tta(inp)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 12)
    print("images per group %d: %.2f ms per image (%s detections)" % (G, 1e3 * best, [len(o["instances"]) for o in out][:3]), flush=True)
