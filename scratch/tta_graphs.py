"""TTA (27 views) ms per image in steady state, HIP graphs on / off (DAFNE_HIP_GRAPHS)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from dafne_amd.modeling.tta import OneStageRCNNWithTTA
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev, cfgname="dota-1.5_r101.yaml", cls_prior=-1.5)
tta = OneStageRCNNWithTTA(cfg, model)
g = torch.Generator().manual_seed(0)
imgs = [torch.randint(0, 256, (3, 1024, 1024), generator=g, dtype=torch.uint8).to(dev) for _ in range(4)]
ts = []
for rep in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tta([{"image": imgs[rep % 4], "height": 1024, "width": 1024}])
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("graphs", os.environ.get("DAFNE_HIP_GRAPHS"), "ms per image:", " ".join("%.1f" % t for t in ts))
