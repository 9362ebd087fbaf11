"""model(batched_inputs) in a loop with one host sync per call (detectron2's own loop shape at batch 8), device tiles:
images/s.  Engine toggles through the environment (DAFNE_PIPELINE_SPLITS, DAFNE_RP_PAIR_SHARED, DAFNE_SPLIT_SIZES)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
loader = [[{"image": batch[k], "height": 1024, "width": 1024} for k in range(8)] for j in range(30)]
for b in loader[:6]: m(b)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for b in loader: m(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("splits %s pair_shared %s sizes %s: %.1f img/s (%.2f ms per call)" % (cfg.ENGINE.PIPELINE_SPLITS, os.environ.get("DAFNE_RP_PAIR_SHARED", "1"),
          os.environ.get("DAFNE_SPLIT_SIZES", "-"), 240 / dt, 1e3 * dt / 30), flush=True)
