"""Soak: the streamed loop over many batches (device tiles, then host tiles, then mixed sizes): memory growth, rate drift."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.evaluation.inference import DafneEvaluator, inference_on_dataset
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
b = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8)
bd = b.to(d)
def loader(src, n):
    return [[{"image": src[k], "height": 1024, "width": 1024, "image_id": j * 8 + k} for k in range(8)] for j in range(n)]
ev = DafneEvaluator("soak", cfg, distributed=False)
inference_on_dataset(m, loader(bd, 6), ev)
for name, src, n in (("device tiles", bd, 400), ("host tiles", b, 400), ("device tiles", bd, 400)):
    torch.cuda.synchronize()
    a0, r0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    rates = []
    for chunk in range(4):
        st = {}
        inference_on_dataset(m, loader(src, n // 4), ev, st)
        rates.append(st["images_per_sec"])
    torch.cuda.synchronize()
    print("%-13s %d batches: rates %s img/s; allocated %+.1f MB, reserved %+.1f MB (now %.2f GB reserved)"
          % (name, n, ["%.0f" % r for r in rates], (torch.cuda.memory_allocated() - a0) / 1e6, (torch.cuda.memory_reserved() - r0) / 1e6,
             torch.cuda.memory_reserved() / 1e9), flush=True)
# many shapes: plan cache growth
sizes = [(512 + 32 * (i % 9), 640 + 32 * (i % 7)) for i in range(40)]
torch.cuda.synchronize(); r0 = torch.cuda.memory_reserved(); t0 = time.perf_counter()
for h, w in sizes:
    x = torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8).to(d)
    m([{"image": x[k], "height": h, "width": w} for k in range(2)])
torch.cuda.synchronize()
print("40 calls over %d distinct shapes: %.1f s, reserved %+.2f GB" % (len(set(sizes)), time.perf_counter() - t0, (torch.cuda.memory_reserved() - r0) / 1e9))
