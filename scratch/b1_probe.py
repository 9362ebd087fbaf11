"""Batch 1 at 1024^2 (the reference's own loop shape, tools/benchmark.py:117-145): ms per model([image]) call, per streamed call, and
the per-kernel isolated times of the batch-1 plan.  Environment toggles of the engine apply (run once per setting)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.evaluation.inference import inference_on_dataset
d = torch.device("cuda", 0)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 101
cfg, m, _ = bench.build_model(depth, d, seed=0)
g = torch.Generator().manual_seed(7)
img = torch.randint(0, 256, (3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
dd = [{"image": img, "height": 1024, "width": 1024, "image_id": 0}]
for _ in range(5): o = m(dd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): o = m(dd)
torch.cuda.synchronize()
ms_call = 1e3 * (time.perf_counter() - t0) / 200
loader = [dd] * 200
inference_on_dataset(m, loader[:8]); st = {}
inference_on_dataset(m, loader, None, st)
print("batch 1: model call %.3f ms, streamed loop %.3f ms per image" % (ms_call, 1e3 * st["seconds"] / 200))
if os.environ.get("B1_KERNELS", "1") == "1":
    iso = bench.conv_kernel_profile_isolated(m, img.unsqueeze(0))
    tot = 0.0
    for k, v in sorted(iso.items(), key=lambda kv: -kv[1]["ms_per_step"]):
        tot += v["ms_per_step"]
        print("   %-28s %3d launches %7.3f ms  %6.0f TF" % (k, v["launches"], v["ms_per_step"], v["tflops"]))
    print("   sum %.3f ms" % tot)
