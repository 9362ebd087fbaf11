"""How far ahead of the GPU does the host run in the pipelined step?  (no profiler attached)
Per step: host time at entry / exit of detect_packed, GPU time (event) when the step's convolutions end on each
compute stream and when its post-process ends on the side stream.  If host_exit(i+1) < gpu_conv_end(i) the host is
ahead and the step boundary cannot be a host gap."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg, model, sd = bench.build_model(101, dev, seed=0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8).to(dev)
for _ in range(5):
    model.detect_packed(batch, pipelined=True, splits=3)
torch.cuda.synchronize()
st = next(iter(model._pipe.values()))
cs = st["cs"]
e0 = torch.cuda.Event(enable_timing=True)
e0.record(cs[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
e0 = torch.cuda.Event(enable_timing=True)
e0.record(cs[0])
rec = []
N = 16
for i in range(N):
    a = time.perf_counter()
    model.detect_packed(batch, pipelined=True, splits=3)
    b = time.perf_counter()
    evs = []
    for s in cs:
        e = torch.cuda.Event(enable_timing=True)
        e.record(s)
        evs.append(e)
    d = torch.cuda.Event(enable_timing=True)
    d.record(model.side_stream)
    rec.append((a - t0, b - t0, evs, d))
torch.cuda.synchronize()
print("step  host_in  host_out | conv_end per stream            | post_end   (ms since start)")
for i, (a, b, evs, d) in enumerate(rec):
    ce = [e0.elapsed_time(e) for e in evs]
    print("%3d  %7.2f  %7.2f | %7.2f %7.2f %7.2f | %7.2f" % (i, a * 1e3, b * 1e3, ce[0], ce[1], ce[2], e0.elapsed_time(d)))
