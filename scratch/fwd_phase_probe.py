"""Where a synchronous model(batch) call spends its time: pack, enqueue, device wait, Instances -- host clock with a device
synchronisation behind every phase (so the phases do not overlap: the sum is above the real call)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd import postprocess as pp
d = torch.device("cuda", 0)
torch.set_num_threads(16)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
inputs = [{"image": batch[k], "height": 1024, "width": 1024} for k in range(8)]
for _ in range(6): m(inputs)
torch.cuda.synchronize()
acc = [0.0] * 5
N = 24
for _ in range(N):
    t = [time.perf_counter()]
    b, valid, out_hw = m._pack_inputs(inputs); torch.cuda.synchronize(); t.append(time.perf_counter())
    rows, counts = m.detect_packed(b, valid_hw=valid, out_hw=out_hw, pipelined=True, splits=2, even=True); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    insts = pp.rows_to_instances(rows, counts, out_hw); t.append(time.perf_counter())
    for i in range(4): acc[i] += t[i + 1] - t[i]
t0 = time.perf_counter()
for _ in range(N): m(inputs)
torch.cuda.synchronize()
whole = (time.perf_counter() - t0) / N
print("pack %.2f ms | enqueue %.2f | device wait %.2f | Instances %.2f | sum %.2f ; model(inputs) as is: %.2f ms"
      % tuple([1e3 * a / N for a in acc[:4]] + [1e3 * sum(acc[:4]) / N, 1e3 * whole]))
