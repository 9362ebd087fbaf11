"""Host-side enqueue time of one pipelined step (python + ctypes + HIP launch calls), vs the GPU time per step."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8).to(dev)
step = lambda: model.detect_packed(batch, pipelined=True, splits=3)
for _ in range(5): step()
torch.cuda.synchronize()
ts = []
for rep in range(10):
    t0 = time.perf_counter(); step(); step(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append((t1 - t0) / 2)
print("host enqueue per step: min %.2f ms  median %.2f ms" % (1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2]))
dt = bench.time_steps(step, 30, 3, False)
print("GPU-paced step: %.2f ms" % (1e3 * dt / 30))
