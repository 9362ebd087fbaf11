"""Timing experiment: the previous step's decode + NMS enqueued when the current step reaches its head towers (eager launches)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ["DAFNE_HIP_GRAPHS"] = "0"
import torch
import bench
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
b = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
f = lambda: m.detect_packed(b, pipelined=True, splits=2)
for _ in range(8): f()
torch.cuda.synchronize()
def rate(n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return 8 * n / (time.perf_counter() - t0)
for rep in range(2):
    os.environ.pop("DAFNE_EXP_DEFER_POST", None)
    a = rate()
    line = "eager %.1f img/s |" % a
    for mode in (0, 1, 2, 3, 4):
        os.environ["DAFNE_EXP_DEFER_POST"] = "1"
        os.environ["DAFNE_EXP_DEFER_AT"] = str(mode)
        f(); f(); torch.cuda.synchronize()
        c = rate()
        line += " at%d %.1f (%+.2f %%)" % (mode, c, 100 * (c / a - 1))
    print(line, flush=True)
