"""Is the timed layout host-bound?  K pipelined steps: host time until the last enqueue returns vs time until the GPU is done.
usage: host_bound.py [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
for mode, kw in (("pipelined x3", dict(pipelined=True, splits=3)), ("pipelined x3 graphs", dict(pipelined=True, splits=3)),
                 ("pipelined x1", dict(pipelined=True, splits=1)), ("serial", dict())):
    m.use_graphs = "graphs" in mode
    f = lambda: m.detect_packed(batch, **kw)
    for _ in range(5): f()
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        per = []
        for _ in range(steps):
            a = time.perf_counter(); f(); per.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%-13s host enqueue %.3f ms/step (min %.3f)  GPU done %.3f ms/step  -> %.0f img/s; GPU tail after last enqueue %.2f ms"
              % (mode, 1e3 * (t1 - t0) / steps, 1e3 * min(per), 1e3 * (t2 - t0) / steps, 8 * steps / (t2 - t0), 1e3 * (t2 - t1)), flush=True)
