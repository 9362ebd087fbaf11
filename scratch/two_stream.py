import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev)
W = model._weights()
def run(nsplit, steps=20):
    per = 8 // nsplit
    plans = [engine.DensePlan(W, per, 1024, 1024, 101, 15, dev) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    for p in plans: p.stem_in.normal_()
    def step():
        for p, s in zip(plans, streams):
            with torch.cuda.stream(s):
                p.run(_lib.current_stream())
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print("dense only: split %d x batch %d: %.3f ms per 8 images -> %.1f img/s" % (nsplit, per, dt * 1e3, 8 / dt))
run(1); run(2); run(4); run(1)
