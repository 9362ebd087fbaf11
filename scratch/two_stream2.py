import sys, os, time, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd import engine, _lib
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev)
W = model._weights()
def run(nsplit, interleave, steps=20):
    per = 8 // nsplit
    plans = [engine.DensePlan(W, per, 1024, 1024, 101, 15, dev) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    sp = [ctypes.c_void_p(s.cuda_stream) for s in streams]
    for p in plans: p.stem_in.normal_()
    ncalls = len(plans[0].calls)
    def step():
        if interleave:
            for i in range(ncalls):
                for p, s in zip(plans, sp):
                    p.calls[i](s)
        else:
            for p, s in zip(plans, sp):
                for c in p.calls: c(s)
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    th = 0
    for _ in range(steps):
        t0 = time.perf_counter(); step(); th += time.perf_counter() - t0
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print("dense only: split %d x batch %d interleave=%d: %.3f ms per 8 images -> %.1f img/s (host enqueue %.2f ms/step)" % (nsplit, per, interleave, dt * 1e3, 8 / dt, th / steps * 1e3))
run(1, 0); run(2, 0); run(2, 1); run(4, 0); run(4, 1); run(8, 1)
