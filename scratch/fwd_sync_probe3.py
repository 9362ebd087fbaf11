"""Order effect: the three-sub-batch loop first (as bench.py), then the synchronous forward; or the other way round (ORDER=sync)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
torch.set_num_threads(16)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
def sync_loop(tag):
    loader = [[{"image": batch[k], "height": 1024, "width": 1024} for k in range(8)] for j in range(24)]
    for b in loader[:4]: m(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in loader: m(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s: sync forward %.1f img/s (%.2f ms per call)" % (tag, 192 / dt, 1e3 * dt / 24), flush=True)
def async_loop(tag):
    f = lambda: m.detect_packed(batch, pipelined=True, splits=3, defer=True)
    dt = bench.time_steps(f, 30, 5, False, flush_fn=m.flush_deferred)
    print("%s: deferred loop %.1f img/s" % (tag, 240 / dt), flush=True)
    f = lambda: m.detect_packed(batch, pipelined=True, splits=3)
    dt = bench.time_steps(f, 30, 5, False)
    print("%s: immediate loop %.1f img/s" % (tag, 240 / dt), flush=True)
order = os.environ.get("ORDER", "async")
for rep in range(2):
    if order == "async":
        async_loop("rep %d" % rep); sync_loop("rep %d" % rep)
    else:
        sync_loop("rep %d" % rep); async_loop("rep %d" % rep)
