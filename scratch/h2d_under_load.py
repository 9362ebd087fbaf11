"""H2D copy of one batch (25 MB) while the pipelined step keeps the GPU busy: pinned async on streams of either priority, pageable blocking."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8).to(dev)
host = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8)
pinned = host.pin_memory()
for _ in range(3):
    model.detect_packed(batch, pipelined=True, splits=3)
torch.cuda.synchronize()
def load(n=6):
    for _ in range(n):
        model.detect_packed(batch, pipelined=True, splits=3)
for name, prio in (("normal", 0), ("high", -1)):
    s = torch.cuda.Stream(device=dev, priority=prio)
    dst = torch.empty_like(batch)
    for busy in (False, True):
        torch.cuda.synchronize()
        if busy:
            load()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            a.record(s); dst.copy_(pinned, non_blocking=True); b.record(s)
        t1 = time.perf_counter()
        b.synchronize()
        t2 = time.perf_counter()
        print("pinned async, %s-priority stream, gpu %s: issue %.2f ms, done after %.2f ms, event span %.2f ms" % (name, "busy" if busy else "idle", 1e3 * (t1 - t0), 1e3 * (t2 - t0), a.elapsed_time(b)))
        torch.cuda.synchronize()
for busy in (False, True):
    torch.cuda.synchronize()
    if busy:
        load()
    t0 = time.perf_counter()
    d = host.to(dev)
    t1 = time.perf_counter()
    print("pageable blocking .to(), gpu %s: %.2f ms" % ("busy" if busy else "idle", 1e3 * (t1 - t0)))
    torch.cuda.synchronize()
