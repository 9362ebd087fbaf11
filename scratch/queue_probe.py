"""Which compute streams share a hardware queue?  After the three-sub-batch loop has bound the streams: two equal sub-batches with a
host wait per call on every pair of the three compute streams (stream_offset 0 / 1 / 2 = streams (0,1) / (1,2) / (2,0))."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
torch.set_num_threads(16)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
if os.environ.get("ORDER", "async") == "async":
    f = lambda: m.detect_packed(batch, pipelined=True, splits=3, defer=True)
    dt = bench.time_steps(f, 30, 5, False, flush_fn=m.flush_deferred)
    print("deferred 3-sub-batch loop %.1f img/s" % (240 / dt), flush=True)
for off in (0, 1, 2, 0):
    def step():
        r = m.detect_packed(batch, pipelined=True, splits=2, even=True, stream_offset=off)
        torch.cuda.synchronize()
    for _ in range(5): step()
    t0 = time.perf_counter()
    for _ in range(24): step()
    dt = time.perf_counter() - t0
    print("two equal sub-batches on compute streams (%d, %d), host wait per call: %.2f ms per call" % (off, (off + 1) % 3, 1e3 * dt / 24), flush=True)
if os.environ.get("ORDER", "async") != "async":
    f = lambda: m.detect_packed(batch, pipelined=True, splits=3, defer=True)
    dt = bench.time_steps(f, 30, 5, False, flush_fn=m.flush_deferred)
    print("deferred 3-sub-batch loop %.1f img/s" % (240 / dt), flush=True)
