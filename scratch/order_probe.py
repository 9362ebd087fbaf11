"""The order effect of HIP-graph replay: a two-sub-batch step with a host wait (the synchronous forward's layout) first, then the
deferred three-sub-batch loop.  SYNC_GRAPHS=0/1: the first layout eager or from graphs; SYNC_OFF: stream_offset of the first layout."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
torch.set_num_threads(16)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
sg = os.environ.get("SYNC_GRAPHS", "1") == "1"
off = int(os.environ.get("SYNC_OFF", "0"))
ncalls = int(os.environ.get("SYNC_CALLS", "28"))
def step():
    m.detect_packed(batch, pipelined=True, splits=2, even=True, stream_offset=off, graphs=sg)
    torch.cuda.synchronize()
t0 = None
for i in range(ncalls):
    if i == 4: t0 = time.perf_counter()
    step()
if ncalls > 4:
    print("two equal sub-batches, host wait per call, graphs %s, offset %d: %.2f ms per call" % (sg, off, 1e3 * (time.perf_counter() - t0) / (ncalls - 4)), flush=True)
f = lambda: m.detect_packed(batch, pipelined=True, splits=3, defer=True)
for rep in range(2):
    dt = bench.time_steps(f, 30, 5, False, flush_fn=m.flush_deferred)
    print("deferred 3-sub-batch loop (graphs): %.1f img/s" % (240 / dt), flush=True)
