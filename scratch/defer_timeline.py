"""Where the deferred post-process runs: per step, relative to the start of the step's convolutions on stream 0 --
tower start, convolutions end, and the previous step's decode + NMS start / end on the side stream."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.modeling import one_stage_detector as osd
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
b = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
mode = sys.argv[1] if len(sys.argv) > 1 else "defer"
os.environ["DAFNE_HIP_GRAPHS"] = "0"          # eager launches: events can sit between any two launches
f = (lambda: m.detect_packed(b, pipelined=True, splits=2, defer=True)) if mode == "defer" else (lambda: m.detect_packed(b, pipelined=True, splits=2))
for _ in range(8): f()
torch.cuda.synchronize()
rec = []
orig_run = m._run_deferred
def run_deferred(p, evs):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(m.side_stream):
        for ev in list(p["ends"]) + list(evs): m.side_stream.wait_event(ev)
        s.record(m.side_stream)
    r = orig_run(p, evs)
    e.record(m.side_stream)
    rec[-1]["post"] = (s, e)
    return r
m._run_deferred = run_deferred
orig_eager = osd.OneStageDetector._enqueue_eager
def eager(plans, cs, sp, splits, defer):
    t0 = torch.cuda.Event(enable_timing=True); t0.record(cs[0])
    tw = []
    evs = []
    for j in range(max(len(p.calls) for p in plans)):
        for k in range(splits):
            if j < len(plans[k].calls):
                if j == plans[k].head_start:
                    e = torch.cuda.Event(enable_timing=True); e.record(cs[k]); tw.append(e)
                    if defer: evs.append(e)
                plans[k].calls[j](sp[k])
    te = torch.cuda.Event(enable_timing=True); te.record(cs[0])
    rec.append({"t0": t0, "tw": tw, "te": te})
    return evs
osd.OneStageDetector._enqueue_eager = staticmethod(eager)
N = 30
for _ in range(N): f()
m.flush_deferred() if mode == "defer" else None
torch.cuda.synchronize()
rows = []
for i in range(2, N):
    r = rec[i]
    if "post" not in r: continue
    t0 = r["t0"]
    rows.append((t0.elapsed_time(r["tw"][0]), t0.elapsed_time(r["tw"][1]), t0.elapsed_time(r["te"]), t0.elapsed_time(r["post"][0]), t0.elapsed_time(r["post"][1])))
import numpy as np
a = np.array(rows)
print("%s: per step (ms after stream 0 starts its convolutions): towers start %.2f / %.2f, convolutions end %.2f | previous step's decode + NMS: start %.2f, end %.2f (%.2f ms long)"
      % (mode, *np.median(a, 0)[:5], np.median(a[:, 4] - a[:, 3])))
