"""Phase timestamps of the pipelined step WITHOUT a profiler attached (HIP events on the streams themselves): per step, when
each sub-batch stream passes its preprocess and its last launch, and when decode / NMS start and end on the side stream.
usage: pipe_events.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
f = lambda: m.detect_packed(batch, pipelined=True, splits=3)
for _ in range(6): f()
torch.cuda.synchronize()
m._dbg_events = []
ref = torch.cuda.Event(enable_timing=True); ref.record(torch.cuda.current_stream())
for _ in range(steps): f()
torch.cuda.synchronize()
ev = m._dbg_events
m._dbg_events = None
by = {}
for i, tag, k, e in ev:
    by.setdefault(i, {})[(tag, k)] = ref.elapsed_time(e)
ids = sorted(by)
for a, b in zip(ids[2:-1], ids[3:]):
    A, B = by[a], by[b]
    t0 = min(A[("pre", k)] for k in range(3))
    ends = [A[("end", k)] - t0 for k in range(3)]
    pres = [B[("pre", k)] - t0 for k in range(3)]
    print("step %d: pre %s  end %s | dec %.2f..%.2f nms ..%.2f | next pre %s  => idle per stream %s"
          % (a, ["%.2f" % (A[("pre", k)] - t0) for k in range(3)], ["%.2f" % x for x in ends], A[("dec0", 0)] - t0, A[("dec1", 0)] - t0,
             A[("nms1", 0)] - t0, ["%.2f" % x for x in pres], ["%.2f" % (p - e) for p, e in zip(pres, ends)]))
