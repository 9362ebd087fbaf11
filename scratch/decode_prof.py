"""Post-process kernels of the headline batch alone (decode_hist / pick / collect / finalize, NMS, gather): HIP-event time of
decode_packed and select_packed on the bench model's own head outputs.  python scratch/decode_prof.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.modeling.dafne.dafne import head_levels
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
model.detect_packed(batch); torch.cuda.synchronize()
plan = model.plan(8, 1024, 1024)
outs = model.proposal_generator.dafne_outputs
lv = head_levels(plan.head, model.proposal_generator.fpn_strides)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): r = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3, r
td, cand = t(lambda: outs.decode_packed(lv))
ts, _ = t(lambda: outs.select_packed(cand))
print("decode_packed %.1f us, select_packed (NMS + gather) %.1f us per batch of 8" % (td, ts))
