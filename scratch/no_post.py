"""Upper bound of what a faster post-process could give the timed layout: the same step with decode + NMS replaced by nothing
(timing only: the previous step's packed results are handed back)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
b = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
f = lambda: m.detect_packed(b, pipelined=True, splits=2)
for _ in range(8): f()
torch.cuda.synchronize()
def rate(n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return 8 * n / (time.perf_counter() - t0)
outs = m.proposal_generator.dafne_outputs
print("full step           : %.1f / %.1f img/s" % (rate(), rate()))
cand = outs.decode_packed.__func__
keep = {}
orig_dec, orig_sel = outs.decode_packed, outs.select_packed
def dec(levels, out=None, **kw):
    if "c" not in keep: keep["c"] = orig_dec(levels, out=out, **kw)
    return keep["c"]
def sel(c, **kw):
    if "r" not in keep: keep["r"] = orig_sel(c, **kw)
    return keep["r"]
outs.decode_packed, outs.select_packed = dec, sel
f(); torch.cuda.synchronize()
print("no decode, no NMS   : %.1f / %.1f img/s" % (rate(), rate()))
outs.decode_packed = orig_dec
keep.pop("c", None)
print("decode only, no NMS : %.1f / %.1f img/s" % (rate(), rate()))
outs.select_packed = orig_sel
print("full step again     : %.1f / %.1f img/s" % (rate(), rate()))
