"""Timing ablation (wrong results): skip the launches of the pipelined step below DAFNE_EXP_SKIP_GF GFLOP and report what they
were and how long they take alone -- does the step get shorter by their duration (latency-chain-bound) or not (CU-time-bound)?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
f = lambda: m.detect_packed(batch, pipelined=True, splits=3)
for _ in range(4): f()
torch.cuda.synchronize()
st = m._pipe[(8, 1024, 1024, 3)]
for thr in (0.0, 1.0, 3.0, 6.0):
    os.environ["DAFNE_EXP_SKIP_GF"] = str(thr)
    sk = {}
    for c in st["plans"][0][1].calls:
        fl = getattr(c, "flops", 0)
        if thr and 0 < fl < thr * 1e9:
            n = c.kernel_name() if hasattr(c, "kernel_name") else c.name
            sk[n] = sk.get(n, 0) + 1
    res = []
    for rep in range(3):
        res.append(8 * 20 / bench.time_steps(f, 20, 2, False))
    print("skip < %.1f GFLOP: %s  ->  %s img/s" % (thr, sk, ["%.0f" % r for r in res]), flush=True)
