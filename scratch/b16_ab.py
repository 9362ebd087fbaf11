"""bf16 UCAS head at batch 16 (the bf16 side of bench.py's configs4 line) under engine toggles given in the environment."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
dev = torch.device("cuda", 0)
cfgname = sys.argv[1] if len(sys.argv) > 1 else "ucas_aod_r101.yaml"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg, m, sd = bench.build_model(101, dev, seed=0, cfgname=cfgname, cls_prior=-1.5)
b = torch.randint(0, 256, (n, 3, 1024, 1024), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
if cfg.ENGINE.WEIGHT_DTYPE == "fp8_e4m3":
    m.calibrate_fp8(b)
f = lambda: m.detect_packed(b, pipelined=True, splits=3)
best = 0
for rep in range(2):
    dt = bench.time_steps(f, 12, 3, False)
    best = max(best, n * 12 / dt)
print(cfgname, n, {k: os.environ[k] for k in os.environ if k.startswith("DAFNE_")}, "%.1f img/s" % best)
