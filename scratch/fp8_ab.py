"""Config 5 (UCAS-AOD head, R101-FPN, batch 16, fp8 weights) in one process: bf16 model, fp8 model with the 256-input layers on
conv3x3_patch_fp8 (ENGINE.FP8_CONV3X3_KERNEL "patch") and on conv3x3_rp8 ("rp8"); alternating timed blocks.
usage: fp8_ab.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
b16 = torch.randint(0, 256, (16, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
models = {}
SPL = int(os.environ.get("SPLITS", "2"))
for name, cfgname, kern in (("bf16", "ucas_aod_r101.yaml", None), ("fp8 patch", "ucas_aod_r101_fp8.yaml", "patch"), ("fp8 rp8", "ucas_aod_r101_fp8.yaml", "rp8")):
    m = bench.build_model(101, d, seed=0, cfgname=cfgname, cls_prior=-1.5)[1]
    if kern:
        if hasattr(m.cfg, "defrost"): m.cfg.defrost()
        m.cfg.ENGINE.FP8_CONV3X3_KERNEL = kern
        m.invalidate()
    if "fp8" in name:
        m.calibrate_fp8(b16)
    f = (lambda m: (lambda: m.detect_packed(b16, pipelined=True, splits=SPL)))(m)
    for _ in range(6):
        f()
    torch.cuda.synchronize()
    models[name] = f
for rep in range(3):
    for name, f in models.items():
        dt = bench.time_steps(f, steps, 1, False)
        print("%d %-16s %.1f img/s" % (rep, name, 16 * steps / dt), flush=True)
