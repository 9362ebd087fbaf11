"""Config 5 per-kernel times in the timed (2-stream) layout and serial / pipelined rates, fp8 3x3 kernels A/B.
usage: fp8_prof.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
b16 = torch.randint(0, 256, (16, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
for name, kern in (("bf16", None), ("fp8 patch", "patch"), ("fp8 rp8", "rp8")):
    m = bench.build_model(101, d, seed=0, cfgname="ucas_aod_r101_fp8.yaml" if kern else "ucas_aod_r101.yaml", cls_prior=-1.5)[1]
    if kern:
        if hasattr(m.cfg, "defrost"): m.cfg.defrost()
        m.cfg.ENGINE.FP8_CONV3X3_KERNEL = kern
        m.invalidate()
        m.calibrate_fp8(b16)
    fs = lambda: m.detect_packed(b16)
    fp = lambda: m.detect_packed(b16, pipelined=True, splits=2)
    for _ in range(5):
        fs(); fp()
    torch.cuda.synchronize()
    for rep in range(2):
        print("%s: serial %.1f img/s   pipelined %.1f img/s" % (name, 16 * steps / bench.time_steps(fs, steps, 1, False),
                                                                16 * steps / bench.time_steps(fp, steps, 1, False)), flush=True)
    for title, st in (("timed layout", bench.conv_kernel_profile(m, b16, 2)), ("isolated", bench.conv_kernel_profile_isolated(m, b16))):
        print("  -- %s" % title)
        tot = 0.0
        ms = lambda s: s["ms"] if "ms" in s else s["ms_per_step"]
        for k, s in sorted(st.items(), key=lambda kv: -ms(kv[1])):
            tot += ms(s)
            print("  %-24s %3d launches  %7.3f ms  union %7.3f ms  %6.1f us/launch  %5.0f TFLOP/s" % (k, s["launches"], ms(s), s.get("union_ms", 0.0), 1e3 * ms(s) / s["launches"], s["tflops"]))
        print("  total %.3f ms" % tot, flush=True)
    del m
