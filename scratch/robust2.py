"""Shape sweep through the whole path (no crash, finite, well-formed) for the bf16 and fp8 models, both step layouts."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for cfgname, depth in (("dota-1.0_r50.yaml", 50), ("ucas_aod_r101_fp8.yaml", 101), ("hrsc_r50.yaml", 50)):
    cfg, model, sd = bench.build_model(depth, dev, cfgname=cfgname)
    if "fp8" in cfgname:        # explicit calibration (round 3): one mid-size batch pins the activation scales
        model.calibrate_fp8(torch.randint(0, 256, (2, 3, 256, 320), generator=g, dtype=torch.uint8).to(dev))
    for n, h, w, s in [(1, 32, 32, 1), (2, 33, 70, 2), (1, 1024, 1024, 3), (16, 512, 512, 3), (3, 1200, 1184, 3), (9, 64, 2048, 3),
                       (2, 1504, 1504, 2)]:
        b = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8).to(dev)
        for piped in (False, True):
            r, c = model.detect_packed(b, pipelined=piped, splits=s)
            torch.cuda.synchronize()
            k = c.tolist()
            assert all(0 <= x <= r.shape[1] for x in k), k
            for i in range(n):
                assert torch.isfinite(r[i, :k[i]]).all()
        print(cfgname, n, h, w, "ok, counts", k[:3])
    del model
print("ALL OK")
