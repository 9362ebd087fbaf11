"""The WHOLE network (serial plan, every convolution kernel family, GroupNorm-fed towers, split-K) beside torch.matmul on three streams:
head outputs and FPN features against the idle-GPU run, bit for bit."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_gpu_reproducible as TR
import test_inference_loop as T
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfgname = os.environ.get("CFGNAME", "dota-1.0_r50.yaml")
cfg, m = T._gpu_model(cfgname)
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(21)
n, H, W = int(os.environ.get("N", "2")), int(os.environ.get("H", "512")), int(os.environ.get("W", "640"))
img = torch.randint(0, 256, (n, 3, H, W), generator=g, dtype=torch.uint8).cuda()
plan = m.plan(n, H, W)
def snap():
    m.detect_packed(img)
    return [f.t.clone() for f in plan.features] + [t.clone() for lst in (plan.head.logits, plan.head.center, plan.head.delta_ctr) for t in lst]
ref = snap(); torch.cuda.synchronize()
load = TR._MatrixLoad(d)
bad = {}
for it in range(iters):
    load.kick(10)
    s = snap()
    if it % 4 == 3 or it == iters - 1:
        torch.cuda.synchronize()
    # compare after sync (snap tensors are clones on the current stream)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(s, ref)):
        if not torch.equal(a, b): bad[k] = bad.get(k, 0) + 1
print("%d runs of the %s network at %dx%dx%d beside matmul on three streams: tensors that ever differed: %s" % (iters, cfgname, n, H, W, bad or "none"))
