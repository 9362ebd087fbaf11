"""dafne_bottleneck_block_mid_hip (a whole res3 block per launch) at the headline shape: us per launch, K cycled buffer sets.
usage: mid_micro.py [N] [head]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
head = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H = W = 128
d = torch.device("cuda", 0); L = _lib.load(); st = _lib.current_stream()
g = torch.Generator().manual_seed(1)
bf = lambda x: x.to(torch.bfloat16).float()
w2p, b2p = engine.pack_conv(bf(torch.randn(128, 128, 3, 3, generator=g) / 34), torch.randn(128, generator=g) * .2, d)
w3p, b3p = engine.pack_conv(bf(torch.randn(512, 128, 1, 1, generator=g) / 11), torch.randn(512, generator=g) * .2, d)
w1p, b1p = engine.pack_conv(bf(torch.randn(128, 512, 1, 1, generator=g) / 22), torch.randn(128, generator=g) * .2, d)
wf = engine.pack_blk_mid(w2p, w3p, w1p if head else None)
scr = torch.empty(L.dafne_bottleneck_block_mid_scratch_bytes(), dtype=torch.uint8, device=d)
K = 3
sets = []
for k in range(K):
    ua = engine.Act.from_nchw(torch.relu(torch.randn(N, 128, H, W, generator=g)).to(d))
    xa = engine.Act.from_nchw(torch.relu(torch.randn(N, 512, H, W, generator=g)).to(d))
    sets.append((ua, xa, engine.Act(N, H, W, 512, d), engine.Act(N, H, W, 128, d)))
def run(k):
    ua, xa, y, z = sets[k % K]
    _lib.check(L.dafne_bottleneck_block_mid_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p), _lib.ptr(b1p) if head else None,
                                                N, H, W, _lib.ptr(y.t), _lib.ptr(z.t) if head else None, _lib.ptr(scr), scr.numel(), st), "mid")
res = []
for rnd in range(4):
    for k in range(2 * K): run(k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(24): run(k)
    b.record(); torch.cuda.synchronize()
    res.append(a.elapsed_time(b) / 24 * 1e3)
fl = 2 * N * H * W * (128 * 1152 + 2 * 128 * 512)
print("blk_mid N=%d head=%d: %.1f us (median %.1f)  %.0f TFLOP/s, %.2f TB/s algorithmic" % (N, head, min(res), sorted(res)[len(res) // 2], fl / (min(res) * 1e-6) / 1e12,
      N * H * W * (128 * 1.6 + 512 * 2 + (128 if head else 0)) * 2 / (min(res) * 1e-6) / 1e12))
if os.environ.get("DAFNE_MID_STAMPS"):
    torch.cuda.synchronize(); scr.zero_(); run(0); torch.cuda.synchronize()
    s = scr.view(torch.int64)[:64 * 24].reshape(64, 24).cpu()
    names = ["start", "top barrier", "A loop end (wave 0)", "T ready"]
    for h in (0, 1):
        names += ["h%d GEMM1a" % h, "h%d GEMM1b (epi a done)" % h, "h%d epi b done" % h, "h%d loads issued" % h, "h%d barrier" % h, "h%d rows issued" % h, "h%d GEMM2 end" % h]
    names += ["Z rows issued"]
    med = s.median(dim=0).values.tolist()
    print("stamps of the second tile (cycles, median over 64 workgroups):", ", ".join("%s %d" % (n, v) for n, v in zip(names, med)))
