# res4 conv2 / FPN output shapes (one map size, plain bf16 input): bf16 kernel the library picks vs the fp8 patch kernel
# python scratch/fp8_res4_micro.py [batch]
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
from dafne_amd.engine import F_RELU
d = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(0)
def bench(calls, reps=20):
    for c in calls: c(_lib.current_stream())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        for c in calls: c(_lib.current_stream())
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / len(calls) * 1e3
for (H, C, name) in ((64, 256, "res4 conv2 / fpn_output4"), (128, 256, "fpn_output3"), (32, 512, "res5 conv2")):
    wt = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    w, b = engine.pack_conv(wt, torch.randn(C, generator=g) * 0.1, d)
    wq, sc = engine.pack_conv_fp8(wt, d)
    K = 6      # cycled instances: no MALL reuse between launches
    xs = [engine.Act.from_nchw(torch.relu(torch.randn(N, C, H, H, generator=g)).to(d)) for _ in range(K)]
    ys = [engine.Act(N, H, H, C, d) for _ in range(K)]
    cb = [engine.ConvCall(w, b, C, C, 3, 1, 1, F_RELU, [(x.t, y.t, None, H, H, H, H)], N) for x, y in zip(xs, ys)]
    cq = [engine.ConvCall(wq, b, C, C, 3, 1, 1, F_RELU, [(x.t, y.t, None, H, H, H, H)], N, fp8=(sc / 8.0, 8.0)) for x, y in zip(xs, ys)]
    fl = 2 * N * H * H * C * C * 9
    tb, tq = bench(cb), bench(cq)
    print("%-26s batch %d: bf16 %s %.1f us (%.0f TF)   fp8 patch %.1f us (%.0f TF)   x%.2f" % (name, N, cb[0].kernel_name(), tb, fl / tb / 1e6, tq, fl / tq / 1e6, tb / tq))
