"""conv_wr (128 px x 256 ch tiles, weights -> registers, split-K) vs the kernel dafne_conv2d_nhwc_bf16_hip picks, layer by layer
for the small-M layers of R101-FPN at 1024 x 1024.   python scratch/wr_micro.py [batch] [excl 0/1]"""
import ctypes
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
EXCL = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
d = torch.device("cuda", 0)
L = _lib.load()
LAYERS = [  # name, cin, cout, k, stride, Hin, flags, res
    ("res5.0.conv1 s2", 1024, 512, 1, 2, 64, engine.F_RELU, None),
    ("res5.0.shortcut s2", 1024, 2048, 1, 2, 64, 0, None),
    ("res5.b.conv2", 512, 512, 3, 1, 32, engine.F_RELU, None),
    ("res5.b.conv3", 512, 2048, 1, 1, 32, engine.F_RELU | engine.F_RES, "res"),
    ("res5.b.conv1", 2048, 512, 1, 1, 32, engine.F_RELU, None),
    ("lateral5", 2048, 256, 1, 1, 32, 0, None),
    ("output5", 256, 256, 3, 1, 32, 0, None),
    ("p6", 256, 256, 3, 2, 32, 0, None),
    ("p7", 256, 256, 3, 2, 16, 0, None),
    ("lateral4 +up", 1024, 256, 1, 1, 64, engine.F_UP, "up"),
    ("output4 / res4.22.conv2", 256, 256, 3, 1, 64, 0, None),
    ("res4.0.conv1 s2", 512, 256, 1, 2, 128, engine.F_RELU, None),
    ("res4.0.shortcut s2", 512, 1024, 1, 2, 128, 0, None),
    ("res4.22.conv3", 256, 1024, 1, 1, 64, engine.F_RELU | engine.F_RES, "res"),
    ("lateral3 +up", 512, 256, 1, 1, 128, engine.F_UP, "up"),
]
g = torch.Generator().manual_seed(0)
print("batch", B, "exclusive", EXCL)
for name, cin, cout, k, stride, H, flags, rk in LAYERS:
    pad = 1 if k == 3 else 0
    ho, wo = engine.conv_out_hw(H, H, k, stride, pad)
    K = 4                                    # cycled buffer sets (cold-ish L2 for the activations, as in the network)
    ins = [engine.Act.from_nchw(torch.randn(B, cin, H, H, generator=g).to(d)) for _ in range(K)]
    res = None
    if rk == "res":
        res = engine.Act.from_nchw(torch.randn(B, cout, ho, wo, generator=g).to(d))
    elif rk == "up":
        res = engine.Act.from_nchw(torch.randn(B, cout, ho // 2, wo // 2, generator=g).to(d))
    wgt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    wp, bp = engine.pack_conv(wgt, torch.randn(cout, generator=g) * 0.1, d)
    wf = engine.pack_conv_frag(wp)
    ws = engine.WrWorkspace(d)
    outs = [engine.Act(B, ho, wo, cout, d) for _ in range(K)]
    base, wr = [], []
    for i in range(K):
        c = engine.ConvCall(wp, bp, cin, cout, k, stride, pad, flags, [(ins[i].t, outs[i].t, res.t if res is not None else None, H, H, ho, wo)], B,
                            shared_gpu=not EXCL)
        base.append(c)
        assert L.dafne_conv2d_wr_ok(ctypes.byref(c.prm), c.segs)
        wr.append(engine.WrCall(c, wf, ws))
    st = _lib.current_stream()

    def t(calls, reps=20):
        for c in calls:
            c(st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for r in range(reps):
            calls[r % K](st)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    tb, tw = t(base), t(wr)
    fl = base[0].flops
    print("%-26s %-20s %6.1f us %6.0f TF | conv_wr S=%d %6.1f us %6.0f TF  x%.2f" % (name, base[0].kernel_name(), tb, fl / tb / 1e6, wr[0].splits, tw, fl / tw / 1e6, tb / tw))
