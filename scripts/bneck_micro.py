# dafne_bottleneck_body_hip (conv2 3x3 + conv3 + residual + next conv1 in one kernel) against the two launches it replaces
# (conv_igemm 3x3, then dafne_bottleneck_tail_head_hip) at the res4 shape, K cycled buffer sets (beyond the MALL for K >= 2),
# post-ReLU inputs.  usage: bneck_micro.py [N]      DAFNE_BNECK_STAMPS=1 with a -DDAFNE_BNECK_TIMING library prints phase stamps
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = W = 64
g = torch.Generator().manual_seed(5)
w2p, b2p = engine.pack_conv(torch.randn(256, 256, 3, 3, generator=g) / 48, torch.randn(256, generator=g), d)
w3p, b3p = engine.pack_conv(torch.randn(1024, 256, 1, 1, generator=g) / 16, torch.randn(1024, generator=g), d)
w1p, b1p = engine.pack_conv(torch.randn(256, 1024, 1, 1, generator=g) / 32, torch.randn(256, generator=g), d)
wf_b = engine.pack_b2b(w3p, w1p)
wf = engine.pack_bneck(w2p, w3p, w1p)
nscr = L.dafne_bottleneck_body_scratch_bytes()
scr = torch.zeros(nscr, dtype=torch.uint8, device=d)
st = _lib.current_stream()
fl = 2 * N * H * W * (256 * 2304 + 2 * 256 * 1024)
for K in (1, 4):
    sets = []
    for k in range(K):
        ua = engine.Act.from_nchw(torch.relu(torch.randn(N, 256, H, W, generator=g)).to(d))
        xa = engine.Act.from_nchw(torch.relu(torch.randn(N, 1024, H, W, generator=g)).to(d))
        t = engine.Act(N, H, W, 256, d)
        sets.append((ua, xa, t, engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d),
                     engine.ConvCall(w2p, b2p, 256, 256, 3, 1, 1, engine.F_RELU, [(ua.t, t.t, None, H, W, H, W)], N)))
    def run_unfused(k):
        ua, xa, t, y, z, c2 = sets[k % K]
        c2(st)
        _lib.check(L.dafne_bottleneck_tail_head_hip(_lib.ptr(t.t), _lib.ptr(xa.t), _lib.ptr(wf_b), _lib.ptr(b3p), _lib.ptr(b1p), N, H, W, _lib.ptr(y.t), _lib.ptr(z.t), st), "b2b")
    def run_conv2(k):
        sets[k % K][5](st)
    def run_fused(k):
        ua, xa, t, y, z, c2 = sets[k % K]
        _lib.check(L.dafne_bottleneck_body_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b2p), _lib.ptr(b3p), _lib.ptr(b1p), N, H, W, _lib.ptr(y.t), _lib.ptr(z.t), _lib.ptr(scr), nscr, st), "bneck")
    res = {}
    for rnd in range(3):                      # interleaved rounds (guide rule 24)
        for name, fn in (("unfused", run_unfused), ("conv2 alone", run_conv2), ("fused", run_fused)):
            for k in range(2 * K): fn(k)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(24): fn(k)
            b.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(a.elapsed_time(b) / 24)
    for name, v in res.items():
        best = min(v)
        f = fl if name != "conv2 alone" else 2 * N * H * W * 256 * 2304
        print("K=%d %-12s %.1f us (median %.1f)  %.0f TF" % (K, name, best * 1e3, sorted(v)[1] * 1e3, f / (best * 1e-3) / 1e12))
    # same results?
    ua, xa, t, y, z, c2 = sets[0]
    run_unfused(0); torch.cuda.synchronize(); yu, zu = y.t.clone(), z.t.clone()
    y.t.zero_(); z.t.zero_(); run_fused(0); torch.cuda.synchronize()
    print("   bit-identical:", torch.equal(yu, y.t), torch.equal(zu, z.t))
    del sets
if os.environ.get("DAFNE_BNECK_STAMPS"):
    torch.cuda.synchronize()
    s = scr.view(torch.int64)[:64 * 48].reshape(32, 2, 48).cpu()
    print("phase stamps (cycles since kernel entry; median over the first 32 workgroups):")
    names = ["start", "phaseA end", "T ready", "G1(0)", "E(0)", "G2a(0)", "G2b(0)", "chunks done", "Z issued", "drained"]
    for wv, lab in ((0, "wave 0"), (1, "wave 4")):
        med = s[:, wv].median(dim=0).values.tolist()
        print("   %s:" % lab, ", ".join("%s %d" % (n, v) for n, v in zip(names, med)))
        print("   %s (us):" % lab, ", ".join("%s %.2f" % (n, v / 100.0) for n, v in zip(names, med[24:])))
