"""Does an MFMA-bound kernel beside an L2/HBM-bound one beat running each on the whole chip in turn?
Stream A: a head tower layer (conv3x3_rp, GroupNorm on load) of a 4-image sub-batch, workgroups capped at DAFNE_RP_GRID.
Stream B: res4 blocks (conv_bneck) of a 4-image sub-batch (128 tiles = 128 CUs).
Measured: each alone, two of a kind side by side (today's timed layout: the sub-batch streams run in step), and the mix.
usage: corun_micro.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
B = 4
g = torch.Generator().manual_seed(0)
# ---- tower layer (two instances: one per stream)
C = 256
sizes = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
b = torch.randn(C, generator=g) * 0.1
wp, bp = engine.pack_conv(w, b, d); wf = engine.pack_conv3x3_frag(wp)
gamma = torch.ones(C, device=d); beta = torch.zeros(C, device=d)
def tower():
    ins = [engine.Act(B, h, ww, C, d) for h, ww in sizes]
    for a in ins: a.t[:, 1:-1, 1:-1, :] = torch.randn(B, a.h, a.w, C, device=d).to(torch.bfloat16)
    outs = [engine.Act(B, h, ww, C, d) for h, ww in sizes]
    stats = torch.zeros(5, B, C // 8, 2, device=d); stats[..., 1] = 1.0
    segs = [(i.t, o.t, None, i.h, i.w, i.h, i.w) for i, o in zip(ins, outs)]
    partial = torch.zeros(8192, C // 8, 2, device=d)
    return engine.ConvCall(wp, bp, C, C, 3, 1, 1, engine.F_GN | engine.F_GNIN, segs, B, gn_partial=partial, gn_in=(stats, gamma, beta), wfrag=wf, shared_gpu=True)
# ---- res4 block
H = W = 64
w2p, b2p = engine.pack_conv(torch.randn(256, 256, 3, 3, generator=g) / 48, torch.randn(256, generator=g), d)
w3p, b3p = engine.pack_conv(torch.randn(1024, 256, 1, 1, generator=g) / 16, torch.randn(1024, generator=g), d)
w1p, b1p = engine.pack_conv(torch.randn(256, 1024, 1, 1, generator=g) / 32, torch.randn(256, generator=g), d)
wfb = engine.pack_bneck(w2p, w3p, w1p)
nscr = L.dafne_bottleneck_body_scratch_bytes()
def bneck():
    ua = engine.Act.from_nchw(torch.relu(torch.randn(B, 256, H, W, generator=g)).to(d))
    xa = engine.Act.from_nchw(torch.relu(torch.randn(B, 1024, H, W, generator=g)).to(d))
    y, z = engine.Act(B, H, W, 1024, d), engine.Act(B, H, W, 256, d)
    scr = torch.zeros(nscr, dtype=torch.uint8, device=d)
    keep = (ua, xa, y, z, scr)
    def call(st):
        _lib.check(L.dafne_bottleneck_body_hip(_lib.ptr(ua.t), _lib.ptr(xa.t), _lib.ptr(wfb), _lib.ptr(b2p), _lib.ptr(b3p), _lib.ptr(b1p), B, H, W, _lib.ptr(y.t), _lib.ptr(z.t), _lib.ptr(scr), nscr, st), "bneck")
    call.keep = keep
    return call
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
import ctypes
pA, pB = ctypes.c_void_p(sA.cuda_stream), ctypes.c_void_p(sB.cuda_stream)
tw = [tower(), tower()]; bn = [bneck(), bneck()]
REPS = 40
def run(fa, fb, reps=REPS):
    """fa on stream A, fb on stream B (either may be None), `reps` launches each, back to back; -> (time per launch of A, of B, wall)"""
    for f, p in ((fa, pA), (fb, pB)):
        if f:
            for _ in range(3): f(p)
    torch.cuda.synchronize()
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(2)]; eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    if fa: ea[0].record(sA)
    if fb: eb[0].record(sB)
    for _ in range(reps):
        if fa: fa(pA)
        if fb: fb(pB)
    if fa: ea[1].record(sA)
    if fb: eb[1].record(sB)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6 / reps
    return (ea[0].elapsed_time(ea[1]) * 1e3 / reps if fa else 0.0, eb[0].elapsed_time(eb[1]) * 1e3 / reps if fb else 0.0, wall)
print("DAFNE_RP_GRID =", os.environ.get("DAFNE_RP_GRID", "(none: balanced 232)"))
for rep in range(2):
    a, _, _ = run(tw[0], None);  print("tower layer alone                : %.1f us per launch" % a)
    _, b_, _ = run(None, bn[0]); print("res4 block alone (128 CUs)       : %.1f us per launch" % b_)
    a, b_, wl = run(tw[0], tw[1]); print("two tower layers side by side    : %.1f / %.1f us  (wall %.1f per pair)" % (a, b_, wl))
    a, b_, wl = run(bn[0], bn[1]); print("two res4 blocks side by side     : %.1f / %.1f us  (wall %.1f per pair)" % (a, b_, wl))
    # the mix: equal GPU time shares are not guaranteed, so report both rates
    a, b_, wl = run(tw[0], bn[0]); print("tower layer beside res4 blocks   : tower %.1f us, block %.1f us per launch" % (a, b_))
