# per-phase cycle stamps of conv_b2b_kernel (library built with -DDAFNE_B2B_TIMING): median over the first workgroups
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import engine, _lib
L = _lib.load(); d = torch.device("cuda", 0)
N, H, W = 8, 64, 64            # top halo row of d_next = 66 px x 512 B: room for 64 x 12 stamps? 33792 B >= 6144 B
g = torch.Generator().manual_seed(5)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # buffer sets cycled (K >= 2: beyond the Infinity Cache, real HBM traffic)
sets = [(engine.Act.from_nchw(torch.relu(torch.randn(N, 256, H, W, generator=g)).to(d)), engine.Act.from_nchw(torch.relu(torch.randn(N, 1024, H, W, generator=g)).to(d)),
         engine.Act(N, H, W, 1024, d), engine.Act(N, H, W, 256, d)) for _ in range(K)]
w3p, b3p = engine.pack_conv(torch.randn(1024, 256, 1, 1, generator=g) / 16, torch.randn(1024, generator=g), d)
w1p, b1p = engine.pack_conv(torch.randn(256, 1024, 1, 1, generator=g) / 32, torch.randn(256, generator=g), d)
wf = engine.pack_b2b(w3p, w1p)
st = _lib.current_stream()
for it in range(3 * K):
    ta, xa, y, z = sets[it % K]
    _lib.check(L.dafne_bottleneck_tail_head_hip(_lib.ptr(ta.t), _lib.ptr(xa.t), _lib.ptr(wf), _lib.ptr(b3p), _lib.ptr(b1p),
                                                N, H, W, _lib.ptr(y.t), _lib.ptr(z.t), st), "b2b")
torch.cuda.synchronize()
raw = z.t.view(-1)[: 64 * 12 * 4].view(torch.int64).reshape(64, 12).cpu()
names = ["start", "prologue done", "GEMM1(0) done", "epilogue(0) done", "chunk0 done", "chunk1 done", "chunk2 done", "chunk3 done",
         "final epilogue", "all stores acked"]
med = raw.median(0).values.tolist()
prev = 0
for nme, v in zip(names, med):
    print("%-20s %8d cycles  (+%d)" % (nme, v, v - prev)); prev = v
