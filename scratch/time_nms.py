import sys, time, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import rrects
from dafne_amd import _lib
L = _lib.load()
dev = torch.device('cuda', 0)
rng = np.random.default_rng(1234)
for N, m in ((1, 500), (1, 2000), (1, 10000), (8, 10000), (1, 27000)):
    b = np.stack([rrects(m, rng, extent=1024.0) for _ in range(N)])
    s = rng.uniform(0.05, 1, (N, m)).astype(np.float32)
    c = rng.integers(0, 15, (N, m)).astype(np.int32)
    tb, ts, tc = (torch.from_numpy(a).to(dev) for a in (b, s, c))
    tn = torch.full((N,), m, dtype=torch.int32, device=dev)
    keep = torch.empty((N, m), dtype=torch.int64, device=dev); nk = torch.zeros(N, dtype=torch.int32, device=dev)
    nbytes = L.dafne_poly_nms_workspace_bytes(N, m); ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    def run():
        _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), _lib.ptr(tn), N, m, 0.1, 1000,
                   _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, _lib.current_stream()))
    run(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(10): run()
    torch.cuda.synchronize()
    dt = (time.time() - t) / 10
    print("N=%d M=%d: %.3f ms per call, %.3f ms/img, kept %s" % (N, m, dt * 1e3, dt * 1e3 / N, nk.tolist()[:3]))
