# rotated NMS on one SURVEY 8(d) candidate set: python scratch/nms_sets.py KIND M N_IMAGES [REPS]   (run under rocprofv3 for per-kernel times)
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from dafne_amd import _lib
from conftest import nms_candidate_set
kind, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
L = _lib.load(); dev = torch.device("cuda", 0)
rng = np.random.default_rng(1234)
sets = [nms_candidate_set(kind, m, rng) for _ in range(n)]
tb = torch.from_numpy(np.stack([x[0] for x in sets])).to(dev)
ts = torch.from_numpy(np.stack([x[1] for x in sets])).to(dev)
tc = torch.from_numpy(np.stack([x[2] for x in sets]).astype(np.int32)).to(dev)
keep = torch.empty((n, m), dtype=torch.int64, device=dev); nk = torch.zeros(n, dtype=torch.int32, device=dev)
nb = L.dafne_poly_nms_workspace_bytes(n, m); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def run():
    _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), None, n, m, 0.1, 1000, _lib.ptr(keep), _lib.ptr(nk),
                                                  _lib.ptr(ws), nb, _lib.current_stream()))
for _ in range(60): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(reps): run()
e1.record()
torch.cuda.synchronize()
print("events: %.3f ms/call" % (e0.elapsed_time(e1) / reps))
off = L.dafne_poly_nms_stats_offset(n, m, 0)
st = ws[off:off + 16 * n].view(torch.int32).reshape(n, 4).sum(0).tolist()
print("%s M=%d x%d: %.3f ms/call  %.4f ms/img  kept %s  paths fast+ %d fast- %d exact %d ovf %d" % (kind, m, n, 1e3 * (time.perf_counter() - t0) / reps,
      1e3 * (time.perf_counter() - t0) / reps / n, nk.tolist()[:3], *st))
