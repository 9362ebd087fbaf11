import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch, time
from dafne_amd import _lib
sys.path.insert(0, os.path.join(R, "tests"))
from conftest import rrects
L = _lib.load()
dev = torch.device("cuda", 0)
m, n_images = 10000, 8
rng = np.random.default_rng(1234)
b = np.stack([rrects(m, rng, extent=1024.0) for _ in range(n_images)])
s = rng.uniform(0.05, 1, (n_images, m)).astype(np.float32)
c = rng.integers(0, 15, (n_images, m)).astype(np.int32)
tb, ts, tc = (torch.from_numpy(a).to(dev) for a in (b, s, c))
tn = torch.full((n_images,), m, dtype=torch.int32, device=dev)
keep = torch.empty((n_images, m), dtype=torch.int64, device=dev); nk = torch.zeros(n_images, dtype=torch.int32, device=dev)
nbytes = L.dafne_poly_nms_workspace_bytes(n_images, m); ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
_lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), _lib.ptr(tn), n_images, m, 0.1, 1000, _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, _lib.current_stream()))
torch.cuda.synchronize()
nblk = (m + 63) // 64
meta = ws[:n_images * 16].view(torch.int32).reshape(n_images, 4).cpu().numpy()
pc = ws[n_images * 16: n_images * 16 + n_images * nblk * 4].view(torch.int32).reshape(n_images, nblk).cpu().numpy()
print("overflow flags", meta[:, 3].tolist())
print("pairs per row block: mean %.0f max %d, blocks over 1024: %d of %d" % (pc.mean(), pc.max(), (pc > 1024).sum(), pc.size))
print("total pairs per image", pc.sum(1).tolist()[:4], "kept", nk.tolist()[:4])
