import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import oracle
from conftest import rrects
from dafne_amd import _lib
L = _lib.load()
rng = np.random.default_rng(8)
m = 200
b = rrects(m, rng, extent=100.0)
s = rng.uniform(0.05, 1, m).astype(np.float32)
d9 = np.concatenate([b, s[:, None]], 1).astype(np.float32)
dev = torch.device('cuda', 0)
d = torch.from_numpy(d9).to(dev)
keep = torch.empty(m, dtype=torch.int64, device=dev); nk = torch.zeros(1, dtype=torch.int32, device=dev)
nbytes = L.dafne_poly_nms_workspace_bytes(1, m)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
_lib.check(L.dafne_poly_nms_hip(_lib.ptr(d), m, 0.1, _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, _lib.current_stream()))
torch.cuda.synchronize()
Mp = (m + 63)//64*64; nb = Mp//64
def al(x): return (x + 255)//256*256
off = 0
def take(nbytes_):
    global off
    off = al(off); r = off; off += nbytes_; return r
o_meta = take(16); o_flag = take(nb*8); o_order = take(Mp*4); o_sbox = take(Mp*32); o_ss = take(Mp*4)
o_hull = take(Mp*16); o_area = take(Mp*8); o_d9 = take(Mp*36); o_mask = take(Mp*nb*8)
w = ws.cpu().numpy()
order = w[o_order:o_order+Mp*4].view(np.int32)[:m]
mask = w[o_mask:o_mask+Mp*nb*8].view(np.uint64).reshape(Mp, nb)
exp_order = oracle.score_order(d9)
print("order ok", np.array_equal(order, exp_order))
sp = d9[exp_order, :8].astype(np.float64)
bad = 0
for r in range(m):
    iou = oracle.iou_poly_pairs(np.repeat(sp[r:r+1], m, 0), sp)
    for cb in range(r//64, nb):
        word = int(mask[r, cb])
        expw = 0
        for c in range(64):
            gc = cb*64 + c
            if gc < m and gc > r and iou[gc] > 0.1: expw |= 1 << c
        if word != expw:
            bad += 1
            if bad < 10: print("row", r, "cb", cb, hex(word), hex(expw))
print("bad words", bad)
got = keep[:int(nk)].cpu().tolist(); exp = oracle.poly_nms(d9, 0.1)
print(len(got), len(exp), got == exp)
