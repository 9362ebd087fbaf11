import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import oracle
from conftest import rrects
from dafne_amd.modeling.nms import poly_gpu_nms
rng = np.random.default_rng(8)
for m in (300, 320, 384, 385, 448, 512, 640, 1000):
    for ext in (100.0, 512.0):
        b = rrects(m, rng, extent=ext)
        s = rng.uniform(0.05, 1, m).astype(np.float32)
        d9 = np.concatenate([b, s[:, None]], 1).astype(np.float32)
        got = poly_gpu_nms(d9, 0.1, 0); exp = oracle.poly_nms(d9, 0.1)
        first = next((i for i,(a,c) in enumerate(zip(got,exp)) if a!=c), None)
        print(m, ext, len(got), len(exp), got == exp, "first diff", first)
        if got != exp and first is not None:
            order = oracle.score_order(d9)
            pos = {int(o):i for i,o in enumerate(order)}
            print("   exp elem sorted pos", pos[exp[first]], "got elem pos", pos[got[first]])
