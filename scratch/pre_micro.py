"""dafne_preprocess_image_hip alone: us per launch at batch 8 / 4 of 1024^2 (planar uint8).  DAFNE_AMD_LIB=<variant> for A/B."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dafne_amd import _lib
L = _lib.load(); d = torch.device("cuda", 0)
m3 = (ctypes.c_float * 3)(103.53, 116.28, 123.675); s3 = (ctypes.c_float * 3)(1.0, 1.0, 1.0)
for N in (8, 4, 1):
    H = W = 1024
    imgs = [torch.randint(0, 256, (N, 3, H, W), dtype=torch.uint8, device=d) for _ in range(4)]
    outs = [torch.zeros(N, H + 6, W + 6, 4, dtype=torch.bfloat16, device=d) for _ in range(4)]
    st = _lib.current_stream()
    def run(k): _lib.check(L.dafne_preprocess_image_hip(_lib.ptr(imgs[k % 4]), 0, N, H, W, None, m3, s3, H, W, _lib.ptr(outs[k % 4]), st))
    for k in range(8): run(k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(40): run(k)
    b.record(); torch.cuda.synchronize()
    print("batch %d: %.1f us per launch" % (N, a.elapsed_time(b) / 40 * 1e3))
