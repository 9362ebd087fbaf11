"""Run-to-run stability of the sub-batch layout on small images (many tiny launches on concurrent streams): n images through
detect_packed(pipelined=True, splits=S) ITER times; compares the head outputs and the detections of every call with the first call's
and with the serial plan's.  usage: race_probe.py [n] [splits] [iters] [H] [W]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import dafne_amd.modeling  # noqa
from dafne_amd.config import load_cfg
from dafne_amd.registry import build_model
from oracle import model as om
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
H = int(sys.argv[4]) if len(sys.argv) > 4 else 128
W = int(sys.argv[5]) if len(sys.argv) > 5 else 128
cfg = load_cfg(os.path.join(R, "configs", os.environ.get("CFGNAME", "dota-1.0_r50.yaml")))
m = build_model(cfg)
m.load_state_dict(om.make_params(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.DAFNE.NUM_CLASSES, seed=7))
d = torch.device("cuda", 0); m.to(d); m.invalidate()
g = torch.Generator().manual_seed(2)
batch = torch.randint(0, 256, (n, 3, H, W), generator=g, dtype=torch.uint8).to(d)
def heads(ho):
    return [t.clone() for lst in (ho.logits, ho.center, ho.delta_ctr) for t in lst]
rows0, counts0 = m.detect_packed(batch); torch.cuda.synchronize()
h_serial = heads(m._last_head)
ref = None; bad = 0; badser = 0
for it in range(iters):
    rows, counts = m.detect_packed(batch, pipelined=True, splits=S)
    torch.cuda.synchronize()
    hs = heads(m._last_head)
    if ref is None:
        ref = (rows.clone(), counts.clone(), hs)
        same_serial = all(torch.equal(a, b) for a, b in zip(hs, h_serial))
        print("first pipelined call equals the serial plan (head outputs): %s; detections equal: %s" % (same_serial, torch.equal(rows, rows0) and torch.equal(counts, counts0)))
        continue
    diff = [k for k, (a, b) in enumerate(zip(hs, ref[2])) if not torch.equal(a, b)]
    if diff or not torch.equal(counts, ref[1]):
        bad += 1
        if bad <= 5:
            k = diff[0] if diff else -1
            extra = ""
            if diff:
                a, b = hs[k], ref[2][k]
                ne = (a != b).reshape(a.shape[0], -1).sum(1).tolist()
                extra = " tensor %d shape %s mismatching elements per image %s max |d| %.3g" % (k, tuple(a.shape), ne, float((a - b).abs().max()))
            print("iter %d: differs from the first call; counts %s vs %s;%s" % (it, counts.tolist(), ref[1].tolist(), extra))
print("n=%d splits=%d %dx%d: %d of %d calls differ from the first" % (n, S, H, W, bad, iters - 1))
