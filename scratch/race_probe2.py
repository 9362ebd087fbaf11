"""First calls after plan building: a fresh model per round, two pipelined calls (both eager: one per plan set) against the serial plan.
usage: race_probe2.py [n] [splits] [rounds] [H] [W]   env HOSTIMG=1: images start on the host and go through forward()"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import dafne_amd.modeling  # noqa
from dafne_amd.config import load_cfg
from dafne_amd.registry import build_model
from oracle import model as om
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 30
H = int(sys.argv[4]) if len(sys.argv) > 4 else 128
W = int(sys.argv[5]) if len(sys.argv) > 5 else 128
cfg = load_cfg(os.path.join(R, "configs", "dota-1.0_r50.yaml"))
params = om.make_params(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.DAFNE.NUM_CLASSES, seed=7)
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(2)
batch = torch.randint(0, 256, (n, 3, H, W), generator=g, dtype=torch.uint8).to(d)
def heads(ho):
    return [t.clone() for lst in (ho.logits, ho.center, ho.delta_ctr) for t in lst]
names = ["logits%d" % i for i in range(5)] + ["center%d" % i for i in range(5)] + ["delta%d" % i for i in range(5)]
bad = 0
ref = None
for r in range(rounds):
    m = build_model(cfg); m.load_state_dict(params); m.to(d); m.invalidate()
    m.detect_packed(batch); torch.cuda.synchronize()
    hs = heads(m._last_head)
    if ref is None: ref = hs
    assert all(torch.equal(a, b) for a, b in zip(hs, ref)), "serial plan not reproducible"
    for call in range(3):
        m.detect_packed(batch, pipelined=True, splits=S); torch.cuda.synchronize()
        hp = heads(m._last_head)
        diff = [k for k, (a, b) in enumerate(zip(hp, ref)) if not torch.equal(a, b)]
        if diff:
            bad += 1
            a, b = hp[diff[0]], ref[diff[0]]
            ne = (a != b).reshape(a.shape[0], -1).sum(1).tolist()
            print("round %d call %d: %d head tensors differ (%s); first: mismatching elements per image %s of %d, max |d| %.3g"
                  % (r, call, len(diff), ",".join(names[k] for k in diff), ne, a[0].numel(), float((a - b).abs().max())), flush=True)
    del m
print("n=%d splits=%d: %d bad calls in %d rounds x 3" % (n, S, bad, rounds))
