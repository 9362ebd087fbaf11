"""test_batch_invariance_and_determinism in a loop: a fresh model, m(inputs) twice with 3 HOST images (sub-batches 1+1+1, both calls
eager: one per plan set), compared with each other; on a mismatch the head outputs of the two plan sets are compared."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import dafne_amd.modeling  # noqa
from dafne_amd.config import load_cfg
from dafne_amd.registry import build_model
from oracle import model as om
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = torch.device("cuda", 0)
cfg = load_cfg(os.path.join(R, "configs", "dota-1.0_r50.yaml"))
params = om.make_params(50, cfg.MODEL.DAFNE.NUM_CLASSES, seed=7)
g = torch.Generator().manual_seed(2)
ims = [torch.randint(0, 256, (3, 128, 128), generator=g, dtype=torch.uint8) for _ in range(nimg)]
inputs = [{"image": im, "height": 128, "width": 128} for im in ims]
names = ["logits%d" % i for i in range(5)] + ["center%d" % i for i in range(5)] + ["delta%d" % i for i in range(5)]
def heads(ho):
    return [t.clone() for lst in (ho.logits, ho.center, ho.delta_ctr) for t in lst]
bad = 0
keep = []
def poison(seed):
    """Fill the caching allocator's free lists with garbage: blocks of many sizes, random bytes (a bf16 / fp32 reader sees huge values
    and NaNs), then released -- torch.empty buffers of the next model come back dirty, as they do after other tests ran in the process."""
    gg = torch.Generator(device=d).manual_seed(seed)
    blocks = []
    for k in range(10, 29):
        for rep in range(3 if k < 24 else 1):
            t = torch.empty((1 << k) + 512 * rep, dtype=torch.uint8, device=d)
            t.random_(0, 256, generator=gg)
            blocks.append(t)
    torch.cuda.synchronize()
    del blocks
for r in range(rounds):
    if os.environ.get("POISON", "1") == "1":
        poison(r)
    m = build_model(cfg); m.load_state_dict(params); m.to(d); m.invalidate()
    if os.environ.get("KEEP"): keep.append(m)
    a = m(inputs); ha = heads(m._last_head)
    b = m(inputs); hb = heads(m._last_head)
    same = all(torch.equal(x["instances"].pred_corners, y["instances"].pred_corners) for x, y in zip(a, b))
    if not same:
        bad += 1
        diff = [k for k, (p, q) in enumerate(zip(ha, hb)) if not torch.equal(p, q)]
        msg = ""
        if diff:
            p, q = ha[diff[0]], hb[diff[0]]
            msg = "%s: mismatching elements per image %s of %d, max |d| %.3g" % (names[diff[0]], (p != q).reshape(p.shape[0], -1).sum(1).tolist(), p[0].numel(), float((p - q).abs().max()))
        print("round %d: the two calls differ; head tensors differing: %s; %s" % (r, [names[k] for k in diff], msg), flush=True)
print("%d of %d rounds: m(inputs) twice gave different detections" % (bad, rounds))
