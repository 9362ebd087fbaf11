import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import oracle
from oracle import model as om, postprocess as opp
from test_gpu_model import build
from dafne_amd import postprocess as pp
from dafne_amd.modeling.dafne.dafne import head_levels
cfg, m, P = build("dota-1.5_r101.yaml", seed=5)
d = cfg.MODEL.DAFNE
g = torch.Generator().manual_seed(1)
ims = [torch.randint(0, 256, (3, 160, 192), generator=g, dtype=torch.uint8), torch.randint(0, 256, (3, 128, 150), generator=g, dtype=torch.uint8)]
inputs = [{"image": ims[0], "height": 320, "width": 384}, {"image": ims[1], "height": 128, "width": 150}]
out = m(inputs); torch.cuda.synchronize()
plan = m.plan(2, 160, 192); hp = plan.head
cand = pp.decode_levels(head_levels(hp, d.FPN_STRIDES), num_classes=16, pre_nms_thresh=0.05, pre_nms_topk=2000, thresh_with_ctr=False, sort_corners=False)
print("counts", cand.counts.tolist())
for rep in range(3):
    keep, nk = pp.select(cand, 0.1, 1000)
    torch.cuda.synchronize()
    for i in range(2):
        n = int(cand.counts[i])
        b = cand.corners[i, :n].cpu().numpy(); s = cand.scores[i, :n].cpu().numpy(); c = cand.classes[i, :n].cpu().numpy().astype(np.int64)
        exp = opp.batched_nms_poly(b, s, c, 0.1, fast=False)
        exp_fast = opp.batched_nms_poly(b, s, c, 0.1, fast=True)
        got = keep[i, :int(nk[i])].cpu().numpy()
        full = exp
        if len(full) > 1000:
            kth = np.sort(s[full])[len(full) - 1000]; full = full[s[full] >= kth]
        print(rep, i, "n", n, "got", len(got), "exp", len(full), "equal", np.array_equal(got, full), "fast==plain", np.array_equal(exp, exp_fast))
        if not np.array_equal(got, full):
            k = next(j for j in range(min(len(got), len(full))) if got[j] != full[j])
            print("   first diff at", k, got[k], full[k], "scores", s[got[k]], s[full[k]])
print("---- forward output vs oracle")
for i, o in enumerate(out):
    inst = o["instances"]
    levels = []
    for l in range(5):
        lg = hp.logits[l][i].cpu().numpy(); dc = hp.delta_ctr[l][i].cpu().numpy(); ce = hp.center[l][i].cpu().numpy()
        reg = ((np.tile(ce, (1, 1, 4)) + dc[..., :8]).astype(np.float32) * np.float32(hp.scales[l])).astype(np.float32)
        levels.append((np.transpose(lg, (2, 0, 1)), np.transpose(reg, (2, 0, 1)), np.transpose(dc[..., 8:9], (2, 0, 1))))
    det = opp.predict_proposals(levels, d.FPN_STRIDES, thresh=0.05, topk=2000, nms_thresh=0.1, post_topk=1000, thresh_with_ctr=False, sort_corners=False, fast=True)
    hw = tuple(ims[i].shape[1:])
    exp = opp.detector_postprocess(det, hw, (inputs[i]["height"], inputs[i]["width"]), hw)
    gc = inst.pred_classes.cpu().numpy(); gs = inst.scores.cpu().numpy()
    print(i, len(inst), exp["scores"].shape[0], "classes eq", np.array_equal(gc, exp["pred_classes"]), "scores eq", np.array_equal(gs, exp["scores"]))
    if not np.array_equal(gc, exp["pred_classes"]):
        k = next(j for j in range(len(gc)) if gc[j] != exp["pred_classes"][j])
        print("  first diff", k, gc[k-2:k+3], exp["pred_classes"][k-2:k+3], gs[k-2:k+3], exp["scores"][k-2:k+3])
        print("  det pre-postprocess n", det["scores"].shape[0])
