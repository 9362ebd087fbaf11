"""The post-process alone (decode -> select -> gather on the side stream) on FIXED head outputs, while convolutions of another shape keep
the three compute streams busy: every stage's output against the idle-GPU reference.  Finds the first stage that differs."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_inference_loop as T
from dafne_amd.modeling.dafne.dafne import head_levels
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
load = int(os.environ.get("LOAD", "3"))
cfg, m = T._gpu_model()
g = torch.Generator().manual_seed(21)
img = torch.randint(0, 256, (1, 3, 512, 640), generator=g, dtype=torch.uint8).cuda()
bg = torch.randint(0, 256, (1, 3, 448, 576), generator=g, dtype=torch.uint8).cuda()
m.detect_packed(img); torch.cuda.synchronize()
plan = m.plan(1, 512, 640)
outs, strides = m.proposal_generator.dafne_outputs, m.proposal_generator.fpn_strides
sizes = m._dev_const([(512, 640, 512, 640, 512, 640)], torch.float32, (1, 6))
side = torch.cuda.Stream()
def post():
    with torch.cuda.stream(side):
        hsnap = [t.clone() for lst in (plan.head.center, plan.head.delta_ctr, plan.head.logits) for t in lst]
        cand = outs.decode_packed(head_levels(plan.head, strides))
        hsnap2 = [t.clone() for lst in (plan.head.center, plan.head.delta_ctr, plan.head.logits) for t in lst]
        snap = [t.clone() for t in (cand.counts, cand.scores, cand.classes, cand.corners)]
        extra = (cand.locs.clone(), cand.levels.clone())
        from dafne_amd import postprocess as pp
        keep, nk = pp.select(cand, outs.nms_thresh, outs.post_nms_topk_test)
        snap += [nk.clone(), keep.clone()]
        rows, cnt = pp.gather(cand, keep, nk, sizes=sizes, k_cap=outs.packed_k_cap(), scale_corners=True)
        snap += [cnt.clone(), rows.clone(), hsnap, hsnap2, extra]
    return snap
names = ["cand.counts", "cand.scores", "cand.classes", "cand.corners", "num_keep", "keep", "count", "rows"]
ref = post(); torch.cuda.synchronize()
def same(a, b, k):
    if names[k] in ("cand.scores", "cand.classes", "cand.corners"):
        n = int(ref[0][0]); return torch.equal(a[0, :n], b[0, :n])
    if names[k] == "keep":
        n = int(ref[4][0]); return torch.equal(a[0, :n], b[0, :n])
    if names[k] == "rows":
        n = int(ref[6][0]); return torch.equal(a[0, :n], b[0, :n])
    return torch.equal(a, b)
first = {}
rot = 0
pend = []
for it in range(iters):
    for _ in range(load):
        m.detect_packed(bg, pipelined=True, splits=1, defer=True, stream_offset=rot); rot = (rot + 1) % 3
    pend.append(post())
    if len(pend) == 8 or it == iters - 1:
        torch.cuda.synchronize()
        for sn in pend:
            hb = [i for i, (a, b) in enumerate(zip(sn[8], ref[8])) if not torch.equal(a, b)]
            ha = [i for i, (a, b) in enumerate(zip(sn[9], ref[9])) if not torch.equal(a, b)]
            if hb or ha:
                first["HEAD TENSORS before decode %s after %s" % (hb, ha)] = first.get("HEAD TENSORS before decode %s after %s" % (hb, ha), 0) + 1
            for k in range(len(names)):
                if not same(sn[k], ref[k], k):
                    first[names[k]] = first.get(names[k], 0) + 1
                    if sum(first.values()) <= 12 and names[k].startswith("cand."):
                        n = int(ref[0][0])
                        a, b = sn[3][0, :n], ref[3][0, :n]
                        rowsd = (a != b).any(1).nonzero().flatten().tolist()
                        print("  differing candidate rows (%d) %s of %d" % (len(rowsd), rowsd[:80], n))
                        lv = sn[0]  # counts only; level offsets from cand.levels not snapshotted
                        for r in rowsd[:1]:
                            lvl = int(sn[10][1][0, r]); stride = strides[lvl]
                            lx, ly = [float(v) for v in sn[10][0][0, r]]
                            x, y = int((lx - stride // 2) / stride), int((ly - stride // 2) / stride)
                            ce = plan.head.center[lvl][0, y, x].tolist(); dl = plan.head.delta_ctr[lvl][0, y, x, :8].tolist()
                            sc = float(plan.head.scales[lvl])
                            raw = [((lx if j % 2 == 0 else ly) + ((ce[j % 2] + dl[j]) * sc) * stride) for j in range(8)]
                            print("   level %d loc (%d, %d): raw corner order %s" % (lvl, x, y, [round(v, 3) for v in raw]))
                            print("   row %d score %.9g / %.9g class %d / %d\n      got %s\n      ref %s" % (r, float(sn[1][0, r]), float(ref[1][0, r]), int(sn[2][0, r]), int(ref[2][0, r]),
                                  [round(v, 3) for v in a[r].tolist()], [round(v, 3) for v in b[r].tolist()]))
                    break
        pend = []
print("%d post-process runs beside %d background calls each: first differing stage -> count: %s" % (iters, load, first or "none differ"))
