# NMS path counters + per-kernel view of the post-process on the BENCH pipeline's own candidates (random-weight R101)
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, numpy as np
import bench
from dafne_amd import _lib, postprocess as pp
from dafne_amd.modeling.dafne.dafne import head_levels
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(dev)
rows, counts = model.detect_packed(batch); torch.cuda.synchronize()
plan = model.plan(8, 1024, 1024)
d = cfg.MODEL.DAFNE
cand = pp.decode_levels(head_levels(plan.head, d.FPN_STRIDES), num_classes=15, pre_nms_thresh=d.INFERENCE_TH_TEST, pre_nms_topk=d.PRE_NMS_TOPK_TEST,
                        thresh_with_ctr=d.THRESH_WITH_CTR, sort_corners=d.SORT_CORNERS)
print("candidates per image", cand.counts.tolist())
L = _lib.load()
n, m = cand.n, cand.m_cap
keep = torch.empty(n, m, dtype=torch.int64, device=dev); nk = torch.zeros(n, dtype=torch.int32, device=dev)
nb = L.dafne_poly_nms_workspace_bytes(n, m); ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
def run():
    _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(cand.corners), _lib.ptr(cand.scores), _lib.ptr(cand.classes), _lib.ptr(cand.counts), n, m, 0.1, 1000,
                                                  _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nb, _lib.current_stream()))
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
off = L.dafne_poly_nms_stats_offset(n, m, 0)
st = ws[off:off + 16 * n].view(torch.int32).reshape(n, 4).sum(0).tolist()
print("select: %.3f ms/call; paths fast+ %d fast- %d exact %d ovf %d; kept %s" % (e0.elapsed_time(e1) / 50, *st, nk.tolist()))
# geometry of the candidates: how many are strictly convex with edges >= 1 px
c = cand.corners[0, :int(cand.counts[0])].cpu().numpy().reshape(-1, 4, 2).astype(np.float64)
def cross(a, b): return a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
area = 0.5 * sum(cross(c[:, i], c[:, (i + 1) % 4]) for i in range(4))
cc = np.where(area[:, None, None] < 0, c[:, ::-1], c)
ok = np.ones(len(c), bool)
for i in range(4):
    e, f = cc[:, (i + 1) % 4] - cc[:, i], cc[:, (i + 2) % 4] - cc[:, (i + 1) % 4]
    ok &= (cross(e, f) > 1e-3) & ((e ** 2).sum(1) >= 1.0)
print("image 0: strictly convex candidates %.1f %%, |area| median %.1f px^2, zero-ish area %d" % (100 * ok.mean(), np.median(np.abs(area)), (np.abs(area) < 16).sum()))
k = int(cand.counts[0])
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(R, "gpurun_out", "pipe_cand_img0.npz"), corners=cand.corners[0, :k].cpu().numpy(), scores=cand.scores[0, :k].cpu().numpy(),
                    classes=cand.classes[0, :k].cpu().numpy(), levels=cand.levels[0, :k].cpu().numpy())
