import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, time
import bench
from dafne_amd import engine, _lib, postprocess as pp
from dafne_amd.modeling.dafne.dafne import head_levels
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(50, dev)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8).to(dev)
rows, counts = model.detect_packed(batch); torch.cuda.synchronize()
plan = model.plan(8, 1024, 1024)
outs = model.proposal_generator.dafne_outputs
cand = pp.decode_levels(head_levels(plan.head, [8,16,32,64,128]), num_classes=15, pre_nms_thresh=0.05, pre_nms_topk=2000, thresh_with_ctr=True, sort_corners=True)
print("cand counts", cand.counts.tolist())
L = _lib.load()
n, m = cand.n, cand.m_cap
keep = torch.empty(n, m, dtype=torch.int64, device=dev); nk = torch.zeros(n, dtype=torch.int32, device=dev)
nbytes = L.dafne_poly_nms_workspace_bytes(n, m); ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
def run():
    _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(cand.corners), _lib.ptr(cand.scores), _lib.ptr(cand.classes), _lib.ptr(cand.counts), n, m, 0.1, 1000, _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, _lib.current_stream()))
run(); torch.cuda.synchronize()
meta = ws[:n*16].view(torch.int32).reshape(n, 4).cpu()
print("clipped pairs per image", meta[:, 2].tolist(), "kept", nk.tolist())
for lv in range(5):
    sel = cand.levels[0, :int(cand.counts[0])] == lv
    hb = cand.hbox[0, :int(cand.counts[0])][sel]
    print("level", lv, "n", int(sel.sum()), "mean box w,h", float((hb[:,2]-hb[:,0]).mean()), float((hb[:,3]-hb[:,1]).mean()))
t=time.time()
for _ in range(10): run()
torch.cuda.synchronize(); print("select ms/call", (time.time()-t)*100)
