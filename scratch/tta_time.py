"""BASELINE config 4 (DOTA-1.5 R101-FPN, multi-scale + flip TTA: 9 sizes x 3 views = 27 forward passes and one merged
rotated NMS per image): seconds per 1024x1024 image, after a warm-up image (plans for the 9 sizes are built once)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from dafne_amd.modeling.tta import OneStageRCNNWithTTA
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev, cfgname="dota-1.5_r101.yaml", cls_prior=-1.5)
cfg.TEST.AUG.ENABLED = True
tta = OneStageRCNNWithTTA(cfg, model)
print("TTA sizes", cfg.TEST.AUG.MIN_SIZES, "max", cfg.TEST.AUG.MAX_SIZE)
g = torch.Generator().manual_seed(0)
imgs = [torch.randint(0, 256, (3, 1024, 1024), generator=g, dtype=torch.uint8).to(dev) for _ in range(4)]
out = tta([{"image": imgs[0], "height": 1024, "width": 1024}])
torch.cuda.synchronize()
t0 = time.perf_counter()
for im in imgs[1:]:
    out = tta([{"image": im, "height": 1024, "width": 1024}])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("TTA: %.1f ms per image (27 views + merge), %d detections" % (1e3 * dt, len(out[0]["instances"])))
from itertools import count
aug, tf = tta._get_augmented_inputs({"image": imgs[1], "height": 1024, "width": 1024})
outs = tta._batch_inference_packed(aug)
print("boxes entering the merge:", sum(len(o["instances"]) for o in outs))
def T(f, *a):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t)
inp = {"image": imgs[2], "height": 1024, "width": 1024}
for rep in range(2):
    (aug, tf), t1 = T(tta._get_augmented_inputs, inp)
    outs, t2 = T(tta._batch_inference_packed, aug)
    def corners():
        lst = []
        from dafne_amd.structures import Instances
        for output, tfm in zip(outs, tf):
            inst = output["instances"]; pc = inst.pred_corners; n = pc.shape[0]
            orig = tfm.inverse().apply_coords(pc.reshape(-1, 2).to(torch.float64)).reshape(n, 8).to(pc.dtype)
            r = Instances(inst.image_size); r.scores = inst.scores; r.centerness = inst.centerness; r.pred_corners = orig; r.pred_classes = inst.pred_classes
            lst.append(r)
        return Instances.cat(lst)
    inst, t3 = T(corners)
    res, t4 = T(tta._merge_detections, inst)
    print("views %.1f ms | 9 chunks %.1f ms | inverse transforms %.1f ms | merge %.1f ms" % (t1, t2, t3, t4))
# dense part alone: the 9 chunk plans back to back, no post-process
def dense_only():
    for i in range(0, 27, 3):
        b = torch.stack([x["image"] for x in aug[i:i + 3]])
        n, _, h, w = b.shape
        model.plan(n, (h + 31) // 32 * 32, (w + 31) // 32 * 32).run()
_, t5 = T(dense_only); _, t5 = T(dense_only)
print("dense plans of the 9 chunks alone: %.1f ms" % t5)
