import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, numpy as np
from oracle import model as om
import dafne_amd.modeling
from dafne_amd.config import load_cfg
from dafne_amd.registry import build_model
cfg = load_cfg(os.path.join(R, "configs", "dota-1.0_r50.yaml"))
m = build_model(cfg); P = om.make_params(50, 15, seed=3); m.load_state_dict(P); m.to("cuda:0"); m.invalidate()
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (2, 3, 128, 160), generator=g, dtype=torch.uint8)
x, _ = om.preprocess([img[0], img[1]], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
taps = {}; taps32 = {}
with torch.no_grad():
    f_e = om.backbone_forward(P, x, 50, emulate_bf16=True, taps=taps)
    f_32 = om.backbone_forward(P, x, 50, taps=taps32)
feats = m.backbone(x.cuda())
plan = list(m.backbone._plans.values())[0]
rel = lambda a, b: float((a - b).norm() / b.norm())
for k in ("res3", "res4", "res5"):
    e = plan.stage_feats[k].nchw_float().cpu()
    print(k, "engine vs emu", rel(e, taps[k]), " emu vs fp32", rel(taps[k], taps32[k]), " engine vs fp32", rel(e, taps32[k]))
for k in ("p3", "p4", "p5", "p6", "p7"):
    print(k, "engine vs emu", rel(feats[k].cpu(), f_e[k]), " emu vs fp32", rel(f_e[k], f_32[k]), "engine vs fp32", rel(feats[k].cpu(), f_32[k]))
