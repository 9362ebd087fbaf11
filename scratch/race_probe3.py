"""test_tta_packed_chunks_equal_the_reference_style_loop in a loop: the reference-style chunk loop (forward: sub-batch streams, a host
wait per chunk) against the packed chunks (splits=1, rotating streams, one host wait), fresh model every few rounds."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import dafne_amd.modeling  # noqa
from dafne_amd.config import load_cfg
from dafne_amd.registry import build_model
from dafne_amd.modeling.tta import OneStageRCNNWithTTA
from oracle import model as om
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = torch.device("cuda", 0)
cfg = load_cfg(os.path.join(R, "configs", "dota-1.5_r101.yaml"))
cfg.TEST.AUG.MIN_SIZES = [96, 128, 160, 224]
cfg.TEST.AUG.MAX_SIZE = 256
params = om.make_params(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.DAFNE.NUM_CLASSES, seed=21)
g = torch.Generator().manual_seed(9)
img = torch.randint(0, 256, (3, 128, 160), generator=g, dtype=torch.uint8).to(d)
ref = None
bad_a = bad_b = 0
for r in range(rounds):
    if r % 4 == 0:
        m = build_model(cfg); m.load_state_dict(params); m.to(d); m.invalidate()
        tta = OneStageRCNNWithTTA(cfg, m)
        aug, _ = tta._get_augmented_inputs({"image": img, "height": 128, "width": 160})
    a = tta._batch_inference(aug)
    b = tta._batch_inference_packed(aug)
    if ref is None:
        ref = [(x["instances"].pred_corners.clone(), x["instances"].scores.clone()) for x in a]
    da = [k for k, x in enumerate(a) if not (x["instances"].pred_corners.shape == ref[k][0].shape and torch.equal(x["instances"].pred_corners, ref[k][0]))]
    db = [k for k, x in enumerate(b) if not (x["instances"].pred_corners.shape == ref[k][0].shape and torch.equal(x["instances"].pred_corners, ref[k][0]))]
    if da: bad_a += 1; print("round %d: reference-style loop differs in views %s" % (r, da), flush=True)
    if db: bad_b += 1; print("round %d: packed chunks differ in views %s" % (r, db), flush=True)
print("%d rounds: chunk loop (forward) bad %d, packed bad %d" % (rounds, bad_a, bad_b))
