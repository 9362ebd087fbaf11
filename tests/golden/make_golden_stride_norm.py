"""One more predict_proposals fixture from the reference's own dafne_outputs.py: MODEL.DAFNE.ENABLE_FPN_STRIDE_NORM false
(dafne_outputs.py:771-774: the regression is NOT multiplied by the level's stride).  No released config sets it, the key exists
(config/defaults.py:73).  Same inputs layout as make_golden.gen_predict; written to predict_no_stride_norm.npz.

    python tests/golden/make_golden_stride_norm.py        (build container only: imports /root/reference under stubs)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    assert os.path.isdir(mg.REF), "reference tree not present: fixtures can only be made in the build container"
    mg.install_stubs()
    mg.load_ref("dafne.utils.sort_corners")
    mg.load_ref("dafne.layers.deform_conv")
    mg.load_ref("dafne.modeling.losses.utils")
    mg.load_ref("dafne.modeling.losses.smooth_l1")
    mg.load_ref("dafne.modeling.nms.nms")
    outputs_mod = mg.load_ref("dafne.modeling.dafne.dafne_outputs")
    dafne_mod = mg.load_ref("dafne.modeling.dafne.dafne")
    rng = np.random.default_rng(405)
    strides = [8, 16, 32, 64, 128]
    sizes = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]
    res = {}
    for name, cfgfile, over in [("d10_nsn", "dota-1.0_r101_ms.yaml", {"ENABLE_FPN_STRIDE_NORM": False}),
                                ("d15_nsn", "dota-1.5_r101_ms.yaml", {"ENABLE_FPN_STRIDE_NORM": False})]:
        cfg = mg.load_cfg(cfgfile, **over)
        C = cfg.MODEL.DAFNE.NUM_CLASSES
        outs = outputs_mod.DAFNeOutputs(cfg)
        outs.eval()
        N = 2
        logits, regs, ctrs, locs = [], [], [], []
        for (h, w), s in zip(sizes, strides):
            logits.append(torch.from_numpy(rng.normal(-3.0, 2.0, (N, C, h, w)).astype(np.float32)))
            # without the stride multiply the regression is in pixels: scaled so that boxes overlap as in the other fixtures
            regs.append(torch.from_numpy((rng.normal(0, 1.5, (N, 8, h, w)) * s).astype(np.float32)))
            ctrs.append(torch.from_numpy(rng.normal(0, 2.0, (N, 1, h, w)).astype(np.float32)))
            locs.append(dafne_mod.compute_locations(h, w, s, "cpu"))
        with torch.no_grad():
            boxlists = outs.predict_proposals(logits, regs, ctrs, locs, [(256, 256)] * N, [])
        res[name + "_cfg"] = np.array([C, cfg.MODEL.DAFNE.PRE_NMS_TOPK_TEST, cfg.MODEL.DAFNE.POST_NMS_TOPK_TEST,
                                       int(cfg.MODEL.DAFNE.THRESH_WITH_CTR), int(cfg.MODEL.DAFNE.SORT_CORNERS)], np.int64)
        res[name + "_thr"] = np.array([cfg.MODEL.DAFNE.INFERENCE_TH_TEST, cfg.MODEL.DAFNE.NMS_TH])
        for l in range(5):
            res["%s_logits%d" % (name, l)] = logits[l].numpy()
            res["%s_reg%d" % (name, l)] = regs[l].numpy()
            res["%s_ctr%d" % (name, l)] = ctrs[l].numpy()
        for i, bl in enumerate(boxlists):
            f = bl.get_fields()
            res["%s_im%d_pred_boxes" % (name, i)] = f["pred_boxes"].tensor.numpy()
            for k in ("pred_corners", "scores", "centerness", "pred_classes", "locations", "fpn_levels"):
                res["%s_im%d_%s" % (name, i, k)] = f[k].numpy()
            print("predict", name, "im", i, "dets", len(bl))
    np.savez_compressed(os.path.join(HERE, "predict_no_stride_norm.npz"), **res)


if __name__ == "__main__":
    main()
