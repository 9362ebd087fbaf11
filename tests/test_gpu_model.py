"""GPU parity of the whole dense path (ResNet-FPN + DAFNe head) and of the
end-to-end detector vs the torch oracle.

Tolerances (bf16 engine vs the oracle), relative L2 error per tensor:
  * vs the pure fp32 oracle: < 2.5e-2, and no worse than 1.5x the error of the
    oracle's own bf16 emulation (rounding at the engine's store points) -- i.e. the
    engine sits at the bf16 noise floor of a 50-layer network (~1.1e-2 measured)
  * vs the bf16 emulation: < 2.5e-2.  This cannot be much tighter: two bf16
    pipelines that differ only in fp32 summation order decorrelate after a few
    layers (a perturbation eps flips roundings with probability eps/ulp, adding
    sqrt(eps*ulp) error: fixed point eps = ulp), so they end up one noise floor
    apart.  Single-layer parity is tight (tests/test_gpu_conv.py, 2 bf16 ulps).
  * end-to-end post-process given the engine's own head outputs: NMS keys bit-exact,
    coordinates / scores within 1e-3 (BASELINE.json)."""
import os

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import postprocess as opp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    return torch.device("cuda", 0)


def build(cfgname, seed, stride_norm=True):
    import dafne_amd.modeling  # noqa: F401
    from dafne_amd.config import load_cfg
    from dafne_amd.registry import build_model
    cfg = load_cfg(os.path.join(ROOT, "configs", cfgname))
    cfg.MODEL.DAFNE.ENABLE_FPN_STRIDE_NORM = bool(stride_norm)
    m = build_model(cfg)
    P = om.make_params(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.DAFNE.NUM_CLASSES, seed=seed)
    m.load_state_dict(P)
    m.to(dev())
    m.invalidate()
    return cfg, m, P


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def test_backbone_and_head_vs_oracle():
    cfg, m, P = build("dota-1.0_r50.yaml", seed=3)
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (2, 3, 128, 160), generator=g, dtype=torch.uint8)
    x, _ = om.preprocess([img[0], img[1]], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
    with torch.no_grad():
        f_e = om.backbone_forward(P, x, 50, emulate_bf16=True)
        f_32 = om.backbone_forward(P, x, 50)
        h_e = om.head_forward(P, [f_e[k] for k in ("p3", "p4", "p5", "p6", "p7")], emulate_bf16=True)
    feats = m.backbone(x.to(dev()))
    for k in ("p3", "p4", "p5", "p6", "p7"):
        e_emu, e_32, floor = rel(feats[k].cpu(), f_e[k]), rel(feats[k].cpu(), f_32[k]), rel(f_e[k], f_32[k])
        assert e_emu < 2.5e-2 and e_32 < 2.5e-2 and e_32 < 1.5 * floor, (k, e_emu, e_32, floor)
    # head on the ORACLE's (bf16-rounded) features, through the reference-signature forward
    head = m.proposal_generator.dafne_head
    logits, regs, centers, _, ctrs, _, _ = head(None, [f_e[k].to(dev()) for k in ("p3", "p4", "p5", "p6", "p7")])
    for l in range(5):
        assert rel(logits[l].cpu(), h_e[0][l]) < 2.5e-2, ("logits", l, rel(logits[l].cpu(), h_e[0][l]))
        assert rel(regs[l].cpu(), h_e[1][l]) < 2.5e-2, ("reg", l, rel(regs[l].cpu(), h_e[1][l]))
        assert rel(centers[l].cpu(), h_e[2][l]) < 2.5e-2, ("center", l)
        assert rel(ctrs[l].cpu(), h_e[3][l]) < 2.5e-2, ("ctr", l)


@pytest.mark.parametrize("cfgname,stride_norm", [("dota-1.0_r50.yaml", True), ("dota-1.5_r101.yaml", True), ("dota-1.0_r50.yaml", False)])
def test_end_to_end_detections_vs_oracle_postprocess(cfgname, stride_norm):
    """OneStageDetector.forward (fused path) vs: engine head outputs -> numpy oracle
    decode / NMS / cap / detector_postprocess.  stride_norm False: MODEL.DAFNE.ENABLE_FPN_STRIDE_NORM off
    (dafne_outputs.py:771-774; no released config), serial and sub-batch-stream layouts."""
    from dafne_amd.modeling.dafne.dafne import head_levels
    cfg, m, P = build(cfgname, seed=5, stride_norm=stride_norm)
    d = cfg.MODEL.DAFNE
    g = torch.Generator().manual_seed(1)
    ims = [torch.randint(0, 256, (3, 160, 192), generator=g, dtype=torch.uint8),
           torch.randint(0, 256, (3, 128, 150), generator=g, dtype=torch.uint8)]
    inputs = [{"image": ims[0], "height": 320, "width": 384}, {"image": ims[1], "height": 128, "width": 150}]
    out = m(inputs)
    torch.cuda.synchronize()
    hp = m._last_head               # the head outputs of that call (round 5: forward() of >= 2 images runs on the sub-batch layout)
    strides = d.FPN_STRIDES
    for i, o in enumerate(out):
        inst = o["instances"]
        levels = []
        for l in range(5):
            lg = hp.logits[l][i].cpu().numpy()
            dc = hp.delta_ctr[l][i].cpu().numpy()
            ce = hp.center[l][i].cpu().numpy()
            reg = ((np.tile(ce, (1, 1, 4)) + dc[..., :8]).astype(np.float32) * np.float32(hp.scales[l])).astype(np.float32)
            levels.append((np.transpose(lg, (2, 0, 1)), np.transpose(reg, (2, 0, 1)), np.transpose(dc[..., 8:9], (2, 0, 1))))
        det = opp.predict_proposals(levels, strides, thresh=d.INFERENCE_TH_TEST, topk=d.PRE_NMS_TOPK_TEST,
                                    nms_thresh=d.NMS_TH, post_topk=d.POST_NMS_TOPK_TEST,
                                    thresh_with_ctr=d.THRESH_WITH_CTR, sort_corners=d.SORT_CORNERS, fast=True,
                                    stride_norm=stride_norm)
        hw = tuple(ims[i].shape[1:])
        exp = opp.detector_postprocess(det, hw, (inputs[i]["height"], inputs[i]["width"]), hw)
        assert len(inst) == exp["scores"].shape[0] and len(inst) > 0
        assert inst.image_size == (inputs[i]["height"], inputs[i]["width"])
        # The oracle's sigmoid/sqrt (numpy) and the kernel's (expf/sqrtf) differ by an
        # ulp now and then, which may swap two detections whose scores are 1 ulp apart:
        # match detections by their (level, location, class) key, then compare values.
        gs = inst.scores.cpu().numpy()
        assert np.all(np.diff(gs) <= 0)                                  # descending score
        sx = inputs[i]["width"] / hw[1]
        sy = inputs[i]["height"] / hw[0]

        def keys(levels_, locs_, classes_):
            x = np.rint(locs_[:, 0] / sx).astype(np.int64)
            y = np.rint(locs_[:, 1] / sy).astype(np.int64)
            return levels_.astype(np.int64) * (1 << 40) + y * (1 << 24) + x * 64 + classes_.astype(np.int64)
        gk = keys(inst.fpn_levels.cpu().numpy(), inst.locations.cpu().numpy(), inst.pred_classes.cpu().numpy())
        ek = keys(exp["fpn_levels"], exp["locations"], exp["pred_classes"])
        assert len(np.unique(gk)) == len(gk)
        assert np.array_equal(np.sort(gk), np.sort(ek)), "different detection sets"
        go, eo = np.argsort(gk), np.argsort(ek)
        assert np.abs(gs[go] - exp["scores"][eo]).max() < 1e-6
        assert np.abs(inst.pred_corners.cpu().numpy()[go] - exp["pred_corners"][eo]).max() < 1e-3
        assert np.abs(inst.pred_boxes.tensor.cpu().numpy()[go] - exp["pred_boxes"][eo]).max() < 1e-3
        assert np.abs(inst.locations.cpu().numpy()[go] - exp["locations"][eo]).max() < 1e-3
        # positions may differ only where scores are (nearly) tied
        moved = np.nonzero(gk != ek)[0]
        assert all(abs(gs[j] - exp["scores"][j]) < 1e-6 for j in moved)
    if not stride_norm:
        # the sub-batch-stream layout decodes through the same switch: same rows as the serial call
        batch, valid, out_hw = m._pack_inputs(inputs)
        r0, c0 = m.detect_packed(batch, valid_hw=valid, out_hw=out_hw)
        torch.cuda.synchronize()
        r1, c1 = m.detect_packed(batch, valid_hw=valid, out_hw=out_hw, pipelined=True, splits=2)
        torch.cuda.synchronize()
        assert torch.equal(c0, c1)
        for i in range(2):
            assert torch.equal(r0[i, :int(c0[i])], r1[i, :int(c0[i])])


def test_checkpoint_file_to_engine_vs_oracle(tmp_path):
    """F1 (tools/plain_train_net.py:576-578): a detectron2-style `.pth` ({"model": state_dict}) and an MSRA-style
    Caffe2-named `.pkl` trunk go file -> load_weights -> packed engine weights -> features, and match the oracle run on
    the same tensors (pkl: the running statistics the trunk does not ship keep d2's defaults, mean 0 / var 1 - eps)."""
    import pickle
    import dafne_amd.modeling  # noqa: F401
    from test_model_cpu import _d2_to_c2
    from dafne_amd.checkpoint import load_weights
    from dafne_amd.config import load_cfg
    from dafne_amd.registry import build_model
    cfg = load_cfg(os.path.join(ROOT, "configs", "dota-1.0_r50.yaml"))
    P = om.make_params(50, 15, seed=31)
    g = torch.Generator().manual_seed(8)
    img = torch.randint(0, 256, (1, 3, 96, 128), generator=g, dtype=torch.uint8)
    x, _ = om.preprocess([img[0]], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
    # ---- .pth
    pth = str(tmp_path / "model_final.pth")
    torch.save({"model": P, "iteration": 90000}, pth)
    m = build_model(cfg)
    m.to(dev())
    missing, unexpected = load_weights(m, pth, strict=True)
    assert not missing and not unexpected
    with torch.no_grad():
        f_e = om.backbone_forward(P, x, 50, emulate_bf16=True)
        f_32 = om.backbone_forward(P, x, 50)
    feats = m.backbone(x.to(dev()))
    for k in ("p3", "p4", "p5", "p6", "p7"):
        e_emu, e_32, floor = rel(feats[k].cpu(), f_e[k]), rel(feats[k].cpu(), f_32[k]), rel(f_e[k], f_32[k])
        assert e_emu < 2.5e-2 and e_32 < 2.5e-2 and e_32 < 1.5 * floor, (k, e_emu, e_32, floor)
    out = m([{"image": img[0], "height": 96, "width": 128}])[0]["instances"]
    assert len(out) > 0
    # ---- Caffe2-named .pkl trunk over the same head / FPN
    bu = "backbone.bottom_up."
    c2 = {_d2_to_c2(k[len(bu):]): v.numpy() for k, v in P.items()
          if k.startswith(bu) and not k.endswith(("running_mean", "running_var"))}
    c2["fc1000_w"] = np.zeros((1000, 2048), np.float32)
    pkl = str(tmp_path / "R-50.pkl")
    with open(pkl, "wb") as f:
        pickle.dump({"model": c2, "matching_heuristics": True}, f)
    m2 = build_model(cfg)
    m2.to(dev())
    load_weights(m2, {k: v for k, v in P.items() if not k.startswith(bu)})
    missing, unexpected = load_weights(m2, pkl)
    assert not unexpected and all(k.endswith(("running_mean", "running_var")) for k in missing if k.startswith(bu))
    P2 = dict(P)
    for k in P:
        if k.startswith(bu) and k.endswith("running_mean"):
            P2[k] = torch.zeros_like(P[k])
        if k.startswith(bu) and k.endswith("running_var"):
            P2[k] = torch.ones_like(P[k]) - 1e-5
    with torch.no_grad():
        f2_e = om.backbone_forward(P2, x, 50, emulate_bf16=True)
        f2_32 = om.backbone_forward(P2, x, 50)
    feats2 = m2.backbone(x.to(dev()))
    for k in ("p3", "p4", "p5", "p6", "p7"):
        e_emu, e_32, floor = rel(feats2[k].cpu(), f2_e[k]), rel(feats2[k].cpu(), f2_32[k]), rel(f2_e[k], f2_32[k])
        assert e_emu < 2.5e-2 and e_32 < 2.5e-2 and e_32 < 1.5 * floor, (k, e_emu, e_32, floor)
        assert rel(f2_32[k], f_32[k]) > 1e-3       # the two trunks really differ (statistics vs defaults)


def test_batch_invariance_and_determinism():
    cfg, m, P = build("dota-1.0_r50.yaml", seed=7)
    g = torch.Generator().manual_seed(2)
    ims = [torch.randint(0, 256, (3, 128, 128), generator=g, dtype=torch.uint8) for _ in range(3)]
    inputs = [{"image": im, "height": 128, "width": 128} for im in ims]
    a = m(inputs)
    b = m(inputs)
    single = [m([inp])[0] for inp in inputs]
    for x, y, z in zip(a, b, single):
        assert torch.equal(x["instances"].pred_corners, y["instances"].pred_corners)       # run-to-run identical
        assert torch.equal(x["instances"].scores, y["instances"].scores)
        assert torch.equal(x["instances"].pred_corners, z["instances"].pred_corners)       # batch-size independent


def test_split_k_workspace_is_zeroed_when_the_plan_is_built():
    """engine.WrWorkspace (round 5): the split-K workspace of a plan's conv_wr launches exists -- zeroed, on the build stream -- as
    soon as the plan does.  Round 4 zeroed it at the first launch, on torch's current stream instead of the launch's: with small
    images the fill raced the first conv_wr kernels of a sub-batch plan (one call of garbage features in ~1 of 8 processes)."""
    from dafne_amd import engine
    cfg, m, P = build("dota-1.0_r50.yaml", seed=7)
    plan = m.plan(1, 128, 128)
    assert any(isinstance(c, engine.WrCall) for c in plan.calls)
    assert plan.wr_ws.t is not None and not bool(plan.wr_ws.t.any())
    img = torch.randint(0, 256, (3, 3, 128, 128), dtype=torch.uint8, device=dev())
    m.detect_packed(img, pipelined=True, splits=3)
    torch.cuda.synchronize()
    for sets in next(iter(m._pipe.values()))["plans"]:
        for p in sets:
            assert p.wr_ws.t is not None
            assert not bool(p.wr_ws.t[:64 * 1024].any())          # the arrival tickets are zero again after every launch


def test_tta_merge_vs_oracle():
    """OneStageRCNNWithTTA: per-view detections come from the engine; the inverse
    transforms + merged NMS + cap are checked against the numpy/C oracle."""
    from dafne_amd.modeling.tta import DotaDatasetMapperTTA, OneStageRCNNWithTTA
    cfg, m, P = build("dota-1.5_r101.yaml", seed=9)
    cfg.TEST.AUG.MIN_SIZES = [96, 128, 160]
    cfg.TEST.AUG.MAX_SIZE = 192
    g = torch.Generator().manual_seed(4)
    img = torch.randint(0, 256, (3, 128, 160), generator=g, dtype=torch.uint8)
    tta = OneStageRCNNWithTTA(cfg, m)
    inp = {"image": img, "height": 128, "width": 160}
    out = tta([inp])[0]["instances"]
    # expected: run the views one by one, invert with the oracle, merge with the oracle
    views = DotaDatasetMapperTTA(cfg)({**inp, "image": img.cuda()})
    assert len(views) == 9
    dets = []
    for k, v in enumerate(views):
        r = m.inference([{kk: vv for kk, vv in v.items() if kk != "transforms"}], None, do_postprocess=False)[0]["instances"]
        nh, nw = v["image"].shape[1:]
        hf, vf = (k % 3 == 1), (k % 3 == 2)
        c = opp.tta_invert_corners(r.pred_corners.cpu().numpy(), (160 / nw, 128 / nh), hf, vf, (nh, nw))
        dets.append({"pred_corners": c, "scores": r.scores.cpu().numpy(), "centerness": r.centerness.cpu().numpy(),
                     "pred_classes": r.pred_classes.cpu().numpy()})
    d = cfg.MODEL.DAFNE
    exp = opp.select_over_all_levels(opp.cat(dets), d.NMS_TH, d.POST_NMS_TOPK_TEST, fast=True)
    assert len(out) == exp["scores"].shape[0] and len(out) > 0
    assert np.array_equal(out.pred_classes.cpu().numpy(), exp["pred_classes"])
    assert np.abs(out.scores.cpu().numpy() - exp["scores"]).max() == 0
    assert np.abs(out.pred_corners.cpu().numpy() - exp["pred_corners"]).max() < 1e-3


@pytest.mark.parametrize("name", ["d15", "d10_pre", "d15_cap"])
def test_tta_merge_vs_reference_fixture(golden, name):
    """The TTA stage downstream of the detector -- views built by the resize kernel, inverse coordinate maps, merged
    rotated NMS + cap -- against tests/golden/tta_merge.npz = the reference's own tta.py
    (_get_augmented_inputs / _get_augmented_corners / _merge_detections, :232-268) run under stubs on canned per-view
    detections: views pixel-identical, inverse-mapped corners and the merged detections bit-equal, same order."""
    from test_model_cpu import _tta_fixture_outputs
    from dafne_amd.modeling.tta import DotaDatasetMapperTTA, OneStageRCNNWithTTA
    g = golden("tta_merge")
    cfg, m, P = build("dota-1.5_r101.yaml" if name != "d10_pre" else "dota-1.0_r101.yaml", seed=9)
    C, post = [int(v) for v in g[name + "_cfg"]]
    assert cfg.MODEL.DAFNE.NUM_CLASSES == C
    cfg.TEST.AUG.MIN_SIZES = [int(v) for v in g[name + "_min_sizes"]]
    cfg.TEST.AUG.MAX_SIZE = int(g[name + "_max_size"])
    m.proposal_generator.dafne_outputs.post_nms_topk = m.proposal_generator.dafne_outputs.post_nms_topk_test = post
    oh, ow = [int(v) for v in g[name + "_orig_hw"]]
    tta = OneStageRCNNWithTTA(cfg, m)
    aug, tfms = tta._get_augmented_inputs({"image": torch.from_numpy(g[name + "_image"]), "height": oh, "width": ow})
    want = g[name + "_views"]
    assert len(aug) == want.shape[0]
    for k, v in enumerate(aug):
        assert tuple(v["image"].shape[1:]) == (int(want[k, 0]), int(want[k, 1]))
        if "%s_view%d_image" % (name, k) in g:
            assert np.array_equal(v["image"].cpu().numpy(), g["%s_view%d_image" % (name, k)])
    inst = tta._invert_and_concat(_tta_fixture_outputs(g, name, dev()), tfms)
    assert np.array_equal(inst.pred_corners.cpu().numpy(), g[name + "_inv_corners"])
    merged = tta._merge_detections(inst)
    for key in ("pred_corners", "scores", "centerness", "pred_classes"):
        assert np.array_equal(getattr(merged, key).cpu().numpy(), g["%s_merged_%s" % (name, key)]), key


def test_tta_packed_chunks_equal_the_reference_style_loop():
    """OneStageRCNNWithTTA._batch_inference_packed (chunks of views on the pipelined path, one host wait) returns the
    detections of _batch_inference (tta.py:170-179: one model.inference call per chunk)."""
    from dafne_amd.modeling.tta import OneStageRCNNWithTTA
    cfg, m, P = build("dota-1.5_r101.yaml", seed=21)
    cfg.TEST.AUG.MIN_SIZES = [96, 128, 160, 224]
    cfg.TEST.AUG.MAX_SIZE = 256
    tta = OneStageRCNNWithTTA(cfg, m)
    g = torch.Generator().manual_seed(9)
    img = torch.randint(0, 256, (3, 128, 160), generator=g, dtype=torch.uint8).to(dev())
    aug, _ = tta._get_augmented_inputs({"image": img, "height": 128, "width": 160})
    assert len(aug) == 12
    a = tta._batch_inference(aug)
    b = tta._batch_inference_packed(aug)
    assert len(a) == len(b) == 12
    for x, y in zip(a, b):
        ix, iy = x["instances"], y["instances"]
        assert len(ix) == len(iy) and ix.image_size == iy.image_size
        assert torch.equal(ix.pred_corners, iy.pred_corners) and torch.equal(ix.scores, iy.scores)
        assert torch.equal(ix.pred_classes, iy.pred_classes)


def test_tta_groups_of_images_match_the_per_image_calls():
    """OneStageRCNNWithTTA.__call__ on several images of one size runs the same-shape views of up to images_per_group images as
    one detector call (the reference augments image by image: a scheduling choice).  Per image the result EQUALS the per-image
    call's: the same kernels on the same views, and no kernel's arithmetic depends on the batch an image sits in (conv_wr's
    split-K slice count is sized per image, test_conv_wr_slices_follow_the_shape)."""
    from dafne_amd.modeling.tta import OneStageRCNNWithTTA
    cfg, m, P = build("dota-1.5_r101.yaml", seed=21)
    cfg.TEST.AUG.MIN_SIZES = [96, 128, 160]
    cfg.TEST.AUG.MAX_SIZE = 256
    g = torch.Generator().manual_seed(12)
    imgs = [torch.randint(0, 256, (3, 128, 160), generator=g, dtype=torch.uint8).to(dev()) for _ in range(4)]
    imgs.append(torch.randint(0, 256, (3, 96, 128), generator=g, dtype=torch.uint8).to(dev()))
    inputs = [{"image": im, "height": int(im.shape[1]) + 5, "width": int(im.shape[2]) + 3} for im in imgs]
    one = OneStageRCNNWithTTA(cfg, m, images_per_group=1)
    grp = OneStageRCNNWithTTA(cfg, m, images_per_group=3)
    calls = []
    orig = grp._views_packed
    grp._views_packed = lambda views, sizes=None, sync=True: calls.append(list(sizes) if sizes else None) or orig(views, sizes, sync)
    a = one(inputs)
    b = grp(inputs)
    # images 0-2 form one group (3 sizes x 3 variants x 3 images: chunks of 9), image 3 and the smaller image 4 go alone
    assert calls == [[9, 9, 9], [3, 3, 3], [3, 3, 3]]
    assert len(a) == len(b) == 5
    for k, (x, y) in enumerate(zip(a, b)):
        ix, iy = x["instances"], y["instances"]
        assert ix.image_size == iy.image_size == (inputs[k]["height"], inputs[k]["width"])
        assert len(ix) == len(iy) > 0, k
        assert torch.equal(ix.pred_corners, iy.pred_corners) and torch.equal(ix.scores, iy.scores), k
        assert torch.equal(ix.pred_classes, iy.pred_classes) and torch.equal(ix.centerness, iy.centerness), k


@pytest.mark.parametrize("h,w", [(160, 192), (512, 640)])
def test_an_image_gets_the_same_detections_in_any_batch(h, w):
    """Batch invariance of the whole path: image k's packed detections are the same bits at batch 1, in a batch of 3 and in a
    batch of 8 (serial layout and sub-batch streams).  Holds because every kernel's per-output arithmetic AND the choice of
    kernel are functions of the image's own shape -- tile mapping, K order, split-K slice count, and the occupancy heuristics
    count the tiles of a nominal batch of 8 (conv.hip kNominalBatch, conv_wr.hip): at 512 x 640 a batch of 1-3 used to put the
    towers' first layers on the generic tile (other GroupNorm partial sums than the resident-patch kernel's) and the stride-2
    shortcuts on other kernels than a batch of 8."""
    cfg, m, P = build("dota-1.0_r101.yaml", seed=23)
    g = torch.Generator().manual_seed(5)
    b8 = torch.randint(0, 256, (8, 3, h, w), generator=g, dtype=torch.uint8).to(dev())
    r8, c8 = m.detect_packed(b8)
    torch.cuda.synchronize()
    r8, c8 = r8.clone(), c8.clone()
    for lo, hi in ((0, 1), (2, 5), (5, 8), (7, 8)):
        r, c = m.detect_packed(b8[lo:hi].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(c, c8[lo:hi]), (lo, hi)
        for i in range(hi - lo):
            k = int(c[i])
            assert k > 0 and torch.equal(r[i, :k], r8[lo + i, :k]), (lo, hi, i)
    rp, cp = m.detect_packed(b8, pipelined=True, splits=2)
    torch.cuda.synchronize()
    assert torch.equal(cp, c8) and all(torch.equal(rp[i, :int(c8[i])], r8[i, :int(c8[i])]) for i in range(8))
    if h >= 512:
        names = lambda n: [c.kernel_name() for c in m.plan(n, h, w).calls]
        assert names(1) == names(3) == names(8)              # one launch list per image shape


def test_tta_groups_equal_the_per_image_calls_at_view_sizes_that_cross_the_tile_thresholds():
    """The released TTA sizes on a 1024^2 tile give views of 450 .. 1200 pixels; at 450-700 a chunk of 3 views and a chunk of 9
    used to differ in kernel choice.  Two sizes of that range on 640^2 images, groups of 3 against one image per call."""
    from dafne_amd.modeling.tta import OneStageRCNNWithTTA
    cfg, m, P = build("dota-1.5_r101.yaml", seed=21)
    cfg.TEST.AUG.MIN_SIZES = [450, 700]
    cfg.TEST.AUG.MAX_SIZE = 1200
    g = torch.Generator().manual_seed(14)
    inputs = [{"image": torch.randint(0, 256, (3, 640, 640), generator=g, dtype=torch.uint8).to(dev()), "height": 640, "width": 640}
              for _ in range(3)]
    a = OneStageRCNNWithTTA(cfg, m, images_per_group=1)(inputs)
    b = OneStageRCNNWithTTA(cfg, m, images_per_group=3)(inputs)
    for k, (x, y) in enumerate(zip(a, b)):
        ix, iy = x["instances"], y["instances"]
        assert len(ix) == len(iy) > 0, k
        assert torch.equal(ix.pred_corners, iy.pred_corners) and torch.equal(ix.scores, iy.scores), k
        assert torch.equal(ix.pred_classes, iy.pred_classes), k


def test_plan_caches_keep_the_most_recent_shapes():
    """cfg.ENGINE.MAX_PLANS bounds both plan caches (a launch plan holds a whole network's buffers at one batch shape); a shape
    that was dropped is rebuilt on its next use and gives the same detections."""
    cfg, m, P = build("dota-1.0_r50.yaml", seed=29)
    cfg.ENGINE.MAX_PLANS = 3
    g = torch.Generator().manual_seed(2)
    shapes = [(96, 128), (128, 128), (128, 160), (160, 160), (96, 160)]
    imgs = [torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8).to(dev()) for h, w in shapes]
    first = []
    for im in imgs:
        r, c = m.detect_packed(im)
        rp, cp = m.detect_packed(im, pipelined=True, splits=2)
        torch.cuda.synchronize()
        assert torch.equal(c, cp)
        first.append((r.clone(), c.clone()))
        assert len(m._plans) <= 3 and len(m._pipe) <= 3
    assert (2, 96, 128, 0) not in m._plans and (2, 96, 160, 0) in m._plans          # least recently used went first
    r, c = m.detect_packed(imgs[0])                                                # rebuilt
    rp, cp = m.detect_packed(imgs[0], pipelined=True, splits=2)
    torch.cuda.synchronize()
    for rr, cc in ((r, c), (rp, cp)):
        assert torch.equal(cc, first[0][1])
        assert all(torch.equal(rr[i, :int(cc[i])], first[0][0][i, :int(cc[i])]) for i in range(2))
    assert len(m._plans) <= 3 and len(m._pipe) <= 3


def test_pipelined_side_stream_equals_serial():
    cfg, m, P = build("dota-1.0_r50.yaml", seed=13)
    g = torch.Generator().manual_seed(6)
    batches = [torch.randint(0, 256, (3, 3, 128, 160), generator=g, dtype=torch.uint8).to(dev()) for _ in range(4)]
    serial = [m.detect_packed(b) for b in batches]
    torch.cuda.synchronize()
    serial = [(r.clone(), c.clone()) for r, c in serial]
    piped = [m.detect_packed(b, pipelined=True, splits=2) for b in batches]      # no sync in between
    torch.cuda.synchronize()
    for (r0, c0), (r1, c1) in zip(serial, piped):
        assert torch.equal(c0, c1)
        for i in range(3):
            k = int(c0[i])
            assert torch.equal(r0[i, :k], r1[i, :k])
    # scheduling hint: the launches of a plan that has the GPU to itself carry DAFNE_CONV_EXCLUSIVE (small launches then take a
    # whole CU's LDS for an operand ring), the sub-batch plans of the pipelined step do not; results are the same (above)
    from dafne_amd import engine
    alone = [c for c in m.plan(3, 128, 160).calls if isinstance(c, engine.ConvCall)]
    shared = [c for p in m._pipe[(3, 128, 160, 2)]["plans"][0] for c in p.calls if isinstance(c, engine.ConvCall)]
    assert alone and all(c.prm.flags & engine.F_EXCL for c in alone)
    assert shared and not any(c.prm.flags & engine.F_EXCL for c in shared)


@pytest.mark.parametrize("cfgname,h,w", [("hrsc_r50.yaml", 800, 1216), ("ucas_aod_r101.yaml", 256, 320),
                                          ("dota-1.0_r101.yaml", 192, 256)])
def test_released_configs_run_end_to_end(cfgname, h, w):
    """BASELINE.json configs: #1 (HRSC R50, one 800x1216 image) and the other released heads
    (C = 1 / 2 / 15) build and produce well-formed detections."""
    cfg, m, P = build(cfgname, seed=17)
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
    out = m([{"image": img, "height": h, "width": w}])[0]["instances"]
    C = cfg.MODEL.DAFNE.NUM_CLASSES
    assert 0 < len(out) <= cfg.MODEL.DAFNE.POST_NMS_TOPK_TEST + 8
    assert out.pred_corners.shape == (len(out), 8) and out.pred_boxes.tensor.shape == (len(out), 4)
    assert int(out.pred_classes.max()) < C and int(out.pred_classes.min()) >= 0
    s = out.scores.cpu().numpy()
    assert np.all(np.diff(s) <= 0) and s.min() > 0 and s.max() <= 1
    b = out.pred_boxes.tensor.cpu().numpy()
    assert b[:, 0].min() >= 0 and b[:, 2].max() <= w and b[:, 1].min() >= 0 and b[:, 3].max() <= h
    assert torch.isfinite(out.pred_corners).all()


def test_resize_kernel_is_pillow_exact():
    """dafne_resize_bilinear_u8_hip vs the PIL-generated fixture, vs the oracle at the TTA sizes of a tile
    (450..1200 from 1024, here from a 256 crop to keep the CPU side quick), with flips."""
    import os
    import numpy as np
    from dafne_amd.modeling.tta import resize_u8, shortest_edge_size
    from oracle import resize as orz
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_pil.npz"))
    d = torch.device("cuda", 0)
    i = 0
    while "in_%d" % i in G:
        img, want = G["in_%d" % i], G["out_%d" % i]
        got = resize_u8(torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).to(d), want.shape[0], want.shape[1])
        assert np.array_equal(got.cpu().numpy().transpose(1, 2, 0), want), i
        i += 1
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (3, 256, 320), dtype=np.uint8)
    t = torch.from_numpy(img).to(d)
    for s in (113, 200, 256, 300):
        nh, nw = shortest_edge_size(256, 320, s, 360)
        for hf, vf in ((False, False), (True, False), (False, True)):
            got = resize_u8(t, nh, nw, hf, vf).cpu().numpy()
            assert np.array_equal(got, orz.resize_bilinear_u8(img, nh, nw, hf, vf)), (s, hf, vf)
    # full-size TTA case: 1024 -> 450 and 1024 -> 1200, checksum against the oracle on a strip
    big = rng.integers(0, 256, (3, 1024, 1024), dtype=np.uint8)
    tb = torch.from_numpy(big).to(d)
    for s in (450, 1200):
        got = resize_u8(tb, s, s).cpu().numpy()
        ref = orz.resize_bilinear_u8(big[:, :, :], s, s)
        assert np.array_equal(got, ref)


def test_tta_view_sharded_driver_equals_the_wrapper():
    """OneStageRCNNWithTTA.inference_view_sharded (the multi-GPU form: views sharded over ranks, gather, merge on rank 0)
    run by ONE process gives exactly the wrapper's __call__ result -- same kernels on the same views; the 2-rank sharding
    and gather are covered on CPU by tests/test_gather_gloo.py::test_tta_views_sharded_over_two_ranks."""
    from dafne_amd.modeling.tta import OneStageRCNNWithTTA
    cfg, m, P = build("dota-1.5_r101.yaml", seed=9)
    cfg.TEST.AUG.MIN_SIZES = [96, 128, 160]
    cfg.TEST.AUG.MAX_SIZE = 192
    g = torch.Generator().manual_seed(4)
    img = torch.randint(0, 256, (3, 128, 160), generator=g, dtype=torch.uint8)
    tta = OneStageRCNNWithTTA(cfg, m)
    inp = {"image": img, "height": 128, "width": 160}
    a = tta([inp])[0]["instances"]
    b = tta.inference_view_sharded(inp)["instances"]
    assert len(a) == len(b) > 0
    assert a.image_size == b.image_size == (128, 160)          # the ORIGINAL image's size, not the first view's
    assert torch.equal(a.pred_corners, b.pred_corners) and torch.equal(a.scores, b.scores) and torch.equal(a.pred_classes, b.pred_classes)


def test_check_finite_debug_mode_makes_a_poisoned_weight_loud():
    """ENGINE.CHECK_FINITE (round 6, advisor): the branch-free epilogues turn a NaN accumulator of a non-ReLU layer into -inf (max with
    -inf), so a corrupted checkpoint shows as empty / wrong detections; with the switch on, the one-stream detect_packed raises when a
    head output is not finite -- and stays silent on a healthy model."""
    from dafne_amd import _lib
    cfg, m, P = build("dota-1.0_r50.yaml", seed=5)
    cfg.ENGINE.CHECK_FINITE = True
    g = torch.Generator().manual_seed(4)
    img = torch.randint(0, 256, (3, 128, 160), generator=g, dtype=torch.uint8)
    out = m([{"image": img, "height": 128, "width": 160}])
    assert len(out[0]["instances"]) > 0
    bad = {k: v.clone() for k, v in P.items()}
    key = "backbone.fpn_lateral4.weight"
    assert key in bad
    bad[key][3, 5, 0, 0] = float("nan")
    m.load_state_dict(bad)
    m.to(dev())
    m.invalidate()
    with pytest.raises(_lib.DafneHipError, match="CHECK_FINITE"):
        m([{"image": img, "height": 128, "width": 160}])


def test_plan_caches_share_a_byte_budget():
    """cfg.ENGINE.MAX_PLAN_BYTES (round 6, advisor): the one-stream plans and the sub-batch pipelines together hold at most that many
    bytes of device memory (as the allocator counts it around a build); the least recently used entry of EITHER cache goes first, the
    entry just built never; a dropped shape is rebuilt with the same detections."""
    cfg, m, P = build("dota-1.0_r50.yaml", seed=31)
    g = torch.Generator().manual_seed(9)
    shapes = [(96, 128), (128, 128), (128, 160), (160, 160)]
    imgs = [torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8).to(dev()) for h, w in shapes]
    r0, c0 = m.detect_packed(imgs[0])
    torch.cuda.synchronize()
    first = (r0.clone(), c0.clone())
    one = sum(v[1] for v in m._plan_lru.values())
    assert one > 0 and len(m._plan_lru) == 1
    cfg.ENGINE.MAX_PLAN_BYTES = int(2.5 * one)            # room for about two plans of this size
    for im in imgs[1:]:
        m.detect_packed(im)
        m.detect_packed(im, pipelined=True, splits=2)
        torch.cuda.synchronize()
        tot = sum(v[1] for v in m._plan_lru.values())
        assert tot <= cfg.ENGINE.MAX_PLAN_BYTES or len(m._plan_lru) == 1, (tot, len(m._plan_lru))
        assert len(m._plan_lru) == len(m._plans) + len(m._pipe)
    assert (2, 96, 128, 0) not in m._plans
    r, c = m.detect_packed(imgs[0])
    torch.cuda.synchronize()
    assert torch.equal(c, first[1]) and all(torch.equal(r[i, :int(c[i])], first[0][i, :int(c[i])]) for i in range(2))
