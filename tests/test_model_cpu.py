"""CPU: host-side logic of the engine's model surface (no kernels run)."""
import pytest
import torch

from oracle import model as om


def _cfg(name="dota-1.0_r50.yaml"):
    import os
    from dafne_amd.config import load_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return load_cfg(os.path.join(root, "configs", name))


@pytest.mark.parametrize("cfgname,depth,C", [("dota-1.0_r50.yaml", 50, 15), ("dota-1.5_r101.yaml", 101, 16)])
def test_state_dict_keys_match_reference_checkpoint_names(cfgname, depth, C):
    import dafne_amd.modeling  # noqa: F401  (registers the classes)
    from dafne_amd.registry import build_model, META_ARCH_REGISTRY, BACKBONE_REGISTRY, PROPOSAL_GENERATOR_REGISTRY
    assert "OneStageDetector" in META_ARCH_REGISTRY and "DAFNe" in PROPOSAL_GENERATOR_REGISTRY
    assert "build_dafne_resnet_fpn_backbone" in BACKBONE_REGISTRY
    m = build_model(_cfg(cfgname))
    sd = m.state_dict()
    shapes = om.param_shapes(depth, C)
    assert set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert m.backbone.size_divisibility == 32
    assert sorted(m.backbone.output_shape()) == ["p3", "p4", "p5", "p6", "p7"]
    # reference init of the class prior (dafne.py:283-285)
    assert torch.allclose(sd["proposal_generator.dafne_head.cls_logits.bias"], torch.full((C,), -4.59512), atol=1e-4)


def test_config_surface():
    from dafne_amd.config import get_cfg
    cfg = get_cfg()
    d = cfg.MODEL.DAFNE
    assert (d.NUM_CLASSES, d.PRE_NMS_TOPK_TEST, d.POST_NMS_TOPK_TEST, d.NMS_TH, d.INFERENCE_TH_TEST) == \
        (15, 2000, 1000, 0.1, 0.05)
    assert d.FPN_STRIDES == [8, 16, 32, 64, 128] and d.CORNER_PREDICTION == "center-to-corner"
    cfg.merge_from_list(["MODEL.DAFNE.NMS_TH", "0.2", "GLOBAL.HACK", "1.0"])     # unknown keys tolerated
    assert cfg.MODEL.DAFNE.NMS_TH == 0.2 and cfg.GLOBAL.HACK == 1.0
    c = _cfg("dota-1.0_r101.yaml")
    assert c.MODEL.RESNETS.DEPTH == 101 and c.MODEL.DAFNE.THRESH_WITH_CTR is True
    assert c.MODEL.PIXEL_MEAN == [123.675, 116.28, 103.53] and c.INPUT.FORMAT == "BGR"
    assert len(c.TEST.AUG.MIN_SIZES) == 9


def test_cpu_input_fails_loudly():
    import dafne_amd.modeling  # noqa: F401
    from dafne_amd.registry import build_model
    m = build_model(_cfg())
    with pytest.raises(RuntimeError):
        m.backbone(torch.zeros(1, 3, 64, 64))
    with pytest.raises(Exception):
        m([{"image": torch.zeros(3, 64, 64, dtype=torch.uint8), "height": 64, "width": 64}])


def test_tta_mapper_geometry_and_inverse_transforms(monkeypatch):
    """CPU: view list, sizes and inverse coordinate maps of the TTA mapper
    (tta.py:48-135, 244-262) -- no kernels involved."""
    import numpy as np
    from dafne_amd.modeling.tta import DotaDatasetMapperTTA, shortest_edge_size
    from oracle import postprocess as opp
    cfg = _cfg("dota-1.0_r101.yaml")
    assert shortest_edge_size(1024, 1024, 450, 1200) == (450, 450)
    assert shortest_edge_size(600, 1000, 800, 1200) == (720, 1200)      # capped by MAX_SIZE
    mapper = DotaDatasetMapperTTA(cfg)
    img = torch.randint(0, 256, (3, 96, 128), dtype=torch.uint8)
    with pytest.raises(Exception):              # the product resampler has no CPU path
        mapper({"image": img, "height": 96, "width": 128, "image_id": 7})
    # geometry under test here: stand the oracle's Pillow-exact resampler in for the device kernel
    import dafne_amd.modeling.tta as tta_mod
    from oracle import resize as orz
    monkeypatch.setattr(tta_mod, "resize_u8", lambda im, nh, nw, hf=False, vf=False: torch.from_numpy(
        orz.resize_bilinear_u8(im.numpy(), nh, nw, hf, vf)))
    views = mapper({"image": img, "height": 96, "width": 128, "image_id": 7})
    assert len(views) == 27                                               # 9 sizes x {none, hflip, vflip}
    v_plain, v_h, v_v = views[0], views[1], views[2]
    assert v_plain["image"].shape == v_h["image"].shape == v_v["image"].shape
    assert torch.equal(v_h["image"], torch.flip(v_plain["image"], dims=[2]))
    assert torch.equal(v_v["image"], torch.flip(v_plain["image"], dims=[1]))
    nh, nw = v_plain["image"].shape[1:]
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 400, (5, 8)).astype(np.float32)
    for v, hf, vf in ((v_plain, False, False), (v_h, True, False), (v_v, False, True)):
        got = v["transforms"].inverse().apply_coords(torch.from_numpy(c).reshape(-1, 2)).reshape(5, 8)
        assert got.dtype == torch.float32
        exp = opp.tta_invert_corners(c, (128 / nw, 96 / nh), hf, vf, (nh, nw))
        assert np.array_equal(got.numpy(), exp)


def _tta_fixture_outputs(g, name, device="cpu"):
    from dafne_amd.structures import Instances
    outs = []
    for k, (nh, nw, _, _) in enumerate(g[name + "_views"]):
        r = Instances((int(nh), int(nw)))
        for key in ("pred_corners", "scores", "centerness", "pred_classes"):
            setattr(r, key, torch.from_numpy(g["%s_view%d_%s" % (name, k, key)]).to(device))
        outs.append({"instances": r})
    return outs


@pytest.mark.parametrize("name", ["d15", "d10_pre", "d15_cap"])
def test_tta_views_and_inverse_maps_vs_reference_fixture(monkeypatch, golden, name):
    """DotaDatasetMapperTTA + OneStageRCNNWithTTA._invert_and_concat against tests/golden/tta_merge.npz (the
    reference's tta.py run under stubs): same view order / sizes / pixels, inverse-mapped corners bit-equal.  Host
    logic only: the oracle's Pillow-exact resampler stands in for the device kernel."""
    import numpy as np
    import dafne_amd.modeling.tta as tta_mod
    from oracle import resize as orz
    monkeypatch.setattr(tta_mod, "resize_u8", lambda im, nh, nw, hf=False, vf=False: torch.from_numpy(
        orz.resize_bilinear_u8(im.numpy(), nh, nw, hf, vf)))
    g = golden("tta_merge")
    cfg = _cfg("dota-1.5_r101.yaml")
    cfg.TEST.AUG.MIN_SIZES = [int(v) for v in g[name + "_min_sizes"]]
    cfg.TEST.AUG.MAX_SIZE = int(g[name + "_max_size"])
    oh, ow = [int(v) for v in g[name + "_orig_hw"]]
    views = tta_mod.DotaDatasetMapperTTA(cfg)({"image": torch.from_numpy(g[name + "_image"]), "height": oh, "width": ow})
    want = g[name + "_views"]
    assert len(views) == want.shape[0]
    for k, v in enumerate(views):
        assert tuple(v["image"].shape[1:]) == (int(want[k, 0]), int(want[k, 1]))
        if "%s_view%d_image" % (name, k) in g:
            assert np.array_equal(v["image"].numpy(), g["%s_view%d_image" % (name, k)])
    inst = tta_mod.OneStageRCNNWithTTA._invert_and_concat(None, _tta_fixture_outputs(g, name), [v["transforms"] for v in views])
    assert inst.pred_corners.dtype == torch.float32
    assert np.array_equal(inst.pred_corners.numpy(), g[name + "_inv_corners"])


def test_checkpoint_loading_pth_and_c2_pkl(tmp_path):
    """CPU: .pth ({"model": sd}) and Caffe2-named .pkl trunks load into the
    engine's parameter containers (no kernels)."""
    import pickle
    import numpy as np
    import dafne_amd.modeling  # noqa: F401
    from dafne_amd.checkpoint import load_weights, _c2_to_d2
    from dafne_amd.registry import build_model
    m = build_model(_cfg())
    P = om.make_params(50, 15, seed=21)
    pth = str(tmp_path / "model_final.pth")
    torch.save({"model": P, "iteration": 1}, pth)
    missing, unexpected = load_weights(m, pth, strict=True)
    assert not missing and not unexpected
    assert torch.equal(m.state_dict()["backbone.bottom_up.res3.0.conv2.weight"], P["backbone.bottom_up.res3.0.conv2.weight"])
    # Caffe2 names
    assert _c2_to_d2("conv1_w") == "stem.conv1.weight"
    assert _c2_to_d2("res_conv1_bn_s") == "stem.conv1.norm.weight"
    assert _c2_to_d2("res2_0_branch2a_w") == "res2.0.conv1.weight"
    assert _c2_to_d2("res4_22_branch2c_bn_b") == "res4.22.conv3.norm.bias"
    assert _c2_to_d2("res3_0_branch1_bn_s") == "res3.0.shortcut.norm.weight"
    assert _c2_to_d2("fc1000_w") is None
    c2 = {"conv1_w": np.ones((64, 3, 7, 7), np.float32) * 0.5, "res_conv1_bn_s": np.full(64, 2.0, np.float32),
          "res2_0_branch2a_w": np.zeros((64, 64, 1, 1), np.float32), "fc1000_w": np.zeros((1000, 2048), np.float32)}
    pkl = str(tmp_path / "R-50.pkl")
    with open(pkl, "wb") as f:
        pickle.dump({"model": c2, "matching_heuristics": True}, f)
    m2 = build_model(_cfg())
    missing, unexpected = load_weights(m2, pkl)
    assert not unexpected and "backbone.bottom_up.stem.conv1.weight" not in missing
    sd2 = m2.state_dict()
    assert float(sd2["backbone.bottom_up.stem.conv1.weight"].mean()) == 0.5
    assert float(sd2["backbone.bottom_up.stem.conv1.norm.weight"][0]) == 2.0


def _d2_to_c2(name):
    """Test-side inverse of the Caffe2 naming (MSRA R-50.pkl / R-101.pkl keys), written independently of
    checkpoint._c2_to_d2: `backbone.bottom_up.` stripped d2 name -> Caffe2 blob name."""
    parts = name.split(".")
    leaf = {"weight": "w"}
    if parts[0] == "stem":
        if parts[2] == "weight":
            return "conv1_w"
        return "res_conv1_bn_" + {"weight": "s", "bias": "b"}[parts[3]]
    stage, blk, conv = parts[0], parts[1], parts[2]
    br = {"shortcut": "branch1", "conv1": "branch2a", "conv2": "branch2b", "conv3": "branch2c"}[conv]
    if parts[3] == "weight":
        return "%s_%s_%s_w" % (stage, blk, br)
    return "%s_%s_%s_bn_%s" % (stage, blk, br, {"weight": "s", "bias": "b"}[parts[4]])


@pytest.mark.parametrize("depth", [50, 101])
def test_c2_trunk_covers_every_bottom_up_key(tmp_path, depth):
    """A complete MSRA-style trunk (every conv weight + affine BN scale / shift of R-50 / R-101 under its Caffe2 name,
    plus the fc1000 blobs) maps onto EVERY backbone.bottom_up.* parameter; the running statistics it does not ship keep
    d2's defaults (mean 0, var 1 - eps), for which FrozenBN folds to scale = weight exactly
    (tools/plain_train_net.py:576-578 -> DetectionCheckpointer -> c2_model_loading [recalled])."""
    import pickle
    import numpy as np
    import dafne_amd.modeling  # noqa: F401
    from dafne_amd import engine
    from dafne_amd.checkpoint import load_weights
    from dafne_amd.registry import build_model
    cfg = _cfg("dota-1.0_r%d.yaml" % depth)
    m = build_model(cfg)
    own = m.state_dict()
    bu = [k for k in own if k.startswith("backbone.bottom_up.")]
    stats = [k for k in bu if k.endswith(("running_mean", "running_var"))]
    rng = np.random.default_rng(depth)
    c2 = {}
    for k in bu:
        if k in stats:
            continue
        c2[_d2_to_c2(k[len("backbone.bottom_up."):])] = rng.normal(0, 1, tuple(own[k].shape)).astype(np.float32)
    n_convs = {50: 53, 101: 104}[depth]
    assert len(c2) == 3 * n_convs                                    # w, bn_s, bn_b per convolution
    c2["fc1000_w"] = np.zeros((1000, 2048), np.float32)
    c2["fc1000_b"] = np.zeros((1000,), np.float32)
    pkl = str(tmp_path / ("R-%d.pkl" % depth))
    with open(pkl, "wb") as f:
        pickle.dump({"model": c2, "matching_heuristics": True}, f)
    missing, unexpected = load_weights(m, pkl)
    assert not unexpected
    assert sorted(k for k in missing if k.startswith("backbone.bottom_up.")) == sorted(stats)
    sd = m.state_dict()
    for k in bu:
        if k not in stats:
            assert np.array_equal(sd[k].numpy(), c2[_d2_to_c2(k[len("backbone.bottom_up."):])]), k
    k0 = "backbone.bottom_up.res4.5.conv2"
    assert float(sd[k0 + ".norm.running_var"][0]) == np.float32(1.0) - np.float32(1e-5)
    w, b = engine.fold_frozen_bn(sd[k0 + ".weight"], sd[k0 + ".norm.weight"], sd[k0 + ".norm.bias"],
                                 sd[k0 + ".norm.running_mean"], sd[k0 + ".norm.running_var"])
    assert torch.allclose(w, sd[k0 + ".weight"] * sd[k0 + ".norm.weight"][:, None, None, None], rtol=1e-6, atol=0)
    assert torch.equal(b, sd[k0 + ".norm.bias"])


def test_subbatch_bounds_are_unequal_and_cover_the_batch():
    """one_stage_detector.subbatch_bounds (round 5): sub-batches of deliberately unequal size (weights 3:2:3 / 5:3), contiguous,
    covering the batch, every one >= 1 image; tiny batches fall back to the even split."""
    from dafne_amd.modeling.one_stage_detector import subbatch_bounds
    assert subbatch_bounds(8, 3) == [0, 3, 5, 8] and subbatch_bounds(8, 2) == [0, 5, 8]
    assert subbatch_bounds(16, 3) == [0, 6, 10, 16] and subbatch_bounds(12, 3) == [0, 5, 8, 12]
    for n in range(1, 40):
        for s in (1, 2, 3, 4):
            k = min(s, n)
            b = subbatch_bounds(n, k)
            assert b[0] == 0 and b[-1] == n and len(b) == k + 1 and all(b[i + 1] > b[i] for i in range(k)), (n, s, b)
