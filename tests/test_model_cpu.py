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
        got = v["transforms"].inverse().apply_coords(torch.from_numpy(c).reshape(-1, 2).double()).reshape(5, 8).float()
        exp = opp.tta_invert_corners(c, (128 / nw, 96 / nh), hf, vf, (nh, nw))
        assert np.array_equal(got.numpy(), exp)


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
