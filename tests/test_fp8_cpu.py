"""CPU checks of the fp8 (config 5) definition: the e4m3 weight quantiser of the engine equals the oracle's restatement,
known answers of the OCP e4m3 rounding, config plumbing.  (The fp8 MFMA kernel itself: tests/test_gpu_fp8.py.)"""
import os

import pytest
import torch

from oracle import model as om

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engine_quantiser_equals_oracle():
    from dafne_amd import engine
    g = torch.Generator().manual_seed(11)
    for shape, spread in (((32, 64, 3, 3), 0.02), ((16, 128, 1, 1), 3.0), ((8, 3, 7, 7), 1e-4)):
        w = torch.randn(shape, generator=g) * spread * torch.logspace(-2, 2, shape[0])[:, None, None, None]
        w[0] = 0
        q, s, deq = engine.quantize_weight_e4m3(w)
        assert torch.equal(deq, om.quantize_weight_e4m3(w))
        assert torch.equal(q.float() * s[:, None, None, None], deq)
        assert torch.equal(deq.to(torch.bfloat16).float(), deq)            # exact on the bf16 kernels
        assert float(q.float().abs().max()) <= 448.0


def test_e4m3_known_answers():
    """OCP e4m3fn: 3 mantissa bits, max 448, min normal 2^-6, subnormal step 2^-9; round to nearest even."""
    x = torch.tensor([0.0, 1.0, 1.0625, 1.1875, 17.0, 19.0, 448.0, 500.0, -460.0, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10])
    exp = torch.tensor([0.0, 1.0, 1.0, 1.25, 16.0, 20.0, 448.0, 448.0, -448.0, 2.0 ** -9, 0.0, 2.0 ** -8])
    assert torch.equal(om.quantize_act_e4m3(x), exp)


def test_fp8_pack_layout():
    """pack_conv_fp8: byte [cout][slab][kh][kw][c] of the e4m3 weight, scale = power of two."""
    from dafne_amd import engine
    g = torch.Generator().manual_seed(2)
    w = torch.randn(256, 128, 3, 3, generator=g) * 0.05
    qb, s = engine.pack_conv_fp8(w, torch.device("cpu"))
    assert qb.dtype == torch.uint8 and tuple(qb.shape) == (256, 2 * 9 * 64)
    q, s2, _ = engine.quantize_weight_e4m3(w)
    assert torch.equal(s, s2)
    v = qb.view(torch.float8_e4m3fn).float().reshape(256, 2, 3, 3, 64)
    assert torch.equal(v[5, 1, 2, 0, 7], q.float()[5, 64 + 7, 2, 0])
    assert torch.equal(v.permute(0, 1, 4, 2, 3).reshape(256, 128, 3, 3), q.float())


def test_fp8_config_and_unknown_dtype():
    import dafne_amd.modeling  # noqa: F401
    from dafne_amd.config import load_cfg
    from dafne_amd.registry import build_model
    cfg = load_cfg(os.path.join(ROOT, "configs", "ucas_aod_r101_fp8.yaml"))
    assert cfg.ENGINE.WEIGHT_DTYPE == "fp8_e4m3" and cfg.MODEL.DAFNE.NUM_CLASSES == 2 and cfg.MODEL.RESNETS.DEPTH == 101
    m = build_model(cfg)
    assert m.backbone.weight_dtype == "fp8_e4m3" and m.proposal_generator.dafne_head.weight_dtype == "fp8_e4m3"
    bad = load_cfg(os.path.join(ROOT, "configs", "ucas_aod_r101.yaml"), ["ENGINE.WEIGHT_DTYPE", "int4"])
    with pytest.raises(NotImplementedError):
        build_model(bad)


def test_oracle_fp8_head_differs_from_bf16_and_is_deterministic():
    P = om.make_head_params(num_classes=2, seed=4, prefix="h.")
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(1, 256, s, s, generator=g) for s in (8, 4, 2, 1, 1)]
    with torch.no_grad():
        a = om.head_forward(P, feats, prefix="h.", emulate_bf16=True, fp8=True)
        b = om.head_forward(P, feats, prefix="h.", emulate_bf16=True, fp8=True)
        c = om.head_forward(P, feats, prefix="h.", emulate_bf16=True)
    assert all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    assert any(not torch.equal(x, y) for x, y in zip(a[1], c[1]))
    with pytest.raises(AssertionError):
        om.head_forward(P, feats, prefix="h.", fp8=True)          # fp8 is defined on the bf16-emulating path only
