"""CPU: torch restatement of DAFNeHead against the reference's own DAFNeHead
(fixture: tests/golden/head_forward.npz, weights regenerated from the seed)."""
import numpy as np
import pytest
import torch

from oracle import model as om


@pytest.mark.parametrize("name,C", [("d10", 15), ("ucas", 2)])
def test_head_forward_golden(golden, name, C):
    g = golden("head_forward")
    P = om.make_head_params(C, seed=7)
    feats = [torch.from_numpy(g["%s_feat%d" % (name, l)]) for l in range(5)]
    with torch.no_grad():
        logits, reg, center, ctr = om.head_forward(P, feats, prefix="")
    for l in range(5):
        for nm, t in (("logits", logits), ("reg", reg), ("center", center), ("ctr", ctr)):
            ref = g["%s_%s%d" % (name, nm, l)]
            assert np.allclose(t[l].numpy(), ref, atol=2e-5, rtol=1e-5), (nm, l)


def test_backbone_shapes_and_bf16_emulation_close():
    P = om.make_params(50, 15, seed=1)
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        f = om.backbone_forward(P, x, 50)
        fe = om.backbone_forward(P, x, 50, emulate_bf16=True)
    assert [tuple(f[k].shape) for k in ("p3", "p4", "p5", "p6", "p7")] == \
        [(1, 256, 8, 12), (1, 256, 4, 6), (1, 256, 2, 3), (1, 256, 1, 2), (1, 256, 1, 1)]
    for k in f:
        rel = (f[k] - fe[k]).norm() / f[k].norm()
        assert rel < 0.05, (k, float(rel))
