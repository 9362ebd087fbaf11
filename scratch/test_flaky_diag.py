"""Diagnostic copy of test_batch_invariance_and_determinism: on a mismatch between the two eager calls, compare the two plan sets
buffer by buffer (stem input, FPN features, head outputs) per sub-batch."""
import os, sys
import pytest, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_model import build, test_end_to_end_detections_vs_oracle_postprocess as e2e   # noqa
pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("cfgname,stride_norm", [("dota-1.0_r50.yaml", True), ("dota-1.5_r101.yaml", True), ("dota-1.0_r50.yaml", False)])
def test_a_pre(cfgname, stride_norm):
    e2e(cfgname, stride_norm)

def test_b_diag():
    cfg, m, P = build("dota-1.0_r50.yaml", seed=7)
    g = torch.Generator().manual_seed(2)
    ims = [torch.randint(0, 256, (3, 128, 128), generator=g, dtype=torch.uint8) for _ in range(3)]
    inputs = [{"image": im, "height": 128, "width": 128} for im in ims]
    a = m(inputs)
    b = m(inputs)
    torch.cuda.synchronize()
    ok = all(torch.equal(x["instances"].pred_corners, y["instances"].pred_corners) for x, y in zip(a, b))
    if ok:
        return
    st = [v for k, v in m._pipe.items()][0]
    msgs = ["bounds %s" % st["bounds"]]
    for k in range(len(st["plans"][0])):
        p0, p1 = st["plans"][0][k], st["plans"][1][k]
        msgs.append("sub-batch %d: stem_in equal %s" % (k, torch.equal(p0.stem_in, p1.stem_in)))
        for lv, (f0, f1) in enumerate(zip(p0.features, p1.features)):
            ne = int((f0.t != f1.t).sum())
            msgs.append("   feature %d: %d mismatching of %d, max |d| %.4g" % (lv, ne, f0.t.numel(), float((f0.t.float() - f1.t.float()).abs().max())))
    h0, h1 = st["ho"][0], st["ho"][1]
    for name in ("logits", "center", "delta_ctr"):
        for lv, (t0, t1) in enumerate(zip(getattr(h0, name), getattr(h1, name))):
            ne = (t0 != t1).reshape(t0.shape[0], -1).sum(1).tolist()
            if sum(ne):
                msgs.append("   head %s[%d]: mismatching per image %s of %d" % (name, lv, ne, t0[0].numel()))
    # third and fourth call: do the sets now agree with each other / with their first run?
    c = m(inputs); d2 = m(inputs); torch.cuda.synchronize()
    msgs.append("call 3 equals call 1: %s; call 4 equals call 2: %s; call 3 equals call 4: %s" % (
        all(torch.equal(x["instances"].pred_corners, y["instances"].pred_corners) for x, y in zip(a, c)),
        all(torch.equal(x["instances"].pred_corners, y["instances"].pred_corners) for x, y in zip(b, d2)),
        all(torch.equal(x["instances"].pred_corners, y["instances"].pred_corners) for x, y in zip(c, d2))))
    single = [m([inp])[0] for inp in inputs]
    msgs.append("call 1 equals single-image calls: %s; call 2: %s" % (
        [torch.equal(x["instances"].pred_corners, z["instances"].pred_corners) for x, z in zip(a, single)],
        [torch.equal(x["instances"].pred_corners, z["instances"].pred_corners) for x, z in zip(b, single)]))
    raise AssertionError("\n".join(msgs))
