"""Diagnostic copy of test_single_image_batches_rotate_streams_and_plan_sets, after the tests that precede it in the file."""
import os, sys
import pytest, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_inference_loop as T
pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("where", ["host", "device"])
def test_a_pre(where):
    T.test_streamed_loop_equals_forward_per_image(where)

def test_a_pre2():
    T.test_streamed_outputs_carry_a_host_twin()

def test_b_diag():
    from dafne_amd.evaluation.inference import inference_on_dataset
    cfg, m = T._gpu_model()
    g = torch.Generator().manual_seed(21)
    items = [{"image": torch.randint(0, 256, (3, 512, 640), generator=g, dtype=torch.uint8).cuda(), "height": 512, "width": 640, "image_id": i}
             for i in range(13)]
    expected = [m([it])[0] for it in items]
    torch.cuda.synchronize()
    got = inference_on_dataset(m, [[it] for it in items])
    bad = [i for i, (a, e) in enumerate(zip(got, expected)) if not T._same(a, e)]
    if not bad:
        return
    again = [m([it])[0] for it in items]
    got2 = inference_on_dataset(m, [[it] for it in items])
    msg = ["streamed != model([input]) for images %s" % bad]
    msg.append("second round of model([input]) equals the first: %s" % [T._same(a, e) for a, e in zip(again, expected)])
    msg.append("second streamed round equals the second model round: %s" % [T._same(a, e) for a, e in zip(got2, again)])
    msg.append("first streamed round equals the second model round: %s" % [T._same(a, e) for a, e in zip(got, again)])
    raise AssertionError("\n".join(msg))
