"""utils/host.py: the CPU count the engine's host-side pools are sized by (scheduler affinity capped by the cgroup quota)."""
import os

from dafne_amd.utils.host import usable_cpus


def _affinity():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def test_usable_cpus_honours_the_cgroup_v2_quota(tmp_path):
    n = _affinity()
    assert usable_cpus(str(tmp_path)) == n                      # no cgroup files: the affinity
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert usable_cpus(str(tmp_path)) == n                      # unlimited
    (tmp_path / "cpu.max").write_text("150000 100000\n")         # 1.5 CPUs -> 1
    assert usable_cpus(str(tmp_path)) == 1
    (tmp_path / "cpu.max").write_text("1600000 100000\n")        # the MI355X boxes of this project: 16 of 256
    assert usable_cpus(str(tmp_path)) == min(n, 16)
    (tmp_path / "cpu.max").write_text("garbage\n")
    assert usable_cpus(str(tmp_path)) == n


def test_usable_cpus_honours_the_cgroup_v1_quota(tmp_path):
    n = _affinity()
    (tmp_path / "cpu").mkdir()
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert usable_cpus(str(tmp_path)) == n
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("300000\n")
    assert usable_cpus(str(tmp_path)) == min(n, 3)


def test_usable_cpus_on_this_host_is_sane():
    assert 1 <= usable_cpus() <= (os.cpu_count() or 1)


def test_tta_batched_inverse_transforms_equal_the_per_view_loop():
    """tta.py:237-262 inverts every view's transform list on its corners; the wrapper does it for all views at once with per-row
    parameters (un-flip, x * ratio, x * ratio of the pre-resize): the same float32 operations in the same order -> equal bits."""
    import torch
    from dafne_amd.modeling.tta import HFlipT, OneStageRCNNWithTTA, ResizeT, TransformList, VFlipT
    from dafne_amd.structures import Instances
    g = torch.Generator().manual_seed(0)
    tfms, outs = [], []
    for k, (nh, nw) in enumerate([(450, 450), (700, 933), (1200, 1200), (96, 128)]):
        for flip in (None, "h", "v"):
            ops = [ResizeT(1000, 1024, 1024, 1024)] if k % 2 else []
            ops.append(ResizeT(1024, 1024, nh, nw))
            if flip == "h":
                ops.append(HFlipT(nw))
            if flip == "v":
                ops.append(VFlipT(nh))
            tfms.append(TransformList(ops))
            n = int(torch.randint(0, 50, (1,), generator=g))
            inst = Instances((1000, 1024))
            inst.pred_corners = torch.rand(n, 8, generator=g) * 1300 - 50
            inst.scores = torch.rand(n, generator=g)
            inst.centerness = torch.rand(n, generator=g)
            inst.pred_classes = torch.randint(0, 16, (n,), generator=g)
            outs.append({"instances": inst})
    a = OneStageRCNNWithTTA._invert_and_concat_loop(None, outs, tfms)
    b = OneStageRCNNWithTTA._invert_and_concat_fast(outs, tfms)
    assert b is not None and a.image_size == b.image_size and len(a) == len(b)
    for k in ("pred_corners", "scores", "centerness", "pred_classes"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    # a list the fast path does not know (two flips) falls back to the loop
    odd = [TransformList([ResizeT(8, 8, 4, 4), HFlipT(4), VFlipT(4)])]
    assert OneStageRCNNWithTTA._invert_and_concat_fast(outs[:1], odd) is None


def test_instances_host_twin_is_dropped_by_any_edit():
    """`Instances.to("cpu")` hands out the host twin made beside the packed rows only while the device-side fields are the ones it
    was made for: set / remove / attribute assignment and IN-PLACE writes (pred_boxes.scale / clip, `scores *= ..`: what a
    detectron2-style post-processing hook does before evaluator.process, dafne/evaluation/dafne_evaluator.py:44-58) all drop it."""
    import torch
    from dafne_amd.structures import _Boxes, _Instances

    def make():
        inst = _Instances((10, 10), scores=torch.tensor([0.5, 0.25]), pred_boxes=_Boxes(torch.tensor([[0., 0., 4., 4.], [1., 1., 3., 3.]])))
        snap = _Instances((10, 10), scores=torch.tensor([0.5, 0.25]), pred_boxes=_Boxes(torch.tensor([[0., 0., 4., 4.], [1., 1., 3., 3.]])),
                          marker=torch.tensor([1, 1]))
        inst.attach_cpu_twin(lambda: snap)
        return inst
    assert make().to("cpu").has("marker")                                   # untouched: the twin
    assert make().to(torch.device("cpu")).has("marker")
    a = make(); a.scores *= 2.0                                             # in-place write on a field tensor
    r = a.to("cpu"); assert not r.has("marker") and r.scores.tolist() == [1.0, 0.5]
    a = make(); a.pred_boxes.scale(2.0, 2.0)                                # in-place through a view inside Boxes
    r = a.to("cpu"); assert not r.has("marker") and r.pred_boxes.tensor[0].tolist() == [0.0, 0.0, 8.0, 8.0]
    a = make(); a.pred_boxes.clip((2, 2))
    assert not a.to("cpu").has("marker")
    a = make(); a.scores = torch.tensor([0.125, 0.75])                       # attribute assignment
    r = a.to("cpu"); assert not r.has("marker") and r.scores.tolist() == [0.125, 0.75]
    a = make(); a.set("extra", torch.zeros(2))
    assert a.to("cpu").has("extra")
    a = make(); a.remove("scores")
    assert not a.to("cpu").has("scores")
    a = make(); assert not a.to("cpu", non_blocking=True).has("marker")     # any other .to() form copies


def test_fragment_orders_of_the_resident_patch_kernel():
    """engine.pack_conv3x3_frag / pack_conv3x3_frag16 against the element maps include/dafne_amd.h states for dafne_conv3x3_c256_hip
    (32x32x16 form: fragment = k16 step, row lane & 31, K 16 step + 8 (lane >> 5); 16x16x32 form, flag DAFNE_CONV_FRAG16: fragment
    2m + cb, row 16 cb + (lane & 15), K 32 m + 8 (lane >> 4))."""
    import torch
    from dafne_amd import engine
    cout, K = 512, 2304
    w = (torch.arange(cout * K, dtype=torch.float32) % 30011).reshape(cout, K).to(torch.bfloat16)     # (distinct enough: every check is exact)
    f32 = engine.pack_conv3x3_frag(w).reshape(cout // 256, 8, 144, 64, 8)
    f16 = engine.pack_conv3x3_frag16(w).reshape(cout // 256, 8, 144, 64, 8)
    g = torch.Generator().manual_seed(5)
    for _ in range(200):
        nt, wv, j, lane, e = (int(torch.randint(0, n, (1,), generator=g)) for n in (cout // 256, 8, 144, 64, 8))
        assert f32[nt, wv, j, lane, e] == w[nt * 256 + wv * 32 + (lane & 31), 16 * j + 8 * (lane >> 5) + e]
        m, cb = j >> 1, j & 1
        assert f16[nt, wv, j, lane, e] == w[nt * 256 + wv * 32 + 16 * cb + (lane & 15), 32 * m + 8 * (lane >> 4) + e]
    assert engine.F_FRAG16 == 256
