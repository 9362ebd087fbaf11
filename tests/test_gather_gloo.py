"""CPU, world_size 2 over gloo: the N>1 path -- contiguous image shards, the
fixed-layout detection gather to rank 0, max-over-ranks timing reduction."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dafne_amd.evaluation.gather import gather_detections, shard_range, to_predictions
    n_total, k_cap = 6, 32
    lo, hi = shard_range(n_total, rank, world)
    b = hi - lo
    rows = torch.zeros(b, k_cap, 18)
    counts = torch.zeros(b, dtype=torch.int32)
    for j in range(b):
        g = lo + j                      # global image id
        counts[j] = g + 1
        rows[j, : g + 1, 8] = torch.arange(g + 1, 0, -1).float() / 10 + g   # scores tagged with image id
        rows[j, : g + 1, 10] = g
    out = gather_detections(rows, counts, dst=0)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        ra, ca = out
        preds = to_predictions(ra, ca)
        ok = ca.tolist() == [1, 2, 3, 4, 5, 6] and all(
            int(p["labels"][0]) == i and len(p["scores"]) == i + 1 for i, p in enumerate(preds))
        q.put(("rank0", ok, float(t)))
    else:
        q.put(("rank1", out is None, float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_timing_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = dict((r[0], r[1:]) for r in res)
    assert res["rank0"][0] and res["rank1"][0]
    assert res["rank0"][1] == 1.5 and res["rank1"][1] == 1.5      # MAX over ranks


def test_shard_range_covers_everything_once():
    from dafne_amd.evaluation.gather import shard_range
    for n, w in ((64, 8), (7, 2), (3, 8), (0, 4)):
        seen = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))


def _driver_worker(rank, world, port, q):
    """The evaluation driver's shard / batch / pad / gather / order logic with the detector stubbed out
    (tools/eval_net.py's multi-process mode minus the GPU)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dafne_amd.evaluation.driver import inference_on_images, instances_to_rows
    from dafne_amd.evaluation.gather import to_predictions
    from dafne_amd.structures import Boxes, Instances
    k_cap = 16
    results = {}
    for n_total, batch in ((7, 3), (1, 4), (4, 1)):
        calls = []

        def detect_batch(lo, hi):                      # stub detector: image g yields g % 5 + 1 detections tagged with g
            calls.append((lo, hi))
            insts = []
            for g in range(lo, hi):
                k = g % 5 + 1
                inst = Instances((64, 64))
                inst.pred_corners = torch.full((k, 8), float(g))
                inst.scores = torch.arange(k, 0, -1).float() / 10
                inst.centerness = torch.ones(k)
                inst.pred_classes = torch.full((k,), g, dtype=torch.int64)
                inst.pred_boxes = Boxes(torch.zeros(k, 4))
                insts.append(inst)
            return instances_to_rows(insts, k_cap)
        out = inference_on_images(detect_batch, n_total, k_cap, batch_size=batch, rank=rank, world=world)
        per = (n_total + world - 1) // world
        lo, hi = min(rank * per, n_total), min((rank + 1) * per, n_total)
        ok_calls = [c for c in calls] == [(b, min(b + batch, hi)) for b in range(lo, hi, batch)]
        if rank == 0:
            rows, counts = out
            preds = to_predictions(rows, counts, image_ids=list(range(n_total)))
            ok = counts.tolist() == [g % 5 + 1 for g in range(n_total)] and rows.shape == (n_total, k_cap, 18) and all(
                p["image_id"] == g and p["labels"].tolist() == [g] * (g % 5 + 1) and float(p["corners"][0, 0]) == g
                for g, p in enumerate(preds))
            results[(n_total, batch)] = ok and ok_calls
        else:
            results[(n_total, batch)] = out is None and ok_calls
    q.put((rank, all(results.values()), sorted(results)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_evaluation_driver_with_stub_detector():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_single_process_driver_and_instances_roundtrip():
    from dafne_amd import postprocess as pp
    from dafne_amd.evaluation.driver import inference_on_images, instances_to_rows
    from dafne_amd.structures import Boxes, Instances
    inst = Instances((10, 20))
    inst.pred_corners = torch.arange(24.0).reshape(3, 8)
    inst.scores = torch.tensor([0.9, 0.5, 0.1])
    inst.centerness = torch.tensor([0.3, 0.2, 0.1])
    inst.pred_classes = torch.tensor([2, 0, 7])
    inst.pred_boxes = Boxes(torch.arange(12.0).reshape(3, 4))
    rows, counts = instances_to_rows([inst, Instances((10, 20), pred_corners=torch.zeros(0, 8), scores=torch.zeros(0),
                                                       centerness=torch.zeros(0), pred_classes=torch.zeros(0, dtype=torch.int64))], 8)
    assert counts.tolist() == [3, 0]
    back = pp.rows_to_instances(rows, counts, [(10, 20), (10, 20)])
    assert torch.equal(back[0].pred_corners, inst.pred_corners) and torch.equal(back[0].pred_classes, inst.pred_classes)
    assert torch.equal(back[0].pred_boxes.tensor, inst.pred_boxes.tensor) and len(back[1]) == 0
    out = inference_on_images(lambda lo, hi: (rows[lo:hi], counts[lo:hi]), 2, 8, batch_size=1)
    assert torch.equal(out[0], rows) and out[1].tolist() == [3, 0]
    with pytest.raises(Exception):
        instances_to_rows([inst], 2)                     # more detections than the gather capacity


def _amax_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dafne_amd import engine
    # every rank saw other images: per-layer amax differs, rank 1 even has a layer at 0 (an all-dark shard)
    calib = {"res4.0.conv2": 3.5 + rank, "fpn_output3": 40.0 / (rank + 1), "cls_tower.0": 0.0 if rank else 7.25}
    red = engine.reduce_amax_over_ranks(calib)
    scales = {k: engine.act_qscale_from_amax(v) for k, v in red.items()}
    bad = None
    try:                                    # a rank that calibrated another layer set must not be paired silently
        engine.reduce_amax_over_ranks({("x" if rank else "y"): 1.0})
    except RuntimeError as e:
        bad = str(e)
    q.put((rank, red, scales, bad))
    dist.barrier()
    dist.destroy_process_group()


def test_fp8_calibration_is_identical_on_every_rank():
    """the fp8 model's activation scales must not depend on a rank's shard.  calibrate_fp8
    MAX-reduces the per-layer amax over the ranks (engine.reduce_amax_over_ranks); here two ranks with different amax
    dicts end with the same reduced values and the same power-of-two scales."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_amax_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, red0, sc0, bad0), (_, red1, sc1, bad1) = res
    assert red0 == red1 == {"res4.0.conv2": 4.5, "fpn_output3": 40.0, "cls_tower.0": 7.25}
    assert sc0 == sc1 and sc0["fpn_output3"] == 4.0 and sc0["res4.0.conv2"] == 32.0     # 2 * amax * q <= 448
    assert bad0 and bad1 and "different layer sets" in bad0


def test_reduce_amax_without_process_group_is_identity():
    from dafne_amd import engine
    assert engine.reduce_amax_over_ranks({"a": 1.5, "b": 0.0}) == {"a": 1.5, "b": 0.0}


class _StubTTA:
    """OneStageRCNNWithTTA's view-sharded driver with the detector stubbed (CPU, gloo): per-view packed detections are a
    pure function of the view index, the merge is the identity -- what is exercised is the sharding of the 27 views, the
    padded gather in view order, the per-view transforms of rank 0 and the reference's inverse maps."""

    def __new__(cls, cfg, calls):
        from dafne_amd.modeling.tta import DotaDatasetMapperTTA, OneStageRCNNWithTTA

        class Stub(OneStageRCNNWithTTA):
            def __init__(self, cfg, calls):
                torch.nn.Module.__init__(self)
                self.cfg, self.tta_mapper, self.batch_size, self.calls = cfg, DotaDatasetMapperTTA(cfg), 3, calls

            def _view_k_cap(self):
                return 8

            def _detect_view_range(self, input, lo, hi):
                self.calls.append((lo, hi))
                rows = torch.zeros(hi - lo, 8, 18)
                counts = torch.zeros(hi - lo, dtype=torch.int32)
                for v in range(lo, hi):
                    k = v % 5 + 1
                    counts[v - lo] = k
                    for j in range(k):
                        rows[v - lo, j, 0:8] = torch.arange(8, dtype=torch.float32) * 7.0 + v + 0.25 * j      # view coordinates
                        rows[v - lo, j, 8] = 1.0 - 0.01 * v - 0.001 * j
                        rows[v - lo, j, 10] = v % 16
                return rows, counts

            def _merge_detections(self, instances):
                return instances
        return Stub(cfg, calls)


def _tta_cfg():
    from dafne_amd.config import get_cfg
    cfg = get_cfg()
    cfg.TEST.AUG.MIN_SIZES = [450, 500, 600, 700, 800, 900, 1000, 1100, 1200]
    cfg.TEST.AUG.MAX_SIZE = 1200
    return cfg


def _tta_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    tta = _StubTTA(_tta_cfg(), calls)
    img = torch.zeros(3, 1024, 1024, dtype=torch.uint8)
    out = tta.inference_view_sharded({"image": img, "height": 1024, "width": 1024}, rank=rank, world=world, device="cpu")
    if rank == 0:
        inst = out["instances"]
        q.put((rank, calls, inst.pred_corners.clone(), inst.scores.clone(), inst.pred_classes.clone()))
    else:
        q.put((rank, calls, out))
    dist.barrier()
    dist.destroy_process_group()


def test_tta_views_sharded_over_two_ranks():
    """SURVEY 8(e), configs[3]: the 27 views of one image split over the ranks (14 + 13), ONE gather, inverse transforms and
    merge on rank 0 -- same result as one process running all 27 views (tta.py:173-197,237-268)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tta_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [(0, 14)] and res[1][1] == [(14, 27)] and res[1][2] is None
    # one process, all views
    calls = []
    single = _StubTTA(_tta_cfg(), calls).inference_view_sharded(
        {"image": torch.zeros(3, 1024, 1024, dtype=torch.uint8), "height": 1024, "width": 1024}, device="cpu")["instances"]
    assert calls == [(0, 27)]
    assert torch.equal(res[0][2], single.pred_corners) and torch.equal(res[0][3], single.scores) and torch.equal(res[0][4], single.pred_classes)
    assert len(single) == sum(v % 5 + 1 for v in range(27))
    # the inverse maps really ran: view 1 (450 px, hflip) maps x -> (450 - x) * 1024 / 450
    x_view = 0 * 7.0 + 1 + 0.25 * 0
    n0 = 0 % 5 + 1
    assert abs(float(single.pred_corners[n0, 0]) - (450 - x_view) * (1024 / 450)) < 1e-3
