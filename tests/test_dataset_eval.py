"""HRSC2016 / UCAS-AOD evaluation (dafne_amd/evaluation/{hrsc,ucas_aod}_evaluation.py, task1.py) against
tests/golden/eval_datasets.npz, which make_golden_datasets.py produced by running the reference's own parse_gt / xywha2xy4 /
load_annotation / parse_annotation / _generate_task_1_files / voc_eval on synthetic annotation files and predictions.
CPU: annotation parsers and Task1 writers (text in, text out: exact).  GPU: the scoring (polygon IoU on the device kernel)."""
import os
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "eval_datasets.npz"))


def _cfg(use_in_score=True):
    ns = types.SimpleNamespace
    return ns(MODEL=ns(DAFNE=ns(CENTERNESS="oriented", CENTERNESS_USE_IN_SCORE=use_in_score)), TEST=ns(IOU_TH=0.5),
              DATASETS=ns(DOTA_REMOVE_CONTAINER_CRANE=True), OUTPUT_DIR="./output")


def _preds(prefix, images, as_tensor):
    out = []
    for k, img in enumerate(images):
        p = {"file_name": "/data/x/%s.png" % img, "image_id": str(img), "height": 800, "width": 1216}
        for key in ("corners", "labels", "scores", "centerness"):
            a = G["%s_pred%d_%s" % (prefix, k, key)]
            p[key] = torch.from_numpy(a) if as_tensor else a
        out.append(p)
    return out


def _write_hrsc(root):
    os.makedirs(os.path.join(root, "labelXml"), exist_ok=True)
    for img in G["hrsc_images"]:
        with open(os.path.join(root, "labelXml", "%s.xml" % img), "w") as f:
            f.write(str(G["hrsc_xml_%s" % img]))


def _write_ucas(root):
    os.makedirs(os.path.join(root, "Annotations"), exist_ok=True)
    for img in G["ucas_images"]:
        with open(os.path.join(root, "Annotations", "%s.txt" % img), "w") as f:
            f.write(str(G["ucas_txt_%s" % img]))


def test_hrsc_xywha2xy4_and_parse_gt(tmp_path):
    from dafne_amd.evaluation import hrsc_evaluation as he
    got = np.array([he.xywha2xy4(r) for r in G["hrsc_xywha"]])
    assert np.array_equal(got, G["hrsc_xywha_out"])
    _write_hrsc(str(tmp_path))
    for img in G["hrsc_images"]:
        objs = he.parse_gt(os.path.join(str(tmp_path), "labelXml", "%s.xml" % img))
        assert [o["name"] for o in objs] == ["ship"] * len(objs)
        assert np.array_equal(np.array([o["bbox"] for o in objs]), G["hrsc_gt_bbox_%s" % img])
        assert np.array_equal(np.array([o["difficult"] for o in objs]), G["hrsc_gt_difficult_%s" % img])


@pytest.mark.parametrize("as_tensor", [True, False])
@pytest.mark.parametrize("tag,use_in_score", [("", True), ("_noctr", False)])
def test_hrsc_task1_files(tmp_path, tag, use_in_score, as_tensor):
    from dafne_amd.evaluation import hrsc_evaluation as he
    t1 = tmp_path / "Task1"
    t1.mkdir()
    he._generate_task_1_files(None, _preds("hrsc", G["hrsc_images"], as_tensor), str(tmp_path), str(t1), _cfg(use_in_score))
    assert open(os.path.join(str(t1), "Task1_ship.txt")).read().splitlines() == [str(l) for l in G["hrsc_task1_ship" + tag]]
    assert sorted(open(os.path.join(str(tmp_path), "imageset.txt")).read().split("\n")) == [str(v) for v in G["hrsc_imageset" + tag]]


def test_ucas_parse_gt_truncates_and_filters(tmp_path):
    from dafne_amd.evaluation import ucas_aod_evaluation as ue
    _write_ucas(str(tmp_path))
    for img in G["ucas_images"]:
        objs = ue.parse_gt(os.path.join(str(tmp_path), "Annotations", "%s.txt" % img))
        assert [o["name"] for o in objs] == [str(v) for v in G["ucas_gt_name_%s" % img]]
        assert np.array_equal(np.array([o["bbox"] for o in objs], dtype=np.float64).reshape(-1, 8), G["ucas_gt_bbox_%s" % img])
        assert all(o["difficult"] == 0 for o in objs)
        assert len(objs) < len(str(G["ucas_txt_%s" % img]).strip().split("\n"))          # the degenerate boxes are dropped


def test_ucas_task1_files(tmp_path):
    from dafne_amd.evaluation import ucas_aod_evaluation as ue
    t1 = tmp_path / "Task1"
    t1.mkdir()
    ue._generate_task_1_files(None, _preds("ucas", G["ucas_images"], True), str(tmp_path), str(t1), _cfg())
    for c in ("car", "airplane"):
        assert open(os.path.join(str(t1), "Task1_%s.txt" % c)).read().splitlines() == [str(l) for l in G["ucas_task1_" + c]]


def test_get_evaluator_picks_the_dataset_class():
    from dafne_amd.evaluation.inference import get_evaluator
    from dafne_amd.evaluation.dota_evaluation import DotaEvaluator
    from dafne_amd.evaluation.hrsc_evaluation import HrscEvaluator
    from dafne_amd.evaluation.ucas_aod_evaluation import UcasAodEvaluator
    cfg = _cfg()
    for name, cls in (("dota_1_0_val", DotaEvaluator), ("hrsc_test", HrscEvaluator), ("ucas_aod_test", UcasAodEvaluator)):
        ev = get_evaluator(cfg, name, distributed=False)
        assert type(ev) is cls and ev._output_dir == os.path.join("./output", "inference", name)
    with pytest.raises(NotImplementedError):
        get_evaluator(cfg, "icdar15_test")


def test_registered_datasets_yield_the_reference_records(tmp_path):
    """register_hrsc / register_ucas_aod / register_dota (tools/plain_train_net.py:568-570): names, directory layout, record
    fields as the reference's load_hrsc / load_ucas_aod return them (fixture) and as load_dota_json builds them; the metadata
    the evaluators read; get_evaluator finds it by name."""
    import json
    from dafne_amd.data import DatasetCatalog, MetadataCatalog, register_all
    from dafne_amd.evaluation.inference import get_evaluator
    ns = types.SimpleNamespace
    data = str(tmp_path)
    hroot, uroot = os.path.join(data, "hrsc"), os.path.join(data, "UCAS-AOD")
    _write_hrsc(hroot)
    _write_ucas(uroot)
    for root, images in ((hroot, G["hrsc_images"]), (uroot, G["ucas_images"])):
        os.makedirs(os.path.join(root, "ImageSets"))
        with open(os.path.join(root, "ImageSets", "test.txt"), "w") as f:
            f.write("\n".join(str(i) for i in images) + "\n")
    droot = os.path.join(data, "dota_1_5_split", "val1024")
    os.makedirs(droot)
    with open(os.path.join(droot, "DOTA1_5_val1024.json"), "w") as f:
        json.dump({"images": [{"id": 7, "file_name": "P0007__1__0___0.png", "height": 1024, "width": 1024},
                              {"id": 2, "file_name": "P0002__1__824___0.png", "height": 1024, "width": 1024}],
                   "annotations": [], "categories": []}, f)
    register_all(ns(DEBUG=ns(OVERFIT_NUM_IMAGES=-1)), data_dir=data)
    recs = DatasetCatalog.get("hrsc_test")
    assert [os.path.relpath(r["file_name"], hroot) for r in recs] == [str(v) for v in G["hrsc_rec_file"]]
    assert [r["image_id"] for r in recs] == G["hrsc_rec_id"].tolist()
    assert [[r["width"], r["height"]] for r in recs] == G["hrsc_rec_wh"].tolist()
    recs = DatasetCatalog.get("dota_1_5_val_1024")
    assert [r["image_id"] for r in recs] == [2, 7] and recs[0]["file_name"] == os.path.join(droot, "images", "P0002__1__824___0.png")
    m = MetadataCatalog.get("dota_1_5_val_1024")
    assert m.root_dir == droot and m.is_test is False and m.evaluator_type == "dota"
    assert MetadataCatalog.get("hrsc_test").is_test and MetadataCatalog.get("hrsc_test").root_dir == hroot
    ev = get_evaluator(_cfg(), "hrsc_test", distributed=False)
    assert ev._metadata is MetadataCatalog.get("hrsc_test")
    register_all(ns(DEBUG=ns(OVERFIT_NUM_IMAGES=2)), data_dir=data)                  # re-registration replaces; first N images
    recs = DatasetCatalog.get("ucas_aod_test")
    assert [os.path.relpath(r["file_name"], uroot) for r in recs] == [str(v) for v in G["ucas_rec_file"]]
    assert [r["image_id"] for r in recs] == [str(v) for v in G["ucas_rec_id"]]
    with pytest.raises(KeyError):
        os.environ.pop("DAFNE_DATA_DIR", None)
        register_all(None)


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("thr", [0.5, 0.75])
def test_do_hrsc_evaluation_matches_reference(tmp_path, thr):
    from dafne_amd.evaluation import hrsc_evaluation as he
    from dafne_amd.evaluation.voc_eval import voc_eval
    root = tmp_path / "hrsc"
    _write_hrsc(str(root))
    out = tmp_path / "out"
    out.mkdir()
    cfg = _cfg()
    cfg.TEST.IOU_TH = thr
    results = {}
    he.do_hrsc_evaluation("hrsc_test", types.SimpleNamespace(root_dir=str(root), is_test=True), _preds("hrsc", G["hrsc_images"], True),
                          str(out), None, results, cfg)
    tag = "%d" % int(thr * 100)
    assert results["task1"]["ship"] == float(G["hrsc_ap_" + tag]) and results["task1"]["map"] == results["task1"]["ship"]
    rec, prec, ap, _ = voc_eval(os.path.join(str(out), "Task1", "Task1_{:s}.txt"), os.path.join(str(root), "labelXml", "{:s}.xml"),
                                os.path.join(str(out), "imageset.txt"), "ship", ovthresh=thr, use_07_metric=True, parse_gt=he.parse_gt)
    assert np.array_equal(rec, G["hrsc_rec_" + tag]) and np.array_equal(prec, G["hrsc_prec_" + tag])
    assert os.path.exists(os.path.join(str(out), "results.txt")) and os.path.exists(os.path.join(str(out), "scores_overlap.csv"))


@pytest.mark.gpu
def test_ucas_evaluator_end_to_end_matches_reference(tmp_path):
    """get_evaluator -> process(inputs, outputs) with Instances -> evaluate(): the reference's results dict."""
    from dafne_amd.evaluation.inference import get_evaluator
    from dafne_amd.structures import Boxes, Instances
    root = tmp_path / "UCAS-AOD"
    _write_ucas(str(root))
    ev = get_evaluator(_cfg(), "ucas_aod_test", output_folder=str(tmp_path / "out"),
                       metadata=types.SimpleNamespace(root_dir=str(root), is_test=True), distributed=False)
    ev.reset()
    for p in _preds("ucas", G["ucas_images"], True):
        inst = Instances((p["height"], p["width"]))
        inst.pred_corners, inst.scores, inst.centerness, inst.pred_classes = p["corners"], p["scores"], p["centerness"], p["labels"]
        inst.pred_boxes = Boxes(torch.zeros(len(p["scores"]), 4))
        ev.process([{"image_id": p["image_id"], "file_name": p["file_name"], "height": p["height"], "width": p["width"]}], [{"instances": inst}])
    res = ev.evaluate()
    assert set(res) == {"task1"}
    for c in ("car", "airplane"):
        assert res["task1"][c] == float(G["ucas_ap_" + c])
    assert res["task1"]["map"] == pytest.approx((float(G["ucas_ap_car"]) + float(G["ucas_ap_airplane"])) / 2, abs=1e-15)
    assert os.path.exists(os.path.join(str(tmp_path / "out"), "instances_predictions.pth"))


@pytest.mark.gpu
def test_reference_shaped_chain_on_a_registered_dataset(tmp_path, monkeypatch):
    """do_test's chain (tools/plain_train_net.py:316-336,568-570) on a tiny HRSC-layout dataset: register_hrsc ->
    build_test_loader(cfg, "hrsc_test") -> inference_on_dataset(model, loader, get_evaluator(cfg, "hrsc_test")) ->
    {"task1": {"ship": ap, "map": ap}}; the Task1 file holds every detection of every image under the image's id."""
    from PIL import Image
    import dafne_amd.modeling  # noqa: F401
    from dafne_amd.config import load_cfg
    from dafne_amd.data import build_test_loader, register_hrsc
    from dafne_amd.evaluation.inference import get_evaluator, inference_on_dataset
    from dafne_amd.registry import build_model
    from oracle import model as om
    root = tmp_path / "hrsc"
    _write_hrsc(str(root))
    os.makedirs(str(root / "images"))
    os.makedirs(str(root / "ImageSets"))
    rng = np.random.default_rng(5)
    ids = [str(v) for v in G["hrsc_images"]]
    for img, (w, h) in zip(ids, G["hrsc_rec_wh"].tolist()):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(str(root / "images" / ("%d.bmp" % int(img))))
    with open(str(root / "ImageSets" / "test.txt"), "w") as f:
        f.write("\n".join(ids) + "\n")
    cfg = load_cfg(os.path.join(ROOT, "configs", "hrsc_r50.yaml"))
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 256, 512
    cfg.OUTPUT_DIR = str(tmp_path / "out")
    monkeypatch.setenv("DAFNE_DATA_DIR", str(tmp_path))
    register_hrsc(cfg)
    m = build_model(cfg)
    m.load_state_dict(om.make_params(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.DAFNE.NUM_CLASSES, seed=3))
    m.to(torch.device("cuda", 0))
    m.invalidate()
    loader = build_test_loader(cfg, "hrsc_test", batch_size=2, device=torch.device("cuda", 0), num_workers=2)
    ev = get_evaluator(cfg, "hrsc_test", distributed=False)
    res = inference_on_dataset(m, loader, ev)
    assert set(res) == {"task1"} and set(res["task1"]) == {"ship", "map"} and 0.0 <= res["task1"]["ship"] <= 1.0
    out = os.path.join(cfg.OUTPUT_DIR, "inference", "hrsc_test")
    lines = open(os.path.join(out, "Task1", "Task1_ship.txt")).read().splitlines()
    preds = torch.load(os.path.join(out, "instances_predictions.pth"), weights_only=False)
    assert len(preds) == 3 and len(lines) == sum(len(p["scores"]) for p in preds) > 0
    assert {l.split(" ")[0] for l in lines} <= {str(int(i)) for i in ids}
    assert [(p["height"], p["width"]) for p in preds] == [(h, w) for w, h in G["hrsc_rec_wh"].tolist()]      # the files' own sizes
