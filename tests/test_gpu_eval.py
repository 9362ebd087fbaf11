"""GPU: tile ResultMerge and voc_eval on the device kernels vs the reference-generated fixture and
the CPU oracle (bit-exact keep lists / rec / prec / ap)."""
import os

import numpy as np
import pytest

import oracle
from oracle import evaluation as oev
from conftest import rrects

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_merge.npz"))


@pytest.mark.parametrize("cls", ["plane", "ship", "small-vehicle"])
def test_mergesingle_matches_reference(cls, tmp_path):
    from dafne_amd.evaluation import result_merge as rm
    src = tmp_path / ("Task1_%s.txt" % cls)
    src.write_text("\n".join(str(x) for x in G["merge_in_" + cls]) + "\n")
    dst = tmp_path / "merged"
    dst.mkdir()
    rm.mergesingle(str(dst), rm.py_cpu_nms_poly_fast, str(src))
    got = (dst / ("Task1_%s.txt" % cls)).read_text().splitlines()
    assert got == [str(x) for x in G["merge_out_" + cls]]


def test_mergebypoly_directory(tmp_path):
    from dafne_amd.evaluation import result_merge as rm
    src, dst = tmp_path / "Task1", tmp_path / "Task1_merged"
    src.mkdir()
    dst.mkdir()
    for cls in ("plane", "ship"):
        (src / ("Task1_%s.txt" % cls)).write_text("\n".join(str(x) for x in G["merge_in_" + cls]) + "\n")
    rm.mergebypoly(str(src), str(dst))
    for cls in ("plane", "ship"):
        assert (dst / ("Task1_%s.txt" % cls)).read_text().splitlines() == [str(x) for x in G["merge_out_" + cls]]


@pytest.mark.parametrize("m,seed", [(1, 0), (2, 1), (63, 2), (64, 3), (65, 4), (700, 5), (5000, 6)])
@pytest.mark.parametrize("strict", [True, False])
def test_f64_nms_vs_oracle(m, seed, strict):
    """fp64 rows that are NOT fp32-representable (offset/rate arithmetic), quantised scores (ties),
    a dense cluster, exact duplicates and zero-area rows."""
    from dafne_amd.evaluation import result_merge as rm
    rng = np.random.default_rng(seed)
    b = rrects(m, rng, extent=300.0 if m > 100 else 60.0).astype(np.float64)
    b = (b + np.array([824.0, 1648.0] * 4)) / 0.3 + rng.normal(0, 1e-9, b.shape)
    s = np.round(rng.uniform(0.05, 1, m), 2)
    if m > 4:
        b[3] = b[1]
        b[4] = np.tile(b[2, :2], 4)           # zero-area
    d = np.concatenate([b, s[:, None]], 1)
    got = rm._merge_nms_batched([d], 0.1, strict)[0]
    assert got == oracle.poly_nms_f64(d, 0.1, strict)


def test_f64_nms_batched_ragged():
    from dafne_amd.evaluation import result_merge as rm
    rng = np.random.default_rng(11)
    arrays = []
    for m in (0, 1, 130, 64, 900, 3, 257):
        b = rrects(m, rng, extent=200.0).astype(np.float64) / 0.5 if m else np.zeros((0, 8))
        arrays.append(np.concatenate([b, rng.uniform(0.05, 1, (m, 1))], 1))
    got = rm._merge_nms_batched(arrays, 0.1, True)
    assert got == [oracle.poly_nms_f64(a, 0.1, True) for a in arrays]


def _write_val(tmp_path):
    lab = tmp_path / "labelTxt"
    lab.mkdir()
    for img, txt in zip(G["val_images"], G["val_gt"]):
        (lab / (str(img) + ".txt")).write_text(str(txt))
    (tmp_path / "imageset.txt").write_text("\n".join(str(x) for x in G["val_images"]))
    for cls in ("plane", "ship"):
        (tmp_path / ("Task1_%s.txt" % cls)).write_text("\n".join(str(x) for x in G["val_det_" + cls]) + "\n")
    return str(tmp_path / "Task1_{:s}.txt"), str(lab / "{:s}.txt"), str(tmp_path / "imageset.txt")


@pytest.mark.parametrize("cls", ["plane", "ship"])
@pytest.mark.parametrize("thr", [0.5, 0.75])
def test_voc_eval_matches_reference(cls, thr, tmp_path):
    from dafne_amd.evaluation.voc_eval import voc_eval
    from dafne_amd.evaluation.dota_evaluation import parse_gt
    det, anno, imgset = _write_val(tmp_path)
    rec, prec, ap, so = voc_eval(det, anno, imgset, cls, ovthresh=thr, use_07_metric=True, parse_gt=parse_gt)
    tag = "%s_%d" % (cls, int(thr * 100))
    assert np.array_equal(rec, G["val_rec_" + tag]) and np.array_equal(prec, G["val_prec_" + tag])
    assert ap == float(G["val_ap_" + tag])
    assert all(len(r) == 4 and r[3] == cls for r in so)


def test_do_dota_evaluation_end_to_end(tmp_path):
    """predictions -> Task1 files -> voc_eval on the device -> results['task1'] (val mode) and the
    tile merge (test mode)."""
    import types
    from dafne_amd.evaluation import dota_evaluation as de
    det, anno, imgset = _write_val(tmp_path)
    cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(DAFNE=types.SimpleNamespace(CENTERNESS="none", CENTERNESS_USE_IN_SCORE=False)),
                                DATASETS=types.SimpleNamespace(DOTA_REMOVE_CONTAINER_CRANE=True), TEST=types.SimpleNamespace(IOU_TH=0.5))
    preds = []
    for img in G["val_images"]:
        rows = [l.split(" ") for c in ("plane", "ship") for l in map(str, G["val_det_" + c]) if l.startswith(str(img) + " ")]
        labs = [0 if any(str(l).startswith(" ".join(r[:3])) for l in G["val_det_plane"]) else 6 for r in rows]
        preds.append({"file_name": "/x/%s.png" % img, "height": 1024, "width": 1024,
                      "corners": np.array([[float(v) for v in r[2:]] for r in rows], dtype=np.float32),
                      "labels": np.array(labs), "scores": np.array([float(r[1]) for r in rows], dtype=np.float32),
                      "centerness": np.ones(len(rows), dtype=np.float32)})
    out = tmp_path / "out"
    out.mkdir()
    meta = types.SimpleNamespace(is_test=False, root_dir=str(tmp_path))
    results = {}
    de.do_dota_evaluation("dota_1_0_val", meta, preds, str(out), None, results, cfg)
    r = results["task1"]
    assert r["plane"] == pytest.approx(float(G["val_ap_plane_50"]), abs=1e-12)
    assert r["ship"] == pytest.approx(float(G["val_ap_ship_50"]), abs=1e-12)
    assert r["map"] == pytest.approx((r["plane"] + r["ship"]) / 15.0)
    meta.is_test = True
    de.do_dota_evaluation("dota_1_0_test", meta, preds, str(out), None, {}, cfg)
    assert os.path.exists(os.path.join(str(out), "Task1_merged", "Task1_plane.txt"))
