"""CPU: numpy restatement of decode / corner sort / select against fixtures made
by running the reference's dafne_outputs.py + sort_corners.py under stubs."""
import numpy as np
import pytest

from oracle import postprocess as pp

VARIANTS = ["d10", "d15", "hrsc", "ucas", "d10_topk", "d15_topk"]
NO_STRIDE_NORM = ["d10_nsn", "d15_nsn"]


def test_sort_quadrilateral_golden(golden):
    g = golden("sort_corners")
    assert np.array_equal(pp.sort_quadrilateral(g["boxes"]), g["sorted"])
    # SURVEY appendix B KATs observed on the reference
    assert pp.sort_quadrilateral(np.array([[0, 0, 1, 0, 2, 0, 3, 0]], np.float32)).tolist() == [[0] * 8]
    assert pp.sort_quadrilateral(np.array([[1, 1, 0, 0, 0, 1, 1, 0]], np.float32)).tolist() == \
        [[0, 0, 0, 1, 1, 1, 1, 0]]


def test_compute_locations():
    loc = pp.compute_locations(2, 3, 8)
    assert loc.tolist() == [[4, 4], [12, 4], [20, 4], [4, 12], [12, 12], [20, 12]]


def _key(d):
    return d["fpn_levels"] * (1 << 32) + d["locations"][:, 1].astype(np.int64) * (1 << 20) \
        + d["locations"][:, 0].astype(np.int64) * 32 + d["pred_classes"]


@pytest.mark.parametrize("name", VARIANTS + NO_STRIDE_NORM)
def test_predict_proposals_golden(golden, name):
    """The *_nsn variants: MODEL.DAFNE.ENABLE_FPN_STRIDE_NORM false (dafne_outputs.py:771-774), fixture of
    make_golden_stride_norm.py."""
    nsn = name in NO_STRIDE_NORM
    g = golden("predict_no_stride_norm" if nsn else "predict_proposals")
    C, topk, post, twc, sortc = [int(v) for v in g[name + "_cfg"]]
    thr, nms_thr = [float(v) for v in g[name + "_thr"]]
    strides = [8, 16, 32, 64, 128]
    for im in range(2):
        levels = [(g["%s_logits%d" % (name, l)][im], g["%s_reg%d" % (name, l)][im],
                   g["%s_ctr%d" % (name, l)][im]) for l in range(5)]
        det = pp.predict_proposals(levels, strides, thresh=thr, topk=topk, nms_thresh=nms_thr,
                                   post_topk=post, thresh_with_ctr=bool(twc), sort_corners=bool(sortc), stride_norm=not nsn)
        ref = {k: g["%s_im%d_%s" % (name, im, k)] for k in
               ("pred_boxes", "pred_corners", "scores", "centerness", "pred_classes", "locations", "fpn_levels")}
        # same detections, same (descending-score) order; keyed compare guards ties
        assert det["scores"].shape == ref["scores"].shape
        assert np.array_equal(_key(det), _key(ref))
        assert np.allclose(det["scores"], ref["scores"], atol=1e-6)
        assert np.allclose(det["pred_corners"], ref["pred_corners"], atol=1e-3)
        assert np.allclose(det["pred_boxes"], ref["pred_boxes"], atol=1e-3)
        assert np.allclose(det["centerness"], ref["centerness"], atol=1e-6)


def test_detector_postprocess_drops_empty_and_scales():
    det = {"pred_boxes": np.array([[10, 10, 20, 20], [-30, 5, -10, 9], [90, 90, 130, 130]], np.float32),
           "pred_corners": np.tile(np.arange(8, dtype=np.float32), (3, 1)),
           "scores": np.array([.9, .8, .7], np.float32), "centerness": np.ones(3, np.float32),
           "pred_classes": np.zeros(3, np.int64), "locations": np.ones((3, 2), np.float32),
           "fpn_levels": np.zeros(3, np.int64)}
    out = pp.detector_postprocess(det, (100, 100), (200, 50), (100, 100))
    assert out["scores"].tolist() == pytest.approx([.9, .7])
    assert out["pred_boxes"].tolist() == [[5, 20, 10, 40], [45, 180, 50, 200]]
    assert out["pred_corners"][0].tolist() == [0, 2, 1, 6, 2, 10, 3, 14]


TTA_CASES = ["d15", "d10_pre", "d15_cap"]


def _tta_fixture_views(g, name):
    """Per view: (detections in the view's frame, (h, w, hflip, vflip))."""
    views = g[name + "_views"]
    dets = []
    for k in range(views.shape[0]):
        dets.append({key: g["%s_view%d_%s" % (name, k, key)] for key in ("pred_corners", "scores", "centerness", "pred_classes")})
    return views, dets


@pytest.mark.parametrize("name", TTA_CASES)
def test_tta_inverse_and_merge_golden(golden, name):
    """oracle tta_invert_corners + select_over_all_levels vs the reference's tta.py (_get_augmented_corners /
    _merge_detections run under stubs, tests/golden/make_golden_tta.py): the inverse-mapped corners are bit-equal
    (float32 arithmetic, un-flip -> un-resize -> un-pre-resize) and the merged detections are the same rows in the
    same order."""
    g = golden("tta_merge")
    views, dets = _tta_fixture_views(g, name)
    h, w = g[name + "_image"].shape[1:]
    oh, ow = [int(v) for v in g[name + "_orig_hw"]]
    pre = None if (oh, ow) == (h, w) else (ow * 1.0 / w, oh * 1.0 / h)
    inv = []
    for (nh, nw, hf, vf), d in zip(views, dets):
        c = pp.tta_invert_corners(d["pred_corners"], (w * 1.0 / nw, h * 1.0 / nh), bool(hf), bool(vf), (int(nh), int(nw)),
                                  pre_scale_xy=pre)
        inv.append({**d, "pred_corners": c})
    allv = pp.cat(inv)
    assert allv["pred_corners"].dtype == np.float32
    assert np.array_equal(allv["pred_corners"], g[name + "_inv_corners"])
    C, post = [int(v) for v in g[name + "_cfg"]]
    out = pp.select_over_all_levels(allv, float(g[name + "_nms_th"]), post)
    for key in ("pred_corners", "scores", "centerness", "pred_classes"):
        assert np.array_equal(out[key], g["%s_merged_%s" % (name, key)]), key
    # the fast (hull pre-filtered) oracle NMS gives the same merge
    out2 = pp.select_over_all_levels(allv, float(g[name + "_nms_th"]), post, fast=True)
    assert np.array_equal(out2["pred_corners"], out["pred_corners"])


def test_tta_view_list_golden(golden):
    """DotaDatasetMapperTTA's view order and sizes (reference tta.py:71-135 over a d2 ResizeShortestEdge stand-in):
    per MIN_SIZES entry [resize, resize+hflip, resize+vflip]; sizes by the shortest-edge rule with the MAX_SIZE cap."""
    from dafne_amd.modeling.tta import shortest_edge_size
    g = golden("tta_merge")
    for name in TTA_CASES:
        h, w = g[name + "_image"].shape[1:]
        want = []
        for s in g[name + "_min_sizes"]:
            nh, nw = shortest_edge_size(h, w, int(s), int(g[name + "_max_size"]))
            want += [(nh, nw, 0, 0), (nh, nw, 1, 0), (nh, nw, 0, 1)]
        assert [tuple(int(x) for x in v) for v in g[name + "_views"]] == want
