"""The test-time data path (dafne_amd/data/loader.py): the counterpart of `build_test_loader(cfg, dataset_name)`
(tools/plain_train_net.py:280-313) + DAFNeDatasetMapper's inference fields (dafne/data/datasets/dafne_dataset_mapper.py:13-19;
detectron2 DatasetMapper / read_image / ResizeShortestEdge [recalled]).  CPU: records, decode semantics, the resize-shape rule,
batching / sharding, the no-CPU-resize error.  GPU: the device resize is bit-exact to PIL's, and files -> loader ->
inference_on_dataset equals the detector called on the same decoded arrays."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Image = pytest.importorskip("PIL.Image")


def _cfg(name="dota-1.0_r50.yaml", **inp):
    from dafne_amd.config import load_cfg
    cfg = load_cfg(os.path.join(ROOT, "configs", name))
    if hasattr(cfg, "defrost"):
        cfg.defrost()
    for k, v in inp.items():
        setattr(cfg.INPUT, k, v)
    return cfg


def _smooth(rng, h, w):
    low = rng.integers(0, 256, (max(h // 16, 2), max(w // 16, 2), 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(low).resize((w, h), Image.BILINEAR)).astype(np.int32)
    return np.clip(img + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)


def _write(tmp, sizes, ext=".png", seed=0):
    rng = np.random.default_rng(seed)
    arrs = []
    for i, (h, w) in enumerate(sizes):
        a = _smooth(rng, h, w)
        Image.fromarray(a).save(os.path.join(tmp, "T%04d%s" % (i, ext)))
        arrs.append(a)
    return arrs


def test_list_image_records_is_sorted_recursive_and_filters_extensions(tmp_path):
    from dafne_amd.data import list_image_records
    (tmp_path / "b").mkdir()
    for rel in ("z.png", "a.JPG", "b/c.bmp", "notes.txt", "b/readme.md"):
        p = tmp_path / rel
        if rel.endswith((".txt", ".md")):
            p.write_text("x")
        else:
            Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(str(p))
    recs = list_image_records(str(tmp_path))
    assert [os.path.relpath(r["file_name"], str(tmp_path)) for r in recs] == ["a.JPG", "z.png", os.path.join("b", "c.bmp")]
    assert [r["image_id"] for r in recs] == ["a", "z", "c"]
    with pytest.raises(FileNotFoundError):
        list_image_records(str(tmp_path / "missing"))


def test_read_image_is_pil_rgb_reversed_and_applies_exif_orientation(tmp_path):
    from dafne_amd.data import read_image
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    Image.fromarray(a).save(p)
    b = read_image(p, "BGR")
    assert b.dtype == np.uint8 and b.flags["C_CONTIGUOUS"] and np.array_equal(b, a[:, :, ::-1])
    assert np.array_equal(read_image(p, "RGB"), a)
    # a grey file becomes three equal channels (convert("RGB")), as detectron2's convert_PIL_to_numpy does
    g = str(tmp_path / "g.png")
    Image.fromarray(a[:, :, 0]).save(g)
    bg = read_image(g, "BGR")
    assert bg.shape == (20, 30, 3) and np.array_equal(bg[:, :, 0], a[:, :, 0]) and np.array_equal(bg[:, :, 2], a[:, :, 0])
    # EXIF orientation 6 (rotate 270 to display): the decoded array is the transposed picture (_apply_exif_orientation)
    j = str(tmp_path / "o.jpg")
    ex = Image.Exif()
    ex[0x0112] = 6
    Image.fromarray(a).save(j, exif=ex, quality=95)
    assert read_image(j, "BGR").shape == (30, 20, 3)
    with pytest.raises(NotImplementedError):
        read_image(p, "YUV-BT.601")


def test_inference_resize_shape_follows_resize_shortest_edge_and_the_both_quirk():
    from dafne_amd.data import inference_resize_shape
    cfg = _cfg(MIN_SIZE_TEST=800, MAX_SIZE_TEST=1333)
    # ResizeShortestEdge.get_output_shape [recalled]: scale the short side to 800, cap the long side at 1333, round half up
    assert inference_resize_shape(cfg, 480, 640) == (800, 1067)
    assert inference_resize_shape(cfg, 640, 480) == (1067, 800)
    assert inference_resize_shape(cfg, 500, 2000) == (333, 1333)
    assert inference_resize_shape(cfg, 800, 1216) == (800, 1216)
    assert inference_resize_shape(_cfg(), 1024, 1024) == (1024, 1024)              # DOTA tiles: identity
    assert inference_resize_shape(_cfg(MIN_SIZE_TEST=0), 77, 99) == (77, 99)       # size 0 disables the resize
    assert inference_resize_shape(_cfg(MIN_SIZE_TEST=(608,), MAX_SIZE_TEST=2000), 304, 400) == (608, 800)
    with pytest.raises(NotImplementedError):
        inference_resize_shape(_cfg(MIN_SIZE_TEST=(608, 800)), 304, 400)
    # "both": T.Resize((h, w)) with the *_TRAIN keys, as plain_train_net.py:298-301 reads them
    both = _cfg(RESIZE_TYPE="both", RESIZE_HEIGHT_TRAIN=512, RESIZE_WIDTH_TRAIN=768)
    assert inference_resize_shape(both, 100, 100) == (512, 768)
    with pytest.raises(RuntimeError, match="Invalid resize-type"):
        inference_resize_shape(_cfg(RESIZE_TYPE="longest"), 10, 10)


def test_loader_batches_in_order_and_shards_like_the_inference_sampler(tmp_path):
    from dafne_amd.data import DatasetCatalog, build_test_loader
    sizes = [(48, 64)] * 7
    arrs = _write(str(tmp_path), sizes)
    cfg = _cfg(MIN_SIZE_TEST=48, MAX_SIZE_TEST=64)
    ld = build_test_loader(cfg, str(tmp_path), batch_size=3, num_workers=3)
    assert len(ld) == 3
    seen = [x for batch in ld for x in batch]
    assert [len(b) for b in ld] == [3, 3, 1]
    assert [x["image_id"] for x in seen] == ["T%04d" % i for i in range(7)]
    for x, a in zip(seen, arrs):
        assert x["image"].dtype == torch.uint8 and tuple(x["image"].shape) == (3, 48, 64) and not x["image"].is_cuda
        assert (x["height"], x["width"]) == (48, 64) and os.path.isfile(x["file_name"])
        assert np.array_equal(x["image"].numpy(), a[:, :, ::-1].transpose(2, 0, 1))            # BGR, CHW
    # contiguous shards: rank r gets [r * ceil(n / world), ...)  (gather.shard_range = detectron2's InferenceSampler)
    r0 = [x["image_id"] for b in build_test_loader(cfg, str(tmp_path), batch_size=2, shard=(0, 2)) for x in b]
    r1 = [x["image_id"] for b in build_test_loader(cfg, str(tmp_path), batch_size=2, shard=(1, 2)) for x in b]
    assert r0 == [x["image_id"] for x in seen[:4]] and r1 == [x["image_id"] for x in seen[4:]]
    assert len(build_test_loader(cfg, str(tmp_path), batch_size=2, shard=(3, 4))) == 1
    assert list(build_test_loader(cfg, str(tmp_path), batch_size=2, shard=(7, 8))) == []        # an empty shard yields nothing
    # by registered name, with record sizes that are checked against the files
    recs = [{"file_name": x["file_name"], "image_id": i, "height": 48, "width": 64, "annotations": [1, 2]} for i, x in enumerate(seen)]
    DatasetCatalog.remove("unit_tiles")
    DatasetCatalog.register("unit_tiles", recs)
    try:
        named = [x for b in build_test_loader(cfg, "unit_tiles", batch_size=4) for x in b]
        assert [x["image_id"] for x in named] == list(range(7)) and all("annotations" not in x for x in named)
        with pytest.raises(KeyError):
            DatasetCatalog.register("unit_tiles", recs)
        recs[2]["width"] = 65
        DatasetCatalog.remove("unit_tiles")
        DatasetCatalog.register("unit_tiles", recs)
        with pytest.raises(ValueError, match="mismatched width"):
            list(build_test_loader(cfg, "unit_tiles", batch_size=4))
    finally:
        DatasetCatalog.remove("unit_tiles")
    with pytest.raises(KeyError):
        build_test_loader(cfg, "no_such_dataset")


@pytest.mark.parametrize("backend,workers", [("process", 2), ("thread", 3), ("process", 0)])
def test_decode_backends_yield_the_same_batches(tmp_path, backend, workers):
    from dafne_amd.data import build_test_loader
    arrs = _write(str(tmp_path), [(32, 40)] * 5, seed=9)
    cfg = _cfg(MIN_SIZE_TEST=32, MAX_SIZE_TEST=40)
    ld = build_test_loader(cfg, str(tmp_path), batch_size=2, num_workers=workers, backend=backend)
    for _ in range(2):                                      # a loader can be iterated again
        got = [x for b in ld for x in b]
        assert [x["image_id"] for x in got] == ["T%04d" % i for i in range(5)]
        for x, a in zip(got, arrs):
            assert np.array_equal(x["image"].numpy(), a[:, :, ::-1].transpose(2, 0, 1))
    with pytest.raises(ValueError):
        build_test_loader(cfg, str(tmp_path), backend="fibers")


def test_a_resize_without_a_gpu_fails_loudly(tmp_path):
    """No CPU fallback: an image that needs the test-time resize needs the HIP resampler."""
    from dafne_amd import _lib
    from dafne_amd.data import build_test_loader
    _write(str(tmp_path), [(40, 60)])
    cfg = _cfg(MIN_SIZE_TEST=80, MAX_SIZE_TEST=200)
    with pytest.raises(_lib.DafneHipError, match="no CPU path"):
        list(build_test_loader(cfg, str(tmp_path), batch_size=1))


@pytest.mark.gpu
def test_device_resize_of_the_loader_is_pil_bilinear_bit_exact(tmp_path):
    """ResizeTransform.apply_image on uint8 = PIL.Image.resize(BILINEAR) [recalled]: the loader's one-launch resize + HWC->CHW
    gives exactly those bytes, for an upscale, a downscale and the max-size cap; `height` / `width` stay the file's."""
    from dafne_amd.data import build_test_loader, inference_resize_shape
    sizes = [(120, 200), (333, 190), (96, 640), (256, 256)]
    arrs = _write(str(tmp_path), sizes, seed=3)
    cfg = _cfg(MIN_SIZE_TEST=256, MAX_SIZE_TEST=512)
    dev = torch.device("cuda", 0)
    out = [x for b in build_test_loader(cfg, str(tmp_path), batch_size=3, device=dev, num_workers=2) for x in b]
    torch.cuda.synchronize()
    assert len(out) == 4
    for x, a, (h, w) in zip(out, arrs, sizes):
        nh, nw = inference_resize_shape(cfg, h, w)
        want = np.asarray(Image.fromarray(a[:, :, ::-1]).resize((nw, nh), Image.BILINEAR))
        assert x["image"].is_cuda and tuple(x["image"].shape) == (3, nh, nw) and (x["height"], x["width"]) == (h, w)
        assert np.array_equal(x["image"].cpu().numpy(), want.transpose(2, 0, 1)), (h, w, nh, nw)


@pytest.mark.gpu
def test_files_through_the_loader_equal_the_detector_on_the_decoded_arrays(tmp_path):
    """do_test's shape (plain_train_net.py:316-336): inference_on_dataset(model, build_test_loader(cfg, dir), evaluator).  The
    predictions equal model(batch) on the same decoded + PIL-resized arrays, image by image; the evaluator receives the file's
    own id / size."""
    from dafne_amd.data import build_test_loader, inference_resize_shape
    from dafne_amd.evaluation.inference import DafneEvaluator, inference_on_dataset
    from test_inference_loop import _gpu_model, _same
    cfg, m = _gpu_model(splits=1)
    if hasattr(cfg, "defrost"):
        cfg.defrost()
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 128, 224
    sizes = [(100, 150), (128, 160), (200, 120), (128, 160), (64, 224)]
    arrs = _write(str(tmp_path), sizes, seed=5)
    dev = torch.device("cuda", 0)
    expected = []
    for i in range(0, len(sizes), 2):
        batch = []
        for a, (h, w) in zip(arrs[i:i + 2], sizes[i:i + 2]):
            nh, nw = inference_resize_shape(cfg, h, w)
            r = np.asarray(Image.fromarray(a[:, :, ::-1]).resize((nw, nh), Image.BILINEAR))
            batch.append({"image": torch.from_numpy(np.ascontiguousarray(r.transpose(2, 0, 1))).to(dev), "height": h, "width": w})
        expected += m(batch)
    torch.cuda.synchronize()
    got = inference_on_dataset(m, build_test_loader(cfg, str(tmp_path), batch_size=2, device=dev), None)
    assert len(got) == len(sizes) and all(len(o["instances"]) > 0 for o in got)
    for i, (a, e) in enumerate(zip(got, expected)):
        assert _same(a, e), i
    k_cap = m.proposal_generator.dafne_outputs.packed_k_cap()
    ev = DafneEvaluator("tiles", cfg, distributed=False, k_cap=k_cap, device=dev)
    res = inference_on_dataset(m, build_test_loader(cfg, str(tmp_path), batch_size=2, device=dev), ev)
    preds = res["predictions"]
    assert [p["image_id"] for p in preds] == ["T%04d" % i for i in range(5)]
    assert [(p["height"], p["width"]) for p in preds] == sizes
    for p, e in zip(preds, expected):
        assert len(p["scores"]) == len(e["instances"])


@pytest.mark.gpu
def test_tta_wrapper_over_the_loader_equals_tta_on_the_decoded_arrays(tmp_path):
    """do_test_with_TTA's shape (plain_train_net.py:338-356): the same test loader, the model wrapped in OneStageRCNNWithTTA.
    The loader hands the wrapper the single-scale-resized image with the file's own height / width; the wrapper's views start from
    that image (tta.py:71-99, `pre` transform)."""
    from dafne_amd.data import build_test_loader, inference_resize_shape
    from dafne_amd.evaluation.inference import inference_on_dataset
    from dafne_amd.modeling.tta import OneStageRCNNWithTTA
    from test_inference_loop import _gpu_model, _same
    cfg, m = _gpu_model("dota-1.5_r101.yaml", splits=1)
    if hasattr(cfg, "defrost"):
        cfg.defrost()
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 128, 224
    cfg.TEST.AUG.MIN_SIZES, cfg.TEST.AUG.MAX_SIZE = [96, 160], 256
    sizes = [(100, 150), (128, 160), (128, 160)]
    arrs = _write(str(tmp_path), sizes, seed=8)
    dev = torch.device("cuda", 0)
    tta = OneStageRCNNWithTTA(cfg, m)
    inputs = []
    for a, (h, w) in zip(arrs, sizes):
        nh, nw = inference_resize_shape(cfg, h, w)
        r = np.asarray(Image.fromarray(a[:, :, ::-1]).resize((nw, nh), Image.BILINEAR))
        inputs.append({"image": torch.from_numpy(np.ascontiguousarray(r.transpose(2, 0, 1))).to(dev), "height": h, "width": w})
    expected = tta(inputs[:2]) + tta(inputs[2:])
    torch.cuda.synchronize()
    got = inference_on_dataset(tta, build_test_loader(cfg, str(tmp_path), batch_size=2, device=dev), None)
    assert len(got) == 3 and all(len(o["instances"]) > 0 for o in got)
    for i, (a, e) in enumerate(zip(got, expected)):
        assert a["instances"].image_size == sizes[i]
        assert torch.equal(a["instances"].pred_corners, e["instances"].pred_corners), i
        assert torch.equal(a["instances"].scores, e["instances"].scores), i
