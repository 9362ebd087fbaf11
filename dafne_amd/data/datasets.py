"""The released datasets by name, test side: `register_dota / register_hrsc / register_ucas_aod` (tools/plain_train_net.py:
568-570) fill `DatasetCatalog` with the records the test loader reads and `MetadataCatalog` with what the evaluators read
(`root_dir`, `is_test`), under the reference's names and directory layout below $DAFNE_DATA_DIR:

  dota_{1,1_5}_{train,val,test}_{600,800,1024,1300,1600,2048}    dota_<v>_split/<split><size>/{images/, DOTA<v>_<split><size>.json, labelTxt/}
                                                                  (dafne/data/datasets/dota.py:362-384; COCO-style json: the "images" list)
  hrsc_{train,val,test,trainval}                                  hrsc/{images/<id>.bmp, labelXml/<id>.xml, ImageSets/<split>.txt}   (hrsc2016.py:54-82,186-199)
  ucas_aod_{train,val,test,trainval}                              UCAS-AOD/{AllImages/<id>.png, Annotations/<id>.txt, ImageSets/<split>.txt}
                                                                  (ucas_aod.py:75-89,131-147,183-196)

A record holds the image side only -- file_name, image_id, height / width where the reference takes them from the annotation
files -- the ground truth is read by the evaluators themselves (evaluation/*_evaluation.py); the training-side fields
(annotations, the random "_mini" subsets) are outside the inference path.  DEBUG.OVERFIT_NUM_IMAGES > 0 keeps the first N images
as in the reference.
"""
import json
import os
import types
import xml.etree.ElementTree as ET

from .loader import DatasetCatalog

__all__ = ["MetadataCatalog", "register_dota", "register_hrsc", "register_ucas_aod", "register_all",
           "load_dota_images", "load_hrsc_images", "load_ucas_aod_images"]


class _Metadata:
    """name -> namespace (detectron2's MetadataCatalog [recalled]: get(name) creates an empty entry, .set(**kw) fills it)."""

    def __init__(self):
        self._m = {}

    def get(self, name):
        if name not in self._m:
            ns = types.SimpleNamespace(name=name)
            ns.set = lambda _ns=ns, **kw: (_ns.__dict__.update(kw), _ns)[1]
            self._m[name] = ns
        return self._m[name]

    def __contains__(self, name):
        return name in self._m

    def remove(self, name):
        self._m.pop(name, None)


MetadataCatalog = _Metadata()


def _first_n(items, cfg):
    n = int(getattr(getattr(cfg, "DEBUG", None), "OVERFIT_NUM_IMAGES", -1)) if cfg is not None else -1
    return items[:n] if n > 0 else items


def load_dota_images(json_file, image_root, cfg=None):
    """The image half of load_dota_json (dota.py:124-131,213-218): images sorted by id."""
    with open(json_file) as f:
        imgs = sorted(json.load(f)["images"], key=lambda d: d["id"])
    return [{"file_name": os.path.join(image_root, d["file_name"]), "height": d["height"], "width": d["width"], "image_id": d["id"]}
            for d in _first_n(imgs, cfg)]


def load_hrsc_images(root, image_set, cfg=None):
    """The image half of load_hrsc (hrsc2016.py:54-82)."""
    out = []
    for s in ([image_set] if isinstance(image_set, str) else image_set):
        with open(os.path.join(root, "ImageSets", "%s.txt" % s)) as f:
            lines = f.read().splitlines()
        for line in _first_n(lines, cfg):
            img_id = int(line)
            a = ET.parse(os.path.join(root, "labelXml", "%d.xml" % img_id)).getroot()
            out.append({"file_name": os.path.join(root, "images", "%d.bmp" % img_id), "image_id": img_id,
                        "width": int(a.find("Img_SizeWidth").text), "height": int(a.find("Img_SizeHeight").text)})
    return out


def load_ucas_aod_images(root, image_set, cfg=None):
    """The image half of load_ucas_aod / parse_annotation (ucas_aod.py:75-89,131-147).  The reference opens every image for
    its size; the test mapper takes it from the decoded file."""
    out = []
    for s in ([image_set] if isinstance(image_set, str) else image_set):
        with open(os.path.join(root, "ImageSets", "%s.txt" % s)) as f:
            lines = f.read().splitlines()
        for img_id in _first_n(lines, cfg):
            out.append({"file_name": os.path.join(root, "AllImages", "%s.png" % img_id), "image_id": img_id[1:]})   # "P0001" -> "0001"
    return out


def _data_dir(data_dir):
    if data_dir is None:
        if "DAFNE_DATA_DIR" not in os.environ:
            raise KeyError("DAFNE_DATA_DIR is not set (the reference reads the datasets' root from it)")
        data_dir = os.environ["DAFNE_DATA_DIR"]
    return data_dir


def _register(name, loader, evaluator_type, root_dir, image_root, **extra):
    if name in DatasetCatalog:
        DatasetCatalog.remove(name)
    DatasetCatalog.register(name, loader)
    MetadataCatalog.get(name).set(evaluator_type=evaluator_type, root_dir=root_dir, image_root=image_root,
                                  is_test="test" in name, **extra)


def register_dota(cfg=None, data_dir=None):
    data_dir = _data_dir(data_dir)
    for version in ("1", "1_5"):
        for split in ("train", "val", "test"):
            for size in ("600", "800", "1024", "1300", "1600", "2048"):
                name = "dota_%s_%s_%s" % (version, split, size)
                root = os.path.join(data_dir, "dota_%s_split" % version, split + size)
                jf = os.path.join(root, "DOTA%s_%s%s.json" % (version, split, size))
                img = os.path.join(root, "images")
                _register(name, lambda jf=jf, img=img: load_dota_images(jf, img, cfg), "dota", root, img, json_file=jf)


def register_hrsc(cfg=None, data_dir=None):
    root = os.path.join(_data_dir(data_dir), "hrsc")
    for split in ("train", "val", "test", "trainval"):
        _register("hrsc_%s" % split, lambda split=split: load_hrsc_images(root, split, cfg), "hrsc", root, os.path.join(root, "images"))


def register_ucas_aod(cfg=None, data_dir=None):
    root = os.path.join(_data_dir(data_dir), "UCAS-AOD")
    for split in ("train", "val", "test", "trainval"):
        # (the reference's metadata says images/; its loader reads AllImages/, ucas_aod.py:79)
        _register("ucas_aod_%s" % split, lambda split=split: load_ucas_aod_images(root, split, cfg), "ucas_aod", root,
                  os.path.join(root, "images"))


def register_all(cfg=None, data_dir=None):
    register_dota(cfg, data_dir)
    register_hrsc(cfg, data_dir)
    register_ucas_aod(cfg, data_dir)
