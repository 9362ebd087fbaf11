#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Runs only where /root/reference exists (never on the GPU box).  The reference's
dafne_outputs.py / dafne.py / nms.py / sort_corners.py are imported from where
they lie, with structural stand-ins for the packages the image lacks
(detectron2, fvcore, poly_nms, poly_overlaps); the fixtures hold numbers only
(inputs + the reference's outputs), never reference text.

  poly_nms.poly_gpu_nms  := greedy loop over the reference's own compiled
                            polyiou.cpp (oracle/_ref), order =
                            argsort(kind="stable")[::-1]  [DOTA_devkit wrapper
                            semantics, SURVEY appendix B]

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402


# --------------------------------------------------------------------------- stubs
class Boxes:
    def __init__(self, tensor):
        self.tensor = tensor

    def __getitem__(self, i):
        return Boxes(self.tensor[i])

    def __len__(self):
        return self.tensor.shape[0]

    @staticmethod
    def cat(lst):
        return Boxes(torch.cat([b.tensor for b in lst], 0))


class Instances:
    def __init__(self, image_size, **kw):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kw.items():
            self._fields[k] = v

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, k, v):
        self._fields[k] = v

    def __getattr__(self, k):
        f = object.__getattribute__(self, "_fields")
        if k in f:
            return f[k]
        raise AttributeError(k)

    def has(self, k):
        return k in self._fields

    def get_fields(self):
        return self._fields

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def __getitem__(self, i):
        r = Instances(self._image_size)
        for k, v in self._fields.items():
            r._fields[k] = v[i]
        return r

    @staticmethod
    def cat(lst):
        r = Instances(lst[0].image_size)
        for k in lst[0]._fields:
            vs = [x._fields[k] for x in lst]
            r._fields[k] = Boxes.cat(vs) if isinstance(vs[0], Boxes) else torch.cat(vs, 0)
        return r


def poly_gpu_nms(dets, thresh, device_id=0):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    m = dets.shape[0]
    order = np.argsort(dets[:, 8], kind="stable")[::-1]
    poly = dets[order, :8].astype(np.float64)
    dead = np.zeros(m, bool)
    keep = []
    for r in range(m):
        if dead[r]:
            continue
        keep.append(int(order[r]))
        rest = np.nonzero(~dead[r + 1:])[0] + r + 1
        if rest.size:
            iou = oracle.ref_iou_poly_pairs(np.repeat(poly[r:r + 1], rest.size, 0), poly[rest])
            dead[rest[iou > thresh]] = True
    return keep


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Registry:
    def register(self, obj=None):
        if obj is None:
            return lambda o: o
        return obj


def install_stubs():
    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

    _mod("detectron2")
    _mod("detectron2.layers", cat=lambda ts, dim=0: torch.cat(ts, dim), ShapeSpec=ShapeSpec,
         NaiveSyncBatchNorm=torch.nn.BatchNorm2d, Conv2d=torch.nn.Conv2d)
    _mod("detectron2.layers.deform_conv", DeformConv=type("DeformConv", (torch.nn.Module,), {}),
         ModulatedDeformConv=type("ModulatedDeformConv", (torch.nn.Module,), {}))
    _mod("detectron2.structures", Instances=Instances, Boxes=Boxes)
    _mod("detectron2.structures.boxes", Boxes=Boxes)
    _mod("detectron2.utils")
    _mod("detectron2.utils.comm", get_world_size=lambda: 1, get_local_rank=lambda: 0)
    sys.modules["detectron2.utils"].comm = sys.modules["detectron2.utils.comm"]
    _mod("detectron2.modeling")
    _mod("detectron2.modeling.proposal_generator")
    _mod("detectron2.modeling.proposal_generator.build", PROPOSAL_GENERATOR_REGISTRY=_Registry())
    _mod("fvcore")
    _mod("fvcore.nn", sigmoid_focal_loss_jit=None, smooth_l1_loss=None)
    _mod("poly_nms", poly_gpu_nms=poly_gpu_nms)
    _mod("poly_overlaps", poly_overlaps=None)
    for pkg in ("dafne", "dafne.modeling", "dafne.modeling.dafne", "dafne.modeling.nms",
                "dafne.modeling.losses", "dafne.utils", "dafne.layers"):
        m = _mod(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split("."))]


def load_ref(modname):
    path = os.path.join(REF, *modname.split(".")) + ".py"
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) else v


def load_cfg(name, **dafne_over):
    with open(os.path.join(REF, "configs", "pre-trained", name)) as f:
        d = yaml.safe_load(f)
    d["MODEL"]["DAFNE"].update(dafne_over)
    return AttrDict(d)


# ------------------------------------------------------------------- generators
def rrects(n, rng, extent=1024.0, lo=8.0, hi=256.0, jitter=0.5):
    c = rng.uniform(0, extent, (n, 2))
    long_side = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
    w, h = long_side, long_side / rng.uniform(1, 6, n)
    a = rng.uniform(0, np.pi, n)
    ca, sa = np.cos(a), np.sin(a)
    ux = np.stack([w / 2 * ca - h / 2 * sa, -w / 2 * ca - h / 2 * sa,
                   -w / 2 * ca + h / 2 * sa, w / 2 * ca + h / 2 * sa], 1)
    uy = np.stack([w / 2 * sa + h / 2 * ca, -w / 2 * sa + h / 2 * ca,
                   -w / 2 * sa - h / 2 * ca, w / 2 * sa - h / 2 * ca], 1)
    p = np.empty((n, 8))
    p[:, 0::2] = c[:, :1] + ux
    p[:, 1::2] = c[:, 1:] + uy
    return (p + rng.normal(0, jitter, p.shape)).astype(np.float32)


def gen_iou(out):
    rng = np.random.default_rng(101)
    n = 3000
    p = rrects(n, rng, extent=200.0).astype(np.float64)
    q = rrects(n, rng, extent=200.0).astype(np.float64)
    # adversarial tail: identical, degenerate (collinear / repeated vertex), touching,
    # clockwise, self-intersecting, tiny, huge-offset copies
    adv_p, adv_q = [], []
    sq = np.array([0, 0, 1, 0, 1, 1, 0, 1], float)
    adv_p += [sq, sq, sq, sq[::-1].reshape(4, 2)[:, ::-1].reshape(-1), sq, sq * 1e-5, sq]
    adv_q += [sq + 0.5, sq, sq + np.array([1, 0] * 4), sq + 0.25, sq + 2, sq * 1e-5, sq * 0]
    deg = np.array([686, 2976, 709, 2976, 724, 2976, 701, 2976], float)   # polyiou.cpp:137
    adv_p += [deg, deg, np.array([0, 0, 2, 2, 2, 0, 0, 2.0]), np.array([1, 1, 1, 1, 1, 1, 1, 1.0])]
    adv_q += [deg, sq, np.array([0, 0, 2, 0, 2, 2, 0, 2.0]), np.array([1, 1, 1, 1, 1, 1, 1, 1.0])]
    # rbox (1,1,2,10,0) vs (2,1,2,10,0)  (poly_overlaps_test.py:7-24) -> 1/3
    adv_p += [np.array([0, -4, 2, -4, 2, 6, 0, 6.0])]
    adv_q += [np.array([1, -4, 3, -4, 3, 6, 1, 6.0])]
    for k in range(200):   # near-coincident / shared-edge / large class offset cases
        b = rrects(1, rng, extent=64.0)[0].astype(np.float64)
        off = float(rng.integers(0, 16)) * 1300.0
        adv_p.append(b + off)
        adv_q.append(b + off + rng.choice([0.0, 1e-9, 1e-6, 1e-3, 0.5]) * rng.normal(size=8))
    p = np.concatenate([p, np.stack(adv_p)])
    q = np.concatenate([q, np.stack(adv_q)])
    # the NMS only ever sees float32-representable inputs
    p[: n // 2] = p[: n // 2].astype(np.float32)
    q[: n // 2] = q[: n // 2].astype(np.float32)
    iou = oracle.ref_iou_poly_pairs(p, q)
    np.savez_compressed(out, p=p, q=q, iou=iou)
    print("iou fixture:", p.shape, "nonzero", float((iou > 0).mean()))


def gen_sort(out, sc):
    rng = np.random.default_rng(202)
    b = rng.normal(0, 10, (512, 8)).astype(np.float32)
    kat = np.array([[0, 0, 1, 0, 2, 0, 3, 0], [1, 1, 0, 0, 0, 1, 1, 0],
                    [0, 0, 0, 0, 0, 0, 0, 0], [2, 2, 2, 2, 1, 1, 3, 3],
                    [0, 0, 1, 1, 1, 0, 0, 1], [5, 5, 5, 7, 5, 6, 6, 6]], np.float32)
    r = rrects(256, rng, extent=512.0)
    b = np.concatenate([b, kat, r, np.round(r)])
    o = sc.sort_quadrilateral(torch.from_numpy(b)).numpy()
    np.savez_compressed(out, boxes=b, sorted=o)
    print("sort fixture:", b.shape)


def gen_nms(out, nms_mod):
    rng = np.random.default_rng(303)
    cases = {}

    def run(name, boxes, scores, classes, thr=0.1):
        keep = nms_mod.batched_nms_poly(torch.from_numpy(boxes), torch.from_numpy(scores),
                                        torch.from_numpy(classes), thr)
        cases[name + "_boxes"] = boxes
        cases[name + "_scores"] = scores
        cases[name + "_classes"] = classes
        cases[name + "_thr"] = np.float64(thr)
        cases[name + "_keep"] = np.asarray(keep, np.int64)
        print("nms case", name, "M", len(scores), "kept", len(keep))

    for m in (1, 2, 63, 64, 65, 300, 1000):
        b = rrects(m, rng, extent=256.0 if m <= 300 else 512.0)
        s = rng.uniform(0.05, 1, m).astype(np.float32)
        c = rng.integers(0, 15, m).astype(np.int64)
        run("rand%d" % m, b, s, c)
    # ties in score, duplicates, classes 4/5 merged, degenerate boxes, other thresholds
    m = 200
    b = rrects(m, rng, extent=128.0)
    b[50:100] = b[0:50]                                   # exact duplicates
    s = np.round(rng.uniform(0.05, 1, m), 1).astype(np.float32)   # many ties
    c = rng.choice([4, 5], m).astype(np.int64)
    run("ties45", b, s, c)
    b = rrects(m, rng, extent=128.0)
    b[::7, 2:] = np.tile(b[::7, :2], (1, 3))              # point-degenerate quads
    b[3::7, 4:6] = b[3::7, 0:2]                           # triangles
    s = rng.uniform(0.05, 1, m).astype(np.float32)
    c = rng.integers(0, 16, m).astype(np.int64)
    run("degenerate", b, s, c)
    # ResultMerge.py:54-63 known answer: two identical degenerate dets -> keep [0]
    d = np.array([[6.86e2, 2.976e3, 7.09e2, 2.976e3, 7.24e2, 2.976e3, 7.01e2, 2.976e3]] * 2, np.float32)
    run("kat_resultmerge", d, np.array([2.7137e-3, 2.7097e-3], np.float32), np.zeros(2, np.int64))
    b = rrects(400, rng, extent=200.0)
    s = rng.uniform(0.05, 1, 400).astype(np.float32)
    c = np.zeros(400, np.int64)
    run("thr05_oneclass", b, s, c, thr=0.5)
    b = (rrects(400, rng, extent=200.0) - 100.0).astype(np.float32)   # negative coords
    run("negcoords", b, s, rng.integers(0, 16, 400).astype(np.int64))
    np.savez_compressed(out, **cases)


def gen_predict(out, outputs_mod):
    rng = np.random.default_rng(404)
    res = {}
    strides = [8, 16, 32, 64, 128]
    sizes = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]
    variants = [
        ("d10", "dota-1.0_r101_ms.yaml", {}),                        # C15 ctr-thresh sort
        ("d15", "dota-1.5_r101_ms.yaml", {}),                        # C16 no-ctr-thresh nosort
        ("hrsc", "hrsc_r50_ms.yaml", {}),                            # C1 sort
        ("ucas", "ucas_aod_r101_ms.yaml", {}),                       # C2 nosort
        ("d10_topk", "dota-1.0_r101_ms.yaml", {"PRE_NMS_TOPK_TEST": 150, "POST_NMS_TOPK_TEST": 100}),
        ("d15_topk", "dota-1.5_r101_ms.yaml", {"PRE_NMS_TOPK_TEST": 150, "POST_NMS_TOPK_TEST": 100}),
    ]
    for name, cfgfile, over in variants:
        cfg = load_cfg(cfgfile, **over)
        C = cfg.MODEL.DAFNE.NUM_CLASSES
        outs = outputs_mod.DAFNeOutputs(cfg)
        outs.eval()
        N = 2
        logits, regs, ctrs, locs = [], [], [], []
        dafne_mod = sys.modules["dafne.modeling.dafne.dafne"]
        for (h, w), s in zip(sizes, strides):
            logits.append(torch.from_numpy(rng.normal(-3.0, 2.0, (N, C, h, w)).astype(np.float32)))
            regs.append(torch.from_numpy(rng.normal(0, 1.5, (N, 8, h, w)).astype(np.float32)))
            ctrs.append(torch.from_numpy(rng.normal(0, 2.0, (N, 1, h, w)).astype(np.float32)))
            locs.append(dafne_mod.compute_locations(h, w, s, "cpu"))
        with torch.no_grad():
            boxlists = outs.predict_proposals(logits, regs, ctrs, locs, [(256, 256)] * N, [])
        res[name + "_cfg"] = np.array([C, cfg.MODEL.DAFNE.PRE_NMS_TOPK_TEST,
                                       cfg.MODEL.DAFNE.POST_NMS_TOPK_TEST,
                                       int(cfg.MODEL.DAFNE.THRESH_WITH_CTR),
                                       int(cfg.MODEL.DAFNE.SORT_CORNERS)], np.int64)
        res[name + "_thr"] = np.array([cfg.MODEL.DAFNE.INFERENCE_TH_TEST, cfg.MODEL.DAFNE.NMS_TH])
        for l in range(5):
            res["%s_logits%d" % (name, l)] = logits[l].numpy()
            res["%s_reg%d" % (name, l)] = regs[l].numpy()
            res["%s_ctr%d" % (name, l)] = ctrs[l].numpy()
        for i, bl in enumerate(boxlists):
            f = bl.get_fields()
            res["%s_im%d_pred_boxes" % (name, i)] = f["pred_boxes"].tensor.numpy()
            for k in ("pred_corners", "scores", "centerness", "pred_classes", "locations", "fpn_levels"):
                res["%s_im%d_%s" % (name, i, k)] = f[k].numpy()
            print("predict", name, "im", i, "dets", len(bl))
    np.savez_compressed(out, **res)


def gen_head(out, dafne_mod):
    """DAFNeHead forward for seeded weights.  Weights are NOT stored: both sides
    regenerate them with oracle.model.fill_params(seed)."""
    from oracle.model import fill_params
    res = {}
    for name, cfgfile in (("d10", "dota-1.0_r101_ms.yaml"), ("ucas", "ucas_aod_r101_ms.yaml")):
        cfg = load_cfg(cfgfile)
        SS = sys.modules["detectron2.layers"].ShapeSpec
        head = dafne_mod.DAFNeHead(cfg, [SS(channels=256)] * 5)
        head.eval()
        fill_params(head, seed=7)
        rng = np.random.default_rng(505)
        feats = [torch.from_numpy(rng.normal(0, 1, (2, 256, h, w)).astype(np.float32))
                 for h, w in ((12, 16), (6, 8), (3, 4), (2, 2), (1, 1))]
        with torch.no_grad():
            logits, reg, center, _, ctr, _, _ = head(None, feats, None, False)
        for l in range(5):
            res["%s_feat%d" % (name, l)] = feats[l].numpy()
            res["%s_logits%d" % (name, l)] = logits[l].numpy()
            res["%s_reg%d" % (name, l)] = reg[l].numpy()
            res["%s_center%d" % (name, l)] = center[l].numpy()
            res["%s_ctr%d" % (name, l)] = ctr[l].numpy()
        print("head", name, "params", sum(p.numel() for p in head.parameters()))
    np.savez_compressed(out, **res)


def main():
    assert os.path.isdir(REF), "reference tree not present: fixtures can only be made in the build container"
    assert oracle.ref_lib() is not None
    install_stubs()
    sc = load_ref("dafne.utils.sort_corners")
    load_ref("dafne.layers.deform_conv")
    load_ref("dafne.modeling.losses.utils")
    load_ref("dafne.modeling.losses.smooth_l1")
    nms_mod = load_ref("dafne.modeling.nms.nms")
    outputs_mod = load_ref("dafne.modeling.dafne.dafne_outputs")
    dafne_mod = load_ref("dafne.modeling.dafne.dafne")
    torch.manual_seed(0)
    gen_iou(os.path.join(HERE, "iou_pairs.npz"))
    gen_sort(os.path.join(HERE, "sort_corners.npz"), sc)
    gen_nms(os.path.join(HERE, "nms_cases.npz"), nms_mod)
    gen_predict(os.path.join(HERE, "predict_proposals.npz"), outputs_mod)
    gen_head(os.path.join(HERE, "head_forward.npz"), dafne_mod)


if __name__ == "__main__":
    main()
