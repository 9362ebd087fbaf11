"""Minimal stand-ins for the detectron2 structures the path touches.

Same attribute / method surface as detectron2.structures.{Instances, Boxes,
ImageList} and detectron2.layers.ShapeSpec for what DAFNe uses (reference call
sites: dafne/modeling/dafne/dafne_outputs.py:879-903, one_stage_detector.py:79-107).
When real detectron2 is importable its classes are used instead.
"""
from collections import namedtuple

import torch

try:  # pragma: no cover - detectron2 is absent in the build image
    from detectron2.layers import ShapeSpec as _D2ShapeSpec
    from detectron2.structures import Boxes as _D2Boxes, ImageList as _D2ImageList, Instances as _D2Instances
    HAVE_D2 = True
except Exception:  # noqa: BLE001
    HAVE_D2 = False


class _ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


class _Boxes:
    def __init__(self, tensor):
        if tensor.numel() == 0:
            tensor = tensor.reshape(-1, 4)
        self.tensor = tensor

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        t = self.tensor[item]
        return _Boxes(t.reshape(1, -1) if t.dim() == 1 else t)

    def to(self, *a, **k):
        return _Boxes(self.tensor.to(*a, **k))

    def clone(self):
        return _Boxes(self.tensor.clone())

    @property
    def device(self):
        return self.tensor.device

    def scale(self, sx, sy):
        self.tensor[:, 0::2] *= sx
        self.tensor[:, 1::2] *= sy

    def clip(self, box_size):
        h, w = box_size
        self.tensor[:, 0::2] = self.tensor[:, 0::2].clamp(min=0, max=w)
        self.tensor[:, 1::2] = self.tensor[:, 1::2].clamp(min=0, max=h)

    def nonempty(self, threshold=0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    @staticmethod
    def cat(lst):
        return _Boxes(torch.cat([b.tensor for b in lst], dim=0))


class _Instances:
    def __init__(self, image_size, **kwargs):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        fields = object.__getattribute__(self, "_fields")
        if name not in fields:
            raise AttributeError("Cannot find field '%s' in the given Instances!" % name)
        return fields[name]

    def set(self, name, value):
        n = len(value)
        if len(self._fields):
            assert len(self) == n, "Adding a field of length %d to Instances of length %d" % (n, len(self))
        self._fields[name] = value
        self.__dict__.pop("_cpu_twin", None)          # the host twin describes the fields it was made beside: gone with any edit

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]
        self.__dict__.pop("_cpu_twin", None)

    @staticmethod
    def _field_state(fields):
        """Identity and in-place version of every field tensor (torch bumps `_version` on every in-place write, views included:
        pred_boxes.scale() / .clip(), `scores *= ...`): what the host twin is valid for."""
        st = []
        for n, v in fields.items():
            t = v.tensor if hasattr(v, "tensor") else v
            st.append((n, id(v), id(t), getattr(t, "_version", None)))
        return tuple(st)

    def attach_cpu_twin(self, make):
        """`make()` builds this object's host copy from rows that are already on the host (postprocess.rows_to_instances).  It is
        handed out by `.to("cpu")` only while no field has been set, removed, replaced or written in place since."""
        object.__setattr__(self, "_cpu_twin", (make, self._field_state(self._fields)))

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *a, **k):
        # a host twin (postprocess.rows_to_instances(host_rows=...): the streamed loop copies a batch's packed rows to pinned host
        # memory behind its NMS, once): `.to("cpu")` -- what every evaluator does per image -- is then no device copy at all
        twin = self.__dict__.get("_cpu_twin")
        if twin is not None and not k and len(a) == 1 and isinstance(a[0], (str, torch.device)) and torch.device(a[0]).type == "cpu":
            if twin[1] == self._field_state(self._fields):
                return twin[0]()
            self.__dict__.pop("_cpu_twin", None)      # a detectron2-style hook edited the device fields: copy what is there now
        r = _Instances(self._image_size)
        for n, v in self._fields.items():
            r.set(n, v.to(*a, **k) if hasattr(v, "to") else v)
        return r

    def __getitem__(self, item):
        if isinstance(item, int):
            item = slice(item, None, len(self)) if item >= 0 else slice(item, None, len(self))
        r = _Instances(self._image_size)
        for n, v in self._fields.items():
            r.set(n, v[item])
        return r

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    @staticmethod
    def cat(lst):
        assert len(lst) > 0
        if len(lst) == 1:
            return lst[0]
        r = _Instances(lst[0].image_size)
        for k in lst[0]._fields:
            vs = [i.get(k) for i in lst]
            if isinstance(vs[0], torch.Tensor):
                v = torch.cat(vs, dim=0)
            elif hasattr(type(vs[0]), "cat"):
                v = type(vs[0]).cat(vs)
            else:
                v = [x for l in vs for x in l]
            r.set(k, v)
        return r

    def __repr__(self):
        return "Instances(num_instances=%d, image_size=%s, fields=%s)" % (
            len(self), self._image_size, list(self._fields))


class _ImageList:
    """d2 ImageList.from_tensors: zero pad bottom/right to the batch max rounded
    up to a multiple of size_divisibility; remembers the unpadded sizes."""

    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        sizes = [(t.shape[-2], t.shape[-1]) for t in tensors]
        H = max(s[0] for s in sizes)
        W = max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            H = (H + d - 1) // d * d
            W = (W + d - 1) // d * d
        out = tensors[0].new_full((len(tensors), tensors[0].shape[0], H, W), pad_value)
        for k, t in enumerate(tensors):
            out[k, :, : t.shape[-2], : t.shape[-1]].copy_(t)
        return _ImageList(out, sizes)


if HAVE_D2:  # pragma: no cover
    ShapeSpec, Boxes, Instances, ImageList = _D2ShapeSpec, _D2Boxes, _D2Instances, _D2ImageList
else:
    ShapeSpec, Boxes, Instances, ImageList = _ShapeSpec, _Boxes, _Instances, _ImageList
