"""The test-time data path in front of the detector: image files -> `batched_inputs`.

Reference: `build_test_loader(cfg, dataset_name)` (tools/plain_train_net.py:280-313) builds ONE test-time augmentation --
`T.ResizeShortestEdge(MIN_SIZE_TEST, MAX_SIZE_TEST, "choice")` for INPUT.RESIZE_TYPE "shortest-edge", `T.Resize((h, w))`
for "both" (with the *_TRAIN height / width keys: :298-301) -- and hands it to `DAFNeDatasetMapper`
(dafne/data/datasets/dafne_dataset_mapper.py:13-19), i.e. detectron2's `DatasetMapper.__call__` [recalled]:
`read_image(file_name, format=INPUT.FORMAT)` (PIL decode, EXIF orientation, RGB -> BGR), the augmentation
(`ResizeTransform.apply_image` = `PIL.Image.resize(BILINEAR)` on uint8), `image` as a uint8 CHW tensor; `height` / `width`
stay the file's own size (what `detector_postprocess` scales the corners back to).  The ground-truth part of the mapper
(:20-45) feeds the losses only; the evaluators read their annotations themselves (evaluation/dota_evaluation.py).

Here: files are decoded on host worker threads (PIL releases the interpreter lock while it inflates; PNG decoding is serial
work with no GPU form), go through pinned staging buffers to the device as HWC bytes, and the resize + HWC -> CHW transpose
are ONE launch of
`dafne_resize_bilinear_u8_hip` (csrc/resize.hip: bit-exact to Pillow's 8-bit resampler, tests/test_oracle_resize.py), on the
caller's stream -- so `inference_on_dataset(model, build_test_loader(cfg, name), evaluator)` reads like the reference's
`do_test` (:316-336).  Every rank iterates its contiguous shard (detectron2's InferenceSampler [recalled]).
There is no CPU resize path: an image that needs resizing needs the GPU.

Measured (MI355X box: 256 visible CPUs under a cgroup quota of 16; 1024^2 PNG tiles of 2 MB, 27.7 ms of decode each on one core;
R101-FPN, batches of 8, scratch/loader_probe.py): 14 decode threads 440-450 images/s -- the decode supply of the 14 CPUs, with
forward_streamed at 3.5 ms of host time per batch; the same arrays pre-decoded 1160-1240 images/s (the device half -- staging,
upload, transpose -- costs 1-2 ms per batch).  Worker PROCESSES (backend "process": a torch DataLoader, what detectron2's
build_detection_test_loader uses [recalled]) reach 200 images/s: with forked children alive every forward_streamed call takes
23 ms instead of 3.5.  More workers than the CPU quota allows get the whole process group throttled (32 processes: 84 ms per
call), hence the cap at utils.host.usable_cpus() - 2.  Feeding one MI355X from PNG files takes ~40 cores of decode.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.utils.data

from .. import _lib
from ..evaluation.gather import shard_range
from ..modeling.tta import shortest_edge_size
from ..utils.host import usable_cpus

__all__ = ["DAFNeTestMapper", "DatasetCatalog", "InferenceLoader", "build_test_loader", "list_image_records", "read_image",
           "inference_resize_shape"]

IMAGE_EXTENSIONS = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")


class _Catalog:
    """name -> list of records (detectron2's DatasetCatalog [recalled], the part a test loader needs): a record is a dict
    with `file_name`, `image_id` and optionally `height` / `width` (checked against the decoded file, as
    detection_utils.check_image_size does)."""

    def __init__(self):
        self._reg = {}

    def register(self, name, records_or_fn):
        if name in self._reg:
            raise KeyError("dataset %r is already registered" % (name,))
        self._reg[name] = records_or_fn

    def register_image_dir(self, name, root):
        self.register(name, lambda: list_image_records(root))

    def get(self, name):
        if name not in self._reg:
            raise KeyError("dataset %r is not registered (DatasetCatalog.register / register_image_dir)" % (name,))
        r = self._reg[name]
        return list(r() if callable(r) else r)

    def remove(self, name):
        self._reg.pop(name, None)

    def __contains__(self, name):
        return name in self._reg


DatasetCatalog = _Catalog()


def list_image_records(root):
    """Every image file under `root` (sorted, recursive): {"file_name", "image_id"}; image_id = the file's stem, which is what
    the DOTA / HRSC evaluators key their Task1 lines and annotation files on (dota_evaluation.py)."""
    if not os.path.isdir(root):
        raise FileNotFoundError("image directory %r does not exist" % (root,))
    out = []
    for d, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            if f.lower().endswith(IMAGE_EXTENSIONS):
                out.append({"file_name": os.path.join(d, f), "image_id": os.path.splitext(f)[0]})
    return out


def read_image(file_name, fmt="BGR"):
    """detectron2 detection_utils.read_image [recalled]: PIL decode, EXIF orientation applied, converted to RGB (or L),
    channel order flipped for "BGR".  Returns a contiguous uint8 HWC array."""
    from PIL import Image, ImageOps
    with Image.open(file_name) as im:
        im = ImageOps.exif_transpose(im)
        if fmt == "L":
            return np.ascontiguousarray(np.asarray(im.convert("L"))[:, :, None])
        if fmt not in ("BGR", "RGB"):
            raise NotImplementedError("INPUT.FORMAT %r (the released configs use BGR)" % (fmt,))
        a = np.asarray(im.convert("RGB"))
    if fmt == "BGR":
        a = a[:, :, ::-1]
    return np.ascontiguousarray(a)


def inference_resize_shape(cfg, h, w):
    """The (new_h, new_w) of the reference's single test-time augmentation for an (h, w) image (plain_train_net.py:292-304)."""
    rt = cfg.INPUT.RESIZE_TYPE
    if rt == "shortest-edge":
        size = cfg.INPUT.MIN_SIZE_TEST
        if isinstance(size, (list, tuple)):
            if len(size) != 1:
                raise NotImplementedError("INPUT.MIN_SIZE_TEST with several sizes is a random choice per image in the reference; "
                                          "the released configs give one")
            size = size[0]
        if int(size) == 0:                          # detectron2: size 0 disables the resize
            return h, w
        return shortest_edge_size(h, w, int(size), int(cfg.INPUT.MAX_SIZE_TEST))
    if rt == "both":
        nh, nw = int(cfg.INPUT.RESIZE_HEIGHT_TRAIN), int(cfg.INPUT.RESIZE_WIDTH_TRAIN)      # sic: the *_TRAIN keys (:299-300)
        if nh <= 0 or nw <= 0:
            raise ValueError("INPUT.RESIZE_TYPE 'both' needs INPUT.RESIZE_HEIGHT_TRAIN / RESIZE_WIDTH_TRAIN")
        return nh, nw
    raise RuntimeError("Invalid resize-type: %s" % (rt,))


def _to_chw_resized(hwc, new_h, new_w):
    """device uint8 [H,W,C] -> device uint8 [C,new_h,new_w]: resize (Pillow-exact bilinear) and transpose in one launch; a
    same-size image is only transposed."""
    h, w, c = (int(v) for v in hwc.shape)
    if (h, w) == (new_h, new_w):
        return hwc.permute(2, 0, 1).contiguous()
    L = _lib.load()
    with torch.cuda.device(hwc.device):
        out = torch.empty((c, new_h, new_w), dtype=torch.uint8, device=hwc.device)
        nbytes = L.dafne_resize_workspace_bytes(c, h, new_w)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=hwc.device)
        _lib.check(L.dafne_resize_bilinear_u8_hip(_lib.ptr(hwc), 1, c, h, w, new_h, new_w, 0, 0, _lib.ptr(out), _lib.ptr(ws), nbytes,
                                                  _lib.current_stream()), "dafne_resize_bilinear_u8_hip")
    return out


class DAFNeTestMapper:
    """record -> {"file_name", "image_id", "height", "width", "image"}: the inference fields of DAFNeDatasetMapper.
    `decode` is the host half (safe on a worker thread), `finish` the device half (upload, resize, transpose)."""

    def __init__(self, cfg, device=None):
        self.cfg = cfg
        self.image_format = cfg.INPUT.FORMAT
        self.device = torch.device(device) if device is not None else None

    def decode(self, record):
        img = read_image(record["file_name"], self.image_format)
        h, w = int(img.shape[0]), int(img.shape[1])
        for k, v in (("height", h), ("width", w)):
            if k in record and int(record[k]) != v:              # detection_utils.check_image_size [recalled]
                raise ValueError("mismatched %s for %s: the record says %d, the file has %d" % (k, record["file_name"], int(record[k]), v))
        return img

    def finish(self, record, img, staging=None):
        if not torch.is_tensor(img):
            img = torch.from_numpy(img)
        h, w = int(img.shape[0]), int(img.shape[1])
        nh, nw = inference_resize_shape(self.cfg, h, w)
        out = {k: v for k, v in record.items() if k != "annotations"}
        out["height"], out["width"] = h, w
        dev = self.device
        if dev is None or dev.type != "cuda":
            if (nh, nw) != (h, w):
                raise _lib.DafneHipError("test loader: %s needs a resize to %dx%d and the MI355X engine has no CPU path" % (record["file_name"], nh, nw))
            out["image"] = img.permute(2, 0, 1).contiguous()
            return out
        t = img
        if staging is not None:
            t = staging.stage(t)
        out["image"] = _to_chw_resized(t.to(dev, non_blocking=True), nh, nw)
        if staging is not None:
            staging.mark(dev)
        return out

    def __call__(self, record):
        return self.finish(record, self.decode(record))


class _Staging:
    """A ring of pinned host buffers for the uploads: slot k is rewritten only after the copy that last read it has finished."""

    def __init__(self, slots):
        self.slots, self.i = [None] * slots, 0
        self.events = [None] * slots
        self._cur = None

    def stage(self, t):
        # the staging copies run on torch's intra-op pool, which inference_on_dataset caps for the duration of the loop
        # (utils.host.capped_torch_threads; it used to be shrunk here, process-wide and for good)
        k = self.i % len(self.slots)
        self.i += 1
        n = t.numel()
        if self.events[k] is not None:
            self.events[k].synchronize()
        if self.slots[k] is None or self.slots[k].numel() < n:
            self.slots[k] = torch.empty(max(n, 1 << 22), dtype=torch.uint8).pin_memory()
        buf = self.slots[k][:n].view(t.shape)
        buf.copy_(t)
        self._cur = k
        return buf

    def mark(self, dev):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.events[self._cur] = ev


class _DecodeDataset(torch.utils.data.Dataset):
    """record index -> decoded uint8 HWC tensor (the host half of the mapper; runs in the DataLoader's worker processes)."""

    def __init__(self, records, mapper):
        self.records, self.mapper = records, mapper

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        return torch.from_numpy(self.mapper.decode(self.records[i]))


class InferenceLoader:
    """Iterable of `batched_inputs` (lists of mapper outputs) over this rank's shard of `records`.  Decoding runs ahead of
    the consumer on `num_workers` threads (capped at utils.host.usable_cpus() - 2; `prefetch_batches` batches in flight);
    num_workers = 0 decodes in the calling thread; backend "process" uses a torch DataLoader's worker processes instead
    (measured slower next to a live HIP context, see the module text).
    len() = batches of this rank."""

    def __init__(self, cfg, records, batch_size=1, device=None, num_workers=8, prefetch_batches=2, shard=None, backend="thread"):
        if batch_size < 1:
            raise ValueError("batch_size must be positive")
        if backend not in ("process", "thread"):
            raise ValueError("backend must be 'process' or 'thread'")
        self.mapper = DAFNeTestMapper(cfg, device)
        if shard is None:
            import torch.distributed as dist
            shard = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        lo, hi = shard_range(len(records), shard[0], shard[1])
        self.records = list(records[lo:hi])
        self.batch_size = int(batch_size)
        # never more decoders than CPUs this process may use, less two for the enqueueing thread and the HIP runtime: beyond the
        # cgroup quota the scheduler throttles the whole group (32 workers under a quota of 16: forward_streamed took 84 ms)
        self.num_workers = max(0, min(int(num_workers), max(1, usable_cpus() - 2)))
        self.prefetch_batches = max(1, int(prefetch_batches))
        self.backend = backend
        self._staging = _Staging(2 * self.batch_size) if self.mapper.device is not None and self.mapper.device.type == "cuda" else None

    def __len__(self):
        return (len(self.records) + self.batch_size - 1) // self.batch_size

    def _decoded_batches(self):
        recs = self.records
        if self.num_workers == 0 or not recs:
            for i in range(0, len(recs), self.batch_size):
                yield [torch.from_numpy(self.mapper.decode(r)) for r in recs[i:i + self.batch_size]]
        elif self.backend == "process":
            host_mapper = DAFNeTestMapper(self.mapper.cfg, None)           # what the workers need: no device handle crosses the fork
            nw = min(self.num_workers, len(recs))
            # one IMAGE per work item (batch_size=None): the first batch is ready after one decode, not after a worker has
            # decoded eight; the DataLoader hands the samples back in order
            per_worker = max(2, -(-self.prefetch_batches * self.batch_size // nw))
            dl = torch.utils.data.DataLoader(_DecodeDataset(recs, host_mapper), batch_size=None, shuffle=False, num_workers=nw,
                                             prefetch_factor=per_worker)
            cur = []
            for t in dl:
                cur.append(t)
                if len(cur) == self.batch_size:
                    yield cur
                    cur = []
            if cur:
                yield cur
        else:
            prefetch = self.prefetch_batches * self.batch_size
            with ThreadPoolExecutor(max_workers=self.num_workers, thread_name_prefix="dafne-decode") as pool:
                pending, nxt, done = [], 0, 0
                while done < len(recs):
                    while nxt < len(recs) and len(pending) < prefetch + self.batch_size:
                        pending.append(pool.submit(self.mapper.decode, recs[nxt]))
                        nxt += 1
                    n = min(self.batch_size, len(recs) - done)
                    yield [torch.from_numpy(pending.pop(0).result()) for _ in range(n)]
                    done += n

    def __iter__(self):
        done = 0
        for imgs in self._decoded_batches():
            batch = [self.mapper.finish(self.records[done + k], im, self._staging) for k, im in enumerate(imgs)]
            done += len(imgs)
            yield batch


def build_test_loader(cfg, dataset_name, batch_size=1, device=None, num_workers=8, prefetch_batches=2, shard=None, backend="thread"):
    """`build_test_loader(cfg, dataset_name)` of tools/plain_train_net.py:280-313.  dataset_name: a name registered in
    `DatasetCatalog`, or an image directory.  batch_size: images per batch and GPU (detectron2's test loader yields 1; the
    engine's timed layout is quoted on 8).  device: where the mapper leaves `image` (None: the host, valid only when no image
    needs a resize)."""
    if dataset_name in DatasetCatalog:
        records = DatasetCatalog.get(dataset_name)
    elif os.path.isdir(dataset_name):
        records = list_image_records(dataset_name)
    else:
        raise KeyError("%r is neither a registered dataset nor an image directory" % (dataset_name,))
    return InferenceLoader(cfg, records, batch_size=batch_size, device=device, num_workers=num_workers,
                           prefetch_batches=prefetch_batches, shard=shard, backend=backend)
