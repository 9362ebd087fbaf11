"""`OneStageDetector` meta-architecture on the MI355X engine.

Same registry name and call contract as the reference
(dafne/modeling/one_stage_detector.py:34-107 on top of detectron2's
ProposalNetwork [recalled, SURVEY appendix B]): a list of {"image": uint8 CHW BGR,
"height", "width"} in, a list of {"instances": Instances} out.

forward() is one fused device pipeline -- uint8 image -> normalise/pad kernel ->
ResNet-FPN -> head -> decode/top-k -> rotated NMS -> rescale/clip/gather -- with a
single host sync at the very end (detection counts).
"""
import ctypes
import os

import torch
from torch import nn

from .. import _lib, engine
from ..utils.host import usable_cpus
from .. import postprocess as pp
from ..registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY
from ..structures import ImageList
from .dafne.dafne import head_levels


_STREAMS = {}


SUBBATCH_WEIGHTS = {2: (5, 3), 3: (3, 2, 3)}


def subbatch_bounds(n, splits):
    """Image ranges of the `splits` sub-batches of a batch of n: deliberately UNEQUAL sizes (weights 3 : 2 : 3, or 5 : 3 for two;
    largest-remainder apportionment, every sub-batch >= 1 image).  Equal sub-batches run the same launch list at the same pace,
    so the streams stay in phase and every kernel shares the chip with its own twin (tower beside tower: both at the power limit;
    res4 beside res4: both write-bound in the same phase).  Sub-batches of different lengths drift against each other and the
    kernel kinds mix: at batch 8, R101, 1024^2: 4+4 1333, 5+3 1361, 3+2+3 1365-1374 images/s (R50: 1636 / 1670 / 1667; batch 16:
    8+8 1388, 6+4+6 1396; scratch/r05_uneq_ab.sh, profiles/NOTES_r05.md).  An image's result does not depend on its sub-batch
    (test_an_image_gets_the_same_detections_in_any_batch).  DAFNE_SPLIT_SIZES="a,b,.." overrides (A/B runs)."""
    env = os.environ.get("DAFNE_SPLIT_SIZES")
    if env:
        sz = [int(v) for v in env.split(",")]
        if sum(sz) == n and len(sz) == splits and min(sz) >= 1:
            return [sum(sz[:k]) for k in range(splits + 1)]
    w = SUBBATCH_WEIGHTS.get(splits)
    if w is None or n < 2 * splits:
        return [(k * n) // splits for k in range(splits + 1)]
    tot = float(sum(w))
    sz = [int(n * wk / tot) for wk in w]
    rem = sorted(range(splits), key=lambda k: (-(n * w[k] / tot - sz[k]), k))
    for k in rem[: n - sum(sz)]:
        sz[k] += 1
    return [sum(sz[:k]) for k in range(splits + 1)]


def _shared_stream(device, kind, k):
    """Process-wide HIP streams of the pipelined path.  They are shared by every detector instance (TTA
    wrapper, a second model in the same process): each new stream may land on a hardware queue that
    is already in use, and two sub-batches on one queue serialise."""
    key = (str(device), kind, k)
    if key not in _STREAMS:
        # ALL of them are created and bound to their hardware queues together, at the first request (a throw-away submission each
        # and a host wait).  ROCm binds a stream to a hardware queue at its first submission, and graph capture brings streams of
        # its own: a compute stream that was first used AFTER another layout's graphs had been captured (model(batch) three
        # times, then the streamed loop) landed on a queue it shared, and two of the three sub-batches serialised -- 1200
        # instead of 1395 img/s for the rest of the process (round 5, scratch/order_probe.py)
        made = []
        for kd, kk in [("compute", 0), ("compute", 1), ("compute", 2), ("side", 0)] + [(kind, k)]:
            kkey = (str(device), kd, kk)
            if kkey not in _STREAMS:
                _STREAMS[kkey] = torch.cuda.Stream(device=device, priority=-1 if kd == "compute" else 0)
                made.append(_STREAMS[kkey])
        for st in made:
            with torch.cuda.stream(st):
                torch.zeros(1, device=device)
        for st in made:
            st.synchronize()
    return _STREAMS[key]


@META_ARCH_REGISTRY.register()
class OneStageDetector(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, None)
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(
            cfg, self.backbone.output_shape())
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD, dtype=torch.float32).view(-1, 1, 1), False)
        self.top_module = None
        if cfg.MODEL.TOP_MODULE.NAME:
            raise NotImplementedError("MODEL.TOP_MODULE is not used by any released config")
        self.depth = cfg.MODEL.RESNETS.DEPTH
        self._packed = None
        self._plans = {}
        self._graphs = {}
        self.__dict__.pop("_plan_lru", None)      # (its entries reference the caches they describe)
        self._act_q8 = None         # fp8 model: calibrated activation scales {weight key: in_qscale} (calibrate_fp8)
        self.side_stream = None
        self._consts = {}
        self._last_head = None
        self.use_graphs = False     # optional: replay each sub-batch's dense plan from a HIP graph (no gain
                                    # measured at batch 8: the GPU, not the host, is the bottleneck)
        self.eval()

    @property
    def device(self):
        return self.pixel_mean.device

    # ------------------------------------------------------------ engine state
    def invalidate(self):
        """Call after changing parameters: weights are re-packed lazily."""
        self._packed = None
        self._plans = {}
        self._graphs = {}
        self.__dict__.pop("_plan_lru", None)      # (its entries reference the caches they describe)
        self._act_q8 = None
        self._pending = None
        self.__dict__.pop("_deferred", None)
        self.__dict__.pop("_deferred_res", None)
        self.__dict__.pop("_stream_q", None)
        self.__dict__.pop("_staging", None)          # pinned / device staging of forward_streamed's host tiles
        self.__dict__.pop("_counts_ring", None)
        self._consts = {}
        if hasattr(self, "_pipe"):
            self._pipe = {}
        self.backbone.invalidate()
        self.proposal_generator.dafne_head.invalidate()

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def _weights(self):
        if self._packed is None:
            self._packed = engine.pack_model_weights(self.state_dict(), self.depth, self.device,
                                                     weight_dtype=self.cfg.ENGINE.WEIGHT_DTYPE,
                                                     fp8_kernel=self.cfg.ENGINE.FP8_CONV3X3_KERNEL)
        return self._packed

    # ------------------------------------------------------------ fp8 activation scales (BASELINE config 5)
    def fp8_act_scales(self):
        """{weight key: in_qscale} of the layers whose plain (not GroupNorm-fed) input is quantised to e4m3 on load:
        res4 / res5 3x3 layers, FPN output convolutions, the two tower layers that read FPN features.  None until
        calibrate_fp8 / set_fp8_act_scales ran.  Persist them with the weights (they are part of the model)."""
        return None if self._act_q8 is None else dict(self._act_q8)

    def set_fp8_act_scales(self, scales):
        """Install persisted activation scales ({weight key: in_qscale}, what fp8_act_scales() returned / what
        checkpoint.load_weights finds under "fp8_act_scales") instead of calibrating: the scales are part of the fp8 model."""
        if self.cfg.ENGINE.WEIGHT_DTYPE != "fp8_e4m3":
            raise RuntimeError("set_fp8_act_scales: ENGINE.WEIGHT_DTYPE is %r" % (self.cfg.ENGINE.WEIGHT_DTYPE,))
        scales = self.check_fp8_act_scale_keys(scales)
        engine.check_act_qscales(scales)
        self._act_q8 = scales
        self._packed["act_q8"] = dict(self._act_q8)
        self._plans = {}
        self._graphs = {}
        self.__dict__.pop("_plan_lru", None)      # (its entries reference the caches they describe)
        if hasattr(self, "_pipe"):
            self._pipe = {}

    def check_fp8_act_scale_keys(self, scales):
        """Does `scales` ({weight key: in_qscale}) name exactly the layers this model quantises on load?  Raises ValueError
        otherwise, touches nothing (checkpoint.load_weights calls it BEFORE copying a tensor: the layer set depends on the
        architecture -- the packed-weight keys -- not on the values).  Returns the scales as {str: float}."""
        P = self._weights()
        scales = {str(k): float(v) for k, v in dict(scales).items()}
        # the layers calibrate_fp8 would scale: every 3x3 layer with e4m3 weights whose input is NOT a GroupNorm output
        # (res4 / res5 conv2, FPN outputs, the two tower layers that read FPN features)
        want = set(k[:-4] for k in P if k.endswith(".fp8") and not k.endswith(".frag")
                   and (k.startswith("res") or k.startswith("fpn_output") or k in ("cls_tower.0.fp8", "center_tower.0.fp8")))
        # ... plus whatever else a calibration plan of this build probes (corners_tower.0 when the center tower's GroupNorm is
        # not fused into it: DAFNE_FUSE_GN=0 / a kernel-selection fallback): any layer with e4m3 weights may carry a scale,
        # the plain-input ones above must
        allowed = set(k[:-4] for k in P if k.endswith(".fp8") and not k.endswith(".frag"))
        if not (want <= set(scales) <= allowed):
            raise ValueError("fp8 activation scales for %d layers, the model has %d plain-input fp8 layers (missing %s, unknown %s)"
                             % (len(scales), len(want), sorted(want - set(scales))[:4], sorted(set(scales) - allowed)[:4]))
        return scales

    def calibrate_fp8(self, images_u8, valid_hw=None, layout_hwc=False, group=None):
        """Static activation calibration of the fp8 model on one batch (uint8 CUDA images as detect_packed takes them): a
        calibration plan runs those layers on the bf16 kernels and records max |input| per layer; in_qscale = the largest
        power of two with 2 * amax * in_qscale <= 448 (engine.act_qscale_from_amax).  Explicit by default
        (cfg.ENGINE.FP8_ACT_CALIBRATION == "explicit": detect_packed raises until this or set_fp8_act_scales ran).  In a
        multi-process job (torch.distributed initialised) the per-layer amax values are MAX-reduced over the ranks of
        `group` before the scales are derived, so every rank serves the SAME quantised model whatever its shard holds:
        every rank must call this (it is a collective)."""
        if self.cfg.ENGINE.WEIGHT_DTYPE != "fp8_e4m3":
            raise RuntimeError("calibrate_fp8: ENGINE.WEIGHT_DTYPE is %r" % (self.cfg.ENGINE.WEIGHT_DTYPE,))
        L = _lib.load()
        if layout_hwc:
            n, h, w, _ = images_u8.shape
        else:
            n, _, h, w = images_u8.shape
        hn, wn = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        P = dict(self._weights())
        P.pop("act_q8", None)
        calib = {}
        nc = self.proposal_generator.dafne_head.num_classes
        with torch.cuda.device(images_u8.device):
            plan = engine.DensePlan(P, n, hn, wn, self.depth, nc, self.device, calib=calib)
            mean = (ctypes.c_float * 3)(*[float(v) for v in self.cfg.MODEL.PIXEL_MEAN])
            std = (ctypes.c_float * 3)(*[float(v) for v in self.cfg.MODEL.PIXEL_STD])
            vt = None
            if valid_hw is not None and any(tuple(v) != (h, w) for v in valid_hw):
                vt = torch.tensor([tuple(v) for v in valid_hw], dtype=torch.int32).to(self.device)
            _lib.check(L.dafne_preprocess_image_hip(_lib.ptr(images_u8.contiguous()), int(layout_hwc), n, h, w, _lib.ptr(vt),
                                                    mean, std, hn, wn, _lib.ptr(plan.stem_in), _lib.current_stream()),
                       "dafne_preprocess_image_hip")
            plan.run()
            torch.cuda.synchronize()
        calib = engine.reduce_amax_over_ranks(calib, group, device=self.device)
        self._act_q8 = {k: engine.act_qscale_from_amax(v) for k, v in calib.items()}
        self._packed["act_q8"] = dict(self._act_q8)
        self._plans = {}
        self._graphs = {}
        self.__dict__.pop("_plan_lru", None)      # (its entries reference the caches they describe)
        if hasattr(self, "_pipe"):
            self._pipe = {}
        return dict(self._act_q8)

    def _dev_const(self, values, dtype, shape):
        """Small constant device tensors (per-image sizes) are uploaded once and reused: a
        pageable host->device copy would make the host wait for the stream every call."""
        key = (dtype, shape, tuple(map(tuple, values)))
        t = self._consts.get(key)
        if t is None:
            t = torch.tensor(values, dtype=dtype).reshape(shape).to(self.device)
            if len(self._consts) > 256:
                self._consts.clear()
            self._consts[key] = t
        return t

    def _lru_get(self, cache, key, build):
        """cache[key], built on first use; the caches of launch plans (buffers of a whole network at one batch shape: 0.3-5 GB
        each) keep the cfg.ENGINE.MAX_PLANS most recently used shapes EACH and, together (the one-stream plans and the sub-batch
        pipelines share one budget; advisor, round 5), at most cfg.ENGINE.MAX_PLAN_BYTES of device memory as the allocator counts it
        around the build -- on datasets where nearly every batch has its own (H, W) the entry count alone let the caches grow to
        whatever 48 + 48 shapes happen to weigh.  Before an entry is dropped the device is synchronised: its launches may still be
        queued, and its buffers go back to the allocator."""
        import collections
        lru = self.__dict__.setdefault("_plan_lru", collections.OrderedDict())      # (id(cache), key) -> [cache, bytes], oldest first
        tag = (id(cache), key)
        if key in cache:
            cache[key] = cache.pop(key)              # dicts keep insertion order: most recently used last
            if tag in lru:
                lru.move_to_end(tag)
            return cache[key]
        cap = max(2, int(getattr(self.cfg.ENGINE, "MAX_PLANS", 48)))
        if len(cache) >= cap:
            torch.cuda.synchronize(self.device)
            while len(cache) >= cap:
                old = next(iter(cache))
                cache.pop(old)
                lru.pop((id(cache), old), None)
        on_gpu = self.device.type == "cuda"
        self._weights()                              # (packed once per model, outside a plan's account)
        m0 = torch.cuda.memory_allocated(self.device) if on_gpu else 0
        cache[key] = build()
        lru[tag] = [cache, max(0, (torch.cuda.memory_allocated(self.device) if on_gpu else 0) - m0)]
        cap_bytes = int(getattr(self.cfg.ENGINE, "MAX_PLAN_BYTES", 0) or 0)
        if cap_bytes > 0 and sum(v[1] for v in lru.values()) > cap_bytes and len(lru) > 1:
            torch.cuda.synchronize(self.device)
            for old in list(lru):
                if old == tag or sum(v[1] for v in lru.values()) <= cap_bytes:
                    break
                c, _ = lru.pop(old)
                c.pop(old[1], None)
        return cache[key]

    def plan(self, n, h, w, slot=0, graph=False):
        def build():
            nc = self.proposal_generator.dafne_head.num_classes
            return engine.DensePlan(self._weights(), n, h, w, self.depth, nc, self.device)
        p = self._lru_get(self._plans, (n, h, w, slot), build)
        if graph:
            p.capture()
        return p

    # ------------------------------------------------------------ fused path
    def detect_packed(self, images_u8, valid_hw=None, out_hw=None, layout_hwc=False, do_postprocess=True,
                      pipelined=False, splits=1, stream_offset=0, graphs=None, defer=False, even=False, post_per_split=False):
        """images_u8: device uint8 [N,3,H,W] (or [N,H,W,3] with layout_hwc) BGR.
        valid_hw: optional per-image (h, w) true sizes; out_hw: optional per-image
        requested output (height, width).  Returns (rows [N,k_cap,18], counts [N])
        on the device, no host synchronisation.

        pipelined=True: the dense part runs as `splits` contiguous sub-batches on their own
        HIP streams, enqueued layer by layer (independent images: the prologue / write-burst
        bubbles of one sub-batch's kernel are filled by another's); decode, rotated NMS and
        gather run once on the whole batch on a side stream, overlapping the next call's
        convolutions (two plan sets / head-output buffers alternate).
        The returned tensors are then produced on `self.side_stream`: wait on it (or
        torch.cuda.synchronize()) before reading them.  stream_offset rotates the compute streams the
        sub-batches use: consecutive calls with different offsets (the TTA wrapper's chunks) run concurrently.
        graphs: replay the sub-batches' dense launches from HIP graphs (None: cfg.ENGINE.HIP_GRAPHS).  Worth it for a loop over
        one shape (host enqueue 1.9 -> 0.4 ms per step); the TTA wrapper's 27 views of 9 shapes pass False (GPU-bound at 26.7
        ms per image either way, and every shape's capture costs ~10 ms).
        defer=True (pipelined only; a LOOP's form): this call enqueues its convolutions and then the decode / NMS / rescale of
        the PREVIOUS deferred call, which starts on the side stream when this call's sub-batches reach their head towers --
        the persistent tower kernel leaves CUs idle (232 of 256 workgroups) that the post-process kernels fill, while beside
        the backbone's chip-wide launches they cost 3.7 % of the step (scratch/no_post.py, defer_post.py: +1.4 .. +4 %).
        Returns the PREVIOUS call's (rows, counts) -- None on the first call; flush_deferred() enqueues and returns the last.
        even=True: sub-batches of EQUAL size (a lone step with a host wait behind it ends when its longest stream does; the unequal
        sizes of subbatch_bounds pay only in a loop whose steps overlap -- or with post_per_split).
        post_per_split=True (immediate form only; forward()'s, round 6): decode / rotated NMS / rescale run per SUB-BATCH on the side
        stream, smallest sub-batch first, each as soon as its own convolutions are done -- with unequal sub-batches the post-process
        of the short one runs under the long one's convolutions and only the long one's own share trails the step.  Per-image
        results are those of the whole-batch post-process (every image is decoded and suppressed on its own)."""
        if not images_u8.is_cuda or images_u8.dtype != torch.uint8:
            raise RuntimeError("detect_packed needs a uint8 CUDA tensor (the MI355X engine has no CPU path)")
        if self.cfg.ENGINE.WEIGHT_DTYPE == "fp8_e4m3" and self._act_q8 is None:
            mode = self.cfg.ENGINE.FP8_ACT_CALIBRATION
            if mode == "first_batch":
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    # calibrate_fp8 is a collective (amax MAX-reduce): hidden inside the first detect_packed it deadlocks as
                    # soon as one rank has an empty shard and never gets here
                    raise RuntimeError("ENGINE.FP8_ACT_CALIBRATION='first_batch' is single-process only: in a multi-process job "
                                       "call calibrate_fp8(batch) on EVERY rank (a collective) or install set_fp8_act_scales")
                self.calibrate_fp8(images_u8, valid_hw=valid_hw, layout_hwc=layout_hwc)
            elif mode != "off":
                raise RuntimeError("fp8 model without activation scales: call calibrate_fp8(batch) or set_fp8_act_scales(scales) "
                                   "first (ENGINE.FP8_ACT_CALIBRATION=%r; 'off' runs only the GroupNorm-fed tower layers in "
                                   "e4m3, 'first_batch' calibrates on whatever batch comes first)" % (mode,))
        L = _lib.load()
        if layout_hwc:
            n, h, w, _ = images_u8.shape
        else:
            n, _, h, w = images_u8.shape
        hn, wn = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        mean = (ctypes.c_float * 3)(*[float(v) for v in self.cfg.MODEL.PIXEL_MEAN])
        std = (ctypes.c_float * 3)(*[float(v) for v in self.cfg.MODEL.PIXEL_STD])
        all_full = valid_hw is None or all(tuple(v) == (h, w) for v in valid_hw)
        if valid_hw is None:
            valid_hw = [(h, w)] * n
        if out_hw is None:
            out_hw = valid_hw
        sizes = self._dev_const([(vh, vw, oh, ow, vh, vw) for (vh, vw), (oh, ow) in zip(valid_hw, out_hw)],
                                torch.float32, (n, 6))
        outs = self.proposal_generator.dafne_outputs
        strides = self.proposal_generator.fpn_strides

        def dense(imgs, lo, hi, plan):
            vt = None if all_full else self._dev_const(valid_hw[lo:hi], torch.int32, (hi - lo, 2))
            _lib.check(L.dafne_preprocess_image_hip(_lib.ptr(imgs), int(layout_hwc), hi - lo, h, w, _lib.ptr(vt),
                                                    mean, std, hn, wn, _lib.ptr(plan.stem_in),
                                                    _lib.current_stream()), "dafne_preprocess_image_hip")
            plan.run()

        if defer and not pipelined:
            raise ValueError("detect_packed(defer=True) is a form of the pipelined step")
        with torch.cuda.device(images_u8.device):
            images_u8 = images_u8.contiguous()
            if not defer and self.__dict__.get("_deferred") is not None:
                # a deferred post-process is pending and this call is not part of that loop: run it first (its plan set's
                # head outputs must be read before anything reuses them); flush_deferred() hands the result out
                self.__dict__["_deferred_res"] = self._run_deferred(self.__dict__.pop("_deferred"), ())
            if not pipelined:
                plan = self.plan(n, hn, wn)
                genv = os.environ.get("DAFNE_HIP_GRAPHS")
                if (self.cfg.ENGINE.HIP_GRAPHS if graphs is None else bool(graphs)) and genv != "0" and plan.graph is None:
                    # a shape's SECOND call captures its launch list (the first ran eagerly: streams bind to hardware queues
                    # at their first submission, see the pipelined branch): model([one image]) in a loop -- the reference's
                    # own evaluation / benchmark loop, tools/benchmark.py:117-145 -- is host-bound when ~200 launches are
                    # enqueued one by one (bench.py latency_b1)
                    plan._uses = getattr(plan, "_uses", 0) + 1
                    if plan._uses >= 2:
                        plan.capture()
                dense(images_u8, 0, n, plan)
                self._last_head = plan.head          # (tests: the head outputs the returned detections were decoded from)
                if getattr(self.cfg.ENGINE, "CHECK_FINITE", False):
                    # the FPN maps first: behind them GroupNorm-on-load computes max(a x + b, 0), which turns a NaN into 0 again
                    for l, (ft, lg, dc, ce) in enumerate(zip(plan.features, plan.head.logits, plan.head.delta_ctr, plan.head.center)):
                        for nme, t in (("FPN map", ft.t), ("logits", lg), ("delta / ctrness", dc), ("center", ce)):
                            if not bool(torch.isfinite(t).all()):
                                raise _lib.DafneHipError("ENGINE.CHECK_FINITE: non-finite %s at level %d (weights / input overflow? a NaN "
                                                         "accumulator of a non-ReLU layer turns into -inf)" % (nme, l))
                return outs.predict_packed(head_levels(plan.head, strides), sizes=sizes,
                                           scale_corners=do_postprocess)
            # ---- pipeline: [preprocess, convs (split over `splits` streams), decode] | [NMS, gather]
            splits = max(1, min(int(splits), n))
            main = torch.cuda.current_stream()
            if self.side_stream is None:
                self.side_stream = _shared_stream(images_u8.device, "side", 0)
                self._pipe = {}
            key = (n, hn, wn, splits, "even") if (even and splits > 1) else (n, hn, wn, splits)
            if post_per_split and not defer and splits > 1:
                key = key + ("pps",)

            def build_pipe():
                nc = self.proposal_generator.dafne_head.num_classes
                bounds = [(k * n) // splits for k in range(splits + 1)] if even else subbatch_bounds(n, splits)
                # two complete plan sets (A/B) with their own head-output buffers: decode + NMS of
                # call i run on the side stream while call i+1's convolutions already write set B
                hos, plan_sets = [], []
                # plan sets (each with its own head-output buffers): two for a batch that is split over the compute streams; FOUR
                # for a batch too small to split (one image per call), whose consecutive calls run on alternating streams -- a
                # set is reusable only when its post-process has finished, which starts at the NEXT call's head towers
                nsets = 2 if splits > 1 else max(2, int(os.environ.get("DAFNE_B1_SETS", "4")))
                for _ in range(nsets):
                    ho = engine.HeadOutputs(n, hn, wn, nc, self._weights()["scales"], self.device)
                    hos.append(ho)
                    plan_sets.append([engine.DensePlan(self._weights(), bounds[k + 1] - bounds[k], hn, wn, self.depth,
                                                       nc, self.device, head_outputs=ho.views(bounds[k], bounds[k + 1]),
                                                       shared_gpu=True)
                                      for k in range(splits)])
                # compute streams are high-priority: ROCm maps normal-priority streams onto a small shared set
                # of hardware queues, and two sub-batches landing on one queue serialise (dense part alone,
                # 4 splits: 714 -> 960 img/s); the high-priority pool gives each its own queue.  With the
                # post-process stream in the mix 3 splits measured best (scratch/split_sweep.sh)
                return {"i": 0, "cs": [_shared_stream(images_u8.device, "compute", k) for k in range(splits)], "ho": hos,
                        "plans": plan_sets, "bounds": bounds, "cand": [None] * nsets, "done": [None] * nsets, "runs": [0] * nsets,
                        "cand_split": [[None] * splits for _ in range(nsets)]}
            if key not in self._pipe:
                self._lru_get(self._pipe, key, build_pipe)
                # plan building enqueued buffer fills, weight packing and uploads on the caller's stream: finished before any
                # sub-batch stream touches the plans (once per shape; the steps themselves never wait for the host).  The
                # streams are ordered behind it by `inputs_ready` as well; this makes a new shape's first step independent of it
                torch.cuda.synchronize(images_u8.device)
            else:
                self._lru_get(self._pipe, key, build_pipe)
            st = self._pipe[key]
            slot = st["i"] % len(st["plans"])
            st["i"] += 1
            cs, plans, bounds = st["cs"], st["plans"][slot], st["bounds"]
            self._last_head = st["ho"][slot]
            if stream_offset:
                cs = [_shared_stream(images_u8.device, "compute", (int(stream_offset) + k) % 3) for k in range(splits)]
            inputs_ready = torch.cuda.Event()
            inputs_ready.record(main)
            vts = []
            dbg = getattr(self, "_dbg_events", None)        # scratch/pipe_events.py: timestamps of the step's phases

            def mark(tag, k, stream):
                if dbg is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(stream)
                    dbg.append((st["i"], tag, k, e))
            for k in range(splits):
                lo, hi = bounds[k], bounds[k + 1]
                cs[k].wait_event(inputs_ready)
                if st["done"][slot] is not None:
                    cs[k].wait_event(st["done"][slot])    # side stream finished with plan set `slot` (2 calls ago)
                # the batch was allocated on the caller's stream but is read on cs[k], possibly long after the caller
                # dropped it (the read queues behind the previous call's network): tell the caching allocator
                images_u8.record_stream(cs[k])
                with torch.cuda.stream(cs[k]):
                    vt = None if all_full else self._dev_const(valid_hw[lo:hi], torch.int32, (hi - lo, 2))
                    vts.append(vt)
                    sub = images_u8[lo:hi]
                    _lib.check(L.dafne_preprocess_image_hip(_lib.ptr(sub), int(layout_hwc), hi - lo, h, w, _lib.ptr(vt),
                                                            mean, std, hn, wn, _lib.ptr(plans[k].stem_in),
                                                            _lib.current_stream()), "dafne_preprocess_image_hip")
                mark("pre", k, cs[k])
            # layer-interleaved enqueue: launch j of every sub-batch before launch j+1, so that the
            # prologue / epilogue bubbles of one sub-batch's kernel are filled by another's
            sp = [ctypes.c_void_p(s.cuda_stream) for s in cs]
            # (sub-batches of different sizes may differ by a launch: the library picks kernels by tile count)
            genv = os.environ.get("DAFNE_HIP_GRAPHS")                # A/B runs: 1 / 0 overrides the config
            want_graphs = self.cfg.ENGINE.HIP_GRAPHS if graphs is None else bool(graphs)
            if (self.use_graphs or want_graphs or genv == "1") and genv != "0" and graphs is not False:
                # every sub-batch's ~200 dense launches replayed from ONE HIP graph per stream (captured on first use: kernels,
                # arguments and buffers of a plan are static): ~10 host calls per step instead of ~600.  The launch order
                # inside a stream is the plan's; across streams the hardware queues interleave as before (the host runs
                # many steps ahead either way).
                # A plan set's FIRST step always runs eagerly and graphs are captured from its second use on: ROCm binds a
                # stream to a hardware queue at its first submission, and torch's capture stream, created first, took one of
                # the few queues the three sub-batch streams need for themselves (measured: capture in the very first step
                # left the whole process at 1010-1050 images/s, eager and graph steps alike; eager first: 1300 both ways).
                eager_first = st["runs"]
                if eager_first[slot] == 0:
                    tower_evs = self._enqueue_eager(plans, cs, sp, splits, defer)
                elif defer:
                    tower_evs = []
                    for k in range(splits):
                        with torch.cuda.stream(cs[k]):
                            plans[k].capture_parts()
                            plans[k].graph_parts[0].replay()
                            e = torch.cuda.Event()
                            e.record(cs[k])                   # this sub-batch is at its head towers
                            tower_evs.append(e)
                            plans[k].graph_parts[1].replay()
                else:
                    tower_evs = []
                    for k in range(splits):
                        if plans[k].graph is None:
                            with torch.cuda.stream(cs[k]):
                                plans[k].capture()
                        with torch.cuda.stream(cs[k]):
                            plans[k].graph.replay()
                eager_first[slot] += 1
            else:
                tower_evs = self._enqueue_eager(plans, cs, sp, splits, defer)
            if defer:
                ends = []
                for k in range(splits):
                    ev = torch.cuda.Event()
                    ev.record(cs[k])
                    ends.append(ev)
                prev = self.__dict__.get("_deferred")
                self.__dict__["_deferred"] = {"st": st, "slot": slot, "ends": ends, "sizes": sizes, "do": do_postprocess, "main": main,
                                              "strides": strides}
                return self._run_deferred(prev, tower_evs) if prev is not None else None
            for k in range(splits):
                mark("end", k, cs[k])
            if post_per_split and splits > 1:
                evs = []
                for k in range(splits):
                    ev = torch.cuda.Event()
                    ev.record(cs[k])
                    evs.append(ev)
                with torch.cuda.stream(self.side_stream):
                    parts = [None] * splits
                    for k in sorted(range(splits), key=lambda q: (bounds[q + 1] - bounds[q], q)):      # the shortest sub-batch ends first
                        lo, hi = bounds[k], bounds[k + 1]
                        self.side_stream.wait_event(evs[k])
                        cand = outs.decode_packed(head_levels(plans[k].head, strides), out=st["cand_split"][slot][k])
                        st["cand_split"][slot][k] = cand
                        parts[k] = outs.select_packed(cand, sizes=sizes[lo:hi], scale_corners=do_postprocess)
                    res = (torch.cat([r for r, _ in parts]), torch.cat([c for _, c in parts]))
                    done = torch.cuda.Event()
                    done.record(self.side_stream)
                st["done"][slot] = done
                for t in res:
                    t.record_stream(main)
                return res
            with torch.cuda.stream(self.side_stream):
                for k in range(splits):
                    ev = torch.cuda.Event()
                    ev.record(cs[k])
                    self.side_stream.wait_event(ev)
                mark("dec0", 0, self.side_stream)
                cand = outs.decode_packed(head_levels(st["ho"][slot], strides), out=st["cand"][slot])
                st["cand"][slot] = cand
                mark("dec1", 0, self.side_stream)
                res = outs.select_packed(cand, sizes=sizes, scale_corners=do_postprocess)
                mark("nms1", 0, self.side_stream)
                done = torch.cuda.Event()
                done.record(self.side_stream)
            st["done"][slot] = done
            for t in res:                     # allocated on the side stream, consumed on the caller's
                t.record_stream(main)
            return res

    @staticmethod
    def _enqueue_eager(plans, cs, sp, splits, defer):
        """Launch j of every sub-batch before launch j + 1; defer: an event per stream where its head begins."""
        evs = []
        for j in range(max(len(p.calls) for p in plans)):
            for k in range(splits):
                if j < len(plans[k].calls):
                    if defer and j == plans[k].head_start:
                        e = torch.cuda.Event()
                        e.record(cs[k])
                        evs.append(e)
                    plans[k].calls[j](sp[k])
        return evs

    def _run_deferred(self, p, tower_evs):
        """Decode + rotated NMS + rescale of a deferred step on the side stream: behind that step's convolutions and behind
        `tower_evs` (the step after it has reached its head towers)."""
        st, slot = p["st"], p["slot"]
        outs = self.proposal_generator.dafne_outputs
        with torch.cuda.stream(self.side_stream):
            for ev in list(p["ends"]) + list(tower_evs):
                self.side_stream.wait_event(ev)
            cand = outs.decode_packed(head_levels(st["ho"][slot], p["strides"]), out=st["cand"][slot])
            st["cand"][slot] = cand
            res = outs.select_packed(cand, sizes=p["sizes"], scale_corners=p["do"])
            done = torch.cuda.Event()
            done.record(self.side_stream)
        st["done"][slot] = done
        for t in res:                     # allocated on the side stream, consumed on the caller's
            t.record_stream(p["main"])
        return res

    def flush_deferred(self):
        """(rows, counts) of the last detect_packed(defer=True) call -- its post-process is enqueued now -- or None."""
        p = self.__dict__.pop("_deferred", None)
        if p is not None:
            return self._run_deferred(p, ())
        return self.__dict__.pop("_deferred_res", None)

    def _pack_inputs(self, batched_inputs, staged=False):
        """list[{"image": uint8 CHW BGR, "height", "width"}] -> (device uint8 batch [n,3,H,W], valid (h, w) per image, requested
        output (height, width) per image): ImageList.from_tensors' zero padding to the batch maximum (one_stage_detector.py:
        100-107); the normalisation and the /32 padding happen in the engine's load kernel.
        staged: host images go through a pinned staging buffer and an asynchronous copy (two buffers alternate; the host
        does not wait), so that the upload of batch i + 1 runs under the network of batch i.  Images that already live on
        the device are stacked there."""
        dev = self.device
        imgs = [x["image"] for x in batched_inputs]
        n = len(imgs)
        if not all(i.dtype == torch.uint8 for i in imgs):
            raise NotImplementedError("the engine takes uint8 images, as the reference's data loader yields them")
        hs = [int(i.shape[1]) for i in imgs]
        ws = [int(i.shape[2]) for i in imgs]
        H, W = max(hs), max(ws)
        same = all(h == H and w == W for h, w in zip(hs, ws))
        if all(i.is_cuda for i in imgs):
            batch = torch.stack(imgs) if same else torch.zeros(n, 3, H, W, dtype=torch.uint8, device=dev)
            if not same:
                for k, im in enumerate(imgs):
                    batch[k, :, : hs[k], : ws[k]] = im
        elif staged:
            # the staging copies below run on torch's intra-op pool; inference_on_dataset caps that pool for the duration of
            # the loop (utils.host.capped_torch_threads: with one worker per visible CPU the workers' spin-wait starves the HIP
            # runtime's completion threads -- 50-200 ms stalls every few batches, 107-170 images/s instead of 1085)
            st = self.__dict__.get("_staging")
            if st is None:
                st = self.__dict__.setdefault("_staging", {"i": 0, "pin": [None, None], "free": [None, None], "dev": [None, None, None]})
            slot = st["i"] & 1
            st["i"] += 1
            need = n * 3 * H * W
            # ONE flat pinned buffer per slot and one flat device buffer per ring position, grown to the largest batch seen and
            # sliced: shortest-edge resizing (HRSC, UCAS-AOD) gives nearly every batch its own (H, W), and a buffer set per
            # shape would pin host memory and hold HBM without bound (advisor, round 4)
            if st["free"][slot] is not None:
                st["free"][slot].synchronize()           # the copy that last read this buffer (two calls ago) is done
            if st["pin"][slot] is None or st["pin"][slot].numel() < need:
                st["pin"][slot] = torch.empty(need + need // 4, dtype=torch.uint8).pin_memory()
            pinned = st["pin"][slot][:need].view(n, 3, H, W)
            if not same:
                pinned.zero_()
            for k, im in enumerate(imgs):
                pinned[k, :, : hs[k], : ws[k]].copy_(im)
            # the copy rides the CALLER's stream, which carries nothing else in the streamed loop (the network runs on the
            # sub-batch streams, the post-process on the side stream), so it overlaps the previous batch; a stream of its own
            # would be the fifth one on four hardware queues (two of them then share a queue and serialise).
            # Device batches come from a ring of three persistent buffers (this call's, the batch in flight, the one whose
            # results the previous call handed back: forward_streamed waits for batch i - 1's post-process before it
            # returns, so the buffer of batch i - 2 is free).  A buffer that is outgrown goes back to the caching allocator,
            # which keeps it until the streams it was recorded on (detect_packed: record_stream) are past it.
            d = st["i"] % 3
            if st["dev"][d] is None or st["dev"][d].numel() < need:
                st["dev"][d] = torch.empty(need + need // 4, dtype=torch.uint8, device=dev)
            batch = st["dev"][d][:need].view(n, 3, H, W)
            batch.copy_(pinned, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
            st["free"][slot] = done
        elif n == 1:
            batch = imgs[0].to(dev, non_blocking=True).unsqueeze(0)
        else:
            batch = torch.zeros(n, 3, H, W, dtype=torch.uint8, device=dev)
            for k, im in enumerate(imgs):
                batch[k, :, : hs[k], : ws[k]] = im.to(dev, non_blocking=True)
        valid = list(zip(hs, ws))
        out_hw = [(int(x.get("height", hs[k])), int(x.get("width", ws[k]))) for k, x in enumerate(batched_inputs)]
        return batch, valid, out_hw

    def forward(self, batched_inputs, do_postprocess=True):
        if self.training:
            raise NotImplementedError("training is outside the scope of the MI355X inference engine")
        batch, valid, out_hw = self._pack_inputs(batched_inputs)
        splits = min(int(os.environ.get("DAFNE_FWD_SPLITS", "2")), max(1, int(self.cfg.ENGINE.PIPELINE_SPLITS)))
        if len(batched_inputs) >= 2 and splits >= 2:
            # the call detectron2's loop makes (tools/plain_train_net.py:316-336: outputs = model(inputs)) runs on sub-batch
            # streams as well, immediate post-process.  TWO sub-batches of EQUAL size: the host waits behind every call, so a
            # call ends when its longest stream does (batch 8, R101, 1024^2: 4+4 6.80 ms per call, 5+3 6.87, 3+2+3 7.80;
            # scratch/fwd_sync_probe.py) -- the unequal three-way split of the streamed loop pays only where steps overlap.
            # Result-neutral: an image gets the same bits in any batch composition (DESIGN section 5,
            # test_an_image_gets_the_same_detections_in_any_batch)
            # (round 6: the post-process per sub-batch, the short sub-batch's under the long one's convolutions -- detect_packed(
            # post_per_split=True), DAFNE_FWD_PPS=1 -- measured 7.05-7.16 ms per call of 8 against 6.94 for this form (4+4, one
            # whole-batch post-process): two decode / NMS passes cost more than the overlap returns; scripts/fwd_sync_probe.py)
            pps = os.environ.get("DAFNE_FWD_PPS", "0") == "1"
            rows, counts = self.detect_packed(batch, valid_hw=valid, out_hw=out_hw, do_postprocess=do_postprocess,
                                              pipelined=True, splits=splits, even=not pps, post_per_split=pps)
            # the rows are produced on the side stream and the host needs them now: a HOST wait on that stream.  (A device-side
            # wait_stream of the caller's stream, followed by the read-back's own wait, cost 0.85 ms per call once all the
            # shared streams are bound: 7.70 against 6.84 ms per call of 8, scratch/fwd_phase_probe.py)
            self.side_stream.synchronize()
        else:
            rows, counts = self.detect_packed(batch, valid_hw=valid, out_hw=out_hw, do_postprocess=do_postprocess)
        insts = pp.rows_to_instances(rows, counts, out_hw)
        return [{"instances": r} for r in insts]

    # ------------------------------------------------------------ streamed form of forward()
    def forward_streamed(self, batched_inputs, do_postprocess=True):
        """forward() for an evaluation LOOP (detectron2's inference_on_dataset: `for inputs in loader: outputs = model(inputs);
        evaluator.process(inputs, outputs)`, called from tools/plain_train_net.py:316-336): ENQUEUES this batch on the layout
        bench.py times -- cfg.ENGINE.PIPELINE_SPLITS sub-batches on concurrent streams; the decode / rotated NMS / rescale of
        the PREVIOUS batch starts on the side stream when this batch reaches its head towers (detect_packed(defer=True)) -- and
        returns the outputs of the OLDEST batch in flight once two later ones are enqueued: call i returns batch i - 2's outputs
        (None on the first two calls), whose post-process was enqueued a whole call ago and has finished without the host
        waiting.  flush() hands out the batches still in flight, oldest first, one per call, then None.
        Same per-image results as forward() (an image gets the same bits in any batch: DESIGN section 5).
        evaluation.inference.inference_on_dataset drives this."""
        if self.training:
            raise NotImplementedError("training is outside the scope of the MI355X inference engine")
        splits = max(1, int(self.cfg.ENGINE.PIPELINE_SPLITS))
        batch, valid, out_hw = self._pack_inputs(batched_inputs, staged=True)
        q = self.__dict__.setdefault("_stream_q", [])       # batches in flight, oldest first: [out_hw, packed results or None]
        # a batch too small to split (the reference's own loop: ONE image per call, tools/plain_train_net.py:316-336) alternates
        # between two compute streams, call by call: the convolutions of image i + 1 run beside those of image i on the other
        # plan set -- at batch 1 a res4 block is 32 tiles for 256 CUs
        rot = 0
        if min(splits, len(batched_inputs)) == 1 and os.environ.get("DAFNE_B1_ROTATE", "1") != "0":
            rot = self.__dict__.get("_stream_rot", 0)
            self.__dict__["_stream_rot"] = (rot + 1) % max(1, int(os.environ.get("DAFNE_B1_STREAMS", "3")))
        res = self.detect_packed(batch, valid_hw=valid, out_hw=out_hw, do_postprocess=do_postprocess,
                                 pipelined=True, splits=splits, defer=True, stream_offset=rot)
        if res is not None:                                  # the previous call's batch: its post-process was enqueued just now
            self._stage_counts(q[-1], res)
        q.append([out_hw, None])
        if len(q) > 2:
            return self._finish_streamed(q.pop(0))
        return None

    def flush(self):
        """Outputs of the oldest forward_streamed() batch still in flight (None when there is none): call until None."""
        q = self.__dict__.get("_stream_q") or []
        if not q:
            return None
        if q[-1][1] is None:                                 # the newest batch's post-process has not been enqueued yet
            res = self.flush_deferred()
            if res is not None:
                self._stage_counts(q[-1], res)
        return self._finish_streamed(q.pop(0))

    def _stage_counts(self, entry, res):
        """The detection counts of a batch go to a pinned host buffer behind its NMS (4 bytes per image)."""
        rows, counts = res
        with torch.cuda.stream(self.side_stream):
            ring = self.__dict__.setdefault("_counts_ring", {"i": 0, "buf": [None] * 4})
            ck = ring["i"] % 4                                  # four pinned buffers: up to three batches in flight + the one the
            ring["i"] += 1                                      # caller may still be reading; grown to the largest batch, sliced
            if ring["buf"][ck] is None or ring["buf"][ck].numel() < counts.numel() or ring["buf"][ck].dtype != counts.dtype:
                ring["buf"][ck] = torch.empty(max(counts.numel(), 16), dtype=counts.dtype).pin_memory()
            counts_h = ring["buf"][ck][:counts.numel()].view(counts.shape)
            counts_h.copy_(counts, non_blocking=True)
            # ... and the packed rows themselves (batch x k_cap x 18 fp32: 0.6 MB for 8 images): the Instances handed out later
            # carry a host twin, so the evaluator's per-image .to(cpu) copies nothing (postprocess.rows_to_instances)
            rb = ring.setdefault("rows", [None] * 4)
            if rb[ck] is None or rb[ck].numel() < rows.numel() or rb[ck].dtype != rows.dtype:
                rb[ck] = torch.empty(max(rows.numel(), 1024), dtype=rows.dtype).pin_memory()
            rows_h = rb[ck][:rows.numel()].view(rows.shape)
            rows_h.copy_(rows, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.side_stream)
        entry[1] = (rows, counts_h, ready, rows_h)

    @staticmethod
    def _finish_streamed(entry):
        out_hw, packed = entry
        if packed is None:
            raise RuntimeError("forward_streamed: a batch left the queue before its post-process was enqueued")
        rows, counts_h, ready, rows_h = packed
        ready.synchronize()                      # that batch's post-process (side stream) is done; later batches keep running
        return [{"instances": r} for r in pp.rows_to_instances(rows, counts_h.clone(), out_hw, host_rows=rows_h)]

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        assert not self.training
        return self.forward(batched_inputs, do_postprocess)

    def preprocess_image(self, batched_inputs):
        """Normalize, pad and batch the input images (one_stage_detector.py:100-107);
        kept for API parity -- forward() fuses this into the stem's load kernel."""
        images = [x["image"].to(self.device) for x in batched_inputs]
        images = [(x.float() - self.pixel_mean) / self.pixel_std for x in images]
        return ImageList.from_tensors(images, self.backbone.size_divisibility)
