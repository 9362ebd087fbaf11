"""Config surface of the engine: `get_cfg()` with the reference's key names.

Drop-in for `dafne.config.get_cfg()` (dafne/config/config.py:4-13,
dafne/config/defaults.py:8-151) for every key the inference path reads, plus
the detectron2 keys it depends on (MODEL.RESNETS.*, MODEL.FPN.*, INPUT.*,
MODEL.PIXEL_MEAN/STD, TEST.AUG.*).  YAML files with `_BASE_` inheritance and the
reference's full dumps (configs/pre-trained/*.yaml) load as they are; unknown
keys (e.g. GLOBAL.HACK) are accepted rather than rejected.
"""
import copy
import os

import yaml


class CfgNode(dict):
    """Attribute-style nested dict (the subset of yacs.CfgNode the path uses)."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v
        object.__setattr__(self, "_frozen", False)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError("Attempted to set %s on a frozen CfgNode" % k)
        self[k] = v

    def clone(self):
        c = CfgNode(copy.deepcopy(dict(self)))
        return c

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_dict(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path):
        self.merge_from_dict(_load_yaml_with_base(path))

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "override list must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            if isinstance(val, str):
                try:
                    val = yaml.safe_load(val)
                except yaml.YAMLError:
                    pass
            node[parts[-1]] = val

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self))


def _load_yaml_with_base(path):
    with open(path) as f:
        d = yaml.safe_load(f) or {}
    base = d.pop("_BASE_", None)
    if base:
        if not os.path.isabs(base):
            base = os.path.join(os.path.dirname(path), base)
        merged = CfgNode(_load_yaml_with_base(base))
        merged.merge_from_dict(d)
        return merged
    return d


def _defaults():
    dafne = dict(
        # head geometry
        NUM_CLASSES=15, IN_FEATURES=["p3", "p4", "p5", "p6", "p7"], FPN_STRIDES=[8, 16, 32, 64, 128],
        TOP_LEVELS=2, NUM_CLS_CONVS=4, NUM_BOX_CONVS=4, NUM_SHARE_CONVS=0, NORM="GN",
        USE_SCALE=True, USE_RELU=True, USE_DEFORMABLE=False, PRIOR_PROB=0.01,
        CORNER_PREDICTION="center-to-corner", CORNER_TOWER_ON_CENTER_TOWER=True,
        MERGE_CORNER_CENTER_PRED=False, CENTERNESS="oriented", CENTERNESS_ALPHA=5,
        CENTERNESS_USE_IN_SCORE=True, CTR_ON_REG=True, YIELD_PROPOSAL=False,
        # inference post-process
        INFERENCE_TH_TEST=0.05, INFERENCE_TH_TRAIN=0.05, NMS_TH=0.1,
        PRE_NMS_TOPK_TEST=2000, PRE_NMS_TOPK_TRAIN=2000, POST_NMS_TOPK_TEST=1000,
        POST_NMS_TOPK_TRAIN=1000, THRESH_WITH_CTR=False, SORT_CORNERS=True,
        SORT_CORNERS_DATALOADER=True, ENABLE_FPN_STRIDE_NORM=True,
        # training-only keys, kept so reference YAMLs merge cleanly
        LOSS_SMOOTH_L1_BETA=1.0 / 9.0, ENABLE_LOSS_MODULATION=True, ENABLE_LOSS_LOG=True,
        ENABLE_LEVEL_SIZE_FILTERING=True, ENABLE_IN_BOX_CHECK=True, LOSS_ALPHA=0.25, LOSS_GAMMA=2.0,
        SIZES_OF_INTEREST=[64, 128, 256, 512], LOSS_LAMBDA_NORM=True,
        LOSS_LAMBDA=dict(CORNERS=1.0, BOX=1.0, LTRB=1.0, CTR=1.0, CLS=1.0, CENTER=1.0),
        CENTER_SAMPLE=True, CENTER_SAMPLE_ONLY=False, COMBINE_CENTER_SAMPLE=True, POS_RADIUS=2.0,
        LOC_LOSS_TYPE="smoothl1",
    )
    model = dict(
        META_ARCHITECTURE="OneStageDetector", DEVICE="cuda", WEIGHTS="",
        PIXEL_MEAN=[103.53, 116.28, 123.675], PIXEL_STD=[1.0, 1.0, 1.0],
        KEYPOINT_ON=False, LOAD_PROPOSALS=False, MASK_ON=False, MOBILENET=False,
        BACKBONE=dict(NAME="build_dafne_resnet_fpn_backbone", FREEZE_AT=2, ANTI_ALIAS=False),
        RESNETS=dict(DEPTH=50, NORM="FrozenBN", NUM_GROUPS=1, WIDTH_PER_GROUP=64,
                     STRIDE_IN_1X1=True, RES5_DILATION=1, RES2_OUT_CHANNELS=256,
                     STEM_OUT_CHANNELS=64, OUT_FEATURES=["res3", "res4", "res5"],
                     DEFORM_INTERVAL=1, DEFORM_ON_PER_STAGE=[False] * 4, DEFORM_MODULATED=False,
                     DEFORM_NUM_GROUPS=1),
        FPN=dict(IN_FEATURES=["res3", "res4", "res5"], OUT_CHANNELS=256, NORM="", FUSE_TYPE="sum"),
        PROPOSAL_GENERATOR=dict(NAME="DAFNe", MIN_SIZE=0),
        TOP_MODULE=dict(NAME="", DIM=16),
        DAFNE=dafne,
    )
    return dict(
        VERSION=2, EXPERIMENT_NAME="dafne", OUTPUT_DIR="./output", SEED=-1,
        MODEL=model,
        INPUT=dict(FORMAT="BGR", MIN_SIZE_TEST=1024, MAX_SIZE_TEST=1024, MIN_SIZE_TRAIN=[1024],
                   MAX_SIZE_TRAIN=1024, RESIZE_TYPE="shortest-edge",
                   # dafne/config/defaults.py:10,26-27,121-134 (data-loader keys; kept so cfg reads never fail)
                   HFLIP_TRAIN=True, MIN_AREA=10, MIN_SIDE=2, ROTATION_AUG_ANGLES=[0.0, 90.0, 180.0, 270.0],
                   RESIZE_HEIGHT_TRAIN=0, RESIZE_WIDTH_TRAIN=0, RESIZE_HEIGHT_TEST=0, RESIZE_WIDTH_TEST=0,
                   ROTATION_AUG_SAMPLE_STYLE="choice", USE_COLOR_AUGMENTATIONS=False),
        # DOTA_REMOVE_CONTAINER_CRANE: dafne/config/defaults.py:148, read by dota_evaluation.py:120,312
        DATASETS=dict(TRAIN=[], TEST=[], DOTA_REMOVE_CONTAINER_CRANE=False),
        DATALOADER=dict(NUM_WORKERS=4),
        DEBUG=dict(OVERFIT_NUM_IMAGES=-1),
        SOLVER=dict(IMS_PER_BATCH=8, AMP=dict(ENABLED=False), OPTIMIZER="sgd"),
        TEST=dict(DETECTIONS_PER_IMAGE=2000, IOU_TH=0.5, NUM_PRED_VIS=20, EXPECTED_RESULTS=[],
                  AUG=dict(ENABLED=False, MIN_SIZES=[1024], MAX_SIZE=1200, FLIP=True,
                           HFLIP=True, VFLIP=True, ROTATION_ANGLES=[])),
        # engine-specific knobs (not in the reference)
        ENGINE=dict(WEIGHT_DTYPE="bf16", ACT_DTYPE="bf16",
                    # fp8 model (the activation scales are part of the model: they change its outputs).  "explicit" (default):
                    # detect_packed RAISES until calibrate_fp8(batch) or set_fp8_act_scales(scales) has been called -- results
                    # never depend on image order, batch size or world size; "off" = only the GroupNorm-fed tower layers take e4m3
                    # activations; "first_batch" = opt-in convenience: calibrate on the first batch detect_packed sees
                    FP8_ACT_CALIBRATION="explicit",
                    # fp8 model: the kernel of its 3x3 layers -- "patch" (conv3x3_patch_fp8_kernel), the only one since round 6
                    # (rounds 3-5 also had "rp8", a resident-patch form that won alone and lost in the timed layout: removed).  A second
                    # kernel would round the same sums differently, so the choice is part of the model: the key stays
                    FP8_CONV3X3_KERNEL="patch",
                    # sub-batches on concurrent streams in the streamed evaluation loop (OneStageDetector.forward_streamed /
                    # evaluation.inference.inference_on_dataset): the layout bench.py times.  Round 5: three sub-batches of UNEQUAL
                    # size (one_stage_detector.subbatch_bounds: 8 images = 3 + 2 + 3), so that the streams drift out of phase
                    PIPELINE_SPLITS=int(os.environ.get("DAFNE_PIPELINE_SPLITS", "3")), MAX_PLANS=48,
                    # ... and together at most this many bytes of device memory (one-stream plans + sub-batch pipelines; 0 = no byte
                    # bound).  96 GB: a third of the MI355X's 288 GB for launch plans, the rest for the caller
                    MAX_PLAN_BYTES=96 << 30,
                    # replay every sub-batch's dense launches from a HIP graph in the pipelined / streamed step (one host call
                    # per stream and step instead of ~200)
                    HIP_GRAPHS=True,
                    # debug: after every detect_packed of the one-stream path, assert that all head outputs are finite (one host
                    # sync per call).  The convolution epilogues implement "no ReLU" as max(v, -inf): under IEEE maxNum a NaN
                    # accumulator of a non-ReLU layer (FPN lateral / output, P6, tower convolutions in front of GroupNorm) becomes
                    # -inf instead of propagating, so a corrupted checkpoint or an overflow shows as empty / wrong detections rather
                    # than as NaNs -- this switch makes it loud
                    CHECK_FINITE=False),
    )


def get_cfg():
    """A fresh default config (same call as dafne.config.get_cfg)."""
    return CfgNode(_defaults())


def load_cfg(path, opts=None):
    cfg = get_cfg()
    cfg.merge_from_file(path)
    if opts:
        cfg.merge_from_list(list(opts))
    return cfg
