from .backbone import build_dafne_resnet_fpn_backbone  # noqa: F401
from .dafne import DAFNe  # noqa: F401
from .one_stage_detector import OneStageDetector  # noqa: F401
