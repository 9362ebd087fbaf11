from .fpn import build_dafne_resnet_fpn_backbone  # noqa: F401
