"""dafne_amd: MI355X (gfx950) inference path of the DAFNe oriented detector.

Python surface mirrors the reference's (`dafne.config.get_cfg`,
`dafne.modeling.{OneStageDetector, build_dafne_resnet_fpn_backbone, DAFNe}`,
`dafne.modeling.nms.nms.{ml_nms, batched_nms_poly}`, `poly_nms.poly_gpu_nms`);
all arithmetic runs in hand-written HIP kernels behind include/dafne_amd.h.
"""
__version__ = "0.1.0"
