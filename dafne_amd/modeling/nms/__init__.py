from .nms import batched_nms_poly, ml_nms, poly_gpu_nms, poly_iou_pairs  # noqa: F401
