from .loader import (DAFNeTestMapper, DatasetCatalog, InferenceLoader, build_test_loader, inference_resize_shape, list_image_records,
                     read_image)

from .datasets import MetadataCatalog, register_all, register_dota, register_hrsc, register_ucas_aod

__all__ = ["MetadataCatalog", "register_all", "register_dota", "register_hrsc", "register_ucas_aod", "DAFNeTestMapper", "DatasetCatalog", "InferenceLoader", "build_test_loader", "inference_resize_shape", "list_image_records",
           "read_image"]
