from .loader import (DAFNeTestMapper, DatasetCatalog, InferenceLoader, build_test_loader, inference_resize_shape, list_image_records,
                     read_image)

__all__ = ["DAFNeTestMapper", "DatasetCatalog", "InferenceLoader", "build_test_loader", "inference_resize_shape", "list_image_records",
           "read_image"]
