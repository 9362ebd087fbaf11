#!/usr/bin/env python3
"""tests/golden/resize_pil.npz: outputs of PIL.Image.resize(BILINEAR) (the resampler behind detectron2's
ResizeShortestEdge for uint8 images) on seeded random images.  Run where Pillow is importable."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(7)
fx = {"pil_version": np.array(Image.__version__ if hasattr(Image, "__version__") else "?")}
CASES = [(37, 53, 19, 91), (64, 64, 64, 100), (100, 80, 45, 80), (128, 96, 320, 240), (97, 131, 31, 40), (50, 50, 50, 50),
         (200, 120, 7, 5)]
for i, (h, w, nh, nw) in enumerate(CASES):
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    out = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    fx["in_%d" % i], fx["out_%d" % i] = img, out
np.savez_compressed(os.path.join(HERE, "resize_pil.npz"), **fx)
print("wrote resize_pil.npz,", len(CASES), "cases")
