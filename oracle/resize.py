"""oracle/resize.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of Pillow's 8-bit bilinear `Image.resize` (libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc), the resampler detectron2's
ResizeShortestEdge applies to uint8 images [recalled], which dafne/modeling/tta.py:71-99 uses for its views.
Pillow is not part of /root/reference; the restatement is pinned against Pillow itself (importable in the build
container: tests/test_oracle_resize.py) and against tests/golden/resize_pil.npz (outputs of PIL 12.2.0).
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """-> (xmin [out], n [out], k [out, ksize] int32)"""
    scale = float(np.float32(in_size)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmins = np.zeros(out_size, dtype=np.int64)
    ns = np.zeros(out_size, dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(xmax, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmins[xx], ns[xx] = xmin, xmax
    return xmins, ns, kk


def _pass(img, out_size):
    """Resample the LAST axis of a uint8 array to out_size."""
    in_size = img.shape[-1]
    xmins, ns, kk = coeffs(in_size, out_size)
    out = np.empty(img.shape[:-1] + (out_size,), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        n = ns[xx]
        acc = (1 << (PRECISION_BITS - 1)) + (src[..., xmins[xx]:xmins[xx] + n] * kk[xx, :n].astype(np.int64)).sum(-1)
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def resize_bilinear_u8(img_chw, new_h, new_w, hflip=False, vflip=False):
    """uint8 [C,H,W] -> uint8 [C,new_h,new_w]: horizontal pass, uint8 intermediate, vertical pass."""
    img = np.ascontiguousarray(img_chw, dtype=np.uint8)
    tmp = _pass(img, new_w) if new_w != img.shape[2] else img
    out = _pass(tmp.transpose(0, 2, 1), new_h).transpose(0, 2, 1) if new_h != img.shape[1] else tmp
    if hflip:
        out = out[:, :, ::-1]
    if vflip:
        out = out[:, ::-1, :]
    return np.ascontiguousarray(out)
