"""Shared loader of the shims: binds libdafne_amd.so with ctypes (nothing else of this repository is imported).
Library path: $DAFNE_AMD_LIB, else ../dafne_amd/libdafne_amd.so next to this directory."""
import ctypes
import os

import torch  # noqa: F401  -- BEFORE the CDLL: the library must bind to the HIP runtime copy torch ships

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DAFNE_AMD_LIB") or os.path.join(os.path.dirname(_HERE), "dafne_amd", "libdafne_amd.so")
_L = None


def lib():
    global _L
    if _L is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libdafne_amd.so not found at %s (build it with `python -m dafne_amd.build`; there is no "
                              "CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cd, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
        L.dafne_last_error.restype = ctypes.c_char_p
        L.dafne_poly_nms_workspace_bytes.restype = cs
        L.dafne_poly_nms_workspace_bytes.argtypes = [ci, ci]
        L.dafne_poly_nms_hip.restype = ci
        L.dafne_poly_nms_hip.argtypes = [vp, ci, cd, vp, vp, vp, cs, ci, vp]
        L.dafne_poly_iou_pairs_hip.restype = ci
        L.dafne_poly_iou_pairs_hip.argtypes = [vp, vp, ctypes.c_int64, vp, vp]
        L.dafne_poly_nms_f64_workspace_bytes.restype = cs
        L.dafne_poly_nms_f64_workspace_bytes.argtypes = [ci, ci]
        L.dafne_poly_nms_f64_batched_hip.restype = ci
        L.dafne_poly_nms_f64_batched_hip.argtypes = [vp, vp, ci, ci, cd, ci, vp, vp, vp, cs, ci, vp]
        _L = L
    return _L


def check(rc, what):
    if rc:
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, lib().dafne_last_error().decode(errors="replace")))
