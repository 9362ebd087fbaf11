"""ctypes binding of libdafne_amd.so (include/dafne_amd.h).

There is no fallback: if the HIP library is missing or a call fails, the caller
gets an exception.  Nothing here (or anywhere under dafne_amd/) touches oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DAFNE_AMD_LIB") or os.path.join(_HERE, "libdafne_amd.so")   # override: debug builds
_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_size_t = ctypes.c_size_t
c_double = ctypes.c_double
c_float = ctypes.c_float
c_u32 = ctypes.c_uint32
c_i32 = ctypes.c_int32


class LevelDesc(ctypes.Structure):
    _fields_ = [("d_logits", c_void_p), ("d_delta", c_void_p), ("d_center", c_void_p),
                ("d_ctrness", c_void_p), ("logits_ps", c_i32), ("delta_ps", c_i32),
                ("center_ps", c_i32), ("ctrness_ps", c_i32), ("H", c_i32), ("W", c_i32), ("stride", c_i32),
                ("scale", c_float)]


class DecodeParams(ctypes.Structure):
    _fields_ = [("n_images", c_i32), ("n_levels", c_i32), ("n_classes", c_i32),
                ("pre_nms_topk", c_i32), ("pre_nms_thresh", c_float), ("thresh_with_ctr", c_i32),
                ("sort_corners", c_i32), ("m_cap", c_i32)]


class ConvSeg(ctypes.Structure):
    _fields_ = [("d_in", c_void_p), ("d_out", c_void_p), ("d_res", c_void_p),
                ("Hin", c_i32), ("Win", c_i32), ("Hout", c_i32), ("Wout", c_i32)]


class ConvParams(ctypes.Structure):
    _fields_ = [("n_images", c_i32), ("n_segs", c_i32), ("Cin", c_i32), ("Cout", c_i32),
                ("KH", c_i32), ("KW", c_i32), ("stride", c_i32), ("pad", c_i32),
                ("flags", c_u32), ("d_weight", c_void_p), ("d_bias", c_void_p),
                ("d_gn_partial", c_void_p), ("d_in_gn_stats", c_void_p), ("d_in_gn_gamma", c_void_p),
                ("d_in_gn_beta", c_void_p), ("d_gn_stats_out", c_void_p), ("d_gn_counters", c_void_p),
                ("gn_eps", c_float)]


class GnSeg(ctypes.Structure):
    _fields_ = [("d_x", c_void_p), ("H", c_i32), ("W", c_i32), ("tile0", c_i32), ("tiles_per_img", c_i32)]


# name -> (restype, argtypes): every symbol include/dafne_amd.h declares
SIGNATURES = {
    "dafne_abi_version": (c_int, []),
    "dafne_last_error": (ctypes.c_char_p, []),
    "dafne_poly_iou_pairs_hip": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "dafne_poly_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dafne_poly_nms_stats_offset": (c_size_t, [c_int, c_int, c_int]),
    "dafne_poly_nms_hip": (c_int, [c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dafne_poly_nms_batched_hip": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_int, c_void_p,
                                           c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dafne_poly_nms_f64_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dafne_poly_nms_f64_batched_hip": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_int, c_void_p,
                                               c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dafne_select_over_all_levels_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                 c_double, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                                 c_int, c_void_p]),
    "dafne_decode_workspace_bytes": (c_size_t, [ctypes.POINTER(DecodeParams), ctypes.POINTER(LevelDesc)]),
    "dafne_decode_levels_hip": (c_int, [ctypes.POINTER(DecodeParams), ctypes.POINTER(LevelDesc)] +
                                [c_void_p] * 8 + [c_void_p, c_size_t, c_void_p]),
    "dafne_sort_quadrilateral_hip": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "dafne_gather_detections_hip": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                             c_void_p]),
    "dafne_conv2d_nhwc_bf16_hip": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p]),
    "dafne_conv2d_nhwc_fp8w_hip": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p, c_float, c_void_p]),
    "dafne_conv2d_num_tiles": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_conv2d_fp8w_num_tiles": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_conv2d_fp8w_tiles_per_image": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p]),
    "dafne_resize_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dafne_resize_bilinear_u8_hip": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                             c_void_p, c_size_t, c_void_p]),
    "dafne_conv2d_kernel_id": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_conv2d_tiles_per_image": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p]),
    "dafne_groupnorm_finalize_hip": (c_int, [ctypes.POINTER(GnSeg), c_int, c_int, c_int, c_void_p, c_void_p,
                                             ctypes.c_float, c_void_p]),
    "dafne_conv2d_cout_pad": (c_int, [c_int]),
    "dafne_conv2d_tile_pixels": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_preprocess_image_hip": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                           ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_int, c_int,
                                           c_void_p, c_void_p]),
    "dafne_conv3x3_c64_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dafne_conv3x3_c256_ok": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_conv3x3_c256_num_tiles": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_conv3x3_c256_tiles_per_image": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p]),
    "dafne_conv3x3_c256_pair_hip": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p,
                                            ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p, c_void_p, c_size_t, c_void_p]),
    "dafne_conv3x3_c256_scratch_bytes": (c_size_t, []),
    "dafne_conv3x3_c256_hip": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p, c_void_p, c_size_t, c_void_p]),
    "dafne_conv2d_wr_ok": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_conv2d_wr_splits": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_conv2d_wr_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg)]),
    "dafne_conv2d_wr_hip": (c_int, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvSeg), c_void_p, c_void_p, c_size_t, c_void_p]),
    "dafne_bottleneck_body_scratch_bytes": (c_size_t, []),
    "dafne_bottleneck_body_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dafne_bottleneck_tail_head_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                               c_void_p, c_void_p, c_void_p]),
    "dafne_bottleneck_tail_head_narrow_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                      c_void_p, c_void_p, c_void_p]),
    "dafne_bottleneck_tail_head_mid_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                   c_void_p, c_void_p, c_void_p]),
    "dafne_bottleneck_proj_tail_head_narrow_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                           c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dafne_bottleneck_block_narrow_scratch_bytes": (c_size_t, []),
    "dafne_bottleneck_block_narrow_hip": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dafne_bottleneck_block_mid_scratch_bytes": (c_size_t, []),
    "dafne_bottleneck_block_mid_hip": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dafne_stem_pool_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dafne_stem_pool_conv1_hip": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p]),
    "dafne_maxpool3x3s2_nhwc_bf16_hip": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dafne_groupnorm_relu_nhwc_bf16_hip": (c_int, [ctypes.POINTER(GnSeg), c_int, c_int, c_int, c_void_p, c_void_p,
                                                   c_void_p, c_void_p, c_float, c_void_p]),
    "dafne_relu_copy_bf16_hip": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
}


DET_ROW = 18


class DafneHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise (loudly) if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DafneHipError(
                "libdafne_amd.so not found at %s -- build it with `python -m dafne_amd.build` "
                "(there is no CPU fallback for the MI355X path)" % LIB_PATH)
        # torch first: its wheel ships its own libamdhip64.so.7 / libhsa-runtime64; the library must bind to
        # THAT runtime instance (same soname -> the loader reuses it).  Loaded the other way round, torch
        # and this library end up on two HIP runtimes and the second one sees no device.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)   # AttributeError if the .so lacks a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dafne_last_error().decode(errors="replace")
        raise DafneHipError("%s failed (code %d): %s" % (what or "dafne_amd call", rc, msg))


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
