"""Drop-in for the external CUDA extension `poly_nms` (DOTA_devkit/poly_nms_gpu) the reference imports at
dafne/modeling/nms/nms.py:6 and calls at :91:

    from poly_nms import poly_gpu_nms
    keep = poly_gpu_nms(boxes_np, iou_threshold, comm.get_local_rank())

Put this directory on PYTHONPATH.  Same arguments, same return (list of kept row indices, descending score; equal
scores: larger row index first = np.argsort(kind="stable")[::-1]); the work is `dafne_poly_nms_hip` of libdafne_amd.so
on the MI355X (fp64 polyiou.cpp arithmetic on the float32 rows).  No CPU path: without a GPU the call raises.
"""
import numpy as np
import torch

from _dafne_amd_lib import check, lib

__all__ = ["poly_gpu_nms"]


def poly_gpu_nms(dets, thresh, device_id=0):
    dets = np.ascontiguousarray(dets, dtype=np.float32)          # [M, 9]: 8 corner coordinates + score
    if dets.ndim != 2 or dets.shape[1] != 9:
        raise ValueError("dets must be [M, 9]")
    m = dets.shape[0]
    if m == 0:
        return []
    L = lib()
    dev = torch.device("cuda", device_id)
    with torch.cuda.device(dev):
        d = torch.from_numpy(dets).to(dev)
        keep = torch.empty(m, dtype=torch.int64, device=dev)
        n = torch.zeros(1, dtype=torch.int32, device=dev)
        nbytes = L.dafne_poly_nms_workspace_bytes(1, m)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(L.dafne_poly_nms_hip(d.data_ptr(), m, float(thresh), keep.data_ptr(), n.data_ptr(), ws.data_ptr(), nbytes, 0,
                                   torch.cuda.current_stream().cuda_stream), "dafne_poly_nms_hip")
        return keep[: int(n.item())].cpu().tolist()
