"""Drop-in for the SWIG module `polyiou` (tools/prepare_dota/polyiou.cpp + polyiou.i) as the reference uses it:

    polyiou.iou_poly(polyiou.VectorDouble(BBGT_keep[index]), polyiou.VectorDouble(bb))      # dafne/evaluation/voc_eval.py:184
    polyiou.VectorDouble([x1, y1, ..., x4, y4]); polyiou.iou_poly(polys[i], polys[j])      # ResultMerge_multi_process.py:38-43,100

Put this directory on PYTHONPATH.  `iou_poly` returns the fp64 value polyiou.cpp:112-133 computes, bit for bit,
evaluated by `dafne_poly_iou_pairs_hip` on the MI355X.  One pair per call costs a device round trip;
`iou_poly_pairs(P, Q)` ([n,8] each) is the batched form the engine's own voc_eval / ResultMerge use.
No CPU path: without a GPU the call raises.
"""
import numpy as np
import torch

from _dafne_amd_lib import check, lib

__all__ = ["VectorDouble", "iou_poly", "iou_poly_pairs"]


class VectorDouble(list):
    """std::vector<double> stand-in: VectorDouble(iterable of numbers); push_back / size like the SWIG proxy."""

    def __init__(self, it=()):
        super().__init__(float(v) for v in it)

    def push_back(self, v):
        self.append(float(v))

    def size(self):
        return len(self)


def iou_poly_pairs(p, q, device_id=0):
    p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 8)
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 8)
    if p.shape != q.shape:
        raise ValueError("p and q must both be [n, 8]")
    n = p.shape[0]
    if n == 0:
        return np.zeros(0, np.float64)
    dev = torch.device("cuda", device_id)
    with torch.cuda.device(dev):
        tp, tq = torch.from_numpy(p).to(dev), torch.from_numpy(q).to(dev)
        out = torch.empty(n, dtype=torch.float64, device=dev)
        check(lib().dafne_poly_iou_pairs_hip(tp.data_ptr(), tq.data_ptr(), n, out.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "dafne_poly_iou_pairs_hip")
        return out.cpu().numpy()


def iou_poly(p, q):
    if len(p) != 8 or len(q) != 8:
        raise ValueError("iou_poly takes two quadrilaterals (8 numbers each)")
    return float(iou_poly_pairs(np.asarray(p, np.float64)[None], np.asarray(q, np.float64)[None])[0])
