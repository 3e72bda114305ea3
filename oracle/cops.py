"""ctypes wrappers of oracle/mmcv_ops.c (tests only)."""
import ctypes

import torch

from . import build as _build

_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        _lib.oracle_nms.restype = ctypes.c_int
    return _lib


def roi_align(feat, rois, out_size, spatial_scale, sampling_ratio=0, aligned=True):
    """mmcv.ops.RoIAlign(pool_mode='avg') restatement.  feat [N,C,H,W], rois [K,5] -> [K,C,ph,pw]."""
    lib = _load()
    feat = feat.detach().float().contiguous()
    rois = rois.detach().float().contiguous()
    ph, pw = (out_size, out_size) if isinstance(out_size, int) else out_size
    N, C, H, W = feat.shape
    K = rois.shape[0]
    out = torch.zeros((K, C, ph, pw), dtype=torch.float32)
    if K:
        lib.oracle_roi_align(ctypes.c_void_p(feat.data_ptr()), N, C, H, W,
                             ctypes.c_void_p(rois.data_ptr()), K, ph, pw,
                             ctypes.c_float(spatial_scale), int(sampling_ratio), int(bool(aligned)),
                             ctypes.c_void_p(out.data_ptr()))
    return out


def nms(boxes, scores, iou_threshold):
    """mmcv.ops.nms(offset=0) restatement: returns (dets [k,5], keep idx [k]) in score order.
    Tie order: stable sort (lower index first) -- the canonical order of SURVEY.md App. D.10."""
    lib = _load()
    boxes = boxes.detach().float().contiguous()
    scores = scores.detach().float().contiguous()
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0, 5)), torch.zeros((0,), dtype=torch.long)
    order = torch.sort(scores, descending=True, stable=True)[1].contiguous()
    keep = torch.empty(n, dtype=torch.int64)
    k = lib.oracle_nms(ctypes.c_void_p(boxes.data_ptr()), ctypes.c_void_p(order.data_ptr()), n,
                       ctypes.c_float(iou_threshold), ctypes.c_void_p(keep.data_ptr()))
    keep = keep[:k]
    return torch.cat([boxes[keep], scores[keep, None]], 1), keep
