"""Evaluation hand-off: `encode_mask_results` of the reference
(mmdet/structures/mask/utils.py:38-53 -> pycocotools.mask.encode) with the run-length counting done on the GPU
(`rsp_mask_rle`); only the compression of the counts to COCO's ASCII string (cocoapi maskApi.c rleToString, a few
hundred integers per instance) runs on the host.  The result is what CocoMetric.process stores per instance
(coco_metric.py:346-391): dict(size=[h, w], counts=bytes)."""
from . import ops


def _counts_to_string(cnts):
    out = bytearray()
    for i, x in enumerate(cnts):
        if i > 2:
            x -= cnts[i - 2]
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return bytes(out)


def encode_mask_results(masks):
    """masks: bool tensor [k, H, W] on the HIP device -> list of k RLE dicts (same as the reference's function)."""
    ops.require_device(masks.device)
    k, h, w = masks.shape
    counts, n = ops.mask_rle_counts(masks)
    counts, n = counts.cpu().tolist(), n.cpu().tolist()
    return [dict(size=[int(h), int(w)], counts=_counts_to_string(counts[i][:n[i]])) for i in range(k)]
