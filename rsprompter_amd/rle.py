"""Evaluation hand-off: `encode_mask_results` of the reference
(mmdet/structures/mask/utils.py:38-53 -> pycocotools.mask.encode): run-length counting (`rsp_mask_rle`) and the
compression of the counts to COCO's ASCII string (cocoapi maskApi.c rleToString -> `rsp_rle_to_string`) both run on the
GPU; the host receives finished strings.  The result is what CocoMetric.process stores per instance
(coco_metric.py:346-391): dict(size=[h, w], counts=bytes)."""
import torch

from . import ops


def counts_to_string(cnts):
    """rleToString on a Python list (the definition the device kernel is tested against; not on the product path)."""
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


_counts_to_string = counts_to_string      # round-2 name


def encode_rle_strings(masks, cap=4096, flat_cap=None):
    """bool [k, H, W] on the device -> (flat uint8 tensor, offsets int64 [k + 1]) on the HOST: string i is
    flat[offs[i]:offs[i + 1]].  Two device kernels, one host synchronisation for the sizes; grows its capacities and
    retries when a mask has more runs / the strings more bytes than assumed."""
    ops.require_device(masks.device)
    k, h, w = masks.shape
    dev = masks.device
    if k == 0:
        return torch.zeros((0,), dtype=torch.uint8), torch.zeros((1,), dtype=torch.int64)
    flat_cap = flat_cap or 2 * cap * k
    while True:
        counts = torch.empty((k, cap), dtype=torch.int32, device=dev)
        ws = torch.empty((k, cap), dtype=torch.int32, device=dev)
        n = torch.empty((k,), dtype=torch.int32, device=dev)
        ops.mask_rle_into(masks, counts, ws, n)
        lens, offs, flat = ops.rle_to_string(counts, n, k, flat_cap)
        need_runs = int((-n).max().item())
        offs_h = offs.cpu()
        if need_runs > 0:
            cap = 1 << (need_runs - 1).bit_length()
            flat_cap = max(flat_cap, 2 * cap * k)
            continue
        if int(offs_h[-1]) > flat_cap:
            flat_cap = int(offs_h[-1])
            continue
        return flat[:int(offs_h[-1])].cpu(), offs_h


def encode_mask_results(masks):
    """masks: bool tensor [k, H, W] on the HIP device -> list of k RLE dicts (same as the reference's function)."""
    k, h, w = masks.shape
    flat, offs = encode_rle_strings(masks)
    buf, o = flat.numpy().tobytes(), offs.tolist()
    return [dict(size=[int(h), int(w)], counts=buf[o[i]:o[i + 1]]) for i in range(k)]
