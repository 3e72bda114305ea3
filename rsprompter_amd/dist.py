"""Multi-GPU data parallelism for the inference path: one process per GPU, images sharded by
batch, ONE collective -- the all-gather of per-image instance results.

Reference equivalent: tools/dist_test.sh (1 proc / GPU) + DefaultSampler round-robin sharding
(configs/rsprompter/_base_/rsprompter_anchor.py:269) + CocoMetric.process (per-rank RLE,
mmdet/evaluation/metrics/coco_metric.py:346-391) + mmengine `collect_results`.
Here the payload is bit-packed masks + boxes/scores/labels, exchanged in two RCCL all-gathers
(counts, then records padded to the global max count) over xGMI.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # "nccl" IS RCCL on ROCm
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_indices(n_items, rank, world):
    """mmengine DefaultSampler(shuffle=False) semantics: round-robin by rank, padded by wrap-around."""
    total = (n_items + world - 1) // world * world
    idx = list(range(n_items)) + list(range(total - n_items))
    return idx[rank:total:world]


def pack_results(results_list, pack_fn, max_k):
    """per-image InstanceData (bboxes, scores, labels, masks) -> fixed-shape device tensors."""
    n = len(results_list)
    dev = results_list[0].bboxes.device if n else torch.device('cpu')
    hw = results_list[0].masks.shape[-2:] if n else (0, 0)
    nb = hw[0] * hw[1] // 8
    counts = torch.tensor([len(r.bboxes) for r in results_list], dtype=torch.int32, device=dev)
    rec = torch.zeros((n, max_k, 6), dtype=torch.float32, device=dev)      # x1 y1 x2 y2 score label
    masks = torch.zeros((n, max_k, nb), dtype=torch.uint8, device=dev)
    for i, r in enumerate(results_list):
        k = min(len(r.bboxes), max_k)
        if k == 0:
            continue
        rec[i, :k, :4] = r.bboxes[:k]
        rec[i, :k, 4] = r.scores[:k]
        rec[i, :k, 5] = r.labels[:k].to(torch.float32)
        masks[i, :k] = pack_fn(r.masks[:k])
    return counts, rec, masks


def all_gather_results(results_list, pack_fn=None, group=None):
    """Gather every rank's per-image results on every rank.  Returns dict(counts [W*n], records
    [W*n, K, 6], masks [W*n, K, HW/8], mask_hw) ordered rank-major."""
    if pack_fn is None:
        from . import ops
        pack_fn = ops.pack_masks
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    local_max = max([len(r.bboxes) for r in results_list] + [0])
    dev = results_list[0].bboxes.device
    if world > 1:
        mx = torch.tensor([local_max], dtype=torch.int32, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)      # 4-byte exchange to size the payload
        max_k = int(mx.item())
    else:
        max_k = local_max
    counts, rec, masks = pack_results(results_list, pack_fn, max(max_k, 1))
    hw = tuple(results_list[0].masks.shape[-2:])
    if world == 1:
        return dict(counts=counts, records=rec, masks=masks, mask_hw=hw)
    n = counts.shape[0]
    g_counts = torch.empty((world * n,), dtype=counts.dtype, device=dev)
    g_rec = torch.empty((world * n,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=dev)
    g_masks = torch.empty((world * n,) + tuple(masks.shape[1:]), dtype=masks.dtype, device=dev)
    dist.all_gather_into_tensor(g_counts, counts, group=group)
    dist.all_gather_into_tensor(g_rec, rec, group=group)
    dist.all_gather_into_tensor(g_masks, masks, group=group)
    return dict(counts=g_counts, records=g_rec, masks=g_masks, mask_hw=hw)
