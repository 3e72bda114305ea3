"""Multi-GPU data parallelism for the inference path: one process per GPU, images sharded by
batch, ONE collective -- the all-gather of per-image instance results.

Reference equivalent: tools/dist_test.sh (1 proc / GPU) + DefaultSampler round-robin sharding
(configs/rsprompter/_base_/rsprompter_anchor.py:269) + CocoMetric.process (per-rank RLE,
mmdet/evaluation/metrics/coco_metric.py:346-391) + mmengine `collect_results`.
Two exchanges are provided:
  * `gather_results` (the default hand-off): what the reference's ranks exchange -- per-image records plus COCO RLE
    run lengths (`rsp_mask_rle`, KBs per instance instead of 128 KiB of packed bits), images of DIFFERENT sizes in one
    batch (rescale=True puts masks at each image's own `ori_shape`), issued on a side stream so that the next step's
    kernels overlap it, and returned in DATASET order with the sampler's wrap-around padding dropped -- exactly what
    mmengine `collect_results` does with the per-rank lists.
  * `all_gather_results`: dense bit-packed masks for consumers that want pixels on every rank (all images one size).
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
    """per-image InstanceData (bboxes, scores, labels, masks) -> fixed-shape device tensors (dense exchange: every
    image must have the same mask size, a multiple of 8 pixels; use `gather_results` otherwise)."""
    n = len(results_list)
    dev = results_list[0].bboxes.device if n else torch.device('cpu')
    hw = results_list[0].masks.shape[-2:] if n else (0, 0)
    for r in results_list:
        if tuple(r.masks.shape[-2:]) != tuple(hw):
            raise ValueError('all_gather_results needs one mask size per batch (got %s and %s): images with their own '
                             'ori_shape go through gather_results (RLE)' % (tuple(hw), tuple(r.masks.shape[-2:])))
    if (hw[0] * hw[1]) % 8:
        raise ValueError('bit-packed exchange needs H*W % 8 == 0; use gather_results (RLE)')
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


# ----------------------------------------------------------------------------- RLE exchange (default hand-off)
def _rle_device(masks):
    from . import ops
    return ops.mask_rle_counts(masks)


def _pad_to(t, n):
    if t.shape[0] == n:
        return t
    out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


class PendingGather:
    """Handle of an exchange in flight (device work queued on `stream`); `collect()` finishes it on the host."""

    def __init__(self, fn):
        self._fn, self._out = fn, None

    def collect(self):
        if self._fn is not None:
            self._out, self._fn = self._fn(), None
        return self._out


def gather_results(results_list, dataset_size=None, rle_fn=None, group=None, stream=None, compress=True):
    """Gather every rank's per-image results (records + COCO RLE) on every rank.

    results_list: this rank's InstanceData list (bboxes, scores, labels, masks bool [k, H_i, W_i]); mask sizes may
    differ from image to image.  Images are assumed sharded round-robin (`shard_indices`): the i-th image of rank r
    is dataset item i * world + r.  Returns a list of per-image dicts in dataset order, truncated to `dataset_size`
    (the sampler's wrap-around duplicates are dropped): dict(bboxes f32 [k,4], scores f32 [k], labels i64 [k],
    masks=[dict(size=[h, w], counts=bytes)]) -- the `pred` CocoMetric.process builds (coco_metric.py:346-391).
    stream: a side torch.cuda.Stream; the RLE kernel, the packing and the collectives are queued there (after the
    work already on the current stream), so the caller can launch the next step before `collect()`-ing:
        h = gather_results(out, stream=side, ...)   ->  PendingGather;  ...next test_step...;  res = h.collect()
    """
    rle_fn = rle_fn or _rle_device
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_img = len(results_list)
    dev = results_list[0].bboxes.device if n_img else torch.device('cpu')
    use_stream = stream is not None and dev.type == 'cuda'
    if use_stream:
        stream.wait_stream(torch.cuda.current_stream(dev))
    ctx = torch.cuda.stream(stream) if use_stream else _null()
    with ctx:
        ks = [int(r.bboxes.shape[0]) for r in results_list]
        meta = torch.tensor([[k, r.masks.shape[-2], r.masks.shape[-1]] for k, r in zip(ks, results_list)],
                            dtype=torch.int32).reshape(n_img, 3)
        rec_parts, cnt_parts, len_parts = [], [], []
        for r, k in zip(results_list, ks):
            if k == 0:
                continue
            rec_parts.append(torch.cat([r.bboxes.float(), r.scores.float()[:, None], r.labels.float()[:, None]], 1))
            counts, n = rle_fn(r.masks)                       # [k, cap] int32, [k] int32 (device)
            cnt_parts.append((counts, n))
            len_parts.append(n)
        rec = torch.cat(rec_parts, 0) if rec_parts else torch.zeros((0, 6), dtype=torch.float32, device=dev)
        run_len = torch.cat(len_parts, 0).to(torch.int32) if len_parts else torch.zeros((0,), dtype=torch.int32, device=dev)

    def finish():
        if use_stream:
            stream.synchronize()                              # only the side stream: the compute stream keeps running
        ctx2 = torch.cuda.stream(stream) if use_stream else _null()
        with ctx2:
            # ragged run-length lists -> one flat int32 buffer (instance-major): one masked select per image
            flat = [c[torch.arange(c.shape[1], device=c.device)[None, :] < nn_[:, None].to(torch.int64)]
                    for c, nn_ in cnt_parts]
            runs = torch.cat(flat, 0).to(torch.int32) if flat else torch.zeros((0,), dtype=torch.int32, device=dev)
            sizes = torch.tensor([n_img, rec.shape[0], runs.shape[0]], dtype=torch.int64, device=dev)
            if world > 1:
                all_sizes = torch.empty((world, 3), dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(all_sizes.view(-1), sizes, group=group)
                all_sizes = all_sizes.cpu()
                mx = all_sizes.max(0).values.tolist()
                bufs = []
                for t, n in ((meta.to(dev), mx[0]), (rec, mx[1]), (run_len, mx[1]), (runs, mx[2])):
                    t = _pad_to(t.contiguous(), max(int(n), 1))
                    g = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=dev)
                    dist.all_gather_into_tensor(g.view(-1), t.view(-1), group=group)
                    bufs.append(g.cpu())
                g_meta, g_rec, g_len, g_runs = bufs
            else:
                all_sizes = sizes.cpu()[None]
                g_meta, g_rec, g_len, g_runs = meta[None], rec.cpu()[None], run_len.cpu()[None], runs.cpu()[None]
        from .rle import _counts_to_string
        per_rank = []
        for w in range(world):
            ni, _, _ = [int(v) for v in all_sizes[w]]
            imgs, i0, r0 = [], 0, 0
            for i in range(ni):
                k, h, wd = [int(v) for v in g_meta[w, i]]
                ls = g_len[w, i0:i0 + k].tolist()
                rles = []
                for ln in ls:
                    cnts = g_runs[w, r0:r0 + ln].tolist()
                    rles.append(dict(size=[h, wd], counts=_counts_to_string(cnts) if compress else cnts))
                    r0 += ln
                rr = g_rec[w, i0:i0 + k]
                imgs.append(dict(bboxes=rr[:, :4].clone(), scores=rr[:, 4].clone(), labels=rr[:, 5].long(), masks=rles))
                i0 += k
            per_rank.append(imgs)
        # mmengine collect_results: interleave the per-rank lists (zip), then cut to the dataset size
        ordered = []
        for i in range(max(len(p) for p in per_rank)):
            for w in range(world):
                if i < len(per_rank[w]):
                    ordered.append(per_rank[w][i])
        if dataset_size is not None:
            ordered = ordered[:dataset_size]
        return ordered

    return PendingGather(finish) if use_stream else finish()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
