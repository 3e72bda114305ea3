"""Multi-GPU data parallelism for the inference path: one process per GPU, images sharded by
batch, ONE collective -- the all-gather of per-image instance results.

Reference equivalent: tools/dist_test.sh (1 proc / GPU) + DefaultSampler round-robin sharding
(configs/rsprompter/_base_/rsprompter_anchor.py:269) + CocoMetric.process (per-rank RLE,
mmdet/evaluation/metrics/coco_metric.py:346-391) + mmengine `collect_results`.
Two exchanges are provided:
  * `gather_results` (the default hand-off): what the reference's ranks exchange -- per-image records plus the COCO RLE
    strings of the masks, both produced on the device (`rsp_mask_rle`, `rsp_rle_to_string`: a few hundred bytes per
    instance instead of 128 KiB of packed bits), images of DIFFERENT sizes in one batch (rescale=True puts masks at each
    image's own `ori_shape`), queued completely on a side stream so that the next step's kernels overlap it, gathered to
    rank 0 like mmengine `collect_results`, and handed back as a lazy list in DATASET order with the sampler's
    wrap-around padding dropped.
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
# Reference: every rank RLE-encodes its own predictions (CocoMetric.process, coco_metric.py:346-391, through
# encode_mask_results, structures/mask/utils.py:38-53) and mmengine `collect_results` brings the per-rank lists to RANK 0
# only (tools/dist_test.sh:11-22 launches one process per GPU).  Here:
#   * run-length counting AND the COCO string compression run on the device (rsp_mask_rle, rsp_rle_to_string): what
#     travels is finished strings, a few hundred bytes per instance;
#   * everything a step contributes is queued at `gather_results` time on the caller's side stream -- codec kernels, a
#     32-byte header all-gather, four fixed-capacity gathers to the destination rank, the copy into pinned host memory --
#     with no host synchronisation (host-made pieces travel through pinned staging buffers), so it runs behind the next
#     step's kernels;
#   * `collect()` waits for that stream's event and hands back a LAZY sequence: the per-image dicts (and the Python bytes
#     of an instance's string) are built when the consumer indexes them, never in a per-step loop;
#   * capacities (images, instances, string bytes per rank and step, runs per mask) are agreed from the all-gathered
#     headers: a step that does not fit is re-sent with doubled capacities by every rank (they all see the same headers).
class ExchangeState:
    """Capacities of the fixed-shape exchange + the pinned host buffers of the destination rank (one per process group)."""

    def __init__(self):
        self.img_cap = self.inst_cap = self.byte_cap = 0
        self.run_cap = 4096
        self.host = {}
        self.live = None             # weakref to the GatheredResults that still aliases `host` (see GatheredResults._detach)
        self.in_flight = False       # an exchange queued on a side stream whose collect() has not run: it owns `host`
        # set by a CANCELLED exchange whose headers exceeded the agreed capacities (world > 1): the ranks that collected it
        # queued a re-send and grew their capacities, this rank did neither -- the next exchange must not issue collectives
        self.poisoned = None

    def fits(self, need):
        return need[0] <= self.img_cap and need[1] <= self.inst_cap and need[2] <= self.byte_cap and need[3] <= 0

    def grow(self, need):
        p2 = lambda v, lo: max(lo, 1 << (max(int(v), 1) * 5 // 4).bit_length())
        self.img_cap = max(self.img_cap, p2(need[0], 1))
        self.inst_cap = max(self.inst_cap, p2(need[1], 16))
        self.byte_cap = max(self.byte_cap, p2(need[2], 1 << 16))
        if need[3] > 0:
            self.run_cap = max(self.run_cap, 1 << (int(need[3]) - 1).bit_length())


_STATES = {}


def release_state(group=None):
    """Forget the ExchangeState (capacities, pinned buffers) of a process group -- call it before destroying the group;
    `_STATES` otherwise keeps the group object and its buffers alive for the life of the process."""
    _STATES.pop(id(group) if group is not None else 0, None)


def _state_of(group):
    """one ExchangeState per process group.  The entry keeps a reference to the group object, so its id() cannot be handed
    to another group while the state exists."""
    key = id(group) if group is not None else 0
    if key not in _STATES:
        _STATES[key] = (group, ExchangeState())
    return _STATES[key][1]


class DeviceCodec:
    """results -> (records, COCO RLE strings) with the two HIP kernels; nothing in here touches the host."""

    def encode(self, results_list, run_cap, byte_cap, dev):
        from . import ops
        ks = [int(r.bboxes.shape[0]) for r in results_list]
        K = sum(ks)
        counts = torch.empty((max(K, 1), run_cap), dtype=torch.int32, device=dev)
        ws = torch.empty((max(K, 1), run_cap), dtype=torch.int32, device=dev)
        n = torch.ones((max(K, 1),), dtype=torch.int32, device=dev)
        i0 = 0
        for r, k in zip(results_list, ks):
            if k:
                ops.mask_rle_into(r.masks, counts[i0:i0 + k], ws[i0:i0 + k], n[i0:i0 + k])
            i0 += k
        lens, offs, flat = ops.rle_to_string(counts, n, K, byte_cap)
        runs_needed = (-n[:K]).clamp_min(0).max() if K else torch.zeros((), dtype=torch.int32, device=dev)
        return lens[:K], flat, offs[K], runs_needed


class PendingGather:
    """Handle of an exchange in flight (device work queued on `stream`); `collect()` finishes it on the host.

    A handle that is dropped without `collect()` -- an exception in the step loop between the two calls -- must not leave
    its process group blocked: `cancel()` (also run by `__del__`) waits for the side stream's queued work, so that the
    pinned buffers are no longer written to, and releases the group's state for the next exchange.  Nothing is
    unpacked; if this step's headers did not fit the agreed capacities the re-send that `collect()` would have queued
    on every rank is skipped on this one -- cancel on all ranks or on none.  A cancel that finds such headers (world > 1)
    marks the group's ExchangeState poisoned: the next gather_results raises instead of issuing mismatched collectives."""

    def __init__(self, fn, abandon=None):
        self._fn, self._out, self._abandon = fn, None, abandon

    def collect(self):
        if self._fn is not None:
            self._out, self._fn, self._abandon = self._fn(), None, None
        return self._out

    def cancel(self):
        if self._fn is not None and self._abandon is not None:
            try:
                self._abandon()
            finally:
                self._fn = self._abandon = None

    def __del__(self):
        try:
            self.cancel()
        except Exception:                  # interpreter shutdown: the CUDA context may already be gone
            pass


class GatheredResults:
    """What the destination rank holds after an exchange: the gathered flat buffers (host) + index tables.  Behaves like
    the list `collect_results` returns -- dataset order, the sampler's wrap-around duplicates dropped -- but builds
    item j (dict(bboxes, scores, labels, masks=[dict(size=[h, w], counts=bytes)])) only when it is asked for."""

    def __init__(self, headers, meta, rec, lens, flat, dataset_size=None):
        import numpy as np
        # per-rank views; on the GPU path they alias the exchange's PINNED buffers until the next exchange needs those
        # (`_detach`, called by gather_results) -- a consumer that drops the results at once (bench.py) never pays a copy,
        # one that keeps them gets private memory holding the used parts only
        world = headers.shape[0]
        self._meta, self._rec = [meta[w] for w in range(world)], [rec[w] for w in range(world)]
        self._lens, self._flat = [lens[w] for w in range(world)], [flat[w] for w in range(world)]
        self._used = [(int(headers[w, 0]), int(headers[w, 1]), int(headers[w, 2])) for w in range(world)]
        n_img = [int(headers[w, 0]) for w in range(world)]
        self._inst0 = [np.concatenate([[0], np.cumsum(meta[w, :n_img[w], 0].astype(np.int64))]) for w in range(world)]
        self._byte0 = [np.concatenate([[0], np.cumsum(lens[w, :int(headers[w, 1])].astype(np.int64))]) for w in range(world)]
        # mmengine collect_results: interleave the per-rank lists (zip), then cut to the dataset size
        order = [(w, i) for i in range(max(n_img + [0])) for w in range(world) if i < n_img[w]]
        self._order = order if dataset_size is None else order[:dataset_size]
        self.n_instances = int(sum(self._inst0[w][i + 1] - self._inst0[w][i] for w, i in self._order))
        self.n_bytes = int(sum(int(headers[w, 2]) for w in range(world)))

    def _detach(self):
        """own the data: copy the used part of every rank's buffers out of the shared pinned memory"""
        for w, (ni, nk, nb) in enumerate(self._used):
            self._meta[w], self._rec[w] = self._meta[w][:ni].copy(), self._rec[w][:nk].copy()
            self._lens[w], self._flat[w] = self._lens[w][:nk].copy(), self._flat[w][:nb].copy()

    def __len__(self):
        return len(self._order)

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self[i] for i in range(*j.indices(len(self)))]
        w, i = self._order[j]
        k, h, wd = (int(v) for v in self._meta[w][i])
        i0 = int(self._inst0[w][i])
        rr = torch.from_numpy(self._rec[w][i0:i0 + k].copy())
        b = self._byte0[w]
        buf = self._flat[w]
        masks = [dict(size=[h, wd], counts=buf[int(b[i0 + t]):int(b[i0 + t + 1])].tobytes()) for t in range(k)]
        return dict(bboxes=rr[:, :4].clone(), scores=rr[:, 4].clone(), labels=rr[:, 5].long(), masks=masks)

    def __iter__(self):
        return (self[j] for j in range(len(self)))


def _pad_rows(t, n):
    if t.shape[0] == n:
        return t.contiguous()
    out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    m = min(n, t.shape[0])
    out[:m] = t[:m]
    return out


def gather_results(results_list, dataset_size=None, group=None, stream=None, dst=0, codec=None, state=None, device=None,
                   codec_on_side_stream=True):
    """Bring every rank's per-image results (records + COCO RLE strings) to rank `dst` (None: to every rank).

    results_list: this rank's InstanceData list (bboxes, scores, labels, masks bool [k, H_i, W_i]); mask sizes may
    differ from image to image.  Images are assumed sharded round-robin (`shard_indices`): the i-th image of rank r is
    dataset item i * world + r.  Returns (from `collect()` when a side `stream` is given, directly otherwise) a
    `GatheredResults` on the destination rank(s) and None elsewhere, as mmengine `collect_results` does.
        h = gather_results(out, stream=side, ...)   ->  PendingGather;  ...next test_step...;  res = h.collect()
    codec: the (records, strings) encoder; the default runs the HIP kernels (tests on CPU inject a numpy one).
    device: where the exchange buffers live when this rank has NO image this step (default: the current CUDA device under
    an nccl group, else the CPU) -- the collectives of all ranks must run on the same kind of tensor.
    One exchange per process group may be in flight: queue the next one after `collect()` of the previous."""
    import numpy as np
    codec = codec or DeviceCodec()
    state = state or _state_of(group)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_img = len(results_list)
    if state.live is not None:                         # the previous results are still in use: give them their own memory
        prev = state.live()
        if prev is not None:
            prev._detach()
        state.live = None
    if state.poisoned:
        raise RuntimeError('gather_results: ' + state.poisoned)
    if state.in_flight:
        raise RuntimeError('gather_results: the previous exchange of this process group has not been collected '
                           '(its pinned host buffers would be overwritten); call collect() first')
    if n_img:
        dev = results_list[0].bboxes.device
    elif device is not None:
        dev = torch.device(device)
    elif dist.is_initialized() and torch.cuda.is_available() and 'nccl' in str(dist.get_backend(group)):
        dev = torch.device('cuda', torch.cuda.current_device())
    else:
        dev = torch.device('cpu')
    on_gpu = dev.type == 'cuda'
    use_stream = stream is not None and on_gpu
    to_me = dst is None or rank == dst
    ks = [int(r.bboxes.shape[0]) for r in results_list]
    K = sum(ks)
    if state.img_cap == 0:
        # first exchange of this process group: one synchronous MAX over the ranks so that everybody starts from the
        # SAME capacities (afterwards they only change through the all-gathered headers)
        need0 = torch.tensor([n_img, K, 512 * K, 0], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(need0, op=dist.ReduceOp.MAX, group=group)
        state.grow(need0.tolist())
    meta_h = torch.tensor([[k, r.masks.shape[-2], r.masks.shape[-1]] for k, r in zip(ks, results_list)],
                          dtype=torch.int32).reshape(n_img, 3)
    # host-made pieces travel through PINNED memory: a `torch.tensor(..., device=dev)` is a synchronous copy on the side
    # stream, i.e. it would hold the host until the whole step before it has run -- the interpreter could never get ahead
    # of the GPU again (measured round 4: 8.8 ms of idle compute stream per step with the exchange on)
    nk_h = torch.tensor([n_img, K], dtype=torch.int64)
    if on_gpu:
        # (staging buffers of the state, allocated once: the previous exchange's copies out of them completed before its
        # collect() returned)
        st = state.host.get('stage')
        if st is None or st[0].shape[0] < max(n_img, 1):
            st = state.host['stage'] = (torch.empty((max(n_img, 1) * 2, 3), dtype=torch.int32).pin_memory(),
                                        torch.empty((2,), dtype=torch.int64).pin_memory())
        st[0][:n_img].copy_(meta_h)
        st[1].copy_(nk_h)
        meta_h, nk_h = st[0][:n_img], st[1]

    def queue():
        """all device work + collectives of one attempt; returns the tensors collect() reads and an event.
        codec_on_side_stream=False runs the codec (run lengths + strings of this step's masks) on the compute stream, in
        order behind the step that made the masks, and only the collectives and the copy into pinned host memory on the
        side stream.  Measured round 4 (tests/test_gpu_dist.py, ViT-H x 8 tiles, 800 NOISE masks of the synthetic weights:
        127 k runs and 135 KB of string each, 108 MB per step -- trained masks are a few hundred runs): 169.6 ms per step
        without the exchange, 173.4 with the codec on the side stream, 174.8 on the compute stream."""
        side_codec = use_stream and codec_on_side_stream
        if side_codec:
            stream.wait_stream(torch.cuda.current_stream(dev))
            for r in results_list:                          # produced on the compute stream, read on `stream`
                for t in (r.bboxes, r.scores, r.labels, r.masks):
                    t.record_stream(stream)
        with (torch.cuda.stream(stream) if side_codec else _null()):
            lens, flat, total, runs_needed = codec.encode(results_list, state.run_cap, state.byte_cap, dev)
            rec_parts = [torch.cat([r.bboxes.float(), r.scores.float()[:, None], r.labels.float()[:, None]], 1)
                         for r, k in zip(results_list, ks) if k]
            rec = torch.cat(rec_parts, 0) if rec_parts else torch.zeros((0, 6), dtype=torch.float32, device=dev)
            header = torch.cat([nk_h.to(dev, non_blocking=True), total.to(torch.int64).reshape(1),
                                runs_needed.to(torch.int64).reshape(1)])
            payload = [_pad_rows(meta_h.to(dev, non_blocking=True), state.img_cap), _pad_rows(rec, state.inst_cap),
                       _pad_rows(lens.to(torch.int32), state.inst_cap), _pad_rows(flat, state.byte_cap)]
        if use_stream and not side_codec:
            stream.wait_stream(torch.cuda.current_stream(dev))
            for t in [header] + payload:                    # produced on the compute stream, read on `stream`
                t.record_stream(stream)
        with (torch.cuda.stream(stream) if use_stream else _null()):
            if world > 1:
                headers = torch.empty((world, 4), dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(headers.view(-1), header, group=group)
                got = []
                for t in payload:
                    if dst is None:
                        g = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=dev)
                        dist.all_gather_into_tensor(g.view(-1), t.view(-1), group=group)
                        got.append(g)
                    else:
                        parts = [torch.empty_like(t) for _ in range(world)] if to_me else None
                        dist.gather(t, parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
                        got.append(torch.stack(parts, 0) if to_me else None)
            else:
                headers, got = header[None], [t[None] for t in payload]
            ev = None
            if on_gpu:                                      # device -> pinned host, still on the side stream
                def pinned(name, t):
                    h = state.host.get(name)
                    if h is None or h.shape != t.shape or h.dtype != t.dtype:
                        h = state.host[name] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    return h
                headers = pinned('headers', headers)
                if to_me:
                    got = [pinned(f'p{i}', t) for i, t in enumerate(got)]
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
        return headers, got, ev

    inflight = queue()
    state.in_flight = bool(use_stream)

    def finish():
        nonlocal inflight
        try:
            return _finish()
        finally:
            state.in_flight = False

    def _finish():
        nonlocal inflight
        while True:
            headers, got, ev = inflight
            if ev is not None:
                ev.synchronize()                            # the side stream's work only: the compute stream keeps running
            hd = headers.numpy().copy()
            need = [int(v) for v in hd.max(0)]
            if state.fits(need):
                break
            state.grow(need)                                # every rank sees the same headers -> the same decision
            inflight = queue()
        if not to_me:
            return None
        meta, rec, lens, flat = (t.numpy() for t in got)
        out = GatheredResults(hd, meta, rec, lens, flat, dataset_size)
        if on_gpu:                                          # views of the pinned buffers the next exchange reuses: the
            import weakref                                  # results detach themselves then, if anybody still holds them
            state.live = weakref.ref(out)
        return out

    def abandon():
        nonlocal inflight
        try:
            headers, _, ev = inflight
            if ev is not None:
                ev.synchronize()
            # the headers are read even though nothing is unpacked: every rank that collects this step compares them with
            # the capacities and, if they do not fit, queues a second round of collectives and grows -- a rank that only
            # cancelled would issue mismatched collectives at its next exchange (hang or corruption)
            if world > 1:
                need = [int(v) for v in headers.cpu().numpy().max(0)]
                if not state.fits(need):
                    state.poisoned = ('a cancelled exchange of this process group needed capacities ' + str(need) +
                                      ' above the agreed ones: the other ranks re-sent and grew, this rank did not -- '
                                      'destroy the group / call release_state(group) on every rank')
        finally:
            state.in_flight = False

    return PendingGather(finish, abandon) if use_stream else finish()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
