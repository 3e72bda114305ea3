"""-m gpu: the device side of the result exchange: rsp_pack_bits against numpy.packbits, and
all_gather_results on device tensors -- single process, and two processes sharing cuda:0 over gloo (RCCL refuses two
ranks on one device; the collectives are the same calls bench.py issues over RCCL on an 8-GPU node)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _make_results(rank, n_imgs, dev, hw=(64, 96)):
    from rsprompter_amd.structures import InstanceData
    out = []
    for i in range(n_imgs):
        g = torch.Generator().manual_seed(100 * rank + i)
        k = [3, 0, 5, 2][(2 * rank + i) % 4]
        out.append(InstanceData(bboxes=(torch.rand(k, 4, generator=g) * 50).to(dev), scores=torch.rand(k, generator=g).to(dev),
                                labels=torch.randint(0, 10, (k,), generator=g).to(dev),
                                masks=(torch.rand(k, *hw, generator=g) > 0.5).to(dev)))
    return out


def test_pack_masks_matches_numpy(dev):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(3)
    m = torch.rand(7, 1024, 1024, generator=g) > 0.3
    got = ops.pack_masks(m.to(dev)).cpu().numpy()
    ref = np.packbits(m.reshape(7, -1).numpy().astype(np.uint8), axis=1, bitorder='little')
    assert np.array_equal(got, ref)


def test_all_gather_results_single_process_on_device(dev):
    from rsprompter_amd import dist as rdist
    res = _make_results(0, 2, dev)
    g = rdist.all_gather_results(res)
    assert g['counts'].tolist() == [3, 0] and g['records'].shape[1] == 3
    bits = np.unpackbits(g['masks'][0, :3].cpu().numpy(), axis=1, bitorder='little').astype(bool)
    assert np.array_equal(bits.reshape(3, *g['mask_hw']), res[0].masks.cpu().numpy())


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rsprompter_amd import dist as rdist
    dev = torch.device('cuda:0')
    try:
        rdist.init_from_env(backend='gloo')
        g = rdist.all_gather_results(_make_results(rank, 2, dev))
        ret[rank] = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in g.items()}
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # reported to the parent, which decides between failure and "backend cannot do this"
        ret[rank] = repr(e)


def test_all_gather_results_two_ranks_on_one_device():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    if isinstance(ret[0], str) or isinstance(ret[1], str):
        msg = f'{ret[0]} / {ret[1]}'
        if 'gloo' in msg.lower() or 'not supported' in msg.lower() or 'not implemented' in msg.lower():
            pytest.skip(f'gloo cannot run this collective on device tensors here: {msg[:200]}')
        raise AssertionError(msg)
    g0, g1 = ret[0], ret[1]
    for k in ('counts', 'records', 'masks'):
        assert torch.equal(g0[k], g1[k])
    assert g0['counts'].tolist() == [3, 0, 5, 2] and g0['records'].shape[1] == 5
    cpu = torch.device('cpu')
    for rank in range(2):
        for i, r in enumerate(_make_results(rank, 2, cpu)):
            j, k = rank * 2 + i, len(r.bboxes)
            assert torch.equal(g0['records'][j, :k, :4], r.bboxes)
            if k:
                bits = np.unpackbits(g0['masks'][j, :k].numpy(), axis=1, bitorder='little').astype(bool)
                assert np.array_equal(bits.reshape(k, *g0['mask_hw']), r.masks.numpy())


def test_mask_rle_matches_coco_restatement(dev):
    """rsp_mask_rle + host string compression against oracle/rle.py (pinned on the reference's RLE strings)."""
    import json
    from oracle import rle
    from rsprompter_amd.rle import encode_mask_results
    g = torch.Generator().manual_seed(8)
    masks = torch.zeros(6, 720, 1280, dtype=torch.bool)
    masks[0, 100:150, 100:150] = True
    masks[1] = torch.rand(720, 1280, generator=g) > 0.5                  # ~460k runs: exercises the capacity retry
    masks[2, :, 0] = True
    masks[3, 0, :] = True
    masks[4] = True
    blob = torch.rand(720, 1280, generator=g)
    masks[5] = torch.nn.functional.avg_pool2d(blob[None, None], 31, 1, 15)[0, 0] > 0.5
    got = encode_mask_results(masks.to(dev))
    for i in range(masks.shape[0]):
        ref = rle.encode(masks[i].numpy())
        assert got[i]['size'] == ref['size'] and got[i]['counts'] == ref['counts'], i
    d = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'coco_rle_strings.json')))
    it = d['items'][0]
    assert got[0]['counts'].decode() == it['counts']                      # the reference's own string for this box


def _blob_results(rank, n_imgs, dev):
    """per-image results with their own mask sizes and blob-like masks (tens of runs each)"""
    from rsprompter_amd.structures import InstanceData
    out = []
    for i in range(n_imgs):
        g = torch.Generator().manual_seed(900 + 10 * rank + i)
        k = [3, 0, 5, 2][(2 * rank + i) % 4]
        hw = [(96, 128), (70, 50), (128, 64), (33, 77)][(rank + i) % 4]
        noise = torch.rand(k, 1, *hw, generator=g)
        masks = torch.nn.functional.avg_pool2d(noise, 9, 1, 4)[:, 0] > 0.5 if k else torch.zeros(0, *hw, dtype=torch.bool)
        out.append(InstanceData(bboxes=(torch.rand(k, 4, generator=g) * 50).to(dev), scores=torch.rand(k, generator=g).to(dev),
                                labels=torch.randint(0, 10, (k,), generator=g).to(dev), masks=masks.to(dev)))
    return out


def test_gather_results_device_codec_side_stream(dev):
    """the default hand-off on the device, single process: RLE counts + COCO strings by the HIP kernels on a side
    stream, pinned-host copy, lazy view; strings decode to the masks (oracle/rle.py)."""
    from oracle import rle as orle
    from rsprompter_amd import dist as rdist
    res = _blob_results(0, 4, dev)
    side = torch.cuda.Stream(device=dev)
    state = rdist.ExchangeState()
    for rep in range(2):
        h = rdist.gather_results(res, stream=side, state=state)
        got = h.collect()
        assert isinstance(got, rdist.GatheredResults) and len(got) == 4
        for g, r in zip(got, res):
            assert torch.equal(g['bboxes'], r.bboxes.cpu()) and torch.equal(g['labels'], r.labels.cpu())
            for j, rle in enumerate(g['masks']):
                dec = orle.rle_decode(orle.rle_from_string(rle['counts']), *rle['size'])
                assert np.array_equal(dec, r.masks[j].cpu().numpy())


def test_gather_results_dropped_handle_releases_the_group(dev):
    """an exchange whose handle is dropped without collect() (an exception between the two calls) must not block the next
    one: the handle's cancel() / __del__ waits for the side stream and clears the in-flight mark."""
    from rsprompter_amd import dist as rdist
    res = _blob_results(0, 2, dev)
    side = torch.cuda.Stream(device=dev)
    state = rdist.ExchangeState()
    h = rdist.gather_results(res, stream=side, state=state)
    assert state.in_flight
    with pytest.raises(RuntimeError):                      # still guarded while the handle lives
        rdist.gather_results(res, stream=side, state=state)
    del h                                                  # dropped: __del__ -> cancel()
    assert not state.in_flight
    h2 = rdist.gather_results(res, stream=side, state=state)
    h2.cancel()
    assert not state.in_flight and h2.collect() is None
    got = rdist.gather_results(res, stream=side, state=state).collect()
    assert len(got) == 2 and torch.equal(got[0]['bboxes'], res[0].bboxes.cpu())
    rdist._state_of(None)
    rdist.release_state(None)
    assert 0 not in rdist._STATES


def _worker_rle(rank, world, port, ret, dst=0):
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rsprompter_amd import dist as rdist
    dev = torch.device('cuda:0')
    try:
        rdist.init_from_env(backend='gloo')
        side = torch.cuda.Stream(device=dev)
        got = rdist.gather_results(_blob_results(rank, 2, dev), dataset_size=4, stream=side, dst=dst).collect()
        ret[rank] = None if got is None else list(got)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        ret[rank] = repr(e)


@pytest.mark.parametrize('dst', [0, None])
def test_gather_results_two_ranks_on_one_device(dst):
    """two processes sharing cuda:0 over gloo (the calls bench.py issues over RCCL): results reach rank 0 only (dst = 0:
    mmengine collect_results, `bench.py --exchange gather`) or EVERY rank (dst = None: the north star's all-gather,
    `--exchange allgather`), in dataset order (item j = image j // 2 of rank j % 2)."""
    from oracle import rle as orle
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rle, args=(2, port, ret, dst), nprocs=2, join=True)
    if isinstance(ret[0], str) or isinstance(ret[1], str):
        msg = f'{ret[0]} / {ret[1]}'
        if 'gloo' in msg.lower() or 'not supported' in msg.lower() or 'not implemented' in msg.lower():
            pytest.skip(f'gloo cannot run this collective on device tensors here: {msg[:200]}')
        raise AssertionError(msg)
    assert (ret[1] is None) == (dst == 0) and len(ret[0]) == 4
    cpu = torch.device('cpu')
    for rank in ((0,) if dst == 0 else (0, 1)):
        assert len(ret[rank]) == 4
        for j, g in enumerate(ret[rank]):
            r = _blob_results(j % 2, 2, cpu)[j // 2]
            assert torch.equal(g['bboxes'], r.bboxes) and len(g['masks']) == len(r.bboxes)
            for t, rle in enumerate(g['masks']):
                dec = orle.rle_decode(orle.rle_from_string(rle['counts']), *rle['size'])
                assert np.array_equal(dec, r.masks[t].numpy())


def _step_loop(model, imgs, metas, dev, world, n_steps, group_kw, stats=None):
    """bench.py's step loop: the exchange of step i is queued on a side stream when the step ends and collected after step
    i + 1 has been launched.  Returns (gathered results per step on the destination rank, local results of the last step,
    timing records)."""
    import time
    from rsprompter_amd import dist as rdist
    from rsprompter_amd.structures import DetDataSample
    side = torch.cuda.Stream(device=dev)
    pending, gathered, timing = None, [], []
    local = None
    for it in range(n_steps):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        out = model.test_step(dict(inputs=imgs, data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
        ev1.record()
        local = [o.pred_instances for o in out]
        if pending is not None:
            gathered.append(pending[0].collect())            # step it - 1, AFTER step it has been launched
            timing[-1]['side_end'] = pending[1]
        h = rdist.gather_results(local, dataset_size=world * len(imgs), stream=side, dst=0, **group_kw)
        ev_side = torch.cuda.Event(enable_timing=True)
        ev_side.record(side)
        pending = (h, ev_side)
        timing.append(dict(host_s=time.perf_counter() - t0, start=ev0, end=ev1))
    gathered.append(pending[0].collect())
    if stats is not None and gathered[-1] is not None:
        stats.update(n_instances=gathered[-1].n_instances, n_bytes=gathered[-1].n_bytes)
    timing[-1]['side_end'] = pending[1]
    torch.cuda.synchronize()
    return gathered, local, timing


def _build_vit(dev, arch='base'):
    import warnings
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import rsprompter_anchor
    from rsprompter_amd.synth import synth_state_dict
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(rsprompter_anchor(arch, 10))
    model.load_state_dict(synth_state_dict(model, seed=0), strict=True)
    return model.to(dev)


@pytest.mark.parametrize('arch', ['huge', 'base'])
def test_bench_step_loop_exchange_costs_no_gpu_time(dev, arch):
    """The weak-scaling preconditions of `bench.py --gpus N` that one GPU can show (no 8-GPU node was available to the
    driver), on the headline configuration (rsprompter_anchor ViT-H, 8 tiles per GPU: the configs[3] slice) and on the
    lightest one (ViT-B, configs[1]): the loop WITH the result exchange -- device RLE codec, header / payload collectives,
    pinned-host copy, all queued on a side stream when a step ends and collected after the next step has been launched --
    must take (nearly) the time of the loop WITHOUT it: (b) the exchange of step i finishes on its stream after step i + 1
    has started on the compute stream, (c) the interpreter's work for it (~4 ms, which the compute stream does wait for:
    the two host syncs inside a step keep the interpreter from running ahead across steps) stays small.
    The synthetic weights make this the WORST case for the codec: their masks are noise (127 k runs, 135 KB of COCO
    string each: 108 MB per step and rank; trained masks are a few hundred runs).  Round 4 found and removed 8.8 ms of idle
    compute stream per step (synchronous `torch.tensor(..., device=)` copies on the side stream), a 256 MB host memcpy per
    collect() and a byte-wise RLE kernel; what is left (2.2 % per step on ViT-H) is printed, the assertion is a loose
    regression guard."""
    import time
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images, synth_metas
    model = _build_vit(dev, arch)
    imgs = [im.to(dev) for im in synth_images(8, seed=1234)]
    metas = synth_metas(8)
    _step_loop(model, imgs, metas, dev, 1, 2, {})                                      # warm-up (packing, allocator)
    n = 4
    plain, with_main, with_side = [], [], []
    stats = {}
    for rnd in range(2):                                   # interleaved rounds, best of each: the box's clock drifts by a few %
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model.test_step(dict(inputs=imgs, data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
        torch.cuda.synchronize()
        plain.append(1e3 * (time.perf_counter() - t0) / n)
        t0 = time.perf_counter()
        gathered, local, timing = _step_loop(model, imgs, metas, dev, 1, n, dict(codec_on_side_stream=True), stats)
        with_side.append(1e3 * (time.perf_counter() - t0) / n)
        t0 = time.perf_counter()
        _step_loop(model, imgs, metas, dev, 1, n, dict(codec_on_side_stream=False))
        with_main.append(1e3 * (time.perf_counter() - t0) / n)
    plain_ms, with_ms, with_side_ms = min(plain), min(with_main), min(with_side)
    gaps = [timing[i]['end'].elapsed_time(timing[i + 1]['start']) for i in range(len(timing) - 1)]
    side_after_next_start = [timing[i + 1]['start'].elapsed_time(timing[i]['side_end']) for i in range(len(timing) - 1)]
    print(f'ViT-{arch} x 8 tiles, {stats}: {plain_ms:.1f} ms / step without the exchange, {with_side_ms:.1f} ms with it (codec on the '
          f'compute stream instead of the side stream: {with_ms:.1f}); compute stream idle '
          f'between steps {["%.2f" % v for v in gaps]} ms; exchange of step i ends {["%.2f" % v for v in side_after_next_start]} '
          f'ms after step i + 1 started')
    assert all(len(g) == 8 for g in gathered)
    assert min(side_after_next_start) > 0.0
    # the measured overheads (2.2 % on ViT-H, 5.8 % on ViT-B alone on a box: DESIGN.md section 7) are REPORTED above; the
    # assertion is a regression guard only -- wall-clock ratios inside a whole-suite run move by several per cent, and the
    # defects this test found were 10-40 % effects
    assert with_side_ms < 1.25 * plain_ms + 2.0, (with_side_ms, plain_ms)


def _worker_loop(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rsprompter_amd import dist as rdist
    from rsprompter_amd.synth import synth_images, synth_metas
    dev = torch.device('cuda:0')
    try:
        rdist.init_from_env(backend='gloo')
        model = _build_vit(dev)
        imgs = [im.to(dev) for im in synth_images(2, seed=1234 + 1000 * rank)]       # bench.py's per-rank fixture
        gathered, local, timing = _step_loop(model, imgs, synth_metas(2), dev, world, 3, {})
        ret[rank] = dict(gathered=None if gathered[0] is None else [[dict(b=g['bboxes'], n=len(g['masks'])) for g in step]
                                                                     for step in gathered],
                         local=[r.bboxes.cpu() for r in local])
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        ret[rank] = repr(e)


def test_two_rank_bench_loop_on_one_device_dataset_order():
    """`bench.py --gpus 2` in miniature: two processes (here sharing cuda:0 over gloo -- RCCL refuses two ranks on one
    device), each with its own ViT-B model and its own two tiles, three pipelined steps.  (a) rank 0 receives, every step,
    the four images in DATASET order (item j = image j // 2 of rank j % 2: mmengine DefaultSampler + collect_results),
    rank 1 receives nothing."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_loop, args=(2, port, ret), nprocs=2, join=True)
    if isinstance(ret[0], str) or isinstance(ret[1], str):
        msg = f'{ret[0]} / {ret[1]}'
        if 'gloo' in msg.lower() or 'not supported' in msg.lower() or 'not implemented' in msg.lower():
            pytest.skip(f'gloo cannot run this collective on device tensors here: {msg[:200]}')
        raise AssertionError(msg)
    assert ret[1]['gathered'] is None and len(ret[0]['gathered']) == 3
    for step in ret[0]['gathered']:
        assert len(step) == 4
        for j, g in enumerate(step):
            want = ret[j % 2]['local'][j // 2]
            assert g['n'] == want.shape[0] and torch.equal(g['b'], want), (j, g['n'], want.shape)
