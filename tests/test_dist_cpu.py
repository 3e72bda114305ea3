"""CPU, world_size 2 over gloo: the one collective of the inference path -- the all-gather of per-image
instance results -- plus the DefaultSampler-style sharding.  The HIP bit-pack kernel is
replaced by numpy.packbits(bitorder='little') (same bit order) because there is no GPU here."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _np_pack(masks):
    k = masks.shape[0]
    return torch.from_numpy(np.packbits(masks.reshape(k, -1).numpy().astype(np.uint8), axis=1, bitorder='little'))


def _make_results(rank, n_imgs, hw=(16, 24)):
    from rsprompter_amd.structures import InstanceData
    out = []
    for i in range(n_imgs):
        g = torch.Generator().manual_seed(100 * rank + i)
        k = [3, 0, 5, 2][(2 * rank + i) % 4]
        out.append(InstanceData(bboxes=torch.rand(k, 4, generator=g) * 50, scores=torch.rand(k, generator=g),
                                labels=torch.randint(0, 10, (k,), generator=g),
                                masks=torch.rand(k, *hw, generator=g) > 0.5))
    return out


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rsprompter_amd import dist as rdist
    r, _, w = rdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    res = _make_results(rank, 2)
    g = rdist.all_gather_results(res, pack_fn=_np_pack)
    ret[rank] = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in g.items()}
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_all_gather_results_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    g0, g1 = ret[0], ret[1]
    for k in ('counts', 'records', 'masks'):
        assert torch.equal(g0[k], g1[k])                 # every rank holds the same gathered result
    assert g0['counts'].tolist() == [3, 0, 5, 2]          # rank-major order
    K = g0['records'].shape[1]
    assert K == 5                                        # padded to the global max count
    for rank in range(world):
        for i, r in enumerate(_make_results(rank, 2)):
            j = rank * 2 + i
            k = len(r.bboxes)
            assert torch.equal(g0['records'][j, :k, :4], r.bboxes)
            assert torch.equal(g0['records'][j, :k, 4], r.scores)
            assert torch.equal(g0['records'][j, :k, 5].long(), r.labels)
            if k:
                bits = np.unpackbits(g0['masks'][j, :k].numpy(), axis=1, bitorder='little').astype(bool)
                assert np.array_equal(bits.reshape(k, *g0['mask_hw']), r.masks.numpy())
            assert not g0['records'][j, k:].any() and not g0['masks'][j, k:].any()


def test_shard_indices_round_robin():
    from rsprompter_amd.dist import shard_indices
    assert shard_indices(8, 0, 2) == [0, 2, 4, 6] and shard_indices(8, 1, 2) == [1, 3, 5, 7]
    parts = [shard_indices(10, r, 4) for r in range(4)]      # padded by wrap-around like DefaultSampler
    assert all(len(p) == 3 for p in parts)
    assert sorted(sum(parts, []))[:10] == [0, 0, 1, 1, 2, 3, 4, 5, 6, 7] or set(sum(parts, [])) == set(range(10))


def test_single_process_path_needs_no_process_group():
    from rsprompter_amd import dist as rdist
    g = rdist.all_gather_results(_make_results(0, 2), pack_fn=_np_pack)
    assert g['counts'].tolist() == [3, 0] and g['records'].shape[1] == 3


# ----------------------------------------------------------------------------- RLE exchange (default hand-off)
def _np_rle(masks):
    """CPU stand-in of ops.mask_rle_counts: COCO run lengths of the column-major pixel stream, first run = zeros."""
    k = masks.shape[0]
    rows = []
    for i in range(k):
        flat = masks[i].numpy().astype(np.uint8).T.reshape(-1)             # column-major
        change = np.flatnonzero(np.diff(flat)) + 1
        edges = np.concatenate([[0], change, [flat.size]])
        runs = np.diff(edges).tolist()
        if flat.size and flat[0] == 1:
            runs = [0] + runs
        rows.append(runs)
    cap = max([len(r) for r in rows] + [1])
    counts = torch.zeros((k, cap), dtype=torch.int32)
    for i, r in enumerate(rows):
        counts[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
    return counts, torch.tensor([len(r) for r in rows], dtype=torch.int32)


class NumpyCodec:
    """CPU stand-in of dist.DeviceCodec (rsp_mask_rle + rsp_rle_to_string) for the gloo tests: same contract, incl. the
    capacity reports (runs_needed > 0 / total > byte_cap mean "did not fit")."""

    def encode(self, results_list, run_cap, byte_cap, dev):
        from rsprompter_amd.rle import counts_to_string
        strs, runs_needed = [], 0
        for r in results_list:
            if r.bboxes.shape[0] == 0:
                continue
            counts, n = _np_rle(r.masks)
            for i in range(counts.shape[0]):
                ln = int(n[i])
                if ln > run_cap:
                    runs_needed = max(runs_needed, ln)
                    strs.append(b'')
                else:
                    strs.append(counts_to_string(counts[i, :ln].tolist()))
        lens = torch.tensor([len(x) for x in strs], dtype=torch.int32)
        total = int(lens.sum())
        flat = torch.zeros((max(byte_cap, 1),), dtype=torch.uint8)
        o = 0
        for x in strs:
            if o + len(x) <= byte_cap:
                flat[o:o + len(x)] = torch.frombuffer(bytearray(x), dtype=torch.uint8) if x else flat[o:o]
            o += len(x)
        return lens, flat, torch.tensor(total, dtype=torch.int64), torch.tensor(runs_needed, dtype=torch.int64)


def _hetero_results(item):
    """dataset item -> results with its OWN mask size (NWPU-style: every image has its own ori_shape)."""
    from rsprompter_amd.structures import InstanceData
    g = torch.Generator().manual_seed(500 + item)
    k = [3, 0, 5, 2, 1][item % 5]
    hw = [(16, 24), (9, 13), (30, 7), (11, 11), (5, 40)][item % 5]          # incl. H*W % 8 != 0
    return InstanceData(bboxes=torch.rand(k, 4, generator=g) * 50, scores=torch.rand(k, generator=g),
                        labels=torch.randint(0, 10, (k,), generator=g), masks=torch.rand(k, *hw, generator=g) > 0.5)


def _worker_rle(rank, world, port, n_items, dst, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rsprompter_amd import dist as rdist
    rdist.init_from_env(backend='gloo')
    state = rdist.ExchangeState()
    state.run_cap = 64                      # small on purpose: item 2 (30x7 noise) and the byte capacity have to grow
    out = []
    for rep in range(2):                    # second exchange: capacities already agreed, no growth
        mine = [_hetero_results(i) for i in rdist.shard_indices(n_items, rank, world)]
        got = rdist.gather_results(mine, dataset_size=n_items, codec=NumpyCodec(), dst=dst, state=state)
        out.append(None if got is None else list(got))
    ret[rank] = (out, (state.img_cap, state.inst_cap, state.byte_cap, state.run_cap))
    dist.barrier()
    dist.destroy_process_group()


def _check_items(got, n_items):
    from oracle import rle as orle
    assert len(got) == n_items                                     # padded duplicate dropped
    for i, g in enumerate(got):
        r = _hetero_results(i)
        assert torch.equal(g['bboxes'], r.bboxes) and torch.equal(g['scores'], r.scores)
        assert torch.equal(g['labels'], r.labels) and len(g['masks']) == len(r.bboxes)
        for j, rle in enumerate(g['masks']):
            assert rle['size'] == list(r.masks.shape[-2:])
            dec = orle.rle_decode(orle.rle_from_string(rle['counts']), *rle['size'])
            assert np.array_equal(dec, r.masks[j].numpy())


@pytest.mark.parametrize('dst', [0, None])
def test_gather_results_rle_heterogeneous_sizes_world2(dst):
    """world 2, 5 items (5 % 2 != 0: rank 1 gets a wrap-around duplicate), every item with its own mask size: results
    arrive on rank 0 only (mmengine collect_results) -- or on every rank with dst=None -- in DATASET order, truncated to
    the dataset size, masks as COCO RLE strings that decode to the input; the capacities start too small and every rank
    grows them identically from the all-gathered headers."""
    world, n_items = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rle, args=(world, _free_port(), n_items, dst, ret), nprocs=world, join=True)
    assert ret[0][1] == ret[1][1]                                      # identical capacities on both ranks
    for rank in range(world):
        for got in ret[rank][0]:
            if dst is None or rank == dst:
                _check_items(got, n_items)
            else:
                assert got is None


def test_gather_results_single_process_and_lazy_view():
    from rsprompter_amd import dist as rdist
    mine = [_hetero_results(i) for i in range(5)]
    got = rdist.gather_results(mine, codec=NumpyCodec(), state=rdist.ExchangeState())
    assert isinstance(got, rdist.GatheredResults) and got.n_instances == 11
    _check_items(list(got), 5)
    _check_items(got[0:5], 5)


def test_collect_host_budget_world8_800_instances():
    """What rank 0 does per step at world = 8 with 8 x 100 instances per rank (the bench shape): build the lazy view of
    the gathered buffers.  No per-instance Python work: the budget is 20 ms (it takes well under one)."""
    import time
    from rsprompter_amd import dist as rdist
    world, n_img, k = 8, 8, 100
    rng = np.random.default_rng(0)
    headers = np.zeros((world, 4), dtype=np.int64)
    lens = rng.integers(200, 1200, size=(world, 1024)).astype(np.int32)
    meta = np.zeros((world, 8, 3), dtype=np.int32)
    meta[:, :, 0], meta[:, :, 1], meta[:, :, 2] = k, 1024, 1024
    headers[:, 0], headers[:, 1] = n_img, n_img * k
    headers[:, 2] = lens[:, :n_img * k].sum(1)
    rec = rng.random((world, 1024, 6), dtype=np.float32)
    flat = rng.integers(48, 112, size=(world, 1 << 20), dtype=np.uint8)
    t = time.perf_counter()
    for _ in range(10):
        g = rdist.GatheredResults(headers, meta, rec, lens, flat, dataset_size=world * n_img)
    dt = (time.perf_counter() - t) / 10
    assert len(g) == 64 and g.n_instances == 6400
    assert dt < 0.020, f'collect-side host work {dt * 1e3:.2f} ms'
    item = g[9]                                                        # dataset item 9 = rank 1, image 1
    assert item['bboxes'].shape == (100, 4) and len(item['masks']) == 100
    b0 = int(lens[1, :100].sum())
    assert item['masks'][0]['counts'] == flat[1, b0:b0 + int(lens[1, 100])].tobytes()


def test_dense_exchange_rejects_heterogeneous_sizes():
    from rsprompter_amd import dist as rdist
    with pytest.raises(ValueError):
        rdist.all_gather_results([_hetero_results(0), _hetero_results(2)], pack_fn=_np_pack)


def test_poisoned_exchange_state_raises_instead_of_issuing_collectives():
    """a rank that CANCELLED an exchange whose headers exceeded the agreed capacities skipped the re-send and the
    capacity growth its peers performed; PendingGather.cancel() marks the group's state, and the next gather_results on it
    must raise (mismatched collectives would hang or corrupt) until release_state() / a new group."""
    from rsprompter_amd import dist as rdist
    state = rdist.ExchangeState()
    res = [_hetero_results(0)]
    assert len(rdist.gather_results(res, codec=NumpyCodec(), state=state)) == 1        # a healthy state works
    state.poisoned = 'a cancelled exchange of this process group needed capacities [9, 9, 9, 0] above the agreed ones'
    with pytest.raises(RuntimeError, match='cancelled exchange'):
        rdist.gather_results(res, codec=NumpyCodec(), state=state)
    assert len(rdist.gather_results(res, codec=NumpyCodec(), state=rdist.ExchangeState())) == 1   # a fresh state again
