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


def _hetero_results(item):
    """dataset item -> results with its OWN mask size (NWPU-style: every image has its own ori_shape)."""
    from rsprompter_amd.structures import InstanceData
    g = torch.Generator().manual_seed(500 + item)
    k = [3, 0, 5, 2, 1][item % 5]
    hw = [(16, 24), (9, 13), (30, 7), (11, 11), (5, 40)][item % 5]          # incl. H*W % 8 != 0
    return InstanceData(bboxes=torch.rand(k, 4, generator=g) * 50, scores=torch.rand(k, generator=g),
                        labels=torch.randint(0, 10, (k,), generator=g), masks=torch.rand(k, *hw, generator=g) > 0.5)


def _worker_rle(rank, world, port, n_items, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rsprompter_amd import dist as rdist
    rdist.init_from_env(backend='gloo')
    mine = [_hetero_results(i) for i in rdist.shard_indices(n_items, rank, world)]
    ret[rank] = rdist.gather_results(mine, dataset_size=n_items, rle_fn=_np_rle)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_rle_heterogeneous_sizes_world2():
    """world 2, 5 items (5 % 2 != 0: rank 1 gets a wrap-around duplicate), every item with its own mask size:
    results come back in DATASET order, truncated to the dataset size, masks as COCO RLE that decode to the input."""
    from oracle import rle as orle
    world, n_items = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rle, args=(world, _free_port(), n_items, ret), nprocs=world, join=True)
    for rank in range(world):
        got = ret[rank]
        assert len(got) == n_items                                     # padded duplicate dropped
        for i, g in enumerate(got):
            r = _hetero_results(i)
            assert torch.equal(g['bboxes'], r.bboxes) and torch.equal(g['scores'], r.scores)
            assert torch.equal(g['labels'], r.labels) and len(g['masks']) == len(r.bboxes)
            for j, rle in enumerate(g['masks']):
                assert rle['size'] == list(r.masks.shape[-2:])
                dec = orle.rle_decode(orle.rle_from_string(rle['counts']), *rle['size'])
                assert np.array_equal(dec, r.masks[j].numpy())


def test_dense_exchange_rejects_heterogeneous_sizes():
    from rsprompter_amd import dist as rdist
    with pytest.raises(ValueError):
        rdist.all_gather_results([_hetero_results(0), _hetero_results(2)], pack_fn=_np_pack)
