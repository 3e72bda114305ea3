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
