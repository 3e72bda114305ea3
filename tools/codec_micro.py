"""The result codec of the multi-GPU exchange (rsprompter_amd/dist.py DeviceCodec: rsp_mask_rle per image +
rsp_rle_to_string) on one step's worth of results -- 8 images x 100 masks of 1024 x 1024 -- on an idle GPU:
ms per stage.   python tools/codec_micro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import dist as rdist, ops  # noqa: E402
from rsprompter_amd.structures import InstanceData  # noqa: E402

dev = torch.device('cuda:0')


def timed(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator().manual_seed(0)
res = []
for i in range(8):
    noise = torch.rand(100, 1, 4, 4, generator=g)
    m = torch.nn.functional.interpolate(noise, size=(1024, 1024), mode='bilinear')[:, 0] > 0.55     # blob-like masks
    res.append(InstanceData(bboxes=torch.rand(100, 4).to(dev), scores=torch.rand(100).to(dev),
                            labels=torch.zeros(100, dtype=torch.long).to(dev), masks=m.to(dev)))
K, cap = 800, 4096
counts = torch.empty((K, cap), dtype=torch.int32, device=dev)
ws = torch.empty((K, cap), dtype=torch.int32, device=dev)
n = torch.ones((K,), dtype=torch.int32, device=dev)


def rle_all():
    for i, r in enumerate(res):
        ops.mask_rle_into(r.masks, counts[100 * i:100 * i + 100], ws[100 * i:100 * i + 100], n[100 * i:100 * i + 100])


print(f'rsp_mask_rle, 8 x 100 masks of 1024 x 1024: {timed(rle_all):.3f} ms  (runs per mask: mean {float(n.float().mean()):.0f}, max {int(n.max())})')
print(f'rsp_rle_to_string (800 instances): {timed(lambda: ops.rle_to_string(counts, n, K, 1 << 20)):.3f} ms')
codec = rdist.DeviceCodec()
print(f'DeviceCodec.encode (both + buffers): {timed(lambda: codec.encode(res, cap, 1 << 20, dev)):.3f} ms')
state = rdist.ExchangeState()
side = torch.cuda.Stream(device=dev)
print(f'gather_results on a side stream + collect (world 1): {timed(lambda: rdist.gather_results(res, stream=side, state=state).collect()):.3f} ms '
      f'(capacities: {state.img_cap} images, {state.inst_cap} instances, {state.byte_cap} bytes, {state.run_cap} runs)')
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
h = rdist.gather_results(res, stream=side, state=state)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
h.collect()
t3 = time.perf_counter()
print(f'host: queue {1e3 * (t1 - t0):.2f} ms, device drain {1e3 * (t2 - t1):.2f} ms, collect {1e3 * (t3 - t2):.2f} ms')
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        rdist.gather_results(res, stream=side, state=state).collect()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=60))
