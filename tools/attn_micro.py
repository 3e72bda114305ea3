"""micro benchmark of the ViT attention kernels for rocprofv3 PMC runs: python tools/attn_micro.py Bp S nh dh iters"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402

Bp, S, nh, dh, iters = [int(v) for v in sys.argv[1:6]]
dev = torch.device('cuda:0')
T = S * S
qkv = torch.randn(Bp * T, 3 * nh * dh, device=dev)
rph = torch.randn(2 * S - 1, dh, device=dev) * 0.1
rpw = torch.randn(2 * S - 1, dh, device=dev) * 0.1
rel = ops.vit_relpos(qkv, rph, rpw, Bp, S, nh, dh)
for _ in range(2):
    ops.vit_attention(qkv, rel, Bp, S, nh, dh, dh ** -0.5, planes=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.vit_attention(qkv, rel, Bp, S, nh, dh, dh ** -0.5, planes=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f'Bp={Bp} S={S} nh={nh} dh={dh}: {ms:.3f} ms  {4.0 * Bp * nh * T * T * dh / ms / 1e9:.1f} TFLOP/s')
