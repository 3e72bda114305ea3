"""micro benchmark of one GEMM config for rocprofv3 PMC runs: python tools/gemm_micro.py M N K hint iters"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402

M, N, K, hint, iters = [int(v) for v in sys.argv[1:6]]
dev = torch.device('cuda:0')
a = ops.to_planes(torch.randn(M, K, device=dev))
w = ops.PackedWeight(torch.randn(N, K) * 0.02, torch.zeros(N), device=dev)
out = torch.empty(M, N, device=dev)
for _ in range(iters):
    ops.gemm(a, w, out=out, tile_hint=hint)
torch.cuda.synchronize()
