"""PMC micro-suite for the GEMM kernels: python tools/gemm_pmc_micro.py [iters]
runs {8192^3, lin1 shape} x {r2 256x256 (hint 17), s2 (40), s2 noDMA (44), s2 hotDMA (56)}; rows of the counter CSV are
told apart by kernel name (template arguments) and grid size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (M, N, K) in [(8192, 8192, 8192), (32768, 5120, 1280)]:
    a = ops.to_planes(torch.randn(M, K, device=dev))
    w = ops.PackedWeight(torch.randn(N, K) * 0.02, torch.zeros(N), device=dev)
    out = torch.empty(M, N, device=dev)
    for hint in (17, 40, 44, 56):
        for _ in range(iters):
            ops.gemm(a, w, out=out, tile_hint=hint)
    torch.cuda.synchronize()
    del a, w, out
