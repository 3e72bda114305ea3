"""Ad-hoc GPU probe: GEMM tile sweep -> gpurun_out/probe.json (not part of the product)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    dev = torch.device('cuda:0')
    res = {}
    shapes = [(32768, 3072, 768), (32768, 768, 3072), (39200, 2304, 768), (32768, 768, 768), (8192, 8192, 8192),
              (524288, 256, 2304)]
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device=dev)
        ap = ops.to_planes(a)
        w = ops.PackedWeight(torch.randn(N, K) * 0.02, torch.zeros(N), device=dev)
        out = torch.empty(M, N, device=dev)
        row = {}
        for hint in (1, 2, 3, 8, 10):
            try:
                t = timeit(lambda: ops.gemm(ap, w, out=out, tile_hint=hint))
                row[f'dma_tile{hint}'] = round(2.0 * M * N * K / t / 1e12, 1)
            except Exception as e:
                row[f'dma_tile{hint}'] = repr(e)[:60]
        t = timeit(lambda: ops.gemm(a, w, out=out, dma=False))
        row['regstage_128'] = round(2.0 * M * N * K / t / 1e12, 1)
        res[f'gemm_{M}x{N}x{K}'] = row
        print(f'gemm_{M}x{N}x{K}', row, flush=True)
        del a, ap, w, out
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/probe.json', 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
