"""Time the plane-path GEMM over (shape x tile hint): python tools/gemm_sweep.py [M N K [res]] ...
Default: the skinny SAM-decoder shapes and the ViT-B encoder shapes of the bench workload."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
SHAPES = [(3276800, 256, 128, 1), (3276800, 256, 256, 0), (3276800, 128, 256, 0),
          (32768, 3072, 768, 0), (32768, 768, 3072, 1), (39200, 2304, 768, 0), (39200, 768, 768, 1),
          (32768, 768, 768, 1)]
HINTS = [(0, 'auto'), (14, 'P128x128'), (18, 'Q256x128'), (17, 'Q256x256'), (17 | 8 << 8, 'Q256x256 g8'), (21, 'P128x64 nbuf3'),
         (22, 'P128x64 nbuf4')]


def run(M, N, K, with_res, iters=5):
    a = ops.to_planes(torch.randn(M, K, device=dev))
    w = ops.PackedWeight(torch.randn(N, K) * 0.02, torch.zeros(N), device=dev)
    out = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev) if with_res else None
    line = f'M={M} N={N} K={K} res={with_res}:'
    ref = None
    for hint, name in HINTS:
        if (hint & 0xff) in (2, 4, 12, 13, 16, 18, 19) and N <= 64 or (hint & 0xff) in (3, 9, 11, 15, 17) and N <= 128:
            continue
        out.zero_()
        ops.gemm(a, w, out=out, res=res, tile_hint=hint)
        torch.cuda.synchronize()
        if 'noDMA' not in name:
            if ref is None:
                ref = out.clone()
            elif not torch.equal(ref, out):
                line += f'  !!MISMATCH {name} {float((ref - out).abs().max()):.3e}'

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm(a, w, out=out, res=res, tile_hint=hint)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 2.0 * M * N * K / ms / 1e9
        gb = 4.0 * (M * K + M * N * (2 if with_res else 1)) / ms / 1e6
        line += f'  [{name}] {ms:.3f} ms {tf:.0f} TF {gb:.0f} GB/s'
    print(line, flush=True)


if __name__ == '__main__':
    args = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(args[i:i + 4]) for i in range(0, len(args), 4)] if args else SHAPES
    for sh in shapes:
        run(*sh)
