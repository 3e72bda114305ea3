"""Round-2 GEMM experiments on the ViT-H encoder shapes (epilogues as the encoder uses them):
  python tools/gemm_exp.py            -> one line per (shape, variant): ms, TFLOP/s
variants: tile hint | group_m << 8 (grouped tile order), 31/32 = pass-major MFMA order."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
D, MLP, Mg, Mw = 1280, 5120, 32768, 39200


def timed(fn, iters=6):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    torch.manual_seed(0)
    mk = lambda n, k: ops.PackedWeight(torch.randn(n, k) / k ** 0.5, torch.randn(n) * 0.05, device=dev)
    xg, xw = ops.to_planes(torch.randn(Mg, D, device=dev)), ops.to_planes(torch.randn(Mw, D, device=dev))
    xm = ops.to_planes(torch.randn(Mg, MLP, device=dev))
    res = torch.randn(Mg, D, device=dev)
    w_qkv, w_proj, w_lin1, w_lin2 = mk(3 * D, D), mk(D, D), mk(MLP, D), mk(D, MLP)
    o_q = torch.empty(Mw, 3 * D, device=dev)
    o_x = torch.empty(Mg, D, device=dev)
    cases = {
        'qkv_window M=39200 N=3840 K=1280': (Mw, 3 * D, D, lambda h: ops.gemm(xw, w_qkv, out=o_q, tile_hint=h)),
        'proj M=32768 N=1280 K=1280 +res': (Mg, D, D, lambda h: ops.gemm(xg, w_proj, out=o_x, res=res, tile_hint=h)),
        'lin1 M=32768 N=5120 K=1280 gelu planes': (Mg, MLP, D, lambda h: ops.gemm(xg, w_lin1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=h)),
        'lin2 M=32768 N=1280 K=5120 +res': (Mg, D, MLP, lambda h: ops.gemm(xm, w_lin2, out=o_x, res=res, tile_hint=h)),
    }
    variants = [('auto', 0), ('256x256', 17), ('256x128', 18), ('128x128', 14),
                ('256x256 g4', 17 | 4 << 8), ('256x256 g8', 17 | 8 << 8), ('256x128 g4', 18 | 4 << 8),
                ('256x128 g8', 18 | 8 << 8), ('256x128 g16', 18 | 16 << 8), ('128x128 g8', 14 | 8 << 8),
                ('256x256 passmajor', 31), ('256x128 passmajor', 32), ('256x256 pm g8', 31 | 8 << 8),
                ('256x128 pm g8', 32 | 8 << 8)]
    for name, (M, N, K, fn) in cases.items():
        line = name + ':'
        for vn, h in variants:
            try:
                ms = timed(lambda: fn(h))
                line += f'  [{vn}] {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f}'
            except Exception as e:
                line += f'  [{vn}] ERR {str(e)[:40]}'
        print(line, flush=True)


if __name__ == '__main__':
    main()
