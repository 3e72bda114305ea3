"""fp16x3 vs fp8-corrected GEMM (RSP_PLANE_F8) on the ViT-H encoder shapes, epilogues as the encoder uses them:
  python tools/gemm_f8_exp.py   -> one line per shape: ms and TFLOP/s for both products on the auto tile (+ hints)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402
from gemm_exp import timed  # noqa: E402

dev = torch.device('cuda:0')
D, MLP, Mg, Mw = 1280, 5120, 32768, 39200


def main():
    torch.manual_seed(0)
    res = torch.randn(Mg, D, device=dev)
    o_q = torch.empty(Mw, 3 * D, device=dev)
    o_x = torch.empty(Mg, D, device=dev)
    for f8 in (False, True):
        mk = lambda n, k: ops.PackedWeight(torch.randn(n, k) / k ** 0.5, torch.randn(n) * 0.05, device=dev, f8=f8)
        xg = ops.to_planes(torch.randn(Mg, D, device=dev), f8=f8)
        xw = ops.to_planes(torch.randn(Mw, D, device=dev), f8=f8)
        xm = ops.to_planes(torch.randn(Mg, MLP, device=dev), f8=f8)
        w_qkv, w_proj, w_lin1, w_lin2 = mk(3 * D, D), mk(D, D), mk(MLP, D), mk(D, MLP)
        cases = {
            'qkv_window M=39200 N=3840 K=1280 (q fp32 | kv planes)': (Mw, 3 * D, D, lambda h: ops.gemm(
                xw, w_qkv, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)),
            'proj M=32768 N=1280 K=1280 +res': (Mg, D, D, lambda h: ops.gemm(xg, w_proj, out=o_x, res=res, tile_hint=h)),
            'lin1 M=32768 N=5120 K=1280 gelu planes': (Mg, MLP, D, lambda h: ops.gemm(
                xg, w_lin1, act=ops.ACT_GELU, out_planes=True, out_f32=False, out_f8=f8, tile_hint=h)),
            'lin2 M=32768 N=1280 K=5120 +res': (Mg, D, MLP, lambda h: ops.gemm(xm, w_lin2, out=o_x, res=res, tile_hint=h)),
        }
        for name, (M, N, K, fn) in cases.items():
            line = ('f16+f8 ' if f8 else 'f16x3  ') + name + ':'
            for vn, h in (('auto', 0), ('256x256', 17), ('256x128', 18), ('128x128', 14)):
                ms = timed(lambda: fn(h))
                line += f'  [{vn}] {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f}'
            print(line, flush=True)


if __name__ == '__main__':
    main()
