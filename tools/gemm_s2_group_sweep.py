"""Tile-order sweep of the s2 GEMM: group_m (M-tiles per group, tile_hint bits 8..15) per encoder shape, interleaved
rounds in one process.  python tools/gemm_s2_group_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402
from tools.gemm_s2_exp import timed_rounds, mk, D, MLP, Mg  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
xg = ops.to_planes(torch.randn(Mg, D, device=dev))
xm = ops.to_planes(torch.randn(Mg, MLP, device=dev))
res = torch.randn(Mg, D, device=dev)
o_x = torch.empty(Mg, D, device=dev)
w_qkv, w_proj, w_lin1, w_lin2 = mk(3 * D, D), mk(D, D), mk(MLP, D), mk(D, MLP)
cases = {
    'qkv_global': (3 * D, D, lambda h: ops.gemm(xg, w_qkv, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)),
    'proj': (D, D, lambda h: ops.gemm(xg, w_proj, out=o_x, res=res, tile_hint=h)),
    'lin1': (MLP, D, lambda h: ops.gemm(xg, w_lin1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=h)),
    'lin2': (D, MLP, lambda h: ops.gemm(xm, w_lin2, out=o_x, res=res, tile_hint=h)),
}
for name, (N, K, fn) in cases.items():
    ms = timed_rounds({f'g{g}': (lambda g=g: fn(g << 8)) for g in (1, 2, 4, 8, 16, 32, 64)})
    print(name + ':  ' + '  '.join(f'[{k}] {t:.3f} ms {2.0 * Mg * N * K / t / 1e9:.0f}' for k, t in ms.items()), flush=True)
