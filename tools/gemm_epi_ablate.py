"""What does the gemm_s2 epilogue cost, and which part?  lin1 / proj shapes with: full epilogue, epilogue without its
stores (tile hint +2), without GELU (act none), without anything (+8).  python tools/gemm_epi_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402
from tools.gemm_s2_exp import D, MLP, Mg, S2, dev, mk, timed_rounds  # noqa: E402

torch.manual_seed(0)
xg = ops.to_planes(torch.randn(Mg, D, device=dev))
w1, wp = mk(MLP, D), mk(D, D)
res = torch.randn(Mg, D, device=dev)
o_x = torch.empty(Mg, D, device=dev)
fns = {
    'lin1 gelu planes': lambda: ops.gemm(xg, w1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=S2),
    'lin1 gelu planes, no stores*': lambda: ops.gemm(xg, w1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=S2 + 2),
    'lin1 no act planes': lambda: ops.gemm(xg, w1, out_planes=True, out_f32=False, tile_hint=S2),
    'lin1 no act, no stores*': lambda: ops.gemm(xg, w1, out_planes=True, out_f32=False, tile_hint=S2 + 2),
    'lin1 no epilogue*': lambda: ops.gemm(xg, w1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=S2 + 8),
    'lin1 r2': lambda: ops.gemm(xg, w1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=17),
    'proj res f32': lambda: ops.gemm(xg, wp, out=o_x, res=res, tile_hint=S2),
    'proj res, no stores*': lambda: ops.gemm(xg, wp, out=o_x, res=res, tile_hint=S2 + 2),
    'proj no epilogue*': lambda: ops.gemm(xg, wp, out=o_x, res=res, tile_hint=S2 + 8),
    'proj no res': lambda: ops.gemm(xg, wp, out=o_x, tile_hint=S2),
    'proj r2': lambda: ops.gemm(xg, wp, out=o_x, res=res, tile_hint=1),
}
ms = timed_rounds(fns)
for k, t in ms.items():
    n, kk = (MLP, D) if k.startswith('lin1') else (D, D)
    print(f'{k:32s} {t:.3f} ms  {2.0 * Mg * n * kk / t / 1e9:.0f} TFLOP/s', flush=True)
