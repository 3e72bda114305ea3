"""Window attention at the encoder shapes (csrc/attn_win.hip): ms per layer call, padded-window TFLOP/s (4 Bp nh 196^2 dh)
and the TFLOP/s of the queries actually evaluated; fused rel-pos form (256 persistent blocks; other grid sizes for comparison)
against the rel-tensor form (+ the stand-alone rel-pos kernel it needs).

  python tools/attn_win_micro.py [iters]                 one line per (arch, form)
  python tools/attn_win_micro.py pmc <arch> <iters>      only the fused product form in a loop (rocprofv3 --pmc workload)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
ARCH = {'huge': (16, 80), 'large': (16, 64), 'base': (12, 64)}


def timed(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def setup(arch, B=8):
    nh, dh = ARCH[arch]
    S, D, Bp = 14, nh * dh, B * 25
    q = torch.randn(Bp * 196, D, device=dev)
    kv = ops.to_planes(torch.randn(Bp * 196, 2 * D, device=dev))
    rph, rpw = torch.randn(27, dh, device=dev) * 0.1, torch.randn(27, dh, device=dev) * 0.1
    tab = ops.pack_relpos_tables(rph, rpw, S, dh)
    return nh, dh, Bp, q, kv, rph, rpw, tab


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'pmc':
        arch, iters = sys.argv[2], int(sys.argv[3])
        nh, dh, Bp, q, kv, rph, rpw, tab = setup(arch)
        for _ in range(iters):
            ops.vit_window_attention(q, kv, tab, Bp, nh, dh, dh ** -0.5, planes=True, win_grid=(5, 8))
        torch.cuda.synchronize()
        return
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for arch in ('huge', 'large', 'base'):
        nh, dh, Bp, q, kv, rph, rpw, tab = setup(arch)
        sc = dh ** -0.5
        pad = 4.0 * Bp * nh * 196 * 196 * dh
        real = pad * 4096 / 4900                      # 64 x 64 real tokens of the 5 x 5 x 196 padded ones are queries
        rows = torch.arange(Bp * 196, device=dev, dtype=torch.int32)
        forms = {
            'fused, product (768 blocks)': lambda: ops.vit_window_attention(q, kv, tab, Bp, nh, dh, sc, planes=True, win_grid=(5, 8)),
            'fused, 256 blocks (one per CU)': lambda: ops.vit_window_attention(q, kv, tab, Bp, nh, dh, sc, planes=True, win_grid=(5, 8), variant=16),
            'fused, 512 blocks (two rounds)': lambda: ops.vit_window_attention(q, kv, tab, Bp, nh, dh, sc, planes=True, win_grid=(5, 8), variant=32),
            'fused, 768 blocks': lambda: ops.vit_window_attention(q, kv, tab, Bp, nh, dh, sc, planes=True, win_grid=(5, 8), variant=48),
            'fused, 1024 blocks': lambda: ops.vit_window_attention(q, kv, tab, Bp, nh, dh, sc, planes=True, win_grid=(5, 8), variant=64),
            'fused, every query (no grid)': lambda: ops.vit_window_attention(q, kv, tab, Bp, nh, dh, sc, planes=True),
        }
        rel = ops.vit_relpos(q, rph, rpw, Bp, 14, nh, dh, q_ld=nh * dh)
        forms['rel tensor given (attention only)'] = lambda: ops.vit_attention_planes(q, kv, rel, Bp, 14, nh, dh, sc, planes=True, win_grid=(5, 8))
        forms['stand-alone rel-pos kernel (all rows)'] = lambda: ops.vit_relpos(q, rph, rpw, Bp, 14, nh, dh, q_ld=nh * dh)
        for name, fn in forms.items():
            ms = timed(fn, iters)
            extra = '' if 'rel-pos kernel' in name else f'  {pad / ms / 1e9:6.1f} TFLOP/s padded, {real / ms / 1e9:6.1f} evaluated'
            print(f'ViT-{arch} B=8 (Bp={Bp}, nh={nh}, dh={dh}) {name:42s} {ms:.3f} ms{extra}', flush=True)


if __name__ == '__main__':
    main()
