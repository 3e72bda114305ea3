"""Probe: does de-phasing the two waves of a SIMD help the window attention?  (round-1 window-resident kernel, no barrier
in its loop; RSP_ATTN_SKEW = s_sleep units applied once to waves 4..6)"""
import os, sys, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from rsprompter_amd import ops
    dev = torch.device('cuda:0')
    Bp, S, nh, dh = 200, 14, 16, 80
    qkv = torch.randn(Bp * S * S, 3 * nh * dh, device=dev)
    rel = torch.randn(Bp * nh, S * S, 2 * S, device=dev) * 0.1
    for _ in range(3):
        ops.vit_attention(qkv, rel, Bp, S, nh, dh, dh ** -0.5, planes=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.vit_attention(qkv, rel, Bp, S, nh, dh, dh ** -0.5, planes=True)
    e1.record(); torch.cuda.synchronize()
    print('skew', os.environ.get('RSP_ATTN_SKEW', '0'), 'ms', round(e0.elapsed_time(e1) / 10, 4), flush=True)
else:
    for sk in (0, 1, 2, 3, 4, 6, 8, 12):
        subprocess.call([sys.executable, __file__, 'run'], env=dict(os.environ, RSP_ATTN_SKEW=str(sk)))
