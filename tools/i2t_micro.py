"""rsp_sam_i2t_fused at the bench shape (R = 800 prompt sets, T = 10 tokens, N = 4096 positions), layer-0 form (per-image
queries / residual through the RoI map) and layer-1 form (per-RoI plane residual): ms per call, GB/s of its algorithmic
traffic.  `--valu` asks for the fp32 copy of the result as well, which the round-2 VALU form serves (A/B).
python tools/i2t_micro.py [R] [--valu]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
VALU = '--valu' in sys.argv
_pos = [a for a in sys.argv[1:] if not a.startswith('--')]
R = int(_pos[0]) if _pos else 800
T, N, B = 10, 4096, 8
g = torch.Generator().manual_seed(0)
roi_img = (torch.arange(R) % B).to(torch.int32).to(dev)
k, v = torch.randn(R * T, 128, generator=g).to(dev), torch.randn(R * T, 128, generator=g).to(dev)
wo, bo = (torch.randn(256, 128, generator=g) / 128 ** 0.5).to(dev), torch.randn(256, generator=g).to(dev)
gamma, beta = torch.randn(256, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
q0, res0 = torch.randn(B * N, 128, generator=g).to(dev) * 2, torch.randn(B * N, 256, generator=g).to(dev) * 3
kw = dict(R=R, T=T, N=N, scale=0.25, eps=1e-6, f32=VALU)


def timed(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


pl0 = ops.sam_i2t_fused(q0, k, v, wo, bo, gamma, beta, q_map=roi_img, res=res0, res_map=roi_img, **kw)
if VALU:
    pl0 = pl0[1]
ms0 = timed(lambda: ops.sam_i2t_fused(q0, k, v, wo, bo, gamma, beta, q_map=roi_img, res=res0, res_map=roi_img, **kw))
q1 = torch.randn(R * N, 128, device=dev)
ms1 = timed(lambda: ops.sam_i2t_fused(q1, k, v, wo, bo, gamma, beta, res_planes=pl0, **kw))
tag = 'VALU (round 2; serves fp32-copy requests)' if VALU else 'MFMA (round 3, default)'
print(f'{tag}: layer-0 form {ms0:.3f} ms ({R * N * 1024 / ms0 / 1e6:.0f} GB/s written), layer-1 form {ms1:.3f} ms '
      f'({R * N * 2560 / ms1 / 1e6:.0f} GB/s moved); checksum {float(pl0.hi.float().abs().mean()):.6f} {float(pl0.lo.float().abs().mean()):.6f}')
