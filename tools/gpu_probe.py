"""Ad-hoc GPU probe: kernel timings -> gpurun_out/probe.json (not part of the product)."""
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
    res = {'device': torch.cuda.get_device_name(0)}
    for (M, N, K) in [(32768, 3072, 768), (32768, 768, 3072), (39200, 2304, 768), (32768, 768, 768),
                      (8192, 8192, 8192), (524288, 256, 2304)]:
        a = torch.randn(M, K, device=dev)
        w = ops.PackedWeight(torch.randn(N, K) * 0.02, torch.zeros(N), device=dev)
        out = torch.empty(M, N, device=dev)
        t = timeit(lambda: ops.gemm(a, w, out=out))
        res[f'gemm_{M}x{N}x{K}'] = dict(ms=t * 1e3, tflops=2.0 * M * N * K / t / 1e12)
        del a, w, out
    # attention: ViT-B global (B=8) and windowed
    for (Bp, S, nh, dh) in [(8, 64, 12, 64), (200, 14, 12, 64)]:
        qkv = torch.randn(Bp, S * S, 3, nh, dh, device=dev)
        rph = torch.randn(2 * S - 1, dh, device=dev) * 0.05
        rpw = torch.randn(2 * S - 1, dh, device=dev) * 0.05
        t0 = timeit(lambda: ops.vit_relpos(qkv, rph, rpw, Bp, S, nh, dh))
        rel = ops.vit_relpos(qkv, rph, rpw, Bp, S, nh, dh)
        t1 = timeit(lambda: ops.vit_attention(qkv, rel, Bp, S, nh, dh, dh ** -0.5))
        fl = 4.0 * Bp * nh * (S * S) ** 2 * dh
        res[f'attn_S{S}_Bp{Bp}'] = dict(relpos_ms=t0 * 1e3, attn_ms=t1 * 1e3, attn_tflops=fl / t1 / 1e12)
    x = torch.randn(32768, 768, device=dev)
    g = torch.ones(768, device=dev)
    t = timeit(lambda: ops.layernorm(x, g, g))
    res['layernorm_32768x768'] = dict(ms=t * 1e3, gbps=2 * x.numel() * 4 / t / 1e9)
    # whole encoder
    from rsprompter_amd.sam_encoder import RSSamVisionEncoder
    from rsprompter_amd.synth import synth_state_dict
    for arch, B in [('base', 8)]:
        m = RSSamVisionEncoder(f'sam_vit_{arch}', extra_config=dict(output_hidden_states=True))
        m.vision_encoder.load_state_dict(synth_state_dict(m.vision_encoder, 0))
        m = m.to(dev)
        xin = torch.randn(B, 3, 1024, 1024, device=dev)
        t = timeit(lambda: m(xin), n=3, warm=1)
        res[f'encoder_{arch}_B{B}'] = dict(ms=t * 1e3, img_per_s=B / t)
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/probe.json', 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
