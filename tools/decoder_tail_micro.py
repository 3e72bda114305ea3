"""The per-RoI passes of the SAM mask decoder at the bench shape (R = 800 prompt sets x 4096 keys x 256 channels): ms per call
of the token -> image attention in its projected form (K | V GEMM + sam_t2i_kernel), folded (csrc/t2i_fold.hip, variants 0 -
3), and of the upscaler tail as two kernels (ConvTranspose GEMM with its LayerNorm epilogue + sam_upscale2_kernel) and as one
(sam_upscale_fused_kernel).  The folded variants 1 - 3 and the fused upscaler have been verified on the lane-level emulator only
(tests/test_wave_emu_cpu.py): this tool is their first timing.

  python tools/decoder_tail_micro.py [R] [iters]        one line per form (R <= 1023: the folded kernel's 32-bit offsets)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402
from rsprompter_amd.sam_decoder import SamMaskDecoderHIP  # noqa: E402
from rsprompter_amd.synth import synth_state_dict  # noqa: E402


def timed(fn, iters, dev):
    fn(); fn()
    if dev.type != 'cuda':
        return float('nan')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main(dev=None, R=800, iters=10, h=64, w=64, T=10):
    dev = dev or torch.device('cuda:0')
    N = h * w
    dec = SamMaskDecoderHIP()
    dec.load_state_dict(synth_state_dict(dec, 1))
    dec = dec.to(dev)
    dec._pack()
    P = dec._packed
    g = torch.Generator().manual_seed(0)
    keys_pl = ops.to_planes((torch.randn(R * N, 256, generator=g) * 1.5).to(dev))
    pe_t = dec._pe_terms(torch.randn(N, 256, generator=g).to(dev).contiguous())
    tq = (torch.randn(R * T, 128, generator=g) * 2).to(dev)
    hy = torch.randn(R, 32, generator=g).to(dev)
    ln = dec.upscale_layer_norm
    ao = torch.empty((R * T, 128), dtype=torch.float32, device=dev)

    def projected():
        kv = ops.gemm(keys_pl, P['final.kv_proj'], bias=None, res=pe_t['final.kv_proj'], res_mod=N)
        dec._t2i(tq, kv, ao, R, T, N)

    def folded(variant):
        dec.t2i_fold_variant = variant
        return dec._t2i_folded('final', tq, keys_pl, pe_t, R, T, N)

    def two_kernels():
        up = ops.conv_transpose2x2(keys_pl.view(R, h, w, 256), *P['up1'], act=ops.ACT_GELU, ln=(ln.weight, ln.bias, 1e-6))
        return ops.conv_transpose2x2(up, *P['up2'], act=ops.ACT_GELU, hyper=hy)

    def fused():
        return ops.sam_upscale_fused(keys_pl, P['up1'][0], P['up1'][1], ln.weight, ln.bias, 1e-6, P['up2p'][0], P['up2p'][1],
                                     hy, h, w)

    gb = R * N * 1024 / 1e9
    for name, fn in (('token -> image: K | V GEMM + sam_t2i_kernel', projected),
                     ('token -> image: folded, variant 0 (DMA burst; measured in round 4)', lambda: folded(0)),
                     ('token -> image: folded, variant 1 (DMA spread between the MFMAs)', lambda: folded(1)),
                     ('token -> image: folded, variant 2 (burst, one score accumulator, reads 2 steps ahead)', lambda: folded(2)),
                     ('token -> image: folded, variant 3 (spread, one score accumulator, reads 2 steps ahead)', lambda: folded(3)),
                     ('upscaler tail: ConvT GEMM + LN epilogue, then sam_upscale2_kernel', two_kernels),
                     ('upscaler tail: sam_upscale_fused_kernel', fused)):
        ms = timed(fn, iters, dev)
        print(f'R={R} {name:<72s} {ms:7.3f} ms   ({gb:.2f} GB of key planes: {gb / ms if ms == ms else 0:.2f} TB/s if read once)')
    d = float((two_kernels() - fused()).abs().max())
    print(f'fused vs two-kernel upscaler: max abs difference {d:.2e}')


if __name__ == '__main__':
    main(R=int(sys.argv[1]) if len(sys.argv) > 1 else 800, iters=int(sys.argv[2]) if len(sys.argv) > 2 else 10)
