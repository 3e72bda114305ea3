"""One process, several kernels in a fixed order -- the workload of the rocprofv3 PMC passes (tools/pmc_round2.sh).

  python tools/pmc_suite.py [--arch huge] [--batch 8] [--iters 3] [--what gemm,attn,ln]

Runs, `iters` times each: the encoder GEMM shapes of the bench workload (qkv windowed / qkv global / proj / lin1 /
lin2 with the epilogues the encoder uses), the global and the windowed attention (+ rel-pos) and the LayerNorm of one
layer.  Prints one JSON manifest line per kernel group with the HIP-event time, so that the counter CSVs (which only
carry kernel names and grid sizes) can be joined with shapes in tools/pmc_report.py."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402
from rsprompter_amd.nnutil import SAM_ARCH  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='huge')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--what', default='gemm,attn,ln')
    ap.add_argument('--only', default='', help='comma list of GEMM group names (default: all)')
    args = ap.parse_args()
    a = SAM_ARCH[args.arch]
    D, nh, mlp = a['hidden'], a['heads'], a['mlp']
    dh = D // nh
    B, T = args.batch, 4096
    Mg, Mw = B * T, B * 25 * 196
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    what = args.what.split(',')
    if 'gemm' in what:
        xg = ops.to_planes(torch.randn(Mg, D, device=dev))
        xw = ops.to_planes(torch.randn(Mw, D, device=dev))
        xm = ops.to_planes(torch.randn(Mg, mlp, device=dev))
        res = torch.randn(Mg, D, device=dev)
        mk = lambda n, k: ops.PackedWeight(torch.randn(n, k) / k ** 0.5, torch.randn(n) * 0.05, device=dev)
        w_qkv, w_proj, w_lin1, w_lin2 = mk(3 * D, D), mk(D, D), mk(mlp, D), mk(D, mlp)
        o_x = torch.empty(Mg, D, device=dev)
        # the encoder's forms since round 3 (sam_encoder.py): the windowed qkv scatters its token rows to window order
        # (the padded rows are a bias fill), the windowed proj gathers them back; q leaves as fp32, K | V as planes
        g_, w_ = 64, 14
        nw = (g_ + w_ - 1) // w_
        idx = torch.arange(B * nw * nw * w_ * w_)
        yy, xx = ((idx // (w_ * w_ * nw)) % nw) * w_ + (idx // w_) % w_, ((idx // (w_ * w_)) % nw) * w_ + idx % w_
        real = (yy < g_) & (xx < g_)
        t2w = torch.empty(Mg, dtype=torch.int64)
        t2w[((idx // (w_ * w_ * nw * nw)) * g_ * g_ + yy * g_ + xx)[real]] = idx[real]
        t2w = t2w.to(torch.int32).to(dev)
        cases = [
            ('qkv_window_scatter', Mg, 3 * D, D,
             lambda: ops.gemm(xg, w_qkv, c_rowmap=t2w, out_rows=Mw, out_planes=True, c_ncols=D, pl_col0=D)),
            ('qkv_global', Mg, 3 * D, D, lambda: ops.gemm(xg, w_qkv, out_planes=True, c_ncols=D, pl_col0=D)),
            ('proj_window_gather', Mg, D, D, lambda: ops.gemm(xw, w_proj, out=o_x, res=res, a_rowmap=t2w, M=Mg)),
            ('proj_global', Mg, D, D, lambda: ops.gemm(xg, w_proj, out=o_x, res=res)),
            ('lin1_gelu_planes', Mg, mlp, D, lambda: ops.gemm(xg, w_lin1, act=ops.ACT_GELU, out_planes=True, out_f32=False)),
            ('lin2_res', Mg, D, mlp, lambda: ops.gemm(xm, w_lin2, out=o_x, res=res)),
        ]
        only = [s_ for s_ in args.only.split(',') if s_]
        for name, m, n, k, fn in cases:
            if only and name not in only:
                continue
            ms = timed(fn, args.iters)
            print(json.dumps(dict(group=name, kernel='rsp_gemm (gemm_f16x3_pp_kernel where its tiles fill the CUs, else gemm_f16x3_s2_kernel)', M=m, N=n, K=k, ms=round(ms, 4),
                                  tflops=round(2.0 * m * n * k / ms / 1e9, 1), launches=args.iters + 1)), flush=True)
    if 'attn' in what:
        for name, Bp, S in (('attn_global', B, 64), ('attn_window', B * 25, 14)):
            Tt = S * S
            q = torch.randn(Bp * Tt, D, device=dev)
            kv = ops.to_planes(torch.randn(Bp * Tt, 2 * D, device=dev))
            rph = torch.randn(2 * S - 1, dh, device=dev) * 0.05
            rpw = torch.randn(2 * S - 1, dh, device=dev) * 0.05

            tab = ops.pack_relpos_tables(rph, rpw, S, dh) if S == 14 else None

            def fn():      # the encoder's sequence: global layers rel-pos terms + plane-fed attention (csrc/attn_stream.hip),
                if S == 14:    # windowed layers ONE kernel (csrc/attn_win.hip), padded queries of the 5 x 5 grid skipped
                    return ops.vit_window_attention(q, kv, tab, Bp, nh, dh, dh ** -0.5, planes=True, win_grid=(5, 8))
                rel = ops.vit_relpos(q, rph, rpw, Bp, S, nh, dh, q_ld=D)
                return ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, dh ** -0.5, planes=True)
            ms = timed(fn, args.iters)
            fl = 4.0 * Bp * nh * Tt * Tt * dh
            print(json.dumps(dict(group=name, Bp=Bp, S=S, nh=nh, dh=dh, ms_with_relpos=round(ms, 4),
                                  tflops_with_relpos=round(fl / ms / 1e9, 1), launches=args.iters + 1)), flush=True)
            del q, kv
    if 'ln' in what:
        x = torch.randn(Mg, D, device=dev)
        g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        ms = timed(lambda: ops.layernorm(x, g, b, 1e-6, planes=True, f32=False), args.iters)
        print(json.dumps(dict(group='layernorm_planes', rows=Mg, C=D, ms=round(ms, 4),
                              gbps=round(8.0 * x.numel() / ms / 1e6, 1), launches=args.iters + 1)), flush=True)


if __name__ == '__main__':
    main()
