"""Round-5 GEMM experiment: the ping-pong kernel (csrc/gemm_pp.hip, tile hint 200) against the two-blocks-per-CU kernel
(csrc/gemm_s2.hip, hint 40) on the ViT-H / ViT-L / ViT-B encoder shapes at the bench batches: bit-equality of the results,
then interleaved timing (median of rounds), TFLOP/s = 2MNK / t.

  python tools/gemm_pp_exp.py [huge|large|base] [batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402
from tools.gemm_s2_exp import mk, same, timed_rounds, window_inverse, window_map  # noqa: E402

dev = torch.device('cuda:0')
ARCH = {'huge': (1280, 5120), 'large': (1024, 4096), 'base': (768, 3072)}


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else 'huge'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    D, MLP = ARCH[arch]
    Mg = B * 4096
    torch.manual_seed(0)
    xg = ops.to_planes(torch.randn(Mg, D, device=dev))
    xm = ops.to_planes(torch.randn(Mg, MLP, device=dev))
    res = torch.randn(Mg, D, device=dev)
    rm = window_map(B)
    Mw = rm.numel()
    t2w, _ = window_inverse(rm)
    w_qkv, w_proj, w_lin1, w_lin2 = mk(3 * D, D), mk(D, D), mk(MLP, D), mk(D, MLP)
    o_x = torch.empty(Mg, D, device=dev)
    att_w = ops.to_planes(torch.randn(Mw, D, device=dev))
    cases = {
        f'qkv_window_scatter M={Mg} N={3 * D} K={D}':
            (Mg, 3 * D, D, lambda h: ops.gemm(xg, w_qkv, c_rowmap=t2w, out_rows=Mw, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)),
        f'qkv_global M={Mg} N={3 * D} K={D}':
            (Mg, 3 * D, D, lambda h: ops.gemm(xg, w_qkv, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)),
        f'lin1 M={Mg} N={MLP} K={D} gelu planes':
            (Mg, MLP, D, lambda h: ops.gemm(xg, w_lin1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=h)),
        f'lin2 M={Mg} N={D} K={MLP} +res': (Mg, D, MLP, lambda h: ops.gemm(xm, w_lin2, out=o_x, res=res, tile_hint=h)),
        f'proj M={Mg} N={D} K={D} +res': (Mg, D, D, lambda h: ops.gemm(xg, w_proj, out=o_x, res=res, tile_hint=h)),
        f'proj_window_gather M={Mg} N={D} K={D} +res':
            (Mg, D, D, lambda h: ops.gemm(att_w, w_proj, res=res, a_rowmap=t2w, M=Mg, out=o_x, tile_hint=h)),
    }
    variants = [('auto', 0), ('s2', 40), ('pp', 200), ('pp gm4', 200 | (4 << 8)), ('pp128', 201), ('pp128 gm4', 201 | (4 << 8)),
                ('pp128 gm16', 201 | (16 << 8))]

    def guarded(fn, h):
        def run():
            try:
                return fn(h)
            except RuntimeError:
                return None                              # not a descriptor that kernel implements
        return run
    for name, (M, N, K, fn) in cases.items():
        a = fn(40)
        a = a.clone() if isinstance(a, torch.Tensor) else a
        ok = True
        for h in (200, 201):
            b = guarded(fn, h)()
            torch.cuda.synchronize()
            if b is None:
                continue
            if 'scatter' in name:        # rows nobody maps to are never written: compare the mapped rows
                rows = t2w.long()
                flat = lambda o: [o[0][rows], o[1].hi[:, rows], o[1].lo[:, rows]]
                ok &= all(torch.equal(x, y) for x, y in zip(flat(a), flat(b)))
            else:
                ok &= same(a, b)
        ms = timed_rounds({vn: guarded(fn, h) for vn, h in variants}, rounds=7, iters=4)
        print(f'{"OK  " if ok else "FAIL"} {name}:  ' + '  '.join(f'[{vn}] {t:.3f} ms {2.0 * M * N * K / t / 1e9:.0f}' for vn, t in ms.items()), flush=True)




def ablate(arch='huge', B=8):
    """development build (RSP_DEV_BUILD=1): lin1 and lin2 / proj with the K loop's DMA removed (hint 204), the epilogue
    removed (208), both (212) -- wrong results on purpose, they price the parts; then the time stamps (232)"""
    import ctypes
    import statistics
    from rsprompter_amd import _lib
    D, MLP = ARCH[arch]
    Mg = B * 4096
    torch.manual_seed(0)
    xg = ops.to_planes(torch.randn(Mg, D, device=dev))
    xm = ops.to_planes(torch.randn(Mg, MLP, device=dev))
    res = torch.randn(Mg, D, device=dev)
    w_proj, w_lin1, w_lin2, w_qkv = mk(D, D), mk(MLP, D), mk(D, MLP), mk(3 * D, D)
    o_x = torch.empty(Mg, D, device=dev)
    cases = {
        'lin1': (Mg, MLP, D, lambda h: ops.gemm(xg, w_lin1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=h)),
        'qkv': (Mg, 3 * D, D, lambda h: ops.gemm(xg, w_qkv, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)),
        'lin2': (Mg, D, MLP, lambda h: ops.gemm(xm, w_lin2, out=o_x, res=res, tile_hint=h)),
        'proj': (Mg, D, D, lambda h: ops.gemm(xg, w_proj, out=o_x, res=res, tile_hint=h)),
    }
    variants = [('s2', 40), ('pp', 200), ('pp noDMA*', 204), ('pp128', 201), ('pp128 noDMA*', 205)]
    def guarded(fn, h):
        def run():
            try:
                return fn(h)
            except RuntimeError:
                return None
        return run
    for name, (M, N, K, fn) in cases.items():
        if sys.argv[3:] and sys.argv[3] == 'alone':
            break
        ms = timed_rounds({vn: guarded(fn, h) for vn, h in variants}, rounds=5, iters=4)
        print(f'{name} M={M} N={N} K={K}:  ' + '  '.join(f'[{vn}] {t:.3f} ms {2.0 * M * N * K / t / 1e9:.0f}' for vn, t in ms.items()), flush=True)
    lib = _lib.load()
    buf = torch.zeros(256 * 16 * 2 * 4, dtype=torch.int64, device=dev)
    lib.rsp_debug_pp_trace.argtypes = [ctypes.c_void_p]
    lib.rsp_debug_pp_trace.restype = None
    sel = sys.argv[3:] and sys.argv[3] == 'alone'
    runs = (('lin1', 216), ('qkv', 216), ('proj', 216), ('lin2', 216), ('proj', 217), ('lin2', 217))
    if sel:      # how long is an epilogue when fewer CUs run: 1 / 4 / 16 / 32 blocks per XCD
        runs = tuple((n, 216 | (c << 16)) for n in ('lin1', 'qkv', 'proj', 'lin2') for c in (1, 4, 16, 32))
    for name, hint in runs:
        fn0 = cases[name][3]
        fn = lambda h, fn0=fn0, hint=hint: fn0(hint)
        fn(216); fn(216)
        torch.cuda.synchronize()
        lib.rsp_debug_pp_trace(buf.data_ptr())
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(216); e1.record()
        torch.cuda.synchronize()
        lib.rsp_debug_pp_trace(None)
        t = buf.view(256, 16, 2, 4).cpu()
        ms = e0.elapsed_time(e1)
        loops = [[], []]; epis = [[], []]; gaps = []
        buf_blocks = 256
        for b in range(buf_blocks):
            for g in range(2):
                for k in range(16):
                    if int(t[b, k, g, 0]) == 0:
                        break
                    loops[g].append(int(t[b, k, g, 1] - t[b, k, g, 0]))
                    epis[g].append(int(t[b, k, g, 2] - t[b, k, g, 1]))
                    if k > 0:
                        gaps.append(int(t[b, k, g, 0] - t[b, k - 1, g, 2]))
        t0 = int(t[:, 0, :, 0][t[:, 0, :, 0] > 0].min()); t1 = int(t[:, :, :, 2].max())
        nk = cases[name][2] // 16
        print(f'--- trace {name} hint {hint}: {ms:.3f} ms, {t1 - t0} cycles first start -> last end = {(t1 - t0) / ms / 1e3:.0f} MHz; tiles {len(loops[0])}; '
              f'K loop cycles median group0 {statistics.median(loops[0]):.0f} group1 {statistics.median(loops[1]):.0f} '
              f'(= {statistics.median(loops[0]) / (2 * nk):.0f} per phase; ideal {768 if (hint & 1) == 0 else 384}); epilogue median g0 {statistics.median(epis[0]):.0f} '
              f'g1 {statistics.median(epis[1]):.0f} (min {min(epis[1])}, max {max(epis[1])}); next-tile start gap median '
              f'{statistics.median(gaps) if gaps else 0:.0f}')
        b0 = t[0]
        print('    block 0: ' + ' | '.join(f'{int(b0[k, 0, 0] - t0)}+{int(b0[k, 0, 1] - b0[k, 0, 0])}+{int(b0[k, 0, 2] - b0[k, 0, 1])}'
                                          f' / g1 +{int(b0[k, 1, 1] - b0[k, 1, 0])}+{int(b0[k, 1, 2] - b0[k, 1, 1])}' for k in range(16) if int(b0[k, 0, 0])))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'ablate':
    ablate(*(sys.argv[2:3] or ['huge']))
    sys.exit(0)

if __name__ == '__main__':
    main()
