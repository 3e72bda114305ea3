"""Round-3 GEMM experiments: the two-blocks-per-CU persistent kernel (csrc/gemm_s2.hip, tile hints 40 + variant)
against the round-2 kernels (csrc/gemm_dma.hip) on the ViT-H encoder shapes.

  python tools/gemm_s2_exp.py check     -> bit-exact comparison of both kernels over the epilogue / loader modes
  python tools/gemm_s2_exp.py time      -> one line per (shape, variant): ms, TFLOP/s (interleaved rounds, median)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsprompter_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
D, MLP, Mg, Mw = 1280, 5120, 32768, 39200
S2 = 40


def window_map(B, g=64, w=14):
    nw = (g + w - 1) // w
    idx = torch.arange(B * nw * nw * w * w, dtype=torch.int64)
    ix, iy = idx % w, (idx // w) % w
    wx, wy = (idx // (w * w)) % nw, (idx // (w * w * nw)) % nw
    b = idx // (w * w * nw * nw)
    y, x = wy * w + iy, wx * w + ix
    src = torch.where((y < g) & (x < g), b * g * g + y * g + x, torch.full_like(idx, -1))
    return src.to(torch.int32).to(dev)


def window_inverse(rm):
    """token row -> window-order row, and the padded window rows (the maps SamVisionEncoderHIP uses since round 3)"""
    rm = rm.long().cpu()
    real = rm >= 0
    t2w = torch.empty(int(real.sum()), dtype=torch.int64)
    t2w[rm[real]] = torch.arange(rm.numel())[real]
    return t2w.to(torch.int32).to(dev), torch.arange(rm.numel())[~real].to(torch.int32).to(dev)


def mk(n, k, bias=True):
    return ops.PackedWeight(torch.randn(n, k) / k ** 0.5, torch.randn(n) * 0.05 if bias else None, device=dev)


def same(a, b):
    if isinstance(a, tuple):
        return all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, ops.Planes):
        return torch.equal(a.hi, b.hi) and torch.equal(a.lo, b.lo)
    return torch.equal(a, b)


def check():
    torch.manual_seed(0)
    ok_all = True

    def run(name, fn):
        nonlocal ok_all
        ref = fn(1)                                          # tile hint 1: the round-2 rule
        for tag, h in (('specialised', S2), ('generic', S2 + 64), ('specialised + non-temporal stores', S2 + 1)):
            try:
                got = fn(h)
            except RuntimeError:
                continue                                   # that variant is not instantiated for this epilogue
            torch.cuda.synchronize()
            ok = same(ref, got)
            if not ok:
                flat = lambda o: [o] if isinstance(o, torch.Tensor) else ([o.hi, o.lo] if isinstance(o, ops.Planes) else sum([flat(x) for x in o], []))
                for n_, (a_, b_) in enumerate(zip(flat(ref), flat(got))):
                    bad = (a_ != b_)
                    if bool(bad.any()):
                        print(f'   part {n_} {tuple(a_.shape)}: {int(bad.sum())} of {bad.numel()} differ, first at '
                              f'{bad.nonzero()[:3].tolist()}, ref {a_[bad][:3].tolist()} got {b_[bad][:3].tolist()}')
            print(f'{"OK  " if ok else "FAIL"} {name} [{tag}]', flush=True)
            ok_all &= ok

    # (a) plain, ragged M and N (N % 4 == 0), short K
    for (M, N, K) in [(1000, 384, 256), (257, 132, 64), (5000, 1280, 1280), (300, 64, 128), (256, 128, 96 + 32)]:
        a = ops.to_planes(torch.randn(M, K, device=dev))
        w = mk(N, K)
        run(f'plain M={M} N={N} K={K} bias', lambda h: ops.gemm(a, w, tile_hint=h))
        run(f'plain M={M} N={N} K={K} relu', lambda h: ops.gemm(a, w, act=ops.ACT_RELU, tile_hint=h))
    # (b) window gather + scatter + residual (qkv / proj of a windowed layer, B = 2)
    B = 2
    rm = window_map(B)
    Mwin = rm.numel()
    x = torch.randn(B * 4096, D, device=dev)
    xp = ops.to_planes(x)
    wq = mk(3 * D, D)
    run('qkv windowed: a_rowmap, q fp32 + K|V planes (c_ncols / pl_col0)',
        lambda h: ops.gemm(xp, wq, a_rowmap=rm, M=Mwin, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h))
    run('qkv global: q fp32 + K|V planes', lambda h: ops.gemm(xp, wq, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h))
    t2w, pads = window_inverse(rm)

    def qkv_scatter(h):
        q, kv = ops.gemm(xp, wq, c_rowmap=t2w, out_rows=Mwin, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)
        ops.fill_bias_rows(wq.bias, pads, 3 * D, out=q, planes=kv, c_ncols=D, pl_col0=D)
        return q, kv
    run('qkv windowed, scatter form: c_rowmap to window order + bias fill of the padded rows', qkv_scatter)
    ref_q, ref_kv = ops.gemm(xp, wq, a_rowmap=rm, M=Mwin, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=1)
    got_q, got_kv = qkv_scatter(0)
    ok = same((ref_q, ref_kv), (got_q, got_kv))
    print(f'{"OK  " if ok else "FAIL"} qkv windowed: scatter form == gather form (padded rows multiplied), bit for bit', flush=True)
    ok_all &= ok
    att = ops.to_planes(torch.randn(Mwin, D, device=dev))
    wp = mk(D, D)
    run('proj windowed, gather form: a_rowmap from window order + fp32 residual',
        lambda h: ops.gemm(att, wp, res=x, a_rowmap=t2w, M=B * 4096, tile_hint=h))
    ok = torch.equal(ops.gemm(att, wp, res=x, a_rowmap=t2w, M=B * 4096),
                     ops.gemm(att, wp, res=x, c_rowmap=rm, out_rows=B * 4096, out=torch.zeros(B * 4096, D, device=dev), tile_hint=1))
    print(f'{"OK  " if ok else "FAIL"} proj windowed: gather form == scatter form, bit for bit', flush=True)
    ok_all &= ok
    run('proj windowed: c_rowmap scatter + fp32 residual',
        lambda h: ops.gemm(att, wp, res=x, c_rowmap=rm, out_rows=B * 4096, out=torch.zeros(B * 4096, D, device=dev), tile_hint=h))
    # (c) lin1 / lin2
    w1, w2 = mk(MLP, D), mk(D, MLP)
    run('lin1: GELU, planes only', lambda h: ops.gemm(xp, w1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=h))
    hm = ops.to_planes(torch.randn(B * 4096, MLP, device=dev))
    run('lin2: fp32 residual', lambda h: ops.gemm(hm, w2, res=x, tile_hint=h))
    run('lin2: fp32 + planes out', lambda h: ops.gemm(hm, w2, res=x, out_planes=True, tile_hint=h))
    # (d) plane residual, broadcast residual (res_mod), residual batch map
    rp = ops.to_planes(torch.randn(B * 4096, D, device=dev))
    run('plane residual', lambda h: ops.gemm(hm, w2, res=rp, tile_hint=h))
    pos = torch.randn(4096, D, device=dev)
    run('broadcast residual (res_mod)', lambda h: ops.gemm(xp, wp, res=pos, res_mod=4096, tile_hint=h))
    bmap = torch.tensor([1, 0, 1, 1, 0], dtype=torch.int32, device=dev)
    a5 = ops.to_planes(torch.randn(5 * 1024, 256, device=dev))
    w5 = mk(256, 256)
    r5 = torch.randn(2 * 1024, 256, device=dev)
    run('residual batch map (res_bmap)', lambda h: ops.gemm(a5, w5, res=r5, res_bmap=bmap, res_brows=1024, tile_hint=h))
    # (e) weight that is a row slice of activation planes (b_rows)
    mf = ops.to_planes(torch.randn(700, 256, device=dev))
    run('PlaneWeight (b_rows)', lambda h: ops.gemm(a5, ops.PlaneWeight(mf, r0=100, n=512), tile_hint=h))
    print('ALL OK' if ok_all else 'SOME FAILED', flush=True)
    return ok_all


def timed_rounds(fns, rounds=5, iters=4):
    """fns: {name: callable}; interleaved rounds, median ms per call"""
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / iters)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


def time_all():
    torch.manual_seed(0)
    xg = ops.to_planes(torch.randn(Mg, D, device=dev))
    xm = ops.to_planes(torch.randn(Mg, MLP, device=dev))
    res = torch.randn(Mg, D, device=dev)
    rm = window_map(8)
    t2w, _ = window_inverse(rm)
    w_qkv, w_proj, w_lin1, w_lin2 = mk(3 * D, D), mk(D, D), mk(MLP, D), mk(D, MLP)
    o_x = torch.empty(Mg, D, device=dev)
    att_w = ops.to_planes(torch.randn(Mw, D, device=dev))
    cases = {
        'qkv_window M=39200 N=3840 K=1280 (rowmap, q f32 + KV planes)':
            (Mw, 3 * D, D, lambda h: ops.gemm(xg, w_qkv, a_rowmap=rm, M=Mw, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)),
        'qkv_window_scatter M=32768 N=3840 K=1280 (token rows -> window order, q f32 + KV planes)':
            (Mg, 3 * D, D, lambda h: ops.gemm(xg, w_qkv, c_rowmap=t2w, out_rows=Mw, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)),
        'proj_window_gather M=32768 N=1280 K=1280 +res (A rows from window order)':
            (Mg, D, D, lambda h: ops.gemm(att_w, w_proj, res=res, a_rowmap=t2w, M=Mg, out=o_x, tile_hint=h)),
        'qkv_global M=32768 N=3840 K=1280':
            (Mg, 3 * D, D, lambda h: ops.gemm(xg, w_qkv, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h)),
        'proj_window M=39200 N=1280 K=1280 +res scatter':
            (Mw, D, D, lambda h: ops.gemm(att_w, w_proj, res=res, c_rowmap=rm, out_rows=Mg, out=o_x, tile_hint=h)),
        'proj M=32768 N=1280 K=1280 +res': (Mg, D, D, lambda h: ops.gemm(xg, w_proj, out=o_x, res=res, tile_hint=h)),
        'lin1 M=32768 N=5120 K=1280 gelu planes':
            (Mg, MLP, D, lambda h: ops.gemm(xg, w_lin1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=h)),
        'lin2 M=32768 N=1280 K=5120 +res': (Mg, D, MLP, lambda h: ops.gemm(xm, w_lin2, out=o_x, res=res, tile_hint=h)),
    }
    variants = [('product', 0), ('r2 auto', 1), ('r2 256x256', 17), ('s2', S2), ('s2 generic-epi', S2 + 64),
                ('s2 prio-DMA', S2 + 1), ('s2 DMA-burst', S2 + 33), ('s2 noDMA*', S2 + 4), ('s2 noEpi*', S2 + 8), ('s2 hotDMA*', S2 + 16)]
    only = {'qkv_window_scatter': ('product', 'r2 auto', 'r2 256x256', 's2', 's2 generic-epi'),
            'proj_window_gather': ('product', 'r2 auto', 'r2 256x256', 's2', 's2 generic-epi'),
            'qkv_window': ('product', 'r2 auto', 'r2 256x256', 's2', 's2 generic-epi', 's2 prio-DMA', 's2 DMA-burst'),
            'qkv_global': ('product', 'r2 auto', 'r2 256x256', 's2', 's2 generic-epi', 's2 prio-DMA', 's2 DMA-burst')}   # variants exist for proj / lin shapes
    for name, (M, N, K, fn) in cases.items():
        vs = [(vn, h) for vn, h in variants if name.split(' ')[0] not in only or vn in only[name.split(' ')[0]]]
        def guarded(h):
            def run():
                try:
                    fn(h)
                except RuntimeError:
                    pass                                   # variant not instantiated for this epilogue
            return run
        ms = timed_rounds({vn: guarded(h) for vn, h in vs})
        print(name + ':  ' + '  '.join(f'[{vn}] {t:.3f} ms {2.0 * M * N * K / t / 1e9:.0f}' for vn, t in ms.items()), flush=True)
    # long-K square: main loop only
    n = 8192
    a = ops.to_planes(torch.randn(n, n, device=dev))
    w = mk(n, n, bias=False)
    o = torch.empty(n, n, device=dev)
    vs = [(vn, h) for vn, h in variants if vn in ('product', 'r2 auto', 'r2 256x256', 's2')]     # plain epilogue: no variants
    ms = timed_rounds({vn: (lambda h=h: ops.gemm(a, w, out=o, tile_hint=h)) for vn, h in vs}, rounds=3, iters=2)
    print('8192^3:  ' + '  '.join(f'[{vn}] {t:.3f} ms {2.0 * n ** 3 / t / 1e9:.0f}' for vn, t in ms.items()), flush=True)


def trace(hint=S2 + 32, shape='lin1'):
    """time stamps of every (block, tile): who shares a CU, whether co-resident blocks stagger, what the phases cost"""
    import ctypes
    from rsprompter_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(512 * 16 * 4, dtype=torch.int64, device=dev)
    lib.rsp_debug_s2_trace.argtypes = [ctypes.c_void_p]
    lib.rsp_debug_s2_trace.restype = None
    lib.rsp_debug_s2_trace(buf.data_ptr())
    torch.manual_seed(0)
    xg = ops.to_planes(torch.randn(Mg, D, device=dev))
    if shape == 'lin1':
        w = mk(MLP, D)
        fn = lambda: ops.gemm(xg, w, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=hint)
    else:
        w = mk(D, D)
        res = torch.randn(Mg, D, device=dev)
        fn = lambda: ops.gemm(xg, w, res=res, tile_hint=hint)
    fn(); fn()
    torch.cuda.synchronize()
    buf.zero_()
    fn()
    torch.cuda.synchronize()
    lib.rsp_debug_s2_trace(None)
    t = buf.view(512, 16, 4).cpu()
    t0 = int(t[:, 0, 0][t[:, 0, 0] > 0].min())
    print(f'--- trace {shape} hint {hint}: per block: xcc se cu tg | tiles: start / loop / epilogue (cycles since first start)')
    rows = []
    for b in range(512):
        if int(t[b, 0, 0]) == 0:
            continue
        hw = int(t[b, 0, 3]) & 0xffffffff
        xcc = (int(t[b, 0, 3]) >> 32) & 0xf
        cu, se, tg, wv, simd = (hw >> 8) & 0xf, (hw >> 13) & 0x7, (hw >> 16) & 0xf, hw & 0xf, (hw >> 4) & 3
        sh = (hw >> 12) & 1
        tl = []
        for k in range(16):
            if int(t[b, k, 0]) == 0:
                break
            tl.append((int(t[b, k, 0]) - t0, int(t[b, k, 1]) - int(t[b, k, 0]), int(t[b, k, 2]) - int(t[b, k, 1])))
        rows.append(((xcc, se, sh, cu, tg), b, wv, simd, tl))
    rows.sort()
    for key, b, wv, simd, tl in rows[:48]:
        print(f'xcc{key[0]} se{key[1]} sh{key[2]} cu{key[3]:2d} tg{key[4]:2d} blk{b:3d} wslot{wv} simd{simd}: ' +
              ' '.join(f'{a}/{l}/{e}' for a, l, e in tl[:6]))
    import statistics
    loops = [l for r in rows for (_, l, _) in r[4]]
    epis = [e for r in rows for (_, _, e) in r[4]]
    print(f'tiles {len(loops)}: main loop cycles median {statistics.median(loops):.0f} (min {min(loops)}, max {max(loops)}); '
          f'epilogue median {statistics.median(epis):.0f} (min {min(epis)}, max {max(epis)})')
    # co-residency: blocks per (xcc, se, sh, cu)
    from collections import Counter
    c = Counter(r[0][:4] for r in rows)
    print('blocks per CU histogram:', Counter(c.values()))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'both'
    if what == 'trace':
        trace(S2 + 32, 'lin1'); trace(S2 + 33, 'lin1'); trace(S2 + 33, 'proj')
        sys.exit(0)
    if what in ('check', 'both'):
        if not check() and what == 'both':
            print('correctness failed: timing anyway (numbers are for structure only)')
    if what in ('time', 'both'):
        time_all()
