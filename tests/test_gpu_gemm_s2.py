"""-m gpu: gemm_f16x3_s2_kernel (csrc/gemm_s2.hip) -- the kernel that carries two thirds of the ViT-H step -- on its own.

Every compile-time epilogue specialisation and the run-time form, the loader modes (row gather with padded rows, weight
taken from activation planes), ragged M / N, column-range outputs, K in {128, 1280, 5120}: each case is compared with
the fp64 product (the Linear calls it replaces: HF:803-831 qkv / proj, HF:132-143 MLP) AND must equal, bit for bit, what
gemm_f16x3_dma_kernel (tile hint 1, the round-2 rule) produces from the same planes.  Tile hints: 40 = this kernel with
the epilogue its dispatcher picks, 104 = this kernel with the run-time epilogue, 0 = rsp_gemm's own choice.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.quick]      # quick: the kernel-level tier (`-m "gpu and quick"`, < 2 min)

S2, S2_GENERIC, R2, PP, PP128 = 40, 104, 1, 200, 201
E_RES, E_GELU, E_C, E_PL, E_RMAP, E_GENERIC = 1, 2, 4, 8, 16, 64


def _pl64(p):
    """Planes -> fp64 [rows, K] on the CPU"""
    kb, rows, _ = p.hi.shape
    v = (p.hi.double() + p.lo.double()).cpu() * 2.0 ** -p.scale_log2
    return v.permute(1, 0, 2).reshape(rows, kb * 32)[:, :p.shape[-1]]


def _flat(o):
    from rsprompter_amd import ops
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, ops.Planes):
        return [o.hi, o.lo]
    return sum([_flat(x) for x in o], [])


def _same(a, b, rows=None):
    """bit-equal (rows: only these output rows were written -- fp32 [rows, N] tensors and [K/32, rows, 32] planes)"""
    def sel(t):
        if rows is None:
            return t
        return t[:, rows.to(t.device)] if t.dtype == torch.float16 else t[rows.to(t.device)]
    return all(torch.equal(sel(x), sel(y)) for x, y in zip(_flat(a), _flat(b)))


def _rel(got, ref, rows=None):
    """max abs error / max |ref| (rows: compare these output rows; ref may already be restricted to them)"""
    got = got.double().cpu() if isinstance(got, torch.Tensor) else _pl64(got)
    if rows is not None:
        got = got[rows]
        ref = ref[rows] if ref.shape[0] != got.shape[0] else ref
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def _mk(g, n, k, dev, bias=True):
    from rsprompter_amd import ops
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g) * 0.3 if bias else None
    return w, b, ops.PackedWeight(w, b, device=dev)


def _check(fn, refs, want_epi, rows=None, tol=2e-6):
    """fn(hint) -> output(s); refs: fp64 references in the same order; want_epi: the specialisation hint 40 must run"""
    assert fn(S2, plan_only=True) == want_epi, (fn(S2, plan_only=True), want_epi)
    base = fn(R2)                                         # gemm_f16x3_dma_kernel
    assert fn(R2, plan_only=True) == -1
    for o, r in zip(base if isinstance(base, tuple) else (base,), refs):
        assert _rel(o, r, rows) < tol
    # PP / PP128: gemm_f16x3_pp_kernel (csrc/gemm_pp.hip, K >= 128) with its 256 x 256 and 128 x 256 tiles, also with one
    # block per XCD (several tiles per block)
    for hint in (S2, S2_GENERIC, PP, PP128, PP | (1 << 16), PP128 | (1 << 16)):
        try:
            got = fn(hint)
        except RuntimeError:
            assert (hint & 0xff) in (PP, PP128) and fn(hint, plan_only='pp') == 0   # not a descriptor that kernel implements
            continue
        torch.cuda.synchronize()
        for o, r in zip(got if isinstance(got, tuple) else (got,), refs):
            assert _rel(o, r, rows) < tol, hint
        assert _same(base, got, rows), f'hint {hint}: differs from gemm_f16x3_dma_kernel'


@pytest.mark.parametrize('M,N,K', [(1000, 384, 256), (300, 64, 128), (513, 1280, 1280), (700, 256, 5120), (256, 128, 128)])
def test_s2_plain_and_residual(dev, M, N, K):
    """E_C (bias only) and E_C | E_RES (proj of a global layer, lin2, patch embed) incl. the broadcast residual"""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w, b, pw = _mk(g, N, K, dev)
    res = torch.randn(M, N, generator=g)
    ap = ops.to_planes(a.to(dev))
    ref = a.double() @ w.double().t() + b.double()
    _check(lambda h, **kw: ops.gemm(ap, pw, tile_hint=h, **kw), [ref], E_C)
    rd = res.to(dev)
    _check(lambda h, **kw: ops.gemm(ap, pw, res=rd, tile_hint=h, **kw), [ref + res.double()], E_C | E_RES)
    _check(lambda h, **kw: ops.gemm(ap, pw, bias=None, tile_hint=h, **kw), [ref - b.double()], E_C)
    if M % 100 == 0:
        pos = torch.randn(100, N, generator=g)
        pd = pos.to(dev)
        _check(lambda h, **kw: ops.gemm(ap, pw, res=pd, res_mod=100, tile_hint=h, **kw),
               [ref + pos.double().repeat(M // 100, 1)], E_C | E_RES)


@pytest.mark.parametrize('M,N,K', [(900, 320, 256), (515, 1280, 1280)])
def test_s2_gelu_and_plane_outputs(dev, M, N, K):
    """E_PL | E_GELU (lin1), E_PL, E_C | E_GELU, fp32 + planes of the same columns"""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(7 + M)
    a = torch.randn(M, K, generator=g) * 1.5
    w, b, pw = _mk(g, N, K, dev)
    ap = ops.to_planes(a.to(dev))
    ref = a.double() @ w.double().t() + b.double()
    # plane outputs carry ~22 bits of the fp32 value: same tolerance class
    _check(lambda h, **kw: ops.gemm(ap, pw, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=h, **kw),
           [F.gelu(ref)], E_PL | E_GELU)
    _check(lambda h, **kw: ops.gemm(ap, pw, out_planes=True, out_f32=False, tile_hint=h, **kw), [ref], E_PL)
    _check(lambda h, **kw: ops.gemm(ap, pw, act=ops.ACT_GELU, tile_hint=h, **kw), [F.gelu(ref)], E_C | E_GELU)
    _check(lambda h, **kw: ops.gemm(ap, pw, out_planes=True, tile_hint=h, **kw), [ref, ref], E_C | E_PL)


@pytest.mark.parametrize('D,K,M', [(128, 256, 777), (320, 1280, 600)])
def test_s2_column_ranges_and_row_maps(dev, D, K, M):
    """the qkv hand-off: q columns fp32 (c_ncols), K | V columns as planes (pl_col0) -- plain (E_C | E_PL), with the
    token -> window-order scatter (E_C | E_PL | E_RMAP); proj of a windowed layer: row gather from window order with
    padded rows + residual (E_C | E_RES), and the scatter form with holes (E_C | E_RES | E_RMAP)"""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(11 + D)
    a = torch.randn(M, K, generator=g)
    w, b, pw = _mk(g, 3 * D, K, dev)
    ap = ops.to_planes(a.to(dev))
    ref = a.double() @ w.double().t() + b.double()
    _check(lambda h, **kw: ops.gemm(ap, pw, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h, **kw),
           [ref[:, :D], ref[:, D:]], E_C | E_PL)
    # scatter to a longer, permuted row space with unwritten rows (the padded window rows)
    Mout = M + 150
    perm = torch.randperm(Mout, generator=g)[:M].to(torch.int32)
    pd = perm.to(dev)

    def qkv_scatter(h, **kw):
        if kw:
            return ops.gemm(ap, pw, c_rowmap=pd, out_rows=Mout, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h, **kw)
        q, kv = ops.gemm(ap, pw, c_rowmap=pd, out_rows=Mout, out_planes=True, c_ncols=D, pl_col0=D, tile_hint=h,
                         out=torch.zeros(Mout, D, device=dev))
        return q, kv
    sref = torch.zeros(Mout, 3 * D, dtype=torch.float64)
    sref[perm.long()] = ref
    written = perm.long().sort().values
    got = qkv_scatter(S2)
    assert _rel(got[0], sref[:, :D]) < 2e-6 and _rel(got[1], sref[:, D:], rows=written) < 2e-6
    _check(qkv_scatter, [sref[:, :D], sref[:, D:]], E_C | E_PL | E_RMAP, rows=written)
    # gather with padded (negative) rows + fp32 residual
    w2, b2, pw2 = _mk(g, D * 2, K, dev)
    amap = torch.randint(-1, M, (M + 90,), generator=g, dtype=torch.int32)
    res = torch.randn(M + 90, 2 * D, generator=g)
    gref = torch.where((amap >= 0)[:, None], a.double()[amap.clamp(min=0).long()], torch.zeros(1, dtype=torch.float64))
    gref = gref @ w2.double().t() + b2.double() + res.double()
    amd, rd = amap.to(dev), res.to(dev)
    _check(lambda h, **kw: ops.gemm(ap, pw2, a_rowmap=amd, M=M + 90, res=rd, tile_hint=h, **kw), [gref], E_C | E_RES)
    # scatter with holes (c_rowmap < 0: row dropped) + residual indexed by the DESTINATION row
    cmap = torch.where(torch.rand(M, generator=g) < 0.2, torch.full((M,), -1, dtype=torch.int32),
                       torch.randperm(M, generator=g).to(torch.int32))
    res2 = torch.randn(M, 2 * D, generator=g)
    full = a.double() @ w2.double().t() + b2.double()
    sref2 = torch.full((M, 2 * D), 7.0, dtype=torch.float64)
    keep = cmap >= 0
    sref2[cmap[keep].long()] = full[keep] + res2.double()[cmap[keep].long()]
    cmd, r2d = cmap.to(dev), res2.to(dev)
    _check(lambda h, **kw: ops.gemm(ap, pw2, c_rowmap=cmd, res=r2d, out=None if kw else torch.full((M, 2 * D), 7.0, device=dev),
                                    tile_hint=h, **kw), [sref2], E_C | E_RES | E_RMAP)


def test_s2_generic_epilogue_modes(dev):
    """what only the run-time epilogue serves: ragged N (N % 64 != 0), ReLU, plane residual, residual batch map, weight
    rows taken from activation planes"""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K = 1000, 132, 256
    a = torch.randn(M, K, generator=g)
    w, b, pw = _mk(g, N, K, dev)
    ap = ops.to_planes(a.to(dev))
    ref = a.double() @ w.double().t() + b.double()
    _check(lambda h, **kw: ops.gemm(ap, pw, tile_hint=h, **kw), [ref], E_GENERIC)
    w3, b3, pw3 = _mk(g, 192, K, dev)
    ref3 = a.double() @ w3.double().t() + b3.double()
    _check(lambda h, **kw: ops.gemm(ap, pw3, act=ops.ACT_RELU, tile_hint=h, **kw), [ref3.clamp(min=0)], E_GENERIC)
    rp = torch.randn(M, 192, generator=g)
    rpl = ops.to_planes(rp.to(dev))
    _check(lambda h, **kw: ops.gemm(ap, pw3, res=rpl, tile_hint=h, **kw), [ref3 + _pl64(rpl)], E_GENERIC)
    bmap = torch.tensor([1, 0, 1, 1, 0], dtype=torch.int32)
    r5 = torch.randn(2 * 200, 192, generator=g)
    bd, r5d = bmap.to(dev), r5.to(dev)
    rr = r5.view(2, 200, 192)[bmap.long()].reshape(1000, 192).double()
    _check(lambda h, **kw: ops.gemm(ap, pw3, res=r5d, res_bmap=bd, res_brows=200, tile_hint=h, **kw), [ref3 + rr], E_GENERIC)
    mf = torch.randn(700, K, generator=g)
    mfp = ops.to_planes(mf.to(dev))
    pwm = ops.PlaneWeight(mfp, r0=100, n=512)
    _check(lambda h, **kw: ops.gemm(ap, pwm, tile_hint=h, **kw), [a.double() @ _pl64(mfp)[100:612].t() * 1.0], E_C, tol=4e-6)


def test_s2_is_the_product_choice_for_the_encoder_shapes(dev):
    """rsp_gemm's own choice (tile hint 0) for the ViT shapes at two images: the s2 kernel with the specialisation the
    layer needs, or (round 5) the ping-pong kernel where its tiles fill the CUs -- and the result is the same bits as with
    the round-2 kernel"""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(5)
    T, D = 8192, 1280                   # two 1024-px images of ViT-H: every encoder Linear has >= 256 tiles of 256 x 128
    x = torch.randn(T, D, generator=g)
    xp = ops.to_planes(x.to(dev))
    xd = x.to(dev)
    sub = torch.arange(0, T, 17)        # fp64 reference on a row sample (the bit-equality below covers every row)
    w, b, wq = _mk(g, 3 * D, D, dev)
    assert ops.gemm(xp, wq, out_planes=True, c_ncols=D, pl_col0=D, plan_only=True) == E_C | E_PL
    w1, b1, pw1 = _mk(g, 4 * D, D, dev)
    assert (ops.gemm(xp, pw1, act=ops.ACT_GELU, out_planes=True, out_f32=False, plan_only=True) == E_PL | E_GELU or
            ops.gemm(xp, pw1, act=ops.ACT_GELU, out_planes=True, out_f32=False, plan_only='pp') in (128, 256))
    h0 = ops.gemm(xp, pw1, act=ops.ACT_GELU, out_planes=True, out_f32=False)
    assert _same(h0, ops.gemm(xp, pw1, act=ops.ACT_GELU, out_planes=True, out_f32=False, tile_hint=R2))
    assert _rel(h0, F.gelu(x[sub].double() @ w1.double().t() + b1.double()), rows=sub) < 2e-6
    w2, b2, pw2 = _mk(g, D, 4 * D, dev)
    assert ops.gemm(h0, pw2, res=xd, plan_only=True) == E_C | E_RES or ops.gemm(h0, pw2, res=xd, plan_only='pp') in (128, 256)
    y = ops.gemm(h0, pw2, res=xd)
    assert _same(y, ops.gemm(h0, pw2, res=xd, tile_hint=R2))
    assert _rel(y, _pl64(h0)[sub] @ w2.double().t() + b2.double() + x[sub].double(), rows=sub) < 2e-6
    # a small GEMM stays with the gemm_dma.hip tiles
    assert ops.gemm(ops.to_planes(x[:300].to(dev)), wq, plan_only=True) == -1


def test_s2_concurrent_streams_and_ticket_ring(dev):
    """the launch state of the persistent kernel (per-launch tile tickets from a per-device ring, one process-wide atomic
    slot counter): GEMMs overlapping on three streams give the same bits as run one after the other, and more launches
    than the ring has slots (1024) re-use re-armed tickets"""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(9)
    shapes = [(2048, 1280, 256), (4096, 768, 768), (1500, 3840, 128)]
    work = []
    for (M, N, K) in shapes:
        a = ops.to_planes(torch.randn(M, K, generator=g).to(dev))
        _, _, pw = _mk(g, N, K, dev)
        work.append((a, pw, ops.gemm(a, pw, tile_hint=S2)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in shapes]
    for rnd in range(6):
        outs = []
        for st, (a, pw, _) in zip(streams, work):
            with torch.cuda.stream(st):
                outs.append([ops.gemm(a, pw, tile_hint=S2) for _ in range(4)])
        torch.cuda.synchronize()
        for os_, (_, _, ref) in zip(outs, work):
            for o in os_:
                assert torch.equal(o, ref), rnd
    a, pw, ref = work[0]
    out = torch.empty_like(ref)
    for i in range(1100):
        ops.gemm(a, pw, out=out, tile_hint=S2)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
