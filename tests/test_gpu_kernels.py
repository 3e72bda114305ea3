"""-m gpu: every C-ABI kernel against a CPU fp64/fp32 restatement of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.quick]      # quick: the kernel-level tier (`-m "gpu and quick"`, < 2 min)


def _rel_err(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))


def test_gemm_plain_and_epilogues(dev):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in [(300, 200, 96), (128, 32, 64), (1000, 51, 1024), (77, 768, 768), (4100, 30, 256)]:
        a = torch.randn(M, K, generator=g) * 2.0
        w = torch.randn(N, K, generator=g) * 0.05      # asymmetric: catches row/col swaps
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        pw = ops.PackedWeight(w, b, device=dev)
        ref = a.double() @ w.double().t() + b.double()
        got = ops.gemm(a.to(dev), pw)
        assert _rel_err(got, ref) < 2e-6, (M, N, K)
        got = ops.gemm(a.to(dev), pw, act=ops.ACT_RELU, res=r.to(dev))
        assert _rel_err(got, ref.clamp(min=0) + r.double()) < 2e-6
        got = ops.gemm(a.to(dev), pw, act=ops.ACT_GELU)
        assert _rel_err(got, F.gelu(ref)) < 2e-6
        got = ops.gemm(a.to(dev), pw, bias=None)
        assert _rel_err(got, ref - b.double()) < 2e-6


def test_gemm_rowmaps_and_broadcast_residual(dev):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(1)
    M, N, K, R = 500, 96, 64, 300
    a = torch.randn(R, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.1
    pw = ops.PackedWeight(w, None, device=dev)
    amap = torch.randint(-1, R, (M,), generator=g, dtype=torch.int32)
    ref = torch.zeros(M, N, dtype=torch.float64)
    sel = amap >= 0
    ref[sel] = a[amap[sel].long()].double() @ w.double().t()
    got = ops.gemm(a.to(dev), pw, a_rowmap=amap.to(dev), M=M)
    assert _rel_err(got, ref) < 2e-6
    # scatter: permutation with holes
    perm = torch.randperm(M, generator=g).to(torch.int32)
    cmap = torch.where(torch.rand(M, generator=g) < 0.2, torch.full((M,), -1, dtype=torch.int32), perm)
    out = torch.full((M, N), 7.0, device=dev)
    a2 = torch.randn(M, K, generator=g)
    res = torch.randn(M, N, generator=g)
    ops.gemm(a2.to(dev), pw, out=out, c_rowmap=cmap.to(dev), res=res.to(dev))
    ref = torch.full((M, N), 7.0, dtype=torch.float64)
    full = a2.double() @ w.double().t()
    keep = cmap >= 0
    ref[cmap[keep].long()] = full[keep] + res.double()[cmap[keep].long()]
    assert _rel_err(out, ref) < 2e-6
    # broadcast residual (pos_embed): res row = crow % res_mod
    res2 = torch.randn(50, N, generator=g)
    got = ops.gemm(a2.to(dev), pw, res=res2.to(dev), res_mod=50)
    ref = full + res2.double()[torch.arange(M) % 50]
    assert _rel_err(got, ref) < 2e-6


@pytest.mark.parametrize('stride', [1, 2])
def test_gemm_conv3x3(dev, stride):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(2)
    B, H, W, C, O = 2, 14, 18, 64, 40
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(O, C, 3, 3, generator=g) * 0.05
    b = torch.randn(O, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
    pw = ops.PackedWeight(w.permute(0, 2, 3, 1).reshape(O, -1), b, device=dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.gemm(xh, pw, conv=(3, stride, 1))
    Ho, Wo = ref.shape[-2:]
    got = got.view(B, Ho, Wo, O).permute(0, 3, 1, 2)
    assert _rel_err(got, ref) < 2e-6


def test_gemm_small_values_precision(dev):
    """fp16 sub-normal range check: tiny activations must keep ~fp32 accuracy."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K = 256, 128, 512
    a = torch.randn(M, K, generator=g) * 1e-3
    w = torch.randn(N, K, generator=g) * 1e-3
    pw = ops.PackedWeight(w, None, device=dev)
    ref = a.double() @ w.double().t()
    for e in (0, 6, 12):
        got = ops.gemm(a.to(dev), pw, a_scale_log2=e)
        err = _rel_err(got, ref)
        print('small-value gemm a_scale_log2', e, 'rel err', err)
    assert _rel_err(ops.gemm(a.to(dev), pw, a_scale_log2=12), ref) < 5e-6


def test_layernorm(dev):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(4)
    for C in (32, 64, 256, 768, 1280):
        x = torch.randn(1001, C, generator=g) * 3 + 1
        w = torch.randn(C, generator=g)
        b = torch.randn(C, generator=g)
        ref = F.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-6)
        got = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6)
        assert float((got.cpu().double() - ref).abs().max()) < 2e-5
        got = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6, act=ops.ACT_GELU)
        assert float((got.cpu().double() - F.gelu(ref)).abs().max()) < 2e-5
    # plane outputs at every multiple of 128 up to 1408 (the eight-column kernel has five widths; 384 / 640 / 896 / 1152 once
    # ran its C = 1280 instantiation: ADVICE r5) and the widths only the other kernels take
    for C in (256, 320, 384, 512, 640, 768, 896, 1024, 1152, 1280, 1408):
        x = torch.randn(203, C, generator=g) * 3 + 1
        w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
        ref = F.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-6)
        y, pl = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6, planes=True)
        assert float((y.cpu().double() - ref).abs().max()) < 2e-5, C
        assert float((_planes_to_f32(pl).cpu().double() - ref).abs().max()) < 2e-5, C


def _ref_vit_attention(qkv, rph, rpw, S, nh, dh, scale):
    """fp64 restatement of HF:803-831 + 761-801 on a [Bp, T, 3, nh, dh] tensor."""
    Bp, T = qkv.shape[:2]
    q, k, v = qkv.double().permute(2, 0, 3, 1, 4).reshape(3, Bp * nh, T, dh).unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1)
    idx = torch.arange(S)[:, None] - torch.arange(S)[None, :] + (S - 1)
    Rh, Rw = rph.double()[idx], rpw.double()[idx]
    rq = q.reshape(Bp * nh, S, S, dh)
    rel_h = torch.einsum('bhwc,hkc->bhwk', rq, Rh)
    rel_w = torch.einsum('bhwc,wkc->bhwk', rq, Rw)
    attn = (attn.view(Bp * nh, S, S, S, S) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(Bp * nh, T, T)
    attn = attn.softmax(-1)
    out = (attn @ v).view(Bp, nh, S, S, dh).permute(0, 2, 3, 1, 4).reshape(Bp, T, nh * dh)
    rel = torch.cat([rel_h, rel_w], -1).reshape(Bp * nh, T, 2 * S)
    return out, rel


@pytest.mark.parametrize('S,nh,dh,Bp', [(14, 3, 64, 5), (64, 2, 64, 1), (14, 2, 80, 3), (64, 1, 80, 1), (32, 2, 64, 2), (32, 1, 80, 2), (64, 3, 64, 2)])
def test_vit_attention(dev, S, nh, dh, Bp):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(5)
    T = S * S
    qkv = torch.randn(Bp, T, 3, nh, dh, generator=g)
    qkv[:, :, 0] *= 2.0   # sharper softmax
    rph = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    rpw = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    scale = dh ** -0.5
    ref, ref_rel = _ref_vit_attention(qkv, rph, rpw, S, nh, dh, scale)
    d = qkv.to(dev).contiguous()
    rel = ops.vit_relpos(d, rph.to(dev), rpw.to(dev), Bp, S, nh, dh)
    assert float((rel.cpu().double() - ref_rel).abs().max()) < 2e-5
    out = ops.vit_attention(d, rel, Bp, S, nh, dh, scale).view(Bp, T, nh * dh)
    err = float((out.cpu().double() - ref).abs().max())
    vmax = float(qkv[:, :, 2].abs().max())
    print('vit_attention', S, nh, dh, 'max abs err', err, 'max|v|', vmax)
    assert err < 2e-5          # both products fp16x3: fp32-class


@pytest.mark.parametrize('S,nh,dh,Bp', [(14, 3, 64, 5), (64, 2, 64, 1), (14, 2, 80, 3), (64, 2, 80, 1), (32, 2, 64, 2),
                                        (32, 2, 80, 2), (64, 4, 64, 2), (14, 16, 80, 9)])
def test_vit_attention_planes(dev, S, nh, dh, Bp):
    """the plane-fed attention (csrc/attn_stream.hip): q fp32 [Bp*T, D], K | V as the KB32 fp16 planes of [Bp*T, 2D]
    (DMA key tiles, transposing LDS reads for V) against the fp64 restatement of HF:803-831."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(50 + S + dh)
    T, D = S * S, nh * dh
    qkv = torch.randn(Bp, T, 3, nh, dh, generator=g)
    qkv[:, :, 0] *= 2.0   # sharper softmax
    rph = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    rpw = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    scale = dh ** -0.5
    ref, ref_rel = _ref_vit_attention(qkv, rph, rpw, S, nh, dh, scale)
    q = qkv[:, :, 0].reshape(Bp * T, D).contiguous().to(dev)
    kv = ops.to_planes(qkv[:, :, 1:].reshape(Bp * T, 2 * D).contiguous().to(dev))
    rel = ops.vit_relpos(q, rph.to(dev), rpw.to(dev), Bp, S, nh, dh, q_ld=D)
    assert float((rel.cpu().double() - ref_rel).abs().max()) < 2e-5
    out = ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale).view(Bp, T, D)
    err = float((out.cpu().double() - ref).abs().max())
    print('vit_attention_planes', S, nh, dh, Bp, 'max abs err', err)
    assert err < 2e-5
    pl = ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale, planes=True)
    assert float((_planes_to_f32(pl).view(Bp, T, D) - ref).abs().max()) < 2e-5


@pytest.mark.parametrize('nw,real,nh,dh,B', [(5, 8, 2, 80, 1), (3, 4, 3, 64, 2), (2, 14, 2, 80, 1)])
def test_vit_attention_planes_skips_padded_queries(dev, nw, real, nh, dh, B):
    """rsp_vit_attention_planes_ex with the window grid known (HF:900-922: 64-grid -> 5 x 5 windows, 8 real rows / columns
    in the last ones; 32-grid -> 3 x 3, 4): the real tokens' outputs equal the plain kernel's / the fp64 restatement's,
    padded tokens still act as keys, their own outputs are left unwritten."""
    from rsprompter_amd import ops
    S = 14
    g = torch.Generator().manual_seed(700 + nw + dh)
    Bp, T, D = B * nw * nw, S * S, nh * dh
    qkv = torch.randn(Bp, T, 3, nh, dh, generator=g)
    qkv[:, :, 0] *= 2.0
    rph = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    rpw = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    scale = dh ** -0.5
    ref, _ = _ref_vit_attention(qkv, rph, rpw, S, nh, dh, scale)
    q = qkv[:, :, 0].reshape(Bp * T, D).contiguous().to(dev)
    kv = ops.to_planes(qkv[:, :, 1:].reshape(Bp * T, 2 * D).contiguous().to(dev))
    rel = ops.vit_relpos(q, rph.to(dev), rpw.to(dev), Bp, S, nh, dh, q_ld=D)
    full = ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale).view(Bp, S, S, D).cpu()
    got = ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale, win_grid=(nw, real))
    got = torch.nan_to_num(got, nan=7.0).view(Bp, S, S, D).cpu()          # unwritten rows hold whatever the allocator had
    wi = torch.arange(Bp) % (nw * nw)
    rh = torch.where(wi // nw == nw - 1, real, S)
    cw = torch.where(wi % nw == nw - 1, real, S)
    yy, xx = torch.arange(S)[None, :, None], torch.arange(S)[None, None, :]
    is_real = (yy < rh[:, None, None]) & (xx < cw[:, None, None])           # [Bp, S, S]
    assert int(is_real.sum()) == B * (S * (nw - 1) + real) ** 2
    assert torch.equal(got[is_real], full[is_real])                         # same arithmetic, only the query -> lane map differs
    assert float((got[is_real].double() - ref.view(Bp, S, S, D)[is_real]).abs().max()) < 2e-5
    pl = ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale, planes=True, win_grid=(nw, real))
    assert float((_planes_to_f32(pl).view(Bp, S, S, D)[is_real] - ref.view(Bp, S, S, D)[is_real]).abs().max()) < 2e-5
    # rel-pos terms for a row list (the real tokens, in any order): those rows equal the full evaluation bit for bit
    rows = is_real.reshape(-1).nonzero()[:, 0]
    rows = rows[torch.randperm(rows.numel(), generator=g)].to(torch.int32).to(dev)
    rel_rows = ops.vit_relpos(q, rph.to(dev), rpw.to(dev), Bp, S, nh, dh, q_ld=D, rows=rows)
    sel = is_real.reshape(Bp, T)[:, None, :].expand(Bp, nh, T).reshape(Bp * nh, T)
    assert torch.equal(torch.nan_to_num(rel_rows.cpu(), nan=3.0)[sel], rel.cpu()[sel])
    out2 = ops.vit_attention_planes(q, kv, rel_rows, Bp, S, nh, dh, scale, win_grid=(nw, real)).view(Bp, S, S, D).cpu()
    assert torch.equal(out2[is_real], full[is_real])


@pytest.mark.parametrize('variant', [0, 1])
@pytest.mark.parametrize('nw,real,nh,dh,B', [(5, 8, 2, 80, 1), (3, 4, 3, 64, 2), (2, 14, 16, 80, 1), (0, 0, 3, 64, 3)])
def test_vit_window_attention_fused_relpos(dev, nw, real, nh, dh, B, variant):
    """rsp_vit_window_attention (csrc/attn_win.hip): windowed SamVisionAttention with the decomposed rel-pos terms
    (HF:761-801) computed INSIDE the kernel from the packed tables, the bias added through the matrix cores, lazy online
    softmax -- against the fp64 restatement of HF:803-831; with the window grid known only the real tokens are queries.
    variant 1 = 16 persistent blocks, so that every block walks several windows (the K | V tile ring and the q requests
    cross window boundaries).  nw = 0: grid unknown, every query computed."""
    from rsprompter_amd import ops
    S = 14
    g = torch.Generator().manual_seed(900 + nw + dh)
    Bp, T, D = B * max(nw, 1) ** 2, S * S, nh * dh
    qkv = torch.randn(Bp, T, 3, nh, dh, generator=g)
    qkv[:, :, 0] *= 2.0                       # sharper softmax
    qkv[0, :, 1] *= 3.0                       # one window with a wide score range: the lazy maximum has to move there
    qkv[0, 150:, 1] *= 2.0
    rph = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    rpw = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    scale = dh ** -0.5
    ref, ref_rel = _ref_vit_attention(qkv, rph, rpw, S, nh, dh, scale)
    ref = ref.view(Bp, S, S, D)
    q = qkv[:, :, 0].reshape(Bp * T, D).contiguous().to(dev)
    kv = ops.to_planes(qkv[:, :, 1:].reshape(Bp * T, 2 * D).contiguous().to(dev))
    tab = ops.pack_relpos_tables(rph.to(dev), rpw.to(dev), S, dh)
    # the packed tables are the fp16 hi / lo split of table * 2^6, zero beyond the 27 rows / dh columns
    tb = tab.float().cpu()
    assert float((tb[0, 0] + tb[0, 1])[:27, :dh].sub(rph * 64).abs().max()) < 64 * 2.0 ** -20
    assert float((tb[1, 0] + tb[1, 1])[:27, :dh].sub(rpw * 64).abs().max()) < 64 * 2.0 ** -20
    assert float(tb[:, :, 27:].abs().max()) == 0.0 and float(tb[:, :, :, dh:].abs().max()) == 0.0
    wg = (nw, real) if nw else None
    got = ops.vit_window_attention(q, kv, tab, Bp, nh, dh, scale, win_grid=wg, variant=variant)
    got = torch.nan_to_num(got, nan=7.0).view(Bp, S, S, D).cpu()
    if nw:
        wi = torch.arange(Bp) % (nw * nw)
        rh = torch.where(wi // nw == nw - 1, real, S)
        cw = torch.where(wi % nw == nw - 1, real, S)
        yy, xx = torch.arange(S)[None, :, None], torch.arange(S)[None, None, :]
        is_real = (yy < rh[:, None, None]) & (xx < cw[:, None, None])
    else:
        is_real = torch.ones(Bp, S, S, dtype=torch.bool)
    err = float((got[is_real].double() - ref[is_real]).abs().max())
    print(f'vit_window_attention nw={nw} real={real} nh={nh} dh={dh} variant={variant}: max abs err {err:.2e}')
    assert err < 2e-5
    pl = ops.vit_window_attention(q, kv, tab, Bp, nh, dh, scale, planes=True, win_grid=wg, variant=variant)
    assert float((_planes_to_f32(pl).view(Bp, S, S, D)[is_real] - ref[is_real]).abs().max()) < 2e-5
    # same arithmetic as the rel-tensor form of the kernel up to the rounding of the rel-pos terms themselves
    rel = ops.vit_relpos(q, rph.to(dev), rpw.to(dev), Bp, S, nh, dh, q_ld=D)
    other = ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale, win_grid=wg).view(Bp, S, S, D).cpu()
    assert float((other[is_real] - got[is_real]).abs().max()) < 1e-5


def test_vit_window_attention_refuses_key_planes_with_a_large_exponent(dev):
    """the padded-key mask and the rel-pos bias share the score scale 2^(6 + kv exponent): beyond 4 the mask would stop
    underflowing the softmax, so both window entry points return RSP_EINVAL instead of a silently wrong answer."""
    from rsprompter_amd import ops
    S, nh, dh, Bp = 14, 2, 64, 1
    D = nh * dh
    g = torch.Generator().manual_seed(5)
    q = torch.randn(Bp * S * S, D, generator=g).to(dev)
    kvf = torch.randn(Bp * S * S, 2 * D, generator=g).to(dev)
    tab = ops.pack_relpos_tables((torch.randn(27, dh, generator=g) * 0.2).to(dev), (torch.randn(27, dh, generator=g) * 0.2).to(dev), S, dh)
    ok = ops.vit_window_attention(q, ops.to_planes(kvf, 4), tab, Bp, nh, dh, dh ** -0.5)
    ref = ops.vit_window_attention(q, ops.to_planes(kvf), tab, Bp, nh, dh, dh ** -0.5)
    assert float((ok - ref).abs().max()) < 2e-5
    with pytest.raises(RuntimeError, match='rsp_vit_window_attention'):
        ops.vit_window_attention(q, ops.to_planes(kvf, 8), tab, Bp, nh, dh, dh ** -0.5)


def test_gemm_column_range_outputs(dev):
    """rsp_gemm c_ncols / pl_col0 (the qkv projection's split hand-off): fp32 for the first D columns only, planes for
    the rest, with a row-gather map and padded rows like the windowed layers."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(9)
    M, K, D = 700, 256, 160
    a = torch.randn(500, K, generator=g)
    w = torch.randn(3 * D, K, generator=g) / K ** 0.5
    b = torch.randn(3 * D, generator=g)
    rowmap = torch.randint(-1, 500, (M,), generator=g, dtype=torch.int32)
    src = torch.where((rowmap >= 0)[:, None], a[rowmap.clamp(min=0).long()], torch.zeros(1))
    ref = src.double() @ w.double().t() + b.double()
    pw = ops.PackedWeight(w, b, device=dev)
    q, kv = ops.gemm(ops.to_planes(a.to(dev)), pw, a_rowmap=rowmap.to(dev), M=M, out_planes=True, c_ncols=D, pl_col0=D)
    assert tuple(q.shape) == (M, D) and kv.shape == (M, 2 * D)
    assert float((q.cpu().double() - ref[:, :D]).abs().max()) < 1e-5
    assert float((_planes_to_f32(kv) - ref[:, D:]).abs().max()) < 1e-5


def test_patchify_preprocess(dev):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(6)
    img = torch.randint(0, 256, (3, 60, 50), generator=g, dtype=torch.uint8)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    out = ops.preprocess([img.to(dev)], mean, std, swap_rb=True, pad_divisor=32)
    ref = torch.zeros(1, 3, 64, 64)
    rgb = img[[2, 1, 0]].float()
    ref[0, :, :60, :50] = (rgb - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    assert float((out.cpu() - ref).abs().max()) < 1e-5
    x = torch.randn(2, 3, 64, 32, generator=g)
    p = ops.patchify(x.to(dev), 16).cpu()
    ref = F.unfold(x, 16, stride=16).transpose(1, 2).reshape(-1, 768)
    assert torch.equal(p, ref)


def packed_like(w, dev):
    from rsprompter_amd.necks import convt_weights
    return convt_weights(w.to(dev), None)[0]


def test_conv_transpose_pool_add_sincos(dev):
    from rsprompter_amd import ops
    from rsprompter_amd.necks import convt_weights
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 6, 10, generator=g)
    w = torch.randn(64, 32, 2, 2, generator=g) * 0.1
    b = torch.randn(32, generator=g)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    packed, bias2 = convt_weights(w.to(dev), b.to(dev))
    got = ops.conv_transpose2x2(x.permute(0, 2, 3, 1).contiguous().to(dev), packed, bias2)
    assert _rel_err(got.permute(0, 3, 1, 2), ref) < 2e-6
    got = ops.conv_transpose2x2(x.permute(0, 2, 3, 1).contiguous().to(dev), packed, bias2, act=ops.ACT_GELU)
    assert _rel_err(got.permute(0, 3, 1, 2), F.gelu(ref)) < 2e-6
    # one-GEMM form (columns = (dy, dx, co)) through the plane path: fp32 out, plane out, fused hyper-network dot
    from rsprompter_amd.necks import convt_weights4
    x4 = torch.randn(3, 64, 12, 20, generator=g)
    w4 = torch.randn(64, 32, 2, 2, generator=g) * 0.1
    ref4 = F.conv_transpose2d(x4.double(), w4.double(), b.double(), stride=2)
    pk4, bias4 = convt_weights4(w4.to(dev), b.to(dev))
    x4h = x4.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.conv_transpose2x2(x4h, pk4, bias4)
    assert _rel_err(got.permute(0, 3, 1, 2), ref4) < 2e-6
    pl = ops.conv_transpose2x2(x4h, pk4, bias4, act=ops.ACT_GELU, out_planes=True)
    K = 32
    back = (pl.hi.float() + pl.lo.float()).permute(1, 0, 2).reshape(3, 24, 40, K) / 2.0 ** pl.scale_log2
    assert _rel_err(back.permute(0, 3, 1, 2), F.gelu(ref4)) < 2e-6
    hyp = torch.randn(3, 32, generator=g)
    refm = torch.einsum('bchw,bc->bhw', F.gelu(ref4), hyp.double())
    gotm = ops.conv_transpose2x2(x4h, pk4, bias4, act=ops.ACT_GELU, hyper=hyp.to(dev))
    assert _rel_err(gotm, refm) < 2e-6
    gotm2 = ops.conv_transpose2x2(x4h, packed_like(w4, dev), b.to(dev).repeat(2), act=ops.ACT_GELU, hyper=hyp.to(dev))
    assert _rel_err(gotm2, refm) < 2e-6
    # ConvTranspose + LayerNorm2d(64) + GELU -> planes (first half of the SAM upscaler)
    x5 = torch.randn(2, 96, 9, 16, generator=g)
    w5 = torch.randn(96, 64, 2, 2, generator=g) * 0.1
    b5 = torch.randn(64, generator=g)
    gam, bet = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    r5 = F.conv_transpose2d(x5.double(), w5.double(), b5.double(), stride=2).permute(0, 2, 3, 1)
    r5 = F.gelu(F.layer_norm(r5, (64,), gam.double(), bet.double(), 1e-6))
    pk5, bias5 = convt_weights4(w5.to(dev), b5.to(dev))
    pl = ops.conv_transpose2x2(x5.permute(0, 2, 3, 1).contiguous().to(dev), pk5, bias5, act=ops.ACT_GELU,
                               ln=(gam.to(dev), bet.to(dev), 1e-6))
    back = (pl.hi.float() + pl.lo.float()).permute(1, 0, 2).reshape(2, 18, 32, 64) / 2.0 ** pl.scale_log2
    assert _rel_err(back, r5) < 3e-6
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    assert torch.equal(ops.pool2(xh, 0).permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 2, 2))
    assert torch.equal(ops.pool2(xh, 1).permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 1, stride=2))
    v = torch.randn(5, 64, generator=g)
    a = torch.randn(20, 64, generator=g)
    assert torch.equal(ops.add_rows(a.to(dev), v.to(dev)).cpu(), a + v[torch.arange(20) % 5])
    s = torch.randn(7, 5, 512, generator=g)
    ref = torch.sin(s[..., ::2]) + s[..., 1::2]
    assert float((ops.sincos_pairs(s.to(dev)).cpu() - ref).abs().max()) < 1e-6
    sf = (2.0, 1.5, 2.0, 1.5)
    bx = torch.rand(9, 4, generator=g) * 100
    assert torch.equal(ops.div_boxes(bx.to(dev), sf).cpu(), bx / torch.tensor(sf))


@pytest.mark.parametrize('dh,Tq,Tk,nh', [(16, 10, 4096, 8), (16, 4096, 10, 8), (32, 10, 10, 8), (64, 70, 130, 2)])
def test_generic_attention_with_batch_maps(dev, dh, Tq, Tk, nh):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(8)
    R, Bk = 5, 2
    D = nh * dh
    q = torch.randn(R, Tq, D, generator=g)
    k = torch.randn(Bk, Tk, D, generator=g)
    v = torch.randn(Bk, Tk, D, generator=g)
    kvmap = torch.tensor([0, 0, 1, 1, 1], dtype=torch.int32)
    scale = dh ** -0.5
    qh = q.double().view(R, Tq, nh, dh).transpose(1, 2)
    kh = k.double()[kvmap.long()].view(R, Tk, nh, dh).transpose(1, 2)
    vh = v.double()[kvmap.long()].view(R, Tk, nh, dh).transpose(1, 2)
    ref = ((qh * scale) @ kh.transpose(-1, -2)).softmax(-1) @ vh
    ref = ref.transpose(1, 2).reshape(R, Tq, D)
    out = torch.empty(R, Tq, D, device=dev)
    ops.attention(q.to(dev), k.to(dev), v.to(dev), out, B=R, nh=nh, dh=dh, Tq=Tq, Tk=Tk, scale=scale,
                  q_strides=(Tq * D, D, dh), k_strides=(Tk * D, D, dh), v_strides=(Tk * D, D, dh),
                  o_strides=(Tq * D, D, dh), kv_batch_map=kvmap.to(dev))
    assert float((out.cpu().double() - ref).abs().max()) < 2e-5
    # q_batch_map: queries shared per image, keys per RoI
    q2 = torch.randn(Bk, Tq, D, generator=g)
    k2 = torch.randn(R, Tk, D, generator=g)
    v2 = torch.randn(R, Tk, D, generator=g)
    qh = q2.double()[kvmap.long()].view(R, Tq, nh, dh).transpose(1, 2)
    kh = k2.double().view(R, Tk, nh, dh).transpose(1, 2)
    vh = v2.double().view(R, Tk, nh, dh).transpose(1, 2)
    ref = (((qh * scale) @ kh.transpose(-1, -2)).softmax(-1) @ vh).transpose(1, 2).reshape(R, Tq, D)
    ops.attention(q2.to(dev), k2.to(dev), v2.to(dev), out, B=R, nh=nh, dh=dh, Tq=Tq, Tk=Tk, scale=scale,
                  q_strides=(Tq * D, D, dh), k_strides=(Tk * D, D, dh), v_strides=(Tk * D, D, dh),
                  o_strides=(Tq * D, D, dh), q_batch_map=kvmap.to(dev))
    assert float((out.cpu().double() - ref).abs().max()) < 2e-5


def test_roi_align_matches_oracle(dev):
    from oracle import cops, glue
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(9)
    strides = [4, 8, 16, 32]
    feats = [torch.randn(2, 256, 256 // s * 4, 256 // s * 4, generator=g) for s in strides]  # 1024-px image
    pes = [torch.randn(f.shape[2], f.shape[3], 256, generator=g) for f in feats]
    K = 300
    xy = torch.rand(K, 2, generator=g) * 900
    wh = torch.exp(torch.rand(K, 2, generator=g) * 6.5)          # 1 .. 665 px, spans all 4 levels
    rois = torch.cat([torch.randint(0, 2, (K, 1), generator=g).float(), xy, xy + wh], 1)
    rois[0, 1:] = torch.tensor([-50., -20., 30., 10.])           # partly outside
    rois[1, 1:] = torch.tensor([10., 10., 10., 10.])             # degenerate
    rois[2, 1:] = torch.tensor([1000., 1000., 1100., 1090.])     # beyond the border
    for P in (7, 14):
        ref = glue.roi_extract([f + pe.permute(2, 0, 1)[None] for f, pe in zip(feats, pes)], rois, P, strides)
        got = ops.roi_align([f.permute(0, 2, 3, 1).contiguous().to(dev) for f in feats],
                            [p.to(dev) for p in pes], rois.to(dev), P, strides)
        err = float((got.permute(0, 3, 1, 2).cpu() - ref).abs().max())
        assert err < 1e-4, err
    assert ops.roi_align([f.permute(0, 2, 3, 1).contiguous().to(dev) for f in feats], None,
                         torch.zeros((0, 5), device=dev), 7, strides).shape == (0, 7, 7, 256)


def _rand_boxes(n, g, size=1024.):
    xy = torch.rand(n, 2, generator=g) * size * 0.8
    wh = torch.rand(n, 2, generator=g) * size * 0.3 + 1
    return torch.cat([xy, (xy + wh).clamp(max=size)], 1)


@pytest.mark.parametrize('n,nid,thr,max_out', [(5000, 5, 0.7, 1000), (10000, 10, 0.5, 100), (37, 1, 0.5, 100),
                                                (0, 1, 0.5, 10), (16384, 16, 0.5, 100), (20000, 2, 0.3, 20000),
                                                (40000, 80, 0.6, 40000)])
def test_batched_nms_matches_oracle(dev, n, nid, thr, max_out):
    """mmcv batched_nms (coordinate-offset form; per-id loop + stable re-sort above its split_thr of 10000 boxes) with many
    exact score ties.  Up to 16384 candidates sort in LDS, larger sets in memory (det.hip NMS_LDS_KEYS) with the
    32-word removal registers -- multiclass_nms of many-class heads (bbox_nms.py:12-105)."""
    from oracle import glue
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(10 + n)
    Bn, cap = 2, max(n, 1)
    boxes = torch.zeros(Bn, cap, 4)
    scores = torch.zeros(Bn, cap)
    ids = torch.zeros(Bn, cap, dtype=torch.int32)
    cnt = torch.tensor([n, max(n - 3, 0)], dtype=torch.int32)
    for b in range(Bn):
        boxes[b] = _rand_boxes(cap, g)
        scores[b] = (torch.rand(cap, generator=g) * 50).round() / 50        # many exact ties
        ids[b] = torch.randint(0, nid, (cap,), generator=g, dtype=torch.int32)
    cand = (boxes.to(dev), scores.to(dev), ids.to(dev), torch.arange(cap, dtype=torch.int32).repeat(Bn, 1).to(dev),
            cnt.to(dev))
    out = ops.batched_nms(cand, Bn, cap, thr, max_out)
    for b in range(Bn):
        m = int(cnt[b])
        dets, keep = glue.batched_nms(boxes[b, :m], scores[b, :m], ids[b, :m].long(), thr)
        keep = keep[:max_out]
        k = int(out['count'][b])
        assert k == keep.numel()
        assert torch.equal(out['keep'][b, :k].cpu().long(), keep)
        assert torch.equal(out['boxes'][b, :k].cpu(), boxes[b, :m][keep])


def test_batched_nms_refuses_a_workspace_beyond_the_limit(dev, monkeypatch):
    """the in-memory NMS path needs B * cap^2 / 8 bytes of pair mask: a call beyond ops.NMS_WORKSPACE_LIMIT_BYTES fails with
    the figures in the message, before anything is allocated or launched."""
    from rsprompter_amd import ops
    cap = 40000
    cand = (torch.zeros(1, cap, 4, device=dev), torch.zeros(1, cap, device=dev), torch.zeros(1, cap, dtype=torch.int32, device=dev),
            torch.zeros(1, cap, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
    monkeypatch.setattr(ops, 'NMS_WORKSPACE_LIMIT_BYTES', 64 << 20)
    with pytest.raises(ValueError, match='GiB pair mask'):
        ops.batched_nms(cand, 1, cap, 0.5, 100)
    monkeypatch.setattr(ops, 'NMS_WORKSPACE_LIMIT_BYTES', 8 << 30)
    assert int(ops.batched_nms(cand, 1, cap, 0.5, 100)['count'][0]) == 0


def test_rpn_topk_ties_and_small_levels(dev):
    """rpn_head.py:198-212: stable descending sort, top nms_pre; levels with n <= nms_pre keep natural order."""
    from rsprompter_amd import _lib, ops
    import ctypes
    g = torch.Generator().manual_seed(11)
    Bn, A, LD, k = 2, 6, 32, 1000
    sizes = [(64, 64), (16, 16), (8, 8), (4, 4)]
    heads = []
    for li, (H, W) in enumerate(sizes):
        h = torch.randn(Bn * H * W, LD, generator=g)
        if li == 0:
            h[:, :A] = (h[:, :A] * 4).round() / 4          # heavy ties
        if li == 1:
            h[:, :A] = 0.25                                 # constant input: pure index order
        heads.append(h)
    lib = _lib.load()
    d = _lib.RspRpnDesc()
    dheads = [h.to(dev) for h in heads]
    for i, (hd, (H, W)) in enumerate(zip(dheads, sizes)):
        d.head[i] = hd.data_ptr(); d.H[i], d.W[i], d.stride[i] = H, W, 4.0 * 2 ** i
    d.ld, d.A, d.nms_pre, d.num_levels = LD, A, k, len(sizes)
    L = len(sizes)
    sel_idx = torch.full((Bn, L, k), -7, dtype=torch.int32, device=dev)
    sel_score = torch.zeros((Bn, L, k), device=dev)
    sel_cnt = torch.zeros((Bn, L), dtype=torch.int32, device=dev)
    _lib.check(lib.rsp_rpn_topk(d, Bn, sel_idx.data_ptr(), sel_score.data_ptr(), sel_cnt.data_ptr(),
                                ops._stream()), 'topk')
    for b in range(Bn):
        for li, (H, W) in enumerate(sizes):
            n = H * W * A
            logits = heads[li].view(Bn, H * W, LD)[b, :, :A].reshape(-1)
            sc = logits.sigmoid()
            cnt = int(sel_cnt[b, li])
            if n > k:
                ranked, inds = sc.sort(descending=True, stable=True)
                assert cnt == k
                assert torch.equal(sel_idx[b, li, :k].cpu().long(), inds[:k]), (b, li)
                assert float((sel_score[b, li, :k].cpu() - ranked[:k]).abs().max()) < 1e-6
            else:
                assert cnt == n
                assert torch.equal(sel_idx[b, li, :n].cpu().long(), torch.arange(n))


def test_mask_post_matches_reference_formula(dev):
    from oracle import glue
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(12)
    low = torch.randn(3, 1, 256, 256, generator=g) * 3
    for meta in (dict(ori_shape=(1024, 1024), scale_factor=(1.0, 1.0), batch_input_shape=(1024, 1024)),
                 dict(ori_shape=(512, 512), scale_factor=(2.0, 2.0), batch_input_shape=(1024, 1024)),
                 dict(ori_shape=(600, 400), scale_factor=(1.5, 1.5), batch_input_shape=(1024, 1024))):
        boxes = torch.rand(3, 4, generator=g) * 100
        ref_mask, _, ref_prob = glue.mask_postprocess_single(low, boxes.clone(), meta, 0.5, True)
        sf_w, sf_h = meta['scale_factor']
        h, w = meta['ori_shape']
        crop = (min(int(h * sf_h), 1024), min(int(w * sf_w), 1024))
        got, prob = ops.mask_post(low[:, 0].contiguous().to(dev), (1024, 1024), crop, (h, w), 0.5, want_prob=True)
        assert float((prob.cpu() - ref_prob).abs().max()) < 2e-6
        assert float((got.cpu() != ref_mask).float().mean()) < 1e-5
    # mask_thr_binary < 0 (models.py:1779-1783, "for visualization and debugging"): uint8 soft masks (p * 255 truncated);
    # a probability 2e-6 from the oracle's may truncate one count lower / higher where p * 255 is within 5e-4 of an integer
    from rsprompter_amd.anchor_heads import RSPrompterAnchorMaskHead
    from rsprompter_amd.structures import InstanceData
    r = InstanceData()
    r.bboxes = boxes.clone().to(dev)
    soft = RSPrompterAnchorMaskHead._predict_by_feat_single(None, low.to(dev), r, meta, dict(mask_thr_binary=-1), rescale=True)
    ref_soft = (ref_prob * 255).to(torch.uint8)
    assert soft.dtype == torch.uint8 and soft.shape == ref_soft.shape
    d = (soft.cpu().int() - ref_soft.int()).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-4


def test_hyper_mask(dev):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(13)
    up = torch.randn(3, 1000, 32, generator=g)
    hy = torch.randn(3, 32, generator=g)
    ref = torch.einsum('rpc,rc->rp', up.double(), hy.double())
    assert float((ops.hyper_mask(up.to(dev), hy.to(dev)).cpu().double() - ref).abs().max()) < 1e-4


def _planes_to_f32(p):
    """KB32 planes [K/32][rows][32] -> fp64 [rows, K]"""
    v = (p.hi.float() + p.lo.float()).cpu().double() / 2.0 ** p.scale_log2
    kb, rows, _ = v.shape
    return v.permute(1, 0, 2).reshape(rows, kb * 32).reshape(p.shape)


def test_plane_path_gemm_layernorm_attention(dev):
    """fp16 (hi, lo) planes: DMA GEMM (plain / gather / implicit conv / plane output), LN and attention emitters."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(21)
    for (M, N, K) in [(300, 200, 96), (1000, 768, 768), (130, 40, 64), (257, 30, 256), (515, 160, 128)]:
        a = torch.randn(M, K, generator=g) * 2
        w = torch.randn(N, K, generator=g) * 0.05
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        pw = ops.PackedWeight(w, b, device=dev)
        ref = F.gelu(a.double() @ w.double().t() + b.double()) + r.double()
        ap = ops.to_planes(a.to(dev))
        assert float((_planes_to_f32(ap) - a.double()).abs().max()) < 1e-6
        if N % 32 == 0:
            out, pl = ops.gemm(ap, pw, act=ops.ACT_GELU, res=r.to(dev), out_planes=True)
            assert float((_planes_to_f32(pl) - ref).abs().max() / ref.abs().max()) < 2e-6
        else:
            out = ops.gemm(ap, pw, act=ops.ACT_GELU, res=r.to(dev))
        assert _rel_err(out, ref) < 2e-6
    # gather + scatter maps on the DMA path
    R, M, N, K = 300, 500, 160, 128
    a = torch.randn(R, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.1
    pw = ops.PackedWeight(w, None, device=dev)
    amap = torch.randint(-1, R, (M,), generator=g, dtype=torch.int32)
    ref = torch.zeros(M, N, dtype=torch.float64)
    sel = amap >= 0
    ref[sel] = a[amap[sel].long()].double() @ w.double().t()
    got = ops.gemm(ops.to_planes(a.to(dev)), pw, a_rowmap=amap.to(dev), M=M)
    assert _rel_err(got, ref) < 2e-6
    # implicit 3x3 conv from planes
    for stride in (1, 2):
        x = torch.randn(2, 64, 14, 18, generator=g)
        cw = torch.randn(140, 64, 3, 3, generator=g) * 0.05
        ref = F.conv2d(x.double(), cw.double(), None, stride=stride, padding=1)
        pw = ops.PackedWeight(cw.permute(0, 2, 3, 1).reshape(140, -1), None, device=dev)
        xp = ops.to_planes(x.permute(0, 2, 3, 1).contiguous().to(dev))
        got = ops.gemm(xp, pw, conv=(3, stride, 1))
        Ho, Wo = ref.shape[-2:]
        assert _rel_err(got.view(2, Ho, Wo, 140).permute(0, 3, 1, 2), ref) < 2e-6
    # LayerNorm emitting planes (all C classes incl. the small-C kernel)
    for C in (32, 64, 256, 768):
        x = torch.randn(777, C, generator=g) * 3 + 1
        wt = torch.randn(C, generator=g); bs = torch.randn(C, generator=g)
        ref = F.layer_norm(x.double(), (C,), wt.double(), bs.double(), 1e-6)
        y, pl = ops.layernorm(x.to(dev), wt.to(dev), bs.to(dev), 1e-6, planes=True)
        assert float((y.cpu().double() - ref).abs().max()) < 2e-5
        assert float((_planes_to_f32(pl) - ref).abs().max()) < 2e-5
    # attention emitting planes
    S, nh, dh, Bp = 14, 2, 64, 3
    T = S * S
    qkv = torch.randn(Bp, T, 3, nh, dh, generator=g)
    rph = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    rpw = torch.randn(2 * S - 1, dh, generator=g) * 0.2
    ref, _ = _ref_vit_attention(qkv, rph, rpw, S, nh, dh, dh ** -0.5)
    d = qkv.to(dev).contiguous()
    rel = ops.vit_relpos(d, rph.to(dev), rpw.to(dev), Bp, S, nh, dh)
    pl = ops.vit_attention(d, rel, Bp, S, nh, dh, dh ** -0.5, planes=True)
    assert float((_planes_to_f32(pl).view(Bp, T, nh * dh) - ref).abs().max()) < 2e-5


def test_outlier_activations_do_not_poison_the_split(dev):
    """real SAM checkpoints carry outlier activations (MLP hidden layer, residual stream).  The fp16 split
    must neither overflow to inf (NaN rows) nor lose the outliers: |x| up to 1e4 is inside the exact range of the
    default pre-scale, beyond 16376 the split saturates gracefully (finite, bounded error) instead of producing inf."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(77)
    M, K, N = 512, 256, 192
    a = torch.randn(M, K, generator=g)
    a[torch.rand(M, K, generator=g) < 0.002] *= 3000.0          # |x| ~ 1e3 - 1e4 outliers in ~0.2 % of the entries
    a[5, 7] = 9.9e3
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    pw = ops.PackedWeight(w, b, device=dev)
    for inp in (ops.to_planes(a.to(dev)), a.to(dev)):                           # DMA plane path and fp32-A path
        got = ops.gemm(inp, pw).cpu().double()
        assert bool(torch.isfinite(got).all())
        # error model of the split: every operand carries ~22 bits (hi + lo; the a_lo b_lo term is dropped), so
        # |err| <= 2^-19 sum_k |a_k| |w_k| with margin (measured: 2.4 x 2^-21)
        bound = (a.double().abs() @ w.double().abs().t()) * 2.0 ** -19 + 1e-6
        assert bool(((got - ref).abs() <= bound).all()), float(((got - ref).abs() / bound).max())
    # LayerNorm output planes of a row with an outlier, and a residual-stream tensor far beyond the range
    x = torch.randn(64, 256, generator=g)
    x[3, 9] = 5.0e3
    wt, bs = torch.ones(256), torch.zeros(256)
    y, pl = ops.layernorm(x.to(dev), wt.to(dev), bs.to(dev), 1e-6, planes=True)
    assert float((_planes_to_f32(pl) - y.cpu().double()).abs().max()) < 1e-5
    big = torch.randn(64, 256, generator=g)
    big[0, 0], big[1, 1] = 3.0e4, -1.0e6                                          # beyond the exact range: saturate
    p2 = ops.to_planes(big.to(dev))
    back = _planes_to_f32(p2)
    assert bool(torch.isfinite(back).all())
    assert abs(float(back[0, 0]) - 3.0e4) < 3.0e4 * 2.0 ** -10 and float(back[1, 1]) < -3.2e4
    mask = torch.ones_like(big, dtype=torch.bool)
    mask[0, 0] = mask[1, 1] = False
    assert float((back - big.double()).abs()[mask].max()) < 1e-5


@pytest.mark.parametrize('hint', [0, 1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 17, 18, 19, 20])
def test_plane_gemm_tile_variants(dev, hint):
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(30 + hint)
    for (M, N, K) in [(1000, 700, 256), (513, 257, 96), (300, 3072, 64), (700, 300, 32)]:
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) * 0.05
        b = torch.randn(N, generator=g)
        pw = ops.PackedWeight(w, b, device=dev)
        ref = a.double() @ w.double().t() + b.double()
        got = ops.gemm(ops.to_planes(a.to(dev)), pw, tile_hint=hint)
        assert _rel_err(got, ref) < 2e-6, (hint, M, N, K)
    x = torch.randn(2, 64, 20, 24, generator=g)
    cw = torch.randn(300, 64, 3, 3, generator=g) * 0.05
    ref = F.conv2d(x.double(), cw.double(), None, stride=1, padding=1)
    pw = ops.PackedWeight(cw.permute(0, 2, 3, 1).reshape(300, -1), None, device=dev)
    got = ops.gemm(ops.to_planes(x.permute(0, 2, 3, 1).contiguous().to(dev)), pw, conv=(3, 1, 1), tile_hint=hint)
    assert _rel_err(got.view(2, 20, 24, 300).permute(0, 3, 1, 2), ref) < 2e-6


@pytest.mark.parametrize('T', [7, 10, 12])
def test_sam_cross_attention_kernels(dev, T):
    """rsp_sam_t2i_attention / rsp_sam_i2t_attention against fp64 softmax attention (HF:243-288 semantics)."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(50 + T)
    R, Rimg, N = 5, 2, 1000 if T != 10 else 4096
    scale = 16 ** -0.5
    q = torch.randn(R, T, 128, generator=g)
    kv = torch.randn(Rimg, N, 256, generator=g)
    kv[..., :128] *= 2.0          # spread the scores so that the online softmax rescaling is exercised
    mp = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32)
    kk = kv[mp.long()][..., :128].view(R, N, 8, 16).permute(0, 2, 1, 3).double()
    vv = kv[mp.long()][..., 128:].view(R, N, 8, 16).permute(0, 2, 1, 3).double()
    qq = q.view(R, T, 8, 16).permute(0, 2, 1, 3).double()
    ref = ((qq * scale) @ kk.transpose(-1, -2)).softmax(-1) @ vv
    ref = ref.permute(0, 2, 1, 3).reshape(R, T, 128)
    out = torch.empty(R * T, 128, device=dev)
    ops.sam_t2i_attention(q.view(R * T, 128).to(dev), kv.view(Rimg * N, 256).to(dev), out, R=R, T=T, N=N, scale=scale,
                          kv_map=mp.to(dev))
    assert float((out.cpu().view(R, T, 128) - ref).abs().max()) < 2e-6
    # image -> token
    qi = torch.randn(Rimg, N, 128, generator=g) * 2.0
    kt, vt = torch.randn(R, T, 128, generator=g), torch.randn(R, T, 128, generator=g)
    qq = qi[mp.long()].view(R, N, 8, 16).permute(0, 2, 1, 3).double()
    kk = kt.view(R, T, 8, 16).permute(0, 2, 1, 3).double()
    vv = vt.view(R, T, 8, 16).permute(0, 2, 1, 3).double()
    ref = (((qq * scale) @ kk.transpose(-1, -2)).softmax(-1) @ vv).permute(0, 2, 1, 3).reshape(R * N, 128)
    o32 = torch.empty(R * N, 128, device=dev)
    pl = ops.empty_planes((R * N, 128), dev)
    ops.sam_i2t_attention(qi.view(Rimg * N, 128).to(dev), kt.view(R * T, 128).to(dev), vt.view(R * T, 128).to(dev),
                          R=R, T=T, N=N, scale=scale, q_map=mp.to(dev), out=o32, out_planes=pl)
    assert float((o32.cpu() - ref).abs().max()) < 2e-6
    assert float((_planes_to_f32(pl) - ref).abs().max()) < 4e-6


def test_gemm_plane_residual(dev):
    """residual handed over as fp16 planes (hi + lo) instead of fp32, incl. the RoI -> image row map"""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(91)
    M, K, N = 3 * 200, 128, 256
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    res = torch.randn(2 * 200, N, generator=g) * 3
    mp = torch.tensor([1, 0, 1], dtype=torch.int32)
    ref = a.double() @ w.double().t() + b.double() + res.view(2, 200, N)[mp.long()].reshape(M, N).double()
    pw = ops.PackedWeight(w, b, device=dev)
    out = ops.gemm(ops.to_planes(a.to(dev)), pw, res=ops.to_planes(res.to(dev)), res_bmap=mp.to(dev), res_brows=200)
    assert _rel_err(out, ref) < 2e-6


def _emu_f8corr(a, w, ea, ew):
    """CPU emulation of the fp8-corrected product (include/rsp_hip.h "Plane format word"): fp16 hi . hi plus the two
    cross terms with e4m3 operands under the static storage scales 2^5 (lo) / 2^-7 (hi); fp64 accumulation."""
    def parts(x, e):
        xs = x.double() * 2.0 ** e
        hi = xs.float().clamp(-65504, 65504).half().double()
        lo = xs - hi
        lo8 = (lo * 32).float().clamp(-448, 448).to(torch.float8_e4m3fn).double() / 32
        hi8 = (hi / 128).float().clamp(-448, 448).to(torch.float8_e4m3fn).double() * 128
        return hi, lo8, hi8
    ah, al8, ah8 = parts(a, ea)
    wh, wl8, wh8 = parts(w, ew)
    return (ah @ wh.t() + al8 @ wh8.t() + ah8 @ wl8.t()) * 2.0 ** -(ea + ew)


@pytest.mark.parametrize('hint', [14, 18, 17])
def test_gemm_fp8_corrected_product(dev, hint):
    """RSP_PLANE_F8 GEMM (fp16 hi.hi + one K = 64 fp8 MFMA for both cross terms) on the three tiles it is built for:
    the kernel against a CPU emulation of the same arithmetic (data path: cat8 layout, lane halves, block scales) and
    against the exact product (error class 2^-15 of sum |a||w|)."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(21 + hint)
    M, N, K = 600, 512, 1280
    a = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))       # rows of different magnitude
    a[5, 7] = 900.0                                                                       # an outlier activation
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    pw = ops.PackedWeight(w, b, device=dev, f8=True)
    pa = ops.to_planes(a.to(dev), f8=True)
    assert pa.f8 and pw.f8
    out = ops.gemm(pa, pw, tile_hint=hint).cpu().double()
    exact = a.double() @ w.double().t() + b.double()
    emu = _emu_f8corr(a, w, pa.scale_log2, pw.scale_log2) + b.double()
    mag = (a.double().abs() @ w.double().abs().t())
    e_emu = float(((out - emu).abs() / mag).max())
    e_exact = float(((out - exact).abs() / mag).max())
    print(f'tile hint {hint}: vs emulation {e_emu:.2e}, vs exact {e_exact:.2e} (relative to sum |a||w|)')
    assert e_emu < 2e-6            # fp32 accumulation noise only
    assert e_exact < 2.0 ** -13    # 2^-15 class; worst element of 300 k
    with pytest.raises(ValueError):
        ops.gemm(ops.to_planes(a.to(dev)), pw)            # plain planes against an f8 weight


def test_gemm_fp8_chain_with_rowmap_and_f8_output(dev):
    """two chained f8 GEMMs like lin1 -> GELU -> lin2: the first reads a row-gathered A and writes cat8 planes from
    its epilogue, the second consumes them; LayerNorm's cat8 output feeds the first."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(5)
    rows, M, K, H = 300, 400, 256, 512
    x = torch.randn(rows, K, generator=g) * 3 + 0.5
    gamma, beta = torch.randn(K, generator=g), torch.randn(K, generator=g)
    w1, b1 = torch.randn(H, K, generator=g) / K ** 0.5, torch.randn(H, generator=g)
    w2, b2 = torch.randn(K, H, generator=g) / H ** 0.5, torch.randn(K, generator=g)
    rowmap = torch.randint(-1, rows, (M,), generator=g, dtype=torch.int32)
    xn = F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-6)
    src = torch.where((rowmap >= 0)[:, None], xn[rowmap.clamp(min=0).long()], torch.zeros(1, dtype=torch.float64))
    h = F.gelu(src @ w1.double().t() + b1.double())
    ref = h @ w2.double().t() + b2.double()
    pl = ops.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), 1e-6, planes=True, f32=False, f8=True)
    hm = ops.gemm(pl, ops.PackedWeight(w1, b1, device=dev, f8=True), a_rowmap=rowmap.to(dev), M=M, act=ops.ACT_GELU,
                  out_planes=True, out_f32=False, out_f8=True)
    assert hm.f8
    out = ops.gemm(hm, ops.PackedWeight(w2, b2, device=dev, f8=True)).cpu().double()
    e = float((out - ref).abs().max())
    print(f'f8 chain LN -> lin1 -> GELU -> lin2: max err {e:.2e} (range {float(ref.abs().max()):.1f})')
    assert e < 5e-4


@pytest.mark.parametrize('form', ['valu', 'mfma'])
@pytest.mark.parametrize('T,N,planes_res', [(10, 600, False), (10, 1024, True), (7, 520, True), (3, 64, False)])
def test_sam_i2t_fused_matches_composition(dev, T, N, planes_res, form):
    """rsp_sam_i2t_fused = LayerNorm(residual + out_proj(image -> token attention)) (HF:340-348) against the fp64
    composition of the plain pieces: per-image queries / residual through RoI maps (layer 0) and per-RoI plane
    residual (layer 1); N not a multiple of the block's 512 positions; T on both kernel instantiations."""
    from rsprompter_amd import ops
    # form 'valu': the request for an fp32 copy of the result is served by the VALU form of the kernel (csrc/samattn.hip)
    g = torch.Generator().manual_seed(100 + T)
    R, B = 5, 2
    roi_img = torch.tensor([0, 0, 1, 1, 1], dtype=torch.int32)
    scale = 16 ** -0.5
    k, v = torch.randn(R * T, 128, generator=g), torch.randn(R * T, 128, generator=g)
    wo, bo = torch.randn(256, 128, generator=g) / 128 ** 0.5, torch.randn(256, generator=g)
    gamma, beta = torch.randn(256, generator=g), torch.randn(256, generator=g)
    if planes_res:
        q = torch.randn(R * N, 128, generator=g) * 2
        res = torch.randn(R * N, 256, generator=g) * 3
        qq, rr = q.view(R, N, 128), res.view(R, N, 256)
    else:
        q = torch.randn(B * N, 128, generator=g) * 2
        res = torch.randn(B * N, 256, generator=g) * 3
        qq, rr = q.view(B, N, 128)[roi_img.long()], res.view(B, N, 256)[roi_img.long()]
    qh = qq.double().view(R, N, 8, 16).permute(0, 2, 1, 3)
    kh = k.double().view(R, T, 8, 16).permute(0, 2, 1, 3)
    vh = v.double().view(R, T, 8, 16).permute(0, 2, 1, 3)
    att = ((qh * scale) @ kh.transpose(-1, -2)).softmax(-1) @ vh
    y = att.permute(0, 2, 1, 3).reshape(R, N, 128) @ wo.double().t() + bo.double() + rr.double()
    ref = F.layer_norm(y, (256,), gamma.double(), beta.double(), 1e-6).reshape(R * N, 256)
    # the matrix-core form writes planes only (a request for the fp32 copy is served by the VALU form)
    kw = dict(R=R, T=T, N=N, scale=scale, eps=1e-6, planes=True, f32=(form == 'valu'))
    args = [t.to(dev) for t in (q, k, v, wo, bo, gamma, beta)]
    if planes_res:
        got = ops.sam_i2t_fused(*args, res_planes=ops.to_planes(res.to(dev)), **kw)
    else:
        got = ops.sam_i2t_fused(*args, q_map=roi_img.to(dev), res=res.to(dev), res_map=roi_img.to(dev), **kw)
    out, pl = got if form == 'valu' else (None, got)
    e32 = float((out.cpu().double() - ref).abs().max()) if out is not None else 0.0
    epl = float((_planes_to_f32(pl) - ref).abs().max())
    print(f'sam_i2t_fused[{form}] T={T} N={N} planes_res={planes_res}: fp32 err {e32:.2e}, planes err {epl:.2e}')
    assert e32 < 2e-5 and epl < 2e-5
    with pytest.raises(RuntimeError):
        ops.sam_i2t_fused(*[t.to(dev) for t in (q, torch.randn(R * 11, 128), torch.randn(R * 11, 128), wo, bo, gamma, beta)],
                          R=R, T=11, N=N, scale=scale, res=torch.zeros(R * N, 256, device=dev))


@pytest.mark.parametrize('R,N,T', [(3, 64, 10), (2, 256, 7), (1, 96, 12), (2, 64, 3)])
def test_sam_t2i_fold_matches_fp64_attention(dev, R, N, T):
    """token -> image attention with the K | V projections of the per-RoI keys folded into the kernel
    (csrc/t2i_fold.hip; SamMaskDecoderHIP._t2i_folded): q' = Wk_h^T tq, scores over the key planes + the PEK term, softmax,
    sum_n p keys[n], v_proj afterwards -- against the fp64 statement of HF:326-331 (k = k_proj(keys + pe), v = v_proj(keys),
    8 heads x 16, scale 16^-0.5) and against the unfolded kernels on the same planes."""
    from rsprompter_amd import ops
    from rsprompter_amd.sam_decoder import SamMaskDecoderHIP
    from rsprompter_amd.synth import synth_state_dict
    dec = SamMaskDecoderHIP()
    dec.load_state_dict(synth_state_dict(dec, 21))
    dec = dec.to(dev)
    dec._pack()
    g = torch.Generator().manual_seed(R * 1000 + N + T)
    keys = torch.randn(R * N, 256, generator=g) * 1.5
    pe = torch.randn(N, 256, generator=g)
    tq = torch.randn(R * T, 128, generator=g) * 2.0
    at = dec.transformer.final_attn_token_to_image
    Wk, bk, Wv, bv = (t.detach().double().cpu() for t in (at.k_proj.weight, at.k_proj.bias, at.v_proj.weight, at.v_proj.bias))
    K = ((keys.view(R, N, 256) + pe[None]).double() @ Wk.t() + bk).view(R, N, 8, 16).permute(0, 2, 1, 3)
    V = (keys.view(R, N, 256).double() @ Wv.t() + bv).view(R, N, 8, 16).permute(0, 2, 1, 3)
    Q = tq.double().view(R, T, 8, 16).permute(0, 2, 1, 3)
    ref = (((Q * 0.25) @ K.transpose(-1, -2)).softmax(-1) @ V).permute(0, 2, 1, 3).reshape(R * T, 128)
    keys_pl = ops.to_planes(keys.to(dev))
    pe_t = dec._pe_terms(pe.to(dev).contiguous())
    got = dec._t2i_folded('final', tq.to(dev), keys_pl, pe_t, R, T, N)
    e_fold = float((got.cpu().double() - ref).abs().max())
    # the unfolded kernels on the same planes
    kv = ops.gemm(keys_pl, dec._packed['final.kv_proj'], bias=None, res=pe_t['final.kv_proj'], res_mod=N)
    ao = torch.empty((R * T, 128), dtype=torch.float32, device=dev)
    dec._t2i(tq.to(dev), kv, ao, R, T, N)
    e_unf = float((ao.cpu().double() - ref).abs().max())
    print(f't2i fold R={R} N={N} T={T}: folded err {e_fold:.2e}, unfolded err {e_unf:.2e} (max |ref| {float(ref.abs().max()):.2f})')
    assert e_fold < 2e-5 and e_unf < 2e-5
