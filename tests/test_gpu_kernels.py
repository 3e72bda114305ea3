"""-m gpu: every C-ABI kernel against a CPU fp64/fp32 restatement of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize('S,nh,dh,Bp', [(14, 3, 64, 5), (64, 2, 64, 1), (14, 2, 80, 3), (64, 1, 80, 1), (32, 2, 64, 2)])
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
    print('vit_attention', S, nh, dh, 'max abs err', err)
    assert err < 2e-5


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
