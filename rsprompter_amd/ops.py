"""Thin Python wrappers: torch tensors (device memory + stream plumbing only)
-> C-ABI calls into librsp_hip.so.  No arithmetic happens in torch here.
"""
import math

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3
DEFAULT_A_SCALE_LOG2 = 6


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _chk_f32(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise ValueError(f"{name}: expected a float32 CUDA/HIP tensor, got {t.dtype} on {t.device}")


class PackedWeight:
    """A GEMM weight [N, K] pre-split into fp16 hi/lo planes on the device.

    `scale_log2` is the power-of-two applied before the split so that the
    largest |w| lands near 2^14 (well inside fp16 range, lo plane normal).
    """

    def __init__(self, w, bias=None, device=None):
        w = w.detach().to(torch.float32)
        if w.dim() != 2:
            raise ValueError("PackedWeight expects a 2-D [N, K] matrix")
        device = device or w.device
        n, k = w.shape
        kpad = (k + 31) // 32 * 32
        if kpad != k:
            w = torch.nn.functional.pad(w, (0, kpad - k))
        amax = float(w.abs().max()) if w.numel() else 0.0
        if amax > 0 and math.isfinite(amax):
            e = int(math.floor(math.log2(16384.0 / amax)))
            e = max(-20, min(24, e))
        else:
            e = 0
        wd = w.contiguous().to(device)
        self.N, self.K = n, kpad
        self.scale_log2 = e
        self.hi = torch.empty((n, kpad), dtype=torch.float16, device=device)
        self.lo = torch.empty((n, kpad), dtype=torch.float16, device=device)
        lib = _lib.load()
        _lib.check(lib.rsp_split_f16(wd.data_ptr(), self.hi.data_ptr(), self.lo.data_ptr(),
                                     wd.numel(), e, _stream()), "rsp_split_f16")
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous().to(device)


def gemm(a, w, *, out=None, bias="auto", res=None, act=ACT_NONE, a_rowmap=None, c_rowmap=None,
         M=None, out_rows=None, res_mod=0, a_scale_log2=DEFAULT_A_SCALE_LOG2, conv=None):
    """C = act(A @ W^T + bias) + res   (see RspGemmDesc in include/rsp_hip.h).

    a: [rows, K] fp32 (row stride = a.stride(0)) or, with conv=(k, stride, pad),
       an NHWC tensor [B, H, W, C].
    """
    lib = _lib.load()
    _chk_f32(a, "a")
    d = _lib.RspGemmDesc()
    if conv is not None:
        k, stride, pad = conv
        if a.dim() != 4 or not a.is_contiguous():
            raise ValueError("conv gemm expects a contiguous NHWC tensor")
        B, H, W, C = a.shape
        Ho = (H + 2 * pad - k) // stride + 1
        Wo = (W + 2 * pad - k) // stride + 1
        m = B * Ho * Wo
        d.conv_k, d.conv_stride, d.conv_pad = k, stride, pad
        d.conv_H, d.conv_W, d.conv_C, d.conv_Ho, d.conv_Wo = H, W, C, Ho, Wo
        d.lda = C
        if w.K != k * k * C:
            raise ValueError(f"conv weight K={w.K} != {k}*{k}*{C}")
    else:
        if a.dim() != 2 or a.stride(1) != 1:
            raise ValueError("gemm expects a 2-D row-major A")
        m = a.shape[0] if M is None else M
        if a.shape[1] != w.K:
            raise ValueError(f"A has K={a.shape[1]}, weight has K={w.K}")
        d.lda = a.stride(0)
    n = w.N
    if out is None:
        rows = m if out_rows is None else out_rows
        out = torch.empty((rows, n), dtype=torch.float32, device=a.device)
    _chk_f32(out, "out")
    if bias == "auto":
        bias = w.bias
    if res is not None:
        _chk_f32(res, "res")
        d.ldr = res.stride(0)
    d.A, d.Bhi, d.Blo, d.C = a.data_ptr(), w.hi.data_ptr(), w.lo.data_ptr(), out.data_ptr()
    d.bias, d.res = _ptr(bias), _ptr(res)
    d.a_rowmap, d.c_rowmap = _ptr(a_rowmap), _ptr(c_rowmap)
    d.M, d.N, d.K = m, n, w.K
    d.ldc = out.stride(0)
    d.res_mod = res_mod
    d.act = act
    d.a_scale_log2 = a_scale_log2
    d.alpha = math.ldexp(1.0, -(a_scale_log2 + w.scale_log2))
    _lib.check(lib.rsp_gemm(d, _stream()), "rsp_gemm")
    return out


def layernorm(x, gamma, beta, eps=1e-6, act=ACT_NONE, out=None):
    lib = _lib.load()
    _chk_f32(x, "x")
    if not x.is_contiguous():
        raise ValueError("layernorm expects a contiguous tensor")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.rsp_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                 rows, C, eps, act, _stream()), "rsp_layernorm")
    return out


def vit_relpos(qkv, rel_pos_h, rel_pos_w, Bp, S, nh, dh):
    lib = _lib.load()
    rel = torch.empty((Bp * nh, S * S, 2 * S), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.rsp_vit_relpos(qkv.data_ptr(), rel_pos_h.data_ptr(), rel_pos_w.data_ptr(),
                                  rel.data_ptr(), Bp, S, nh, dh, _stream()), "rsp_vit_relpos")
    return rel


def vit_attention(qkv, rel, Bp, S, nh, dh, scale):
    lib = _lib.load()
    out = torch.empty((Bp * S * S, nh * dh), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.rsp_vit_attention(qkv.data_ptr(), rel.data_ptr(), out.data_ptr(), Bp, S, nh, dh,
                                     scale, _stream()), "rsp_vit_attention")
    return out


def patchify(img, patch):
    """NCHW fp32 image batch -> [B*gh*gw, C*p*p] patch rows (k = (c, ky, kx))."""
    lib = _lib.load()
    _chk_f32(img, "img")
    B, C, H, W = img.shape
    out = torch.empty((B * (H // patch) * (W // patch), C * patch * patch), dtype=torch.float32,
                      device=img.device)
    _lib.check(lib.rsp_patchify(img.data_ptr(), out.data_ptr(), B, C, H, W, patch, _stream()),
               "rsp_patchify")
    return out


def preprocess(imgs, mean, std, swap_rb, pad_divisor=1, pad_value=0.0, device=None):
    """DetDataPreprocessor: list of CHW uint8/fp32 tensors -> normalised fp32 [B,3,Hp,Wp]."""
    import ctypes
    lib = _lib.load()
    device = device or imgs[0].device
    Hm = max(int(i.shape[1]) for i in imgs)
    Wm = max(int(i.shape[2]) for i in imgs)
    Hp = (Hm + pad_divisor - 1) // pad_divisor * pad_divisor
    Wp = (Wm + pad_divisor - 1) // pad_divisor * pad_divisor
    out = torch.empty((len(imgs), 3, Hp, Wp), dtype=torch.float32, device=device)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    for b, im in enumerate(imgs):
        im = im.to(device).contiguous()
        if im.dtype == torch.uint8:
            is_u8 = 1
        else:
            im = im.to(torch.float32)
            is_u8 = 0
        _lib.check(lib.rsp_preprocess(im.data_ptr(), is_u8, out[b].data_ptr(), int(im.shape[1]),
                                      int(im.shape[2]), Hp, Wp, m3, s3, 1 if swap_rb else 0,
                                      float(pad_value), _stream()), "rsp_preprocess")
    return out
