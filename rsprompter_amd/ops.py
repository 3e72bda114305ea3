"""Thin Python wrappers: torch tensors (device memory + stream plumbing only)
-> C-ABI calls into librsp_hip.so.  No arithmetic happens in torch here.
"""
import math
import os

import torch

from . import _lib as _lib_real

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID, ACT_RELU_POST = 0, 1, 2, 3, 4
# Power-of-two pre-scale of activation planes (x * 2^e is what gets split into fp16 hi + lo).  e = 2 keeps |x| < 16376
# exactly representable (the split saturates beyond, rsp_common.h) -- headroom for the outlier activations of real SAM
# checkpoints (MLP hidden layer, residual stream) and for the RSFeatureAggregator's running sum over 16 layers, which
# passed 1023 (the old e = 6 limit) on the ViT-H fixture and turned whole rows into NaN.  The price is that `lo` goes
# subnormal for |x| < 0.03: an absolute error floor of 7.5e-9 per element, invisible next to the 1e-5 fp32 noise floor.
DEFAULT_A_SCALE_LOG2 = 2
DEBUG_FINITE = bool(int(__import__('os').environ.get('RSP_DEBUG_FINITE', '0')))


class Profiler:
    """Per-kernel HIP-event timing on the launch stream (bench.py's roofline leg; off by default)."""

    def __init__(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, nbytes, e0, e1 in self.records:
            a = agg.setdefault(name, dict(ms=0.0, calls=0, flops=0.0, bytes=0.0))
            a['ms'] += e0.elapsed_time(e1)
            a['calls'] += 1
            a['flops'] += flops
            a['bytes'] += nbytes
        return agg


_prof = None


def set_profiler(p):
    global _prof
    _prof = p


_timed_depth = 0


def _timed(name, flops, nbytes, fn, detail=None):
    global _timed_depth
    if _prof is None:
        return fn()
    if detail is not None and getattr(_prof, 'shapes', False):
        name = f'{name} {detail}'
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _timed_depth += 1
    try:
        r = fn()
    finally:
        _timed_depth -= 1
    e1.record()
    _prof.records.append((name, float(flops), float(nbytes), e0, e1))
    return r


class _ProfiledLib:
    """librsp_hip.so as seen by this module: with a profiler installed every entry point that is not already inside
    an annotated `_timed` region is bracketed by HIP events under its own symbol name, so that the per-step kernel
    table of bench.py accounts for ALL native calls, not only the ones carrying a FLOP/byte model."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        f = getattr(self._real, name)
        if _prof is None or _timed_depth > 0 or not name.startswith('rsp_'):
            return f

        def call(*a):
            return _timed(name, 0, 0, lambda: f(*a))
        return call


class _LibModule:
    """Stand-in for the `_lib` module inside ops.py (same attributes; `load()` returns the profiled view)."""

    def __getattr__(self, name):
        return getattr(_lib_real, name)

    @staticmethod
    def load():
        return _ProfiledLib(_lib_real.load())


_lib = _LibModule()


def require_device(dev):
    if torch.device(dev).type != 'cuda':
        raise RuntimeError('rsprompter_amd runs on the HIP device only (there is no CPU fallback); '
                           'move the model with .to("cuda")')


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _is_device(t):
    return t.is_cuda


def _chk_f32(t, name):
    if t.dtype != torch.float32 or not _is_device(t):
        raise ValueError(f"{name}: expected a float32 CUDA/HIP tensor, got {t.dtype} on {t.device}")


class PackedWeight:
    """A GEMM weight [N, K] pre-split into fp16 hi/lo planes on the device.

    `scale_log2` is the power-of-two applied before the split so that the
    largest |w| lands near 2^14 (well inside fp16 range, lo plane normal).
    """

    def __init__(self, w, bias=None, device=None, f8=False):
        """f8=True: the second plane is the cat8 plane of the fp8-corrected product (plane format word, rsp_hip.h)."""
        w = w.detach().to(torch.float32)
        if w.dim() != 2:
            raise ValueError("PackedWeight expects a 2-D [N, K] matrix")
        self.f8 = bool(f8)
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
        # KB32 layout [K/32][N][32]: every 32-wide K slice of the weight is one contiguous run
        self.hi = torch.empty((kpad // 32, n, 32), dtype=torch.float16, device=device)
        self.lo = torch.empty((kpad // 32, n, 32), dtype=torch.float16, device=device)
        lib = _lib.load()
        _lib.check(lib.rsp_split_f16_kb32(wd.data_ptr(), self.hi.data_ptr(), self.lo.data_ptr(),
                                          n, kpad, plane_word(e, self.f8), _stream()), "rsp_split_f16_kb32")
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous().to(device)


PLANE_F8 = 0x100        # include/rsp_hip.h "Plane format word"
# Opt-in fast mode (RSP_F8CORR=1, or set ops.F8_CORR before the model is built / first run): the four big GEMMs of every
# encoder block run fp16 hi.hi + ONE fp8 MFMA carrying both correction terms (2 units of matrix time instead of 3).
# Measured: GEMMs 1.15-1.25x faster, image embeddings 1.0-1.4e-4 instead of 1.2-1.6e-5 off the fp32
# reference -- inside the 1e-3 budget, but the masked query decoder then flips 10 instead of 4 attention-mask decisions
# on the ViT-H + LoRA fixture, so the default stays the three-pass fp16x3 product.
# RSP_F8CORR=mlp (round 6 study): only lin1 / lin2 of every block.
F8_CORR = {'0': False, '': False, '1': True, 'all': True, 'mlp': 'mlp'}[os.environ.get('RSP_F8CORR', '0')]


def plane_word(scale_log2, f8=False):
    return (int(scale_log2) & 0xff) | (PLANE_F8 if f8 else 0)


class Planes:
    """An fp32 matrix x [rows, K] held as two fp16 tensors hi = f16(x * 2^e), lo = f16(x * 2^e - hi) in the
    K-blocked "KB32" layout [K/32][rows][32].  Same bytes as fp32; lets the GEMM stream its A operand
    HBM -> LDS with the DMA engine in contiguous 1-KiB bursts.  `shape` is the LOGICAL shape
    (leading dims are free to re-factor: rows = prod(shape[:-1])).
    f8=True: `lo` holds the cat8 plane [e4m3(lo) x 32 | e4m3(hi) x 32] of the fp8-corrected product instead (same
    bytes); such planes feed GEMMs whose weight was packed with f8=True and nothing else."""

    def __init__(self, hi, lo, shape, scale_log2=DEFAULT_A_SCALE_LOG2, f8=False):
        self.hi, self.lo, self.shape, self.scale_log2, self.f8 = hi, lo, tuple(shape), scale_log2, bool(f8)

    @property
    def word(self):
        return plane_word(self.scale_log2, self.f8)

    @property
    def device(self):
        return self.hi.device

    @property
    def rows(self):
        return self.hi.shape[1]

    def view(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        n = 1
        for v in shape[:-1]:
            n *= v
        if shape[-1] != self.shape[-1] or n != self.rows:
            raise ValueError(f'cannot view planes {self.shape} as {shape}')
        return Planes(self.hi, self.lo, shape, self.scale_log2, self.f8)

    reshape = view


def empty_planes(shape, device, scale_log2=DEFAULT_A_SCALE_LOG2, f8=False):
    shape = tuple(shape)
    K = shape[-1]
    if K % 32:
        raise ValueError('planes need K % 32 == 0')
    rows = 1
    for v in shape[:-1]:
        rows *= v
    return Planes(torch.empty((K // 32, rows, 32), dtype=torch.float16, device=device),
                  torch.empty((K // 32, rows, 32), dtype=torch.float16, device=device), shape, scale_log2, f8)


def to_planes(x, scale_log2=DEFAULT_A_SCALE_LOG2, f8=False):
    """fp32 tensor -> Planes (one extra HBM pass; producers that can emit planes directly avoid it)."""
    lib = _lib.load()
    _chk_f32(x, "x")
    x = x if x.is_contiguous() else x.contiguous()
    p = empty_planes(x.shape, x.device, scale_log2, f8)
    _timed('split_f16_kernel', 0, 8.0 * x.numel(),
           lambda: _lib.check(lib.rsp_split_f16_kb32(x.data_ptr(), p.hi.data_ptr(), p.lo.data_ptr(), p.rows,
                                                     x.shape[-1], p.word, _stream()), "rsp_split_f16_kb32"))
    return p


class PlaneWeight:
    """A GEMM 'weight' that is itself an activation held as Planes (rows [r0, r0+n) of a plane tensor),
    e.g. mask_feature in `einsum('bqc,bchw->bqhw')` (models.py:357)."""

    def __init__(self, planes, r0=0, n=None):
        self.N = planes.rows - r0 if n is None else n
        self.K = planes.shape[-1]
        self.scale_log2 = planes.scale_log2
        self.f8 = planes.f8
        self.b_rows = planes.rows
        self.hi = planes.hi[:, r0:, :]
        self.lo = planes.lo[:, r0:, :]
        self.bias = None


def _dma_tile_name(m, n, hint=0, conv=False, k=1 << 30, f8=False):
    """Mirror of the tile choice in rsp_gemm_dma_dispatch (gemm_dma.hip) - profiler labels only."""
    nblk = lambda bm, bn: -(-n // bn) * -(-m // bm)
    small = '128x128' if n > 64 else ('128x64' if n > 32 else '128x32')
    hint &= 0xff
    if hint == 1:
        hint = 0                     # "the round-2 rule"
    if hint in (3, 9, 10, 11, 15, 17, 31):
        return '256x256'
    if hint in (2, 4, 12, 13, 16, 18, 19, 32):
        return '256x128'
    if hint != 0:
        return small
    if conv:
        return '256x256' if n > 128 and nblk(256, 256) >= 1024 else small
    if f8 and n > 128 and nblk(256, 256) >= 512:
        return '256x256'
    if k <= 256:
        return '256x256' if n > 128 and nblk(256, 256) >= 1024 else small
    if n <= 64:
        return small
    cost = lambda bm, bn, bpc, eff: -(-nblk(bm, bn) // (256 * bpc)) * bm * bn * bpc / eff
    best, name = cost(128, 128, 2, 0.88), '128x128'
    c = cost(256, 128, 1, 0.96)
    if c < best:
        best, name = c, '256x128'
    if n > 128 and (cost(256, 256, 1, 1.0) <= best or nblk(256, 256) >= 1024 or (nblk(256, 256) >= 512 and k >= 2048)):
        name = '256x256'
    return name


PP_AUTO = True      # False: rsp_gemm's choice of the ping-pong kernel (csrc/gemm_pp.hip) is overridden by gemm_s2 (A/B runs)


def gemm(a, w, *, out=None, bias="auto", res=None, act=ACT_NONE, a_rowmap=None, c_rowmap=None,
         M=None, out_rows=None, res_mod=0, a_scale_log2=DEFAULT_A_SCALE_LOG2, conv=None,
         res_bmap=None, res_brows=0, out_planes=False, out_f32=True, dma="auto", tile_hint=0, c_ncols=0, pl_col0=0,
         out_f8=False, plan_only=False):
    """C = act(A @ W^T + bias) + res   (see RspGemmDesc in include/rsp_hip.h).

    plan_only=True launches nothing and returns rsp_gemm_s2_epilogue(desc): -1 = a gemm_dma.hip / gemm.hip tile serves
    the call, otherwise the epilogue form of gemm_f16x3_s2_kernel (tests assert which specialisation they exercise).

    a: [rows, K] fp32 (row stride = a.stride(0)) or, with conv=(k, stride, pad),
       an NHWC tensor [B, H, W, C].
    A weight packed with f8=True selects the fp8-corrected product: `a` must then be f8 Planes (or an fp32 tensor, which
    is converted); out_f8=True writes the output planes in that format for the next such GEMM.
    """
    lib = _lib.load()
    w_f8 = bool(getattr(w, 'f8', False))
    if not isinstance(a, Planes):
        _chk_f32(a, "a")
        # the DMA fast path wants fp16 planes: convert once when the GEMM is big enough to amortise it
        if (dma is True or (dma == "auto" and w.N > 64 and w.K >= 128)) and a.is_contiguous() \
                and a.shape[-1] % 32 == 0:
            a = to_planes(a, a_scale_log2, w_f8)
    is_planes = isinstance(a, Planes)
    if is_planes:
        a_scale_log2 = a.scale_log2
    if (a.f8 if is_planes else False) != w_f8:
        raise ValueError('fp8-corrected product: A planes and the weight must both be packed with f8=True')
    d = _lib.RspGemmDesc()
    if conv is not None:
        k, stride, pad = conv
        if len(a.shape) != 4 or not (is_planes or a.is_contiguous()):
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
        if len(a.shape) != 2 or (not is_planes and a.stride(1) != 1):
            raise ValueError("gemm expects a 2-D row-major A")
        m = a.shape[0] if M is None else M
        if a.shape[1] != w.K:
            raise ValueError(f"A has K={a.shape[1]}, weight has K={w.K}")
        d.lda = a.shape[1] if is_planes else a.stride(0)
    n = w.N
    rows = m if out_rows is None else out_rows
    pl = None
    if (c_ncols or pl_col0) and not (is_planes and out_planes):
        raise ValueError('column-range outputs (c_ncols / pl_col0) belong to the plane path with out_planes=True')
    if out_planes:
        if isinstance(out_planes, Planes):         # a caller-owned plane tensor (rows the GEMM does not map keep their contents)
            pl = out_planes
            if tuple(pl.shape) != (rows, n - pl_col0) or pl.f8 != bool(out_f8):
                raise ValueError('out_planes: a Planes tensor of the output\'s shape and format')
        else:
            pl = empty_planes((rows, n - pl_col0), a.device, f8=out_f8)
        d.Chi, d.Clo, d.c_scale_log2, d.c_rows = pl.hi.data_ptr(), pl.lo.data_ptr(), pl.word, rows
    d.c_ncols, d.pl_col0 = c_ncols, pl_col0
    if out is None and out_f32:
        # (a scattered output leaves the rows nobody maps to unwritten: zeroed under the finite check, else never read)
        alloc = torch.zeros if (DEBUG_FINITE and c_rowmap is not None) else torch.empty
        out = alloc((rows, c_ncols or n), dtype=torch.float32, device=a.device)
    if out is not None:
        _chk_f32(out, "out")
    if bias == "auto":
        bias = w.bias
    if isinstance(res, Planes):
        if not is_planes:
            raise ValueError('a plane residual needs the plane path')
        d.res_hi, d.res_lo, d.res_scale_log2, d.res_rows = res.hi.data_ptr(), res.lo.data_ptr(), res.scale_log2, res.rows
        res = None
    if res is not None:
        _chk_f32(res, "res")
        d.ldr = res.stride(0)
    if is_planes:
        d.Ahi, d.Alo, d.a_rows = a.hi.data_ptr(), a.lo.data_ptr(), a.rows
    else:
        d.A = a.data_ptr()
    d.Bhi, d.Blo, d.C = w.hi.data_ptr(), w.lo.data_ptr(), _ptr(out)
    d.bias, d.res = _ptr(bias), _ptr(res)
    d.a_rowmap, d.c_rowmap = _ptr(a_rowmap), _ptr(c_rowmap)
    d.M, d.N, d.K = m, n, w.K
    d.ldc = out.stride(0) if out is not None else n
    d.res_mod = res_mod
    d.res_bmap, d.res_brows = _ptr(res_bmap), res_brows
    d.b_rows = getattr(w, 'b_rows', 0)
    d.tile_hint = tile_hint
    if d.c_rows == 0:
        d.c_rows = rows          # also bounds C when rows are scattered (c_rowmap)
    d.act = act
    d.a_scale_log2 = plane_word(a_scale_log2, w_f8)
    d.alpha = math.ldexp(1.0, -(a_scale_log2 + w.scale_log2))
    if not PP_AUTO and is_planes and tile_hint == 0 and _lib_real.load().rsp_gemm_uses_pp(d):
        d.tile_hint = 40        # A/B switch (tools/ab_bench.py "ops.PP_AUTO=False"): the round-3 kernel for these shapes
    if plan_only == 'pp':       # 256 / 128: gemm_f16x3_pp_kernel (csrc/gemm_pp.hip) with that tile serves the call, else 0
        return int(_lib_real.load().rsp_gemm_uses_pp(d)) if is_planes else 0
    if plan_only:
        return int(_lib_real.load().rsp_gemm_s2_epilogue(d)) if is_planes else -1
    tile = _dma_tile_name(m, n, tile_hint, conv is not None, w.K, w_f8) if is_planes else ('128x128' if n > 64 else ('128x64' if n > 32 else '128x32'))
    kname = ('gemm_f16f8_dma_kernel' if w_f8 else 'gemm_f16x3_dma_kernel') if is_planes else 'gemm_f16x3_kernel'
    if is_planes and _prof is not None and _lib_real.load().rsp_gemm_uses_s2(d):
        kname, tile = 'gemm_f16x3_s2_kernel', '256x128'         # the kernel rsp_gemm really launches (profiler label)
    elif is_planes and _prof is not None and _lib_real.load().rsp_gemm_uses_pp(d):
        kname, tile = 'gemm_f16x3_pp_kernel', f'{_lib_real.load().rsp_gemm_uses_pp(d)}x256'
    _timed(f'{kname}<{tile}>', 2.0 * m * n * w.K, 4.0 * (m * w.K + m * n) + 4.0 * n * w.K,
           lambda: _lib.check(lib.rsp_gemm(d, _stream()), "rsp_gemm"),
           detail=f'M={m} N={n} K={w.K}' + (' conv' if conv is not None else ''))
    if DEBUG_FINITE and out is not None and not bool(torch.isfinite(out).all()):
        raise FloatingPointError(f'rsp_gemm produced non-finite values (M={m} N={n} K={w.K})')
    if out_planes:
        return (out, pl) if out_f32 else pl
    return out


def layernorm(x, gamma, beta, eps=1e-6, act=ACT_NONE, out=None, planes=False, f32=True, f8=False):
    """planes=True additionally returns the result as fp16 Planes (f32=False: planes only; f8: cat8 second plane)."""
    lib = _lib.load()
    _chk_f32(x, "x")
    if not x.is_contiguous():
        raise ValueError("layernorm expects a contiguous tensor")
    C = x.shape[-1]
    rows = x.numel() // C
    if planes:
        pl = empty_planes(x.shape, x.device, f8=f8)
        y = torch.empty_like(x) if f32 else None
        _timed('layernorm_kernel', 0, 8.0 * x.numel() + (4.0 * x.numel() if f32 else 0),
               lambda: _lib.check(lib.rsp_layernorm_ex(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(y),
                                                       pl.hi.data_ptr(), pl.lo.data_ptr(), pl.word, rows, C,
                                                       eps, act, _stream()), "rsp_layernorm_ex"))
        return (y, pl) if f32 else pl
    if out is None:
        out = torch.empty_like(x)
    _timed('layernorm_kernel', 0, 8.0 * x.numel(),
           lambda: _lib.check(lib.rsp_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                                rows, C, eps, act, _stream()), "rsp_layernorm"))
    return out


def vit_relpos(qkv, rel_pos_h, rel_pos_w, Bp, S, nh, dh, q_ld=None, rows=None):
    """q_ld: row stride of the tensor holding q in its first nh*dh columns (default: the [.., 3, nh, dh] qkv matrix).
    rows (int32, windowed layers): only these q rows are evaluated (the real tokens of padded windows); the other rows of
    the result stay unwritten."""
    lib = _lib.load()
    rel = torch.empty((Bp * nh, S * S, 2 * S), dtype=torch.float32, device=qkv.device)
    q_ld = 3 * nh * dh if q_ld is None else q_ld
    n_rows = 0 if rows is None else int(rows.shape[0])
    _timed('vit_relpos_kernel', 2.0 * Bp * nh * S * S * 2 * S * dh, 0,
           lambda: _lib.check(lib.rsp_vit_relpos_rows(qkv.data_ptr(), q_ld, rel_pos_h.data_ptr(), rel_pos_w.data_ptr(),
                                                      rel.data_ptr(), Bp, S, nh, dh, _ptr(rows), n_rows, _stream()),
                              "rsp_vit_relpos_rows"))
    return rel


def vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale, planes=False, f8=False, win_grid=None):
    """SAM ViT attention with q as fp32 rows [Bp*T, nh*dh] and K | V as the fp16 Planes [Bp*T, 2*nh*dh] the qkv GEMM
    wrote (gemm(..., out_planes=True, c_ncols=D, pl_col0=D)).  planes=True: the output only as Planes.
    win_grid=(windows per image side, real rows / columns of the last window): the outputs of the padded tokens of
    window_partition are not computed (their rows of the result are left unwritten)."""
    lib = _lib.load()
    T, D = S * S, nh * dh
    if not isinstance(kv, Planes) or kv.shape[-1] != 2 * D or kv.rows < Bp * T:
        raise ValueError('kv must be the K | V planes of the qkv GEMM')
    _chk_f32(q, 'q')
    out = None if planes else torch.empty((Bp * T, D), dtype=torch.float32, device=q.device)
    pl = empty_planes((Bp * T, D), q.device, f8=f8) if planes else None
    hi, lo, e = (pl.hi.data_ptr(), pl.lo.data_ptr(), pl.word) if planes else (0, 0, 0)
    if kv.f8:
        raise ValueError('K | V planes are consumed as fp16 hi / lo planes (qkv GEMM: out_f8=False)')
    kind = 'global' if T >= 1024 else 'window'
    wn, wr = (int(win_grid[0]), int(win_grid[1])) if (win_grid is not None and S == 14) else (0, 0)
    _timed('attn_stream_kernel<vit,global>' if S != 14 else 'attn_win_kernel<vit,window,rel given>',
           4.0 * Bp * nh * T * T * dh, 0,
           lambda: _lib.check(lib.rsp_vit_attention_planes_ex(q.data_ptr(), q.stride(0), kv.hi.data_ptr(), kv.lo.data_ptr(),
                                                              kv.rows, kv.scale_log2, rel.data_ptr(), _ptr(out), hi, lo, e,
                                                              Bp, S, nh, dh, scale, wn, wr, _stream()),
                              "rsp_vit_attention_planes_ex"))
    return pl if planes else out


def pack_relpos_tables(rel_pos_h, rel_pos_w, S, dh):
    """the two rel-pos tables of a windowed layer ([2S-1, dh] fp32 each) as the fp16 hi / lo planes rsp_vit_window_attention
    DMAs into LDS: uint16 [2, 2, 32, dh + 8] (rsp_pack_relpos_tables; once per layer at pack time)."""
    lib = _lib.load()
    _chk_f32(rel_pos_h, 'rel_pos_h')
    _chk_f32(rel_pos_w, 'rel_pos_w')
    if tuple(rel_pos_h.shape) != (2 * S - 1, dh) or tuple(rel_pos_w.shape) != (2 * S - 1, dh):
        raise ValueError(f'rel-pos tables must be [{2 * S - 1}, {dh}]')
    out = torch.empty((2, 2, 32, dh + 8), dtype=torch.float16, device=rel_pos_h.device)
    _lib.check(lib.rsp_pack_relpos_tables(rel_pos_h.contiguous().data_ptr(), rel_pos_w.contiguous().data_ptr(), out.data_ptr(),
                                          S, dh, _stream()), 'rsp_pack_relpos_tables')
    return out


def vit_window_attention(q, kv, rel_tab, Bp, nh, dh, scale, planes=False, f8=False, win_grid=None, variant=0):
    """Windowed SamVisionAttention (HF:803-831 with get_decomposed_rel_pos HF:761-801) in ONE kernel: q fp32 rows
    [Bp*196, nh*dh], K | V as the qkv GEMM's fp16 Planes, rel_tab = pack_relpos_tables(...) of the layer; the rel-pos terms
    are computed inside (csrc/attn_win.hip).  win_grid as in vit_attention_planes."""
    lib = _lib.load()
    T, D = 196, nh * dh
    if not isinstance(kv, Planes) or kv.shape[-1] != 2 * D or kv.rows < Bp * T or kv.f8:
        raise ValueError('kv must be the fp16 K | V planes of the qkv GEMM')
    if tuple(rel_tab.shape) != (2, 2, 32, dh + 8) or rel_tab.dtype != torch.float16 or not rel_tab.is_contiguous():
        raise ValueError('rel_tab must come from pack_relpos_tables')
    _chk_f32(q, 'q')
    out = None if planes else torch.empty((Bp * T, D), dtype=torch.float32, device=q.device)
    pl = empty_planes((Bp * T, D), q.device, f8=f8) if planes else None
    hi, lo, e = (pl.hi.data_ptr(), pl.lo.data_ptr(), pl.word) if planes else (0, 0, 0)
    wn, wr = (int(win_grid[0]), int(win_grid[1])) if win_grid is not None else (0, 0)
    # flops: the padded-window figure of SURVEY section 8(d) (bench.py also quotes the evaluated-query figure); the 30 + 28
    # rel-pos / bias MFMAs per wave are not counted
    _timed('attn_win_kernel<vit,window>', 4.0 * Bp * nh * T * T * dh, 0,
           lambda: _lib.check(lib.rsp_vit_window_attention(q.data_ptr(), q.stride(0), kv.hi.data_ptr(), kv.lo.data_ptr(),
                                                           kv.rows, kv.scale_log2, rel_tab.data_ptr(), _ptr(out), hi, lo, e,
                                                           Bp, nh, dh, scale, wn, wr, variant, _stream()),
                              'rsp_vit_window_attention'))
    return pl if planes else out


SAM_I2T_FUSED_MAX_TOKENS = 10   # rsp_sam_i2t_fused (LDS budget); more tokens: sam_i2t_attention + GEMM + layernorm


def sam_i2t_fused(q, k, v, wo, bo, gamma, beta, *, R, T, N, scale, eps=1e-6, q_map=None, res=None, res_map=None,
                  res_planes=None, planes=True, f32=False):
    """LayerNorm(residual + out_proj(image -> token attention)) in one kernel (HF:340-348), out_proj folded into the
    values.  q [Rq*N,128] (q_map: RoI -> row block), k / v [R*T,128], wo [256,128] fp32, bo [256]; residual either fp32
    rows `res` [Rres*N,256] (+ res_map) or Planes of the [R*N,256] tensor.  Returns Planes (and / or fp32)."""
    lib = _lib.load()
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v'), (wo, 'wo'), (bo, 'bo'), (gamma, 'gamma'), (beta, 'beta')):
        _chk_f32(t, n)
        if not t.is_contiguous():
            raise ValueError(f'sam_i2t_fused: {n} must be contiguous')
    if tuple(wo.shape) != (256, 128) or q.shape[-1] != 128:
        raise ValueError('sam_i2t_fused expects internal width 128 and a [256, 128] out_proj weight')
    if (res is None) == (res_planes is None):
        raise ValueError('sam_i2t_fused: exactly one of res / res_planes')
    d = _lib.RspI2tFusedDesc()
    d.q, d.q_map, d.k, d.v, d.wo, d.bo = q.data_ptr(), _ptr(q_map), k.data_ptr(), v.data_ptr(), wo.data_ptr(), bo.data_ptr()
    if res is not None:
        _chk_f32(res, 'res')
        if not res.is_contiguous() or res.shape[-1] != 256:
            raise ValueError('sam_i2t_fused: res must be contiguous rows of 256')
        d.res, d.res_map = res.data_ptr(), _ptr(res_map)
    else:
        if res_planes.f8 or res_planes.rows != R * N or res_planes.shape[-1] != 256:
            raise ValueError('sam_i2t_fused: res_planes must be the fp16 hi / lo planes of the [R*N, 256] tensor')
        d.res_hi, d.res_lo, d.res_scale_log2 = res_planes.hi.data_ptr(), res_planes.lo.data_ptr(), res_planes.scale_log2
    d.gamma, d.beta, d.eps = gamma.data_ptr(), beta.data_ptr(), eps
    out = torch.empty((R * N, 256), dtype=torch.float32, device=q.device) if f32 else None
    pl = empty_planes((R * N, 256), q.device) if planes else None
    d.out = _ptr(out)
    if pl is not None:
        d.out_hi, d.out_lo, d.out_scale_log2 = pl.hi.data_ptr(), pl.lo.data_ptr(), pl.scale_log2
    d.R, d.T, d.N, d.scale = R, T, N, scale
    # algorithmic work: scores + the folded out_proj product; bytes: q + residual + result rows
    _timed('sam_i2t_fused_kernel', 2.0 * R * N * T * (128 + 8 * 256), 4.0 * R * N * (128 + 256 + 256),
           lambda: _lib.check(lib.rsp_sam_i2t_fused(d, _stream()), "rsp_sam_i2t_fused"), detail=f'R={R} T={T} N={N}')
    if planes and f32:
        return out, pl
    return pl if planes else out


_attn_ws = {}


def _attn_workspace(nbytes, device):
    """K / V^T plane scratch of the global-attention path, kept per device (layers run back to back on one stream)."""
    key = str(device)
    buf = _attn_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        _attn_ws[key] = buf
    return buf


def vit_attention(qkv, rel, Bp, S, nh, dh, scale, planes=False):
    """planes=True: return the output only as fp16 Planes (it feeds the proj GEMM's DMA path)."""
    lib = _lib.load()
    kind = 'global' if S * S >= 1024 else 'window'
    fl = 4.0 * Bp * nh * (S * S) ** 2 * dh
    pl = empty_planes((Bp * S * S, nh * dh), qkv.device) if planes else None
    out = None if planes else torch.empty((Bp * S * S, nh * dh), dtype=torch.float32, device=qkv.device)
    hi, lo, e = (pl.hi.data_ptr(), pl.lo.data_ptr(), pl.scale_log2) if planes else (0, 0, 0)
    if S in (64, 32):
        ws = _attn_workspace(int(lib.rsp_vit_attention_global_ws_bytes(Bp, S, nh, dh)), qkv.device)
        _timed(f'attn_global_kernel<vit>', fl, 0,
               lambda: _lib.check(lib.rsp_vit_attention_global(qkv.data_ptr(), rel.data_ptr(), ws.data_ptr(), _ptr(out),
                                                               hi, lo, e, Bp, S, nh, dh, scale, _stream()),
                                  "rsp_vit_attention_global"))
    else:
        _timed(f'attn_kernel<vit,{kind}>', fl, 0,
               lambda: _lib.check(lib.rsp_vit_attention_ex(qkv.data_ptr(), rel.data_ptr(), _ptr(out), hi, lo, e, Bp, S,
                                                           nh, dh, scale, _stream()), "rsp_vit_attention_ex"))
    return pl if planes else out


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


def paste_masks(logits_nhwc, labels, boxes, img_hw, thr=0.5):
    """FCNMaskHead mask paste: logits [k, Hm, Wm, C] (NHWC), labels int [k] or None, boxes [k, 4] -> bool [k, H, W];
    thr < 0: uint8 [k, H, W], the pasted probabilities as (p * 255) truncated (fcn_mask_head.py:390-394)."""
    lib = _lib.load()
    _chk_f32(logits_nhwc, 'logits')
    k, Hm, Wm, C = logits_nhwc.shape
    H, W = int(img_hw[0]), int(img_hw[1])
    out = torch.empty((k, H, W), dtype=torch.bool if thr >= 0 else torch.uint8, device=logits_nhwc.device)
    if k:
        lab = None if labels is None else labels.to(torch.int32).contiguous()
        _lib.check(lib.rsp_paste_masks(logits_nhwc.contiguous().data_ptr(), _ptr(lab), boxes.contiguous().data_ptr(), k, Hm, Wm,
                                       C, H, W, float(thr), out.data_ptr(), _stream()), "rsp_paste_masks")
    return out


def resize_pad(img_hwc, new_hw, pad_hw, pad_val=(0.0, 0.0, 0.0), out=None, normalise=None):
    """Resize(keep_ratio) + Pad of the test pipeline on one decoded HWC image (uint8 / fp32, device tensor) ->
    fp32 [3, Hp, Wp] (see rsp_resize_pad).  normalise = (mean3, std3, swap_rb) fuses the DetDataPreprocessor step."""
    import ctypes
    lib = _lib.load()
    if img_hwc.dim() != 3 or img_hwc.shape[2] != 3 or not _is_device(img_hwc):
        raise ValueError('resize_pad expects an [H, W, 3] device tensor')
    im = img_hwc.contiguous()
    if im.dtype != torch.uint8:
        im = im.to(torch.float32)
    H, W = int(im.shape[0]), int(im.shape[1])
    Hn, Wn = int(new_hw[0]), int(new_hw[1])
    Hp, Wp = int(pad_hw[0]), int(pad_hw[1])
    if out is None:
        out = torch.empty((3, Hp, Wp), dtype=torch.float32, device=im.device)
    p3 = (ctypes.c_float * 3)(*[float(v) for v in pad_val])
    if normalise is None:
        m3 = s3 = None
        nrm, swap = 0, 0
    else:
        m3 = (ctypes.c_float * 3)(*[float(v) for v in normalise[0]])
        s3 = (ctypes.c_float * 3)(*[float(v) for v in normalise[1]])
        nrm, swap = 1, 1 if normalise[2] else 0
    _lib.check(lib.rsp_resize_pad(im.data_ptr(), 1 if im.dtype == torch.uint8 else 0, out.data_ptr(), H, W, Hn, Wn,
                                  Hp, Wp, p3, nrm, swap, m3, s3, _stream()), "rsp_resize_pad")
    return out


def conv_transpose2x2(x_nhwc, w_dy, bias, act=ACT_NONE, a_scale_log2=DEFAULT_A_SCALE_LOG2, out_planes=False,
                      hyper=None, ln=None):
    """ConvTranspose2d(k=2, s=2) on NHWC as GEMMs.

    x_nhwc: fp32 [B, H, W, Cin] or Planes of that logical shape.
    w_dy: a pair of PackedWeights [(dx, co), Cin] (one GEMM per output-row parity, bias tiled x2), or ONE
    PackedWeight [(dy, dx, co), Cin] (all four sub-pixels in one pass over A, bias tiled x4; plane path only).
    Returns [B, 2H, 2W, Cout] (fp32, or Planes when out_planes).  hyper [B, Cout=32]: fuse
    `sum_c act(.)[.., c] * hyper[b, c]` into the epilogue and return [B, 2H, 2W] (the SAM decoder's
    hyper-network product, HF:523-531).  ln=(gamma, beta, eps): LayerNorm2d over the Cout=64 channels of every
    output pixel before `act` (HF:519-520), one-weight form only, returns Planes."""
    B, H, W, Cin = x_nhwc.shape
    four = not isinstance(w_dy, (tuple, list))
    ws = (w_dy,) if four else tuple(w_dy)
    cout = ws[0].N // (4 if four else 2)
    dev = x_nhwc.device
    if not isinstance(x_nhwc, Planes) and (four or out_planes or hyper is not None
                                           or (Cin % 32 == 0 and B * H * W >= 4096)):
        x_nhwc = to_planes(x_nhwc.contiguous(), a_scale_log2)
    a = x_nhwc.view(B * H * W, Cin)
    dys = (-1,) if four else (0, 1)
    if hyper is not None:
        out = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device=dev)
        if four and Cin == 64 and cout == 32 and act == ACT_GELU and bias is not None:
            # the SAM upscaler's last stage: dedicated streaming kernel (weights resident in LDS)
            lib = _lib.load()
            w = ws[0]
            rows = B * H * W
            _timed('sam_upscale2_kernel', 2.0 * rows * 128 * 64, 4.0 * rows * 64 + 4.0 * rows * 4,
                   lambda: _lib.check(lib.rsp_sam_upscale2(a.hi.data_ptr(), a.lo.data_ptr(), rows, a.scale_log2,
                                                           w.hi.data_ptr(), w.lo.data_ptr(), w.scale_log2,
                                                           bias.data_ptr(), hyper.data_ptr(), out.data_ptr(), H * W, W,
                                                           _stream()), "rsp_sam_upscale2"))
            return out
        for w, dy in zip(ws, dys):
            _gemm_ct(a, w, None, bias, act, W, dy, a_scale_log2, hyper=hyper, hd_out=out, hd_rows=H * W)
        return out
    if ln is not None:
        if not four or cout != 64:
            raise ValueError('the fused LayerNorm epilogue needs the one-weight form with Cout == 64')
        out_planes = True
    if out_planes:
        pl = empty_planes((B, 2 * H, 2 * W, cout), dev)
        for w, dy in zip(ws, dys):
            _gemm_ct(a, w, None, bias, act, W, dy, a_scale_log2, planes_out=pl, c_rows=B * 4 * H * W, ln=ln)
        return pl
    out = torch.empty((B, 2 * H, 2 * W, cout), dtype=torch.float32, device=dev)
    o2 = out.view(B * 2 * H * W, 2 * cout)
    for w, dy in zip(ws, dys):
        _gemm_ct(a, w, o2, bias, act, W, dy, a_scale_log2)
    return out


def _gemm_ct(a, w, out, bias, act, ct_W, ct_dy, a_scale_log2, planes_out=None, c_rows=0, hyper=None, hd_out=None,
             hd_rows=0, ln=None):
    lib = _lib.load()
    d = _lib.RspGemmDesc()
    is_pl = isinstance(a, Planes)
    if is_pl:
        d.Ahi, d.Alo, d.a_rows = a.hi.data_ptr(), a.lo.data_ptr(), a.rows
        a_scale_log2 = a.scale_log2
        d.lda = a.shape[1]
    else:
        d.A = a.data_ptr()
        d.lda = a.stride(0)
    d.Bhi, d.Blo, d.C = w.hi.data_ptr(), w.lo.data_ptr(), _ptr(out)
    d.bias = _ptr(bias)
    d.M, d.N, d.K = a.shape[0], w.N, w.K
    d.ldc = out.stride(0) if out is not None else w.N
    d.act = act
    d.a_scale_log2 = a_scale_log2
    d.alpha = math.ldexp(1.0, -(a_scale_log2 + w.scale_log2))
    d.ct_W, d.ct_dy = ct_W, ct_dy
    if planes_out is not None:
        # the convT output matrix is [B*2H*W rows of 2*Cout]; as planes of the NHWC tensor [.., Cout] the row index
        # doubles (dx folds into the row): handled by viewing the plane tensor as [rows, 2*Cout]
        d.Chi, d.Clo, d.c_scale_log2, d.c_rows = planes_out.hi.data_ptr(), planes_out.lo.data_ptr(), planes_out.scale_log2, c_rows
    if hd_out is not None:
        d.hd_hyper, d.hd_out, d.hd_rows = hyper.data_ptr(), hd_out.data_ptr(), hd_rows
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), ln[2]
    _timed('gemm_f16x3_dma_kernel<convT>' if is_pl else 'gemm_f16x3_kernel<convT>', 2.0 * d.M * d.N * d.K, 4.0 * (d.M * d.K + d.M * d.N),
           lambda: _lib.check(lib.rsp_gemm(d, _stream()), "rsp_gemm(convT)"),
           detail=f'M={d.M} N={d.N} K={d.K}' + (' hyper' if hd_out is not None else '') + (' ln' if ln is not None else ''))


def attention(q, k, v, out, *, B, nh, dh, Tq, Tk, scale, q_strides, k_strides, v_strides, o_strides,
              kv_batch_map=None, q_batch_map=None, out_planes=None, mask=None):
    """Generic strided multi-head attention (RspAttnDesc). strides = (batch, token, head) in elements.
    out_planes: optional Planes receiving a KB32 copy of the dense [B*Tq, nh*dh] output (out may be None)."""
    lib = _lib.load()
    d = _lib.RspAttnDesc()
    d.q, d.k, d.v, d.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), _ptr(out)
    if out_planes is not None:
        d.out_hi, d.out_lo, d.out_scale_log2 = out_planes.hi.data_ptr(), out_planes.lo.data_ptr(), out_planes.scale_log2
    d.kv_batch_map = _ptr(kv_batch_map)
    d.q_batch_map = _ptr(q_batch_map)
    d.mask = _ptr(mask)
    d.q_bs, d.q_ts, d.q_hs = q_strides
    d.k_bs, d.k_ts, d.k_hs = k_strides
    d.v_bs, d.v_ts, d.v_hs = v_strides
    d.o_bs, d.o_ts, d.o_hs = o_strides
    d.B, d.nh, d.dh, d.Tq, d.Tk, d.scale = B, nh, dh, Tq, Tk, scale
    _timed('attn_kernel<sam_decoder>', 4.0 * B * nh * Tq * Tk * dh, 0,
           lambda: _lib.check(lib.rsp_attention(d, _stream()), "rsp_attention"),
           detail=f'B={B} nh={nh} dh={dh} Tq={Tq} Tk={Tk}')
    return out if out_planes is None else out_planes


SAM_T2I_MAX_TOKENS = 12     # rsp_sam_t2i_attention; more tokens go through the generic rsp_attention
SAM_I2T_MAX_TOKENS = 16


def sam_t2i_attention(q, kv, out, *, R, T, N, scale, kv_map=None):
    """SAM decoder token->image attention (8 heads x 16, exact fp32). q [R*T,128], kv [Rkv*N,256] = K|V, out [R*T,128]."""
    lib = _lib.load()
    for t, n in ((q, 'q'), (kv, 'kv'), (out, 'out')):
        _chk_f32(t, n)
        if not t.is_contiguous():
            raise ValueError(f'sam_t2i_attention: {n} must be contiguous')
    if q.shape[-1] != 128 or kv.shape[-1] != 256:
        raise ValueError('sam_t2i_attention expects internal width 128 (q) and fused K|V rows of 256')
    _timed('sam_t2i_kernel', 4.0 * R * T * N * 128, 4.0 * R * N * 256,
           lambda: _lib.check(lib.rsp_sam_t2i_attention(q.data_ptr(), kv.data_ptr(), _ptr(kv_map), out.data_ptr(),
                                                        R, T, N, scale, _stream()), "rsp_sam_t2i_attention"),
           detail=f'R={R} T={T} N={N}')
    return out


def sam_upscale_fused(x, w1, bias1, gamma, beta, eps, w2p, bias2, hyper, h, w):
    """x Planes [R*h*w, 256] -> masks [R, 4h, 4w] (ConvT + LN + GELU + ConvT + GELU + hyper dot in one kernel, csrc/upscale.hip);
    w1 = PackedWeight [(dy, dx, co), 256], w2p = PackedWeight [(dy2, dx2, c2), 64] with its K columns in the order
    rsp_sam_upscale_fused documents (sam_decoder._upscale2_k_order)."""
    lib = _lib.load()
    if not isinstance(x, Planes) or x.shape[-1] != 256 or x.f8:
        raise ValueError('sam_upscale_fused: x must be fp16 planes with 256 columns')
    rows = x.rows
    R = rows // (h * w)
    out = torch.empty((R, 4 * h, 4 * w), dtype=torch.float32, device=x.device)
    _timed('sam_upscale_fused_kernel', 2.0 * rows * (256 * 256 + 4 * 128 * 64), 4.0 * rows * 256 + 64.0 * rows,
           lambda: _lib.check(lib.rsp_sam_upscale_fused(x.hi.data_ptr(), x.lo.data_ptr(), x.rows, x.scale_log2,
                                                        w1.hi.data_ptr(), w1.lo.data_ptr(), w1.scale_log2, bias1.data_ptr(),
                                                        gamma.data_ptr(), beta.data_ptr(), float(eps), w2p.hi.data_ptr(),
                                                        w2p.lo.data_ptr(), w2p.scale_log2, bias2.data_ptr(),
                                                        hyper.data_ptr(), out.data_ptr(), rows, h * w, w, _stream()),
                              "rsp_sam_upscale_fused"))
    return out


SAM_T2I_FOLD_MAX_TOKENS = 12    # 8 heads x T query columns fit the kernel's 96


def sam_t2i_fold(keys, pek, qp, tqx, *, R, N, ncols):
    """token -> image attention over the per-RoI key planes with the K | V projections folded in (csrc/t2i_fold.hip):
    keys Planes [R*N, 256]; pek Planes [N, 128] = k_proj(pe) + bias; qp Planes [R*96, 256] and tqx Planes [R*96, 128] (see
    rsp_sam_t2i_fold).  Returns u fp32 [R*96, 256] = sum_n softmax[n] keys[n] per (RoI, column); the rows of the columns
    >= ncols are zero."""
    lib = _lib.load()
    for t, name, k in ((keys, 'keys', 256), (pek, 'pek', 128), (qp, 'qp', 256), (tqx, 'tqx', 128)):
        if not isinstance(t, Planes) or t.shape[-1] != k or t.f8:
            raise ValueError(f'sam_t2i_fold: {name} must be fp16 planes with {k} columns')
    if qp.rows != tqx.rows or qp.rows < R * 96 or keys.rows < R * N or pek.rows != N:
        raise ValueError('sam_t2i_fold: row counts')
    u = torch.zeros((R * 96, 256), dtype=torch.float32, device=keys.device)
    _timed('sam_t2i_fold_kernel', 2.0 * R * N * 96 * (256 + 128 + 256), 4.0 * R * N * 256,
           lambda: _lib.check(lib.rsp_sam_t2i_fold(keys.hi.data_ptr(), keys.lo.data_ptr(), keys.rows, keys.scale_log2,
                                                   pek.hi.data_ptr(), pek.lo.data_ptr(), pek.scale_log2,
                                                   qp.hi.data_ptr(), qp.lo.data_ptr(), qp.scale_log2,
                                                   tqx.hi.data_ptr(), tqx.lo.data_ptr(), tqx.scale_log2, qp.rows,
                                                   u.data_ptr(), R, N, ncols, _stream()), "rsp_sam_t2i_fold"))
    return u


def sam_fold_expand(tq, R, T, scale):
    """projected token queries tq [R*T, 128] -> Planes [R*96, 128] of the block-diagonal, pre-scaled query (rsp_sam_fold_expand)"""
    lib = _lib.load()
    _chk_f32(tq, 'tq')
    if not tq.is_contiguous() or tuple(tq.shape) != (R * T, 128):
        raise ValueError('sam_fold_expand: tq must be a contiguous [R*T, 128] tensor')
    pl = empty_planes((R * 96, 128), tq.device)
    _timed('sam_fold_expand_kernel', 0, 4.0 * tq.numel() + 4.0 * R * 96 * 128,
           lambda: _lib.check(lib.rsp_sam_fold_expand(tq.data_ptr(), pl.hi.data_ptr(), pl.lo.data_ptr(), pl.word, R, T, float(scale),
                                                      _stream()), 'rsp_sam_fold_expand'))
    return pl


def sam_fold_gather(full, R, T):
    """full [R*96, 128] (v_proj applied to every column of the folded attention's result) -> [R*T, 128]: every column's own head"""
    lib = _lib.load()
    _chk_f32(full, 'full')
    if not full.is_contiguous() or tuple(full.shape) != (R * 96, 128):
        raise ValueError('sam_fold_gather: full must be a contiguous [R*96, 128] tensor')
    ao = torch.empty((R * T, 128), dtype=torch.float32, device=full.device)
    _timed('sam_fold_gather_kernel', 0, 8.0 * ao.numel(),
           lambda: _lib.check(lib.rsp_sam_fold_gather(full.data_ptr(), ao.data_ptr(), R, T, _stream()), 'rsp_sam_fold_gather'))
    return ao


def sam_i2t_attention(q, k, v, *, R, T, N, scale, q_map=None, out=None, out_planes=None):
    """SAM decoder image->token attention. q [Rq*N,128], k/v [R*T,128]; result [R*N,128] into `out` and/or Planes."""
    lib = _lib.load()
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v')):
        _chk_f32(t, n)
        if not t.is_contiguous():
            raise ValueError(f'sam_i2t_attention: {n} must be contiguous')
    hi = lo = 0
    e = 0
    if out_planes is not None:
        hi, lo, e = out_planes.hi.data_ptr(), out_planes.lo.data_ptr(), out_planes.scale_log2
    _timed('sam_i2t_kernel', 4.0 * R * T * N * 128, 4.0 * R * N * 128 * 2,
           lambda: _lib.check(lib.rsp_sam_i2t_attention(q.data_ptr(), _ptr(q_map), k.data_ptr(), v.data_ptr(),
                                                        _ptr(out), hi, lo, e, R, T, N, scale, _stream()),
                              "rsp_sam_i2t_attention"),
           detail=f'R={R} T={T} N={N}')
    return out if out_planes is None else out_planes


def roi_align(feats_nhwc, pes, rois, P, strides, finest_scale=56):
    """feats_nhwc: list of [B,H,W,C]; pes: list of [H,W,C] or None; rois [K,5] -> [K,P,P,C]."""
    lib = _lib.load()
    K = rois.shape[0]
    C = feats_nhwc[0].shape[-1]
    out = torch.empty((K, P, P, C), dtype=torch.float32, device=rois.device)
    if K == 0:
        return out
    d = _lib.RspRoiAlignDesc()
    for i, f in enumerate(feats_nhwc):
        _chk_f32(f, "feat")
        if not f.is_contiguous():
            raise ValueError("roi_align expects contiguous NHWC levels")
        d.feat[i] = f.data_ptr()
        d.pe[i] = 0 if pes is None or pes[i] is None else pes[i].data_ptr()
        d.H[i], d.W[i] = f.shape[1], f.shape[2]
        d.spatial_scale[i] = 1.0 / strides[i]
    d.rois, d.out = rois.data_ptr(), out.data_ptr()
    d.K, d.P, d.C, d.num_levels, d.finest_scale = K, P, C, len(feats_nhwc), finest_scale
    _timed('roi_align_kernel', 0, 4.0 * out.numel(),
           lambda: _lib.check(lib.rsp_roi_align(d, _stream()), "rsp_roi_align"))
    return out


def pool2(x_nhwc, mode):
    lib = _lib.load()
    B, H, W, C = x_nhwc.shape
    Ho, Wo = (H // 2, W // 2) if mode == 0 else ((H + 1) // 2, (W + 1) // 2)
    out = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x_nhwc.device)
    _lib.check(lib.rsp_pool2(x_nhwc.data_ptr(), out.data_ptr(), B, H, W, C, mode, _stream()), "rsp_pool2")
    return out


def add_rows(x, v, vmod=None, out=None):
    """x [rows, C] + v[(row % vmod), C]."""
    lib = _lib.load()
    C = x.shape[-1]
    rows = x.numel() // C
    vmod = v.numel() // C if vmod is None else vmod
    out = torch.empty_like(x) if out is None else out
    _lib.check(lib.rsp_add_rows(x.data_ptr(), v.data_ptr(), out.data_ptr(), rows, C, vmod, _stream()),
               "rsp_add_rows")
    return out


def sincos_pairs(x):
    lib = _lib.load()
    out = torch.empty(x.shape[:-1] + (x.shape[-1] // 2,), dtype=torch.float32, device=x.device)
    _lib.check(lib.rsp_sincos_pairs(x.data_ptr(), out.data_ptr(), out.numel(), _stream()), "rsp_sincos_pairs")
    return out


def gather_rows(src, idx):
    lib = _lib.load()
    C = src.shape[-1]
    out = torch.empty((idx.numel(), C), dtype=torch.float32, device=src.device)
    _lib.check(lib.rsp_gather_rows(src.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), C, _stream()),
               "rsp_gather_rows")
    return out


def hyper_mask(up, hyper):
    """up [R, npix, C], hyper [R, C] -> [R, npix]."""
    lib = _lib.load()
    R, npix, C = up.shape
    out = torch.empty((R, npix), dtype=torch.float32, device=up.device)
    _lib.check(lib.rsp_hyper_mask(up.data_ptr(), hyper.data_ptr(), out.data_ptr(), R, npix, C, _stream()),
               "rsp_hyper_mask")
    return out


def mask_post(low_res, batch_input_shape, crop_hw, out_hw, thr, want_prob=False):
    """low_res [k, h, w] logits -> bool [k, out_h, out_w] (+ optional probabilities)."""
    lib = _lib.load()
    k, h, w = low_res.shape
    out = torch.empty((k, out_hw[0], out_hw[1]), dtype=torch.bool, device=low_res.device)
    prob = torch.empty((k, out_hw[0], out_hw[1]), dtype=torch.float32, device=low_res.device) if want_prob else None
    ws = torch.empty_like(low_res)
    _lib.check(lib.rsp_mask_post(low_res.data_ptr(), ws.data_ptr(), k, h, w, batch_input_shape[0], batch_input_shape[1],
                                 crop_hw[0], crop_hw[1], out_hw[0], out_hw[1], thr, out.data_ptr(), _ptr(prob),
                                 _stream()), "rsp_mask_post")
    return (out, prob) if want_prob else out


def mask_post_logits(low_res, img_shape, crop_hw, out_hw, thr=0.0, want_val=False):
    """SAMDet.predict (models.py:1185-1206): low_res [k, h, w] logits -> bool [k, out_h, out_w] = resized logits > thr."""
    lib = _lib.load()
    k, h, w = low_res.shape
    if not low_res.is_contiguous():
        raise ValueError("mask_post_logits expects contiguous logits")
    out = torch.empty((k, out_hw[0], out_hw[1]), dtype=torch.bool, device=low_res.device)
    val = torch.empty((k, out_hw[0], out_hw[1]), dtype=torch.float32, device=low_res.device) if want_val else None
    _lib.check(lib.rsp_mask_post_logits(low_res.data_ptr(), k, h, w, img_shape[0], img_shape[1], crop_hw[0], crop_hw[1],
                                        out_hw[0], out_hw[1], thr, out.data_ptr(), _ptr(val), _stream()),
               "rsp_mask_post_logits")
    return (out, val) if want_val else out


def resnet_stem(x, w_taps, bias):
    """x [B,3,H,W] fp32 NCHW -> relu(bn(conv7x7 s2 p3)) as NHWC [B,Ho,Wo,64] (resnet.py:640-647)."""
    lib = _lib.load()
    _chk_f32(x, "x")
    if x.dim() != 4 or x.shape[1] != 3 or not x.is_contiguous():
        raise ValueError("resnet_stem expects a contiguous [B,3,H,W] tensor")
    B, _, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, 64), dtype=torch.float32, device=x.device)
    _timed('stem_conv_kernel', 2.0 * B * Ho * Wo * 64 * 147, 4.0 * (x.numel() + y.numel()),
           lambda: _lib.check(lib.rsp_resnet_stem(x.data_ptr(), w_taps.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W,
                                                  _stream()), "rsp_resnet_stem"))
    return y


def maxpool_nhwc(x, k=3, s=2, p=1):
    lib = _lib.load()
    _chk_f32(x, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("maxpool_nhwc expects a contiguous NHWC tensor")
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
    _timed('maxpool_kernel', 0, 4.0 * (x.numel() + y.numel()),
           lambda: _lib.check(lib.rsp_maxpool_nhwc(x.data_ptr(), y.data_ptr(), B, H, W, C, k, s, p, _stream()),
                              "rsp_maxpool_nhwc"))
    return y


def upsample_nearest_add_(dst, src):
    """dst [B,H,W,C] += F.interpolate(src [B,h,w,C], size=(H,W), mode='nearest')  (fpn.py:190-204), in place."""
    lib = _lib.load()
    _chk_f32(dst, "dst"); _chk_f32(src, "src")
    if not (dst.is_contiguous() and src.is_contiguous()) or dst.shape[0] != src.shape[0] or dst.shape[3] != src.shape[3]:
        raise ValueError("upsample_nearest_add_ expects contiguous NHWC tensors of the same batch and channels")
    B, H, W, C = dst.shape
    _timed('upsample_add_kernel', 0, 4.0 * (2 * dst.numel() + src.numel()),
           lambda: _lib.check(lib.rsp_upsample_nearest_add(src.data_ptr(), dst.data_ptr(), B, src.shape[1], src.shape[2], H,
                                                           W, C, _stream()), "rsp_upsample_nearest_add"))
    return dst


def sam_embed_boxes(boxes, gauss, pe_top_left, pe_bottom_right, input_size):
    """boxes [n,4] -> [n, 2, 2F] sparse prompt embeddings (HF SamPromptEncoder._embed_boxes)."""
    lib = _lib.load()
    boxes = boxes.contiguous()
    _chk_f32(boxes, "boxes")
    n, F = boxes.shape[0], gauss.shape[1]
    out = torch.empty((n, 2, 2 * F), dtype=torch.float32, device=boxes.device)
    _lib.check(lib.rsp_sam_embed_boxes(boxes.data_ptr(), gauss.data_ptr(), pe_top_left.data_ptr(),
                                       pe_bottom_right.data_ptr(), out.data_ptr(), n, F, input_size[0], input_size[1],
                                       _stream()), "rsp_sam_embed_boxes")
    return out


def box_coder(coder):
    """The RspBoxCoder of a DeltaXYWHBBoxCoder-like object (`means`, `stds`, `max_ratio`, `clip_border`, `add_ctr_clamp`,
    `ctr_clamp`: delta_xywh_bbox_coder.py:71-131)."""
    c = _lib.RspBoxCoder()
    for i in range(4):
        c.means[i], c.stds[i] = float(coder.means[i]), float(coder.stds[i])
    c.max_ratio, c.ctr_clamp = float(coder.max_ratio), float(coder.ctr_clamp)
    c.clip_border, c.add_ctr_clamp = int(bool(coder.clip_border)), int(bool(coder.add_ctr_clamp))
    return c


class RpnSelector:
    """rpn_topk -> rpn_decode -> batched_nms on the device (rpn_head.py:134-304)."""

    def __init__(self, base_anchors, strides, nms_pre, max_per_img, iou_thr, min_bbox_size, coder, device):
        self.base = base_anchors.to(torch.float32).contiguous().to(device)   # [L, A, 4]
        self.strides = list(strides)
        self.L, self.A = self.base.shape[0], self.base.shape[1]
        self.nms_pre, self.max_per_img, self.iou_thr = nms_pre, max_per_img, iou_thr
        self.min_bbox_size, self.coder = min_bbox_size, box_coder(coder)

    def __call__(self, heads, sizes, ld, img_hw):
        """heads: per-level [B*H*W, ld] outputs; sizes: [(H, W)]; img_hw: device [B, 2] float."""
        lib = _lib.load()
        dev = heads[0].device
        B = img_hw.shape[0]
        d = _lib.RspRpnDesc()
        for i, (hd, (H, W)) in enumerate(zip(heads, sizes)):
            d.head[i] = hd.data_ptr()
            d.H[i], d.W[i], d.stride[i] = H, W, float(self.strides[i])
        d.ld, d.A, d.nms_pre, d.num_levels = ld, self.A, self.nms_pre, len(heads)
        d.base_anchors = self.base.data_ptr()
        d.coder, d.min_bbox_size = self.coder, float(self.min_bbox_size)
        L, k = len(heads), self.nms_pre
        sel_idx = torch.empty((B, L, k), dtype=torch.int32, device=dev)
        sel_score = torch.empty((B, L, k), dtype=torch.float32, device=dev)
        sel_cnt = torch.empty((B, L), dtype=torch.int32, device=dev)
        _lib.check(lib.rsp_rpn_topk(d, B, sel_idx.data_ptr(), sel_score.data_ptr(), sel_cnt.data_ptr(),
                                    _stream()), "rsp_rpn_topk")
        cap = L * k
        cand = _cand_buffers(B, cap, dev)
        _lib.check(lib.rsp_rpn_decode(d, B, sel_idx.data_ptr(), sel_score.data_ptr(), sel_cnt.data_ptr(),
                                      img_hw.data_ptr(), cap, *[c.data_ptr() for c in cand], _stream()),
                   "rsp_rpn_decode")
        return batched_nms(cand, B, cap, self.iou_thr, self.max_per_img)


NMS_LDS_CANDIDATES = 16384     # rsp_batched_nms sorts up to here in LDS (det.hip NMS_LDS_KEYS), in memory above
NMS_MAX_CANDIDATES = 131072    # ... and holds this many candidates per image at most
# the pair mask of the in-memory path is quadratic: B * cap * cap / 8 bytes (200 MB per image at 40 k candidates, 2 GiB at
# the kernel's limit).  A call that would need more than this many bytes of workspace is refused with the figures in the
# message instead of failing inside the allocator (settable: a box with spare HBM may raise it).
NMS_WORKSPACE_LIMIT_BYTES = 8 << 30


def _cand_buffers(B, cap, dev):
    return (torch.empty((B, cap, 4), dtype=torch.float32, device=dev),
            torch.empty((B, cap), dtype=torch.float32, device=dev),
            torch.empty((B, cap), dtype=torch.int32, device=dev),
            torch.empty((B, cap), dtype=torch.int32, device=dev),
            torch.empty((B,), dtype=torch.int32, device=dev))


def batched_nms(cand, B, cap, iou_thr, max_out):
    """cand = (boxes [B,cap,4], scores, ids, src, cnt).  returns dict of [B, max_out] outputs + counts."""
    lib = _lib.load()
    boxes, scores, ids, src, cnt = cand
    dev = boxes.device
    ws_bytes = int(lib.rsp_nms_workspace_bytes(B, cap))
    if ws_bytes > NMS_WORKSPACE_LIMIT_BYTES:
        raise ValueError(f'batched_nms: {B} images x {cap} candidates need a {ws_bytes / 2 ** 30:.1f} GiB pair mask '
                         f'(limit ops.NMS_WORKSPACE_LIMIT_BYTES = {NMS_WORKSPACE_LIMIT_BYTES / 2 ** 30:.0f} GiB): raise score_thr, '
                         'lower the number of proposals, or run fewer images per call')
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    keep = torch.empty((B, max_out), dtype=torch.int32, device=dev)
    keep_cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    ob = torch.empty((B, max_out, 4), dtype=torch.float32, device=dev)
    os_ = torch.empty((B, max_out), dtype=torch.float32, device=dev)
    oi = torch.empty((B, max_out), dtype=torch.int32, device=dev)
    osrc = torch.empty((B, max_out), dtype=torch.int32, device=dev)
    _lib.check(lib.rsp_batched_nms(boxes.data_ptr(), scores.data_ptr(), ids.data_ptr(), src.data_ptr(),
                                   cnt.data_ptr(), B, cap, iou_thr, max_out, ws.data_ptr(), keep.data_ptr(),
                                   keep_cnt.data_ptr(), ob.data_ptr(), os_.data_ptr(), oi.data_ptr(),
                                   osrc.data_ptr(), _stream()), "rsp_batched_nms")
    return dict(boxes=ob, scores=os_, ids=oi, src=osrc, count=keep_cnt, keep=keep, cand_count=cnt)


def bbox_post(head, ld, rois, roi_start, img_hw, num_classes, score_thr, coder, iou_thr, max_out,
              scale_factors=None):
    """R-CNN head post-processing + multiclass NMS (bbox_head.py:476-571, bbox_nms.py:12-105).
    scale_factors: per image (w, h) -- `rescale=True` (bbox_head.py:549-554): the decoded, clipped boxes are divided by the
    scale factor BEFORE the NMS, as the reference does."""
    import ctypes
    lib = _lib.load()
    dev = head.device
    B = img_hw.shape[0]
    n_max = int((roi_start[1:] - roi_start[:-1]).max()) if roi_start.numel() > 1 else 0
    cap = max(n_max * num_classes, 1)
    cand = _cand_buffers(B, cap, dev)
    rs = roi_start.to(device=dev, dtype=torch.int32)
    _lib.check(lib.rsp_bbox_post(head.data_ptr(), ld, rois.data_ptr(), rs.data_ptr(), img_hw.data_ptr(), B,
                                 num_classes, score_thr, ctypes.byref(box_coder(coder)), cap, *[c.data_ptr() for c in cand],
                                 _stream()), "rsp_bbox_post")
    if cap > NMS_LDS_CANDIDATES:
        # many classes (the RSPrompter datasets have 1 or 10: never here): size the NMS by the candidates that passed the
        # score threshold -- one host sync -- instead of by RoIs x classes; its pair mask is quadratic in that size
        cmax = max(int(cand[4].max()), 1)
        if cmax > NMS_MAX_CANDIDATES:
            raise ValueError(f'bbox_post: {cmax} of {n_max} RoIs x {num_classes} classes passed score_thr={score_thr}; '
                             f'rsp_batched_nms holds {NMS_MAX_CANDIDATES} candidates per image (det.hip NMS_MAXW_LARGE)')
        if cmax < cap:
            cand = tuple(c[:, :cmax].contiguous() for c in cand[:4]) + (cand[4],)
            cap = cmax
    if scale_factors is not None:
        for b, sf in enumerate(scale_factors):
            inv = (1 / float(sf[0]), 1 / float(sf[1]))          # bbox_head.py:550: python reciprocal, then an fp32 product
            arr = (ctypes.c_float * 4)(inv[0], inv[1], inv[0], inv[1])
            bx = cand[0][b]
            _lib.check(lib.rsp_scale_boxes(bx.data_ptr(), bx.data_ptr(), bx.shape[0], arr, _stream()), "rsp_scale_boxes")
    return batched_nms(cand, B, cap, iou_thr, max_out)


def div_boxes(boxes, sf4):
    """boxes [k,4] / (sf_w, sf_h, sf_w, sf_h)."""
    import ctypes
    lib = _lib.load()
    boxes = boxes.contiguous()
    out = torch.empty_like(boxes)
    arr = (ctypes.c_float * 4)(*[float(v) for v in sf4])
    _lib.check(lib.rsp_div_boxes(boxes.data_ptr(), out.data_ptr(), boxes.shape[0], arr, _stream()), "rsp_div_boxes")
    return out


def fill_bias_rows(bias, rows, N, out=None, planes=None, c_ncols=0, pl_col0=0):
    """rows (int32 indices) of a GEMM output the GEMM did not write because their A row is zero (window padding): fp32
    columns [0, c_ncols or N) of `out` = bias, plane columns [pl_col0, N) = split(bias)  (HF:913-915: qkv(0) = bias)."""
    lib = _lib.load()
    n = int(rows.shape[0])
    if n == 0:
        return
    _lib.check(lib.rsp_fill_bias_rows(_ptr(bias), rows.data_ptr(), n, N, _ptr(out), out.stride(0) if out is not None else 0,
                                      c_ncols, planes.hi.data_ptr() if planes is not None else None,
                                      planes.lo.data_ptr() if planes is not None else None,
                                      planes.rows if planes is not None else 0, pl_col0,
                                      planes.word if planes is not None else 0, _stream()), "rsp_fill_bias_rows")


def scale_boxes(boxes, f4):
    """boxes [k,4] * (f0, f1, f2, f3) in fp32 (scale_boxes, structures/bbox/transforms.py:391-414)."""
    import ctypes
    lib = _lib.load()
    boxes = boxes.contiguous()
    out = torch.empty_like(boxes)
    arr = (ctypes.c_float * 4)(*[float(v) for v in f4])
    _lib.check(lib.rsp_scale_boxes(boxes.data_ptr(), out.data_ptr(), boxes.shape[0], arr, _stream()), "rsp_scale_boxes")
    return out


def pack_masks(masks):
    """bool [k, H, W] -> uint8 [k, H*W/8] (bit i of byte j = pixel 8j+i)."""
    lib = _lib.load()
    k = masks.shape[0]
    n = masks[0].numel() if k else 0
    if n % 8:
        raise ValueError('mask area must be a multiple of 8')
    out = torch.empty((k, n // 8), dtype=torch.uint8, device=masks.device)
    if k:
        m = masks.contiguous()
        _lib.check(lib.rsp_pack_bits(m.data_ptr(), out.data_ptr(), m.numel(), _stream()), "rsp_pack_bits")
    return out


def mask_rle_counts(masks, cap=4096):
    """bool [k, H, W] -> (counts int32 [k, cap'], n int32 [k]): COCO RLE run lengths (column-major stream, first run
    counts zeros).  Retries with a larger capacity when a mask has more runs than `cap`."""
    lib = _lib.load()
    k, H, W = masks.shape
    m = masks.contiguous()
    dev = masks.device
    while True:
        counts = torch.empty((k, cap), dtype=torch.int32, device=dev)
        ws = torch.empty((k, cap), dtype=torch.int32, device=dev)
        n = torch.empty((k,), dtype=torch.int32, device=dev)
        if k:
            _lib.check(lib.rsp_mask_rle(m.data_ptr(), k, H, W, ws.data_ptr(), counts.data_ptr(), n.data_ptr(), cap,
                                        _stream()), "rsp_mask_rle")
            need = int((-n).max().item())
            if need > 0:
                cap = 1 << (need - 1).bit_length()
                continue
        return counts, n


def mask_rle_into(masks, counts, ws, n):
    """rsp_mask_rle into rows of preallocated buffers (counts / ws int32 [k, cap], n int32 [k]); nothing here touches
    the host: n[i] < 0 reports a mask with more than cap runs (-needed), checked by whoever reads n later."""
    lib = _lib.load()
    k, H, W = masks.shape
    if k:
        m = masks.contiguous()
        _lib.check(lib.rsp_mask_rle(m.data_ptr(), k, H, W, ws.data_ptr(), counts.data_ptr(), n.data_ptr(),
                                    counts.shape[1], _stream()), "rsp_mask_rle")


def rle_to_string(counts, n, k, flat_cap):
    """COCO ASCII strings of k run-length lists (rsp_rle_to_string): returns (lens int32 [k], offs int64 [k + 1],
    flat uint8 [flat_cap]) on the device, no host synchronisation; strings beyond flat_cap are not written
    (offs[k] > flat_cap tells)."""
    lib = _lib.load()
    dev = counts.device
    lens = torch.zeros((max(k, 1),), dtype=torch.int32, device=dev)
    offs = torch.zeros((k + 1,), dtype=torch.int64, device=dev)
    flat = torch.empty((max(int(flat_cap), 1),), dtype=torch.uint8, device=dev)
    _lib.check(lib.rsp_rle_to_string(counts.data_ptr(), n.data_ptr(), k, counts.shape[1], lens.data_ptr(),
                                     offs.data_ptr(), flat.data_ptr(), int(flat_cap), _stream()), "rsp_rle_to_string")
    return lens, offs, flat


# ----------------------------------------------------------------------------- query prompter ops
def groupnorm(x, gamma, beta, groups, eps=1e-5, relu=False, add=None):
    """GroupNorm on channels-last [B, ..., C]; `add` (same shape) is added after the norm."""
    lib = _lib.load()
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    ws = torch.empty((B * groups * 2,), dtype=torch.float64, device=x.device)
    _timed('groupnorm_kernels', 0, 12.0 * x.numel(),
           lambda: _lib.check(lib.rsp_groupnorm_nhwc(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(add),
                                                     y.data_ptr(), ws.data_ptr(), B, HW, C, groups, eps,
                                                     1 if relu else 0, _stream()), "rsp_groupnorm_nhwc"))
    return y


def resize_bilinear(x_nhwc, size):
    lib = _lib.load()
    B, H, W, C = x_nhwc.shape
    y = torch.empty((B, size[0], size[1], C), dtype=torch.float32, device=x_nhwc.device)
    _lib.check(lib.rsp_resize_bilinear_nhwc(x_nhwc.data_ptr(), y.data_ptr(), B, H, W, size[0], size[1], C, _stream()),
               "rsp_resize_bilinear_nhwc")
    return y


def msdeform_attn(value, offs_weights, ref_points, B, Ntok, level_hw, head_dim=16):
    """value [B*Ntok, 8*head_dim] -> sampled [B*Ntok, 8*head_dim] (before output_proj); head_dim 16 | 32"""
    import ctypes
    lib = _lib.load()
    if value.shape[-1] != 8 * head_dim:
        raise ValueError(f'msdeform_attn: value rows of {value.shape[-1]} channels, expected 8 x {head_dim}')
    out = torch.empty((B * Ntok, 8 * head_dim), dtype=torch.float32, device=value.device)
    arr = (ctypes.c_int * (2 * len(level_hw)))(*[int(v) for hw in level_hw for v in hw])
    _timed('msda_kernel', 0, 4.0 * out.numel() * 50,
           lambda: _lib.check(lib.rsp_msdeform_attn_ex(value.data_ptr(), offs_weights.data_ptr(), offs_weights.shape[-1],
                                                       ref_points.data_ptr(), out.data_ptr(), B, Ntok, len(level_hw), arr,
                                                       head_dim, _stream()), "rsp_msdeform_attn_ex"))
    return out


def query_attn_mask(mask_pred_plus, size):
    """mask_pred_plus [B, Nq, Hs, Ws] -> uint8 [B, Nq, h*w] (1 = blocked), fully blocked rows cleared."""
    lib = _lib.load()
    B, Nq, Hs, Ws = mask_pred_plus.shape
    m = torch.empty((B, Nq, size[0] * size[1]), dtype=torch.uint8, device=mask_pred_plus.device)
    _lib.check(lib.rsp_query_attn_mask(mask_pred_plus.data_ptr(), m.data_ptr(), B * Nq, Hs, Ws, size[0], size[1],
                                       _stream()), "rsp_query_attn_mask")
    return m


def sam_mask_embed(mask_pred_plus, emb_rows, roi_img, prm, he, we, eps=1e-6):
    """mask_pred_plus [R, 4he, 4we]; emb_rows [B*he*we, C]; prm: dict of the SamMaskEmbedding tensors."""
    lib = _lib.load()
    R = mask_pred_plus.shape[0]
    C = emb_rows.shape[-1]
    out = torch.empty((R * he * we, C), dtype=torch.float32, device=emb_rows.device)
    d = _lib.RspMaskEmbedDesc()
    d.mask_pred_plus, d.image_embeddings, d.roi_img = mask_pred_plus.data_ptr(), emb_rows.data_ptr(), roi_img.data_ptr()
    for k in ('conv1_w', 'conv1_b', 'ln1_w', 'ln1_b', 'conv2_w', 'conv2_b', 'ln2_w', 'ln2_b', 'conv3_w', 'conv3_b'):
        setattr(d, k, prm[k].data_ptr())
    d.out, d.R, d.he, d.we, d.C, d.eps = out.data_ptr(), R, he, we, C, eps
    _timed('mask_embed_kernel', 0, 4.0 * out.numel(),
           lambda: _lib.check(lib.rsp_sam_mask_embed(d, _stream()), "rsp_sam_mask_embed"))
    return out


def query_topk(cls, k):
    """cls [B, Nq, nc+1] logits -> (scores [B,k], flat index [B,k] into Nq*nc), score desc / index asc."""
    lib = _lib.load()
    B, Nq, nc1 = cls.shape
    sc = torch.empty((B, k), dtype=torch.float32, device=cls.device)
    fl = torch.empty((B, k), dtype=torch.int32, device=cls.device)
    _lib.check(lib.rsp_query_topk(cls.data_ptr(), B, Nq, nc1 - 1, k, sc.data_ptr(), fl.data_ptr(), _stream()),
               "rsp_query_topk")
    return sc, fl


def query_mask_post(low_res, qidx, cls_score, batch_input_shape, crop_hw, out_hw, want_logits=False):
    """low_res [Nq, h, w] logits of one image; qidx int32 [k]; returns (masks bool [k,oh,ow], det_scores, bboxes)."""
    lib = _lib.load()
    k = qidx.numel()
    _, h, w = low_res.shape
    dev = low_res.device
    masks = torch.empty((k, out_hw[0], out_hw[1]), dtype=torch.bool, device=dev)
    logits = torch.empty((k, out_hw[0], out_hw[1]), dtype=torch.float32, device=dev) if want_logits else None
    det = torch.empty((k,), dtype=torch.float32, device=dev)
    boxes = torch.empty((k, 4), dtype=torch.float32, device=dev)
    ws = torch.empty((max(k, 1) * 32,), dtype=torch.uint8, device=dev)
    _timed('query_mask_kernel', 0, 1.0 * masks.numel(),
           lambda: _lib.check(lib.rsp_query_mask_post(low_res.data_ptr(), qidx.data_ptr(), cls_score.data_ptr(), k, h, w,
                                                      batch_input_shape[0], batch_input_shape[1], crop_hw[0], crop_hw[1],
                                                      out_hw[0], out_hw[1], ws.data_ptr(), masks.data_ptr(), _ptr(logits),
                                                      det.data_ptr(), boxes.data_ptr(), _stream()), "rsp_query_mask_post"))
    return (masks, det, boxes, logits) if want_logits else (masks, det, boxes)
