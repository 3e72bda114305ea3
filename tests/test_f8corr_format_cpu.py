"""CPU: the numerics of the opt-in fp8-corrected product (include/rsp_hip.h "Plane format word")
emulated in torch -- the same emulation the GPU unit test holds the kernel to (tests/test_gpu_kernels.py) -- so that the
format decisions (e4m3 for both correction operands, static storage scales 2^5 / 2^-7 shared by activation and weight
planes) are pinned by something that runs without a GPU."""
import math

import torch


def parts(x, e):
    xs = x.double() * 2.0 ** e
    hi = xs.float().clamp(-65504, 65504).half().double()
    lo = xs - hi
    lo8 = (lo * 32).float().clamp(-448, 448).to(torch.float8_e4m3fn).double() / 32
    hi8 = (hi / 128).float().clamp(-448, 448).to(torch.float8_e4m3fn).double() * 128
    return hi, lo, lo8, hi8


def weight_exp(w):
    return int(math.floor(math.log2(16384.0 / float(w.abs().max()))))      # ops.PackedWeight


def test_error_class_and_scale_windows():
    g = torch.Generator().manual_seed(0)
    K = 1280
    a = torch.randn(256, K, generator=g) * torch.exp(torch.randn(256, 1, generator=g))     # LayerNorm-like rows, spread
    a[3, 5], a[7, 100] = 1500.0, -3000.0                                                    # outlier activations
    w = torch.randn(192, K, generator=g) / K ** 0.5
    ea, ew = 2, weight_exp(w)
    ah, al, al8, ah8 = parts(a, ea)
    wh, wl, wl8, wh8 = parts(w, ew)
    exact = a.double() @ w.double().t()
    x3 = (ah @ wh.t() + al @ wh.t() + ah @ wl.t()) * 2.0 ** -(ea + ew)          # what the fp16x3 kernel sums (lo.lo dropped)
    f8 = (ah @ wh.t() + al8 @ wh8.t() + ah8 @ wl8.t()) * 2.0 ** -(ea + ew)
    mag = a.double().abs() @ w.double().abs().t()
    e3, e8 = float(((x3 - exact).abs() / mag).max()), float(((f8 - exact).abs() / mag).max())
    assert e3 < 2.0 ** -21                      # 2^-22 class
    assert 2.0 ** -19 < e8 < 2.0 ** -13         # 2^-15 ... 2^-16 class: 8 bits short of fp16x3, 4 better than one fp16 pass
    # the static windows: nothing that matters saturates or flushes
    assert float(ah8.abs().max()) <= 448 * 128 and float((ah8 - ah).abs().max()) <= float(ah.abs().max()) * 2.0 ** -4
    big = ah.abs() >= 2.0                        # elements above hi8's normal range floor (2^-6 * 2^7)
    assert float(((ah8 - ah).abs() / ah.abs())[big].max()) <= 2.0 ** -4
    assert float(al.abs().max()) * 32 <= 448                               # lo of |a| 2^ea < 28672 fits e4m3 after the 2^5 scale
    typical = al.abs() >= 2.0 ** -11                                        # lo8's normal range floor (2^-6 / 2^5)
    assert float(((al8 - al).abs() / al.abs())[typical].max()) <= 2.0 ** -4
    # weights: |w|max sits at 2^13..2^14 after PackedWeight's scale: hi8 <= 128 < 448, lo <= 8 -> lo8 <= 256 < 448
    assert float(wh.abs().max()) / 128 <= 448 and float(wl.abs().max()) * 32 <= 448


def test_one_fp16_pass_is_not_enough_but_fp8_correction_is():
    """the budget argument of DESIGN 3 in one number: dropping the correction costs 2^-12, the fp8 correction 2^-16."""
    g = torch.Generator().manual_seed(1)
    a, w = torch.randn(128, 1024, generator=g), torch.randn(128, 1024, generator=g) / 32
    ah, al, al8, ah8 = parts(a, 2)
    wh, wl, wl8, wh8 = parts(w, weight_exp(w))
    sc = 2.0 ** -(2 + weight_exp(w))
    exact = a.double() @ w.double().t()
    rms = float(exact.pow(2).mean().sqrt())
    e1 = float(((ah @ wh.t()) * sc - exact).pow(2).mean().sqrt()) / rms
    e8 = float(((ah @ wh.t() + al8 @ wh8.t() + ah8 @ wl8.t()) * sc - exact).pow(2).mean().sqrt()) / rms
    assert e1 > 8 * e8 and e8 < 2.0 ** -15
