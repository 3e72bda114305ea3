"""CPU: the GELU of the HIP kernels (rsprompter_amd/csrc/rsp_common.h, rsp_gelu / rsp_gelu4) restated in numpy with the
header's own coefficients, operation by operation in fp32 (an fma is one rounding), against the fp64 definition
x * Phi(x) of nn.GELU (HF "gelu"; reference: `modeling_sam.py` SamMLPBlock / SamMaskDecoder upscaler activations).
The kernels' form is  max(x, 0) - 0.5 |x| 2^-P(min(|x|, 13.5)),  P(u) = u Q(u): what has to hold is (a) the error bound the
header quotes, (b) P monotone on the clamp interval so that the tail cannot turn around, (c) an exactly zero tail."""
import os
import re

import numpy as np
import pytest
from scipy.special import ndtr

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rsprompter_amd', 'csrc', 'rsp_common.h')
f32 = np.float32


def _coefficients():
    src = open(HDR).read()
    c = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r'#define RSP_GELU_C(\d) (\S+?)f\n', src)}
    u_max = float(re.search(r'#define RSP_GELU_U_MAX (\S+?)f\n', src).group(1))
    assert sorted(c) == list(range(1, 9))
    return [c[k] for k in range(1, 9)], u_max


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def gelu_f32(x):
    C, u_max = _coefficients()
    x = x.astype(f32)
    ax = np.abs(x)
    u = np.minimum(ax, f32(u_max))
    q = _fma(np.full_like(u, f32(C[7])), u, np.full_like(u, f32(C[6])))
    for k in range(5, -1, -1):
        q = _fma(q, u, np.full_like(u, f32(C[k])))
    pu = (q * u).astype(f32)
    with np.errstate(under='ignore'):
        e = np.exp2(-pu.astype(np.float64)).astype(f32)
    e = np.where(np.abs(e) < np.finfo(f32).tiny, f32(0), e)        # the device flushes denormal results
    with np.errstate(invalid='ignore'):
        rl = (f32(0.5) * x + f32(0.5) * ax).astype(f32)         # max(x, 0) for finite x, NaN for NaN (and for -inf, as torch)
    return _fma((f32(-0.5) * u).astype(f32), e, rl)


def test_gelu_error_bound_against_fp64():
    rng = np.random.default_rng(7)
    x = np.concatenate([np.linspace(-16, 16, 2_000_001), rng.normal(size=500_000) * 2.0]).astype(f32)
    ref = x.astype(np.float64) * ndtr(x.astype(np.float64))
    err = np.abs(gelu_f32(x).astype(np.float64) - ref)
    assert err.max() < 3.0e-7                                  # half an ulp of a result near 4
    assert err[np.abs(x) < 3].max() < 2.0e-7
    assert (err / np.maximum(np.abs(ref), 1e-3)).max() < 5e-5    # small negative results keep ~4 digits


def test_gelu_exponent_polynomial_is_monotone_and_tail_is_zero():
    C, u_max = _coefficients()
    u = np.linspace(0, u_max, 200_001)
    P = sum(c * u ** (k + 1) for k, c in enumerate(C))
    assert np.diff(P).min() > 0
    assert P[-1] > 150                                          # 2^-P underflows: erfc tail exactly 0
    big = np.array([-13.5, -14, -100, -1e4, -1e30, 13.5, 100, 1e30, 0.0, -0.0], dtype=f32)
    got = gelu_f32(big)
    assert np.array_equal(got[:5], np.zeros(5, f32)) and np.array_equal(got[5:8], big[5:8]) and np.all(got[8:] == 0)


def test_gelu_propagates_nan_and_equals_the_max_form_on_finite_inputs():
    """fmaxf(NaN, 0) = 0 used to turn a NaN activation into 0; 0.5 x + 0.5 |x| keeps it and has the bits of max(x, 0)
    for every finite x"""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.normal(size=200_000) * 4, [0.0, -0.0, 1e-30, -1e-30, 3.4e38, -3.4e38, 65504.0, -65504.0]]).astype(f32)
    assert np.array_equal((f32(0.5) * x + f32(0.5) * np.abs(x)).astype(f32), np.maximum(x, f32(0)))
    got = gelu_f32(np.array([np.nan, np.inf, 1.0], dtype=f32))
    assert np.isnan(got[0]) and got[1] == np.inf and abs(got[2] - 0.8413447) < 1e-6


@pytest.mark.parametrize('n', [64])
def test_gelu_matches_torch_on_activations(n):
    torch = pytest.importorskip('torch')
    x = torch.randn(n, 1024, generator=torch.Generator().manual_seed(3)) * 3
    got = torch.from_numpy(gelu_f32(x.numpy())).double()
    ref = torch.nn.functional.gelu(x.double())
    assert float(((got - ref).abs() / ref.abs().clamp(min=1.0)).max()) < 1.5e-7      # ~ one fp32 rounding of the result
