"""The arithmetic of the hi/lo fp16 split (csrc/conv_f16.hip, PREC 2) restated on the CPU: what the three-MFMA scheme can
and cannot represent, independent of the hardware.  Products of fp16 values are exact in fp32 and the kernels accumulate
in fp32, so the scheme's own error is the REPRESENTATION error of the two planes plus the dropped lo.lo term; this file pins
the bounds DESIGN.md section 2 states:

    activations (no scale):   |x - hi - lo| <= max(2^-22 |x|, 2^-25)            for |x| <= 65504
    weights (scaled by 2^e):  |w s - hi - lo| <= max(2^-22 |w s|, 2^-25),  max |w| s in [2^13, 2^14)
    a dot product:            |sum x w - sum (hi.hi + hi.lo + lo.hi) / s| <= sum (|dx| |w| + |x| |dw| + |lo_x lo_w| / s)

and the two consequences the tests on the GPU rely on: for activations of ordinary magnitude the error is fp32-round-off class
(a few 2^-22 of sum |x w|), and an activation beyond the fp16 range turns hi into inf (the fall-back trigger)."""
import math

import pytest
import torch

from deva.hip import ops


def _split(v):
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    return hi, lo


@pytest.mark.parametrize('scale', [1e-6, 1e-3, 1.0, 37.0, 6.0e4])
def test_activation_planes_represent_x_to_2pow22_or_the_subnormal_floor(scale):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(200000, generator=g) * scale).clamp(-65504.0, 65504.0)
    hi, lo = _split(x)
    err = (x.double() - hi.double() - lo.double()).abs()
    bound = torch.maximum(x.double().abs() * 2.0**-22, torch.tensor(2.0**-25, dtype=torch.float64))
    assert bool((err <= bound).all()), float((err / bound).max())
    assert bool(torch.isfinite(hi.float()).all() and torch.isfinite(lo.float()).all())


def test_an_activation_beyond_the_fp16_range_poisons_hi():
    """|x| > 65504 (after rounding: >= 65520) and non-finite values make hi inf / NaN: the product with any weight, zero
    included, is then inf / NaN and reaches the accumulator -- what the kernel's fall-back flag tests"""
    x = torch.tensor([65519.9, 65520.0, 7.0e4, -1.0e6, float('inf'), float('nan')])
    hi, _ = _split(x)
    assert bool(torch.isfinite(hi[:1].float()).all())            # rounds down to 65504: still representable
    assert not bool(torch.isfinite(hi[1:].float()).any())
    assert not bool(torch.isfinite(hi[1:].float() * 0.0).any())  # a zero weight does not hide it


@pytest.mark.parametrize('cout,cin,k,gain', [(64, 64, 1, 0.05), (32, 96, 3, 2.0), (48, 513, 1, 1e-3)])
def test_weight_planes_and_scale(cout, cin, k, gain):
    g = torch.Generator().manual_seed(cout + cin)
    w = torch.randn(cout, cin, k, k, generator=g) * gain
    w.view(-1)[::9] *= 1e-5
    planes, e = ops.pack_split(w)
    wmax = float(w.abs().max())
    assert 2.0**13 <= wmax * 2.0**e < 2.0**14
    cpad = (cout + 31) // 32 * 32
    pl = planes.view(-1, 2, cpad, 8).double()
    rec = (pl[:, 0] + pl[:, 1]).permute(0, 2, 1).reshape(-1, cpad)[:k * k * cin, :cout]  # [K][cout], K order of the kernels
    taps = k * k
    ref = (w.reshape(cout, cin // 32, 32, taps).permute(1, 3, 2, 0).reshape(-1, cout) if taps > 1 else w.reshape(cout, cin).t()).double() * 2.0**e
    err = (rec - ref).abs()
    assert bool((err <= torch.maximum(ref.abs() * 2.0**-22, torch.tensor(2.0**-25, dtype=torch.float64))).all())
    # in units of the layer's largest weight: 2^-22 at worst, and the floor of the small weights is 2^-25 / 2^13 = 2^-38
    assert float(err.max()) / (wmax * 2.0**e) <= 2.0**-22
    small = ref.abs() < 2.0**-3
    assert bool(small.any()) and float(err[small].max()) / (wmax * 2.0**e) <= 2.0**-38


@pytest.mark.parametrize('in_scale,K', [(1.0, 4608), (0.05, 1024), (300.0, 512), (1e-4, 2304)])
def test_dot_products_of_the_three_term_scheme(in_scale, K):
    """hi.hi + hi.lo + lo.hi in exact arithmetic (fp64 here; the kernels' products are exact in fp32, their sums fp32)
    against the exact dot product: inside the stated bound, and -- for inputs of ordinary magnitude -- within a few 2^-22
    of sum |x w| (the class of an fp32 accumulation's own round-off)"""
    g = torch.Generator().manual_seed(K)
    n = 256
    x = torch.randn(n, K, generator=g) * in_scale
    w = torch.randn(K, generator=g) * (2.0 / K)**0.5
    e = 14 - math.frexp(float(w.abs().max()))[1]
    ws = torch.ldexp(w, torch.tensor(e, dtype=torch.int32))
    xh, xl = (t.double() for t in _split(x))
    wh, wl = (t.double() for t in _split(ws))
    got = ((xh * wh).sum(1) + (xh * wl).sum(1) + (xl * wh).sum(1)) * 2.0**-e
    exact = (x.double() * w.double()).sum(1)
    dx = (x.double() - xh - xl).abs()
    dw = (ws.double() - wh - wl).abs() * 2.0**-e
    bound = (dx * w.double().abs() + x.double().abs() * dw + (xl * wl).abs() * 2.0**-e).sum(1)
    err = (got - exact).abs()
    assert bool((err <= bound * (1 + 1e-9) + 1e-300).all())
    mag = (x.double().abs() * w.double().abs()).sum(1)
    rel = float((err / mag).max())
    if in_scale >= 0.05:
        assert rel <= 4 * 2.0**-22, rel        # fp32-round-off class
    else:
        # tiny activations throughout (|x| ~ 1e-4 << 2^-14): the 2^-25 floor of the lo plane shows, as DESIGN.md says
        assert 2.0**-22 < rel <= 2.0**-25 / (in_scale * 0.5), rel
