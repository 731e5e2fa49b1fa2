"""deva_conv2d (implicit-GEMM fp32 MFMA) against F.conv2d on the CPU: every geometry the network
uses plus ragged / tiny / padded edge cases.  Tolerance: fp32 accumulation-order noise only."""
import os
import zlib

import pytest
import torch

import emu_ops
from deva.hip import ops
from gpu_util import dev, max_err, rand, rel_err, to_dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

# (name, c0, c1, cout, k, stride, pad, batch, H, W, b0_bcast, relu_in, residual('none'|'full'|'bcast'), act, bias, bn)
CASES = [
    ('stem7x7_rgb', 3, 0, 64, 7, 2, 3, 1, 70, 94, False, False, 'none', ops.ACT_RELU, False, True),
    ('stem7x7_rgb+mask', 3, 1, 64, 7, 2, 3, 3, 64, 96, True, False, 'none', ops.ACT_NONE, False, True),
    ('bottleneck1x1', 64, 0, 256, 1, 1, 0, 1, 24, 32, False, False, 'full', ops.ACT_RELU, False, True),
    ('bottleneck3x3s2', 128, 0, 128, 3, 2, 1, 1, 24, 32, False, False, 'none', ops.ACT_RELU, False, True),
    ('down1x1s2', 256, 0, 512, 1, 2, 0, 1, 24, 32, False, False, 'none', ops.ACT_NONE, False, True),
    ('proj1x1_bias', 1024, 0, 512, 1, 1, 0, 1, 6, 8, False, False, 'none', ops.ACT_NONE, True, False),
    ('key3x3_64', 512, 0, 64, 3, 1, 1, 1, 6, 8, False, False, 'none', ops.ACT_NONE, True, False),
    ('shrinkage_cout1', 512, 0, 1, 3, 1, 1, 1, 6, 8, False, False, 'none', ops.ACT_SQUARE_PLUS_ONE, True, False),
    ('selection_sigmoid', 512, 0, 64, 3, 1, 1, 1, 6, 8, False, False, 'none', ops.ACT_SIGMOID, True, False),
    ('fuse_cat_relu_in', 512, 256, 512, 3, 1, 1, 2, 6, 8, True, True, 'none', ops.ACT_NONE, True, False),
    ('fuse_cat_1x1', 512, 512, 512, 1, 1, 0, 3, 6, 8, True, False, 'none', ops.ACT_NONE, True, False),
    ('resblock_conv2_res', 512, 0, 512, 3, 1, 1, 2, 6, 8, False, True, 'full', ops.ACT_NONE, True, False),
    ('compress_513', 512, 1, 512, 1, 1, 0, 2, 6, 8, False, False, 'full', ops.ACT_NONE, True, False),
    ('g4_257_res', 256, 1, 512, 1, 1, 0, 2, 6, 8, False, False, 'full', ops.ACT_NONE, True, False),
    ('gru_transform', 512, 512, 1536, 3, 1, 1, 2, 6, 8, False, False, 'none', ops.ACT_NONE, True, False),
    ('up8_4_256', 256, 0, 256, 3, 1, 1, 2, 24, 32, False, True, 'full', ops.ACT_NONE, True, False),
    ('pred_cout1', 256, 0, 1, 3, 1, 1, 2, 24, 32, False, True, 'none', ops.ACT_NONE, True, False),
    ('cbam_gate7x7', 2, 0, 1, 7, 1, 3, 2, 6, 8, False, False, 'none', ops.ACT_NONE, True, False),
    ('skip_bcast_res', 256, 0, 256, 1, 1, 0, 3, 12, 16, False, False, 'bcast', ops.ACT_NONE, True, False),
    ('ragged_all', 37, 5, 45, 3, 1, 1, 2, 7, 9, False, True, 'full', ops.ACT_RELU, True, False),
    ('one_pixel', 240, 0, 32, 1, 1, 0, 1, 1, 1, False, False, 'none', ops.ACT_NONE, False, False),
    ('gemm_as_conv', 240, 0, 32, 1, 1, 0, 1, 1, 512, False, False, 'none', ops.ACT_NONE, False, False),
    ('big_tile_128', 256, 0, 256, 3, 1, 1, 4, 60, 108, False, False, 'none', ops.ACT_RELU, True, False),
]


CASES += [
    ('vec_rowwrap_3x3', 64, 0, 64, 3, 1, 1, 2, 6, 10, False, True, 'full', ops.ACT_NONE, True, False),
    ('vec_rowwrap_cat', 32, 32, 96, 3, 1, 1, 3, 10, 6, True, False, 'none', ops.ACT_RELU, True, False),
    ('vec_1x1_tail513', 512, 1, 64, 1, 1, 0, 2, 6, 10, False, False, 'none', ops.ACT_NONE, True, False),
    ('vec_7x7_same', 32, 0, 40, 7, 1, 3, 1, 12, 20, False, False, 'none', ops.ACT_NONE, True, False),
]


# the output stage through LDS (conv_epilogue.h: conv_store_tile_vec; rows of 4k pixels, aligned tensors): ragged cout,
# pixel counts that end inside a tile, broadcast / per-image residual, every activation
CASES += [
    ('vecout_cout72_bcast_sigmoid', 64, 0, 72, 1, 1, 0, 3, 6, 10, False, False, 'bcast', ops.ACT_SIGMOID, True, False),
    ('vecout_cout200_res_relu', 128, 0, 200, 3, 1, 1, 2, 20, 36, False, True, 'full', ops.ACT_RELU, True, False),
    ('vecout_cout136_sq1', 96, 32, 136, 3, 1, 1, 1, 34, 44, True, False, 'none', ops.ACT_SQUARE_PLUS_ONE, True, False),
]

# the grouped tile order (conv_epilogue.h: conv_tile_coords) with a last group that is not full: 3 cout tiles of 128 in
# groups of 2 (3x3, 192 tiles), 5 cout tiles of 64 in groups of 3 (1x1, 80 tiles) -- a tile visited twice or never shows
CASES += [
    ('tile_groups_3x3_3of2', 32, 0, 384, 3, 1, 1, 2, 64, 64, False, False, 'none', ops.ACT_NONE, True, False),
    ('tile_groups_1x1_5of3', 64, 0, 320, 1, 1, 0, 1, 32, 32, False, False, 'full', ops.ACT_RELU, True, False),
]


# single output channel on a large guard-banded map: the row-reusing VALU kernel (conv_cout1.hip); the
# unguarded variants of the same cases run on the MFMA tile
CASES += [
    ('pred_rows_big', 96, 0, 1, 3, 1, 1, 2, 96, 128, False, True, 'none', ops.ACT_NONE, True, False),
    ('pred_rows_cat_res', 32, 40, 1, 3, 1, 1, 3, 76, 84, True, False, 'full', ops.ACT_SIGMOID, True, False),
]


def _guarded(x):
    """the tensor inside a NaN-filled buffer with ops.GUARD floats on either side: what ops._alloc
    returns, with poison in the guard bands so that any unmasked over-read shows up as NaN"""
    if x is None:
        return None
    flat = torch.full((x.numel() + 2 * ops.GUARD,), float('nan'), device=dev())
    view = flat[ops.GUARD:ops.GUARD + x.numel()].view(*x.shape)
    view.copy_(x)
    return view


@pytest.mark.parametrize('guarded', [False, True], ids=['scalar_gather', 'vector_gather'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_matches_cpu(case, guarded):
    name, c0, c1, cout, k, stride, pad, batch, H, W, bcast0, relu_in, res, act, bias, bn = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
    cin = c0 + c1
    w = rand(g, cout, cin, k, k, scale=(2.0 / (cin * k * k))**0.5)
    b = rand(g, cout, scale=0.1) if bias else None
    bn_p = None
    if bn:
        bn_p = (torch.rand(cout, generator=g) + 0.5, rand(g, cout, scale=0.1), rand(g, cout, scale=0.1),
                torch.rand(cout, generator=g) + 0.5, 1e-5)
    pc = ops.pack_conv(w, b, bn_p)
    x0 = rand(g, 1 if bcast0 else batch, c0, H, W)
    x1 = rand(g, batch, c1, H, W) if c1 else None
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    residual = None
    if res == 'full':
        residual = rand(g, batch, cout, oh, ow)
    elif res == 'bcast':
        residual = rand(g, 1, cout, oh, ow)
        x0 = rand(g, batch, c0, H, W)
    want = emu_ops.conv2d(pc, x0, x1, stride=stride, pad=pad, relu_in=relu_in, residual=residual, act=act)
    dx0, dx1 = (_guarded(x0), _guarded(x1)) if guarded else (to_dev(x0), to_dev(x1))
    got = ops.conv2d(to_dev(pc), dx0, dx1, stride=stride, pad=pad, relu_in=relu_in,
                     residual=to_dev(residual), act=act)
    torch.cuda.synchronize()
    assert got.shape == want.shape
    err = max_err(got, want)
    assert not torch.isnan(got).any(), f'{name}: guard-band values leaked into the result'
    print(f'{name}: max abs err {err:.3e} (|ref|max {want.abs().max().item():.3e})')
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), (name, err)


# --amp: fp16 operands / fp32 accumulation (csrc/conv_f16.hip) against F.conv2d on the inputs and weights rounded to fp16
# (tests/emu_ops.py: products of fp16 values are exact in fp32, so only the accumulation order differs).  The last
# three cases are shapes the fp16 kernels do not take: they must run exact fp32 although amp is requested.
# (name, c0, c1, cout, k, stride, batch, H, W, bcast0, relu_in, residual, act, takes)
AMP_CASES = [
    ('amp_1x1_small', 64, 0, 64, 1, 1, 1, 6, 8, False, False, 'none', ops.ACT_NONE, True),
    ('amp_1x1_res_relu', 256, 0, 1024, 1, 1, 1, 30, 54, False, False, 'full', ops.ACT_RELU, True),
    ('amp_1x1_cat_bcast', 512, 512, 512, 1, 1, 3, 6, 8, True, False, 'none', ops.ACT_NONE, True),
    ('amp_3x3_rowwrap', 64, 0, 64, 3, 1, 2, 6, 10, False, True, 'full', ops.ACT_NONE, True),
    ('amp_3x3_cat_relu_in', 512, 256 + 64, 512, 3, 1, 2, 6, 8, True, True, 'none', ops.ACT_NONE, True),
    ('amp_3x3_gru', 512, 512, 1536, 3, 1, 2, 6, 8, False, False, 'none', ops.ACT_NONE, True),
    ('amp_3x3_big_tile', 256, 0, 256, 3, 1, 4, 60, 108, False, True, 'full', ops.ACT_NONE, True),
    ('amp_3x3_splitk', 256, 0, 256, 3, 1, 1, 30, 54, False, False, 'none', ops.ACT_RELU, True),
    ('amp_1x1_splitk', 1024, 0, 256, 1, 1, 1, 30, 54, False, False, 'none', ops.ACT_RELU, True),
    ('amp_not_taken_stride2', 128, 0, 128, 3, 2, 1, 24, 32, False, False, 'none', ops.ACT_RELU, False),
    ('amp_not_taken_513', 512, 1, 512, 1, 1, 2, 6, 8, False, False, 'full', ops.ACT_NONE, False),
    ('amp_not_taken_odd_map', 64, 0, 64, 3, 1, 1, 5, 7, False, False, 'none', ops.ACT_NONE, False),
]


@pytest.mark.parametrize('case', AMP_CASES, ids=[c[0] for c in AMP_CASES])
def test_conv_amp_matches_fp16_rounded_cpu(case):
    name, c0, c1, cout, k, stride, batch, H, W, bcast0, relu_in, res, act, takes = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
    cin, pad = c0 + c1, k // 2
    w = rand(g, cout, cin, k, k, scale=(2.0 / (cin * k * k))**0.5)
    b = rand(g, cout, scale=0.1)
    pc = ops.pack_conv(w, b, None, amp=True)
    x0 = rand(g, 1 if bcast0 else batch, c0, H, W)
    x1 = rand(g, batch, c1, H, W) if c1 else None
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    residual = rand(g, batch, cout, oh, ow) if res == 'full' else None
    assert emu_ops.amp_takes(pc, x0, x1, stride, pad) == takes, 'the eligibility rule of the test and of the kernels disagree'
    want = emu_ops.conv2d(pc, x0, x1, stride=stride, pad=pad, relu_in=relu_in, residual=residual, act=act, amp=True)
    exact = emu_ops.conv2d(pc, x0, x1, stride=stride, pad=pad, relu_in=relu_in, residual=residual, act=act)
    got = ops.conv2d(to_dev(pc), _guarded(x0), _guarded(x1), stride=stride, pad=pad, relu_in=relu_in,
                     residual=to_dev(residual), act=act, amp=True)
    torch.cuda.synchronize()
    assert not torch.isnan(got).any(), f'{name}: guard-band values leaked into the result'
    err, scale = max_err(got, want), max(1.0, want.abs().max().item())
    print(f'{name}: max abs err vs the fp16-rounded reference {err:.3e}; the rounding itself moves the output by '
          f'{max_err(want, exact):.3e} (|ref|max {scale:.3e})')
    assert err <= 2e-5 * scale, (name, err)
    if takes:
        assert max_err(want, exact) > 0, 'the case does not exercise the rounding'
    else:
        assert torch.equal(want, exact)


split_takes = emu_ops.split_takes


def _conv64(pc, x0, x1, stride, pad, relu_in, residual, act):
    """the convolution in fp64 on the CPU: what both the fp32 kernels and the split kernels approximate"""
    import torch.nn.functional as F
    batch = max(x0.shape[0], 1 if x1 is None else x1.shape[0], 1 if residual is None else residual.shape[0])
    xs = [x0.expand(batch, -1, -1, -1)] + ([] if x1 is None else [x1.expand(batch, -1, -1, -1)])
    x = torch.cat(xs, 1).double()
    if relu_in:
        x = F.relu(x)
    y = F.conv2d(x, emu_ops._unpack(pc).double(), None if pc.bias is None else pc.bias.double(), stride=stride, padding=pad)
    if residual is not None:
        y = y + residual.double()
    return emu_ops._act(y, act)


# --f16_split: the hi/lo fp16 split (csrc/conv_f16.hip, PREC 2) is held to the SAME bound as the fp32 kernels against the
# SAME reference (fp32 F.conv2d on the CPU), over the same cases; the eligible ones among them run on the f16 pipes
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_split_matches_cpu(case):
    name, c0, c1, cout, k, stride, pad, batch, H, W, bcast0, relu_in, res, act, bias, bn = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
    cin = c0 + c1
    w = rand(g, cout, cin, k, k, scale=(2.0 / (cin * k * k))**0.5)
    b = rand(g, cout, scale=0.1) if bias else None
    bn_p = None
    if bn:
        bn_p = (torch.rand(cout, generator=g) + 0.5, rand(g, cout, scale=0.1), rand(g, cout, scale=0.1),
                torch.rand(cout, generator=g) + 0.5, 1e-5)
    pc = ops.pack_conv(w, b, bn_p, split=True)
    x0 = rand(g, 1 if bcast0 else batch, c0, H, W)
    x1 = rand(g, batch, c1, H, W) if c1 else None
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    residual = None
    if res == 'full':
        residual = rand(g, batch, cout, oh, ow)
    elif res == 'bcast':
        residual = rand(g, 1, cout, oh, ow)
        x0 = rand(g, batch, c0, H, W)
    want = emu_ops.conv2d(pc, x0, x1, stride=stride, pad=pad, relu_in=relu_in, residual=residual, act=act)
    before = ops.split_fallbacks(dev())
    got = ops.conv2d(to_dev(pc), _guarded(x0), _guarded(x1), stride=stride, pad=pad, relu_in=relu_in,
                     residual=to_dev(residual), act=act, split=True)
    torch.cuda.synchronize()
    assert not torch.isnan(got).any(), f'{name}: guard-band values leaked into the result'
    err = max_err(got, want)
    print(f'{name}: split={split_takes(pc, x0, x1, stride, pad)} max abs err {err:.3e} (|ref|max {want.abs().max().item():.3e})')
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), (name, err)
    assert ops.split_fallbacks(dev()) == before, 'inputs inside the fp16 range must not fall back'


# shapes the split kernels take, one per tile / kind / K-loop form; against the fp64 convolution the split result must be
# as close as the fp32 kernels' own (both are fp32-round-off class: the bound is 2x the fp32 kernels' error + 1e-6)
# (name, c0, c1, cout, k, batch, H, W, bcast0, relu_in, residual, act, in_scale)
SPLIT_CASES = [
    ('split_1x1_64tile', 64, 0, 64, 1, 1, 6, 8, False, False, 'none', ops.ACT_NONE, 1.0),
    ('split_1x1_32ch_cat', 32, 96, 72, 1, 3, 6, 10, True, False, 'full', ops.ACT_RELU, 1.0),
    ('split_1x1_res_relu', 256, 0, 1024, 1, 1, 30, 54, False, False, 'full', ops.ACT_RELU, 1.0),
    ('split_3x3_rowwrap', 64, 0, 64, 3, 2, 6, 10, False, True, 'full', ops.ACT_NONE, 1.0),
    ('split_3x3_cat_relu_in', 512, 256 + 32, 512, 3, 2, 6, 8, True, True, 'none', ops.ACT_NONE, 1.0),
    ('split_3x3_gru', 512, 512, 1536, 3, 2, 6, 8, False, False, 'none', ops.ACT_NONE, 1.0),
    ('split_3x3_tile128', 256, 0, 128, 3, 4, 60, 108, False, True, 'full', ops.ACT_NONE, 1.0),
    ('split_3x3_tile256', 256, 0, 256, 3, 5, 120, 216, False, True, 'full', ops.ACT_NONE, 1.0),
    ('split_1x1_tile256', 512, 0, 256, 1, 5, 120, 216, False, False, 'none', ops.ACT_SIGMOID, 1.0),
    ('split_3x3_splitk', 256, 0, 256, 3, 1, 30, 54, False, False, 'none', ops.ACT_RELU, 1.0),
    ('split_1x1_splitk', 1024, 0, 256, 1, 1, 30, 54, False, False, 'none', ops.ACT_RELU, 1.0),
    # activations far below 1 WITHOUT a bias (a bias of 0.1 hid the conv term): below |x| ~ 2^-3 the lo plane of
    # an activation enters the fp16 subnormals and the scheme's error stops shrinking with x -- the absolute floor of
    # 2^-25 |w| per product that include/deva_hip.h states; these cases are held to the fp32 class PLUS that floor
    ('split_small_inputs', 256, 0, 256, 3, 2, 12, 16, False, False, 'none', ops.ACT_NONE, 1e-4),
    ('split_midrange_inputs', 256, 0, 256, 3, 2, 12, 16, False, False, 'none', ops.ACT_NONE, 1e-2),
    ('split_small_inputs_1x1', 512, 0, 128, 1, 2, 12, 16, False, True, 'none', ops.ACT_NONE, 1e-3),
    ('split_large_inputs', 256, 0, 256, 3, 2, 12, 16, False, False, 'none', ops.ACT_NONE, 3e3),
    ('split_ragged_cout', 96, 32, 200, 3, 2, 20, 36, True, True, 'full', ops.ACT_RELU, 1.0),
    # 1x1 with a partial last K step: the one-channel tails of sensory_compress / g4_conv, a single source with a tail
    ('split_1x1_tail513', 512, 1, 512, 1, 2, 6, 8, False, False, 'full', ops.ACT_NONE, 1.0),
    ('split_1x1_tail257_bcast', 256, 1, 512, 1, 3, 30, 54, True, False, 'full', ops.ACT_RELU, 1.0),
    ('split_1x1_single_tail', 72, 0, 136, 1, 2, 10, 12, False, True, 'none', ops.ACT_NONE, 1.0),
    ('split_1x1_tail_31_of_32', 64, 31, 64, 1, 1, 6, 8, False, False, 'none', ops.ACT_SIGMOID, 1.0),
]


@pytest.mark.parametrize('case', SPLIT_CASES, ids=[c[0] for c in SPLIT_CASES])
def test_conv_split_is_fp32_accurate(case):
    name, c0, c1, cout, k, batch, H, W, bcast0, relu_in, res, act, in_scale = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
    cin, pad = c0 + c1, k // 2
    w = rand(g, cout, cin, k, k, scale=(2.0 / (cin * k * k))**0.5)
    w.view(-1)[::5] *= 1e-3  # weights far below the layer's largest: their lo planes sit in the fp16 subnormals
    # cases with scaled-down activations carry no bias (and no residual): the error is measured against the convolution
    # term itself, not against an O(0.1) bias that would hide it
    b = rand(g, cout, scale=0.1) if in_scale >= 1.0 else None
    pc = ops.pack_conv(w, b, None, split=True)
    x0 = rand(g, 1 if bcast0 else batch, c0, H, W, scale=in_scale)
    x1 = rand(g, batch, c1, H, W, scale=in_scale) if c1 else None
    residual = rand(g, batch, cout, H, W) if res == 'full' else None
    assert in_scale >= 1.0 or residual is None
    assert split_takes(pc, x0, x1, 1, pad), 'the case must run on the split kernels'
    want = _conv64(pc, x0, x1, 1, pad, relu_in, residual, act)
    before = ops.split_fallbacks(dev())
    got = ops.conv2d(to_dev(pc), _guarded(x0), _guarded(x1), pad=pad, relu_in=relu_in, residual=to_dev(residual), act=act,
                     split=True)
    f32 = ops.conv2d(to_dev(pc), _guarded(x0), _guarded(x1), pad=pad, relu_in=relu_in, residual=to_dev(residual), act=act)
    torch.cuda.synchronize()
    assert not torch.isnan(got).any(), f'{name}: guard-band values leaked into the result'
    scale = max(1e-30, want.abs().max().item())
    e_split = (got.double().cpu() - want).abs().max().item() / scale
    e_f32 = (f32.double().cpu() - want).abs().max().item() / scale
    # the absolute floor of the activation split: |x - hi - lo| <= 2^-25 once lo is an fp16 subnormal, i.e. at most
    # 2^-25 sum_k |w_mk| per output (worst case over signs; tests/test_split_arithmetic_cpu.py pins the per-element bound).
    # For activations of ordinary magnitude the term is below the fp32 kernels' own round-off; for |x| << 2^-3 it is what
    # is left, and the scheme is then 11..22-bit accurate in x rather than 22-bit
    floor = 2.0**-25 * float(w.abs().sum((1, 2, 3)).max()) / scale
    print(f'{name}: max err / |ref|max against fp64: split {e_split:.3e}, fp32 kernels {e_f32:.3e}; '
          f'absolute floor of the lo plane / |ref|max {floor:.3e}')
    if in_scale >= 1.0:
        assert e_split <= 2.0 * e_f32 + 1e-6, (name, e_split, e_f32)
    else:
        assert e_split <= 2.0 * e_f32 + floor, (name, e_split, e_f32, floor)
    assert ops.split_fallbacks(dev()) == before


@pytest.mark.parametrize('poison', [7.0e4, -1.0e6, float('inf'), float('nan')], ids=['70000', '-1e6', 'inf', 'nan'])
@pytest.mark.parametrize('k', [1, 3])
def test_conv_split_falls_back_beyond_the_fp16_range(poison, k):
    """one input element beyond +-65504 (or non-finite): the split kernel raises its flag and the fp32 kernel launched
    behind it, gated on the flag, produces the output: the fp32 kernels' arithmetic in their plain K order (a persistent
    64x64-tile launch without K slices -- csrc/conv_mfma.hip PERSIST --, so bit-identical to the plain fp32 call wherever
    the tile policy of that call does not slice K, and within the fp32 kernels' own bound of it otherwise)"""
    g = torch.Generator().manual_seed(11 + k)
    w = rand(g, 128, 128, k, k, scale=0.05)
    pc = to_dev(ops.pack_conv(w, rand(g, 128, scale=0.1), None, split=True))
    x = rand(g, 2, 128, 24, 32)
    x[1, 77, 13, 5] = poison
    xd = _guarded(x)
    before = ops.split_fallbacks(dev())
    got = ops.conv2d(pc, xd, pad=k // 2, act=ops.ACT_RELU, split=True)
    f32 = ops.conv2d(pc, xd, pad=k // 2, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    assert ops.split_fallbacks(dev()) == before + 1
    assert torch.equal(got.isnan(), f32.isnan()) and torch.equal(got.isinf(), f32.isinf())
    a, b = got.nan_to_num(0.0, 0.0, 0.0), f32.nan_to_num(0.0, 0.0, 0.0)
    if k == 1:  # 4 K steps on 48 tiles: the plain call runs the same unsliced K loop
        assert torch.equal(a, b)
    assert max_err(a, b) <= 2e-5 * max(1.0, b.abs().max().item())
    # the next call with clean inputs gets a fresh flag
    x[1, 77, 13, 5] = 0.5
    got = ops.conv2d(pc, _guarded(x), pad=k // 2, act=ops.ACT_RELU, split=True)
    torch.cuda.synchronize()
    assert ops.split_fallbacks(dev()) == before + 1 and bool(torch.isfinite(got).all())


@pytest.mark.parametrize('k,cin,cout,relu_in', [(3, 256, 256, True), (3, 128, 192, False), (1, 512, 256, False)])
def test_conv_split_fallback_walks_every_tile(k, cin, cout, relu_in):
    """the gated re-run is a PERSISTENT launch (at most 1 024 workgroups walk over the 64x64 tiles): a layer with several
    thousand tiles, one poisoned input -- every output must come from the fp32 arithmetic.  On these shapes the plain fp32
    call runs 128x128 tiles without K slices, i.e. the same K order: bit-identical, residual and bias included"""
    g = torch.Generator().manual_seed(31 + k + cin)
    pc = to_dev(ops.pack_conv(rand(g, cout, cin, k, k, scale=(2.0 / (cin * k * k))**0.5), rand(g, cout, scale=0.1), None,
                              split=True))
    x = rand(g, 2, cin, 120, 216)
    x[1, 17, 100, 200] = 3.0e5
    xd = _guarded(x)
    res = to_dev(rand(g, 2, cout, 120, 216))
    before = ops.split_fallbacks(dev())
    got = ops.conv2d(pc, xd, pad=k // 2, relu_in=relu_in, residual=res, act=ops.ACT_RELU, split=True)
    f32 = ops.conv2d(pc, xd, pad=k // 2, relu_in=relu_in, residual=res, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    assert ops.split_fallbacks(dev()) == before + 1
    assert torch.equal(got, f32)


# fp32 Winograd F(2x2, 3x3) (csrc/conv_wino.hip): the SAME 2e-5 bound as the direct kernels against the same CPU convolution, and
# against fp64 an error of the fp32 class (measured ~2x the direct kernels').  The library takes the path only for layers with
# >= 192 workgroups of 64 channels x 64 tiles, so the shapes are sized for that.
# (name, c0, c1, cout, batch, H, W, bcast0, relu_in, residual, act, bias)
WINO_CASES = [
    ('wino_plain', 64, 0, 128, 2, 96, 128, False, False, 'none', ops.ACT_NONE, True),
    ('wino_relu_res', 256, 0, 256, 2, 60, 108, False, True, 'full', ops.ACT_NONE, True),
    ('wino_cat_bcast', 32, 40, 192, 3, 64, 96, True, False, 'none', ops.ACT_RELU, True),
    ('wino_ragged_cout_bcast_res', 24, 0, 200, 4, 50, 76, False, True, 'bcast', ops.ACT_SIGMOID, False),
    ('wino_gru_shape', 512, 512, 1536, 2, 30, 54, False, False, 'none', ops.ACT_NONE, True),
    ('wino_narrow', 16, 0, 128, 200, 34, 4, False, False, 'full', ops.ACT_SQUARE_PLUS_ONE, True),
    ('wino_ragged_tiles', 64, 0, 192, 7, 46, 62, False, True, 'none', ops.ACT_RELU, True),
    # odd heights (4K: 2160 / 16 = 135 rows): the last tile row holds one output row
    ('wino_odd_height_res', 32, 0, 256, 4, 45, 64, False, False, 'full', ops.ACT_RELU, True),
    ('wino_odd_height_relu', 32, 32, 200, 5, 27, 72, False, True, 'none', ops.ACT_NONE, False),
]


@pytest.mark.parametrize('case', WINO_CASES, ids=[c[0] for c in WINO_CASES])
def test_conv_wino_matches_cpu(case):
    name, c0, c1, cout, batch, H, W, bcast0, relu_in, res, act, bias = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
    cin = c0 + c1
    w = rand(g, cout, cin, 3, 3, scale=(2.0 / (cin * 9))**0.5)
    b = rand(g, cout, scale=0.1) if bias else None
    pc = ops.pack_conv(w, b, None, wino=True)
    pc_direct = ops.pack_conv(w, b, None)
    assert pc.weight_wino is not None
    x0 = rand(g, 1 if bcast0 else batch, c0, H, W)
    x1 = rand(g, batch, c1, H, W) if c1 else None
    residual = rand(g, batch if res == 'full' else 1, cout, H, W) if res != 'none' else None
    want = emu_ops.conv2d(pc_direct, x0, x1, pad=1, relu_in=relu_in, residual=residual, act=act)
    want64 = _conv64(pc_direct, x0, x1, 1, 1, relu_in, residual, act)
    got = ops.conv2d(to_dev(pc), _guarded(x0), _guarded(x1), pad=1, relu_in=relu_in, residual=to_dev(residual), act=act)
    direct = ops.conv2d(to_dev(pc_direct), _guarded(x0), _guarded(x1), pad=1, relu_in=relu_in, residual=to_dev(residual), act=act)
    torch.cuda.synchronize()
    assert not torch.isnan(got).any(), f'{name}: guard-band values leaked into the result'
    if os.environ.get('DEVA_TEST_DRYRUN') != '1':  # (the emulated ops have one convolution)
        assert not torch.equal(got, direct), f'{name}: the Winograd kernel did not run (bit-identical to the direct kernels)'
    scale = max(1.0, want.abs().max().item())
    err = max_err(got, want)
    e_w = (got.double().cpu() - want64).abs().max().item() / max(1e-30, want64.abs().max().item())
    e_d = (direct.double().cpu() - want64).abs().max().item() / max(1e-30, want64.abs().max().item())
    print(f'{name}: max abs err {err:.3e} (|ref|max {scale:.3e}); against fp64 / |ref|max: Winograd {e_w:.3e}, direct kernels {e_d:.3e}')
    assert err <= 2e-5 * scale, (name, err)
    assert e_w <= 8.0 * e_d + 1e-6, (name, e_w, e_d)


def test_conv_wino_small_layers_stay_on_the_direct_kernels():
    """fewer than 160 workgroups of 64 channels x 64 tiles (most batch-1 layers of a 480p frame): the library ignores the
    transformed weights -- bit-identical to the call without them; so do odd widths and maps of 4 k + 2 pixels"""
    g = torch.Generator().manual_seed(8)
    w = rand(g, 64, 64, 3, 3, scale=0.05)
    pc, pcd = to_dev(ops.pack_conv(w, None, None, wino=True)), to_dev(ops.pack_conv(w, None, None))
    for shape in ((1, 64, 30, 54), (4, 64, 31, 54), (64, 64, 30, 53)):
        x = _guarded(rand(g, *shape))
        assert torch.equal(ops.conv2d(pc, x, pad=1), ops.conv2d(pcd, x, pad=1)), shape


# the 7x7 stride-2 stems on the f16 matrix pipes (csrc/conv_stem.hip): against fp64 like the split convolutions above, and
# against the fp32 kernels' own result for the same layer (image part + mask part, the form the fp32 graph runs)
@pytest.mark.parametrize('cin,batch,H,W,relu,in_scale', [
    (3, 1, 96, 128, True, 1.0), (4, 3, 80, 112, False, 1.0), (4, 1, 16, 16, False, 1.0), (3, 1, 480, 864, True, 2.5),
    (4, 5, 480, 864, False, 1.0), (4, 2, 1088, 1920, False, 1.0), (3, 1, 272, 4 * 130, True, 0.05)])
def test_stem7x7_is_fp32_accurate(cin, batch, H, W, relu, in_scale):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(cin * 1000 + H + batch)
    w = rand(g, 64, cin, 7, 7, scale=(2.0 / (cin * 49))**0.5)
    w.view(-1)[::5] *= 1e-3
    bn = (torch.rand(64, generator=g) + 0.5, rand(g, 64, scale=0.1), rand(g, 64, scale=0.1), torch.rand(64, generator=g) + 0.5, 1e-5)
    ps = ops.pack_stem(w, None, bn, dev())
    pc = ops.pack_conv(w, None, bn)
    image = rand(g, 1, 3, H, W, scale=in_scale)
    masks = torch.rand(batch, 1, H, W, generator=g) if cin == 4 else None
    x = image.expand(batch, -1, -1, -1) if masks is None else torch.cat([image.expand(batch, -1, -1, -1), masks], 1)
    want = F.conv2d(x.double(), emu_ops._unpack(pc).double(), pc.bias.double(), stride=2, padding=3)
    want = F.relu(want) if relu else want
    before = ops.split_fallbacks(dev())
    got = ops.stem7x7(ps, to_dev(image), to_dev(masks), relu=relu)
    f32 = ops.conv2d(to_dev(pc), to_dev(x.contiguous()), stride=2, pad=3, act=ops.ACT_RELU if relu else ops.ACT_NONE)
    torch.cuda.synchronize()
    assert tuple(got.shape) == (batch, 64, H // 2, W // 2) and not torch.isnan(got).any()
    scale = max(1e-30, want.abs().max().item())
    e_stem = (got.double().cpu() - want).abs().max().item() / scale
    e_f32 = (f32.double().cpu() - want).abs().max().item() / scale
    print(f'stem {cin}ch x{batch} {H}x{W}: max err / |ref|max against fp64: stem kernel {e_stem:.3e}, fp32 kernels {e_f32:.3e}')
    assert e_stem <= 2.0 * e_f32 + 1e-6, (e_stem, e_f32)
    assert ops.split_fallbacks(dev()) == before


@pytest.mark.parametrize('poison', [7.0e4, float('inf')], ids=['70000', 'inf'])
def test_stem7x7_recomputes_a_tile_beyond_the_fp16_range(poison):
    """an input beyond +-65504: the workgroup that meets it recomputes ITS tile with fp32 FMAs inside the kernel (no second
    launch) and the call is counted; everywhere the result is the fp32 convolution's"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    w = rand(g, 64, 4, 7, 7, scale=0.1)
    ps, pc = ops.pack_stem(w, rand(g, 64, scale=0.1), None, dev()), ops.pack_conv(w, None, None)
    image, masks = rand(g, 1, 3, 64, 160), torch.rand(2, 1, 64, 160, generator=g)
    masks[1, 0, 40, 100] = poison
    x = torch.cat([image.expand(2, -1, -1, -1), masks], 1)
    want = F.conv2d(x, emu_ops._unpack(pc), ps.bias.cpu(), stride=2, padding=3)
    before = ops.split_fallbacks(dev())
    got = ops.stem7x7(ps, to_dev(image), to_dev(masks)).cpu()
    torch.cuda.synchronize()
    assert ops.split_fallbacks(dev()) == before + 1
    assert torch.equal(got.isfinite(), want.isfinite())
    fin = want.isfinite()
    assert (got[fin] - want[fin]).abs().max().item() <= 2e-5 * max(1.0, want[fin].abs().max().item())


def test_conv_split_in_place_residual_takes_the_fp32_kernels():
    """out aliasing the residual (an in-place residual add) with split=True: the fp32 re-run behind a split launch would
    read the residual after `out` was written, so such a call must run the fp32 kernels alone -- also when an input is
    beyond the fp16 range (the case in which the re-run does its work); bit-identical to the out-of-place fp32 call"""
    g = torch.Generator().manual_seed(21)
    pc = to_dev(ops.pack_conv(rand(g, 128, 128, 3, 3, scale=0.03), rand(g, 128, scale=0.1), None, split=True))
    for poison in (None, 1.0e5):
        x = rand(g, 2, 128, 24, 32)
        if poison is not None:
            x[0, 5, 3, 3] = poison
        xd = _guarded(x)
        res = to_dev(rand(g, 2, 128, 24, 32))
        want = ops.conv2d(pc, xd, pad=1, residual=res.clone(), act=ops.ACT_RELU)
        before = ops.split_fallbacks(dev())
        got = ops.conv2d(pc, xd, pad=1, residual=res, act=ops.ACT_RELU, out=res, split=True)
        torch.cuda.synchronize()
        assert got.data_ptr() == res.data_ptr() and torch.equal(got, want), poison
        assert ops.split_fallbacks(dev()) == before, 'the split kernels must not have run'


def test_conv_split_flag_ring_wraps():
    """more split convolutions than the ring has slots: the raised flags survive in the running total"""
    g = torch.Generator().manual_seed(3)
    pc = to_dev(ops.pack_conv(rand(g, 64, 64, 1, 1, scale=0.1), None, None, split=True))
    x = rand(g, 1, 64, 8, 8)
    x[0, 3, 2, 1] = 1.0e5
    bad, good = _guarded(x), _guarded(rand(g, 1, 64, 8, 8))
    before = ops.split_fallbacks(dev())
    for i in range(ops._SPLIT_RING + 50):
        ops.conv2d(pc, bad if i % 1000 == 0 else good, split=True)
    torch.cuda.synchronize()
    assert ops.split_fallbacks(dev()) == before + (ops._SPLIT_RING + 50 + 999) // 1000


def test_conv_rejects_cpu_tensors():
    pc = ops.pack_conv(torch.zeros(32, 16, 1, 1))
    with pytest.raises(Exception):
        ops.conv2d(pc, torch.zeros(1, 16, 4, 4))


@pytest.mark.parametrize('cin,cout,k,h,w', [(1024, 256, 1, 30, 54), (256, 256, 3, 30, 54), (512, 64, 3, 30, 54),
                                             (256, 1024, 1, 30, 54)])
def test_split_k_is_deterministic(cin, cout, k, h, w):
    """Layers with few output tiles accumulate K ranges in parallel workgroups and a second kernel adds
    the partial sums in split order: every repetition must reproduce the first result bit for bit, with
    other split-K layers using the same workspace in between."""
    g = torch.Generator().manual_seed(cin + cout + k)
    wgt = rand(g, cout, cin, k, k, scale=(2.0 / (cin * k * k))**0.5)
    b = rand(g, cout, scale=0.1)
    pc = to_dev(ops.pack_conv(wgt, b))
    x = ops._alloc((1, cin, h, w), dev())
    x.copy_(rand(g, 1, cin, h, w))
    other_pc = to_dev(ops.pack_conv(rand(g, 128, 512, 1, 1, scale=0.05)))
    other_x = to_dev(rand(g, 1, 512, 16, 16))
    want = emu_ops.conv2d(ops.pack_conv(wgt, b), x.cpu(), None, stride=1, pad=k // 2, act=ops.ACT_RELU)
    first = ops.conv2d(pc, x, pad=k // 2, act=ops.ACT_RELU).clone()
    assert max_err(first, want) <= 2e-5 * max(1.0, want.abs().max().item())
    outs = []
    for it in range(300):
        outs.append(ops.conv2d(pc, x, pad=k // 2, act=ops.ACT_RELU))
        if it % 3 == 0:
            ops.conv2d(other_pc, other_x)
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(o, first)) for o in outs)
    assert bad == 0, f'{bad} of 300 repetitions differ from the first result'
