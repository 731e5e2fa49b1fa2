"""pooling / resampling / CBAM / GRU / aggregate kernels against their PyTorch definitions."""
import pytest
import torch

import emu_ops
from deva.hip import ops
from gpu_util import dev, max_err, rand, to_dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _check(name, got, want, tol=1e-5):
    torch.cuda.synchronize()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = max_err(got, want)
    print(f'{name}: max abs err {err:.3e}')
    assert err <= tol * max(1.0, want.abs().max().item()), (name, err)


# (widths that are multiples of 8 take the four-outputs-per-thread kernel, the others the scalar one)
@pytest.mark.parametrize('shape', [(1, 64, 48, 64), (2, 8, 45, 65), (1, 3, 1, 1), (3, 5, 2, 7), (2, 3, 5, 8), (1, 2, 9, 16), (1, 4, 1, 24)])
@pytest.mark.parametrize('relu', [False, True])
def test_maxpool(shape, relu):
    x = rand(torch.Generator().manual_seed(1), *shape)
    _check('maxpool', ops.maxpool3x3s2(to_dev(x), relu), emu_ops.maxpool3x3s2(x, relu), 0)


@pytest.mark.parametrize('shape', [(2, 16, 6, 8), (1, 3, 1, 1), (3, 5, 7, 9), (2, 4, 5, 4), (1, 2, 3, 6), (3, 8, 68, 120), (1, 2, 1, 4)])
@pytest.mark.parametrize('with_skip', [False, True])
def test_upsample2x_add(shape, with_skip):
    g = torch.Generator().manual_seed(2)
    x = rand(g, *shape)
    skip = rand(g, 1, shape[1], 2 * shape[2], 2 * shape[3]) if with_skip else None
    _check('up2x', ops.upsample2x_add(to_dev(x), to_dev(skip)), emu_ops.upsample2x_add(x, skip))


@pytest.mark.parametrize('shape', [(2, 16, 60, 108), (11, 8, 136, 240), (1, 3, 6, 10), (2, 2, 4, 6)])
def test_upsample2x_add_ds2_is_both_kernels_in_one_pass(shape):
    """deva_upsample2x_add_ds2: the x2 up-sampling + skip AND area_downsample(x, 2) of the input from the same loads --
    bit-identical to the two separate launches"""
    g = torch.Generator().manual_seed(sum(shape))
    x = to_dev(rand(g, *shape))
    skip = to_dev(rand(g, 1, shape[1], 2 * shape[2], 2 * shape[3]))
    up, ds = ops.upsample2x_add_ds2(x, skip)
    torch.cuda.synchronize()
    assert torch.equal(up, ops.upsample2x_add(x, skip)) and torch.equal(ds, ops.area_downsample(x, 2))


@pytest.mark.parametrize('shape,f', [((2, 5, 32, 48), 16), ((3, 7, 8, 12), 2), ((2, 1, 8, 12), 4), ((1, 2, 3, 3), 1)])
def test_area_downsample(shape, f):
    x = rand(torch.Generator().manual_seed(3), *shape)
    _check('area', ops.area_downsample(to_dev(x), f), emu_ops.area_downsample(x, f))


@pytest.mark.parametrize('shape,f', [((3, 16, 272, 480), 4), ((2, 8, 120, 216), 4), ((2, 8, 136, 240), 2), ((3, 4, 60, 108), 2),
                                     ((1, 3, 12, 20), 4), ((2, 2, 8, 12), 2), ((5, 480, 864), 16), ((2, 3, 32, 48), 16)])
def test_area_downsample_vector_path_is_bit_identical_to_the_scalar_kernel(shape, f):
    """factors 2 / 4 on aligned rows run 16-byte loads (2 or 4 outputs per thread); same sums in the same order as the
    one-output-per-thread kernel, which an unaligned view of the same data still takes"""
    g = torch.Generator().manual_seed(sum(shape) + f)
    x = rand(g, *shape)
    n = x.numel()
    buf = torch.zeros(n + 8, device=dev())
    buf[4:4 + n] = x.reshape(-1).to(dev())   # 16-byte aligned view -> vector path
    buf2 = torch.zeros(n + 8, device=dev())
    buf2[1:1 + n] = x.reshape(-1).to(dev())  # 4-byte aligned view -> scalar path
    a = ops.area_downsample(buf[4:4 + n].view(*shape), f)
    b = ops.area_downsample(buf2[1:1 + n].view(*shape), f)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    _check('area', a, emu_ops.area_downsample(x, f))


@pytest.mark.parametrize('k', [1, 3])
@pytest.mark.parametrize('shape', [(1, 256, 120, 216), (3, 64, 60, 108), (2, 5, 14, 22), (1, 3, 6, 10)])
def test_gather_s2_is_the_stride2_convolutions_im2col(shape, k):
    """deva_gather_s2: the taps of a k x k stride-2 convolution as channels (tap-major), exact copies; a 1x1 convolution with
    the weights in [cout][t*C + c] order over it is the stride-2 convolution (checked on the CPU in fp64)"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(k + sum(shape))
    x = rand(g, *shape)
    got = ops.gather_s2(to_dev(x), k)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), emu_ops.gather_s2(x, k))
    w = rand(g, 8, shape[1], k, k)
    ref = F.conv2d(x.double(), w.double(), stride=2, padding=k // 2)
    via = F.conv2d(got.cpu().double(), w.permute(0, 2, 3, 1).reshape(8, -1, 1, 1).double())
    assert (via - ref).abs().max().item() <= 1e-9 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('no', [1, 2, 5])
def test_aggregate_and_softmax(no):
    g = torch.Generator().manual_seed(4)
    logits = rand(g, no, 24, 33, scale=4.0)
    _check('aggregate_sigmoid', ops.aggregate(to_dev(logits), True), emu_ops.aggregate(logits, True), 2e-6)
    prob = torch.rand(no, 24, 33, generator=g)
    prob[0, :3] = 0.0
    prob[0, 3:6] = 1.0  # exercise the clamp
    agg = emu_ops.aggregate(prob)
    _check('aggregate_prob', ops.aggregate(to_dev(prob)), agg, 2e-6)
    onehot = (torch.rand(24, 33, generator=g) * (no + 1)).long()
    onehot = torch.stack([onehot == i + 1 for i in range(no)], 0)
    _check('aggregate_bool', ops.aggregate(to_dev(onehot)), emu_ops.aggregate(onehot), 2e-6)
    _check('softmax_channels', ops.softmax_channels(to_dev(agg)), emu_ops.softmax_channels(agg), 2e-6)


def test_aggregate_without_objects():
    """nothing tracked yet (empty detection round, inference_core.py:196): background-only logits"""
    empty = torch.zeros(0, 24, 33)
    _check('aggregate_empty', ops.aggregate(to_dev(empty)), emu_ops.aggregate(empty), 2e-6)


@pytest.mark.parametrize('c,h,w', [(3, 6, 8), (6, 24, 32), (1, 1, 1), (2, 5, 3)])
def test_upsample4x_softmax(c, h, w):
    x = rand(torch.Generator().manual_seed(5), c, h, w, scale=5.0)
    up, prob = ops.upsample4x_softmax(to_dev(x))
    wup, wprob = emu_ops.upsample4x_softmax(x)
    _check('up4x_logits', up, wup, 2e-6)
    _check('up4x_prob', prob, wprob, 2e-6)
    _, prob2 = ops.upsample4x_softmax(to_dev(x), need_logits=False)
    _check('up4x_prob_only', prob2, wprob, 2e-6)


@pytest.mark.parametrize('b,c,h,w', [(2, 512, 6, 8), (1, 512, 30, 54), (3, 64, 5, 7), (5, 512, 30, 54), (2, 40, 9, 12)])
def test_cbam(b, c, h, w):
    g = torch.Generator().manual_seed(6)
    x = rand(g, b, c, h, w)
    hid = c // 16
    w1, b1 = rand(g, hid, c, scale=c**-0.5), rand(g, hid, scale=0.1)
    w2, b2 = rand(g, c, hid, scale=hid**-0.5), rand(g, c, scale=0.1)
    sp = ops.pack_conv(rand(g, 1, 2, 7, 7, scale=0.2), rand(g, 1, scale=0.1))
    want = emu_ops.cbam(x, w1, b1, w2, b2, sp)
    got = ops.cbam(to_dev(x), to_dev(w1), to_dev(b1), to_dev(w2), to_dev(b2), to_dev(sp))
    _check('cbam', got, want, 1e-5)


@pytest.mark.parametrize('b,c,h,w', [(2, 512, 6, 8), (1, 16, 3, 5)])
def test_gru(b, c, h, w):
    g = torch.Generator().manual_seed(7)
    v, hh = rand(g, b, 3 * c, h, w, scale=2.0), rand(g, b, c, h, w)
    _check('gru', ops.gru_update(to_dev(v), to_dev(hh)), emu_ops.gru_update(v, hh), 2e-6)


@pytest.mark.parametrize('h,w,size,aa', [(720, 1280, (480, 853), True), (1080, 1920, (480, 853), True),
                                         (2160, 3840, (480, 853), True), (240, 320, (480, 640), True),
                                         (720, 1280, (480, 853), False), (97, 131, (50, 203), True),
                                         (480, 854, None, True)])
def test_input_head(h, w, size, aa):
    """uint8 HWC -> normalised, (antialias-)resized CHW vs torchvision's tensor pipeline restated with
    torch (ToTensor + Normalize + F.interpolate(bilinear, antialias))"""
    g = torch.Generator().manual_seed(h + w)
    img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
    # smooth it a little so that neighbouring pixels correlate like in a photograph
    img = ((img.float() + img.float().roll(1, 0) + img.float().roll(1, 1)) / 3).round().to(torch.uint8)
    want = emu_ops.input_head(img, size, antialias=aa)
    got = ops.input_head(to_dev(img), size, antialias=aa)
    assert tuple(got.shape) == tuple(want.shape)
    _check(f'input_head {h}x{w}->{size} aa={aa}', got, want, 3e-6)
    # fused pad_divide_by: same pixels inside an exactly zero border
    pad = (3, 4, 1, 2)
    padded = ops.input_head(to_dev(img), size, antialias=aa, pad=pad)
    want_p = emu_ops.input_head(img, size, antialias=aa, pad=pad)
    assert tuple(padded.shape) == tuple(want_p.shape)
    assert torch.equal(padded[:, 1:padded.shape[1] - 2, 3:padded.shape[2] - 4], got)
    border = padded.clone()
    border[:, 1:padded.shape[1] - 2, 3:padded.shape[2] - 4] = 0
    assert (border == 0).all()


@pytest.mark.parametrize('dtype', [torch.float32, torch.int64, torch.uint8])
@pytest.mark.parametrize('shape,pad', [((3, 480, 854), (5, 5, 0, 0)), ((1080, 1920), (0, 0, 4, 4)), ((2, 3, 37, 51), (6, 7, 5, 6))])
def test_pad2d_matches_f_pad(dtype, shape, pad):
    """deva_pad2d (one launch) against F.pad with zeros: bit-identical; the path of tensor_utils.pad_divide_by"""
    import torch.nn.functional as F
    from deva.utils.tensor_utils import pad_divide_by
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 50).to(dtype)
    got = ops.pad2d(to_dev(x), pad)
    torch.cuda.synchronize()
    assert got.dtype == dtype and torch.equal(got.cpu(), F.pad(x, pad))
    via, p = pad_divide_by(to_dev(x), 16)
    ref, p_ref = pad_divide_by(x, 16)
    assert p == p_ref and torch.equal(via.cpu(), ref)

