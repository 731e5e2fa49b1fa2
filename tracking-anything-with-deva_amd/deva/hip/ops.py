"""Tensor-level wrappers over the C ABI of libdeva_hip.so.

PyTorch is used for device memory (torch.empty on the caching allocator), views and the current
HIP stream only; every arithmetic operation below runs in a hand-written gfx950 kernel.  All
wrappers raise `DevaHipError` on non-HIP / non-contiguous / wrong-dtype tensors: there is no
fallback path.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SQUARE_PLUS_ONE, KLAYOUT_CHUNK32, KLAYOUT_Q4, KLAYOUT_TAP_MAJOR,
               ConvDesc, DevaHipError, check, lib)

__all__ = ['PackedConv', 'pack_conv', 'conv2d', 'split_fallbacks', 'PackedStem', 'pack_stem', 'stem7x7', 'pad2d', 'usage_init', 'gather_s2', 'maxpool3x3s2', 'upsample2x_add', 'upsample2x_add_ds2', 'area_downsample',
           'aggregate', 'softmax_channels', 'upsample4x_softmax', 'cbam', 'gru_update',
           'affinity_topk', 'BankPrep', 'affinity_dense', 'affinity_candidates', 'affinity_merge', 'usage_update', 'readout_sparse', 'bank_append', 'bank_gather_rows',
           'bank_export', 'rank', 'rank_select', 'evict_select', 'similarity_dense', 'softmax_columns',
           'label_histogram', 'merge_paint', 'lut_remap', 'index_mask', 'input_head',
           'ACT_NONE', 'ACT_RELU', 'ACT_SIGMOID', 'ACT_SQUARE_PLUS_ONE']


def require_hip(device, what: str) -> None:
    if torch.device(device).type != 'cuda':
        raise DevaHipError(f'{what} must be on the HIP device to run (is on {device}); '
                           'there is no CPU execution path')


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_RAW_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream(device=None) -> int:
    """raw handle of torch's current stream on `device` (default: the current device).  torch.cuda.current_stream()
    builds a Stream object through four Python layers (6 us, twice per kernel launch = 1 ms of a 3.5-ms frame at
    480p / 1 object); the C entry points behind it return the same handle in 0.3 us."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        if device is not None and not isinstance(device, torch.device):
            device = torch.device(device)  # 'cuda:1', 1
        index = None if device is None else device.index
        return _RAW_STREAM(_RAW_DEVICE() if index is None else index)
    return torch.cuda.current_stream(device).cuda_stream


def _p(t: Optional[torch.Tensor], dtype=torch.float32, name: str = 'tensor') -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise DevaHipError(f'{name} must live on the HIP device (got {t.device}); there is no CPU path')
    if t.dtype != dtype:
        raise DevaHipError(f'{name} must be {dtype} (got {t.dtype})')
    if not t.is_contiguous():
        raise DevaHipError(f'{name} must be contiguous')
    return t.data_ptr()


def _batched(t: torch.Tensor, name: str) -> Tuple[int, int]:
    """pointer and batch stride of an NCHW tensor whose per-item [C,H,W] block is contiguous"""
    if not t.is_cuda or t.dtype != torch.float32:
        raise DevaHipError(f'{name} must be an fp32 HIP tensor')
    if t.dim() != 4 or not (t.is_contiguous() or t[0].is_contiguous()):  # (the first test avoids building a view)
        raise DevaHipError(f'{name} must be [B,C,H,W] with contiguous items')
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else 0)


# ------------------------------------------------------------------------------------------ conv
@dataclass
class PackedConv:
    """weights of one convolution in the kernel's layout; K ordered tap-major (k = tap*Cin + c) or, for
    Cin % 32 == 0 and kernels larger than 1x1, in 32-channel slabs (k = ((c/32)*KH*KW + tap)*32 + c%32);
    stored [K][cout_pad], or k-quad interleaved [ceil(K/4)][cout_pad][4] when k_layout carries KLAYOUT_Q4
    (every convolution with more than one output channel) -- see include/deva_hip.h"""
    weight: torch.Tensor
    bias: Optional[torch.Tensor]
    cin: int
    cout: int
    cout_pad: int
    kh: int
    kw: int
    k_layout: int = KLAYOUT_TAP_MAJOR
    # opt-in amp path: the same weights rounded to fp16 in the layout of csrc/conv_f16.hip (None: the layer stays fp32)
    weight_f16: Optional[torch.Tensor] = None
    # opt-in split path (fp32-accurate on the f16 matrix pipes): hi / lo fp16 planes of weight * 2^split_scale_log2
    # (None: the layer stays on the fp32 kernels)
    weight_split: Optional[torch.Tensor] = None
    split_scale_log2: int = 0
    # fp32 Winograd F(2x2, 3x3) (csrc/conv_wino.hip): the weights transformed by deva_conv_pack_wino (None: the direct kernels)
    weight_wino: Optional[torch.Tensor] = None


def pack_f16(w: torch.Tensor) -> Optional[torch.Tensor]:
    """[cout][cin][kh][kw] fp32 (BatchNorm already folded) -> fp16 weights of the amp kernels, element (k, m) at
    ((k/8)*cout_pad + m)*8 + k%8; K tap-major for 1x1, 64-channel slabs otherwise (k = ((c/64)*taps + tap)*64 + c%64).
    None when the kernels cannot take the layer (cin % 64 != 0, a single output channel).  Same arithmetic as
    deva_conv_pack_f16 (round to nearest even)."""
    cout, cin, kh, kw = w.shape
    taps = kh * kw
    if cin % 64 or cout < 2:
        return None
    cout_pad = (cout + 31) // 32 * 32
    wk = torch.zeros(taps * cin, cout_pad, dtype=torch.float32, device=w.device)
    if taps > 1:
        wk[:, :cout] = w.reshape(cout, cin // 64, 64, taps).permute(1, 3, 2, 0).reshape(-1, cout)
    else:
        wk[:, :cout] = w.reshape(cout, cin).t()
    return wk.view(-1, 8, cout_pad).permute(0, 2, 1).contiguous().to(torch.float16).reshape(-1)


def pack_split(w: torch.Tensor) -> Tuple[Optional[torch.Tensor], int]:
    """[cout][cin][kh][kw] fp32 (BatchNorm already folded) -> (hi / lo fp16 planes of the split kernels, e): with
    s = 2^e such that max|w| * s lies in [2^13, 2^14), hi = fp16(w s), lo = fp16(w s - hi); element (k, plane, m) at
    (((k/8)*2 + plane)*cout_pad + m)*8 + k%8; K tap-major for 1x1, 32-channel slabs otherwise
    (k = ((c/32)*taps + tap)*32 + c%32); a 1x1 layer may have any cin (513, 257: K is padded with zero rows to a multiple
    of 32).  (None, 0) when the kernels cannot take the layer (3x3 with cin % 32 != 0, a single output channel, non-finite
    weights).  Same arithmetic as deva_conv_pack_split."""
    import math
    cout, cin, kh, kw = w.shape
    taps = kh * kw
    if (cin % 32 and taps > 1) or cout < 2 or not bool(torch.isfinite(w).all()):
        return None, 0
    wmax = float(w.abs().max())
    e = 0
    if wmax > 0.0:
        e = max(-120, min(120, 14 - math.frexp(wmax)[1]))
    cout_pad = (cout + 31) // 32 * 32
    wk = torch.zeros((taps * cin + 31) // 32 * 32, cout_pad, dtype=torch.float32, device=w.device)  # (1x1: K padded to 32)
    if taps > 1:
        wk[:, :cout] = w.reshape(cout, cin // 32, 32, taps).permute(1, 3, 2, 0).reshape(-1, cout)
    else:
        wk[:cin, :cout] = w.reshape(cout, cin).t()
    ws = torch.ldexp(wk, torch.tensor(e, dtype=torch.int32, device=w.device))  # exact: a power of two
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    planes = torch.stack([hi.view(-1, 8, cout_pad).permute(0, 2, 1), lo.view(-1, 8, cout_pad).permute(0, 2, 1)], 1)
    return planes.contiguous().reshape(-1), e


def pack_wino(w: torch.Tensor) -> Optional[torch.Tensor]:
    """[cout][cin][3][3] fp32 (BatchNorm folded) -> the Winograd-transformed weights of deva_conv_pack_wino (None when the
    layer is not eligible: another kernel size, cin % 8 != 0)"""
    cout, cin, kh, kw = w.shape
    if (kh, kw) != (3, 3) or cin % 8:
        return None
    w = w.detach().to(torch.float32).cpu().contiguous()
    L = lib()
    n = L.deva_conv_pack_wino(w.data_ptr(), None, cout, cin)
    if n <= 0:
        return None
    out = torch.empty(n, dtype=torch.float32)
    if L.deva_conv_pack_wino(w.data_ptr(), out.data_ptr(), cout, cin) != n:
        raise DevaHipError(f'deva_conv_pack_wino failed: {L.deva_hip_last_error().decode()}')
    return out


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn=None,
              device=None, amp: bool = False, split: bool = False, wino: bool = False) -> PackedConv:
    """One-time weight preparation (model load, not the frame path): fold an eval-mode BatchNorm
    `bn = (gamma, beta, running_mean, running_var, eps)` into the convolution and repack
    [cout][cin][kh][kw] -> [kh*kw*cin][cout_pad]."""
    w = weight.detach().to(torch.float32)
    cout, cin, kh, kw = w.shape
    b = None if bias is None else bias.detach().to(torch.float32)
    if bn is not None:
        gamma, beta, mean, var, eps = bn
        scale = gamma.detach().float() / torch.sqrt(var.detach().float() + eps)
        w = w * scale.view(-1, 1, 1, 1)
        shift = beta.detach().float() - mean.detach().float() * scale
        b = shift if b is None else b * scale + shift
    cout_pad = (cout + 31) // 32 * 32
    packed = torch.zeros(kh * kw * cin, cout_pad, dtype=torch.float32, device=w.device)
    if kh * kw > 1 and cin % 32 == 0:
        layout = KLAYOUT_CHUNK32  # [cin/32][tap][32][cout]
        packed[:, :cout] = w.reshape(cout, cin // 32, 32, kh * kw).permute(1, 3, 2, 0).reshape(-1, cout)
    else:
        layout = KLAYOUT_TAP_MAJOR  # [tap][cin][cout]
        packed[:, :cout] = w.permute(2, 3, 1, 0).reshape(kh * kw * cin, cout)
    if cout > 1:  # k-quad interleave for the lean-loop kernels (csrc/conv_mfma.hip); same arithmetic as deva_conv_pack
        k = packed.shape[0]
        kq = (k + 3) // 4
        if kq * 4 != k:
            packed = torch.cat([packed, torch.zeros(kq * 4 - k, cout_pad, dtype=packed.dtype, device=packed.device)], 0)
        packed = packed.view(kq, 4, cout_pad).permute(0, 2, 1).contiguous().view(kq * 4, cout_pad)
        layout |= KLAYOUT_Q4
    w16 = pack_f16(w) if amp else None
    wsp, e = pack_split(w) if split else (None, 0)
    wwi = pack_wino(w) if wino else None
    if device is not None:
        packed = packed.to(device)
        b = None if b is None else b.to(device)
        w16 = None if w16 is None else w16.to(device)
        wsp = None if wsp is None else wsp.to(device)
        wwi = None if wwi is None else wwi.to(device)
    return PackedConv(packed.contiguous(), None if b is None else b.contiguous(), cin, cout, cout_pad, kh, kw,
                      layout, w16, wsp, e, wwi)


GUARD = 8192  # floats of readable slack on both sides of every tensor this module allocates


def _alloc(shape, device) -> torch.Tensor:
    """fp32 output tensor with GUARD readable floats before and after it: the vector gathers of
    deva_conv2d may touch (and then mask) a few elements beyond the ends of their inputs"""
    n = 1
    for s_ in shape:
        n *= int(s_)
    flat = torch.empty(n + 2 * GUARD, dtype=torch.float32, device=device)
    out = flat[GUARD:GUARD + n].view(*shape)
    # what _guard_elems would compute for this very object (views and slices of it recompute), with the extent it holds
    # for: an in-place metadata operation on the object (as_strided_, set_, resize_) must not keep a stale promise
    out._deva_guard = (GUARD, out.data_ptr(), n)
    return out


def _guard_elems(t: Optional[torch.Tensor]) -> int:
    """readable floats before the first / after the last element of t inside its storage"""
    if t is None:
        return 1 << 30
    known = getattr(t, '_deva_guard', None)  # set by _alloc on the tensor object it returns: 3 us less per operand
    if known is not None and known[1] == t.data_ptr() and known[2] == t.numel() and t.is_contiguous():
        return known[0]
    first = t.storage_offset()
    last = first + sum((int(n) - 1) * int(st) for n, st in zip(t.shape, t.stride()))
    total = t.untyped_storage().nbytes() // 4
    return max(0, min(first, total - 1 - last))


_WORKSPACE = {}
_WORKSPACE_ELEMS = 16 * 1024 * 1024  # 64 MB of split-K scratch per device
_MAX_STREAM_CACHES = 8  # per-(device, stream) scratch caches are bounded: transient side streams must not pin HBM forever


def _make_room(cache: dict) -> None:
    """called before a NEW (device, stream) entry is inserted: drop the oldest entries beyond the bound (the caching
    allocator keeps a dropped buffer alive until the kernels already queued on its stream have run)"""
    while len(cache) >= _MAX_STREAM_CACHES:
        cache.pop(next(iter(cache)))



def _workspace(device) -> torch.Tensor:
    """split-K scratch, one per (device, stream): launches on different streams never share partial sums"""
    key = (device, _stream(device))
    ws = _WORKSPACE.get(key)
    if ws is None:
        _make_room(_WORKSPACE)
        ws = _WORKSPACE[key] = torch.empty(_WORKSPACE_ELEMS, dtype=torch.float32, device=device)
    return ws


_SPLIT_FLAGS = {}
_SPLIT_EVICTED = {}  # device index -> fall-backs counted by rings that _make_room has dropped
_SPLIT_RING = 4096  # flag slots per (device, stream); re-zeroed (one fill launch) every _SPLIT_RING split convolutions


def _dev_index(device) -> int:
    """torch.device('cuda') and torch.device('cuda:0') name the same device when 0 is current"""
    d = torch.device(device)
    if d.type != 'cuda':
        return -1
    return torch.cuda.current_device() if d.index is None else d.index


def _evict_split_rings() -> None:
    """before a NEW ring is inserted: the rings beyond the bound are dropped, their totals folded into _SPLIT_EVICTED
    (after a device synchronise: the ring may belong to another stream with launches in flight)"""
    while len(_SPLIT_FLAGS) >= _MAX_STREAM_CACHES:
        (dev, _), (ring, _) = next(iter(_SPLIT_FLAGS.items()))
        if ring.is_cuda:
            torch.cuda.synchronize(ring.device)
        _SPLIT_EVICTED[_dev_index(dev)] = _SPLIT_EVICTED.get(_dev_index(dev), 0) + int(ring.sum().item())
        _SPLIT_FLAGS.pop(next(iter(_SPLIT_FLAGS)))


def _split_flag(device) -> int:
    """device address of a zeroed int the next split convolution may raise (deva_conv_desc.split_flag): slots of a
    per-(device, stream) ring; when the ring wraps, the raised slots are added to the ring's running total (slot
    _SPLIT_RING, read by `split_fallbacks`) and the ring is cleared -- two launches per _SPLIT_RING convolutions"""
    key = (device, _stream(device))
    ent = _SPLIT_FLAGS.get(key)
    if ent is None:
        _evict_split_rings()
        ent = _SPLIT_FLAGS[key] = [torch.zeros(_SPLIT_RING + 1, dtype=torch.int32, device=device), 0]
    ring, nxt = ent
    if nxt == _SPLIT_RING:
        ring[_SPLIT_RING:].add_(ring[:_SPLIT_RING].sum(dtype=torch.int32))
        ring[:_SPLIT_RING].zero_()
        nxt = 0
    ent[1] = nxt + 1
    return ring.data_ptr() + 4 * nxt


def split_fallbacks(device) -> int:
    """number of split convolutions on `device` (all streams of this process) whose inputs left the fp16 range, so
    that the fp32 kernels behind them produced the output (synchronises: a statistic for tests and the bench)"""
    want = _dev_index(device)
    if torch.device(device).type == 'cuda':
        torch.cuda.synchronize(torch.device('cuda', want))  # rings of OTHER streams (key-encoder prefetch) are read below
    total = _SPLIT_EVICTED.get(want, 0)
    for (dev, _), (ring, _) in _SPLIT_FLAGS.items():
        if _dev_index(dev) == want:
            total += int(ring.sum().item())
    return total


def conv2d(pc: PackedConv, x0: torch.Tensor, x1: Optional[torch.Tensor] = None, *, stride: int = 1,
           pad: int = 0, relu_in: bool = False, residual: Optional[torch.Tensor] = None,
           act: int = ACT_NONE, out: Optional[torch.Tensor] = None, amp: bool = False,
           split: bool = False) -> torch.Tensor:
    """out = act(conv(cat([x0, x1], 1)) + bias + residual); batch-1 operands broadcast.
    amp: fp16 operands (inputs rounded while they are staged, `pc.weight_f16`) with fp32 accumulation where the fp16
    kernels take the shape, exact fp32 otherwise (include/deva_hip.h: deva_conv_desc.amp).
    split: fp32-accurate arithmetic on the f16 matrix pipes (hi/lo fp16 split of both operands, `pc.weight_split`) where
    the split kernels take the shape, the fp32 kernels otherwise and for inputs beyond the fp16 range (amp == 2)."""
    c0 = x0.shape[1]
    c1 = 0 if x1 is None else x1.shape[1]
    if c0 + c1 != pc.cin:
        raise DevaHipError(f'conv2d: {c0}+{c1} input channels, weights expect {pc.cin}')
    if (pc.k_layout & 0xf) == KLAYOUT_CHUNK32 and (c0 % 32 or c1 % 32):
        raise DevaHipError('conv2d: 32-channel-slab weights need both concatenated inputs to be multiples of 32 channels')
    batch = max(x0.shape[0], 1 if x1 is None else x1.shape[0], 1 if residual is None else residual.shape[0])
    h, w = x0.shape[-2:]
    if x1 is not None and tuple(x1.shape[-2:]) != (h, w):
        raise DevaHipError('conv2d: x0/x1 spatial size mismatch')
    for t in (x0, x1, residual):
        if t is not None and t.shape[0] not in (1, batch):
            raise DevaHipError('conv2d: batch sizes must be 1 or equal')
    oh = (h + 2 * pad - pc.kh) // stride + 1
    ow = (w + 2 * pad - pc.kw) // stride + 1
    if out is None:
        out = _alloc((batch, pc.cout, oh, ow), x0.device)
    elif tuple(out.shape) != (batch, pc.cout, oh, ow):
        raise DevaHipError('conv2d: bad output shape')
    d = ConvDesc()
    d.in0, d.in0_batch_stride = _batched(x0, 'x0')
    if x1 is not None:
        d.in1, d.in1_batch_stride = _batched(x1, 'x1')
    else:
        d.in1, d.in1_batch_stride = None, 0
    d.c0, d.c1 = c0, c1
    d.batch, d.height, d.width = batch, h, w
    d.weight = _p(pc.weight, name='packed weight')
    d.bias = _p(pc.bias, name='bias')
    d.cout, d.cout_pad = pc.cout, pc.cout_pad
    d.k_layout = pc.k_layout
    d.kh, d.kw, d.stride, d.pad = pc.kh, pc.kw, stride, pad
    d.relu_in = 1 if relu_in else 0
    if residual is not None:
        if tuple(residual.shape[1:]) != (pc.cout, oh, ow):
            raise DevaHipError('conv2d: residual shape mismatch')
        d.residual, d.residual_batch_stride = _batched(residual, 'residual')
    else:
        d.residual, d.residual_batch_stride = None, 0
    d.act = act
    d.out = _p(out, name='out')
    d.in_guard_elems = min(_guard_elems(x0), _guard_elems(x1), (1 << 31) - 1)
    ws = _workspace(out.device)
    d.workspace, d.workspace_elems = ws.data_ptr(), ws.numel()
    d.split_scale_log2, d.split_flag = 0, None
    if amp and pc.weight_f16 is not None:
        d.weight_f16, d.amp = _p(pc.weight_f16, torch.float16, 'fp16 weight'), 1
    elif split and pc.weight_split is not None:
        d.weight_f16, d.amp = _p(pc.weight_split, torch.float16, 'fp16 hi/lo weight'), 2
        d.split_scale_log2, d.split_flag = pc.split_scale_log2, _split_flag(out.device)
    else:
        d.weight_f16, d.amp = None, 0
    # fp32 Winograd for the layers packed with it (taken by the library only where it pays: big 3x3 stride-1 layers)
    d.weight_wino = _p(pc.weight_wino, name='Winograd weight') if (pc.weight_wino is not None and d.amp == 0) else None
    check(lib().deva_conv2d(d, _stream()), 'deva_conv2d')
    return out


# ------------------------------------------------------------------------------------------ pointwise
@dataclass
class PackedStem:
    """weights of a 7x7 stride-2 stem for deva_stem7x7 (csrc/conv_stem.hip): hi / lo fp16 planes of w * 2^scale_log2 in the
    kernel's K order, the fp32 weights of its in-kernel fall-back, the folded bias"""
    planes: torch.Tensor   # int16 view of the uint16 planes, [cin*8*2*64*8]
    w32: torch.Tensor      # [cin*49, 64]
    bias: Optional[torch.Tensor]
    cin: int
    scale_log2: int


def pack_stem(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn=None, device=None) -> PackedStem:
    """[64][cin = 3 | 4][7][7] (+ eval-mode BatchNorm, folded like pack_conv) -> PackedStem through deva_stem_pack"""
    import ctypes
    w = weight.detach().to(torch.float32).cpu()
    cout, cin, kh, kw = w.shape
    if (cout, kh, kw) != (64, 7, 7) or cin not in (3, 4):
        raise DevaHipError(f'pack_stem: 64 x (3|4) x 7 x 7 weights expected, got {tuple(w.shape)}')
    b = None if bias is None else bias.detach().to(torch.float32).cpu()
    if bn is not None:
        gamma, beta, mean, var, eps = bn
        scale = gamma.detach().float().cpu() / torch.sqrt(var.detach().float().cpu() + eps)
        w = w * scale.view(-1, 1, 1, 1)
        shift = beta.detach().float().cpu() - mean.detach().float().cpu() * scale
        b = shift if b is None else b * scale + shift
    w = w.contiguous()
    L = lib()
    e = ctypes.c_int(0)
    n = L.deva_stem_pack(w.data_ptr(), cin, None, None, ctypes.byref(e))
    if n < 0:
        raise DevaHipError(f'deva_stem_pack failed: {L.deva_hip_last_error().decode()}')
    planes = torch.zeros(n, dtype=torch.int16)
    w32 = torch.zeros(cin * 49, 64, dtype=torch.float32)
    if L.deva_stem_pack(w.data_ptr(), cin, planes.data_ptr(), w32.data_ptr(), ctypes.byref(e)) != n:
        raise DevaHipError(f'deva_stem_pack failed: {L.deva_hip_last_error().decode()}')
    if device is not None:
        planes, w32, b = planes.to(device), w32.to(device), None if b is None else b.to(device)
    return PackedStem(planes, w32, None if b is None else b.contiguous(), cin, int(e.value))


def stem7x7(ps: PackedStem, image: torch.Tensor, masks: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """act(conv7x7 stride 2 pad 3 over cat(image broadcast, masks) + bias): image [1 or B,3,H,W], masks [B,1,H,W] | None
    -> [B,64,H/2,W/2]; fp32-accurate on the f16 matrix pipes (deva_stem7x7), inputs beyond the fp16 range recomputed in fp32
    inside the kernel (counted by `split_fallbacks`)"""
    c1 = 0 if masks is None else masks.shape[1]
    if image.shape[1] + c1 != ps.cin:
        raise DevaHipError(f'stem7x7: {image.shape[1]}+{c1} input channels, weights expect {ps.cin}')
    batch = image.shape[0] if masks is None else masks.shape[0]
    if image.shape[0] not in (1, batch):
        raise DevaHipError('stem7x7: the image batch must be 1 or equal to the masks\'')
    h, w = image.shape[-2:]
    if masks is not None and tuple(masks.shape[-2:]) != (h, w):
        raise DevaHipError('stem7x7: image / mask size mismatch')
    in0, bs0 = _batched(image, 'image')
    in1, bs1 = (None, 0) if masks is None else _batched(masks, 'masks')
    out = _alloc((batch, 64, h // 2, w // 2), image.device)
    check(lib().deva_stem7x7(in0, bs0, image.shape[1], in1, bs1, c1, batch, h, w, _p(ps.planes, torch.int16), _p(ps.w32),
                             ps.scale_log2, _p(ps.bias), int(relu), _p(out), _split_flag(image.device), _stream()),
          'deva_stem7x7')
    return out


def pad2d(x: torch.Tensor, pad: Tuple[int, int, int, int]) -> torch.Tensor:
    """F.pad(x, (left, right, top, bottom)) with zeros on the last two dimensions, one launch (deva_pad2d); fp32 results
    carry the guard bands of `_alloc`"""
    left, right, top, bottom = pad
    h, w = x.shape[-2:]
    oh, ow = h + top + bottom, w + left + right
    if not x.is_cuda or not x.is_contiguous() or x.element_size() not in (1, 4, 8):
        raise DevaHipError('pad2d: a contiguous HIP tensor with 1-, 4- or 8-byte elements expected')
    shape = (*x.shape[:-2], oh, ow)
    out = _alloc(shape, x.device) if x.dtype == torch.float32 else torch.empty(shape, dtype=x.dtype, device=x.device)
    planes = x.numel() // (h * w)
    check(lib().deva_pad2d(x.data_ptr(), out.data_ptr(), x.element_size(), planes, h, w, top, left, oh, ow, _stream()),
          'deva_pad2d')
    return out


def usage_init(use: torch.Tensor, life: torch.Tensor) -> None:
    """use[:] = 0, life[:] = 1e-7 (the counters of freshly appended tokens), one launch"""
    check(lib().deva_usage_init(_p(use), _p(life), use.numel(), _stream()), 'deva_usage_init')


def gather_s2(x: torch.Tensor, kernel: int) -> torch.Tensor:
    """[B,C,H,W] -> [B,k*k*C,OH,OW]: the taps of a k x k stride-2 convolution (k = 1 pad 0, k = 3 pad 1) as channels,
    tap-major (deva_gather_s2); a 1x1 convolution with weights [cout][t*C + c] over it is that convolution"""
    b, c, h, w = x.shape
    pad = kernel // 2
    oh, ow = (h + 2 * pad - kernel) // 2 + 1, (w + 2 * pad - kernel) // 2 + 1
    out = _alloc((b, kernel * kernel * c, oh, ow), x.device)
    check(lib().deva_gather_s2(_p(x), _p(out), b, c, h, w, kernel, _stream()), 'deva_gather_s2')
    return out


def maxpool3x3s2(x: torch.Tensor, relu_after: bool = False) -> torch.Tensor:
    b, c, h, w = x.shape
    out = _alloc((b, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), x.device)
    check(lib().deva_maxpool3x3s2(_p(x), _p(out), b * c, h, w, int(relu_after), _stream()), 'deva_maxpool3x3s2')
    return out


def upsample2x_add(x: torch.Tensor, skip: Optional[torch.Tensor]) -> torch.Tensor:
    """x [B,C,h,w] -> [B,C,2h,2w] bilinear (+ skip [1,C,2h,2w] broadcast over B)"""
    b, c, h, w = x.shape
    if skip is not None and tuple(skip.shape[-3:]) != (c, 2 * h, 2 * w):
        raise DevaHipError('upsample2x_add: skip shape mismatch')
    out = _alloc((b, c, 2 * h, 2 * w), x.device)
    check(lib().deva_upsample2x_add(_p(x), _p(skip), _p(out), b, c, h, w, _stream()), 'deva_upsample2x_add')
    return out


def upsample2x_add_ds2(x: torch.Tensor, skip: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """(upsample2x_add(x, skip), area_downsample(x, 2)) from one pass over x (deva_upsample2x_add_ds2); x [B,C,h,w], h, w even"""
    b, c, h, w = x.shape
    if skip is not None and tuple(skip.shape[-3:]) != (c, 2 * h, 2 * w):
        raise DevaHipError('upsample2x_add_ds2: skip shape mismatch')
    if h % 2 or w % 2:
        raise DevaHipError('upsample2x_add_ds2: even input size expected')
    out = _alloc((b, c, 2 * h, 2 * w), x.device)
    ds2 = _alloc((b, c, h // 2, w // 2), x.device)
    check(lib().deva_upsample2x_add_ds2(_p(x), _p(skip), _p(out), _p(ds2), b, c, h, w, _stream()), 'deva_upsample2x_add_ds2')
    return out, ds2


def area_downsample(x: torch.Tensor, factor: int) -> torch.Tensor:
    """[..., H, W] -> [..., H/factor, W/factor] box mean"""
    h, w = x.shape[-2:]
    planes = x.numel() // (h * w)
    out = _alloc((*x.shape[:-2], h // factor, w // factor), x.device)
    check(lib().deva_area_downsample(_p(x), _p(out), planes, h, w, factor, _stream()), 'deva_area_downsample')
    return out


def aggregate(prob: torch.Tensor, apply_sigmoid: bool = False) -> torch.Tensor:
    """[no, ...] object probabilities (fp32, or uint8/bool one-hot) -> [no+1, ...] logits"""
    no = prob.shape[0]
    pixels = 1
    for d in prob.shape[1:]:
        pixels *= int(d)
    is_u8 = prob.dtype in (torch.uint8, torch.bool)
    if no == 0:  # nothing tracked yet: background-only logits
        require_hip(prob.device, 'aggregate')
        src = None
    elif is_u8:
        src = _p(prob.view(torch.uint8) if prob.dtype == torch.bool else prob, torch.uint8, 'prob')
    else:
        src = _p(prob, name='prob')
    out = torch.empty((no + 1, *prob.shape[1:]), dtype=torch.float32, device=prob.device)
    check(lib().deva_aggregate(src, int(is_u8), int(apply_sigmoid), _p(out), no, pixels, _stream()),
          'deva_aggregate')
    return out


def softmax_channels(x: torch.Tensor) -> torch.Tensor:
    c = x.shape[0]
    out = torch.empty_like(x)
    check(lib().deva_softmax_channels(_p(x), _p(out), c, x.numel() // c, _stream()), 'deva_softmax_channels')
    return out


def upsample4x_softmax(logits: torch.Tensor, need_logits: bool = True):
    """[C,h,w] -> (logits_up [C,4h,4w] or None, prob [C,4h,4w])"""
    c, h, w = logits.shape
    prob = torch.empty((c, 4 * h, 4 * w), dtype=torch.float32, device=logits.device)
    up = torch.empty_like(prob) if need_logits else None
    check(lib().deva_upsample4x_softmax(_p(logits), _p(up), _p(prob), c, h, w, _stream()),
          'deva_upsample4x_softmax')
    return up, prob


def cbam(x: torch.Tensor, w1, b1, w2, b2, spatial: PackedConv) -> torch.Tensor:
    """returns x + CBAM(x) for x [B,C,h,w]  (cbam.py:21-76 + the residual add of
    group_modules.py:149)"""
    b, c, h, w = x.shape
    hw = h * w
    dev = x.device
    avg = torch.empty((b, c), dtype=torch.float32, device=dev)
    mx = torch.empty_like(avg)
    check(lib().deva_global_avgmax(_p(x), _p(avg), _p(mx), b * c, hw, _stream()), 'deva_global_avgmax')
    scale = torch.empty_like(avg)
    check(lib().deva_cbam_mlp(_p(avg), _p(mx), _p(w1), _p(b1), _p(w2), _p(b2), _p(scale), b, c, w1.shape[0],
                              _stream()), 'deva_cbam_mlp')
    pooled = _alloc((b, 2, h, w), dev)
    check(lib().deva_cbam_channel_pool(_p(x), _p(scale), _p(pooled), b, c, hw, _stream()),
          'deva_cbam_channel_pool')
    gate = conv2d(spatial, pooled, pad=spatial.kh // 2)
    out = _alloc(x.shape, dev)
    check(lib().deva_cbam_apply(_p(x), _p(scale), _p(gate), _p(out), b, c, hw, _stream()), 'deva_cbam_apply')
    return out


def gru_update(values: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """values [B,3C,h,w], h [B,C,h,w] -> new h"""
    b, c = h.shape[:2]
    hw = h.shape[-2] * h.shape[-1]
    if values.shape[1] != 3 * c:
        raise DevaHipError('gru_update: values must have 3x the channels of h')
    out = _alloc(h.shape, h.device)
    check(lib().deva_gru_update(_p(values), _p(h), _p(out), b, c, hw, _stream()), 'deva_gru_update')
    return out


# ------------------------------------------------------------------------------------------ memory read
_AFF_WS = {}


def _affinity_workspace(elems: int, device) -> torch.Tensor:
    """hand-over buffer between the two affinity kernels, kept alive (grow-only) per device and stream:
    deva_affinity_topk rewrites every list length and the live part of every list before
    deva_affinity_finalize reads them"""
    key = (device, _stream(device))
    ws = _AFF_WS.get(key)
    if ws is None:
        _make_room(_AFF_WS)
    if ws is None or ws.numel() < elems:
        # grow geometrically: the scratch of deva_affinity_read scales with the bank, which grows every memory frame --
        # an exact-size buffer would be re-allocated (a fresh hipMalloc, the old block parked in the cache) each time
        # (+50 % for small buffers, +12.5 % once the buffer is beyond 256 MiB: the 4K read's scratch is ~0.7 GiB)
        slack = elems // 2 if elems < (32 << 20) else elems // 8
        ws = _AFF_WS[key] = torch.empty((max(elems + slack, 1 << 20),), dtype=torch.int64, device=device)
    return ws


class BankPrep:
    """The bank side of the pre-filtered memory read (mean key, operand scales, fp16 MFMA fragments of every token), kept
    by the caller per bank between reads (include/deva_hip.h: deva_affinity_read_prepared).  `key` is whatever the owner
    uses to tell one state of the bank from another (MemoryManager: the stores' bucket versions and sizes); a read
    whose key equals the key of the read that filled the buffer skips the bank kernels."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.key = None
        self._pending = None

    def prepare(self, nbytes: int, device, key) -> Tuple[int, int]:
        """-> (device address of the buffer, 1 if it holds the operands of this very bank state)"""
        if self.buf is None or self.buf.numel() * 8 < nbytes or self.buf.device != torch.device(device):
            # (a bank grows by one frame of tokens per memory frame: grow with slack like the read scratch)
            self.buf = torch.empty(((nbytes + nbytes // 4) // 8 + 64,), dtype=torch.int64, device=device)
            self.key = None
        valid = int(key is not None and self.key == key)
        # the key is committed by `commit` AFTER the read that fills the buffer has been launched successfully: a call that
        # raises (bad arguments, launch error) must not leave a promise about a buffer that was never filled
        self._pending = key
        self.key = None
        return self.buf.data_ptr(), valid

    def commit(self) -> None:
        self.key = self._pending


def affinity_topk(key_long, shr_long, n_long: int, key_work, shr_work, n_work: int, qk: torch.Tensor,
                  qe: torch.Tensor, k: int, usage_fix: Optional[torch.Tensor] = None,
                  splits: Optional[int] = None, prep: Optional[BankPrep] = None, prep_key=None):
    """Fused similarity -> top-k -> softmax.  key_* token-major [>=n,64] arenas, shr_* [>=n];
    qk/qe [64,hw].  Returns idx int32 [hw,k] (long-then-work token index), weight fp32 [hw,k];
    adds weight*2^40 into usage_fix (uint64 viewed as int64, [>= n_long+n_work]) if given.
    prep / prep_key: the caller's `BankPrep` of this bank and the key of the bank's current state -- reads of an
    unchanged bank then re-use the prepared fp16 operands (bit-identical results)."""
    hw = qk.shape[1]
    if qk.shape[0] != 64 or tuple(qe.shape) != tuple(qk.shape):
        raise DevaHipError('affinity_topk: queries must be [64, hw]')
    L = lib()
    idx = torch.empty((hw, k), dtype=torch.int32, device=qk.device)
    weight = torch.empty((hw, k), dtype=torch.float32, device=qk.device)
    if splits is None:
        # the library picks the kernels: fp16 pre-filter + exact fp32 re-scoring on banks where it pays, else the fp32
        # kernels (also its device-side fall-back); bit-identical results either way
        scratch = _affinity_workspace(L.deva_affinity_read_scratch(n_long + n_work, hw, k), qk.device)
        bank_prep, valid = None, 0
        if prep is not None and L.deva_affinity_prefilter_enabled(n_long + n_work, hw, k):
            bank_prep, valid = prep.prepare(L.deva_affinity_bank_prep_bytes(n_long + n_work), qk.device,
                                            None if prep_key is None else (prep_key, n_long, n_work))
        check(L.deva_affinity_read_prepared(_p(key_long) if n_long else None, _p(shr_long) if n_long else None, n_long,
                                            _p(key_work) if n_work else None, _p(shr_work) if n_work else None, n_work,
                                            _p(qk), _p(qe), hw, k, _p(scratch, torch.int64), _p(idx, torch.int32), _p(weight),
                                            _p(usage_fix, torch.int64), None, None, 0, bank_prep, valid, _stream()),
              'deva_affinity_read_prepared')
        if bank_prep is not None:
            prep.commit()
        return idx, weight
    part = _affinity_workspace(L.deva_affinity_workspace(hw, k, splits), qk.device)
    check(L.deva_affinity_topk(_p(key_long) if n_long else None, _p(shr_long) if n_long else None, n_long,
                               _p(key_work) if n_work else None, _p(shr_work) if n_work else None, n_work,
                               _p(qk), _p(qe), hw, k, splits, _p(part, torch.int64), _stream()),
          'deva_affinity_topk')
    check(L.deva_affinity_finalize(_p(part, torch.int64), hw, k, splits, _p(idx, torch.int32), _p(weight),
                                   _p(usage_fix, torch.int64), _stream()), 'deva_affinity_finalize')
    return idx, weight


def affinity_dense(key_long, shr_long, n_long: int, key_work, shr_work, n_work: int, qk: torch.Tensor, qe: torch.Tensor,
                   k: int, usage_fix: Optional[torch.Tensor] = None):
    """The read on the dense kernel (deva_affinity_dense: 1 <= k <= 64; what `affinity_topk` runs for k > 32).  Same
    arguments and results as `affinity_topk`; for k <= 32 bit-identical to it (tests)."""
    hw = qk.shape[1]
    if qk.shape[0] != 64 or tuple(qe.shape) != tuple(qk.shape):
        raise DevaHipError('affinity_dense: queries must be [64, hw]')
    idx = torch.empty((hw, k), dtype=torch.int32, device=qk.device)
    weight = torch.empty((hw, k), dtype=torch.float32, device=qk.device)
    check(lib().deva_affinity_dense(_p(key_long) if n_long else None, _p(shr_long) if n_long else None, n_long,
                                    _p(key_work) if n_work else None, _p(shr_work) if n_work else None, n_work,
                                    _p(qk), _p(qe), hw, k, _p(idx, torch.int32), _p(weight), _p(usage_fix, torch.int64),
                                    _stream()), 'deva_affinity_dense')
    return idx, weight


def affinity_last_read_flag(device) -> int:
    """test hook: fall-back flag of the last pre-filtered read on the current stream of `device` (synchronises);
    0 = the fp16 pre-filter produced the result, otherwise the fp32 kernels took over (bit 0: non-finite bank,
    1: negative / non-finite query, 2: a candidate sub-list overflowed, 3: too many candidates to re-score)"""
    ws = _AFF_WS.get((device, _stream(device)))
    if ws is None:
        raise DevaHipError('affinity_last_read_flag: no read has run on this stream')
    return int(lib().deva_affinity_read_flag(_p(ws, torch.int64), _stream()))


def usage_update(usage_fix: torch.Tensor, offset: int, use: Optional[torch.Tensor], life: torch.Tensor,
                 n: int) -> None:
    check(lib().deva_usage_update(_p(usage_fix, torch.int64), offset, _p(use), _p(life), n, _stream()),
          'deva_usage_update')


def readout_sparse(idx: torch.Tensor, weight: torch.Tensor, val_long, n_long: int, val_work,
                   out: torch.Tensor, tok_range: Optional[Tuple[int, int]] = None,
                   row_map_long: Optional[torch.Tensor] = None, row_map_work: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out [cv, hw(...)] = sparse readout of token-major values ([>=n, cv] arenas); with tok_range =
    (lo, hi) only the tokens lo <= t < hi contribute (partial read-out of one bank shard); with row maps
    (int32 [>= n], value-sharded storage) token t of a segment is row map[t] of its arena, < 0 = not on this rank"""
    hw, k = idx.shape
    cv = out.shape[0]
    if out.numel() != cv * hw:
        raise DevaHipError('readout_sparse: bad output shape')
    lo, hi = (0, (1 << 31) - 1) if tok_range is None else (int(tok_range[0]), int(tok_range[1]))
    check(lib().deva_readout_sparse(_p(idx, torch.int32), _p(weight), hw, k, _p(val_long) if n_long else None,
                                    n_long, _p(val_work), cv, _p(out), lo, hi,
                                    _p(row_map_long, torch.int32) if n_long else None, _p(row_map_work, torch.int32),
                                    _stream()), 'deva_readout_sparse')
    return out


def affinity_candidates(key_long, shr_long, n_long: int, key_work, shr_work, n_work: int, qk: torch.Tensor,
                        qe: torch.Tensor, k: int, token_offset: int = 0, splits: Optional[int] = None):
    """One shard of a token-sharded bank: the shard's own sorted top-k per query in the hand-over format of
    the affinity kernels -- keys int64 [hw, 64] (order-preserving score bits << 32 | ~(token + offset); the
    first k entries of a list are live), counts int32 [hw]"""
    hw = qk.shape[1]
    if qk.shape[0] != 64 or tuple(qe.shape) != tuple(qk.shape):
        raise DevaHipError('affinity_candidates: queries must be [64, hw]')
    L = lib()
    keys = torch.zeros((hw, 64), dtype=torch.int64, device=qk.device)
    counts = torch.empty((hw,), dtype=torch.int32, device=qk.device)
    if splits is None:
        scratch = _affinity_workspace(L.deva_affinity_read_scratch(n_long + n_work, hw, k), qk.device)
        check(L.deva_affinity_read(_p(key_long) if n_long else None, _p(shr_long) if n_long else None, n_long,
                                   _p(key_work) if n_work else None, _p(shr_work) if n_work else None, n_work,
                                   _p(qk), _p(qe), hw, k, _p(scratch, torch.int64), None, None, None,
                                   _p(keys, torch.int64), _p(counts, torch.int32), int(token_offset), _stream()),
              'deva_affinity_read')
        return keys, counts
    part = _affinity_workspace(L.deva_affinity_workspace(hw, k, splits), qk.device)
    check(L.deva_affinity_topk(_p(key_long) if n_long else None, _p(shr_long) if n_long else None, n_long,
                               _p(key_work) if n_work else None, _p(shr_work) if n_work else None, n_work,
                               _p(qk), _p(qe), hw, k, splits, _p(part, torch.int64), _stream()),
          'deva_affinity_topk')
    check(L.deva_affinity_select(_p(part, torch.int64), hw, k, splits, int(token_offset), _p(keys, torch.int64),
                                 _p(counts, torch.int32), _stream()), 'deva_affinity_select')
    return keys, counts


def affinity_merge(keys: torch.Tensor, counts: torch.Tensor, k: int, usage_fix: Optional[torch.Tensor] = None):
    """keys int64 [lists, hw, 64], counts int32 [lists, hw] (the gathered `affinity_candidates` of every
    shard) -> idx int32 [hw, k], weight fp32 [hw, k] of the exact global top-k (+ usage like affinity_topk)"""
    lists, hw = counts.shape
    idx = torch.empty((hw, k), dtype=torch.int32, device=keys.device)
    weight = torch.empty((hw, k), dtype=torch.float32, device=keys.device)
    check(lib().deva_affinity_merge(_p(keys, torch.int64), _p(counts, torch.int32), hw, k, lists, _p(idx, torch.int32),
                                    _p(weight), _p(usage_fix, torch.int64), _stream()), 'deva_affinity_merge')
    return idx, weight


# ------------------------------------------------------------------------------------------ bank upkeep
def bank_append(src: torch.Tensor, arena: torch.Tensor, row0: int) -> None:
    """src [C, n] channel-major -> arena rows row0..row0+n (token-major [cap, C])"""
    c, n = src.shape
    if arena.shape[1] != c or row0 + n > arena.shape[0]:
        raise DevaHipError('bank_append: arena too small or channel mismatch')
    check(lib().deva_bank_append(_p(src), _p(arena), row0, c, n, _stream()), 'deva_bank_append')


def bank_gather_rows(src: torch.Tensor, rows: Optional[torch.Tensor], dst: torch.Tensor, count: int) -> None:
    c = src.shape[1] if src.dim() == 2 else 1
    check(lib().deva_bank_gather_rows(_p(src), _p(rows, torch.int32), _p(dst), count, c, _stream()),
          'deva_bank_gather_rows')


def bank_export(arena: torch.Tensor, n: int) -> torch.Tensor:
    """first n rows of a token-major arena -> channel-major [C, n] (reference layout)"""
    c = arena.shape[1]
    out = torch.empty((c, n), dtype=torch.float32, device=arena.device)
    if n > 0:
        check(lib().deva_bank_export(_p(arena), _p(out), c, n, _stream()), 'deva_bank_export')
    return out


def rank(x: torch.Tensor, n: int, descending: bool, life: Optional[torch.Tensor] = None):
    """permutation rank of x[:n] (ties by index); with `life`, ranks x/life and returns it too"""
    r = torch.empty((n,), dtype=torch.int32, device=x.device)
    xn = torch.empty((n,), dtype=torch.float32, device=x.device) if life is not None else None
    check(lib().deva_rank(_p(x), _p(life), _p(xn), n, int(descending), _p(r, torch.int32), _stream()),
          'deva_rank')
    return r, xn


def rank_select(rank_t: torch.Tensor, k: int) -> torch.Tensor:
    out = torch.empty((k,), dtype=torch.int32, device=rank_t.device)
    check(lib().deva_rank_select(_p(rank_t, torch.int32), rank_t.numel(), k, _p(out, torch.int32), _stream()),
          'deva_rank_select')
    return out


def evict_select(x: torch.Tensor, rank_asc: torch.Tensor, n_remove: int):
    """survivor rows (x > x at ascending rank n_remove-1), in order; returns (idx int32 [n], count tensor)"""
    n = rank_asc.numel()
    idx = torch.empty((n,), dtype=torch.int32, device=x.device)
    count = torch.zeros((1,), dtype=torch.int32, device=x.device)
    check(lib().deva_evict_select(_p(x), _p(rank_asc, torch.int32), n, n_remove, _p(idx, torch.int32),
                                  _p(count, torch.int32), _stream()), 'deva_evict_select')
    return idx, count


def similarity_dense(key: torch.Tensor, shr: torch.Tensor, sel: torch.Tensor, proto_idx: torch.Tensor,
                     n_cand: int) -> torch.Tensor:
    """dense [n_cand, ld] similarity of candidates vs prototypes, ld = P rounded up to 32 (zero pad)"""
    p = proto_idx.numel()
    ld = (p + 31) // 32 * 32
    sim = torch.zeros((n_cand, ld), dtype=torch.float32, device=key.device)
    check(lib().deva_similarity_dense(_p(key), _p(shr), _p(sel), _p(proto_idx, torch.int32), n_cand, p, ld,
                                      _p(sim), _stream()), 'deva_similarity_dense')
    return sim


def softmax_columns(x: torch.Tensor, p: int) -> torch.Tensor:
    n, ld = x.shape
    check(lib().deva_softmax_columns(_p(x), n, p, ld, _stream()), 'deva_softmax_columns')
    return x


# ------------------------------------------------------------------------------------------ detection merging
def label_histogram(ours: torch.Tensor, news: torch.Tensor, new_ids: torch.Tensor, n_our: int) -> torch.Tensor:
    """joint histogram [n_our+1, n_new+1] (int32) of two int64 index masks; column n_new = unlisted ids"""
    n_new = new_ids.numel()
    counts = torch.zeros((n_our + 1, n_new + 1), dtype=torch.int32, device=ours.device)
    check(lib().deva_label_histogram(_p(ours, torch.int64), _p(news, torch.int64),
                                     _p(new_ids, torch.int64) if n_new else None, n_our, n_new, ours.numel(),
                                     _p(counts, torch.int32), _stream()), 'deva_label_histogram')
    return counts


def merge_paint(ours: torch.Tensor, news: torch.Tensor, new_ids: torch.Tensor, our_order: torch.Tensor,
                our_label: torch.Tensor, new_order: torch.Tensor, new_label: torch.Tensor,
                out_ids: torch.Tensor) -> torch.Tensor:
    """one-hot planes [n_out, *mask.shape] (fp32) of the area-ordered repaint"""
    n_our, n_new, n_out = our_order.numel() - 1, new_ids.numel(), out_ids.numel()
    out = _alloc((n_out, *ours.shape), ours.device)
    if n_out:
        check(lib().deva_merge_paint(_p(ours, torch.int64), _p(news, torch.int64),
                                     _p(new_ids, torch.int64) if n_new else None, n_our, n_new,
                                     _p(our_order, torch.int32), _p(our_label, torch.int64),
                                     _p(new_order, torch.int32) if n_new else None,
                                     _p(new_label, torch.int64) if n_new else None, _p(out_ids, torch.int64), n_out,
                                     ours.numel(), _p(out), _stream()), 'deva_merge_paint')
    return out


def lut_remap(mask: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    """int64 index mask -> lut[mask] (0 for values outside the table)"""
    out = torch.empty_like(mask)
    check(lib().deva_lut_remap(_p(mask, torch.int64), _p(lut, torch.int64), lut.numel(), mask.numel(),
                               _p(out, torch.int64), _stream()), 'deva_lut_remap')
    return out


def index_mask(prob: torch.Tensor, size: Optional[Tuple[int, int]] = None,
               lut: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[C,H,W] probabilities -> int64 [OH,OW] labels: bilinear resize to `size` (align_corners=False)
    if it differs from (H,W), argmax over C, then lut[argmax] if a table is given"""
    c, h, w = prob.shape
    oh, ow = (h, w) if size is None else (int(size[0]), int(size[1]))
    out = torch.empty((oh, ow), dtype=torch.int64, device=prob.device)
    n = 0 if lut is None else lut.numel()
    check(lib().deva_index_mask(_p(prob, name='prob'), c, h, w, oh, ow, _p(lut, torch.int64), n, _p(out, torch.int64),
                                _stream()), 'deva_index_mask')
    return out


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def input_head(image_u8: torch.Tensor, size: Optional[Tuple[int, int]] = None, *, antialias: bool = True,
               mean=IMAGENET_MEAN, std=IMAGENET_STD, pad: Tuple[int, int, int, int] = (0, 0, 0, 0)) -> torch.Tensor:
    """uint8 [H,W,3] frame on the device -> normalised fp32 [3,OH,OW] (ToTensor + Normalize + Resize), placed
    inside a zero border `pad` = (left, right, top, bottom) (pad_divide_by fused)"""
    import ctypes
    if image_u8.dtype != torch.uint8 or image_u8.dim() != 3 or image_u8.shape[2] != 3:
        raise DevaHipError('input_head: expects a uint8 [H, W, 3] frame')
    h, w = image_u8.shape[:2]
    oh, ow = (h, w) if size is None else (int(size[0]), int(size[1]))
    left, right, top, bottom = (int(v) for v in pad)
    out = _alloc((3, oh + top + bottom, ow + left + right), image_u8.device)
    m = (ctypes.c_float * 3)(*mean)
    sd = (ctypes.c_float * 3)(*std)
    check(lib().deva_input_head(_p(image_u8, torch.uint8, 'image'), h, w, m, sd, int(antialias), _p(out), oh, ow,
                                left, right, top, bottom, _stream()), 'deva_input_head')
    return out
