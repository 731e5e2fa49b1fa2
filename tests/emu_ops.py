"""TEST-ONLY stand-ins for `deva.hip.ops` written with plain PyTorch on the CPU.

The build container has no GPU, so the host-side logic of the package (weight packing, BN folding,
graph wiring, arena bookkeeping, the frame state machine) is exercised on the CPU by
monkeypatching these functions over the ctypes wrappers.  They define, in executable form, the
contract each HIP kernel must meet; the `-m gpu` tests then hold the real kernels to the same
oracle.  Nothing here is reachable from the product package.
"""
import torch
import torch.nn.functional as F

from deva.hip import ops as real
from deva.hip.ops import PackedConv

TWO40 = float(2**40)
_real_pack_conv = real.pack_conv


def require_hip(device, what):  # the emulation runs on the CPU
    return None


def pack_conv(weight, bias=None, bn=None, device=None, amp=False, split=False, wino=False):
    return _real_pack_conv(weight, bias, bn, None, amp, split, wino)  # (the Winograd contract is the fp32 convolution itself)


def _unpack(pc: PackedConv):
    w = pc.weight
    if pc.k_layout & real.KLAYOUT_Q4:  # [K/4][cout_pad][4] -> [K][cout_pad]
        k = pc.kh * pc.kw * pc.cin
        w = w.view(-1, pc.cout_pad, 4).permute(0, 2, 1).reshape(-1, pc.cout_pad)[:k]
    w = w[:, :pc.cout]
    if (pc.k_layout & 0xf) == 1:  # [cin/32][tap][32][cout]
        w = w.reshape(pc.cin // 32, pc.kh * pc.kw, 32, pc.cout).permute(3, 0, 2, 1)
        return w.reshape(pc.cout, pc.cin, pc.kh, pc.kw).contiguous()
    return w.reshape(pc.kh, pc.kw, pc.cin, pc.cout).permute(3, 2, 0, 1).contiguous()


def _act(y, act):
    if act == real.ACT_RELU:
        return F.relu(y)
    if act == real.ACT_SIGMOID:
        return torch.sigmoid(y)
    if act == real.ACT_SQUARE_PLUS_ONE:
        return y * y + 1
    return y


def amp_takes(pc, x0, x1, stride, pad):
    """the shapes csrc/conv_f16.hip takes (launch_conv_f16 + the vector-gather geometry of deva_conv2d); everything
    else runs fp32 even under amp"""
    c0, c1 = x0.shape[1], 0 if x1 is None else x1.shape[1]
    h, w = x0.shape[-2:]
    return (pc.weight_f16 is not None and stride == 1 and pc.cout >= 64 and c0 % 64 == 0 and c1 % 64 == 0 and
            ((pc.kh == 1 and pad == 0) or (pc.kh == 3 and pad == 1)) and (h * w) % 4 == 0 and w >= 4)


def split_takes(pc, x0, x1, stride, pad):
    """the shapes the hi/lo split kernels take (csrc/conv_f16.hip: launch_conv_f16 with prec 2 + the vector-gather
    geometry of deva_conv2d); everything else runs the fp32 kernels although split is requested"""
    c0, c1 = x0.shape[1], 0 if x1 is None else x1.shape[1]
    h, w = x0.shape[-2:]
    whole = c0 % 32 == 0 and c1 % 32 == 0
    tail = pc.kh == 1 and (c0 % 32 == 0 if c1 else True)  # 1x1: a partial last K step (513 = 512 + 1 channels) is taken too
    return (pc.weight_split is not None and stride == 1 and pc.cout >= 64 and (whole or tail) and
            ((pc.kh == 1 and pad == 0) or (pc.kh == 3 and pad == 1)) and (h * w) % 4 == 0 and w >= 4)


_SPLIT_FALLBACKS = [0]


def split_fallbacks(device):
    return _SPLIT_FALLBACKS[0]


def conv2d(pc, x0, x1=None, *, stride=1, pad=0, relu_in=False, residual=None, act=real.ACT_NONE, out=None, amp=False,
           split=False):
    # split: the contract of the hi/lo split kernels IS the fp32 convolution (to fp32 round-off); only the fall-back
    # statistic is emulated (an input beyond the fp16 range sends the layer to the fp32 kernels)
    def _overlaps(t):  # (deva_conv2d: an output that overlaps an operand sends a split call to the fp32 kernels alone)
        return (t is not None and out is not None and t.untyped_storage().data_ptr() == out.untyped_storage().data_ptr())
    if split and split_takes(pc, x0, x1, stride, pad) and not any(_overlaps(t) for t in (x0, x1, residual)):
        for t in (x0, x1):
            if t is not None and not bool(((F.relu(t) if relu_in else t).abs() <= 65504.0).all()):
                _SPLIT_FALLBACKS[0] += 1
                break
    batch = max(x0.shape[0], 1 if x1 is None else x1.shape[0], 1 if residual is None else residual.shape[0])
    xs = [x0.expand(batch, -1, -1, -1)]
    if x1 is not None:
        xs.append(x1.expand(batch, -1, -1, -1))
    x = torch.cat(xs, 1)
    assert x.shape[1] == pc.cin
    if relu_in:
        x = F.relu(x)
    w = _unpack(pc)
    if amp and amp_takes(pc, x0, x1, stride, pad):  # fp16 operands (round to nearest even), fp32 products and sums
        x, w = x.half().float(), w.half().float()
    y = F.conv2d(x, w, pc.bias, stride=stride, padding=pad)
    if residual is not None:
        y = y + residual
    y = _act(y, act)
    if out is not None:
        out.copy_(y)
        return out
    return y


_real_pack_stem = real.pack_stem


def pack_stem(weight, bias=None, bn=None, device=None):
    return _real_pack_stem(weight, bias, bn, None)  # (deva_stem_pack is host code)


def stem7x7(ps, image, masks=None, relu=False):
    # contract of deva_stem7x7: the fp32 convolution over cat(image broadcast, masks) (to fp32 round-off); the fall-back
    # statistic counts calls with an input beyond the fp16 range
    batch = image.shape[0] if masks is None else masks.shape[0]
    x = image.expand(batch, -1, -1, -1)
    if masks is not None:
        x = torch.cat([x, masks], 1)
    if not bool((x.abs() <= 65504.0).all()):
        _SPLIT_FALLBACKS[0] += 1
    w = ps.w32.t().reshape(64, ps.cin, 7, 7)
    y = F.conv2d(x, w, ps.bias, stride=2, padding=3)
    return F.relu(y) if relu else y


def upsample2x_add_ds2(x, skip):
    return upsample2x_add(x, skip), area_downsample(x, 2)


def pad2d(x, pad):
    return F.pad(x, pad)


def usage_init(use, life):
    use.zero_()
    life.fill_(1e-7)


def gather_s2(x, kernel):
    b, c, h, w = x.shape
    pad = kernel // 2
    cols = F.unfold(x, kernel, padding=pad, stride=2)  # [B, C*k*k, L] with channel index c*k*k + t
    oh, ow = (h + 2 * pad - kernel) // 2 + 1, (w + 2 * pad - kernel) // 2 + 1
    return cols.view(b, c, kernel * kernel, oh, ow).transpose(1, 2).reshape(b, kernel * kernel * c, oh, ow).contiguous()


def maxpool3x3s2(x, relu_after=False):
    y = F.max_pool2d(x, 3, 2, 1)
    return F.relu(y) if relu_after else y


def upsample2x_add(x, skip):
    y = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    return y if skip is None else skip + y


def area_downsample(x, factor):
    lead = x.shape[:-2]
    y = F.avg_pool2d(x.reshape(1, -1, *x.shape[-2:]), factor)
    return y.reshape(*lead, *y.shape[-2:])


def aggregate(prob, apply_sigmoid=False):
    p = prob.float()
    if apply_sigmoid:
        p = torch.sigmoid(p)
    full = torch.cat([torch.prod(1 - p, dim=0, keepdim=True), p], 0).clamp(1e-7, 1 - 1e-7)
    return torch.log(full / (1 - full))


def softmax_channels(x):
    return torch.softmax(x, dim=0)


def upsample4x_softmax(logits, need_logits=True):
    up = F.interpolate(logits.unsqueeze(0), scale_factor=4, mode='bilinear', align_corners=False)[0]
    return (up if need_logits else None), torch.softmax(up, dim=0)


def cbam(x, w1, b1, w2, b2, spatial):
    hw = x.shape[-2:]

    def mlp(v):
        return F.linear(F.relu(F.linear(v, w1, b1)), w2, b2)

    att = mlp(F.avg_pool2d(x, hw).flatten(1)) + mlp(F.max_pool2d(x, hw).flatten(1))
    xs = x * torch.sigmoid(att)[:, :, None, None]
    pooled = torch.cat([xs.max(1, keepdim=True)[0], xs.mean(1, keepdim=True)], 1)
    gate = conv2d(spatial, pooled, pad=spatial.kh // 2)
    return x + xs * torch.sigmoid(gate)


def gru_update(values, h):
    c = h.shape[1]
    f, u, n = torch.sigmoid(values[:, :c]), torch.sigmoid(values[:, c:2 * c]), torch.tanh(values[:, 2 * c:])
    return f * h * (1 - u) + u * n


def _bank(key_long, n_long, key_work, n_work):
    parts = []
    if n_long:
        parts.append(key_long[:n_long])
    if n_work:
        parts.append(key_work[:n_work])
    return torch.cat(parts, 0)


def affinity_topk(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix=None,
                  splits=None, prep=None, prep_key=None):  # (prepared bank operands: a cache, no arithmetic to emulate)
    mk = _bank(key_long, n_long, key_work, n_work)          # [N,64] token-major
    ms = _bank(shr_long, n_long, shr_work, n_work)          # [N]
    if mk.shape[0] < k:
        raise real.DevaHipError('selected index k out of range')
    if not 1 <= k <= 64:
        raise real.DevaHipError(f'k={k} unsupported (1..64)')
    a_sq = mk.pow(2) @ qe
    two_ab = 2 * (mk @ (qk * qe))
    b_sq = (qe * qk.pow(2)).sum(0, keepdim=True)
    sim = (-a_sq + two_ab - b_sq) * ms[:, None] / 8.0
    vals, idx = torch.topk(sim, k=k, dim=0)                  # [k,hw]
    w = vals.exp()
    w = w / w.sum(0, keepdim=True)
    idx, w = idx.t().contiguous(), w.t().contiguous()
    if usage_fix is not None:
        usage_fix.index_add_(0, idx.reshape(-1), (w.reshape(-1).double() * TWO40).long())
    return idx.int(), w


def affinity_dense(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix=None):
    return affinity_topk(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix)


def usage_update(usage_fix, offset, use, life, n):
    seg = usage_fix[offset:offset + n]
    if use is not None:
        use[:n] += (seg.double() / TWO40).float()
    if life is not None:
        life[:n] += 1
    seg.zero_()


def readout_sparse(idx, weight, val_long, n_long, val_work, out, tok_range=None, row_map_long=None, row_map_work=None):
    hw, k = idx.shape
    cv = out.shape[0]
    w = weight
    if row_map_long is not None or row_map_work is not None:
        # value-sharded storage: token -> local row of its segment's arena, < 0 = another rank's row (weight 0)
        t = idx.long()
        is_long = t < n_long
        seg = torch.where(is_long, t, t - n_long)
        loc = torch.full_like(seg, -1)
        if n_long:
            loc = torch.where(is_long, (row_map_long.long()[seg.clamp(max=row_map_long.numel() - 1)] if row_map_long is not None else seg), loc)
        loc = torch.where(~is_long, (row_map_work.long()[seg.clamp(min=0, max=row_map_work.numel() - 1)] if row_map_work is not None else seg), loc)
        w = torch.where(loc >= 0, w, torch.zeros_like(w))
        rows_l = val_long if n_long else val_work
        g = torch.where(is_long.reshape(-1, 1), rows_l[loc.clamp(min=0).clamp(max=rows_l.shape[0] - 1).reshape(-1)],
                        val_work[loc.clamp(min=0).clamp(max=val_work.shape[0] - 1).reshape(-1)]).reshape(hw, k, cv)
        if tok_range is not None:
            w = torch.where((idx >= tok_range[0]) & (idx < tok_range[1]), w, torch.zeros_like(w))
        out.copy_((g * w[:, :, None]).sum(1).t().reshape(out.shape))
        return out
    n_work_needed = int(idx.max().item()) + 1 - n_long
    vals = _bank(val_long, n_long, val_work, max(n_work_needed, 0))
    g = vals[idx.long().reshape(-1)].reshape(hw, k, cv)
    if tok_range is not None:
        w = torch.where((idx >= tok_range[0]) & (idx < tok_range[1]), weight, torch.zeros_like(weight))
    out.copy_((g * w[:, :, None]).sum(1).t().reshape(out.shape))
    return out


def _orderable(score):
    """order-preserving 32-bit image of an fp32 score (as int64), like affinity.hip:orderable"""
    u = score.contiguous().view(torch.int32).long() & 0xffffffff
    return torch.where(u >= 0x80000000, (~u) & 0xffffffff, u | 0x80000000)


def affinity_candidates(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, token_offset=0,
                        splits=None):
    mk = _bank(key_long, n_long, key_work, n_work)
    ms = _bank(shr_long, n_long, shr_work, n_work)
    a_sq = mk.pow(2) @ qe
    two_ab = 2 * (mk @ (qk * qe))
    b_sq = (qe * qk.pow(2)).sum(0, keepdim=True)
    sim = (-a_sq + two_ab - b_sq) * ms[:, None] / 8.0
    vals, idx = torch.topk(sim, k=k, dim=0)
    hw = qk.shape[1]
    keys = torch.zeros((hw, 64), dtype=torch.int64)
    token = (idx.t().long() + token_offset)
    keys[:, :k] = (_orderable(vals.t()) << 32) | ((~token) & 0xffffffff)
    return keys, torch.full((hw,), k, dtype=torch.int32)


def affinity_merge(keys, counts, k, usage_fix=None):
    lists, hw = counts.shape
    live = torch.arange(64)[None, None, :] < counts[:, :, None]
    flat = torch.where(live, keys, torch.full_like(keys, -(1 << 62))).permute(1, 0, 2).reshape(hw, lists * 64)
    # keys are "unsigned": compare on (score bits, ~token) -- map to a sortable signed value
    signed = torch.where(live.permute(1, 0, 2).reshape(hw, -1), flat ^ (-(1 << 63)), torch.full_like(flat, -(1 << 63)))
    top = torch.topk(signed, k=k, dim=1)[0] ^ (-(1 << 63))
    o = (top >> 32) & 0xffffffff
    token = (~top) & 0xffffffff
    bits = torch.where(o >= 0x80000000, o & 0x7fffffff, (~o) & 0xffffffff)
    score = bits
    score = torch.where(score >= 0x80000000, score - (1 << 32), score).to(torch.int32).view(torch.float32)
    w = score.exp()
    w = w / w.sum(1, keepdim=True)
    if usage_fix is not None:
        usage_fix.index_add_(0, token.reshape(-1), (w.reshape(-1).double() * TWO40).long())
    return token.int(), w


def bank_append(src, arena, row0):
    arena[row0:row0 + src.shape[1]] = src.t()


def bank_gather_rows(src, rows, dst, count):
    dst[:count] = src[:count] if rows is None else src[rows[:count].long()]


def bank_export(arena, n):
    return arena[:n].t().contiguous()


def rank(x, n, descending, life=None):
    v = x[:n] / life[:n] if life is not None else x[:n]
    order = torch.sort(v, descending=descending, stable=True)[1]
    r = torch.empty(n, dtype=torch.int32)
    r[order] = torch.arange(n, dtype=torch.int32)
    return r, (v.clone() if life is not None else None)


def rank_select(rank_t, k):
    out = torch.empty(k, dtype=torch.int32)
    sel = rank_t < k
    out[rank_t[sel].long()] = torch.nonzero(sel).flatten().int()
    return out


def evict_select(x, rank_asc, n_remove):
    n = rank_asc.numel()
    thr = x[:n][rank_asc == n_remove - 1][0]
    keep = torch.nonzero(x[:n] > thr).flatten().int()
    idx = torch.zeros(n, dtype=torch.int32)
    idx[:keep.numel()] = keep
    return idx, torch.tensor([keep.numel()], dtype=torch.int32)


def similarity_dense(key, shr, sel, proto_idx, n_cand):
    p = proto_idx.numel()
    ld = (p + 31) // 32 * 32
    mk = key[:n_cand]
    qk, qe = key[proto_idx.long()].t(), sel[proto_idx.long()].t()   # [64,P]
    a_sq = mk.pow(2) @ qe
    two_ab = 2 * (mk @ (qk * qe))
    b_sq = (qe * qk.pow(2)).sum(0, keepdim=True)
    sim = torch.zeros(n_cand, ld)
    sim[:, :p] = (-a_sq + two_ab - b_sq) * shr[:n_cand, None] / 8.0
    return sim


def softmax_columns(x, p):
    x[:, :p] = torch.softmax(x[:, :p], dim=0)
    return x


def label_histogram(ours, news, new_ids, n_our):
    n_new = new_ids.numel()
    t = ours.reshape(-1).clone()
    t[(t < 0) | (t > n_our)] = 0
    col = torch.full_like(t, n_new)
    for j in range(n_new):
        col[news.reshape(-1) == new_ids[j]] = j
    flat = torch.bincount(t * (n_new + 1) + col, minlength=(n_our + 1) * (n_new + 1))
    return flat.reshape(n_our + 1, n_new + 1).int()


def merge_paint(ours, news, new_ids, our_order, our_label, new_order, new_label, out_ids):
    n_our, n_new = our_order.numel() - 1, new_ids.numel()
    t = ours.clone()
    t[(t < 0) | (t > n_our)] = 0
    order = torch.where(t > 0, our_order.long()[t], torch.full_like(t, -1))
    label = torch.where(order >= 0, our_label[t], torch.zeros_like(t))
    for j in range(n_new):
        sel = (news == new_ids[j]) & (new_order[j].long() >= order) & (new_order[j] >= 0)
        order = torch.where(sel, new_order[j].long().expand_as(order), order)
        label = torch.where(sel, new_label[j].expand_as(label), label)
    planes = [((order >= 0) & (label == o)).float() for o in out_ids]
    return torch.stack(planes, 0) if planes else torch.zeros((0, *ours.shape))


def lut_remap(mask, lut):
    ok = (mask >= 0) & (mask < lut.numel())
    return torch.where(ok, lut[mask.clamp(0, lut.numel() - 1)], torch.zeros_like(mask))


def index_mask(prob, size=None, lut=None):
    import torch.nn.functional as F
    if size is not None and tuple(size) != tuple(prob.shape[-2:]):
        prob = F.interpolate(prob.unsqueeze(1), tuple(size), mode='bilinear', align_corners=False)[:, 0]
    idx = torch.argmax(prob, dim=0)
    return idx if lut is None else lut_remap(idx, lut)


def input_head(image_u8, size=None, *, antialias=True, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
               pad=(0, 0, 0, 0)):
    import torch.nn.functional as F
    x = image_u8.permute(2, 0, 1).float() / 255
    x = (x - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    if size is not None and tuple(size) != tuple(x.shape[-2:]):
        x = F.interpolate(x.unsqueeze(0), tuple(size), mode='bilinear', align_corners=False, antialias=antialias)[0]
    return F.pad(x, tuple(pad))


def install(monkeypatch):
    """patch every public op of deva.hip.ops with its emulation"""
    for name in real.__all__ + ['require_hip']:
        if name in globals() and callable(globals()[name]) and name not in ('PackedConv',):
            monkeypatch.setattr(real, name, globals()[name])
