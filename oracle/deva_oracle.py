"""CPU oracle for DEVA's per-frame temporal-propagation path (fp32, PyTorch-CPU).

TEST INFRASTRUCTURE ONLY.  Nothing under `tracking-anything-with-deva_amd/` may
import this module: only tests/, bench.py's `cpu_baseline` leg and
`__graft_entry__.smoke()` use it, and only as the checker.

Parity status: PINNED.  The reference has no tests or golden vectors of its own
(SURVEY.md §4), so this restatement is pinned against outputs of the reference
itself, produced in the build container by `tests/golden/make_golden.py`
(imports /root/reference with the two shims of SURVEY.md §8c) and committed
under `tests/golden/`.  `tests/test_oracle_golden.py` replays them.

This is a functional restatement (flat state_dict + free functions) of the
algorithm in the reference files below; it shares no module structure with them.

  network      deva/model/network.py:33-173, big_modules.py:23-212,
               modules.py:19-169, group_modules.py:17-152, cbam.py:21-76,
               resnet.py:46-152
  memory ops   deva/model/memory_utils.py:6-76
  memory mgr   deva/inference/memory_manager.py:64-292, kv_memory_store.py:5-276
  frame loop   deva/inference/inference_core.py:55-113,200-290
  detections   deva/inference/inference_core.py:137-198, segment_merging.py:17-143,
               object_manager.py:27-110, object_info.py:11-31
  pad/unpad    deva/utils/tensor_utils.py:7-48
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------------------
# primitive layers (names index the flat state_dict)
# --------------------------------------------------------------------------------------


# --amp of this project's HIP path (NOT the reference's autocast, which also rounds every activation to fp16): the
# convolutions of the value encoder and the mask decoder that the fp16 kernels take (csrc/conv_f16.hip: 1x1 or 3x3/pad 1,
# stride 1, >= 64 output channels, input channels a multiple of 64, map of a multiple of 4 pixels) see their INPUTS and
# WEIGHTS rounded to fp16 (round to nearest even); products, sums, bias, everything else stay fp32.  Off by default.
AMP = False


class amp:
    """`with O.amp():` -- the oracle restates the amp arithmetic inside the block"""

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        global AMP
        self.prev, AMP = AMP, self.on
        return self

    def __exit__(self, *exc):
        global AMP
        AMP = self.prev


def _amp_takes(name: str, x: torch.Tensor, w: torch.Tensor, stride: int, padding: int) -> bool:
    if not AMP or not name.startswith(('mask_encoder.', 'mask_decoder.')):
        return False
    cout, cin, kh, kw = w.shape
    hh, ww = x.shape[-2:]
    return (stride == 1 and cout >= 64 and cin % 64 == 0 and ((kh == 1 and padding == 0) or (kh == 3 and padding == 1))
            and (hh * ww) % 4 == 0 and ww >= 4)


def _conv(P: Params, name: str, x: torch.Tensor, stride: int = 1, padding: int = 0):
    w = P[name + '.weight']
    if _amp_takes(name, x, w, stride, padding):
        x, w = x.half().float(), w.half().float()
    return F.conv2d(x, w, P.get(name + '.bias'), stride=stride, padding=padding)


def _bn(P: Params, name: str, x: torch.Tensor):
    # eval-mode batch norm, eps = 1e-5 (nn.BatchNorm2d default, resnet.py:52)
    return F.batch_norm(x, P[name + '.running_mean'], P[name + '.running_var'],
                        P[name + '.weight'], P[name + '.bias'], False, 0.0, 1e-5)


def _gconv(P: Params, name: str, g: torch.Tensor, padding: int = 0):
    # group_modules.py:41-45 -- fold the object axis into the batch axis
    b, n = g.shape[:2]
    y = _conv(P, name, g.flatten(0, 1), padding=padding)
    return y.view(b, n, *y.shape[1:])


def _resize_groups(g: torch.Tensor, ratio: float, mode: str):
    # group_modules.py:17-38
    b, n = g.shape[:2]
    kw = dict(align_corners=False) if mode == 'bilinear' else {}
    y = F.interpolate(g.flatten(0, 1), scale_factor=ratio, mode=mode, **kw)
    return y.view(b, n, *y.shape[1:])


def _bottleneck(P: Params, pre: str, x: torch.Tensor, stride: int):
    # resnet.py:78-114
    y = F.relu(_bn(P, pre + '.bn1', _conv(P, pre + '.conv1', x)))
    y = F.relu(_bn(P, pre + '.bn2', _conv(P, pre + '.conv2', y, stride=stride, padding=1)))
    y = _bn(P, pre + '.bn3', _conv(P, pre + '.conv3', y))
    if (pre + '.downsample.0.weight') in P:
        x = _bn(P, pre + '.downsample.1', _conv(P, pre + '.downsample.0', x, stride=stride))
    return F.relu(y + x)


def _conv_bn(P: Params, conv: str, bn: str, x: torch.Tensor, stride: int = 1, padding: int = 0):
    """conv -> eval-mode BatchNorm.  Under amp the HIP path rounds the BatchNorm-FOLDED weights (w * gamma / sqrt(var + eps),
    deva/hip/ops.py:pack_conv) to fp16, so the amp restatement folds first, in the same order of operations."""
    w = P[conv + '.weight']
    if not _amp_takes(conv, x, w, stride, padding):
        return _bn(P, bn, _conv(P, conv, x, stride=stride, padding=padding))
    scale = P[bn + '.weight'] / torch.sqrt(P[bn + '.running_var'] + 1e-5)
    shift = P[bn + '.bias'] - P[bn + '.running_mean'] * scale
    bias = P.get(conv + '.bias')
    bias = shift if bias is None else bias * scale + shift
    return F.conv2d(x.half().float(), (w * scale.view(-1, 1, 1, 1)).half().float(), bias, stride=stride, padding=padding)


def _basic_block(P: Params, pre: str, x: torch.Tensor, stride: int):
    # resnet.py:46-75
    y = F.relu(_conv_bn(P, pre + '.conv1', pre + '.bn1', x, stride=stride, padding=1))
    y = _conv_bn(P, pre + '.conv2', pre + '.bn2', y, padding=1)
    if (pre + '.downsample.0.weight') in P:
        x = _bn(P, pre + '.downsample.1', _conv(P, pre + '.downsample.0', x, stride=stride))
    return F.relu(y + x)


def _res_stage(P: Params, pre: str, x: torch.Tensor, blocks: int, stride: int, block_fn):
    for i in range(blocks):
        x = block_fn(P, f'{pre}.{i}', x, stride if i == 0 else 1)
    return x


def _group_res_block(P: Params, pre: str, g: torch.Tensor):
    # group_modules.py:48-67 (3x3, 3x3, optional 1x1 shortcut; relu BEFORE each conv)
    y = _gconv(P, pre + '.conv1', F.relu(g), padding=1)
    y = _gconv(P, pre + '.conv2', F.relu(y), padding=1)
    if (pre + '.downsample.weight') in P:
        g = _gconv(P, pre + '.downsample', g)
    return y + g


def _cbam(P: Params, pre: str, x: torch.Tensor):
    # cbam.py:21-76: channel gate (shared MLP over avg & max pooled vectors) then spatial gate
    def mlp(v):
        v = F.relu(F.linear(v, P[pre + '.ChannelGate.mlp.1.weight'], P[pre + '.ChannelGate.mlp.1.bias']))
        return F.linear(v, P[pre + '.ChannelGate.mlp.3.weight'], P[pre + '.ChannelGate.mlp.3.bias'])

    hw = (x.shape[2], x.shape[3])
    att = mlp(F.avg_pool2d(x, hw, stride=hw).flatten(1)) + mlp(F.max_pool2d(x, hw, stride=hw).flatten(1))
    x = x * torch.sigmoid(att)[:, :, None, None]
    pooled = torch.cat([x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)], 1)
    gate = _conv(P, pre + '.SpatialGate.spatial.conv', pooled, padding=3)
    return x * torch.sigmoid(gate)


def _group_fusion(P: Params, pre: str, x: torch.Tensor, g: torch.Tensor):
    # group_modules.py:133-152: cat(x broadcast, g) -> resblock -> +CBAM -> resblock
    b, n = g.shape[:2]
    g = torch.cat([x.unsqueeze(1).expand(-1, n, -1, -1, -1), g], 2)
    g = _group_res_block(P, pre + '.block1', g)
    r = _cbam(P, pre + '.attention', g.flatten(0, 1)).view_as(g)
    return _group_res_block(P, pre + '.block2', g + r)


def _gru(P: Params, name: str, g: torch.Tensor, h: torch.Tensor):
    # modules.py:141-149 / :162-169 -- "new value before forget gate" GRU variant
    c = h.shape[2]
    v = _gconv(P, name, torch.cat([g, h], 2), padding=1)
    forget = torch.sigmoid(v[:, :, :c])
    update = torch.sigmoid(v[:, :, c:2 * c])
    new = torch.tanh(v[:, :, 2 * c:])
    return forget * h * (1 - update) + update * new


# --------------------------------------------------------------------------------------
# network entry points (DEVA.encode_image / transform_key / encode_mask / segment / aggregate)
# --------------------------------------------------------------------------------------


def encode_image(P: Params, image: torch.Tensor):
    """big_modules.py:42-51.  image [1,3,H,W] -> (f16 [1,512,h,w], f8, f4), key-feature."""
    pe = 'pixel_encoder'
    x = F.relu(_bn(P, pe + '.bn1', _conv(P, pe + '.conv1', image, stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    f4 = _res_stage(P, pe + '.res2', x, 3, 1, _bottleneck)
    f8 = _res_stage(P, pe + '.layer2', f4, 4, 2, _bottleneck)
    f16 = _res_stage(P, pe + '.layer3', f8, 6, 2, _bottleneck)
    return (_conv(P, pe + '.proj1', f16), f8, f4), _conv(P, pe + '.proj2', f16)


def transform_key(P: Params, feat: torch.Tensor):
    """modules.py:73-78.  -> key [1,64,h,w], shrinkage [1,1,h,w] (>=1), selection (0,1)."""
    shrinkage = _conv(P, 'key_proj.d_proj', feat, padding=1)**2 + 1
    selection = torch.sigmoid(_conv(P, 'key_proj.e_proj', feat, padding=1))
    return _conv(P, 'key_proj.key_proj', feat, padding=1), shrinkage, selection


def encode_mask(P: Params, image: torch.Tensor, f16: torch.Tensor, sensory: torch.Tensor,
                masks: torch.Tensor, deep_update: bool = True):
    """big_modules.py:73-127 (chunk_size=-1 fast path).
    image [1,3,H,W]; masks [1,no,H,W]; sensory [1,no,512,h,w] -> value, new sensory."""
    me = 'mask_encoder'
    b, n = masks.shape[:2]
    g = torch.cat([image.unsqueeze(1).expand(-1, n, -1, -1, -1), masks.unsqueeze(2)], 2).flatten(0, 1)
    g = _bn(P, me + '.bn1', _conv(P, me + '.conv1', g, stride=2, padding=3))
    g = F.relu(F.max_pool2d(g, 3, 2, 1))  # maxpool BEFORE relu (big_modules.py:107-110)
    g = _res_stage(P, me + '.layer1', g, 2, 1, _basic_block)
    g = _res_stage(P, me + '.layer2', g, 2, 2, _basic_block)
    g = _res_stage(P, me + '.layer3', g, 2, 2, _basic_block)
    g = g.view(b, n, *g.shape[1:])
    value = _group_fusion(P, me + '.fuser', f16, g)
    if deep_update:
        sensory = _gru(P, me + '.sensory_update.transform', value, sensory)
    return value, sensory


def aggregate(prob: torch.Tensor, dim: int):
    """network.py:33-40: soft aggregation into logits with an explicit background channel."""
    prob = prob.float()
    full = torch.cat([torch.prod(1 - prob, dim=dim, keepdim=True), prob], dim).clamp(1e-7, 1 - 1e-7)
    return torch.log(full / (1 - full))


def segment(P: Params, ms_features: Sequence[torch.Tensor], readout: torch.Tensor,
            sensory: torch.Tensor, last_mask: torch.Tensor, update_sensory: bool = True):
    """network.py:94-173 + big_modules.py:147-212 (no aux, chunk_size=-1, all objects jointly).
    readout/sensory [1,no,512,h,w]; last_mask [1,no,H,W] -> sensory', logits, prob [1,no+1,H,W]."""
    md = 'mask_decoder'
    f16, f8, f4 = ms_features
    b, n = readout.shape[:2]
    last = F.interpolate(last_mask, size=readout.shape[-2:], mode='area').unsqueeze(2)
    d8 = _conv(P, md + '.decoder_feat_proc.transforms.0', f8)
    d4 = _conv(P, md + '.decoder_feat_proc.transforms.1', f4)

    p16 = readout + _gconv(P, md + '.sensory_compress', torch.cat([sensory, last], 2))
    p16 = _group_fusion(P, md + '.fuser', f16, p16)
    p8 = _group_res_block(P, md + '.up_16_8.out_conv',
                          d8.unsqueeze(1) + _resize_groups(p16, 2, 'bilinear'))
    p4 = _group_res_block(P, md + '.up_8_4.out_conv',
                          d4.unsqueeze(1) + _resize_groups(p8, 2, 'bilinear'))
    logits = _conv(P, md + '.pred', F.relu(p4.flatten(0, 1)), padding=1)  # [no,1,H/4,W/4]

    if update_sensory:
        su = md + '.sensory_update'
        p4x = torch.cat([p4, logits.view(b, n, 1, *logits.shape[-2:])], 2)
        g = _gconv(P, su + '.g16_conv', p16) + \
            _gconv(P, su + '.g8_conv', _resize_groups(p8, 1 / 2, 'area')) + \
            _gconv(P, su + '.g4_conv', _resize_groups(p4x, 1 / 4, 'area'))
        sensory = _gru(P, su + '.transform', g, sensory)

    logits = logits.view(b, n, *logits.shape[-2:])
    logits = aggregate(torch.sigmoid(logits), dim=1)
    logits = F.interpolate(logits, scale_factor=4, mode='bilinear', align_corners=False)
    return sensory, logits, F.softmax(logits, dim=1)


# --------------------------------------------------------------------------------------
# memory ops (memory_utils.py)
# --------------------------------------------------------------------------------------


def get_similarity(mk: torch.Tensor, ms: Optional[torch.Tensor], qk: torch.Tensor,
                   qe: Optional[torch.Tensor]):
    """memory_utils.py:6-45.  mk [CK,N], ms [1,N], qk/qe [CK,Q] -> [N,Q] (no batch dim)."""
    ck = mk.shape[0]
    mkt = mk.t()
    if qe is not None:
        a_sq = mkt.pow(2) @ qe
        two_ab = 2 * (mkt @ (qk * qe))
        b_sq = (qe * qk.pow(2)).sum(0, keepdim=True)
        sim = -a_sq + two_ab - b_sq
    else:
        sim = -mk.pow(2).sum(0).unsqueeze(1) + 2 * (mkt @ qk)
    if ms is not None:
        return sim * ms.reshape(-1, 1) / math.sqrt(ck)
    return sim / math.sqrt(ck)


def topk_softmax(sim: torch.Tensor, k: int):
    """memory_utils.py:57-60.  top-k over N per query column, exp WITHOUT max subtraction.
    sim [N,Q] -> values-sorted indices [k,Q], weights [k,Q]."""
    values, indices = torch.topk(sim, k=k, dim=0)
    w = values.exp()
    w = w / w.sum(0, keepdim=True)
    return indices, w


def dense_affinity(sim: torch.Tensor, k: Optional[int]):
    """memory_utils.py:48-76 -> dense [N,Q] affinity and usage [N] (= row sums)."""
    if k is not None:
        idx, w = topk_softmax(sim, k)
        aff = torch.zeros_like(sim).scatter_(0, idx, w)
    else:
        e = torch.exp(sim - sim.max(0, keepdim=True)[0])
        aff = e / e.sum(0, keepdim=True)
    return aff, aff.sum(1)


# --------------------------------------------------------------------------------------
# three-tier memory (memory_manager.py + kv_memory_store.py), restated with one store class
# --------------------------------------------------------------------------------------


class _Store:
    """kv_memory_store.py:5-276, channel-major tensors grown by concatenation."""

    def __init__(self, keep_selection: bool, keep_usage: bool):
        self.keep_selection, self.keep_usage = keep_selection, keep_usage
        self.next_bucket = 0
        self.buckets: Dict[int, List[int]] = {}
        self.k: Dict[int, torch.Tensor] = {}
        self.s: Dict[int, torch.Tensor] = {}
        self.e: Dict[int, torch.Tensor] = {}
        self.use: Dict[int, torch.Tensor] = {}
        self.life: Dict[int, torch.Tensor] = {}
        self.v: Dict[int, torch.Tensor] = {}

    def size(self, b: int) -> int:
        return self.k[b].shape[-1] if b in self.k else 0

    def add(self, key, values: Dict[int, torch.Tensor], shrinkage, selection, bucket_id: int = -1):
        # kv_memory_store.py:35-116
        if bucket_id >= 0:
            touched = {bucket_id}
            exists = bucket_id in self.buckets
            for o, val in values.items():
                self.v[o] = torch.cat([self.v[o], val], -1) if exists else val
            self.buckets[bucket_id] = list(values.keys())
        else:
            touched, fresh = set(), None
            for o, val in values.items():
                if o in self.v:
                    self.v[o] = torch.cat([self.v[o], val], -1)
                    touched.add(next(b for b, objs in self.buckets.items() if o in objs))
                else:
                    self.v[o] = val
                    if fresh is None:
                        fresh = self.next_bucket
                        self.next_bucket += 1
                        self.buckets[fresh] = []
                    self.buckets[fresh].append(o)
                    touched.add(fresh)
        n = key.shape[1]
        for b in self.buckets:
            if b not in touched:
                continue
            cat = (lambda d, x: torch.cat([d[b], x], -1)) if b in self.k else (lambda d, x: x)
            pairs = [(self.k, key), (self.s, shrinkage)]
            if self.keep_selection:
                pairs.append((self.e, selection))
            if self.keep_usage:
                pairs.append((self.use, torch.zeros(n)))
                pairs.append((self.life, torch.zeros(n) + 1e-7))
            for d, x in [(d, cat(d, x)) for d, x in pairs]:
                d[b] = x

    def bump_usage(self, b: int, usage: torch.Tensor):
        # kv_memory_store.py:118-125
        if self.keep_usage:
            self.use[b] = self.use[b] + usage
            self.life[b] = self.life[b] + 1

    def normalized_usage(self, b: int):
        if not self.keep_usage:
            raise RuntimeError('I did not count usage!')
        return self.use[b] / self.life[b]

    def _select(self, b: int, pick):
        """apply `pick` (a function tensor -> tensor along the token axis) to everything in bucket b"""
        self.k[b], self.s[b] = pick(self.k[b]), pick(self.s[b])
        if self.keep_selection:
            self.e[b] = pick(self.e[b])
        if self.keep_usage:
            self.use[b], self.life[b] = pick(self.use[b]), pick(self.life[b])
        for o in self.buckets[b]:
            self.v[o] = pick(self.v[o])

    def drop_range(self, b: int, start: int, end: int, min_size: int):
        # kv_memory_store.py:127-159 (end is negative or 0 meaning "to the end")
        if self.size(b) <= min_size:
            return
        stop = self.size(b) if end == 0 else end
        self._select(b, lambda t: torch.cat([t[..., :start], t[..., stop:]], -1))

    def evict_least_used(self, b: int, max_size: int):
        # kv_memory_store.py:164-185: everything at or below the (size-max)-th smallest usage goes
        usage = self.normalized_usage(b).flatten()
        worst, _ = torch.topk(usage, k=self.size(b) - max_size, largest=False, sorted=True)
        keep = usage > worst[-1]
        self._select(b, lambda t: t[..., keep])

    def purge_except(self, keep_objs):
        # kv_memory_store.py:216-239
        keep_objs = set(keep_objs)
        for b in list(self.buckets):
            self.buckets[b] = [o for o in self.buckets[b] if o in keep_objs]
            if not self.buckets[b]:
                for d in (self.buckets, self.k, self.s, self.e, self.use, self.life):
                    d.pop(b, None)
        self.v = {o: v for o, v in self.v.items() if o in keep_objs}


class OracleMemory:
    """memory_manager.py:14-292."""

    def __init__(self, cfg: Dict):
        self.cfg = cfg
        self.top_k = cfg['top_k']
        self.long_term = cfg['enable_long_term']
        self.count_lt = cfg['enable_long_term_count_usage']
        self.work = _Store(self.long_term, self.long_term)
        self.long = _Store(False, self.count_lt) if self.long_term else None
        self.sensory: Dict[int, torch.Tensor] = {}
        self.engaged = False
        self.HW = None

    def match(self, key: torch.Tensor, selection: torch.Tensor) -> Dict[int, torch.Tensor]:
        # memory_manager.py:91-169
        h, w = key.shape[-2:]
        qk, qe = key[0].flatten(1), selection[0].flatten(1)
        out = {}
        for b, objs in self.work.buckets.items():
            use_long = self.long_term and b in self.long.buckets
            if use_long:
                n_long = self.long.size(b)
                mk = torch.cat([self.long.k[b], self.work.k[b]], -1)
                ms = torch.cat([self.long.s[b], self.work.s[b]], -1)
            else:
                mk, ms = self.work.k[b], self.work.s[b]
            aff, usage = dense_affinity(get_similarity(mk, ms, qk, qe), self.top_k)
            if use_long:
                self.work.bump_usage(b, usage[n_long:])
                if self.count_lt:
                    self.long.bump_usage(b, usage[:n_long])
            elif self.long_term:
                self.work.bump_usage(b, usage)
            for o in objs:
                v = self.work.v[o]
                if use_long and o in self.long.v:
                    v = torch.cat([self.long.v[o], v], -1)
                out[o] = (v @ aff).view(-1, h, w)
        return out

    def add(self, key, shrinkage, value, objects: List[int], selection):
        # memory_manager.py:171-218
        self.engaged = True
        if self.HW is None:
            self.HW = value.shape[-2] * value.shape[-1]
        value = value[0].flatten(2)
        self.work.add(key[0].flatten(1), {o: value[i] for i, o in enumerate(objects)},
                      shrinkage[0].flatten(1), selection[0].flatten(1))
        if not self.long_term:
            return
        t_max = self.cfg['max_mid_term_frames'] * self.HW
        for b in self.work.buckets:
            if self.work.size(b) >= t_max:
                cap = self.cfg['max_long_term_elements'] - self.cfg['num_prototypes']
                if self.long.size(b) >= cap:
                    self.long.evict_least_used(b, cap)
                self._consolidate(b)

    def _consolidate(self, b: int):
        # memory_manager.py:231-276: middle frames -> P prototypes chosen by usage
        hw = self.HW
        t_min = self.cfg['min_mid_term_frames'] * hw
        lo, hi = hw, -t_min + hw
        sl = (lambda t: t[..., lo:]) if hi == 0 else (lambda t: t[..., lo:hi])
        ck, cs, ce = sl(self.work.k[b]), sl(self.work.s[b]), sl(self.work.e[b])
        usage = sl(self.work.normalized_usage(b))
        proto = torch.topk(usage, k=self.cfg['num_prototypes'], dim=-1, sorted=True)[1].flatten()
        aff, _ = dense_affinity(get_similarity(ck, cs, ck[:, proto], ce[:, proto]), None)
        values = {o: sl(self.work.v[o]) @ aff for o in self.work.buckets[b]}
        self.work.drop_range(b, lo, hi, min_size=t_min + hw)
        self.long.add(ck[:, proto], values, cs @ aff, None, bucket_id=b)

    def purge_except(self, keep_objs):
        # memory_manager.py:220-229
        self.work.purge_except(keep_objs)
        if self.long_term and self.long.buckets:
            self.long.purge_except(keep_objs)
        self.sensory = {o: s for o, s in self.sensory.items() if o in keep_objs}
        if not self.work.buckets:
            self.engaged = False


# --------------------------------------------------------------------------------------
# frame loop (inference_core.py) for integer object ids given on annotated frames
# --------------------------------------------------------------------------------------


def pad_to_multiple(x: torch.Tensor, d: int = 16):
    """tensor_utils.py:7-22: symmetric zero pad, odd remainder goes bottom/right."""
    h, w = x.shape[-2:]
    nh, nw = (h + d - 1) // d * d, (w + d - 1) // d * d
    lh, lw = (nh - h) // 2, (nw - w) // 2
    pad = (lw, nw - w - lw, lh, nh - h - lh)
    return F.pad(x, pad), pad


def unpad(x: torch.Tensor, pad):
    """tensor_utils.py:25-48."""
    lw, uw, lh, uh = pad
    h, w = x.shape[-2:]
    return x[..., lh:h - uh, lw:w - uw]


class OracleCore:
    """inference_core.py:17-113,200-290 for the VOS call pattern: `step(image, mask, objects)`
    with hard integer masks on annotated frames, `step(image)` elsewhere."""

    def __init__(self, params: Params, cfg: Dict):
        self.P, self.cfg = params, cfg
        self.memory = OracleMemory(cfg)
        self.objects: List[int] = []  # tmp id i+1 <-> self.objects[i]  (object_manager.py:8-70)
        self.curr_ti, self.last_mem_ti = -1, 0
        self.last_mask = None
        self.trace: Dict[str, torch.Tensor] = {}

    def _segment(self, key, selection, ms, update_sensory=True):
        # inference_core.py:89-113
        ro = self.memory.match(key, selection)
        ro = torch.stack([ro[o] for o in self.objects], 0).unsqueeze(0)
        sens = torch.stack([self.memory.sensory[o] for o in self.objects], 0).unsqueeze(0)
        sens, logits, prob = segment(self.P, ms, ro, sens, self.last_mask, update_sensory)
        self.trace.update(readout=ro, logits=logits)
        if update_sensory:
            for i, o in enumerate(self.objects):
                self.memory.sensory[o] = sens[0, i]
        return prob[0]

    def _add_memory(self, image, ms, prob, key, shrinkage, selection):
        # inference_core.py:55-87
        for o in self.objects:
            if o not in self.memory.sensory:
                self.memory.sensory[o] = torch.zeros(self.cfg['value_dim'], *key.shape[-2:])
        sens = torch.stack([self.memory.sensory[o] for o in self.objects], 0).unsqueeze(0)
        value, sens = encode_mask(self.P, image, ms[0], sens, prob)
        self.memory.add(key, shrinkage, value, self.objects, selection)
        self.last_mem_ti = self.curr_ti
        for i, o in enumerate(self.objects):
            self.memory.sensory[o] = sens[0, i]

    def step(self, image: torch.Tensor, mask: Optional[torch.Tensor] = None,
             objects: Optional[List[int]] = None, end: bool = False) -> torch.Tensor:
        # inference_core.py:200-290
        self.curr_ti += 1
        image, pad = pad_to_multiple(image)
        image = image.unsqueeze(0)
        is_mem = ((self.curr_ti - self.last_mem_ti >= self.cfg['mem_every']) or mask is not None) and not end
        need_seg = mask is None or (len(self.objects) > 0 and not all(o in self.objects for o in objects))
        ms, feat = encode_image(self.P, image)
        key, shrinkage, selection = transform_key(self.P, feat)
        if need_seg:
            prob = self._segment(key, selection, ms, update_sensory=not end)
        if mask is not None:
            n_old = len(self.objects)
            new_tmp = list(range(n_old + 1, n_old + 1 + len(objects)))
            self.objects = self.objects + list(objects)
            mask, _ = pad_to_multiple(mask)
            if need_seg:
                fg = prob[1:]
                fg[:, mask > 0] = 0
                extra = []
                for j, tmp in enumerate(new_tmp):
                    m = (mask == objects[j]).type_as(fg)
                    if tmp >= fg.shape[0]:
                        extra.append(m.unsqueeze(0))
                    else:
                        fg[tmp + 1] = m  # reference indexing quirk, inference_core.py:268-270
                mask = torch.cat([fg, *extra], 0)
            else:
                mask = torch.stack([mask == o for o in objects], 0)
            prob = torch.softmax(aggregate(mask, dim=0), dim=0)
        self.last_mask = prob[1:].unsqueeze(0)
        if is_mem:
            self._add_memory(image, ms, self.last_mask, key, shrinkage, selection)
        return unpad(prob, pad)


# --------------------------------------------------------------------------------------
# detections + propagation (evaluation/eval_with_detections.py call pattern)
# --------------------------------------------------------------------------------------


def merge_detection(forward: torch.Tensor, detected: torch.Tensor, table: List[Dict], segments: List[Dict],
                    history: set, max_num_objects: int = -1, incremental: bool = False):
    """segment_merging.py:17-143 + the ObjectManager side effects it triggers
    (object_manager.py:27-66), on a plain object table.

    forward: H*W mask in tmp ids (position in `table` + 1); detected: H*W mask in detection ids.
    table: tracked objects in tmp order, dicts {id, isthing, cats, poke}; segments: detections
    {id, category_id, isthing}.  `history`: every id ever handed out (collisions are re-drawn from
    np.random like the reference).  Returns the one-hot merged masks [len(table'), H, W] (bool);
    `table` and `history` are updated in place."""
    import numpy as np
    forward, detected = forward.long(), detected.long()
    ours = [forward == (i + 1) for i in range(len(table))]
    if max_num_objects > 0 and len(table) + len(segments) > max_num_objects:
        segments = []  # too many objects: every new detection is denied (segment_merging.py:115-122)
    news = [detected == s['id'] for s in segments]
    our_area = [int(m.sum()) for m in ours]
    new_area = [int(m.sum()) for m in news]
    merged = torch.zeros_like(forward)
    n_tracked_before = len(table)

    for status in (None, False, True):  # untyped, stuff, things are merged separately
        partner: Dict[int, int] = {}  # tracked index -> detection index
        queue = []                     # (area, kind, index) in the reference's insertion order
        for j, seg in enumerate(segments):
            if seg['isthing'] != status:
                continue
            hit = None
            for i in range(n_tracked_before):
                if table[i]['isthing'] != status or i in partner:
                    continue
                inter = int((news[j] & ours[i]).sum())
                if inter < 1e-3:
                    continue
                union = new_area[j] + our_area[i] - inter
                if inter / union > 0.5:
                    hit = (i, union)
                    break
            if hit is None:
                queue.append((new_area[j], 'new', j))
            else:
                partner[hit[0]] = j
                queue.append((hit[1], 'ours', hit[0]))
        for i in range(n_tracked_before):
            if table[i]['isthing'] == status and i not in partner:
                queue.append((our_area[i], 'ours', i))
        # large areas are painted first so that small segments end up on top (stable order on ties)
        for _, kind, idx in sorted(queue, key=lambda item: item[0], reverse=True):
            if kind == 'new':
                seg = segments[idx]
                new_id = seg['id']
                tries = 0
                while new_id in history:
                    new_id = int(np.random.randint(1, 256))
                    tries += 1
                    if tries > 5000:
                        raise ValueError('no free object id')
                history.add(new_id)
                table.append(dict(id=new_id, isthing=seg['isthing'], cats=[seg['category_id']], poke=0))
                merged[news[idx]] = new_id
                continue
            rec = table[idx]
            merged[ours[idx]] = rec['id']
            if idx in partner:
                j = partner[idx]
                merged[news[j]] = rec['id']
                rec['cats'].append(segments[j]['category_id'])
                rec['poke'] = 0
            elif incremental:
                rec['poke'] = rec['poke'] + 1 if our_area[idx] < 1 else 0
            else:
                rec['poke'] += 1
    if not table:
        return torch.zeros((0, *merged.shape), dtype=torch.bool)
    return torch.stack([merged == rec['id'] for rec in table], 0)


class OracleDetectionCore(OracleCore):
    """OracleCore + `incorporate_detection` (inference_core.py:137-198) for the online
    detections-with-propagation setting."""

    def __init__(self, params: Params, cfg: Dict):
        super().__init__(params, cfg)
        self.table: List[Dict] = []
        self.history: set = set()

    def incorporate_detection(self, image: torch.Tensor, new_mask: torch.Tensor, segments: List[Dict],
                              incremental: bool = False) -> torch.Tensor:
        self.curr_ti += 1
        image, pad = pad_to_multiple(image)
        new_mask, _ = pad_to_multiple(new_mask)
        image = image.unsqueeze(0)
        ms, feat = encode_image(self.P, image)
        key, shrinkage, selection = transform_key(self.P, feat)
        if self.memory.engaged:
            forward_prob = self._segment(key, selection, ms)
            forward = torch.argmax(forward_prob, dim=0)
            self.trace.update(forward_prob=unpad(forward_prob, pad))
        else:
            forward = torch.zeros_like(new_mask)
            self.trace.pop('forward_prob', None)
        merged = merge_detection(forward, new_mask, self.table, segments, self.history,
                                 max_num_objects=self.cfg.get('max_num_objects', -1), incremental=incremental)
        # retire objects unseen for too long (object_manager.py:89-110) and their memories
        limit = self.cfg['max_missed_detection_count']
        keep = [i for i, rec in enumerate(self.table) if rec['poke'] <= limit]
        if len(keep) != len(self.table):
            self.table = [self.table[i] for i in keep]
            self.memory.purge_except([rec['id'] for rec in self.table])
            merged = merged[keep]
        self.objects = [rec['id'] for rec in self.table]
        self.last_mask = merged.unsqueeze(0).type_as(key)
        if self.last_mask.shape[1] > 0:  # inference_core.py:67-70: nothing to memorise
            self._add_memory(image, ms, self.last_mask, key, shrinkage, selection)
        return unpad(aggregate(self.last_mask[0], dim=0), pad)
