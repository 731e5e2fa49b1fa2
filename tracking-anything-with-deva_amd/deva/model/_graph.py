"""Parameter tree and execution graph of the DEVA propagation network on libdeva_hip.

Two halves:

* `build_parameter_tree` creates bare `nn.Module` containers whose `state_dict()` has exactly the
  420 tensor names/shapes of the reference checkpoint (`DEVA-propagation.pth`; reference modules
  deva/model/big_modules.py, modules.py, group_modules.py, cbam.py, resnet.py).  The modules are
  parameter holders only -- their `forward` is never called.
* `CompiledGraph` is built once per weight load: it folds every eval-mode BatchNorm into its
  convolution, repacks all weights into the [K][cout] layout of `deva_conv2d`, and exposes the
  four network stages as sequences of HIP kernel launches.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from deva.hip import ops
from deva.hip.ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SQUARE_PLUS_ONE, PackedConv

# ------------------------------------------------------------------------------------------------
# parameter tree (names follow the reference checkpoint)
# ------------------------------------------------------------------------------------------------


def _box(**children) -> nn.Module:
    m = nn.Module()
    for name, child in children.items():
        m.add_module(name, child)
    return m


def _conv(cin, cout, k, stride=1, bias=True) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=bias)


def _resnet_stage(kind: str, cin: int, planes: int, blocks: int, stride: int) -> nn.Sequential:
    """torchvision-style stage; kind 'bottleneck' (expansion 4) or 'basic' (expansion 1)"""
    exp = 4 if kind == 'bottleneck' else 1
    layers = []
    for i in range(blocks):
        s = stride if i == 0 else 1
        inp = cin if i == 0 else planes * exp
        if kind == 'bottleneck':
            blk = _box(conv1=_conv(inp, planes, 1, bias=False), bn1=nn.BatchNorm2d(planes),
                       conv2=_conv(planes, planes, 3, stride=s, bias=False), bn2=nn.BatchNorm2d(planes),
                       conv3=_conv(planes, planes * 4, 1, bias=False), bn3=nn.BatchNorm2d(planes * 4))
        else:
            blk = _box(conv1=_conv(inp, planes, 3, stride=s, bias=False), bn1=nn.BatchNorm2d(planes),
                       conv2=_conv(planes, planes, 3, bias=False), bn2=nn.BatchNorm2d(planes))
        if i == 0 and (s != 1 or inp != planes * exp):
            blk.add_module('downsample', nn.Sequential(_conv(inp, planes * exp, 1, stride=s, bias=False),
                                                       nn.BatchNorm2d(planes * exp)))
        layers.append(blk)
    return nn.Sequential(*layers)


def _group_res_block(cin: int, cout: int) -> nn.Module:
    blk = _box(conv1=_conv(cin, cout, 3), conv2=_conv(cout, cout, 3))
    if cin != cout:
        blk.add_module('downsample', _conv(cin, cout, 1))
    return blk


def _fusion_block(x_dim: int, g_dim: int, mid: int, out: int) -> nn.Module:
    mlp = nn.Sequential(nn.Identity(), nn.Linear(mid, mid // 16), nn.Identity(), nn.Linear(mid // 16, mid))
    attention = _box(ChannelGate=_box(mlp=mlp), SpatialGate=_box(spatial=_box(conv=_conv(2, 1, 7))))
    return _box(block1=_group_res_block(x_dim + g_dim, mid), attention=attention,
                block2=_group_res_block(mid, out))


def build_parameter_tree(pix_feat_dim: int, key_dim: int, value_dim: int) -> Dict[str, nn.Module]:
    pixel_encoder = _box(conv1=_conv(3, 64, 7, stride=2, bias=False), bn1=nn.BatchNorm2d(64),
                         res2=_resnet_stage('bottleneck', 64, 64, 3, 1),
                         layer2=_resnet_stage('bottleneck', 256, 128, 4, 2),
                         layer3=_resnet_stage('bottleneck', 512, 256, 6, 2),
                         proj1=_conv(1024, pix_feat_dim, 1), proj2=_conv(1024, pix_feat_dim, 1))
    mask_encoder = _box(conv1=_conv(4, 64, 7, stride=2, bias=False), bn1=nn.BatchNorm2d(64),
                        layer1=_resnet_stage('basic', 64, 64, 2, 1),
                        layer2=_resnet_stage('basic', 64, 128, 2, 2),
                        layer3=_resnet_stage('basic', 128, 256, 2, 2),
                        fuser=_fusion_block(pix_feat_dim, 256, value_dim, value_dim),
                        sensory_update=_box(transform=_conv(value_dim * 2, value_dim * 3, 3)))
    key_proj = _box(key_proj=_conv(pix_feat_dim, key_dim, 3), d_proj=_conv(pix_feat_dim, 1, 3),
                    e_proj=_conv(pix_feat_dim, key_dim, 3))
    mask_decoder = _box(
        fuser=_fusion_block(512, value_dim, value_dim, value_dim),
        sensory_compress=_conv(value_dim + 1, value_dim, 1),
        sensory_update=_box(g16_conv=_conv(value_dim, 512, 1), g8_conv=_conv(256, 512, 1),
                            g4_conv=_conv(256 + 1, 512, 1), transform=_conv(512 + 512, 512 * 3, 3)),
        decoder_feat_proc=_box(transforms=nn.ModuleList([_conv(512, value_dim, 1), _conv(256, 256, 1)])),
        up_16_8=_box(out_conv=_group_res_block(value_dim, 256)),
        up_8_4=_box(out_conv=_group_res_block(256, 256)),
        pred=_conv(256, 1, 3),
        sensory_linear_pred=_box(projection=_conv(value_dim, 512 + 1, 1)))
    return dict(pixel_encoder=pixel_encoder, mask_encoder=mask_encoder, key_proj=key_proj,
                mask_decoder=mask_decoder)


# ------------------------------------------------------------------------------------------------
# compiled graph
# ------------------------------------------------------------------------------------------------


class CompiledGraph:
    """Packed weights + the launch sequences of the four stages (all tensors fp32 NCHW on HIP)."""

    # modules whose convolutions take fp16 operands under --amp: the value encoder and the mask decoder.  The key
    # encoder and the key projection stay exact fp32 (the memory read's top-k stays bit-faithful), and so do the
    # reference's own fp32 islands (network.py:34 aggregate, big_modules.py:189 pred: a single-channel VALU kernel here)
    AMP_SCOPES = ('mask_encoder.', 'mask_decoder.')
    # modules whose convolutions run the fp32-accurate hi/lo fp16 split on the f16 matrix pipes under --f16_split (the
    # same two: 90 % of the frame's flop at 5 objects); the key encoder and the key projection stay on the fp32 kernels
    # whose FMA order the affinity tests pin, so the inputs of the memory read's top-k do not move by a bit
    SPLIT_SCOPES = AMP_SCOPES
    # modules whose 3x3 stride-1 convolutions carry Winograd-transformed weights on the plain fp32 path (csrc/conv_wino.hip:
    # F(2x2, 3x3), 2.25x fewer MFMAs, round-off BELOW the direct kernels' -- a quarter of the accumulated terms; the library
    # takes it only for layers that fill the chip).  The key encoder's bottleneck 3x3s are among them from 1080p up (layer1 /
    # layer2 at 1080p, layer3 too at 4K: 13 launches = 30 % of a 4K frame on the direct kernels); at 480p they stay below the
    # threshold, so the keys of the headline configuration are the direct kernels' bit for bit.  The key projection (1x1 /
    # 3x3 at 1/16 resolution, 64 + 64 + 64 channels) is never eligible.
    WINO_SCOPES = AMP_SCOPES + ('pixel_encoder.',)

    def __init__(self, sd: Dict[str, torch.Tensor], device: torch.device, amp: bool = False, split: bool = False,
                 split_key_encoder: bool = False, winograd: bool = True):
        ops.require_hip(device, 'DEVA network')
        if amp and split:
            raise ValueError('amp (fp16 operands) and f16_split (fp32-accurate on the f16 pipes) are alternatives: pick one')
        self.device = device
        self.sd = sd
        self.amp = bool(amp)
        self.f16_split = bool(split)
        self.winograd = bool(winograd) and not amp and not split
        # second level of the opt-in: the key encoder (ResNet-50 stages + the two projections; 6 % of the flop at 5 objects,
        # 3.7 ms of a 1080p frame on the fp32 kernels) on the split kernels too.  Its outputs feed the key projection --
        # which stays on the fp32 kernels either way -- so the memory read's inputs then differ from the fp32 run's by
        # fp32 round-off (like any change of accumulation order), no longer bit for bit
        self.split_scopes = self.SPLIT_SCOPES + (('pixel_encoder.',) if split_key_encoder else ())
        if split_key_encoder and not split:
            raise ValueError('f16_split_key_encoder extends f16_split: set both')
        self.convs: Dict[str, PackedConv] = {}
        self.s2: Dict[str, PackedConv] = {}  # stride-2 convolutions of the split scopes as 1x1 convolutions over their taps
        self.vecs: Dict[str, torch.Tensor] = {}
        for name in sd:
            if not name.endswith('.weight') or sd[name].dim() != 4:
                continue
            base = name[:-len('.weight')]
            self.convs[base] = ops.pack_conv(sd[name], sd.get(base + '.bias'), self._bn_after(base), device,
                                             amp=self._amp_of(base), split=self._split_of(base), wino=self._wino_of(base))
        for name, t in sd.items():
            if '.ChannelGate.mlp.' in name:
                self.vecs[name] = t.detach().float().contiguous().to(device)
        # Convolutions over cat(image feature broadcast to every object, per-object feature): by linearity
        # W.[x; g] = Wx.x + Wg.g, and Wx.x is the same for all objects.  With two or more objects the image
        # part is computed ONCE (batch 1) and enters the per-object convolution as its fused residual; the
        # reference (group_modules.py:141-146, big_modules.py:96-101) expands x and convolves it per object.
        # Saves (no-1)/no of 7.6 GF per object in each fuser's first 3x3 convolution at 480p.
        self.split: Dict[str, Tuple[PackedConv, PackedConv]] = {}
        x_dim = sd['pixel_encoder.proj1.weight'].shape[0]  # channels of the image feature fed to both fusers
        for base, cx in (('mask_decoder.fuser.block1.conv1', x_dim), ('mask_decoder.fuser.block1.downsample', x_dim),
                         ('mask_encoder.fuser.block1.conv1', x_dim), ('mask_encoder.fuser.block1.downsample', x_dim),
                         ('mask_encoder.conv1', 3)):
            if base + '.weight' in sd:
                self.split[base] = self._split_pack(base, cx)
        # the 7x7 stride-2 stems under --f16_split*: a direct convolution on the f16 matrix pipes (csrc/conv_stem.hip) instead
        # of the scalar-gather fp32 kernel (3 / 4 input channels give the implicit GEMM nothing to tile); the value encoder's
        # reads cat(image, mask) directly, like the reference (big_modules.py:103-107), instead of image part + mask part
        self.stems: Dict[str, ops.PackedStem] = {}
        for base in ('pixel_encoder.conv1', 'mask_encoder.conv1'):
            if self._split_of(base):
                self.stems[base] = ops.pack_stem(sd[base + '.weight'], sd.get(base + '.bias'), self._bn_after(base), device)
        # the two 1x1 projections of the key encoder read the same feature map: one launch with their output
        # channels side by side (big_modules.py:42-51 runs them one after the other)
        w1, w2 = sd['pixel_encoder.proj1.weight'], sd['pixel_encoder.proj2.weight']
        self.proj_split = w1.shape[0]
        self.convs['pixel_encoder.proj12'] = ops.pack_conv(
            torch.cat([w1, w2], 0), torch.cat([sd['pixel_encoder.proj1.bias'], sd['pixel_encoder.proj2.bias']], 0), None,
            device, split=self._split_of('pixel_encoder.proj12'))

    def _amp_of(self, base: str) -> bool:
        return self.amp and base.startswith(self.AMP_SCOPES)

    def _split_of(self, base: str) -> bool:
        return self.f16_split and base.startswith(self.split_scopes)

    def _wino_of(self, base: str) -> bool:
        return self.winograd and base.startswith(self.WINO_SCOPES)

    def _split_pack(self, base: str, cx: int) -> Tuple[PackedConv, PackedConv]:
        """(image part without bias, per-object part with the bias) of convolution `base`, BatchNorm folded"""
        w, b = self._folded(base)
        kw = dict(amp=self._amp_of(base), split=self._split_of(base), wino=self._wino_of(base))
        return (ops.pack_conv(w[:, :cx].contiguous(), None, None, self.device, **kw),
                ops.pack_conv(w[:, cx:].contiguous(), b, None, self.device, **kw))

    def _conv(self, base: str, *inputs, **kw):
        """convolution `base` of the value encoder / mask decoder: fp16 operands under --amp, the hi/lo split under
        --f16_split, where eligible"""
        if kw.get('stride', 1) == 2 and self._split_of(base) and len(inputs) == 1 and self.convs[base].kh in (1, 3):
            # the split kernels take stride 1 only, and the scalar-gather fp32 kind a stride-2 layer would run instead
            # reaches ~50 TFLOP/s: the taps become channels (ops.gather_s2) and the layer a 1x1 convolution over them
            pc = self._s2_pack(base)
            kw = {k: v for k, v in kw.items() if k not in ('stride', 'pad')}
            return ops.conv2d(pc, ops.gather_s2(inputs[0], self.convs[base].kh), split=True, **kw)
        return ops.conv2d(self.convs[base], *inputs, amp=self._amp_of(base), split=self._split_of(base), **kw)

    def _folded(self, base: str):
        """(weight [cout][cin][kh][kw], bias | None) of convolution `base` with its BatchNorm folded"""
        w = self.sd[base + '.weight'].detach().float()
        b = self.sd.get(base + '.bias')
        b = None if b is None else b.detach().float()
        bn = self._bn_after(base)
        if bn is not None:
            gamma, beta, mean, var, eps = (t.detach().float() if torch.is_tensor(t) else t for t in bn)
            scale = gamma / torch.sqrt(var + eps)
            shift = beta - mean * scale
            w = w * scale.view(-1, 1, 1, 1)
            b = shift if b is None else b * scale + shift
        return w, b

    def _s2_pack(self, base: str) -> PackedConv:
        """the weights of stride-2 convolution `base` as a 1x1 convolution over its taps (channel index t*C + c)"""
        pc = self.s2.get(base)
        if pc is None:
            w, b = self._folded(base)
            cout, cin, kh, kw_ = w.shape
            w1 = w.permute(0, 2, 3, 1).reshape(cout, kh * kw_ * cin, 1, 1).contiguous()
            pc = self.s2[base] = ops.pack_conv(w1, b, None, self.device, split=True)
        return pc

    def _conv_shared_x(self, base: str, x, g, **kw):
        """conv over the virtual cat(x broadcast, g): one launch for a single object, otherwise the image
        part once + the per-object part with it as the fused residual"""
        amp, sp = self._amp_of(base), self._split_of(base)
        if g.shape[0] < 2 or x.shape[0] != 1 or base not in self.split:
            return ops.conv2d(self.convs[base], x, g, amp=amp, split=sp, **kw)
        wx, wg = self.split[base]
        act = kw.pop('act', 0)  # the activation belongs to the sum of the two parts
        shared = ops.conv2d(wx, x, amp=amp, split=sp, **kw)
        return ops.conv2d(wg, g, residual=shared, amp=amp, split=sp, act=act, **kw)

    def _bn_after(self, conv: str):
        """the BatchNorm that follows `conv` in the ResNets: convN -> bnN, downsample.0 -> downsample.1"""
        head, leaf = conv.rsplit('.', 1)
        if leaf.startswith('conv') and (head + '.bn' + leaf[4:] + '.running_mean') in self.sd:
            bn = head + '.bn' + leaf[4:]
        elif leaf == '0' and (head + '.1.running_mean') in self.sd:
            bn = head + '.1'
        else:
            return None
        g = self.sd
        return (g[bn + '.weight'], g[bn + '.bias'], g[bn + '.running_mean'], g[bn + '.running_var'], 1e-5)

    # ---------------------------------------------------------------- building blocks
    def _bottleneck(self, pre: str, x, stride: int):
        c = self.convs
        y = self._conv(pre + '.conv1', x, act=ACT_RELU)
        y = self._conv(pre + '.conv2', y, stride=stride, pad=1, act=ACT_RELU)
        if (pre + '.downsample.0') in c:
            x = self._conv(pre + '.downsample.0', x, stride=stride)
        return self._conv(pre + '.conv3', y, residual=x, act=ACT_RELU)

    def _basic(self, pre: str, x, stride: int):
        c = self.convs
        y = self._conv(pre + '.conv1', x, stride=stride, pad=1, act=ACT_RELU)
        if (pre + '.downsample.0') in c:
            x = self._conv(pre + '.downsample.0', x, stride=stride)
        return self._conv(pre + '.conv2', y, pad=1, residual=x, act=ACT_RELU)

    def _stage(self, pre: str, x, blocks: int, stride: int, block_fn):
        for i in range(blocks):
            x = block_fn(f'{pre}.{i}', x, stride if i == 0 else 1)
        return x

    def _res_block(self, pre: str, g0, g1=None):
        """relu -> 3x3 -> relu -> 3x3, plus (1x1-projected) input; input = virtual cat(g0, g1).  The inner ReLU is
        conv1's output stage (its result feeds conv2 only), so conv2 reads its input as it is."""
        c = self.convs
        if g1 is not None:
            t = self._conv_shared_x(pre + '.conv1', g0, g1, pad=1, relu_in=True, act=ACT_RELU)
            skip = self._conv_shared_x(pre + '.downsample', g0, g1)
        else:
            t = self._conv(pre + '.conv1', g0, pad=1, relu_in=True, act=ACT_RELU)
            skip = self._conv(pre + '.downsample', g0) if (pre + '.downsample') in c else g0
        return self._conv(pre + '.conv2', t, pad=1, residual=skip)

    def _fusion(self, pre: str, x, g):
        """x [1,Cx,h,w] image feature (broadcast over objects), g [no,Cg,h,w]"""
        g = self._res_block(pre + '.block1', x, g)
        a = pre + '.attention.ChannelGate.mlp.'
        g = ops.cbam(g, self.vecs[a + '1.weight'], self.vecs[a + '1.bias'], self.vecs[a + '3.weight'],
                     self.vecs[a + '3.bias'], self.convs[pre + '.attention.SpatialGate.spatial.conv'])
        return self._res_block(pre + '.block2', g)

    def _gru(self, conv: str, g, h):
        return ops.gru_update(self._conv(conv, g, h, pad=1), h)

    # ---------------------------------------------------------------- stages
    def encode_image(self, image):
        c = self.convs
        pe = 'pixel_encoder'
        if (pe + '.conv1') in self.stems:
            x = ops.stem7x7(self.stems[pe + '.conv1'], image, None, relu=True)
        else:
            x = self._conv(pe + '.conv1', image, stride=2, pad=3, act=ACT_RELU)
        x = ops.maxpool3x3s2(x)
        f4 = self._stage(pe + '.res2', x, 3, 1, self._bottleneck)
        f8 = self._stage(pe + '.layer2', f4, 4, 2, self._bottleneck)
        f16 = self._stage(pe + '.layer3', f8, 6, 2, self._bottleneck)
        both = self._conv(pe + '.proj12', f16)  # batch 1: the two channel ranges are contiguous tensors
        return (both[:, :self.proj_split], f8, f4), both[:, self.proj_split:]

    def transform_key(self, feat, need_s: bool, need_e: bool):
        c = self.convs
        shrinkage = self._conv('key_proj.d_proj', feat, pad=1, act=ACT_SQUARE_PLUS_ONE) if need_s else None
        selection = self._conv('key_proj.e_proj', feat, pad=1, act=ACT_SIGMOID) if need_e else None
        return self._conv('key_proj.key_proj', feat, pad=1), shrinkage, selection

    def encode_mask(self, image, f16, sensory, masks, deep_update: bool):
        """image [1,3,H,W]; masks [no,1,H,W]; sensory [no,C,h,w] -> value [no,C,h,w], sensory'"""
        c = self.convs
        me = 'mask_encoder'
        if (me + '.conv1') in self.stems:
            g = ops.stem7x7(self.stems[me + '.conv1'], image, masks)
        else:
            g = self._conv_shared_x(me + '.conv1', image, masks, stride=2, pad=3)
        g = ops.maxpool3x3s2(g, relu_after=True)
        g = self._stage(me + '.layer1', g, 2, 1, self._basic)
        g = self._stage(me + '.layer2', g, 2, 2, self._basic)
        g = self._stage(me + '.layer3', g, 2, 2, self._basic)
        value = self._fusion(me + '.fuser', f16, g)
        if deep_update:
            sensory = self._gru(me + '.sensory_update.transform', value, sensory)
        return value, sensory

    def decode_masks(self, ms_features, readout, sensory, last_mask16, want_p8_ds: bool = False):
        """readout/sensory [no,C,h,w]; last_mask16 [no,1,h,w] -> decoder pyramid p16, p8, p4, the object logits
        [no,1,4h,4w] (everything `segment` returns except the new sensory state) and, with want_p8_ds, p8 at 1/16 for
        `sensory_update` (None otherwise)"""
        c = self.convs
        md = 'mask_decoder'
        f16, f8, f4 = ms_features
        d8 = self._conv(md + '.decoder_feat_proc.transforms.0', f8)
        d4 = self._conv(md + '.decoder_feat_proc.transforms.1', f4)
        p16 = self._conv(md + '.sensory_compress', sensory, last_mask16, residual=readout)
        p16 = self._fusion(md + '.fuser', f16, p16)
        p8 = self._res_block(md + '.up_16_8.out_conv', ops.upsample2x_add(p16, d8))
        if want_p8_ds and p8.shape[-2] % 2 == 0 and p8.shape[-1] % 2 == 0:
            # the sensory update takes p8 at 1/16 (modules.py:121-151): written by the pass that up-samples p8 anyway
            up, p8_ds = ops.upsample2x_add_ds2(p8, d4)
        else:
            up, p8_ds = ops.upsample2x_add(p8, d4), None
        p4 = self._res_block(md + '.up_8_4.out_conv', up)
        logits = self._conv(md + '.pred', p4, pad=1, relu_in=True)
        return p16, p8, p4, logits, p8_ds

    def sensory_update(self, p16, p8, p4, logits, sensory, p8_ds=None):
        """the decoder's GRU update of the sensory memory (modules.py:121-151): needed by the NEXT frame only"""
        c = self.convs
        su = 'mask_decoder.sensory_update'
        g = self._conv(su + '.g16_conv', p16)
        if p8_ds is None:  # (decode_masks(want_p8_ds=True) makes it in its x2 up-sampling pass)
            p8_ds = ops.area_downsample(p8, 2)
        g = self._conv(su + '.g8_conv', p8_ds, residual=g)
        g = self._conv(su + '.g4_conv', ops.area_downsample(p4, 4), ops.area_downsample(logits, 4), residual=g)
        return self._gru(su + '.transform', g, sensory)

    def decode(self, ms_features, readout, sensory, last_mask16, update_sensory: bool):
        """readout/sensory [no,C,h,w]; last_mask16 [no,1,h,w] -> sensory', object logits [no,1,4h,4w]"""
        p16, p8, p4, logits, p8_ds = self.decode_masks(ms_features, readout, sensory, last_mask16, want_p8_ds=update_sensory)
        if update_sensory:
            sensory = self.sensory_update(p16, p8, p4, logits, sensory, p8_ds)
        return sensory, logits
