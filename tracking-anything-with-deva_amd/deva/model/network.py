"""`DEVA`: the network-module interface of the reference (deva/model/network.py:18-190) executed by
hand-written gfx950 kernels (libdeva_hip) instead of ATen.

Same constructor, attributes (`pix_feat_dim`, `key_dim`, `value_dim`, `pixel_encoder`,
`mask_encoder`, `key_proj`, `mask_decoder`), methods and tensor conventions; the state_dict is
key- and shape-compatible with `DEVA-propagation.pth`, so `load_weights(torch.load(path))` works
unchanged.  Inference only: parameters are frozen and eval-mode BatchNorm is folded at load time.
"""
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from deva.hip import ops
from deva.model._graph import CompiledGraph, build_parameter_tree


class DEVA(nn.Module):
    def __init__(self, config: Dict):
        super().__init__()
        self.pix_feat_dim = config['pix_feat_dim']
        self.key_dim = config['key_dim']
        self.value_dim = config['value_dim']
        # --amp (eval_args.py:17; the reference's drivers wrap the frame loop in fp16 autocast, eval_vos.py:137): the
        # value encoder and the mask decoder run their convolutions on fp16 operands with fp32 accumulation; key encoder,
        # key projection, memory read, aggregate and the mask-logit head stay fp32 (deva/model/_graph.py:AMP_SCOPES)
        self.amp = bool(config.get('amp', False))
        # --f16_split (an extension, no counterpart in the reference): the same two modules run fp32-ACCURATE
        # convolutions on the f16 matrix pipes (hi/lo fp16 split of both operands, three MFMAs per block, fp32
        # accumulation; csrc/conv_f16.hip) -- held to the fp32 parity gates, 2.5-3x the fp32-MFMA rate
        self.f16_split = bool(config.get('f16_split', False))
        if self.amp and self.f16_split:
            raise ValueError('--amp and --f16_split are alternatives: pick one')
        # --f16_split_key_encoder: the key encoder on the split kernels as well (deva/model/_graph.py:split_scopes)
        # fp32 path: Winograd F(2x2, 3x3) for the big 3x3 layers of the value encoder / mask decoder (csrc/conv_wino.hip); on by
        # default, --no_winograd keeps every layer on the direct kernels (the arithmetic order of rounds 1-5)
        self.winograd = not bool(config.get('no_winograd', False))
        self.f16_split_key_encoder = bool(config.get('f16_split_key_encoder', False))
        if self.f16_split_key_encoder and not self.f16_split:
            raise ValueError('--f16_split_key_encoder extends --f16_split: pass both')
        for name, module in build_parameter_tree(self.pix_feat_dim, self.key_dim, self.value_dim).items():
            self.add_module(name, module)
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()
        self._graph: Optional[CompiledGraph] = None

    # ------------------------------------------------------------------ weights / placement
    def _apply(self, fn, *args, **kwargs):
        self._graph = None  # .cuda()/.to()/.float() move the parameters: re-pack lazily
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._graph = None
        return super().load_state_dict(*args, **kwargs)

    def load_weights(self, src_dict) -> None:
        # network.py:189-190 (strict load)
        self.load_state_dict(src_dict)

    def graph(self) -> CompiledGraph:
        if self._graph is None:
            device = next(self.parameters()).device
            self._graph = CompiledGraph(self.state_dict(), device, amp=self.amp, split=self.f16_split,
                                        split_key_encoder=self.f16_split_key_encoder, winograd=self.winograd)
        return self._graph

    # ------------------------------------------------------------------ reference API
    def aggregate(self, prob: torch.Tensor, dim: int) -> torch.Tensor:
        """network.py:33-40.  prob: object probabilities with the object axis at `dim`."""
        moved = prob.movedim(dim, 0).contiguous()
        if moved.dtype not in (torch.float32, torch.bool, torch.uint8):
            moved = moved.float()
        return ops.aggregate(moved).movedim(0, dim)

    def encode_image(self, image: torch.Tensor) -> Tuple[Tuple[torch.Tensor, ...], torch.Tensor]:
        """network.py:42-44.  image [1,3,H,W] -> (f16, f8, f4), key feature"""
        return self.graph().encode_image(_f32c(image))

    def transform_key(self, feat: torch.Tensor, *, need_sk: bool = True, need_ek: bool = True):
        """network.py:62-68 -> key [B,CK,h,w], shrinkage [B,1,h,w] | None, selection | None"""
        return self.graph().transform_key(_f32c(feat), need_sk, need_ek)

    def encode_mask(self, image: torch.Tensor, ms_features: Iterable[torch.Tensor], h: torch.Tensor,
                    masks: torch.Tensor, *, is_deep_update: bool = True,
                    chunk_size: int = -1) -> Tuple[torch.Tensor, torch.Tensor]:
        """network.py:46-60.  image [1,3,H,W]; h (sensory) [1,no,C,h,w]; masks [1,no,H,W]
        -> value [1,no,C,h,w], new sensory [1,no,C,h,w]"""
        assert image.shape[0] == 1 and masks.shape[0] == 1, 'batch size 1 (objects are the batch axis)'
        g = self.graph()
        image, f16 = _f32c(image), ms_features[0]
        sens_all = _f32c(h)[0]
        masks_all = _f32c(masks)[0].unsqueeze(1)
        no = masks_all.shape[0]
        step = no if (chunk_size < 1 or chunk_size >= no) else chunk_size
        values, sens = [], []
        for i in range(0, no, step):
            v, s = g.encode_mask(image, f16, sens_all[i:i + step], masks_all[i:i + step], is_deep_update)
            values.append(v)
            sens.append(s)
        value = values[0] if len(values) == 1 else torch.cat(values, 0)
        new_h = sens[0] if len(sens) == 1 else torch.cat(sens, 0)
        return value.unsqueeze(0), new_h.unsqueeze(0)

    def read_memory(self, query_key: torch.Tensor, query_selection: torch.Tensor, memory_key: torch.Tensor,
                    memory_shrinkage: torch.Tensor, memory_value: torch.Tensor) -> torch.Tensor:
        """network.py:72-92, the dense (full-softmax) read used at training time; inference reads
        through `MemoryManager.match_memory`.  query_key / query_selection [B,CK,H,W]; memory_key
        [B,CK,T,H,W]; memory_shrinkage [B,1,T,H,W]; memory_value [B,no,CV,T,H,W] -> [B,no,CV,H,W].
        Per batch element: dense similarity of all memory tokens against the HW queries, column
        softmax with max subtraction, read-out as an fp32-MFMA GEMM (the consolidation kernels)."""
        B, no, cv = memory_value.shape[:3]
        ck, (h, w) = query_key.shape[1], query_key.shape[-2:]
        hw = h * w
        n = memory_key[0, 0].numel()
        dev = query_key.device
        out = torch.empty((B, no, cv, h, w), dtype=torch.float32, device=dev)
        queries = torch.arange(n, n + hw, dtype=torch.int32, device=dev)
        for b in range(B):
            # token-major rows: memory tokens first, the queries after them (the kernel picks its
            # queries by row index; the selection of the memory rows is never read)
            key_rows = torch.empty((n + hw, ck), dtype=torch.float32, device=dev)
            sel_rows = torch.zeros((n + hw, ck), dtype=torch.float32, device=dev)
            ops.bank_append(_f32c(memory_key[b]).reshape(ck, n), key_rows, 0)
            ops.bank_append(_f32c(query_key[b]).reshape(ck, hw), key_rows, n)
            ops.bank_append(_f32c(query_selection[b]).reshape(ck, hw), sel_rows, n)
            shr = _f32c(memory_shrinkage[b]).reshape(n)
            aff = ops.softmax_columns(ops.similarity_dense(key_rows, shr, sel_rows, queries, n), hw)
            ld = aff.shape[1]
            cv_pad = (cv + 31) // 32 * 32
            val_rows = torch.zeros((n, cv_pad), dtype=torch.float32, device=dev)
            for o in range(no):
                # out[c][q] = sum_n value[n][c] * aff[n][q]: the values are the GEMM's [K][cout] weights, the
                # affinity matrix its [K = tokens][pixels = queries] input
                rows = val_rows if cv_pad == cv else torch.empty((n, cv), dtype=torch.float32, device=dev)
                ops.bank_append(_f32c(memory_value[b, o]).reshape(cv, n), rows, 0)
                if cv_pad != cv:
                    val_rows[:, :cv] = rows
                gemm = ops.PackedConv(val_rows, None, n, cv, cv_pad, 1, 1)
                r = ops.conv2d(gemm, aff.view(1, n, 1, ld))
                out[b, o] = r.view(cv, ld)[:, :hw].reshape(cv, h, w)
        return out

    def segment(self, multi_scale_features: Iterable[torch.Tensor], memory_readout: torch.Tensor,
                sensory: torch.Tensor, last_mask: torch.Tensor, *, selector=None, need_aux: bool = False,
                chunk_size: int = -1, update_sensory: bool = True, independent_objects: bool = False):
        """network.py:94-173 (inference form).  memory_readout/sensory [1,no,C,h,w]; last_mask
        [1,no,H,W] -> sensory' [1,no,C,h,w], logits [1,no+1,H,W], prob [1,no+1,H,W]"""
        if need_aux or selector is not None or independent_objects:
            raise NotImplementedError('training-only options of DEVA.segment are not part of the '
                                      'inference path (need_aux / selector / independent_objects)')
        assert memory_readout.shape[0] == 1, 'batch size 1 (objects are the batch axis)'
        g = self.graph()
        ms = tuple(multi_scale_features)
        readout_all, sens_all = _f32c(memory_readout)[0], _f32c(sensory)[0]
        no = readout_all.shape[0]
        last16 = ops.area_downsample(_f32c(last_mask)[0], last_mask.shape[-1] // readout_all.shape[-1])
        last16 = last16.unsqueeze(1)
        step = no if (chunk_size < 1 or chunk_size >= no) else chunk_size
        logits, sens = [], []
        for i in range(0, no, step):
            s, lg = g.decode(ms, readout_all[i:i + step], sens_all[i:i + step], last16[i:i + step],
                             update_sensory)
            logits.append(lg)
            sens.append(s)
        obj_logits = (logits[0] if len(logits) == 1 else torch.cat(logits, 0))[:, 0]
        new_sens = sens[0] if len(sens) == 1 else torch.cat(sens, 0)
        agg = ops.aggregate(obj_logits, apply_sigmoid=True)
        logits_up, prob = ops.upsample4x_softmax(agg)
        return new_sens.unsqueeze(0), logits_up.unsqueeze(0), prob.unsqueeze(0)

    def forward(self, mode: str, *args, **kwargs):
        # network.py:175-187
        if mode == 'encode_image':
            return self.encode_image(*args, **kwargs)
        elif mode == 'transform_key':
            return self.transform_key(*args, **kwargs)
        elif mode == 'encode_mask':
            return self.encode_mask(*args, **kwargs)
        elif mode == 'read_memory':
            return self.read_memory(*args, **kwargs)
        elif mode == 'segment':
            return self.segment(*args, **kwargs)
        raise NotImplementedError(mode)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
