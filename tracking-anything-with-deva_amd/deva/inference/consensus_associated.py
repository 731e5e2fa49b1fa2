"""`spatial_alignment` on the fused gfx950 kernels (deva/inference/consensus_associated.py:16-69):
the one-frame memory read that projects the segmentation of one frame of the voting window onto
another (semi-online consensus).  The reference materialises the HW x HW
similarity and affinity matrices for it (266 MB each at 1080p); here it is one fused
similarity -> top-k -> softmax pass plus a sparse read-out per object, exactly like a frame's
memory read with N = HW.

`find_consensus_with_established_association` (consensus_associated.py:80-160; callers: the referring /
saliency evaluation drivers) is restated on top of it: keyframe choice with one host transfer for the whole
window, score-weighted sum of the projections.
"""
from typing import Dict, List, Optional, Tuple

import torch

from deva.hip import ops
from deva.utils.tensor_utils import pad_divide_by, unpad


def spatial_alignment(src_ti: int, src_image: torch.Tensor, src_mask: torch.Tensor, tar_ti: int,
                      tar_image: torch.Tensor, network, store, config: Dict) -> torch.Tensor:
    """src_image / tar_image: 3*H*W, src_mask: num_objects*H*W  ->  target segmentation
    num_objects*H*W (probabilities, background channel dropped like the reference's `segment`)."""
    num_objects, h, w = src_mask.shape
    src_image = src_image.unsqueeze(0)
    tar_image = tar_image.unsqueeze(0)
    src_mask = src_mask.unsqueeze(0)

    src_ms_features = store.get_ms_features(src_ti, src_image)
    src_key, src_shrinkage, _ = store.get_key(src_ti, src_image)
    tar_ms_features = store.get_ms_features(tar_ti, tar_image)
    tar_key, _, tar_selection = store.get_key(tar_ti, tar_image)

    # memory of the source frame
    h16, w16 = h // 16, w // 16
    hw = h16 * w16
    cv = config['value_dim']
    sensory = torch.zeros((1, num_objects, cv, h16, w16), device=src_key.device)
    value, sensory = network.encode_mask(src_image, src_ms_features, sensory, src_mask,
                                         is_deep_update=True, chunk_size=config['chunk_size'])

    # one-frame bank, token-major like the memory stores
    ck = src_key.shape[1]
    key_rows = torch.empty((hw, ck), dtype=torch.float32, device=src_key.device)
    ops.bank_append(src_key[0].reshape(ck, hw), key_rows, 0)
    shr = src_shrinkage[0].reshape(hw).contiguous()
    idx, weight = ops.affinity_topk(None, None, 0, key_rows, shr, hw, tar_key[0].reshape(ck, hw),
                                    tar_selection[0].reshape(ck, hw), config['top_k'])
    readout = torch.empty((1, num_objects, cv, h16, w16), dtype=torch.float32, device=src_key.device)
    val_rows = torch.empty((hw, cv), dtype=torch.float32, device=src_key.device)
    for o in range(num_objects):
        ops.bank_append(value[0, o].reshape(cv, hw), val_rows, 0)
        ops.readout_sparse(idx, weight, None, 0, val_rows, readout[0, o].view(cv, hw))

    _, _, tar_mask = network.segment(tar_ms_features, readout, sensory, src_mask,
                                     chunk_size=config['chunk_size'], update_sensory=False)
    return tar_mask


def _keyframe_objective_from_mask(mask: torch.Tensor, score, method: str = 'high_foreground'):
    """how good a keyframe a segmentation would make (consensus_associated.py:70-77)"""
    if method == 'high_foreground':
        return (mask > 0.8).float().mean()
    if method == 'score':
        return score
    raise NotImplementedError


def find_consensus_with_established_association(time_indices: List[int], images: List[torch.Tensor],
                                                masks: List[torch.Tensor], network, store, config: Dict,
                                                scores: Optional[List[float]] = None) -> Tuple[int, torch.Tensor]:
    """Consensus of a window whose segments are already associated across frames (channel c is the same object
    in every frame; consensus_associated.py:80-160): pick the keyframe (highest score, or largest confident
    foreground), project every other frame onto it with the fused `spatial_alignment` and return the
    score-weighted sum of the projections.  `images` / `masks` are padded in place like the reference does."""
    pads = (0, 0, 0, 0)
    for i in range(len(images)):
        images[i], pads = pad_divide_by(images[i], 16)
        masks[i], _ = pad_divide_by(masks[i], 16)
    ranked_by_score = scores is not None
    weights = torch.softmax(torch.tensor([1.0] * len(time_indices) if scores is None else list(scores),
                                         dtype=torch.float32) * 2, dim=0).tolist()
    if ranked_by_score:
        objectives = weights
    else:  # one reduction per frame, ONE host transfer for the whole window
        objectives = torch.stack([_keyframe_objective_from_mask(m, None) for m in masks]).tolist()
    key = max(range(len(time_indices)), key=lambda i: (objectives[i], -i))  # first maximum, like the reference's '>'
    key_weight = weights[key] if ranked_by_score else weights[0]
    total = masks[key] * key_weight
    for i, ti in enumerate(time_indices):
        if ti == time_indices[key]:
            continue
        projected = spatial_alignment(ti, images[i], masks[i], time_indices[key], images[key], network, store, config)
        total += projected[0, 1:] * weights[i]
    return time_indices[key], unpad(total, pads)
