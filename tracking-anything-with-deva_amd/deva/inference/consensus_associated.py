"""`spatial_alignment` on the fused gfx950 kernels (deva/inference/consensus_associated.py:16-69):
the one-frame memory read that projects the segmentation of one frame of the voting window onto
another (semi-online consensus).  The reference materialises the HW x HW
similarity and affinity matrices for it (266 MB each at 1080p); here it is one fused
similarity -> top-k -> softmax pass plus a sparse read-out per object, exactly like a frame's
memory read with N = HW.

Everything else of this module (`find_consensus_with_established_association`, keyframe scoring)
is host logic and stays the reference's: when a reference checkout follows this package on
`sys.path` its module of the same name is loaded, its public names are re-exported
from here, and its `spatial_alignment` is rebound to this one so that its own callers use it too.
"""
import importlib.util
import os
from typing import Dict

import torch

from deva.hip import ops


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


def _adopt_reference_module() -> None:
    """re-export the reference's host-side functions of this module, if a checkout is on the path"""
    import deva.inference as pkg
    here = os.path.dirname(os.path.abspath(__file__))
    for d in list(pkg.__path__):
        cand = os.path.join(d, 'consensus_associated.py')
        if os.path.abspath(d) == here or not os.path.isfile(cand):
            continue
        spec = importlib.util.spec_from_file_location('deva.inference._reference_consensus_associated', cand)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        ref.spatial_alignment = spatial_alignment
        for name, obj in vars(ref).items():
            if not name.startswith('__') and name not in globals():
                globals()[name] = obj
        globals()['_keyframe_objective_from_mask'] = ref._keyframe_objective_from_mask
        return


_adopt_reference_module()
