"""`DEVAInferenceCore`: the per-frame state machine of the reference
(deva/inference/inference_core.py:17-290) on the HIP network / memory modules of this package.

Public surface kept verbatim: constructor, attributes (`network, mem_every, enable_long_term,
chunk_size, max_missed_detection_count, max_num_objects, config, curr_ti, last_mem_ti, memory,
object_manager, image_feature_store, last_mask, frame_buffer, pad`) and the methods `step`,
`incorporate_detection`, `add_to_temporary_buffer`, `vote_in_temporary_buffer`, `clear_buffer`,
`enabled_long_id`, `_segment`, `_add_memory`.
"""
import warnings
from typing import Dict, Iterable, List, Literal, Optional, Tuple

import torch

from deva.hip import ops
from deva.inference.image_feature_store import ImageFeatureStore
from deva.inference.memory_manager import MemoryManager
from deva.inference.object_info import ObjectInfo
from deva.inference.object_manager import ObjectManager
from deva.model.network import DEVA
from deva.utils.tensor_utils import pad_divide_by, unpad


class DEVAInferenceCore:
    def __init__(self, network: DEVA, config: Dict, *, image_feature_store: ImageFeatureStore = None):
        self.network = network
        self.mem_every = config['mem_every']
        self.enable_long_term = config['enable_long_term']
        self.chunk_size = config['chunk_size']
        self.max_missed_detection_count = config.get('max_missed_detection_count')
        self.max_num_objects = config.get('max_num_objects')
        self.config = config

        self.curr_ti = -1
        self.last_mem_ti = 0
        self.memory = MemoryManager(config=config)
        self.object_manager = ObjectManager()
        self.image_feature_store = (ImageFeatureStore(self.network)
                                    if image_feature_store is None else image_feature_store)
        self.last_mask = None
        self.pad = None
        self.frame_buffer = []  # online / semi-online processing

    def enabled_long_id(self) -> None:
        # short ids 1..255 (palette PNG) by default; long ids 256..255**3 for panoptic RGB masks
        self.object_manager.use_long_id = True

    @property
    def use_long_id(self):
        return self.object_manager.use_long_id

    # ------------------------------------------------------------------ the two halves of a frame
    def _add_memory(self, image: torch.Tensor, ms_features: Iterable[torch.Tensor], prob: torch.Tensor,
                    key: torch.Tensor, shrinkage: torch.Tensor, selection: torch.Tensor, *,
                    is_deep_update: bool = True) -> None:
        """encode (image, masks) into a memory value and append it (inference_core.py:55-87).
        image 1*3*H*W; prob 1*num_objects*H*W in [0,1]"""
        if prob.shape[1] == 0:
            warnings.warn('Empty object mask!', RuntimeWarning)
            return
        ids = self.object_manager.all_obj_ids
        self.memory.initialize_sensory_if_needed(key, ids)
        value, sensory = self.network.encode_mask(image, ms_features, self.memory.get_sensory(ids), prob,
                                                  is_deep_update=is_deep_update,
                                                  chunk_size=self.chunk_size)
        self.memory.add_memory(key, shrinkage, value, ids, selection=selection)
        self.last_mem_ti = self.curr_ti
        if is_deep_update:
            self.memory.update_sensory(sensory, ids)

    def _segment(self, key: torch.Tensor, selection: torch.Tensor, ms_features: Iterable[torch.Tensor],
                 update_sensory: bool = True) -> torch.Tensor:
        """memory read + decode for every live object (inference_core.py:89-113);
        returns (num_objects+1)*H*W probabilities"""
        if not self.memory.engaged:
            warnings.warn('Trying to segment without any memory!', RuntimeWarning)
            return torch.zeros((1, key.shape[-2] * 16, key.shape[-1] * 16), device=key.device, dtype=key.dtype)
        ids = self.object_manager.all_obj_ids
        readout = self.memory.match_memory(key, selection)
        readout = self.object_manager.realize_dict(readout).unsqueeze(0)
        sensory, _, prob = self.network.segment(ms_features, readout, self.memory.get_sensory(ids),
                                                self.last_mask, chunk_size=self.chunk_size,
                                                update_sensory=update_sensory)
        if update_sensory:
            self.memory.update_sensory(sensory, ids)
        return prob[0]

    # ------------------------------------------------------------------ semi-online buffer
    def add_to_temporary_buffer(self, frame_info) -> None:
        self.frame_buffer.append(frame_info)

    def vote_in_temporary_buffer(
            self, keyframe_selection: Literal['last', 'middle', 'score', 'first'] = 'first'
    ) -> Tuple[int, torch.Tensor, List[ObjectInfo]]:
        # consensus voting (deva/inference/consensus_automatic.py:82) is a caller of this path, not
        # part of it; it is resolved from whichever `deva` tree provides it.
        from deva.inference.consensus_automatic import find_consensus_auto_association
        return find_consensus_auto_association(self.frame_buffer, network=self.network,
                                               store=self.image_feature_store, config=self.config,
                                               keyframe_selection=keyframe_selection)

    def clear_buffer(self) -> None:
        for f in self.frame_buffer:
            self.image_feature_store.delete(f.ti)
        self.frame_buffer = []

    # ------------------------------------------------------------------ detections
    def incorporate_detection(self, image: torch.Tensor, new_mask: torch.Tensor,
                              segments_info: List[ObjectInfo], *, image_ti_override: bool = None,
                              forward_mask: torch.Tensor = None, incremental: bool = False) -> torch.Tensor:
        """merge an image-level detection into the propagated state (inference_core.py:137-198)"""
        from deva.inference.segment_merging import match_and_merge
        self.curr_ti += 1
        image_ti = self.curr_ti if image_ti_override is None else image_ti_override

        image, self.pad = pad_divide_by(image, 16)
        new_mask, _ = pad_divide_by(new_mask, 16)
        image = image.unsqueeze(0)
        ms_features = self.image_feature_store.get_ms_features(image_ti, image)
        key, shrinkage, selection = self.image_feature_store.get_key(image_ti, image)

        if forward_mask is None:
            if self.memory.engaged:
                forward_mask = torch.argmax(self._segment(key, selection, ms_features), dim=0)
            else:
                forward_mask = torch.zeros_like(new_mask)

        merged_mask = match_and_merge(forward_mask, new_mask, self.object_manager, segments_info,
                                      max_num_objects=self.max_num_objects, incremental_mode=incremental)
        purged, tmp_keep_idx, obj_keep_idx = self.object_manager.purge_inactive_objects(
            self.max_missed_detection_count)
        if purged:
            self.memory.purge_except(obj_keep_idx)
            merged_mask = merged_mask[[i - 1 for i in tmp_keep_idx]]

        self.last_mask = merged_mask.unsqueeze(0).type_as(key)
        self._add_memory(image, ms_features, self.last_mask, key, shrinkage, selection)
        pred_prob_with_bg = self.network.aggregate(self.last_mask[0], dim=0)
        self.image_feature_store.delete(image_ti)
        return unpad(pred_prob_with_bg, self.pad)

    # ------------------------------------------------------------------ propagation
    def step(self, image: torch.Tensor, mask: torch.Tensor = None, objects: Optional[List[int]] = None, *,
             hard_mask: bool = True, end: bool = False, image_ti_override: bool = None,
             delete_buffer: bool = True) -> torch.Tensor:
        """
        image: 3*H*W (ImageNet-normalised)
        mask: H*W index mask, or len(objects)*H*W soft masks (hard_mask=False), or None
        objects: object ids in mask order; None (soft masks only) means 1..mask.shape[0]
        end: last frame of the sequence -- skip memory/sensory updates
        returns (num_objects+1)*H*W probabilities, channel 0 = background  (inference_core.py:200-290)
        """
        if objects is None and mask is not None:
            assert not hard_mask
            objects = list(range(1, mask.shape[0] + 1))

        self.curr_ti += 1
        image_ti = self.curr_ti if image_ti_override is None else image_ti_override

        image, self.pad = pad_divide_by(image, 16)
        image = image.unsqueeze(0)

        is_mem_frame = ((self.curr_ti - self.last_mem_ti >= self.mem_every) or (mask is not None)) and (not end)
        # segment when no mask is given, or when the given mask does not cover every known object
        need_segment = (mask is None) or (not self.object_manager.has_all(objects)
                                          and self.object_manager.num_obj > 0)

        ms_features = self.image_feature_store.get_ms_features(image_ti, image)
        key, shrinkage, selection = self.image_feature_store.get_key(image_ti, image)

        if need_segment:
            pred_prob_with_bg = self._segment(key, selection, ms_features, update_sensory=not end)

        if mask is not None:
            new_tmp_ids, _ = self.object_manager.add_new_objects(objects)
            mask, _ = pad_divide_by(mask, 16)
            if need_segment:
                # merge the prediction with the (partial) input mask; input pixels win
                fg = pred_prob_with_bg[1:]
                if hard_mask:
                    fg[:, mask > 0] = 0
                else:
                    fg[:, mask.max(0) > 0.5] = 0
                extra = []
                for mask_id, tmp_id in enumerate(new_tmp_ids):
                    this_mask = (mask == objects[mask_id]).type_as(fg) if hard_mask else mask[tmp_id]
                    if tmp_id >= fg.shape[0]:
                        extra.append(this_mask.unsqueeze(0))
                    else:
                        fg[tmp_id + 1] = this_mask  # reference indexing (inference_core.py:268-270)
                mask = torch.cat([fg, *extra], dim=0)
            elif hard_mask:
                mask = torch.stack([mask == obj for obj in objects], dim=0)  # index mask -> one-hot
            pred_prob_with_bg = ops.softmax_channels(self.network.aggregate(mask, dim=0))

        self.last_mask = pred_prob_with_bg[1:].unsqueeze(0)

        if is_mem_frame:
            self._add_memory(image, ms_features, self.last_mask, key, shrinkage, selection)

        if delete_buffer:
            self.image_feature_store.delete(image_ti)

        return unpad(pred_prob_with_bg, self.pad)
