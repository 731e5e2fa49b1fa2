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

    # ------------------------------------------------------------------ shared frame plumbing
    def _begin_frame(self, image: torch.Tensor, image_ti_override):
        """advance the clock, pad the frame to a multiple of 16 and fetch (or compute) its features:
        -> (frame index used for the feature cache, 1*3*H'*W' image, ms_features, key, shrinkage, selection)"""
        self.curr_ti += 1
        frame_ti = image_ti_override if image_ti_override is not None else self.curr_ti
        padded, self.pad = pad_divide_by(image, 16)
        batch = padded.unsqueeze(0)
        store = self.image_feature_store
        return (frame_ti, batch, store.get_ms_features(frame_ti, batch), *store.get_key(frame_ti, batch))

    # ------------------------------------------------------------------ detections
    def incorporate_detection(self, image: torch.Tensor, new_mask: torch.Tensor,
                              segments_info: List[ObjectInfo], *, image_ti_override: bool = None,
                              forward_mask: torch.Tensor = None, incremental: bool = False) -> torch.Tensor:
        """merge an image-level detection into the propagated state (inference_core.py:137-198):
        propagate (unless the caller did), match detected segments with tracked objects by IoU, retire
        objects that went unseen for too long, and commit the merged masks as a memory frame"""
        from deva.inference.segment_merging import match_and_merge
        if self.memory._shard_group is not None and self.memory._shard_owner is not None:
            # frame-owner mode routes `step` only: non-owner ranks hold no encoder / decoder state to merge into
            raise NotImplementedError('incorporate_detection is not available in frame-owner mode '
                                      '(MemoryManager.shard_queries(owner=r)); use shard_queries() or shard_bank()')
        frame_ti, batch, ms_features, key, shrinkage, selection = self._begin_frame(image, image_ti_override)
        new_mask, _ = pad_divide_by(new_mask, 16)

        if forward_mask is None:
            # argmax over the propagated probabilities in one pass (the output-tail kernel without resize / LUT)
            forward_mask = (ops.index_mask(self._segment(key, selection, ms_features).contiguous())
                            if self.memory.engaged else torch.zeros_like(new_mask))

        merged = match_and_merge(forward_mask, new_mask, self.object_manager, segments_info,
                                 max_num_objects=self.max_num_objects, incremental_mode=incremental)
        anything_purged, tmp_kept, obj_kept = self.object_manager.purge_inactive_objects(
            self.max_missed_detection_count)
        if anything_purged:
            self.memory.purge_except(obj_kept)
            merged = merged[[t - 1 for t in tmp_kept]]  # tmp ids are 1-based channel numbers

        self.last_mask = merged.unsqueeze(0).type_as(key)
        self._add_memory(batch, ms_features, self.last_mask, key, shrinkage, selection)
        self.image_feature_store.delete(frame_ti)
        return unpad(self.network.aggregate(self.last_mask[0], dim=0), self.pad)

    # ------------------------------------------------------------------ propagation
    def _blend_annotation(self, prediction: torch.Tensor, mask: torch.Tensor, objects: List[int],
                          new_tmp_ids: List[int], hard_mask: bool) -> torch.Tensor:
        """an annotation that covers only some objects on top of the propagated prediction
        ((no+1)*H*W): annotated pixels win, channels of newly introduced objects are appended
        (inference_core.py:251-272, including its channel indexing).  For soft masks the reference
        compares the (values, indices) pair of `mask.max(0)` with 0.5 and indexes the annotation by tmp
        id (a TypeError on every call); here the values are compared and channel `pos` of the annotation
        is taken, which is what that code means."""
        fg = prediction[1:]
        annotated = (mask > 0) if hard_mask else (mask.max(0).values > 0.5)
        fg[:, annotated] = 0
        appended = []
        for pos, tmp_id in enumerate(new_tmp_ids):
            channel = (mask == objects[pos]).type_as(fg) if hard_mask else mask[pos].type_as(fg)
            if tmp_id < fg.shape[0]:
                fg[tmp_id + 1] = channel
            else:
                appended.append(channel.unsqueeze(0))
        return torch.cat([fg, *appended], dim=0)

    def _step_frame_owner(self, image, mask, objects, hard_mask, end, image_ti_override, delete_buffer):
        """`step` of ONE clip on several GPUs in frame-owner mode (`MemoryManager.shard_queries(group,
        owner=r)`): every rank of the group calls `step` with the same arguments; only the
        owner runs the key encoder, the mask decoder and the value encoder.  Per frame the owner broadcasts
        the query key / selection, every rank matches and reads out its share of the query columns against
        its replica of the bank, the read-out columns are gathered to the owner and the integer usage
        counters all-reduced; on memory frames the owner broadcasts the new key / shrinkage / selection /
        value rows and every rank appends them (consolidation and eviction then run redundantly on
        identical inputs with deterministic kernels, so the replicas cannot diverge).  The frame state
        machine depends on host-side state only, which is identical on all ranks.  Returns the
        probabilities on the owner, None elsewhere."""
        mem, om = self.memory, self.object_manager
        own = mem.is_frame_owner
        annotated = mask is not None
        self.curr_ti += 1
        frame_ti = image_ti_override if image_ti_override is not None else self.curr_ti
        padded, self.pad = pad_divide_by(image, 16)
        h, w = padded.shape[-2] // 16, padded.shape[-1] // 16
        device = image.device
        ms_features = key = shrinkage = selection = None
        if own:
            batch = padded.unsqueeze(0)
            store = self.image_feature_store
            ms_features = store.get_ms_features(frame_ti, batch)
            key, shrinkage, selection = store.get_key(frame_ti, batch)
        due = self.curr_ti - self.last_mem_ti >= self.mem_every
        commit = (annotated or due) and not end
        propagate = (not annotated) or (om.num_obj > 0 and not om.has_all(objects))

        prob = None
        if propagate:
            if not mem.engaged:
                warnings.warn('Trying to segment without any memory!', RuntimeWarning)
                if own:
                    prob = torch.zeros((1, h * 16, w * 16), device=device, dtype=torch.float32)
            else:
                qk, qe = mem.broadcast_query(key, selection, h, w, device)
                readout = mem.match_memory(qk, qe)
                if own:
                    ids = om.all_obj_ids
                    sensory, _, dec = self.network.segment(ms_features, om.realize_dict(readout).unsqueeze(0),
                                                           mem.get_sensory(ids), self.last_mask,
                                                           chunk_size=self.chunk_size, update_sensory=not end)
                    if not end:
                        mem.update_sensory(sensory, ids)
                    prob = dec[0]
        if annotated:
            new_tmp_ids, _ = om.add_new_objects(objects)
            if own:
                mask, _ = pad_divide_by(mask, 16)
                if propagate:
                    mask = self._blend_annotation(prob, mask, objects, new_tmp_ids, hard_mask)
                elif hard_mask:
                    mask = torch.stack([mask == o for o in objects], dim=0)
                prob = ops.softmax_channels(self.network.aggregate(mask, dim=0))
        if own:
            self.last_mask = prob[1:].unsqueeze(0)
        if commit:
            ids = om.all_obj_ids
            if not ids:
                warnings.warn('Empty object mask!', RuntimeWarning)
            else:
                value = sensory = None
                if own:
                    mem.initialize_sensory_if_needed(key, ids)
                    value, sensory = self.network.encode_mask(batch, ms_features, mem.get_sensory(ids), self.last_mask,
                                                              is_deep_update=True, chunk_size=self.chunk_size)
                key_b, shr_b, val_b, sel_b = mem.broadcast_memory_frame(key, shrinkage, value, selection, ids, h, w,
                                                                        device)
                mem.add_memory(key_b, shr_b, val_b, ids, selection=sel_b)
                self.last_mem_ti = self.curr_ti
                if own:
                    mem.update_sensory(sensory, ids)
        if own and delete_buffer:
            self.image_feature_store.delete(frame_ti)
        return unpad(prob, self.pad) if own else None

    def step(self, image: torch.Tensor, mask: torch.Tensor = None, objects: Optional[List[int]] = None, *,
             hard_mask: bool = True, end: bool = False, image_ti_override: bool = None,
             delete_buffer: bool = True) -> torch.Tensor:
        """One frame (inference_core.py:200-290).

        image: 3*H*W, ImageNet-normalised.  mask: H*W index mask, or len(objects)*H*W soft masks with
        hard_mask=False, or None to propagate only.  objects: ids in mask order (None with soft masks
        means 1..mask.shape[0]).  end: last frame of the sequence -- nothing is written to the memories.
        Returns (num_objects+1)*H*W probabilities at the input size, channel 0 = background."""
        annotated = mask is not None
        if annotated and objects is None:
            assert not hard_mask
            objects = list(range(1, mask.shape[0] + 1))
        if self.memory._shard_group is not None and self.memory._shard_owner is not None:
            return self._step_frame_owner(image, mask, objects, hard_mask, end, image_ti_override, delete_buffer)

        frame_ti, batch, ms_features, key, shrinkage, selection = self._begin_frame(image, image_ti_override)
        due = self.curr_ti - self.last_mem_ti >= self.mem_every
        commit = (annotated or due) and not end
        # propagate unless the annotation covers every object known so far
        om = self.object_manager
        propagate = (not annotated) or (om.num_obj > 0 and not om.has_all(objects))

        prob = self._segment(key, selection, ms_features, update_sensory=not end) if propagate else None
        if annotated:
            new_tmp_ids, _ = om.add_new_objects(objects)
            mask, _ = pad_divide_by(mask, 16)
            if propagate:
                mask = self._blend_annotation(prob, mask, objects, new_tmp_ids, hard_mask)
            elif hard_mask:
                mask = torch.stack([mask == o for o in objects], dim=0)  # index mask -> one-hot
            prob = ops.softmax_channels(self.network.aggregate(mask, dim=0))

        self.last_mask = prob[1:].unsqueeze(0)
        if commit:
            self._add_memory(batch, ms_features, self.last_mask, key, shrinkage, selection)
        if delete_buffer:
            self.image_feature_store.delete(frame_ti)
        return unpad(prob, self.pad)
