"""Object id bookkeeping (interface of deva/inference/object_manager.py:8-168).

Real object ids never change; temporary ids are the 1-based channel positions of the objects in
the network tensors and are re-packed when objects are removed."""
from typing import Dict, List, Set, Tuple, Union

import numpy as np
import torch

from deva.inference.object_info import ObjectInfo


class ObjectManager:
    def __init__(self):
        self.obj_to_tmp_id: Dict[ObjectInfo, int] = {}
        self.tmp_id_to_obj: Dict[int, ObjectInfo] = {}
        self.obj_id_to_obj: Dict[int, ObjectInfo] = {}
        self.all_historical_object_ids: Set[int] = set()  # removed ids stay reserved
        self.use_long_id = False

    def _reindex(self) -> None:
        self.obj_id_to_obj = {obj.id: obj for obj in self.obj_to_tmp_id}

    def _fresh_id(self, wanted: int) -> int:
        """keep `wanted` unless it collides (or is a short id in long-id mode); then draw random
        ids from the active id space until one is free (object_manager.py:40-55)"""
        new_id = wanted
        for _ in range(5001):
            if new_id not in self.all_historical_object_ids and not (self.use_long_id and new_id < 256):
                return new_id
            new_id = np.random.randint(256, 256**3) if self.use_long_id else np.random.randint(1, 256)
        raise ValueError('We cannot find a new ID for this object. Perhaps you should use long ID?')

    def add_new_objects(self, objects: Union[List[ObjectInfo], ObjectInfo, List[int]]) -> Tuple[List[int], List[int]]:
        if not isinstance(objects, list):
            objects = [objects]
        tmp_ids, obj_ids = [], []
        for obj in objects:
            if isinstance(obj, int):
                obj = ObjectInfo(id=obj)
            new_obj = ObjectInfo(id=self._fresh_id(obj.id))
            new_obj.copy_meta_info(obj)
            tmp = len(self.obj_to_tmp_id) + 1
            self.obj_to_tmp_id[new_obj] = tmp
            self.tmp_id_to_obj[tmp] = new_obj
            self.all_historical_object_ids.add(new_obj.id)
            tmp_ids.append(tmp)
            obj_ids.append(new_obj.id)
        self._reindex()
        assert tmp_ids == sorted(tmp_ids), 'tmp id assignment bugged'
        return tmp_ids, obj_ids

    def delete_object(self, obj_ids_to_remove: Union[int, List[int]]) -> None:
        if isinstance(obj_ids_to_remove, int):
            obj_ids_to_remove = [obj_ids_to_remove]
        survivors = [self.tmp_id_to_obj[t] for t in range(1, len(self.obj_to_tmp_id) + 1)
                     if self.tmp_id_to_obj[t].id not in obj_ids_to_remove]
        self.obj_to_tmp_id = {obj: i + 1 for i, obj in enumerate(survivors)}
        self.tmp_id_to_obj = {i + 1: obj for i, obj in enumerate(survivors)}
        self._reindex()

    def purge_inactive_objects(self, max_missed_detection_count: int) -> Tuple[bool, List[int], List[int]]:
        """drop objects missed more than the allowed number of detections; returns
        (anything purged, tmp ids kept (old numbering), object ids kept)"""
        gone = [o.id for o in self.obj_to_tmp_id if o.poke_count > max_missed_detection_count]
        tmp_keep = [t for o, t in self.obj_to_tmp_id.items() if o.poke_count <= max_missed_detection_count]
        obj_keep = [o.id for o in self.obj_to_tmp_id if o.poke_count <= max_missed_detection_count]
        if gone:
            self.delete_object(gone)
        return len(gone) > 0, tmp_keep, obj_keep

    def _tmp_to_obj_table(self, device) -> torch.Tensor:
        table = [0] * (max(self.tmp_id_to_obj, default=0) + 1)
        for tmp_id, obj in self.tmp_id_to_obj.items():
            table[tmp_id] = int(obj.id)
        return torch.tensor(table, dtype=torch.int64, device=device)

    def prob_to_obj_cls(self, prob: torch.Tensor, size=None) -> torch.Tensor:
        """(no+1)*H*W probabilities of `DEVAInferenceCore.step` -> object-id index mask, optionally at
        another resolution: the drivers' F.interpolate(bilinear) -> argmax -> tmp_to_obj_cls tail
        (eval_vos.py:170-181, result_utils.py:98-102) as one kernel, so that H*W labels leave the
        device instead of (no+1)*H*W floats (not part of the reference's interface)"""
        from deva.hip import ops
        return ops.index_mask(prob.contiguous(), size, self._tmp_to_obj_table(prob.device))

    def tmp_to_obj_cls(self, mask) -> torch.Tensor:
        """tmp-id index mask -> object-id index mask"""
        if mask.is_cuda and mask.dtype == torch.int64 and self.tmp_id_to_obj:
            # one relabelling pass on the device instead of one masked assignment per object
            from deva.hip import ops
            return ops.lut_remap(mask.contiguous(), self._tmp_to_obj_table(mask.device))
        new_mask = torch.zeros_like(mask)  # host-side masks (e.g. in the result savers)
        for tmp_id, obj in self.tmp_id_to_obj.items():
            new_mask[mask == tmp_id] = obj.id
        return new_mask

    def get_tmp_to_obj_mapping(self) -> Dict[int, int]:
        return {obj.id: tmp_id for obj, tmp_id in self.tmp_id_to_obj.items()}

    def realize_dict(self, obj_dict: Dict[int, torch.Tensor]) -> torch.Tensor:
        """{object id: tensor} -> stacked tensor in tmp-id order"""
        stack, order = getattr(obj_dict, 'stack', None), getattr(obj_dict, 'order', None)
        if stack is not None and order == [obj.id for obj in self.tmp_id_to_obj.values()]:
            return stack  # MemoryManager.match_memory already produced the rows in this order: no copy
        rows = []
        for obj in self.tmp_id_to_obj.values():
            if obj.id not in obj_dict:
                raise NotImplementedError
            rows.append(obj_dict[obj.id])
        return torch.stack(rows, dim=0)

    def make_one_hot(self, cls_mask: torch.Tensor) -> torch.Tensor:
        planes = [cls_mask == obj.id for obj in self.tmp_id_to_obj.values()]
        if not planes:
            return torch.zeros((0, *cls_mask.shape), dtype=torch.bool, device=cls_mask.device)
        return torch.stack(planes, dim=0)

    def get_current_segments_info(self) -> List[Dict]:
        return [{'category_id': obj.vote_category_id(), 'id': int(obj.id), 'score': obj.vote_score()}
                for obj in self.obj_to_tmp_id]

    @property
    def all_obj_ids(self) -> List[int]:
        return [k.id for k in self.obj_to_tmp_id]

    @property
    def num_obj(self) -> int:
        return len(self.obj_to_tmp_id)

    def has_all(self, objects: List[int]) -> bool:
        # NB the reference compares ints against ObjectInfo keys, which raises AttributeError for an
        # id that is already registered (object_manager.py:161-165 + object_info.py:58); asking
        # about registered ids is answered here instead of raising.
        return all(o in self.obj_id_to_obj for o in objects)

    def find_object_by_id(self, obj_id) -> ObjectInfo:
        return self.obj_id_to_obj[obj_id]
