"""Merging propagated segments with image-level detections (interface of
deva/inference/segment_merging.py:89-143, Section 3.2.2 of the DEVA paper).

The reference evaluates `_get_iou` on boolean mask products for every (detection, object) pair and
syncs with the host for each comparison.  Here one kernel builds the joint label histogram of the
two index masks, the host derives every intersection / area / union from that small integer matrix
(one device->host copy per detection frame) and takes the same greedy decisions in the same order,
and one kernel paints the merged result directly as one-hot planes.
"""
import warnings
from typing import List, Literal

import torch

from deva.hip import ops
from deva.inference.object_info import ObjectInfo
from deva.inference.object_manager import ObjectManager


def match_and_merge(our_mask: torch.Tensor, new_mask: torch.Tensor, object_manager: ObjectManager,
                    new_segments_info: List[ObjectInfo], mode: Literal['iou'] = 'iou',
                    max_num_objects: int = -1, incremental_mode: bool = False) -> torch.Tensor:
    """
    our_mask: H*W propagated mask in temporary ids (0 = background)
    new_mask: H*W detection mask in object ids (the ids of new_segments_info)
    Updates the object manager (new objects, merged category votes, poke counters) exactly like
    the reference and returns the merged segmentation as num_objects*H*W one-hot planes in
    temporary-id order (fp32; the reference returns bool and casts right after).
    incremental_mode: existing objects are only poked when they are not visible at all.
    """
    if mode.lower() != 'iou':
        raise NotImplementedError('Engulf mode is deprecated')
    our_mask = our_mask.long().contiguous()
    new_mask = new_mask.long().contiguous()

    if max_num_objects > 0 and len(object_manager.obj_to_tmp_id) + len(new_segments_info) > max_num_objects:
        warnings.warn('Number of objects exceeded maximum (--max_num_objects); discarding new objects')
        new_segments_info = []

    ours = list(object_manager.obj_to_tmp_id.items())  # (ObjectInfo, tmp id) as of now
    n_our, n_new = len(ours), len(new_segments_info)
    tmp_of = {id(obj): tmp for obj, tmp in ours}
    device = our_mask.device
    new_ids = torch.tensor([int(o.id) for o in new_segments_info], dtype=torch.int64, device=device)

    # every pairwise intersection and every area in one pass + one copy
    counts = ops.label_histogram(our_mask, new_mask, new_ids, n_our).cpu().tolist()
    our_area = [sum(row) for row in counts]                      # by tmp id
    new_area = [sum(counts[t][j] for t in range(n_our + 1)) for j in range(n_new)]
    col_of = {id(o): j for j, o in enumerate(new_segments_info)}

    our_order, our_label = [-1] * (n_our + 1), [0] * (n_our + 1)
    new_order, new_label = [-1] * n_new, [0] * n_new
    step = 0

    for isthing_status in (None, False, True):  # others / stuff / things are merged separately
        matched = {}      # our ObjectInfo (by identity) -> detection
        areas = []        # ((object, is_new), area) in insertion order

        for new_obj in new_segments_info:
            if new_obj.isthing != isthing_status:
                continue
            j = col_of[id(new_obj)]
            for our_obj in list(object_manager.obj_to_tmp_id):
                if our_obj.isthing != isthing_status or id(our_obj) in matched:
                    continue
                inter = counts[tmp_of[id(our_obj)]][j]
                if inter < 1e-3:
                    continue
                union = new_area[j] + our_area[tmp_of[id(our_obj)]] - inter
                if inter / union > 0.5:
                    matched[id(our_obj)] = new_obj
                    areas.append(((our_obj, False), union))
                    break
            else:
                areas.append(((new_obj, True), new_area[j]))

        for our_obj in list(object_manager.obj_to_tmp_id):
            if our_obj.isthing != isthing_status or id(our_obj) in matched:
                continue
            areas.append(((our_obj, False), our_area[tmp_of[id(our_obj)]]))

        # repaint from the largest area to the smallest (stable for equal areas)
        for (obj, is_new), _ in sorted(areas, key=lambda kv: kv[1], reverse=True):
            if is_new:
                _, obj_ids = object_manager.add_new_objects(obj)
                j = col_of[id(obj)]
                new_order[j], new_label[j] = step, int(obj_ids[0])
            else:
                t = tmp_of[id(obj)]
                our_order[t], our_label[t] = step, int(obj.id)
                if id(obj) in matched:
                    new_obj = matched[id(obj)]
                    j = col_of[id(new_obj)]
                    new_order[j], new_label[j] = step, int(obj.id)
                    obj.merge(new_obj)
                    obj.unpoke()
                elif incremental_mode:
                    if our_area[t] < 1:
                        obj.poke()
                    else:
                        obj.unpoke()
                else:
                    obj.poke()
            step += 1

    out_ids = torch.tensor([int(o.id) for o in object_manager.tmp_id_to_obj.values()], dtype=torch.int64,
                           device=device)
    as_i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)  # noqa: E731
    as_i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=device)  # noqa: E731
    return ops.merge_paint(our_mask, new_mask, new_ids, as_i32(our_order), as_i64(our_label),
                           as_i32(new_order), as_i64(new_label), out_ids)
